// ndt_optimizer.hip.h -- row a6: Newton step + More-Thuente line search as a resumable state machine (also a3, a9 helpers).
// Part of libndt_mi355x.so: included by ndt_mi355x.hip inside its anonymous namespace (one translation
// unit; the order of the includes matters).  Not a standalone header.

// ------------------------------------------------------------------------------------------
// a6: Newton step + More-Thuente line search as a resumable state machine
// ------------------------------------------------------------------------------------------

// One Jacobi rotation annihilating a_pq of a symmetric 3x3 kept in scalars; r is the third index.
__device__ __forceinline__ void jacobi_rot(double &app, double &aqq, double &apq, double &arp, double &arq,
                                           double &v0p, double &v0q, double &v1p, double &v1q,
                                           double &v2p, double &v2q) {
  if (apq == 0.0) return;
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  const double app0 = app, aqq0 = aqq;
  app = app0 - t * apq; aqq = aqq0 + t * apq;
  const double arp0 = arp, arq0 = arq;
  arp = c * arp0 - s * arq0; arq = s * arp0 + c * arq0;
  apq = 0.0;
  double a, b;
  a = v0p; b = v0q; v0p = c * a - s * b; v0q = s * a + c * b;
  a = v1p; b = v1q; v1p = c * a - s * b; v1q = s * a + c * b;
  a = v2p; b = v2q; v2p = c * a - s * b; v2q = s * a + c * b;
}

// Symmetric 3x3 pseudo-inverse solve (cyclic Jacobi, all state in registers); stands in for
// JacobiSVD<6x6>::solve on the block-diagonal 6x6 (SURVEY.md 8a note).  Hs = xx xy xt yy yt tt.
__device__ __forceinline__ void solve3(const double Hs[6], double b0, double b1, double b2,
                                       double &x0, double &x1, double &x2) {
  double a00 = Hs[0], a01 = Hs[1], a02 = Hs[2], a11 = Hs[3], a12 = Hs[4], a22 = Hs[5];
  if (a00 != a00 || a01 != a01 || a02 != a02 || a11 != a11 || a12 != a12 || a22 != a22) {
    x0 = x1 = x2 = NAN; return;
  }
  {
    // well-conditioned case: adjugate / determinant
    const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const double det = a00 * c00 + a01 * c01 + a02 * c02;
    const double sc = fmax(fmax(fabs(a00), fabs(a11)), fmax(fabs(a22), fmax(fabs(a01), fmax(fabs(a02), fabs(a12)))));
    if (fabs(det) > 1e-9 * sc * sc * sc && fabs(det) <= DBL_MAX) {
      x0 = (c00 * b0 + c01 * b1 + c02 * b2) / det;
      x1 = (c01 * b0 + c11 * b1 + c12 * b2) / det;
      x2 = (c02 * b0 + c12 * b1 + c22 * b2) / det;
      return;
    }
  }
  // near-singular Hessian: pseudo-inverse through the eigen-decomposition
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(a01) + fabs(a02) + fabs(a12);
    if (off == 0.0) break;
    jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21);   // (p,q) = (0,1), r = 2
    jacobi_rot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22);   // (0,2), r = 1
    jacobi_rot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22);   // (1,2), r = 0
  }
  const double lmax = fmax(fabs(a00), fmax(fabs(a11), fabs(a22)));
  const double thr = lmax * (6.0 * DBL_EPSILON);
  x0 = x1 = x2 = 0.0;
  if (fabs(a00) > thr && !(fabs(a00) < DBL_MIN)) {
    const double pr = (v00 * b0 + v10 * b1 + v20 * b2) / a00;
    x0 += v00 * pr; x1 += v10 * pr; x2 += v20 * pr;
  }
  if (fabs(a11) > thr && !(fabs(a11) < DBL_MIN)) {
    const double pr = (v01 * b0 + v11 * b1 + v21 * b2) / a11;
    x0 += v01 * pr; x1 += v11 * pr; x2 += v21 * pr;
  }
  if (fabs(a22) > thr && !(fabs(a22) < DBL_MIN)) {
    const double pr = (v02 * b0 + v12 * b1 + v22 * b2) / a22;
    x0 += v02 * pr; x1 += v12 * pr; x2 += v22 * pr;
  }
}

// More-Thuente trial value, cases 1-4 (Sun & Yuan 2.4.2 / 2.4.5 / 2.4.52 / 2.4.56).
__device__ __forceinline__ double mt_trial(double a_l, double f_l, double g_l, double a_u, double f_u,
                                        double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    if (fabs(a_c - a_l) < fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    if (fabs(a_c - a_t) >= fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (fabs(g_t) <= fabs(g_l)) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double a_n = (fabs(a_c - a_t) < fabs(a_s - a_t)) ? a_c : a_s;
    double lim = a_t + 0.66 * (a_u - a_t);
    if (a_t > a_l) return (a_n < lim) ? a_n : lim;
    return (lim < a_n) ? a_n : lim;
  } else {
    double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
    double w = sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
  }
}

__device__ __forceinline__ int mt_update(AlignState &S, double a_t, double f_t, double g_t) {
  if (f_t > S.f_l) { S.a_u = a_t; S.f_u = f_t; S.g_u = g_t; return 0; }
  if (g_t * (S.a_l - a_t) > 0) { S.a_l = a_t; S.f_l = f_t; S.g_l = g_t; return 0; }
  if (g_t * (S.a_l - a_t) < 0) {
    S.a_u = S.a_l; S.f_u = S.f_l; S.g_u = S.g_l;
    S.a_l = a_t; S.f_l = f_t; S.g_l = g_t; return 0;
  }
  return 1;
}

__device__ __forceinline__ void set_trial(AlignState &S, const OptParams &P, bool refresh_h) {
  S.xt[0] = S.p[0] + S.dir[0] * S.a_t;
  S.xt[1] = S.p[1] + S.dir[1] * S.a_t;
  S.xt[2] = S.p[2] + S.dir[2] * S.a_t;
  S.need_tf = 1 | ((refresh_h || !P.stale_h_ang) ? 2 : 0);      // -> trial_transforms
}

// The two sincos of a new trial -- the float32 matrix (of the float32 yaw) and the fp64 angle terms of J_E / H_E -- are
// a few hundred dependent instructions each on the one lane that runs the optimiser, behind every pass with the whole
// workgroup waiting: lanes 0 and 1 of the wave compute them side by side (same code, different argument).
__device__ __noinline__ void trial_transforms(AlignState &S, const OptParams &P, int lane, int need) {
  const double yaw = S.xt[2];
  if (P.libm_f32) {
    // the float32 matrix entries as glibc's cosf / sinf give them -- lane 0 the cosine, lane 1 the sine, one code path --
    // then the fp64 angle terms (both lanes walk the short sincos together, lane 1 keeps them)
    const float v = sincosf_glibc((float)yaw, lane == 0 ? 1 : 0);
    double sn, cs;
    sincos_small(yaw, sn, cs);
    if (lane == 0) {
      S.T.c = v; S.T.tx = (float)S.xt[0]; S.T.ty = (float)S.xt[1];
      S.need_tf = 0;
    } else {
      S.T.s = v;
      if (fabs(yaw) < P.snap_thresh) { cs = 1.0; sn = 0.0; }
      S.cj = cs; S.sj = sn;
      if (need & 2) { S.ch = cs; S.sh = sn; }
    }
    return;
  }
  const double arg = lane == 0 ? (double)(float)yaw : yaw;
  double sn, cs;
  sincos_small(arg, sn, cs);
  if (lane == 0) {
    S.T.c = (float)cs; S.T.s = (float)sn; S.T.tx = (float)S.xt[0]; S.T.ty = (float)S.xt[1];
    S.need_tf = 0;
  } else {
    if (fabs(yaw) < P.snap_thresh) { cs = 1.0; sn = 0.0; }
    S.cj = cs; S.sj = sn;
    if (need & 2) { S.ch = cs; S.sh = sn; }
  }
}

// Start (or finish) outer iterations until a derivative pass is needed or the match is done.
__device__ __noinline__ void begin_outer(AlignState &S, const OptParams &P) {
  for (int guard = 0; guard < 1 << 20; ++guard) {   // every turn either asks for a pass or counts an iteration
    double dp0, dp1, dp2;
    solve3(S.H, -S.g[0], -S.g[1], -S.g[2], dp0, dp1, dp2);
    double nrm = sqrt(dp0 * dp0 + dp1 * dp1 + dp2 * dp2);
    if (nrm == 0 || nrm != nrm) { S.converged = (nrm == nrm); S.phase = PH_DONE; return; }
    S.dir[0] = dp0 / nrm; S.dir[1] = dp1 / nrm; S.dir[2] = dp2 / nrm;
    S.phi0 = -S.score;
    S.dphi0 = -(S.g[0] * S.dir[0] + S.g[1] * S.dir[1] + S.g[2] * S.dir[2]);
    double a = 0.0;
    bool need_eval = true;
    if (S.dphi0 >= 0) {
      if (S.dphi0 == 0) need_eval = false;
      else { S.dphi0 *= -1; S.dir[0] *= -1; S.dir[1] *= -1; S.dir[2] *= -1; }
    }
    if (need_eval) {
      S.step_iterations = 0;
      S.a_l = 0; S.a_u = 0;
      S.f_l = S.phi0 - S.phi0 - P.mt_mu * S.dphi0 * S.a_l;
      S.g_l = S.dphi0 - P.mt_mu * S.dphi0;
      S.f_u = S.phi0 - S.phi0 - P.mt_mu * S.dphi0 * S.a_u;
      S.g_u = S.dphi0 - P.mt_mu * S.dphi0;
      S.interval_converged = (P.step_size - P.trans_eps / 2) < 0;
      S.open_interval = 1;
      double a_t = nrm;
      a_t = (P.step_size < a_t) ? P.step_size : a_t;
      a_t = (a_t < P.trans_eps / 2) ? P.trans_eps / 2 : a_t;
      S.a_t = a_t;
      set_trial(S, P, true);
      S.phase = PH_LS_FIRST;
      return;
    }
    // zero directional derivative: step length 0, parameters unchanged
    int over = P.conv_ge ? (S.iters >= P.max_iter) : (S.iters > P.max_iter);
    bool conv = over || (S.iters && (fabs(a) < P.trans_eps));
    S.iters++;
    if (conv) { S.converged = 1; S.phase = PH_DONE; return; }
  }
}

// Consume one derivative pass (score, gradient, Hessian at the current trial transform).  Three leaf functions, called
// one after the other by lane 0 of the owning workgroup: a function that calls another keeps its live values in
// callee-saved registers, which it has to save and reload through scratch memory on every call -- half a microsecond
// on the critical path of every pass.  Leaves that stay inside the caller-saved registers touch no scratch at all.
__device__ __noinline__ void advance_totals(AlignState &S, const MapView &M, const double tot[kAcc], double *trace,
                                            int trace_cap, int *trace_rows) {
  // (all loads first: S and tot are both LDS, and a store to S would otherwise fence the loads behind it)
  const double d1 = M.d1, w = d1 * M.d2;
  const double t0 = tot[0], t1 = tot[1], t2 = tot[2], t3 = tot[3], t4 = tot[4], t5 = tot[5], t6 = tot[6], t7 = tot[7],
               t8 = tot[8], t9 = tot[9], t10 = tot[10], pr = S.pairs;
  const int ev = S.evals, rev = S.ref_evals;
  S.score = -d1 * t0;
  S.g[0] = w * t1; S.g[1] = w * t2; S.g[2] = w * t3;
  S.H[0] = w * t4; S.H[1] = w * t5; S.H[2] = w * t6;
  S.H[3] = w * t7; S.H[4] = w * t8; S.H[5] = w * t9;
  S.pairs = pr + t10;
  S.evals = ev + 1; S.ref_evals = rev + 1;
  if (trace) {
    int row = *trace_rows;
    if (row < trace_cap) {
      double *t = trace + 8 * (size_t)row;
      const double *pp = (S.phase == PH_INIT) ? S.p : S.xt;
      t[0] = (S.phase == PH_INIT) ? 0.0 : S.a_t; t[1] = S.score;
      t[2] = S.g[0]; t[3] = S.g[1]; t[4] = S.g[2]; t[5] = pp[0]; t[6] = pp[1]; t[7] = pp[2];
    }
    *trace_rows = row + 1;
  }
}

// The line search's turn after a pass.  Returns true when an outer iteration has to be started (begin_outer).
__device__ __noinline__ bool advance_step(AlignState &S, const OptParams &P) {
  if (S.phase == PH_INIT) return true;

  const double mu = P.mt_mu, nu = P.mt_nu;
  double phi_t = -S.score;
  double d_phi_t = -(S.g[0] * S.dir[0] + S.g[1] * S.dir[1] + S.g[2] * S.dir[2]);
  double psi_t = phi_t - S.phi0 - mu * S.dphi0 * S.a_t;
  double d_psi_t = d_phi_t - mu * S.dphi0;
  if (S.phase == PH_LS_INNER) {
    if (S.open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      S.open_interval = 0;
      S.f_l = S.f_l + S.phi0 - mu * S.dphi0 * S.a_l; S.g_l = S.g_l + mu * S.dphi0;
      S.f_u = S.f_u + S.phi0 - mu * S.dphi0 * S.a_u; S.g_u = S.g_u + mu * S.dphi0;
    }
    if (S.open_interval) S.interval_converged = mt_update(S, S.a_t, psi_t, d_psi_t);
    else                 S.interval_converged = mt_update(S, S.a_t, phi_t, d_phi_t);
    S.step_iterations++;
  }
  bool more = !S.interval_converged && S.step_iterations < P.mt_max_iter &&
              !(psi_t <= 0 && d_phi_t <= -nu * S.dphi0);
  if (more) {
    double a_t;
    if (S.open_interval) a_t = mt_trial(S.a_l, S.f_l, S.g_l, S.a_u, S.f_u, S.g_u, S.a_t, psi_t, d_psi_t);
    else                 a_t = mt_trial(S.a_l, S.f_l, S.g_l, S.a_u, S.f_u, S.g_u, S.a_t, phi_t, d_phi_t);
    a_t = (P.step_size < a_t) ? P.step_size : a_t;
    a_t = (a_t < P.trans_eps / 2) ? P.trans_eps / 2 : a_t;
    S.a_t = a_t;
    set_trial(S, P, false);
    S.phase = PH_LS_INNER;
    return false;
  }
  // line search done.  The reference now runs a Hessian-only pass when the inner loop ran;
  // the Hessian of the last pass (same cloud, same angle terms) is that Hessian already.
  if (S.step_iterations) S.ref_evals++;
  const double a = S.a_t;
  S.p[0] += S.dir[0] * a; S.p[1] += S.dir[1] * a; S.p[2] += S.dir[2] * a;
  int over = P.conv_ge ? (S.iters >= P.max_iter) : (S.iters > P.max_iter);
  bool conv = over || (S.iters && (fabs(a) < P.trans_eps));
  S.iters++;
  if (conv) { S.converged = 1; S.phase = PH_DONE; return false; }
  return true;
}

// The pass the reference would run next is the pass it has just run: the line search asked for the SAME step length
// again.  More-Thuente's trial value is clamped from below to trans_eps / 2 (PCL: `a_t = std::max (a_t, step_min)`), and once
// a search is down there every further trial is that value -- the same x_t, the same float32 matrix, the same cloud,
// hence, operation for operation, the same score, gradient and Hessian -- until the search has used up its ten
// iterations: on the bench workload 29 of 256 matches end in eleven such passes (the 13-20-pass matches every launch
// waits for; tools/pass_counts.py, LOG R4.9).  The totals of the last pass ARE the totals of that pass: it is counted
// (ref_evals), logged, and not run.
__device__ __noinline__ void repeat_totals(AlignState &S, double *trace, int trace_cap, int *trace_rows) {
  S.ref_evals = S.ref_evals + 1;
  if (trace) {
    int row = *trace_rows;
    if (row < trace_cap) {
      double *t = trace + 8 * (size_t)row;
      t[0] = S.a_t; t[1] = S.score;
      t[2] = S.g[0]; t[3] = S.g[1]; t[4] = S.g[2]; t[5] = S.xt[0]; t[6] = S.xt[1]; t[7] = S.xt[2];
    }
    *trace_rows = row + 1;
  }
}

__device__ __forceinline__ void advance(AlignState &S, const OptParams &P, const MapView &M,
                                        const double tot[kAcc], double *trace, int trace_cap,
                                        int *trace_rows) {
  advance_totals(S, M, tot, trace, trace_cap, trace_rows);
  for (int turn = 0; turn < 64; ++turn) {            // (a line search has mt_max_iter turns at most)
    const double a_prev = S.a_t;
    const int ph_prev = S.phase;
    if (advance_step(S, P)) { begin_outer(S, P); return; }
#ifdef NDT_NO_REPEAT_SKIP
    return;
#endif
    // a new trial inside the same line search (same p, same direction) with the step length of the pass just consumed?
    if (S.phase != PH_LS_INNER || ph_prev == PH_INIT || !(S.a_t == a_prev)) return;
    repeat_totals(S, trace, trace_cap, trace_rows);
  }
}

// The optimiser's start in two pieces: the float32 matrix of the initial guess (init_matrix; per lane: 0 the cosine and the
// translation, 1 the sine -- same code, other argument, as trial_transforms does for every trial) and everything else
// (init_rest: Eigen's rotation().eulerAngles of that matrix -- a float32 two-sided Jacobi SVD, a couple of thousand dependent
// instructions -- and the fp64 angle terms).  The owner runs both on one lane at the head of its set-up (init_state).
__device__ __noinline__ void init_matrix(AlignState &S, const OptParams &P, const double init[3], int lane) {
  // init_guess = Translation3f * AngleAxisf (src/PoseEstimator.cpp:22-24)
  const float yaw = (float)init[2];
  if (P.libm_f32) {
    const float v = sincosf_glibc(yaw, lane == 0 ? 1 : 0);
    if (lane == 0) S.T.c = v; else S.T.s = v;
  } else {
    double sd, cd;
    sincos_small((double)yaw, sd, cd);
    if (lane == 0) S.T.c = (float)cd; else S.T.s = (float)sd;
  }
  if (lane == 0) { S.T.tx = (float)init[0]; S.T.ty = (float)init[1]; }
}

__device__ __noinline__ void init_rest(AlignState &S, const OptParams &P, double n_points) {
  S.iters = 0; S.evals = 0; S.ref_evals = 0; S.converged = 0; S.step_iterations = 0;
  S.open_interval = 1; S.interval_converged = 0; S.pairs = 0.0; S.n_points = n_points;
  const Tf32 T = S.T;
  // p0 = (translation, rotation().eulerAngles(0,1,2)) of the float matrix: (-0, 0, yaw) -- the yaw as Eigen and the platform's
  // atan2f compute it (libm_f32: ndt_libm_f32.hip.h) or modelled as atan2f(s, c) correctly rounded
  S.p[0] = (double)T.tx; S.p[1] = (double)T.ty;
  S.p[2] = P.libm_f32 ? (double)eigen_init_yaw(T.c, T.s) : (double)(float)atan2((double)T.s, (double)T.c);
  S.xt[0] = S.p[0]; S.xt[1] = S.p[1]; S.xt[2] = S.p[2];
  S.dir[0] = S.dir[1] = S.dir[2] = 0.0; S.a_t = 0.0;
  angle_cs(P.snap_thresh, S.p[2], S.cj, S.sj);
  S.ch = S.cj; S.sh = S.sj;
  S.score = 0.0;
  S.g[0] = S.g[1] = S.g[2] = 0.0;                  // (defined also in the record of a scan that never gets a pass: an empty one)
  S.H[0] = S.H[1] = S.H[2] = S.H[3] = S.H[4] = S.H[5] = 0.0;
  S.need_tf = 0;
  S.phase = PH_INIT;
}


// both on one lane (scans that do not take the register-resident set-up)
__device__ __forceinline__ void init_state(AlignState &S, const OptParams &P, const double init[3], double n_points) {
  init_matrix(S, P, init, 0);
  init_matrix(S, P, init, 1);
  init_rest(S, P, n_points);
}

// a9: src/PoseEstimator.cpp:31-35 on the float32 entries; asinf/acosf modelled as correctly rounded.
__device__ __noinline__ double yaw_from_T(float T00, float T10) {   // (once per match: kept out of the kernel body, whose loops would hoist its constants into spilled registers)
  if (T00 > 0 && T10 > 0) return (double)(float)asin((double)T10);
  if (T00 > 0 && T10 < 0) return (double)(float)asin((double)T10);
  if (T00 < 0 && T10 > 0) return (double)(float)acos((double)T00);
  return (double)(float)acos((double)T00) * (-1.0);
}
