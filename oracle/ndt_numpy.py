"""Independent NumPy fp64 restatement of the NDT path, written from the equations
(Magnusson 2009 eqs 6.8-6.13, 6.17-6.21; More & Thuente 1994), NOT from ndt_oracle.c, so the
two can disagree (SURVEY.md 8c "How parity is established without PCL").

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/ndt_oracle.h).
Sized for small cases (python loops over voxels / line-search steps).
"""
import math

import numpy as np

F = np.float32


def gauss_constants(resolution, outlier_ratio=0.55):
    """eq 6.8: c1, c2 mix a Gaussian with a uniform outlier density over one voxel."""
    res = float(F(resolution))
    c1 = 10.0 * (1.0 - outlier_ratio)
    c2 = outlier_ratio / res ** 3
    d3 = -math.log(c2)
    d1 = -math.log(c1 + c2) - d3
    d2 = -2.0 * math.log((-math.log(c1 * math.exp(-0.5) + c2) - d3) / d1)
    return d1, d2


class Cells:
    """Voxel normal distributions (a2).  Dense grid -> compact table in ascending index order."""

    def __init__(self, map_xy, resolution, min_pts=6, eig_mult=0.01, unbiased=False,
                 init_identity=True):
        # defaults = PCL 1.10: Leaf() starts the sum of products at the identity, (n-1)/n normalisation
        xy = np.ascontiguousarray(map_xy, dtype=F)
        self.res = F(resolution)
        self.inv = F(1.0) / self.res
        vx = np.floor(xy[:, 0] * self.inv).astype(np.int64)
        vy = np.floor(xy[:, 1] * self.inv).astype(np.int64)
        self.min_b = (int(math.floor(float(xy[:, 0].min() * self.inv))),
                      int(math.floor(float(xy[:, 1].min() * self.inv))))
        max_b = (int(math.floor(float(xy[:, 0].max() * self.inv))),
                 int(math.floor(float(xy[:, 1].max() * self.inv))))
        self.div = (max_b[0] - self.min_b[0] + 1, max_b[1] - self.min_b[1] + 1)
        key = (vy - self.min_b[1]) * self.div[0] + (vx - self.min_b[0])
        order = np.argsort(key, kind="stable")
        skey = key[order]
        bounds = np.flatnonzero(np.diff(skey)) + 1
        starts = np.concatenate([[0], bounds]); ends = np.concatenate([bounds, [len(skey)]])
        idx, cent, mean, icov, npts = [], [], [], [], []
        for s, e in zip(starts, ends):
            n = e - s
            if n < min_pts:
                continue
            pts32 = xy[order[s:e]]
            c32 = np.cumsum(pts32, axis=0, dtype=F)[-1] / F(n)      # sequential float32 sum
            p = pts32.astype(np.float64)
            mu = p.sum(axis=0) / n
            S = p.T @ p
            if init_identity:
                S = S + np.eye(2)
            if unbiased:
                cov = (S - n * np.outer(mu, mu)) / (n - 1.0)
                czz = (1.0 if init_identity else 0.0) / (n - 1.0)
            else:
                cov = (S / n - np.outer(mu, mu)) * ((n - 1.0) / n)
                czz = (1.0 if init_identity else 0.0) / n * ((n - 1.0) / n)
            cov3 = np.zeros((3, 3)); cov3[:2, :2] = cov; cov3[2, 2] = czz
            w, V = np.linalg.eigh(cov3)
            ok = not (w[0] < 0 or w[1] < 0 or w[2] <= 0)
            ic = np.zeros(3)
            if ok:
                thr = eig_mult * w[2]
                if w[0] < thr:
                    w = w.copy(); w[0] = thr
                    if w[1] < thr:
                        w[1] = thr
                    cov3 = V @ np.diag(w) @ np.linalg.inv(V)
                ic3 = np.linalg.inv(cov3[:2, :2])
                ic = np.array([ic3[0, 0], 0.5 * (ic3[0, 1] + ic3[1, 0]), ic3[1, 1]])
            idx.append(int(skey[s])); cent.append(c32); mean.append(mu); icov.append(ic)
            npts.append(n if ok else -n)
        self.idx = np.array(idx, dtype=np.int64)
        self.cent = np.array(cent, dtype=F).reshape(-1, 2)
        self.mean = np.array(mean).reshape(-1, 2)
        self.icov = np.array(icov).reshape(-1, 3)
        self.npts = np.array(npts, dtype=np.int64)
        self.lookup = {int(k): i for i, k in enumerate(self.idx)}
        self.r2 = F(float(self.res) * float(self.res))

    def neighbours(self, pts32, inclusive=False):
        """(point index, cell index) pairs with ||x' - centroid||^2 < r^2 in float32."""
        pi, ci = [], []
        vx = np.floor(pts32[:, 0] * self.inv).astype(np.int64) - self.min_b[0]
        vy = np.floor(pts32[:, 1] * self.inv).astype(np.int64) - self.min_b[1]
        for i in range(len(pts32)):
            if not (np.isfinite(pts32[i, 0]) and np.isfinite(pts32[i, 1])):
                continue
            for dy in (-1, 0, 1):
                yy = vy[i] + dy
                if yy < 0 or yy >= self.div[1]:
                    continue
                for dx in (-1, 0, 1):
                    xx = vx[i] + dx
                    if xx < 0 or xx >= self.div[0]:
                        continue
                    c = self.lookup.get(int(yy * self.div[0] + xx))
                    if c is None:
                        continue
                    ex = pts32[i, 0] - self.cent[c, 0]; ey = pts32[i, 1] - self.cent[c, 1]
                    d = F(F(ex * ex) + F(ey * ey))
                    if (d <= self.r2) if inclusive else (d < self.r2):
                        pi.append(i); ci.append(c)
        return np.array(pi, dtype=np.int64), np.array(ci, dtype=np.int64)


# std::cos / std::sin on a float (Eigen's AngleAxisf): this machine's libm cosf / sinf -- the default preset's libm_f32 = 1 -- or
# the correctly rounded value (libm_f32 = 0: the PCL <= 1.8 preset's model)
LIBM_F32 = True
_libm = None


def cos_sin_f32(yaw):
    global _libm
    if not LIBM_F32:
        return F(math.cos(float(yaw))), F(math.sin(float(yaw)))
    if _libm is None:
        import ctypes
        import ctypes.util
        _libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        for f in (_libm.cosf, _libm.sinf):
            f.restype = ctypes.c_float
            f.argtypes = [ctypes.c_float]
    return F(_libm.cosf(float(yaw))), F(_libm.sinf(float(yaw)))


def _libm_f32():
    cos_sin_f32(F(0.0))
    for name, nargs in (("atan2f", 2),):
        f = getattr(_libm, name)
        if f.restype is not __import__("ctypes").c_float:
            import ctypes
            f.restype = ctypes.c_float
            f.argtypes = [ctypes.c_float] * nargs
    return _libm


def eigen_init_yaw(c, s):
    """The initial yaw as computeTransformation's prologue gets it: Affine3f.rotation().eulerAngles(0, 1, 2)[2] for the guess
    matrix of a z rotation.  rotation() = U V^T of a float32 two-sided Jacobi SVD of the linear part (Eigen 3.3.90:
    Geometry/Transform.h:1088-1121, SVD/JacobiSVD.h:663-790, misc/RealSvd2x2.h, Jacobi/Jacobi.h); every operation below is a
    float32 operation in Eigen's order (3-term sums as a0 + (a1 + a2)); atan2f / sinf / cosf are this machine's libm."""
    lm = _libm_f32()
    one, zero = F(1.0), F(0.0)
    m22 = F(F(one - c) + c)
    W = np.array([[c, -s, zero], [s, c, zero], [zero, zero, m22]], dtype=F)
    U = np.eye(3, dtype=F); V = np.eye(3, dtype=F)
    scale = F(np.abs(W).max())
    if scale == 0:
        scale = one
    W = (W / scale).astype(F)
    eps2, tiny = F(2.0) * np.finfo(F).eps, np.finfo(F).tiny
    max_diag = F(max(abs(W[0, 0]), abs(W[1, 1]), abs(W[2, 2])))

    def rot(x, y, cc, ss):                       # x <- c x + s y ; y <- -s x + c y   (element-wise float32)
        if cc == one and ss == zero:
            return x, y
        return (F(cc) * x + F(ss) * y).astype(F), (F(-ss) * x + F(cc) * y).astype(F)

    for _ in range(64):
        finished = True
        for pp in (1, 2):
            for q in range(pp):
                thr = max(tiny, F(eps2 * max_diag))
                if not (abs(W[pp, q]) > thr or abs(W[q, pp]) > thr):
                    continue
                finished = False
                m = np.array([[W[pp, pp], W[pp, q]], [W[q, pp], W[q, q]]], dtype=F)
                t = F(m[0, 0] + m[1, 1]); d = F(m[1, 0] - m[0, 1])
                if abs(d) < tiny:
                    r1c, r1s = one, zero
                else:
                    u = F(t / d); tmp = F(np.sqrt(F(one + F(u * u))))
                    r1s = F(one / tmp); r1c = F(u / tmp)
                m[0], m[1] = rot(m[0].copy(), m[1].copy(), r1c, r1s)
                deno = F(F(2.0) * abs(m[0, 1]))
                if deno < tiny:
                    jrc, jrs = one, zero
                else:
                    tau = F(F(m[0, 0] - m[1, 1]) / deno); w = F(np.sqrt(F(F(tau * tau) + one)))
                    tt = F(one / F(tau + w)) if tau > 0 else F(one / F(tau - w))
                    sign_t = one if tt > 0 else F(-1.0)
                    n = F(one / F(np.sqrt(F(F(tt * tt) + one))))
                    jrs = F(F(F(F(-sign_t) * F(m[0, 1] / abs(m[0, 1]))) * abs(tt)) * n); jrc = n
                oc, os_ = jrc, F(-jrs)
                jlc = F(F(r1c * oc) - F(r1s * os_)); jls = F(F(r1c * os_) + F(r1s * oc))
                W[pp], W[q] = rot(W[pp].copy(), W[q].copy(), jlc, jls)
                U[:, pp], U[:, q] = rot(U[:, pp].copy(), U[:, q].copy(), jlc, jls)
                W[:, pp], W[:, q] = rot(W[:, pp].copy(), W[:, q].copy(), jrc, F(-jrs))
                V[:, pp], V[:, q] = rot(V[:, pp].copy(), V[:, q].copy(), jrc, F(-jrs))
                max_diag = F(max(max_diag, abs(W[pp, pp]), abs(W[q, q])))
        if finished:
            break
    sv = np.abs(np.diag(W)).astype(F)
    for i in range(3):
        if W[i, i] < 0:
            U[:, i] = -U[:, i]
    sv = (sv * scale).astype(F)
    for i in range(3):
        pos = i + int(np.argmax(sv[i:]))          # the first maximum, as maxCoeff
        if sv[pos] == 0:
            break
        if pos != i:
            sv[[i, pos]] = sv[[pos, i]]; U[:, [i, pos]] = U[:, [pos, i]]; V[:, [i, pos]] = V[:, [pos, i]]

    def mul_t(A, B):                              # A B^T, coefficient = a0 + (a1 + a2)
        out = np.zeros((3, 3), dtype=F)
        for i in range(3):
            for j in range(3):
                out[i, j] = F(F(A[i, 0] * B[j, 0]) + F(F(A[i, 1] * B[j, 1]) + F(A[i, 2] * B[j, 2])))
        return out
    P = mul_t(U, V)
    h = lambda a, b, cc: F(P[0, a] * F(F(P[1, b] * P[2, cc]) - F(P[1, cc] * P[2, b])))
    x = F(F(h(0, 1, 2) - h(1, 0, 2)) + h(2, 0, 1))
    Mx = U.copy(); Mx[:, 0] = (Mx[:, 0] / x).astype(F)
    R = mul_t(Mx, V)
    res0 = F(lm.atan2f(float(R[1, 2]), float(R[2, 2])))
    c1, s1 = cos_sin_f32(res0)
    num = F(F(s1 * R[2, 0]) - F(c1 * R[1, 0])); den = F(F(c1 * R[1, 1]) - F(s1 * R[2, 1]))
    return F(-F(lm.atan2f(float(num), float(den))))


def transform32(scan32, p, sse=True):
    """x' = R(yaw) x + t in float32 with the float32 matrix of the fp64 parameters (a4)."""
    yaw = F(p[2])
    c, s = cos_sin_f32(yaw)
    tx = F(p[0]); ty = F(p[1])
    x = scan32[:, 0].astype(F); y = scan32[:, 1].astype(F)
    if not sse:
        xo = F(F(c * x) + F(-s * y)) + tx
        yo = F(F(s * x) + F(c * y)) + ty
    else:
        xo = F(c * x) + F(F(-s * y) + tx)
        yo = F(s * x) + F(F(c * y) + ty)
    return np.stack([xo, yo], axis=1).astype(F), (c, s, tx, ty)


def score_grad_hess(cells, scan32, trans32, yaw, d1, d2, snap=10e-5, yaw_h=None, pairs=None):
    """eqs 6.9, 6.12, 6.13 restricted to (tx, ty, yaw)."""
    if yaw_h is None:
        yaw_h = yaw
    cz, sz = (1.0, 0.0) if abs(yaw) < snap else (math.cos(yaw), math.sin(yaw))
    ch, sh = (1.0, 0.0) if abs(yaw_h) < snap else (math.cos(yaw_h), math.sin(yaw_h))
    pi, ci = cells.neighbours(trans32) if pairs is None else pairs
    if len(pi) == 0:
        return 0.0, np.zeros(3), np.zeros((3, 3)), 0
    x = scan32[pi].astype(np.float64)
    q = trans32[pi].astype(np.float64) - cells.mean[ci]
    ic = cells.icov[ci]
    Sinv = np.empty((len(pi), 2, 2))
    Sinv[:, 0, 0] = ic[:, 0]; Sinv[:, 0, 1] = ic[:, 1]; Sinv[:, 1, 0] = ic[:, 1]; Sinv[:, 1, 1] = ic[:, 2]
    m = np.einsum("ni,nij,nj->n", q, Sinv, q)
    e = np.exp(-0.5 * d2 * m)
    keep = ~((d2 * e > 1) | (d2 * e < 0) | np.isnan(e))
    score = float(np.sum((-d1 * e)[keep]))
    # dT/dp: columns for tx, ty, yaw (eq 6.18 restricted)
    J = np.zeros((len(pi), 2, 3))
    J[:, 0, 0] = 1.0; J[:, 1, 1] = 1.0
    J[:, 0, 2] = -x[:, 0] * sz - x[:, 1] * cz
    J[:, 1, 2] = x[:, 0] * cz - x[:, 1] * sz
    # d2T/dyaw2 (eq 6.21 block f)
    h = np.zeros((len(pi), 2))
    h[:, 0] = -x[:, 0] * ch + x[:, 1] * sh
    h[:, 1] = -x[:, 0] * sh - x[:, 1] * ch
    w = (d1 * d2 * e) * keep
    qS = np.einsum("ni,nij->nj", q, Sinv)
    a = np.einsum("nj,njk->nk", qS, J)                      # q^T S^-1 J_i
    g = np.einsum("n,nk->k", w, a)
    JSJ = np.einsum("nik,nij,njl->nkl", J, Sinv, J)
    H = np.einsum("n,nkl->kl", w, -d2 * a[:, :, None] * a[:, None, :] + JSJ)
    H[2, 2] += float(np.sum(w * np.einsum("nj,nj->n", qS, h)))
    return score, g, H, len(pi)


def solve_newton(H, g):
    """delta = pinv(H) (-g) with the SVD rank rule of Eigen::JacobiSVD (6 * eps * s_max)."""
    U, s, Vt = np.linalg.svd(H)
    if not np.all(np.isfinite(s)):
        return np.full(3, np.nan)
    thr = 6 * np.finfo(float).eps * s.max() if s.size else 0.0
    inv = np.array([1.0 / v if v > thr else 0.0 for v in s])
    return Vt.T @ (inv * (U.T @ (-g)))


def _cubic_min(a1, f1, g1, a2, f2, g2):
    """Sun & Yuan 2.4.52/2.4.56: minimiser of the cubic through (a1,f1,g1), (a2,f2,g2)."""
    z = 3 * (f2 - f1) / (a2 - a1) - g2 - g1
    w = np.sqrt(z * z - g2 * g1)
    return a1 + (a2 - a1) * (w - g1 - z) / (g2 - g1 + 2 * w)


def mt_trial(*args):
    """IEEE semantics (0/0 = NaN, comparisons with NaN false) as in the compiled reference:
    a trial value clamped onto an interval end makes the interpolation formulas 0/0."""
    with np.errstate(all="ignore"):
        return float(_mt_trial(*[np.float64(v) for v in args]))


def _mt_trial(a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t):
    if f_t > f_l:                                            # case 1
        a_c = _cubic_min(a_l, f_l, g_l, a_t, f_t, g_t)
        a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t))
        return a_c if abs(a_c - a_l) < abs(a_q - a_l) else 0.5 * (a_q + a_c)
    if g_t * g_l < 0:                                        # case 2
        a_c = _cubic_min(a_l, f_l, g_l, a_t, f_t, g_t)
        a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l
        return a_c if abs(a_c - a_t) >= abs(a_s - a_t) else a_s
    if abs(g_t) <= abs(g_l):                                 # case 3
        a_c = _cubic_min(a_l, f_l, g_l, a_t, f_t, g_t)
        a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l
        nxt = a_c if abs(a_c - a_t) < abs(a_s - a_t) else a_s
        lim = a_t + 0.66 * (a_u - a_t)
        if a_t > a_l:
            return nxt if nxt < lim else lim
        return nxt if lim < nxt else lim
    return _cubic_min(a_u, f_u, g_u, a_t, f_t, g_t)          # case 4


def mt_update(I, a_t, f_t, g_t):
    """I = [a_l, f_l, g_l, a_u, f_u, g_u]; returns True when the interval has converged."""
    if f_t > I[1]:
        I[3:6] = [a_t, f_t, g_t]; return False
    if g_t * (I[0] - a_t) > 0:
        I[0:3] = [a_t, f_t, g_t]; return False
    if g_t * (I[0] - a_t) < 0:
        I[3:6] = I[0:3]; I[0:3] = [a_t, f_t, g_t]; return False
    return True


def yaw_from_T(T00, T10):
    T00 = F(T00); T10 = F(T10)
    if T00 > 0 and T10 != 0:
        return float(F(math.asin(float(T10))))
    if T00 < 0 and T10 > 0:
        return float(F(math.acos(float(T00))))
    return -float(F(math.acos(float(T00))))


def align(cells, scan32, init, resolution, step_size=0.1, trans_eps=0.01, max_iter=35,
          outlier_ratio=0.55, stale_h_ang=False, mu=1e-4, nu=0.9, mt_max=10, sse=True):
    """Newton + More-Thuente driver (Magnusson Algorithm 2 with PCL's step clamp)."""
    d1, d2 = gauss_constants(resolution, outlier_ratio)
    scan32 = np.ascontiguousarray(scan32, dtype=F)
    log = []
    trans, T = transform32(scan32, init, sse)
    yaw0 = eigen_init_yaw(F(T[0]), F(T[1])) if LIBM_F32 else F(math.atan2(float(T[1]), float(T[0])))
    p = np.array([float(T[2]), float(T[3]), float(yaw0)])
    state = {"yaw_h": p[2], "evals": 0}

    def derivs(pp, tr, refresh_h):
        if refresh_h or not stale_h_ang:
            state["yaw_h"] = pp[2]
        state["evals"] += 1
        return score_grad_hess(cells, scan32, tr, pp[2], d1, d2, yaw_h=state["yaw_h"])

    score, g, H, _ = derivs(p, trans, True)
    log.append((0.0, score))
    it, converged = 0, False
    while not converged:
        dp = solve_newton(H, g)
        nrm = float(np.linalg.norm(dp))
        if nrm == 0 or math.isnan(nrm):
            converged = not math.isnan(nrm)
            break
        d = dp / nrm
        phi0, dphi0 = -score, -float(g @ d)
        a = 0.0
        if dphi0 >= 0:
            if dphi0 == 0:
                a = None
            else:
                dphi0, d = -dphi0, -d
        if a is not None:
            I = [0.0, 0.0, dphi0 - mu * dphi0, 0.0, 0.0, dphi0 - mu * dphi0]
            open_iv, done, k = True, (step_size - trans_eps / 2) < 0, 0
            a = max(min(nrm, step_size), trans_eps / 2)
            xt = p + d * a
            trans, T = transform32(scan32, xt, sse)
            score, g, H, _ = derivs(xt, trans, True)
            log.append((a, score))
            phit, dphit = -score, -float(g @ d)
            psit, dpsit = phit - phi0 - mu * dphi0 * a, dphit - mu * dphi0
            while not done and k < mt_max and not (psit <= 0 and dphit <= -nu * dphi0):
                a = mt_trial(*I, a, psit, dpsit) if open_iv else mt_trial(*I, a, phit, dphit)
                a = a if not (step_size < a) else step_size
                a = a if not (a < trans_eps / 2) else trans_eps / 2
                xt = p + d * a
                trans, T = transform32(scan32, xt, sse)
                score, g, _, _ = derivs(xt, trans, False)
                log.append((a, score))
                phit, dphit = -score, -float(g @ d)
                psit, dpsit = phit - phi0 - mu * dphi0 * a, dphit - mu * dphi0
                if open_iv and psit <= 0 and dpsit >= 0:
                    open_iv = False
                    I[1] += phi0 - mu * dphi0 * I[0]; I[2] += mu * dphi0
                    I[4] += phi0 - mu * dphi0 * I[3]; I[5] += mu * dphi0
                done = mt_update(I, a, psit, dpsit) if open_iv else mt_update(I, a, phit, dphit)
                k += 1
            if k:
                _, _, H, _ = score_grad_hess(cells, scan32, trans, xt[2], d1, d2, yaw_h=state["yaw_h"])
                state["evals"] += 1
        else:
            a = 0.0
        p = p + d * a
        if it > max_iter or (it and abs(a) < trans_eps):
            converged = True
        it += 1
    return dict(p=p, T=T, score=score, H=H, iters=it, converged=converged, log=log,
                evals=state["evals"] + 1,
                pose=np.array([float(T[2]), float(T[3]), yaw_from_T(T[0], T[1])]))


def fitness(map_xy, scan32, T, sse=True):
    """Mean float32 squared distance to the nearest raw map point (a7), brute force."""
    c, s, tx, ty = T
    x = scan32[:, 0].astype(F); y = scan32[:, 1].astype(F)
    if not sse:
        qx = F(F(c * x) + F(-s * y)) + tx
        qy = F(F(s * x) + F(c * y)) + ty
    else:
        qx = F(c * x) + F(F(-s * y) + tx)
        qy = F(s * x) + F(F(c * y) + ty)
    m = np.ascontiguousarray(map_xy, dtype=F)
    tot = 0.0
    for i in range(len(qx)):
        ex = qx[i] - m[:, 0]; ey = qy[i] - m[:, 1]
        d = F(ex * ex) + F(ey * ey)
        tot += float(d.min())
    return tot / len(qx)
