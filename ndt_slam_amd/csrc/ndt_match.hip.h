// ndt_match.hip.h -- the match kernel (rows a3-a6, a8, a9; a7 follows in ndt_fitness.hip.h): LDS window, spatial sort, units, work sharing between workgroups.
// Part of libndt_mi355x.so: included by ndt_mi355x.hip inside its anonymous namespace (one translation
// unit; the order of the includes matters).  Not a standalone header.

// ------------------------------------------------------------------------------------------
// the match kernel
//
// One workgroup per CU.  Every scan has an OWNER workgroup that holds the optimiser state in LDS
// and runs the whole match on the device.  A derivative pass is cut into
// kUnits units of points; each unit is reduced on its own and the pass total is the sum of the
// unit totals in a fixed order, so the result does not depend on who computed which unit.
// A workgroup whose own scans are finished becomes a HELPER: it attaches to an unfinished scan,
// stages that scan's window in its own LDS, registers, and from then on computes its static share
// of the units of every pass the owner opens.  Matches differ widely in the number of passes they
// need (mean ~7, max ~20 on the bench workload), so without helpers most of the chip idles
// behind the slowest scans.
//
// Inter-workgroup hand-off (cdna_hip_programming.md Guideline 16): every shared word (epoch word, pose halves,
// unit totals, ready counter) is read and written ONLY with agent-scope relaxed atomics (sc1 loads /
// write-through stores); the per-pass words validate themselves (payload + epoch tag in one 64-bit word, see
// ScanCtl), so nothing in a pass is ordered by a flag; the one bulk hand-off (the owner's
// ordered scan copy, marked-cell bitmap and window geometry) uses plain stores, drained
// (s_waitcnt vmcnt(0)) + agent release fence on the owner and an agent acquire fence on the helper.  No workgroup ever waits for a
// workgroup that is not running: a helper is only counted in after it has registered, at which
// point it does nothing but poll the scan's epoch word; helpers themselves only poll.
// Every spin is bounded by a watchdog that raises the abort word.
// ------------------------------------------------------------------------------------------
constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / 64;
#ifndef NDT_KSUB
#define NDT_KSUB 4
#endif
constexpr int kSub = NDT_KSUB;               // a lane's points are cut into kSub runs -> kSub units per wave
constexpr int kUnits = kWaves * kSub;        // units per pass
#ifndef NDT_MAX_HELPERS_BUILD
#define NDT_MAX_HELPERS_BUILD 15
#endif
constexpr int kMaxHelpers = NDT_MAX_HELPERS_BUILD;   // helper workgroups per scan, hard limit (64 units: 4 each)
// Default (NDT_OPT_MAX_HELPERS): beyond three helpers a pass hardly gets shorter (one unit on one wave takes 7.5 us),
// and a workgroup that finds every unfinished scan at its limit leaves -- its CU goes to whatever is queued behind
// the launch (in the bench: the map build of the next step).  8 against 15: same kernel time, 1.6 % more matches/s.
constexpr int kDefaultHelpers = 8;
constexpr int kBatchHelpers = 2;           // launches with a scan for every workgroup (ndt_mi355x.hip: launch_align)
#ifndef NDT_IDLE_MAX
#define NDT_IDLE_MAX 800           // idle helper back-off: 4 us doubling up to 8 us (100 MHz ticks)
#endif
#ifndef NDT_XCD_BONUS
#define NDT_XCD_BONUS 6          // passes' worth of preference for scans owned on the helper's own XCD (0: off)
#endif
#ifndef NDT_NEED_SLOPE
#define NDT_NEED_SLOPE 10       // passes still to run per unit of (1 - score per point / best score per point of a finished scan)
#endif
#ifndef NDT_HELPER_PENALTY
#define NDT_HELPER_PENALTY 12    // passes a scan must be ahead by before it gets one more helper than another
#endif
#ifndef NDT_BASE_HELPERS
#define NDT_BASE_HELPERS 7
#endif
#ifndef NDT_POLL2
#define NDT_POLL2 0      // owner: two polls of a unit-total word in flight
#endif
#ifndef NDT_POLL2H
#define NDT_POLL2H 0     // helper: two polls of the epoch line in flight
#endif
#ifndef NDT_OWNER_LEAD
#define NDT_OWNER_LEAD 0
#endif
constexpr int kOwnerLead = NDT_OWNER_LEAD;   // units of a shared pass the owner computes on top of its round-robin share
                                             // (the helpers' share reaches it a hand-off latency after its own)
constexpr int kBaseHelpers = NDT_BASE_HELPERS;              // ... while more scans are unfinished than workgroups / 8
constexpr unsigned kEpochDone = 0xFFFFFFFFu;
#ifndef NDT_FIRST_PASS_WAIT
#define NDT_FIRST_PASS_WAIT 2500
#endif
constexpr unsigned long long kFirstPassWait = NDT_FIRST_PASS_WAIT;   // ticks of the 100 MHz wall clock (25 us)
constexpr unsigned long long kWatchTicks = 400000000ull;   // ~4 s of the 100 MHz wall clock

typedef unsigned long long u64;
typedef unsigned int u32;

// Phase timers of the match kernel exist only in diagnostic builds (-DNDT_DIAG): a dozen 64-bit running sums live
// across every loop of the kernel cost registers the pass loop needs.
#ifdef NDT_DIAG
constexpr bool kProf = true;
#else
constexpr bool kProf = false;
#endif
// diagnostic builds: thread 0 of the owner stamps the setup phases of a scan (ticks since the scan was taken)
// diagnostic builds: timeline of the first kProfPasses shared passes of every scan, 16 words each (absolute ticks):
// opened, owner's units done, collected, helpers counted in, then (seen, done) of the helpers of rank 0..5
constexpr int kProfPasses = 24;
constexpr int kProfTimeline = kProfPasses * 16;
#define NDT_STAMP(st_, t0_, k_) do { if (kProf && (st_) && threadIdx.x == 0) (st_)[k_] = wall_clock64() - (t0_); } while (0)

// Per-scan control block: two 128-byte lines, so that the words polled by the helpers (line 0, written by the owner
// only) never share a line with the counters the helpers modify (line 1).
//
// Every word that hands data from one workgroup to another inside a pass is SELF-VALIDATING: 32 bits of payload under
// a 32-bit epoch tag, written and read as one aligned 64-bit word (single-copy atomic).  A reader polls the words
// themselves until each carries the tag of the pass it waits for: no flag that "covers" other stores, hence no drain
// on the writer and no second, dependent load on the reader -- one one-way latency per direction and pass (round 1 and
// most of round 2: pose block, drain, epoch word | poll, pose load || totals, drain, arrival add | poll, totals load).
//   owner -> helpers: line 0 = the epoch word + the twelve 32-bit halves of the pose block of the pass;
//   helper -> owner : unit totals, 24 words per unit (the halves of its 12 fp64 sums), utot[b][unit][24].
// Tags are the scan's epoch counter (2, 3, ... one per shared pass); all tagged words are zero at kernel start (cleared
// by the last kernel of the previous launch, fitness_reduce_kernel, or the memset in front of a first launch).
//
// The epoch word describes the split of a pass: units [0, ubeg) are the owner's, units [ubeg, uend) go round-robin over
// the owner and the first `h` registered helpers -- participant k (0 = owner, k = helper rank + 1) computes the units
// ubeg + k + j*(h+1).  The assignment is static (no claim atomics: a same-address agent-scope
// read-modify-write costs ~0.1 us and 128 waves used to queue on it every pass); it is safe because a
// helper only counts once it has registered in `ready`, after which it does nothing but poll line 0.
constexpr int kPoseWords = 12;               // Tf32 (4 x float32) + cj, sj, ch, sh (4 x fp64) as 32-bit halves
constexpr int kUnitWords = 24;               // 12 fp64 sums of a unit as 32-bit halves
struct alignas(128) ScanCtl {
  u64 ticket;        // line 0: epoch << 32 | h << 16 | uend << 8 | ubeg.  epoch 0: not open; 1: open for joining; kEpochDone: finished
  u64 pose[kPoseWords];   //    epoch << 32 | half k of the pass's PassPose
  u64 pad0_[3];
  u32 helpers;       // line 1: helper workgroups attached; geometry published by the owner's release
  int region[6];
  u32 passes;        //         passes the owner has run so far (helpers go where most were needed)
  u32 ready;         //         helpers whose window is staged; rank = order of registration
  u32 use_sorted;    //         1: passes read the scan from the sorted scratch copy
  u32 claimed;       //         1: a workgroup owns this scan (compare-and-swap; see "claims" in the kernel)
  u32 owner_xcd;     //         1 + XCD of the owning workgroup (helpers of the same XCD share its L2)
  u32 spp;           //         float bits: score per point after the last pass (what is left to do shows in it: see the helper's choice)
  int pad2_[19];
};
static_assert(sizeof(ScanCtl) == 256, "ScanCtl is two 128-byte lines");

struct WsHeader { u32 done; u32 abort; u32 next; u32 best_spp; u32 pad[28]; };   // next: scans handed out beyond the first gridDim.x; best_spp: float bits, best score per point of a finished scan
static_assert(sizeof(WsHeader) == 128, "WsHeader");

#define NDT_RLX __ATOMIC_RELAXED
#define NDT_AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ u64 ld64(const u64 *p) { return __hip_atomic_load(p, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ u32 ld32(const u32 *p) { return __hip_atomic_load(p, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ void st64(u64 *p, u64 v) { __hip_atomic_store(p, v, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ void st32(u32 *p, u32 v) { __hip_atomic_store(p, v, NDT_RLX, NDT_AGENT); }
// reads through a memory-side read-modify-write: never served from a stale L2 line of this XCD
__device__ __forceinline__ u64 rd64_fresh(u64 *p) { return __hip_atomic_fetch_add(p, 0ull, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ u32 rd32_fresh(u32 *p) { return __hip_atomic_fetch_add(p, 0u, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// 16- / 8-byte stores that write through to memory (sc1: what an agent-scope atomic store is on gfx942 / gfx950), for bulk
// data another XCD will read: once they are drained no agent-scope release -- a write-back of the XCD's whole L2 -- is needed
__device__ __forceinline__ void st_wt_f4(float4 *p, float4 v) {
  const ndt_f4v x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void st_wt_f2(float2 *p, float2 v) {
  const ndt_f2v x = {v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}

// A workgroup barrier that orders LDS traffic only: __syncthreads() carries workgroup-scope fences over ALL memory, i.e. an
// s_waitcnt vmcnt(0) in front of the barrier -- every global load or store a wave still has in flight is waited for right there.
// The set-up phases keep loads in flight across their LDS barriers on purpose (the records' lines asked for early, the
// ordered copy's stores draining); their barriers fence the `local` address space alone: s_waitcnt lgkmcnt(0) + s_barrier.
__device__ __forceinline__ void sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_down(v, o); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_down(v, o); v = t > v ? t : v; }
  return v;
}

// pose block of the pass being computed (LDS copy)
struct PassPose { Tf32 T; double cj, sj, ch, sh; };
static_assert(offsetof(ScanCtl, pose) == 8 && sizeof(PassPose) == 4 * kPoseWords && offsetof(PassPose, cj) == 16,
              "line 0 of ScanCtl: the epoch word, then the 32-bit halves of PassPose in memory order");

struct Lds {
  AlignState S;
  PassPose PP;
  Region RG;
  int sbox[4];
  int swave[kWaves + 1];
  int sflag[4];
  double wpart[kUnits * 12];       // unit totals this workgroup computed in the open pass (set-up: marked-cell bitmap + ballots, 4 KiB)
  double wtmp[kWaves * 12];        // helper waves: the unit just computed, before it is published
  double tot[12];                  // pass totals
  unsigned long long own_mask;     // units of the open pass computed by this workgroup
  unsigned long long hpose[8];     // helper: pose block of the open epoch, staged by wave 0
  unsigned long long hword;        // helper: epoch word seen by wave 0
  int hrank;                       // helper: order of registration on its scan
  int jnext;                       // units of the open segment handed out so far
  int steal;                       // helper search: nearest scan nobody has claimed yet (its workgroup is not resident)
  unsigned diag[2];                // diagnostic: ticks of fill_window's first two phases
  int clipped;                     // owner: the scan's voxel bounding box did not fit the window
  // what a pass needs besides PP and RG, read by pass_units (a separate function: see there)
  MapView M;                       // copies of the kernel arguments: what the functions around the kernel body read (a
  OptParams P;                     //   by-value kernel argument whose address is taken is spilled to scratch by every lane)
  const float2 *pts;               // the scan as the passes read it (ordered scratch copy, or the input)
  int npts;
  double etab[64];
};

// The match kernel's LDS lives at namespace scope so that the routines around the kernel body (pass_units, which can be
// built as a function of its own) can reach it by name: the window pool (slot table + voxel records of one scan) and the control / scratch block.
__shared__ Lds g_L;
__shared__ uint4 g_pool[kPoolBytes / 16];

__device__ __forceinline__ Window window_of(const Region &R, const uint4 *pool) {
  Window W;
  W.R = R;
  W.slot = reinterpret_cast<const unsigned short *>(pool);
  W.ent = reinterpret_cast<const CellEntry *>(reinterpret_cast<const char *>(pool) +
                                              ((R.rw * R.rh * 2 + 15) / 16) * 16);
  return W;
}

// thread 0: window geometry from the bounding box of the scan's voxel coordinates (L.sbox) -> L.RG, L.clipped
__device__ __forceinline__ void region_from_bbox(const MapView &M, const Tf32 &T0, Lds &L) {
  {
    Region r = {0, 0, 0, 0, 0, 0};
    L.clipped = 0;
    if (L.sbox[0] <= L.sbox[2]) {
      // The bbox plus the slack, wherever it lies: cells outside the map's grid are simply empty (fill_window), and a window
      // that does not depend on the grid's extent keeps the order of the scan copy -- hence every sum, to the last bit --
      // independent of it (ndt_params::grid_margin widens the grid; round 4).  A bbox too large for the slot table is cut
      // down around the SENSOR's voxel (the first pose's translation: the scan surrounds it, whatever a stray far return
      // does to the bbox), kept inside the bbox; flagged NDT_FLAG_REGION_CLIPPED.
      long long x0 = (long long)L.sbox[0] - kRegionMargin, x1 = (long long)L.sbox[2] + kRegionMargin;
      long long y0 = (long long)L.sbox[1] - kRegionMargin, y1 = (long long)L.sbox[3] + kRegionMargin;
      long long w = x1 - x0 + 1, h = y1 - y0 + 1;
      if (w > 0 && h > 0) {
        if (w * h > kRegionCells) {
          L.clipped = 1;
          const long long w2 = w > 128 ? 128 : w;
          long long h2 = kRegionCells / w2; if (h2 > h) h2 = h;
          const float sx = fminf(fmaxf(floorf(T0.tx * M.inv_leaf), -1.0e9f), 1.0e9f);    // (NaN -> -1e9, as everywhere)
          const float sy = fminf(fmaxf(floorf(T0.ty * M.inv_leaf), -1.0e9f), 1.0e9f);
          long long cx0 = ((long long)(int)sx - M.min_bx) - w2 / 2, cy0 = ((long long)(int)sy - M.min_by) - h2 / 2;
          cx0 = cx0 < x0 ? x0 : (cx0 > x1 - w2 + 1 ? x1 - w2 + 1 : cx0);
          cy0 = cy0 < y0 ? y0 : (cy0 > y1 - h2 + 1 ? y1 - h2 + 1 : cy0);
          x0 = cx0; y0 = cy0; w = w2; h = h2;
        }
        const long long far = 1ll << 30;          // (a scan of nothing but far-away returns: keep the int arithmetic of the passes in range)
        x0 = x0 < -far ? -far : (x0 > far ? far : x0); y0 = y0 < -far ? -far : (y0 > far ? far : y0);
        r.x0 = (int)x0; r.y0 = (int)y0; r.rw = (int)w; r.rh = (int)h;
      }
    }
    const int slot_bytes = ((r.rw * r.rh * 2 + 15) / 16) * 16;
    int cap = (kPoolBytes - slot_bytes) / (int)sizeof(CellEntry) - 2;   // last two = sentinels
    r.cap = cap > 0xFFF0 ? 0xFFF0 : cap;
    L.RG = r;
  }
}

// Owner: bounding box of the scan's voxel coordinates at the first pose -> window geometry.
template <bool SSE>
__device__ __noinline__ void compute_region(const MapView &M, const Tf32 &T0, const float2 *__restrict__ scan,
                                               int n, Lds &L) {
  if (threadIdx.x == 0) { L.sbox[0] = INT_MAX; L.sbox[1] = INT_MAX; L.sbox[2] = INT_MIN; L.sbox[3] = INT_MIN; }
  __syncthreads();
  int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const float2 pt = scan[i];
    float xt, yt;
    tf_apply_t<SSE>(T0, pt.x, pt.y, xt, yt);
    if (!finite2(xt, yt)) continue;
    const float fx = fminf(fmaxf(floorf(xt * M.inv_leaf), -1.0e9f), 1.0e9f);
    const float fy = fminf(fmaxf(floorf(yt * M.inv_leaf), -1.0e9f), 1.0e9f);
    const int ix = (int)fx - M.min_bx, iy = (int)fy - M.min_by;
    mnx = ix < mnx ? ix : mnx; mxx = ix > mxx ? ix : mxx;
    mny = iy < mny ? iy : mny; mxy = iy > mxy ? iy : mxy;
  }
  mnx = wave_min_i(mnx); mny = wave_min_i(mny); mxx = wave_max_i(mxx); mxy = wave_max_i(mxy);
  if ((threadIdx.x & 63) == 0 && mnx <= mxx) {
    atomicMin(&L.sbox[0], mnx); atomicMin(&L.sbox[1], mny); atomicMax(&L.sbox[2], mxx); atomicMax(&L.sbox[3], mxy);
  }
  __syncthreads();
  if (threadIdx.x == 0) region_from_bbox(M, T0, L);
  __syncthreads();
}

// Owner and helpers: fill the slot table and the compact record table of window L.RG from the map
// and the marked-cell bitmap in L.wmap.  Slots are numbered in row-major order of the window, so
// the content depends only on the map, the geometry and the bitmap.  Slot values: < cap a resident
// record; cap = voxel outside the search set (centroid +inf); cap + 1 = occupied voxel without an
// LDS record (centroid -inf).  Sets L.RG.nspill = occupied voxels left without a record.
// Cells are walked 1024 at a time with consecutive lanes on consecutive cells (coalesced centroid
// and record reads); the row-major numbering comes from wave ballots kept in LDS.
//
// Two routines: fill_window_plan works out WHICH voxels get a record -- occupancy words, dilation of the marked cells, the
// prefix sums that number them -- and touches only the control block (L.wpart, L.wtmp, L.swave, L.sbox); fill_window_fetch
// writes the slot table and fetches the records into the pool.  (Round 5 tried the plan in the middle of the owner's ordering
// phases with the records' lines asked for at once, so that the fetch would find them in L2: no gain -- the fetch was
// bound by its own LDS instructions and address arithmetic, not by the loads, which arrive while the slot table is
// written; LOG R5.1.)
__device__ __forceinline__ int window_rounds(const Region &r) { return (r.rw * r.rh + kBlock - 1) / kBlock; }   // <= kRegionCells / kBlock = 16

__device__ __noinline__ void fill_window_plan(const MapView &M, Lds &L, u64 *stamps = nullptr, u64 t0s = 0) {
  const Region r = L.RG;
  const unsigned *wmap = reinterpret_cast<const unsigned *>(L.wpart);
  const u64 t_fill0 = kProf ? wall_clock64() : 0;
  const int ncell = r.rw * r.rh;
  const int rounds = window_rounds(r);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 *keepw = reinterpret_cast<u64 *>(L.wpart) + 256;        // [rounds][kWaves] ballots (wmap uses the first 2 KiB)
  u64 *occw = keepw + 256;
  int *base = reinterpret_cast<int *>(L.wtmp);                // [256] exclusive prefix of the kept counts
  // which window cells are in the map's search set: the window's occupancy bitmap in cell order (64 cells per
  // word = what a wave ballot over 64 consecutive cells would give), each 32-bit half assembled by one thread from
  // the map's occupancy words -- a run of window cells in one row is a run of bits of the map's bitmap.
  // (One load per CELL, 16 per thread, took 12 us here.)
  {
    constexpr int kWords32 = kRegionCells / 32;
    unsigned *occ32 = reinterpret_cast<unsigned *>(occw);
    const int rw1 = max(r.rw, 1);
    for (int i = threadIdx.x; i < kWords32; i += kBlock) {
      unsigned word = 0u;
      int c = 32 * i, filled = 0;
      int ly = c / rw1, lx = c - ly * rw1;
      for (int piece = 0; piece < 34 && filled < 32 && c < ncell; ++piece) {
        const int len = min(32 - filled, r.rw - lx);             // cells of row ly from lx on
        const int my = r.y0 + ly, a = r.x0 + lx;
        const int lo = max(a, 0), hi = min(a + len, M.div_x);
        if (my >= 0 && my < M.div_y && lo < hi) {
          const size_t g = (size_t)my * M.div_x + lo;
          const unsigned w0 = M.occ[g >> 5], w1 = M.occ[(g >> 5) + 1];    // (the bitmap has two spare words)
          const u64 two = ((u64)w1 << 32) | w0;
          const unsigned bits = (unsigned)(two >> (g & 31)) & (hi - lo >= 32 ? 0xFFFFFFFFu : ((1u << (hi - lo)) - 1u));
          word |= bits << (filled + (lo - a));
        }
        filled += len; c += len; lx = 0; ++ly;
      }
      occ32[i] = word;
    }
  }
  NDT_STAMP(stamps, t0s, 7);
  // marked cells dilated by two cells in x and y, on whole words: a voxel gets an LDS record when it
  // is in the search set and within two cells of a cell a scan point fell in.  (Rows are not word
  // aligned, so a mark in the first or last two columns of the window also reaches the end of the
  // neighbouring row: a few more records, nothing else.)
  unsigned *dx = reinterpret_cast<unsigned *>(keepw);       // 512 words, reused for the result
  constexpr int kWords = kRegionCells / 32;
  auto word_at = [&](const unsigned *a, int i) { return (i >= 0 && i < kWords) ? a[i] : 0u; };
  if (threadIdx.x < kWords) {
    const int i = threadIdx.x;
    const unsigned w = wmap[i], pv = word_at(wmap, i - 1), nx = word_at(wmap, i + 1);
    dx[i] = w | (w << 1) | (w << 2) | (w >> 1) | (w >> 2) | (pv >> 31) | (pv >> 30) | (nx << 31) | (nx << 30);
  }
  __syncthreads();
  unsigned kword = 0;
  if (threadIdx.x < kWords) {
    const int i = threadIdx.x;
    kword = dx[i];
#pragma unroll
    for (int m = 1; m <= 2; ++m) {
      const int sft = m * r.rw, q = sft >> 5, b = sft & 31;
      // bits moved towards higher cell numbers (from the row(s) above) and towards lower ones (below)
      kword |= (word_at(dx, i - q) << b) | (b ? (word_at(dx, i - q - 1) >> (32 - b)) : 0u);
      kword |= (word_at(dx, i + q) >> b) | (b ? (word_at(dx, i + q + 1) << (32 - b)) : 0u);
    }
    kword &= reinterpret_cast<const unsigned *>(occw)[i];
  }
  __syncthreads();
  if (threadIdx.x < kWords) dx[threadIdx.x] = kword;        // = keepw, two words per ballot word
  __syncthreads();
  if (kProf && threadIdx.x == 0) L.diag[0] = (unsigned)(wall_clock64() - t_fill0);
  NDT_STAMP(stamps, t0s, 8);
  // exclusive prefix of the kept counts over the rounds * kWaves ballot words (cell order)
  const int nword = rounds * kWaves;                           // <= 256
  if (threadIdx.x < 256) {
    const int mine = (int)threadIdx.x < nword ? __builtin_popcountll(keepw[threadIdx.x]) : 0;
    const int skip = (int)threadIdx.x < nword ? __builtin_popcountll(occw[threadIdx.x] & ~keepw[threadIdx.x]) : 0;
    int incl = mine, sk = skip;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o); if (lane >= o) incl += t;
      sk += __shfl_xor(sk, o);
    }
    base[threadIdx.x] = incl - mine;
    if (lane == 63) { L.swave[wave] = incl; L.sbox[wave] = sk; }
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    int add = 0;
    for (int w = 0; w < wave; ++w) add += L.swave[w];
    base[threadIdx.x] += add;
  }
  if (threadIdx.x == 0) {
    const int kept = L.swave[0] + L.swave[1] + L.swave[2] + L.swave[3];
    const int skipped = L.sbox[0] + L.sbox[1] + L.sbox[2] + L.sbox[3];
    L.RG.nspill = skipped + (kept > r.cap ? kept - r.cap : 0);
    if (kProf) L.diag[1] = (unsigned)(wall_clock64() - t_fill0);
  }
  __syncthreads();
  NDT_STAMP(stamps, t0s, 9);
}

constexpr int kRecPer = 4;
static_assert(kPoolBytes / (int)sizeof(CellEntry) <= kRecPer * kBlock, "a thread fetches at most kRecPer records");

// Round 5.  Until round 4: sixteen rounds of one cell per thread for the slot table (a 2-byte LDS store per cell: the LDS pipe
// takes one instruction per ~5 cycles whatever its width -- 3.3 us), and every record's cell found from its number by a binary
// search over the prefix sums + a search for the n-th set bit (eight dependent LDS reads and ~100 instructions per record:
// 2.2 us).  Now a thread takes EIGHT consecutive cells -- one byte of each ballot word, their eight slot numbers counted up in
// registers, one 16-byte store -- and leaves the padded-grid index of every cell that gets a record in the first word of that
// record's own place; the record-major pass (all of a thread's loads in flight together, as before) reads its cells from there.
__device__ __noinline__ void fill_window_fetch(const MapView &M, Lds &L, uint4 *pool, u64 *stamps = nullptr, u64 t0s = 0) {
  const Region r = L.RG;
  uint4 *slot8 = pool;                                   // the slot table, eight 16-bit entries per 16-byte word
  CellEntry *ent = reinterpret_cast<CellEntry *>(reinterpret_cast<char *>(pool) + ((r.rw * r.rh * 2 + 15) / 16) * 16);
  const int ncell = r.rw * r.rh;
  const u64 *keepw = reinterpret_cast<const u64 *>(L.wpart) + 256;
  const u64 *occw = keepw + 256;
  const int *base = reinterpret_cast<const int *>(L.wtmp);
  if (threadIdx.x == 0) {
    CellEntry z; z.cent = make_float2(INFINITY, INFINITY); z.mx = z.my = z.i00 = z.i01 = z.i11 = 0.0;
    ent[r.cap] = z;                             // voxels outside the search set
    z.cent = make_float2(-INFINITY, -INFINITY);
    ent[r.cap + 1] = z;                         // occupied voxels without an LDS record
  }
  const int rw1 = max(r.rw, 1);
  for (int g = threadIdx.x; 8 * g < ncell; g += kBlock) {               // at most kRegionCells / 8 / kBlock = 2 rounds
    const int w = g >> 3, sh = (g & 7) << 3;
    const u64 kw = keepw[w], ow = occw[w];                              // (bits of cells >= ncell are zero: fill_window_plan)
    const unsigned kb = (unsigned)(kw >> sh) & 0xFFu, ob = (unsigned)(ow >> sh) & 0xFFu;
    int next = base[w] + __builtin_popcountll(kw & ((1ull << sh) - 1ull));
    const int c0 = 8 * g;
    int ly = c0 / rw1, lx = c0 - ly * rw1;
    unsigned sl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned v = (unsigned)r.cap;
      if ((ob >> i) & 1u) {
        v = (unsigned)r.cap + 1u;
        if ((kb >> i) & 1u) {
          if (next < r.cap) {
            v = (unsigned)next;
            // this record's cell in the padded grid, left in the record's own place until the record arrives
            *reinterpret_cast<unsigned *>(ent + next) = (unsigned)((size_t)(r.y0 + ly + 2) * M.gw + (r.x0 + lx + 2));
          }
          ++next;
        }
      }
      sl[i] = v;
      if (++lx == r.rw) { lx = 0; ++ly; }
    }
    slot8[g] = make_uint4(sl[0] | (sl[1] << 16), sl[2] | (sl[3] << 16), sl[4] | (sl[5] << 16), sl[6] | (sl[7] << 16));
  }
  sync_lds();
  NDT_STAMP(stamps, t0s, 14);
  // records: thread t fetches records t, t + 1024, ... (at most kRecPer), ALL of its loads in flight together
  // (walking the cells round by round, four rounds of loads at a time, took 14 us: four dependent latencies)
  const int kept = min(L.swave[0] + L.swave[1] + L.swave[2] + L.swave[3], r.cap);
  int nx[kRecPer]; double2 ra[kRecPer], rb[kRecPer], rc[kRecPer];     // rc.y: the float32 centroid (map_build: write_voxel)
#pragma unroll
  for (int u = 0; u < kRecPer; ++u) {
    const int k = (int)threadIdx.x + u * kBlock;
    nx[u] = -1;
    if (k < kept) {
      const size_t pg = *reinterpret_cast<const unsigned *>(ent + k);
      const double *rec = M.rec + pg * 8;
      ra[u] = gld_d2(rec); rb[u] = gld_d2(rec + 2); rc[u] = gld_d2(rec + 4);
      nx[u] = k;
    }
  }
  NDT_STAMP(stamps, t0s, 15);
#pragma unroll
  for (int u = 0; u < kRecPer; ++u) {
    if (nx[u] >= 0) {
      const u64 cb = (u64)__double_as_longlong(rc[u].y);
      CellEntry E; E.cent = make_float2(__uint_as_float((unsigned)cb), __uint_as_float((unsigned)(cb >> 32)));
      E.mx = ra[u].x; E.my = ra[u].y; E.i00 = rb[u].x; E.i01 = rb[u].y; E.i11 = rc[u].x;
      ent[nx[u]] = E;
    }
  }
  __syncthreads();                                 // (all memory: whatever this workgroup stored before is drained here too -- the late publication relies on it)
}

__device__ __forceinline__ void fill_window(const MapView &M, Lds &L, uint4 *pool, u64 *stamps = nullptr, u64 t0s = 0) {
  fill_window_plan(M, L, stamps, t0s);
  fill_window_fetch(M, L, pool, stamps, t0s);
}

// Owner: spatial order of the scan.  The points are sorted by the window cell they fall in at the
// first pose (row-major cell order, input order kept inside a cell) and written to the scratch copy
// every pass reads.  The 64 lanes of a wave then always work on neighbouring points -- a rigid
// transform keeps neighbours together, so this holds at every later pose too -- which means: equal
// in-radius voxel counts (the pair loop runs max-over-lanes times), LDS probes that hit the same few
// slots and records (broadcast instead of bank conflicts), and in the fitness kernel bucket loads that
// share cache lines.  The cell histogram also yields the marked-cell bitmap (L.wmap) that
// fill_window and the helpers use.  Uses the LDS pool as scratch (before the window is staged).
// Returns false (bitmap still produced, scratch copy not written) when the scan is too large for it.
constexpr int kSortMax = 20000;             // LDS room for one word per point; point numbers < 2^15
constexpr int kSortRegs = 10 * kBlock;      // order_scan_regs: the scan in registers, two words per point in LDS beside the counters
static_assert(((kRegionCells + 1 + 3) & ~3) * 4 + 2 * kSortRegs * 4 <= kPoolBytes, "counters + entries + places in the LDS pool");
template <bool SSE>
__device__ __noinline__ bool sort_points(const MapView &M, const Tf32 &T0, const float2 *__restrict__ scan,
                                            int n, Lds &L, uint4 *pool, float2 *__restrict__ sp,
                                            u64 *stamps = nullptr, u64 t0s = 0) {
  const Region r = L.RG;
  const int ncell = r.rw * r.rh;
  unsigned *wmap = reinterpret_cast<unsigned *>(L.wpart);
  unsigned *hist = reinterpret_cast<unsigned *>(pool);                 // ncell + 1 counters (last: outside the window)
  unsigned *idx = hist + ((ncell + 1 + 3) & ~3);
  for (int i = threadIdx.x; i <= ncell; i += kBlock) hist[i] = 0u;
  for (int i = threadIdx.x; i < kRegionCells / 32; i += kBlock) wmap[i] = 0u;
  __syncthreads();
  auto key_of = [&](float2 pt) {
    float xt, yt;
    tf_apply_t<SSE>(T0, pt.x, pt.y, xt, yt);
    if (!finite2(xt, yt)) return ncell;
    const float fx = fminf(fmaxf(floorf(xt * M.inv_leaf), -1.0e9f), 1.0e9f);
    const float fy = fminf(fmaxf(floorf(yt * M.inv_leaf), -1.0e9f), 1.0e9f);
    const int lx = (int)fx - M.min_bx - r.x0, ly = (int)fy - M.min_by - r.y0;
    if (lx < 0 || lx >= r.rw || ly < 0 || ly >= r.rh) return ncell;
    return ly * r.rw + lx;
  };
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * kBlock) {      // four loads in flight
    float2 pt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pt[u] = scan[min(i0 + u * kBlock, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i0 + u * kBlock < n) atomicAdd(&hist[key_of(pt[u])], 1u);
  }
  __syncthreads();
  NDT_STAMP(stamps, t0s, 2);
  // marked-cell bitmap: 64 consecutive cells per wave ballot (bank-conflict-free reads)
  {
    u64 *wmap64 = reinterpret_cast<u64 *>(wmap);
    for (int c0 = (int)(threadIdx.x & ~63u); c0 < ncell; c0 += kBlock) {
      const int c = c0 + (int)(threadIdx.x & 63u);
      const u64 bits = __ballot(c < ncell && hist[c] != 0u);
      if ((threadIdx.x & 63u) == 0u) wmap64[c0 >> 6] = bits;
    }
  }
  const bool do_sort = sp != nullptr && n <= kSortMax;
  if (!do_sort) { __syncthreads(); return false; }
  // exclusive scan of the ncell + 1 counters
  const int per = (ncell + 1 + kBlock - 1) / kBlock;
  const int c0 = min((int)threadIdx.x * per, ncell + 1), c1 = min(c0 + per, ncell + 1);
  unsigned mine = 0;
  for (int c = c0; c < c1; ++c) mine += hist[c];
  unsigned incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if ((int)(threadIdx.x & 63) >= o) incl += t; }
  if ((threadIdx.x & 63) == 63) L.swave[threadIdx.x >> 6] = (int)incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < kWaves; ++w) { const int t = L.swave[w]; L.swave[w] = run; run += t; }
  }
  __syncthreads();
  {
    unsigned run = (unsigned)L.swave[threadIdx.x >> 6] + incl - mine;
    for (int c = c0; c < c1; ++c) { const unsigned t = hist[c]; hist[c] = run; run += t; }
  }
  __syncthreads();
  NDT_STAMP(stamps, t0s, 3);
  // scatter (cell, point number) packed in one word; afterwards hist[c] = end of cell c
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * kBlock) {
    float2 pt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pt[u] = scan[min(i0 + u * kBlock, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u * kBlock >= n) break;
      const int key = key_of(pt[u]);
      idx[atomicAdd(&hist[key], 1u)] = ((unsigned)key << 15) | (unsigned)(i0 + u * kBlock);
    }
  }
  __syncthreads();
  NDT_STAMP(stamps, t0s, 4);
  // input order inside a cell (the atomics above arrive in any order): every entry finds its rank among
  // the entries of its cell -- neighbouring lanes read the same short segment -- and its point goes
  // straight to that place of the scratch copy
  for (int p0 = threadIdx.x; p0 < n; p0 += 4 * kBlock) {
    int dstpos[4]; float2 pt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = p0 + u * kBlock;
      dstpos[u] = -1;
      if (pp < n) {
        const unsigned v = idx[pp];
        const int key = (int)(v >> 15);
        const int s0 = key ? (int)hist[key - 1] : 0, s1 = (int)hist[key];
        int rank = 0;
        for (int a = s0; a < s1; ++a) rank += idx[a] < v ? 1 : 0;
        dstpos[u] = s0 + rank;
        pt[u] = scan[v & 0x7FFFu];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (dstpos[u] >= 0) sp[dstpos[u]] = pt[u];
  }
  __syncthreads();
  return true;
}

// Owner: optimiser start + window geometry + spatial order of a scan of at most PER * kBlock points in ONE routine that
// keeps the scan in registers (round 3; compute_region + sort_points below remain for longer scans).  The separate
// routines read the scan from memory four times -- bounding box, histogram, scatter, and a gather by point number in the
// ranking phase, a few loads in flight each, every round a trip to L2 -- and every phase was a chain of dependent LDS
// round trips between two barriers: 33 us of the 50 us a scan's set-up took.  Here every thread loads its PER points
// once (all loads in flight; the counters are cleared and two lanes form the float32 matrix of the first pose while they
// travel), a thread's run of cell counters is read into registers once for both the marked-cell bitmap and the offsets, and
// every point goes from its register to its place in an LDS image of the ordered copy: no gather.
//
// Round 5 -- the order inside a cell.  It only has to be the SAME whatever the scheduling.  Rounds 3-4 kept the input order
// and paid for it by COUNTING: every entry its rank among the entries of its cell, 8 us of a 40-us set-up.  Now it is
// (wave, round, lane) of the thread that holds the point -- the order in which the scatter hands the places out when the
// waves take their turns one after the other: a wave's PER atomics on a cell counter are issued in round order and the LDS
// executes one wave's instructions in order; between the lanes of ONE instruction that hit the same counter the LDS serialises
// -- in lane order on this hardware, which the ISA does not promise, so every entry carries its sequence number and the order
// that came out is checked (one comparison per entry: the entries of the whole array must ascend); if it ever does not, the
// places are found by counting after all.  The turns are sixteen LDS-only barriers, 3.3 us (a token in LDS that the waves
// poll for their turn: 6 us).  Results differ from rounds 3-4 in the last bits of the sums (other order inside the cells), not
// in the float32 transforms or the iteration counts (every parity test, all configurations).
// The barriers of these phases order LDS traffic only (sync_lds): the ordered copy's stores to memory drain behind them.
#ifndef NDT_ORDER_INLINE
#define NDT_ORDER_INLINE __noinline__
#endif
template <bool SSE, int PER>
__device__ NDT_ORDER_INLINE void order_scan_regs(const MapView &M, const OptParams &P, const double *__restrict__ init,
                                             const float2 *__restrict__ scan, int n, Lds &L, uint4 *pool,
                                             float2 *__restrict__ sp, u64 *stamps = nullptr, u64 t0s = 0,
                                             ScanCtl *open_ctl = nullptr, unsigned *open_map = nullptr) {
  // The LDS pipe takes one wave instruction every ~4 cycles whatever its width (tools/repro/rates.hip), and these phases
  // are nothing but LDS traffic: counters, entries and places are moved 16 bytes at a time.
  constexpr int kRun = 20;                                             // counters per thread: five 16-byte words
  constexpr int kHistWords = (kRegionCells + 1 + 3) & ~3;              // ncell + 1 counters (last: outside the window), padded
  static_assert(kRun * kBlock >= kHistWords && kRun % 4 == 0, "every counter belongs to a thread's run");
  const int wv = (int)(threadIdx.x >> 6), ln = (int)(threadIdx.x & 63);
  // the optimiser's start first, while nothing else is live: one lane, ~5 us of dependent scalar code (Eigen's rotation() of the
  // first pose the bulk of it).  The other waves' loads are on their way meanwhile; wave 0 asks for its points behind it (a
  // call in front of which loads are in flight waits for them).  Round 5 tried the long part on the last wave beside the
  // scatter's turns, and all of it inlined behind the loads: slower both ways -- this phase is the 20 MB of scans that
  // the 256 workgroups of a launch ask for at the same moment, not the scalar code (LOG R5.1).
  if (threadIdx.x == 0) init_state(L.S, P, init, (double)n);
  float2 pt[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) pt[u] = gld_f2(scan + min((int)threadIdx.x + u * kBlock, n - 1));
  // ... while the first touch of the scan is on its way: cleared counters
  unsigned *wmap = reinterpret_cast<unsigned *>(L.wpart);
  unsigned *hist = reinterpret_cast<unsigned *>(pool);
  unsigned *idx = hist + kHistWords;                                   // n entries (cell << 15 | sequence number)
  unsigned *inv = idx + kSortRegs;                                     // (repair only) place of the point with sequence number q
#pragma unroll
  for (int k = 0; k < (kHistWords / 4 + kBlock - 1) / kBlock; ++k) {
    const int i = (int)threadIdx.x + k * kBlock;
    if (i < kHistWords / 4) pool[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (threadIdx.x < kRegionCells / 32) wmap[threadIdx.x] = 0u;
  if (threadIdx.x == 64) {
    L.sbox[0] = INT_MAX; L.sbox[1] = INT_MAX; L.sbox[2] = INT_MIN; L.sbox[3] = INT_MIN;
    L.sflag[0] = 0;                                                    // raised when the scatter's order has to be repaired
  }
  __syncthreads();
  NDT_STAMP(stamps, t0s, 0);
  const Tf32 T0 = L.S.T;
  // voxel coordinates at the first pose
  auto vox = [&](float2 p, int &ix, int &iy) {
    float xt, yt;
    tf_apply_t<SSE>(T0, p.x, p.y, xt, yt);
    const float fx = fminf(fmaxf(floorf(xt * M.inv_leaf), -1.0e9f), 1.0e9f);
    const float fy = fminf(fmaxf(floorf(yt * M.inv_leaf), -1.0e9f), 1.0e9f);
    ix = (int)fx - M.min_bx; iy = (int)fy - M.min_by;
    return finite2(xt, yt);
  };
  {
    int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      int ix, iy;
      if (vox(pt[u], ix, iy) && (int)threadIdx.x + u * kBlock < n) {
        mnx = ix < mnx ? ix : mnx; mxx = ix > mxx ? ix : mxx;
        mny = iy < mny ? iy : mny; mxy = iy > mxy ? iy : mxy;
      }
    }
    mnx = wave_min_dpp(mnx); mny = wave_min_dpp(mny); mxx = wave_max_dpp(mxx); mxy = wave_max_dpp(mxy);
    if (ln == 63 && mnx <= mxx) {
      atomicMin(&L.sbox[0], mnx); atomicMin(&L.sbox[1], mny); atomicMax(&L.sbox[2], mxx); atomicMax(&L.sbox[3], mxy);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) region_from_bbox(M, L.S.T, L);
  __syncthreads();
  NDT_STAMP(stamps, t0s, 1);
  const Region r = L.RG;
  const int ncell = r.rw * r.rh;
  int key[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    key[u] = -1;
    if ((int)threadIdx.x + u * kBlock < n) {
      int ix, iy;
      const bool fin = vox(pt[u], ix, iy);
      const int lx = ix - r.x0, ly = iy - r.y0;
      key[u] = (!fin || lx < 0 || lx >= r.rw || ly < 0 || ly >= r.rh) ? ncell : ly * r.rw + lx;
      atomicAdd(&hist[key[u]], 1u);
    }
  }
  sync_lds();
  NDT_STAMP(stamps, t0s, 2);
  // A thread's run of kRun counters, read once (counters past ncell are zero): its bits of the marked-cell bitmap (the
  // run touches at most two words) and its part of the exclusive scan (DPP scan over the wave, the waves' totals through LDS).
  {
    const int c0 = (int)threadIdx.x * kRun;
    const bool live = c0 < kHistWords;
    unsigned cnt[kRun];
#pragma unroll
    for (int k = 0; k < kRun / 4; ++k) {
      const uint4 q = live ? pool[(c0 >> 2) + k] : make_uint4(0u, 0u, 0u, 0u);
      cnt[4 * k] = q.x; cnt[4 * k + 1] = q.y; cnt[4 * k + 2] = q.z; cnt[4 * k + 3] = q.w;
    }
    unsigned mine = 0; u64 bits = 0;
#pragma unroll
    for (int k = 0; k < kRun; ++k) { mine += cnt[k]; if (cnt[k] != 0u && c0 + k < ncell) bits |= 1ull << k; }
    bits <<= (c0 & 31);
    if ((unsigned)bits) atomicOr(&wmap[c0 >> 5], (unsigned)bits);
    if ((unsigned)(bits >> 32)) atomicOr(&wmap[(c0 >> 5) + 1], (unsigned)(bits >> 32));
    const unsigned incl = wave_incl_scan(mine);
    if (ln == 63) L.swave[wv] = (int)incl;
    sync_lds();
    unsigned run = incl - mine;
    for (int w = 0; w < wv; ++w) run += (unsigned)L.swave[w];
    if (live) {
#pragma unroll
      for (int k = 0; k < kRun / 4; ++k) {
        uint4 q;
        q.x = run; run += cnt[4 * k]; q.y = run; run += cnt[4 * k + 1]; q.z = run; run += cnt[4 * k + 2]; q.w = run; run += cnt[4 * k + 3];
        pool[(c0 >> 2) + k] = q;
      }
    }
  }
  sync_lds();
  NDT_STAMP(stamps, t0s, 3);
  // A launch with idle workgroups from the start (fewer scans than workgroups: one scan at a time, the reference's own use) opens
  // the scan for joining HERE (round 5): the window's geometry and the marked cells are all a helper needs to stage its window --
  // it does that while this workgroup still puts the points in order, and is registered before the first pass instead of
  // 8 us behind it (the ordered copy is only read once a pass has been opened, behind this workgroup's own staging, whose
  // last barrier drains the copy's stores).  Everything stored write-through and drained before the ticket, as in the kernel.
  if (open_ctl) {
    for (int i = threadIdx.x; i < kRegionCells / 32; i += kBlock) st32(&open_map[i], wmap[i]);
    if (threadIdx.x == 0) {
      const Region r = L.RG;
      st32((u32 *)&open_ctl->region[0], (u32)r.x0); st32((u32 *)&open_ctl->region[1], (u32)r.y0); st32((u32 *)&open_ctl->region[2], (u32)r.rw);
      st32((u32 *)&open_ctl->region[3], (u32)r.rh); st32((u32 *)&open_ctl->region[4], (u32)r.cap); st32((u32 *)&open_ctl->region[5], 0u);
      st32(&open_ctl->use_sorted, 1u);
    }
    drain_vmem();
    __syncthreads();
    if (threadIdx.x == 0) st64(&open_ctl->ticket, (u64)1 << 32);       // epoch 1: open for joining, nothing to compute yet
  }
  // Places: the waves in turn (see the head of the routine).
  unsigned place[PER];
  for (int w = 0; w < kWaves; ++w) {
    if (wv == w) {
#pragma unroll
      for (int u = 0; u < PER; ++u) place[u] = key[u] >= 0 ? atomicAdd(&hist[key[u]], 1u) : 0u;
    }
    sync_lds();                                                        // (the returns are in: the next wave's atomics come behind)
  }
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (key[u] >= 0) idx[place[u]] = ((unsigned)key[u] << 15) | (unsigned)((wv * PER + u) * 64 + ln);
  sync_lds();
  NDT_STAMP(stamps, t0s, 4);
  {
    bool bad = false;
    unsigned va[PER], vb[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {                                    // all in flight
      const int j = (int)threadIdx.x + u * kBlock;
      va[u] = idx[min(j, n - 1)]; vb[u] = idx[min(j + 1, n - 1)];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) bad |= ((int)threadIdx.x + u * kBlock + 1 < n) && !(va[u] < vb[u]);
    if (__ballot(bad) != 0ull && ln == 0) L.sflag[0] = 1;              // (raised by any wave)
  }
  sync_lds();
  NDT_STAMP(stamps, t0s, 11);
  if (L.sflag[0]) {
    // not in sequence: every entry finds its rank among the entries of its cell by counting (consecutive lanes on consecutive
    // entries, i.e. mostly on the same cell: equal trip counts, broadcast reads, four entries of the cell per LDS read);
    // hist[c] = end of cell c by now.  The place of the point with sequence number q goes to inv[q].
    const uint4 *idx4 = reinterpret_cast<const uint4 *>(idx);
    unsigned v[PER]; int s0[PER], s1[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) v[u] = idx[min((int)threadIdx.x + u * kBlock, n - 1)];      // all in flight
#pragma unroll
    for (int u = 0; u < PER; ++u) {                                                            // all in flight
      const int k = (int)(v[u] >> 15);
      s0[u] = (int)hist[max(k - 1, 0)]; s1[u] = (int)hist[k];
      if (k == 0) s0[u] = 0;
      if ((int)threadIdx.x + u * kBlock >= n) s1[u] = s0[u];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      int rank = 0;
      for (int q = s0[u] >> 2; q <= (s1[u] - 1) >> 2 && s1[u] > s0[u]; ++q) {   // aligned quads covering [s0, s1)
        const uint4 e = idx4[q];
        const int a = q << 2;
        rank += (a >= s0[u] && a < s1[u] && e.x < v[u]) ? 1 : 0;
        rank += (a + 1 >= s0[u] && a + 1 < s1[u] && e.y < v[u]) ? 1 : 0;
        rank += (a + 2 >= s0[u] && a + 2 < s1[u] && e.z < v[u]) ? 1 : 0;
        rank += (a + 3 >= s0[u] && a + 3 < s1[u] && e.w < v[u]) ? 1 : 0;
      }
      if (s1[u] > s0[u]) inv[v[u] & 0x7FFFu] = (unsigned)(s0[u] + rank);
    }
    sync_lds();
#pragma unroll
    for (int u = 0; u < PER; ++u) place[u] = key[u] >= 0 ? inv[(wv * PER + u) * 64 + ln] : 0u;
    sync_lds();
  }
  float2 *stage = reinterpret_cast<float2 *>(pool);                    // n points over the dead counters and entries
  static_assert(kSortRegs * 8 <= (kHistWords + kSortRegs) * 4, "the LDS image of the ordered copy must not reach the places");
#pragma unroll
  for (int u = 0; u < PER; ++u) if (key[u] >= 0) stage[place[u]] = pt[u];
  sync_lds();                                      // the image is complete: copy_out_ordered sends it to memory
}

// The LDS image of the ordered copy (order_scan_regs) to the scratch copy in memory, in coalesced 16-byte stores,
// write-through: helpers on other XCDs read this copy (the publishing step drains the stores).  A routine of its own, and a
// small one (round 5): a routine that keeps values in callee-saved registers reloads them behind an s_waitcnt vmcnt(0) on its
// way out -- inside order_scan_regs that wait sat right behind these stores and held the workgroup until all 80 KB had
// landed (2-4 us).  Issued from here, the stores drain while the window's slot table is written.
__device__ __noinline__ void copy_out_ordered(const uint4 *pool, float2 *__restrict__ sp, int n) {
  const float4 *st4 = reinterpret_cast<const float4 *>(pool);
  const float2 *stage = reinterpret_cast<const float2 *>(pool);
  float4 *sp4 = reinterpret_cast<float4 *>(sp);
  const bool aligned = (reinterpret_cast<size_t>(sp) & 15) == 0;
  if (aligned) {
    for (int i = threadIdx.x; i < n / 2; i += kBlock) st_wt_f4(sp4 + i, st4[i]);
    if (threadIdx.x == 0 && (n & 1)) st_wt_f2(sp + (n - 1), stage[n - 1]);
  } else {
    for (int i = threadIdx.x; i < n; i += kBlock) st_wt_f2(sp + i, stage[i]);
  }
  sync_lds();                                      // (the image has been read: the pool is the window's from here on)
}

// Sum of 12 per-lane values over the 64 lanes of a wave in a fixed order: a butterfly in which every exchange also
// halves the number of values a lane carries (12 -> 6 -> 3 -> 2 -> 1).  The total of value j ends in the lanes whose
// bits select j; those lanes store it to dst[j] (LDS).
//
// Round 3: the exchanges are gfx950 cross-lane VALU moves instead of 24 x 2 ds_bpermute_b32 through the LDS pipe (each
// stage waited ~100 cycles for its shuffles: ~1 us per unit on a wave that has nothing else to issue -- the helpers'
// lone waves).  v_permlane32_swap / v_permlane16_swap (new in CDNA4) exchange the halves / odd-even rows of TWO
// registers in one instruction, which is exactly the "keep one half, send the other" step, so the selects vanish too;
// the xor-8 / xor-4 / xor-2 / xor-1 exchanges are DPP row rotations and quad permutes with bank masks.  Same pairs of
// lanes, same additions: the sums are bit-identical to the shuffle version (fp addition commutes).
__device__ __forceinline__ unsigned dlo(double x) { return (unsigned)__double_as_longlong(x); }
__device__ __forceinline__ unsigned dhi(double x) { return (unsigned)((u64)__double_as_longlong(x) >> 32); }
__device__ __forceinline__ double dmk(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((u64)hi << 32) | lo)); }
// lanes 0..31: x(l) + x(l + 32);  lanes 32..63: y(l - 32) + y(l)
__device__ __forceinline__ double swap32_add(double x, double y) {
  const auto a = __builtin_amdgcn_permlane32_swap(dlo(x), dlo(y), false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(dhi(x), dhi(y), false, false);
  return dmk(a[0], b[0]) + dmk(a[1], b[1]);
}
// rows of 16 lanes; even rows: x(l) + x(l + 16);  odd rows: y(l - 16) + y(l)
__device__ __forceinline__ double swap16_add(double x, double y) {
  const auto a = __builtin_amdgcn_permlane16_swap(dlo(x), dlo(y), false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(dhi(x), dhi(y), false, false);
  return dmk(a[0], b[0]) + dmk(a[1], b[1]);
}
// DPP move of a double: lanes enabled by BANKS (one bit per group of four lanes of a row) take src from the lane CTRL
// names, the others keep `old`
template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_d(double old, double src) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)dlo(old), (int)dlo(src), CTRL, 0xF, BANKS, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)dhi(old), (int)dhi(src), CTRL, 0xF, BANKS, false);
  return dmk(lo, hi);
}
constexpr int kDppRor4 = 0x124, kDppRor8 = 0x128, kDppRor12 = 0x12C;    // row_ror:n -- lane l reads lane (l - n) mod 16 of its row
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E;                         // quad_perm [1,0,3,2] / [2,3,0,1]
// lanes whose bit `8` (BIT8) or bit `4` is clear: x(l) + x(l + d);  set: y(l - d) + y(l)      (d = 8 or 4)
__device__ __forceinline__ double xor8_add(double x, double y) {
  const double keep = dpp_d<kDppRor8, 0xC>(x, y);     // lanes 8..15 of a row: y(l - 8)
  const double recv = dpp_d<kDppRor8, 0x3>(y, x);     // lanes 0..7:          x(l + 8)
  return keep + recv;
}
__device__ __forceinline__ double xor4_add(double x, double y) {
  const double keep = dpp_d<kDppRor4, 0xA>(x, y);     // lanes 4..7, 12..15: y(l - 4)
  const double recv = dpp_d<kDppRor12, 0x5>(y, x);    // lanes 0..3, 8..11:  x(l + 4)
  return keep + recv;
}
__device__ __forceinline__ void wave_reduce12(const double (&a)[12], int lane, double *__restrict__ dst) {
  const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
  double k[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) k[i] = swap32_add(a[i], a[i + 6]);          // values 0..5 (b5 = 0) or 6..11 (b5 = 1)
  double m[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) m[i] = swap16_add(k[i], k[i + 3]);          // 0..2 or 3..5 of those
  const double p0 = xor8_add(m[0], m[1]);                                 // value 0 or 1 of the triple
  const double p1 = m[2] + dpp_d<kDppRor8, 0xF>(m[2], m[2]);              // value 2
  double r = xor4_add(p0, p1);
  r += dpp_d<kDppXor2, 0xF>(r, r);
  r += dpp_d<kDppXor1, 0xF>(r, r);
  const int idx = (b5 ? 6 : 0) + (b4 ? 3 : 0) + (b2 ? 2 : (b3 ? 1 : 0));
  if ((lane & 3) == 0 && !(b2 && b3)) dst[idx] = r;
}

// Units of a pass: unit u = (virtual wave w = u % kWaves, run q = u / kWaves) is the lane set
// {w*64 .. w*64+63} walking the q-th run of its points i = w*64 + lane + k*kBlock,
// k in [q*run, (q+1)*run), of the (ordered) scan.  Any physical wave of any workgroup can compute
// a unit; its sums are reduced over the 64 lanes in a fixed order, and a pass total is the sum of
// the kUnits unit totals in unit order -- the same arithmetic whether the owner computed all units
// itself or helpers computed some.
// This routine computes the consecutive runs [q0, q1) of virtual wave w in ONE walk over k (the
// point prefetch keeps running across run boundaries) and leaves the 12 sums of run q at
// dst[(q - q0) * dst_stride .. +12) (LDS).
// wave-uniform values read from LDS land in VGPRs; these move them to SGPRs (the pass loop is short of VGPRs)
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uniform_d(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

__device__ __forceinline__ unsigned uniform_u(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ const T *uniform_p(const T *p) {
  const u64 b = (u64)p;
  return (const T *)(((u64)uniform_u((unsigned)(b >> 32)) << 32) | uniform_u((unsigned)b));
}

// The pass loop needs the register file to itself.  In round 1's kernel -- everything inlined into one persistent
// body, with a dozen 64-bit diagnostic timers alive across every loop -- it inherited what the kernel keeps alive around
// it and the compiler spilled inside the point loop (a dozen scratch reloads per point, each behind an
// s_waitcnt vmcnt(0)).  Two things fixed that: the phases AROUND the loop became functions of their own
// (compute_region, sort_points, fill_window, advance: their registers no longer overlap the loop's), and the loop takes
// its inputs from LDS (g_L.M, g_L.RG, g_L.PP, g_L.pts), moved to SGPRs once per call, instead of from values that would
// have to stay live across the whole kernel.  For a while this routine was a function too (-DNDT_PASS_INLINE=__noinline__
// still builds that): same speed, but every call saved and restored 18 VGPRs per lane through scratch -- 300 MB of
// write traffic per launch -- so it is inlined again, spill-free now (the kernel's 67 spilled VGPRs are in cold code).
#ifndef NDT_PASS_INLINE
#define NDT_PASS_INLINE __forceinline__
#endif
// Two ways of being used (once per wave and pass):
//   step == 0: solo pass -- wave `first` walks its own kSub units (first, 0..kSub-1) in one go, totals to L.wpart;
//   step  > 0: shared pass -- the wave takes units (j from the workgroup's LDS counter) 0 .. lead-1, then
//              first + (j - lead) * step until they reach uend; totals to L.wpart (owner, vtot == nullptr) or, tagged,
//              straight to the scan's unit totals in HBM (helper; lead = 0).
template <bool SSE, bool INCL, bool CHK>
__device__ NDT_PASS_INLINE void pass_units(int first_in, int step_in, int uend_in, u64 *vtot_in, unsigned tag_in, int lead_in) {
  const int first = (int)uniform_u((unsigned)first_in), step = (int)uniform_u((unsigned)step_in);
  const int uend = (int)uniform_u((unsigned)uend_in), lead = (int)uniform_u((unsigned)lead_in);
  const u64 tag = (u64)uniform_u(tag_in) << 32;
  u64 *const vtot = (u64 *)uniform_p(vtot_in);
  Lds &L = g_L;
  MapView M;
  M.inv_leaf = uniform_f(L.M.inv_leaf); M.leaf = uniform_f(L.M.leaf); M.r2 = uniform_f(L.M.r2);
  M.radius_inclusive = INCL; M.transform_sse = SSE;
  M.min_bx = (int)uniform_u((unsigned)L.M.min_bx); M.min_by = (int)uniform_u((unsigned)L.M.min_by);
  M.div_x = (int)uniform_u((unsigned)L.M.div_x); M.div_y = (int)uniform_u((unsigned)L.M.div_y);
  M.gw = (int)uniform_u((unsigned)L.M.gw); M.gh = (int)uniform_u((unsigned)L.M.gh);
  M.cent = uniform_p(L.M.cent); M.rec = uniform_p(L.M.rec); M.occ = uniform_p(L.M.occ);
  M.pt_start = uniform_p(L.M.pt_start); M.pts = uniform_p(L.M.pts);
  M.d1 = uniform_d(L.M.d1); M.d2 = uniform_d(L.M.d2); M.e_hi = uniform_d(L.M.e_hi);
  Region R;
  R.x0 = (int)uniform_u((unsigned)L.RG.x0); R.y0 = (int)uniform_u((unsigned)L.RG.y0);
  R.rw = (int)uniform_u((unsigned)L.RG.rw); R.rh = (int)uniform_u((unsigned)L.RG.rh);
  R.cap = (int)uniform_u((unsigned)L.RG.cap); R.nspill = (int)uniform_u((unsigned)L.RG.nspill);
  const Window W = window_of(R, g_pool);
  const double *__restrict__ etab = L.etab;
  const PassPose &pp_in = L.PP;
  const float2 *__restrict__ pts = uniform_p(L.pts);
  const int n = (int)uniform_u((unsigned)L.npts);
  PassPose pp;
  pp.T.c = uniform_f(pp_in.T.c); pp.T.s = uniform_f(pp_in.T.s); pp.T.tx = uniform_f(pp_in.T.tx); pp.T.ty = uniform_f(pp_in.T.ty);
  pp.cj = uniform_d(pp_in.cj); pp.sj = uniform_d(pp_in.sj); pp.ch = uniform_d(pp_in.ch); pp.sh = uniform_d(pp_in.sh);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, last = n - 1;
  const int per_lane = (n + kBlock - 1) / kBlock;          // points of the longest lane
#ifdef NDT_RUNS_CEIL
  const int run = (per_lane + kSub - 1) / kSub;              // runs of equal length, the last one short (10 rounds: 3 3 3 1)
  auto kstart = [&](int q) { return min(per_lane, q * run); };
#else
  // runs as equal as they get (10 rounds: 3 3 2 2): in a pass shared by two workgroups every wave then walks a long and
  // a short unit -- five rounds -- instead of half the waves two long ones (round 3)
  const int rbase = per_lane / kSub, rextra = per_lane % kSub;
  auto kstart = [&](int q) { return q * rbase + min(q, rextra); };
#endif
  for (int it = 0; it <= kUnits; ++it) {                   // counted (tools/repro/ticket2.hip)
    int w, q0, q1, dst_stride;
    double *dst;
    if (step == 0) {
      if (it > 0) break;
      w = first; q0 = 0; q1 = kSub; dst = L.wpart + first * 12; dst_stride = kWaves * 12;
    } else {
      int j = 0;
      if (lane == 0) j = atomicAdd(&L.jnext, 1);
      j = __builtin_amdgcn_readfirstlane(j);
      const int u = j < lead ? j : first + (j - lead) * step;
      if (u >= uend) break;
      w = u % kWaves; q0 = u / kWaves; q1 = q0 + 1; dst_stride = 0;
      dst = vtot ? L.wtmp + wave * 12 : L.wpart + u * 12;
    }
    const int kbeg = kstart(q0), kend = kstart(q1);
    const int base = w * 64 + lane;
    Acc A = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0u};
    // (loads at a 32-bit byte offset from the uniform base: one shift per address instead of a sign extension + 64-bit add)
    float2 p0 = gld_f2_at(pts, (unsigned)min(base + kbeg * kBlock, last) * 8u), p1 = gld_f2_at(pts, (unsigned)min(base + (kbeg + 1) * kBlock, last) * 8u);
    int q = q0, kb = kstart(q0 + 1);                       // end of the current run
#pragma nounroll
    for (int k = kbeg; k < kend; ++k) {
      const float2 p2 = gld_f2_at(pts, (unsigned)min(base + (k + 2) * kBlock, last) * 8u);
      if ((k + 1) * kBlock > n) {                          // (uniform) only the scan's last round has lanes past the end
        if (base + k * kBlock >= n) p0.x = NAN;            // past the end: contributes nothing
      }
      eval_point<SSE, INCL, CHK>(M, W, etab, pp.T, p0.x, p0.y, pp.cj, pp.sj, pp.ch, pp.sh, A);
      p0 = p1; p1 = p2;
      if (k + 1 == kb) {                                   // run q complete (uniform across the wave)
        const double a[12] = {A.e, A.g0, A.g1, A.g2, A.hxx, A.hxy, A.hxt, A.hyy, A.hyt, A.htt, (double)A.pairs, 0.0};
        wave_reduce12(a, lane, dst + (q - q0) * dst_stride);
        A = Acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0u};
        ++q; kb = kstart(q + 1);
      }
    }
    for (; q < q1; ++q) {                                  // empty runs (short scans)
      if (lane < 12) dst[(q - q0) * dst_stride + lane] = 0.0;
    }
    if (step != 0 && vtot) {                               // helper: publish the unit (write-through stores), each
      const int u = q0 * kWaves + w;                       // half of a sum under the tag of the pass
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < kUnitWords) {
        const u64 bits = (u64)__double_as_longlong(dst[lane >> 1]);
        st64(&vtot[u * kUnitWords + lane], tag | (u64)(u32)((lane & 1) ? (bits >> 32) : bits));
      }
    }
  }
}

// Bound on every spin, measured from the start of that spin (a launch may legitimately run longer
// than any bound); looked at once per 64 polls (the abort word is one line shared by the chip).
__device__ __forceinline__ bool watchdog(WsHeader *hdr, u64 spin_start, unsigned &polls) {
  if ((++polls & 63u) != 0u) return false;
  if (ld32(&hdr->abort)) return true;
  if (wall_clock64() - spin_start > kWatchTicks) { st32(&hdr->abort, 1u); return true; }
  return false;
}

__device__ __forceinline__ u64 wave_bcast64(u64 v) {   // lane 0's value to the whole wave
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}

// ------------------------------------------------------------------------------------------
// A batch prepared ahead (round 5; ndt_align_batch_prepare_dev).  What an owner does with a scan before it can stage the
// window -- the optimiser's start, the window geometry, the spatial order, the marked-cell bitmap -- depends on the scan,
// its initial guess and the map's grid only, not on anything the matches compute: a caller with a stream of batches can have
// it done for batch i + 1 while the matches of batch i run out (the CUs their workgroups leave are idle otherwise).
// ndt_order_kernel runs the owner's own routines (order_scan_regs, copy_out_ordered) and leaves per scan one PrepRec, the
// bitmap and the ordered copy; an owner that finds a record for its scan starts at the window's staging: 33 -> ~10 us of
// set-up inside the match kernel.  Same routines, same data: the records of a prepared batch are byte-identical.
// ------------------------------------------------------------------------------------------
struct alignas(16) PrepRec {
  int region[6];                 // Region of the scan's window
  int clipped, ok;               // ok: 1 = this scan has been prepared (0: the owner does it itself, e.g. a scan beyond kSortRegs points)
  AlignState S;                  // the optimiser's start (init_state)
};
constexpr int kPrepWords = (int)(sizeof(PrepRec) / 4);
static_assert(sizeof(PrepRec) % 16 == 0 && kPrepWords <= kBlock, "PrepRec is copied word by word by one workgroup");

template <bool SSE>
__global__ void __launch_bounds__(kBlock)
ndt_order_kernel(MapView M, OptParams P, const float *__restrict__ scans, const unsigned long long *__restrict__ offsets,
                 int B, int shared_scan, const double *__restrict__ inits, float2 *__restrict__ sorted,
                 PrepRec *__restrict__ prep, unsigned *__restrict__ prep_map /* [B][kRegionCells / 32] */) {
  Lds &L = g_L;
  uint4 *const pool = g_pool;
  if (threadIdx.x == 64) { L.M = M; L.P = P; }
  __syncthreads();
  for (int b = (int)blockIdx.x; b < B; b += (int)gridDim.x) {
    const u64 o0 = shared_scan ? offsets[0] : offsets[b];
    const u64 o1 = shared_scan ? offsets[1] : offsets[b + 1];
    const int n = (int)(o1 - o0);
    PrepRec *R = prep + b;
    if (!(n > 0 && n <= kSortRegs)) {                          // (uniform) left to the owner's streaming routines
      if (threadIdx.x == 0) R->ok = 0;
      continue;
    }
    const float2 *scan = reinterpret_cast<const float2 *>(scans) + o0;
    float2 *sp = shared_scan ? sorted + (size_t)b * (size_t)n : sorted + o0;
    order_scan_regs<SSE, kSortRegs / kBlock>(L.M, L.P, inits + 3 * (size_t)b, scan, n, L, pool, sp);
    copy_out_ordered(pool, sp, n);
    if (threadIdx.x == 0) {
      const Region r = L.RG;
      R->region[0] = r.x0; R->region[1] = r.y0; R->region[2] = r.rw; R->region[3] = r.rh; R->region[4] = r.cap; R->region[5] = 0;
      R->clipped = L.clipped; R->ok = 1;
    }
    {
      const u32 *src = reinterpret_cast<const u32 *>(&L.S);
      u32 *dst = reinterpret_cast<u32 *>(&R->S);
      if (threadIdx.x < sizeof(AlignState) / 4) dst[threadIdx.x] = src[threadIdx.x];
      const unsigned *wmap = reinterpret_cast<const unsigned *>(L.wpart);
      unsigned *gm = prep_map + (size_t)b * (kRegionCells / 32);
      for (int i = threadIdx.x; i < kRegionCells / 32; i += kBlock) gm[i] = wmap[i];
    }
    __syncthreads();
  }
}

template <bool SSE, bool INCL, bool CHK>
__global__ void __launch_bounds__(kBlock)
ndt_align_kernel(MapView M, OptParams P, const float *__restrict__ scans,
                 const unsigned long long *__restrict__ offsets, int B, int shared_scan,
                 const double *__restrict__ inits, ndt_result *__restrict__ results,
                 double *__restrict__ trace, int trace_cap, int *__restrict__ trace_rows,
                 float2 *__restrict__ sorted /* scratch, same offsets as scans; may be null */,
                 unsigned char *__restrict__ ws /* WsHeader, ScanCtl[B], unit totals[B][kUnits][24], marked-cell bitmaps[B][kRegionCells/32] */,
                 int allow_helpers /* 0: none; else max helper workgroups per scan */,
                 unsigned long long *__restrict__ prof /* diagnostic: 8 words per scan */,
                 const PrepRec *__restrict__ prep /* batch prepared ahead (ndt_order_kernel), or null */,
                 const unsigned *__restrict__ prep_map) {
  Lds &L = g_L;
  uint4 *const pool = g_pool;
  WsHeader *hdr = reinterpret_cast<WsHeader *>(ws);
  ScanCtl *ctl = reinterpret_cast<ScanCtl *>(ws + sizeof(WsHeader));
  u64 *utot = reinterpret_cast<u64 *>(ws + sizeof(WsHeader) + (size_t)B * sizeof(ScanCtl));
  unsigned *wantmap = reinterpret_cast<unsigned *>(ws + sizeof(WsHeader) + (size_t)B * sizeof(ScanCtl) +
                                                  (size_t)B * kUnits * kUnitWords * sizeof(u64));
  const u64 t_start = kProf ? wall_clock64() : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 64) L.etab[threadIdx.x] = c_exp2_tab[threadIdx.x];
  if (threadIdx.x == 64) { L.M = M; L.P = P; }
  bool aborted = false;

  // =========================== owner of scans blockIdx.x, + gridDim.x, ... ===========================
  // scans are taken from a queue: the first gridDim.x by workgroup number, the rest in the order
  // workgroups become free (results do not depend on who owns which scan).
  // Claims (round 3): the first gridDim.x scans belong to the workgroups of the same number only once those have won a
  // compare-and-swap on the scan's `claimed` word.  A launch is NOT guaranteed one resident workgroup per CU -- another
  // queue's kernel may hold CUs -- and a workgroup that is not resident cannot start the scan its number stands for; the
  // resident ones used to finish their own scans and then poll that scan's control word for as long as the foreign kernel
  // ran (tests/test_gpu_robustness.py: 20 ms).  Now a workgroup that has used the queue up first looks for a scan nobody
  // has claimed and takes it over; the late workgroup finds its scan taken and moves on: no wait in the kernel depends
  // on a workgroup that is not running.  (Scans of the queue are handed out by a counter to workgroups that are running.)
  // With fewer scans than workgroups the first 4 B workgroups bid for a scan each -- workgroup j B + r for scan (r + j) mod B,
  // so that a scan's (up to) four bidders sit on different XCDs: whichever of them is resident first owns the scan, the
  // others find it taken and help (results do not depend on the owner); a scan waits for a workgroup that is not running
  // only if NONE of its bidders is.  tests/test_gpu_robustness.py: B = 24.
  int preclaimed = -1;
  int b0 = (int)blockIdx.x;
  if (allow_helpers && B > 0 && B < (int)gridDim.x && blockIdx.x < 4u * (unsigned)B)
    b0 = (int)((blockIdx.x % (unsigned)B + blockIdx.x / (unsigned)B) % (unsigned)B);
  for (int b = b0; b < B && !aborted;) {
    const u64 o0 = shared_scan ? offsets[0] : offsets[b];
    const u64 o1 = shared_scan ? offsets[1] : offsets[b + 1];
    const int n = (int)(o1 - o0);
    const float2 *scan = reinterpret_cast<const float2 *>(scans) + o0;
    double *tr = trace ? trace + (size_t)b * trace_cap * 8 : nullptr;
    ScanCtl *C = ctl + b;
    u64 *mytot = utot + (size_t)b * kUnits * kUnitWords;
    __syncthreads();
    // scans that fit the register-resident set-up (order_scan_regs) get their optimiser state set up in there, under the
    // latency of the scan's first touch
    const bool reg_path = n > 0 && sorted != nullptr && n <= kSortRegs;
    const bool prepared = reg_path && prep != nullptr && prep[b].ok != 0;        // (uniform: one word per scan)
    // fewer scans than workgroups: idle workgroups from the start -- the scan is opened to them in the middle of its ordering
    const bool early_open = allow_helpers != 0 && reg_path && !prepared && B < (int)gridDim.x;
    if (threadIdx.x == 0) {
      u32 expect = 0u;
      const bool won = !allow_helpers || b >= (int)gridDim.x || b == preclaimed ||
                       __hip_atomic_compare_exchange_strong(&C->claimed, &expect, 1u, NDT_RLX, NDT_RLX, NDT_AGENT);
      L.sflag[3] = won ? 1 : 0;
      if (won) {
        if (NDT_XCD_BONUS && allow_helpers) st32(&C->owner_xcd, 1u + (blockIdx.x & 7u));
        if (!reg_path) init_state(L.S, L.P, inits + 3 * (size_t)b, (double)n);
        if (trace_rows) trace_rows[b] = 0;
        if (n <= 0) { L.S.phase = PH_DONE; L.S.converged = 0; }
      }
    }
    __syncthreads();
    if (!L.sflag[3]) {                        // taken over by another workgroup while this one was not resident: next scan
      __syncthreads();
      if (threadIdx.x == 0) L.sflag[3] = (int)gridDim.x + (int)__hip_atomic_fetch_add(&hdr->next, 1u, NDT_RLX, NDT_AGENT);
      __syncthreads();
      b = L.sflag[3];
      continue;
    }
    const float2 *pts = scan;
    u64 *const stamps = (kProf && prof) ? prof + 16 * (size_t)B + 16 * (size_t)b : nullptr;
    const u64 t0s = kProf ? wall_clock64() : 0;
    if (n > 0) {
      const u64 q0 = kProf ? wall_clock64() : 0;
      if (!reg_path) NDT_STAMP(stamps, t0s, 0);
      // scratch copy: at the scan's own offsets, or (every match uses scan 0) one slot per match
      float2 *sp = sorted ? (shared_scan ? sorted + (size_t)b * (size_t)n : sorted + o0) : nullptr;
      u64 q1 = 0;
      if (prepared) {
        // ordered ahead of the launch: the optimiser's start, the geometry and the bitmap come from the record, the copy is in place
        const PrepRec *R = prep + b;
        if (threadIdx.x < sizeof(AlignState) / 4)
          reinterpret_cast<u32 *>(&L.S)[threadIdx.x] = reinterpret_cast<const u32 *>(&R->S)[threadIdx.x];
        if (threadIdx.x == 64) {
          Region r; r.x0 = R->region[0]; r.y0 = R->region[1]; r.rw = R->region[2]; r.rh = R->region[3]; r.cap = R->region[4]; r.nspill = 0;
          L.RG = r; L.clipped = R->clipped;
        }
        unsigned *wmap = reinterpret_cast<unsigned *>(L.wpart);
        const unsigned *gm = prep_map + (size_t)b * (kRegionCells / 32);
        for (int i = threadIdx.x; i < kRegionCells / 32; i += kBlock) wmap[i] = gm[i];
        pts = sp;
        __syncthreads();
        NDT_STAMP(stamps, t0s, 0);
      } else if (reg_path) {
        order_scan_regs<SSE, kSortRegs / kBlock>(L.M, L.P, inits + 3 * (size_t)b, scan, n, L, pool, sp, stamps, t0s,
                                                 early_open ? C : nullptr, early_open ? wantmap + (size_t)b * (kRegionCells / 32) : nullptr);
        pts = sp;
      } else {
        compute_region<SSE>(L.M, L.S.T, scan, n, L);
        q1 = kProf ? wall_clock64() : 0;
        NDT_STAMP(stamps, t0s, 1);
        if (sort_points<SSE>(L.M, L.S.T, scan, n, L, pool, sp, stamps, t0s)) pts = sp;
      }
      if (kProf && q1 == 0) q1 = wall_clock64();
      NDT_STAMP(stamps, t0s, 5);
      const u64 q2 = kProf ? wall_clock64() : 0;
      // Which voxels get a record (control block only; its loads of the map's occupancy words are in front of the copy's
      // stores: a wait for a load is a wait for every store issued before it), then the ordered copy on its way to memory.
      fill_window_plan(L.M, L, stamps, t0s);
      if (reg_path && !prepared) copy_out_ordered(pool, sp, n);
      if (allow_helpers && !early_open) {          // helpers rebuild the same window from this bitmap
        const unsigned *wmap = reinterpret_cast<const unsigned *>(L.wpart);
        unsigned *gw = wantmap + (size_t)b * (kRegionCells / 32);
        for (int i = threadIdx.x; i < kRegionCells / 32; i += kBlock) st32(&gw[i], wmap[i]);
      }
      // Publication: geometry + marked cells + ordered copy.  Everything a helper will read is stored write-through (geometry
      // and marked cells here, the ordered copy above) and drained by every wave before the ticket is written, so
      // the scan is opened without an agent-scope release -- a write-back of this XCD's whole L2, 3-4 us on the critical path of
      // every scan.  Scans ordered by the streaming routines (plain stores) keep the release.
      //   * a launch with idle workgroups from the start (fewer scans than workgroups: one scan at a time) publishes BEFORE
      //     staging the own window, so that the others stage theirs meanwhile;
      //   * a launch with a scan for every workgroup has nobody to publish to yet: the ticket is written BEHIND the window's
      //     staging, whose last barrier has drained the stores anyway -- the 2 us of waiting for the copy's 80 KB to land
      //     go by while the slot table is written (round 5).
      const bool late_publish = reg_path && B >= (int)gridDim.x;
      if (allow_helpers && !early_open && threadIdx.x == 0) {
        const Region r = L.RG;
        st32((u32 *)&C->region[0], (u32)r.x0); st32((u32 *)&C->region[1], (u32)r.y0); st32((u32 *)&C->region[2], (u32)r.rw);
        st32((u32 *)&C->region[3], (u32)r.rh); st32((u32 *)&C->region[4], (u32)r.cap); st32((u32 *)&C->region[5], (u32)r.nspill);
        st32(&C->use_sorted, (pts != scan) ? 1u : 0u);
      }
      if (allow_helpers && !late_publish && !early_open) {
        drain_vmem();
        __syncthreads();
        if (threadIdx.x == 0) {
          if (!reg_path) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            drain_vmem();
          }
          st64(&C->ticket, (u64)1 << 32);                     // epoch 1: open for joining, nothing to compute (h = 0)
        }
      }
      NDT_STAMP(stamps, t0s, 6);
      fill_window_fetch(L.M, L, pool, stamps, t0s);         // (its last barrier waits for every wave's outstanding stores)
      if (allow_helpers && late_publish && threadIdx.x == 0) st64(&C->ticket, (u64)1 << 32);
      NDT_STAMP(stamps, t0s, 10);
      const u64 q3 = kProf ? wall_clock64() : 0;
      if (kProf && prof && threadIdx.x == 0) {
        prof[8 * (size_t)B + 8 * (size_t)b + 6] = ((q1 - q0) << 32) | ((q2 - q1) & 0xFFFFFFFFull);
        prof[8 * (size_t)B + 8 * (size_t)b + 7] = ((q3 - q2) << 32) | ((u64)(L.diag[0] & 0xFFFFu) << 16) | (u64)(L.diag[1] & 0xFFFFu);
      }
    }
    unsigned epoch = 1;
    if (threadIdx.x == 0) {
      int ready = 0;
      // A launch with fewer scans than workgroups (one scan at a time: the reference's own use) has helpers from the
      // start: they stage their windows while this workgroup finishes its own (ready ~13 us after it).  Waiting for them
      // -- at most kFirstPassWait ticks -- makes the first pass a shared one: 10 us instead of 34 (one scan: 0.215 ->
      // 0.200 ms).
      if (allow_helpers && n > 0 && (int)gridDim.x > B && b < (int)gridDim.x) {
        const int want = min(allow_helpers, ((int)gridDim.x - B) / B);   // helpers every scan of the launch can count on
        const u64 w0 = wall_clock64();
        // (only when every scan can have its full complement: with one to three helpers per scan the wait does not pay --
        // B = 64 / 128: +6 / +3 %)
        while (want >= allow_helpers && ready < want && wall_clock64() - w0 < kFirstPassWait) {
          ready = (int)rd32_fresh(&C->ready);
          __builtin_amdgcn_s_sleep(8);
        }
      }
      L.sflag[1] = ready; L.pts = pts; L.npts = n;       // sflag[1]: registered helpers (refreshed during every advance)
      L.sflag[2] = 0;                                    // set by a thread whose wait ran into the watchdog
    }
    u64 t_eval = 0, t_adv = 0, tt0 = 0, tt1 = 0, t_wait = 0, t_first_shared = 0, t_fit = 0;
    u64 ts1 = 0, ts2 = 0, ts3 = 0, a_pro = 0, a_own = 0, a_wait = 0, a_comb = 0, a_adv = 0, a_n = 0;   // shared derivative passes (diagnostic)
    const u64 t_scan0 = kProf ? wall_clock64() - t_start : 0;
    unsigned n_shared = 0, n_helped = 0;
    // ---- derivative passes until the optimiser stops (the fitness score is a kernel of its own: ndt_fitness.hip.h) ----
    while (n > 0 && L.S.phase != PH_DONE) {
      if (kProf && prof) tt0 = wall_clock64();
      // A pass is solo (one walk per wave) or split over the registered helpers.
      int pass_h = 0;
      if (threadIdx.x == 0) {
        const PassPose pp = {L.S.T, L.S.cj, L.S.sj, L.S.ch, L.S.sh};
        L.PP = pp;
        const int h = allow_helpers ? min(L.sflag[1], kMaxHelpers) : 0;
        L.sflag[0] = h;
        L.jnext = 0;
        if (allow_helpers) {
          st32(&C->passes, (u32)L.S.evals);
          const float spp = (float)(L.S.score / (double)n);
          st32(&C->spp, spp > 0.f ? __float_as_uint(spp) : 0u);
        }
        if (h > 0) {                            // open an epoch: the pose halves and the epoch word, all under its tag
          const u64 tg = (u64)(epoch + 1) << 32;
          const u64 dj = (u64)__double_as_longlong(pp.cj), ds = (u64)__double_as_longlong(pp.sj);
          const u64 eh = (u64)__double_as_longlong(pp.ch), es = (u64)__double_as_longlong(pp.sh);
          st64(&C->pose[0], tg | (u64)__float_as_uint(pp.T.c));  st64(&C->pose[1], tg | (u64)__float_as_uint(pp.T.s));
          st64(&C->pose[2], tg | (u64)__float_as_uint(pp.T.tx)); st64(&C->pose[3], tg | (u64)__float_as_uint(pp.T.ty));
          st64(&C->pose[4], tg | (dj & 0xFFFFFFFFull));  st64(&C->pose[5], tg | (dj >> 32));
          st64(&C->pose[6], tg | (ds & 0xFFFFFFFFull));  st64(&C->pose[7], tg | (ds >> 32));
          st64(&C->pose[8], tg | (eh & 0xFFFFFFFFull));  st64(&C->pose[9], tg | (eh >> 32));
          st64(&C->pose[10], tg | (es & 0xFFFFFFFFull)); st64(&C->pose[11], tg | (es >> 32));
          st64(&C->ticket, tg | ((u64)h << 16) | ((u64)kUnits << 8) | (u64)kOwnerLead);
          if (kProf && prof && epoch - 1 < (unsigned)kProfPasses) {
            u64 *tl = prof + 32 * (size_t)B + ((size_t)b * kProfPasses + (epoch - 1)) * 16;
            tl[0] = wall_clock64(); tl[3] = (u64)h;
          }
        }
      }
      __syncthreads();
      if (kProf && prof) ts1 = wall_clock64();
      const int nhelp = L.sflag[0];
      if (nhelp <= 0) {
        // solo pass: wave w computes its own units (w, 0..kSub-1) in one walk
        pass_units<SSE, INCL, CHK>(wave, 0, kUnits, nullptr, 0u, 0);
      } else {
        // this workgroup's units 0 .. kOwnerLead-1 and kOwnerLead + j*(nhelp+1), handed to its waves from an LDS counter
        ++epoch;
        pass_h = nhelp;
        pass_units<SSE, INCL, CHK>(kOwnerLead, nhelp + 1, kUnits, nullptr, 0u, kOwnerLead);
        if (kProf && prof) ts2 = wall_clock64();
        // the helpers' units: every thread polls the words it will copy (24 per unit) until they carry this epoch's
        // tag (every counted helper is polling the epoch word or computing)
        u32 *const wp32 = reinterpret_cast<u32 *>(L.wpart);
        unsigned polls = 0;
        u64 w0 = 0;
        bool bad = false;
        for (int wi = threadIdx.x; wi < kUnits * kUnitWords && !bad; wi += kBlock) {
          const int u = wi / kUnitWords;
          if (u < kOwnerLead || (u - kOwnerLead) % (nhelp + 1) == 0) continue;      // own unit
#if NDT_POLL2
          u64 w = 0, w_next = ld64(&mytot[wi]);          // two polls in flight: half the time between looks
#else
          u64 w = 0;
#endif
          for (unsigned it = 0; it < 0x40000000u; ++it) {
#if NDT_POLL2
            w = w_next;
            w_next = ld64(&mytot[wi]);
#else
            w = ld64(&mytot[wi]);
#endif
            if ((u32)(w >> 32) == epoch) break;
            if ((++polls & 63u) == 0u) {           // bound on the spin, as in watchdog(); the clock is read lazily
              if (ld32(&hdr->abort)) { bad = true; break; }
              if (w0 == 0) w0 = wall_clock64();
              else if (wall_clock64() - w0 > kWatchTicks) { st32(&hdr->abort, 1u); bad = true; break; }
            }
            __builtin_amdgcn_s_sleep(1);
          }
          wp32[wi] = (u32)w;
        }
        if (bad) L.sflag[2] = 1;
        if (kProf && threadIdx.x == 0) { if (n_shared == 0) t_first_shared = ts2 - t_start; n_shared += 1; }
      }
      __syncthreads();
      if (kProf && prof) ts3 = wall_clock64();
      if (kProf && prof && threadIdx.x == 0 && pass_h > 0 && epoch - 2 < (unsigned)kProfPasses) {
        u64 *tl = prof + 32 * (size_t)B + ((size_t)b * kProfPasses + (epoch - 2)) * 16;
        tl[1] = ts2; tl[2] = ts3;
      }
      if (L.sflag[2]) { aborted = true; break; }
      // pass total: the units in kSub groups of 16, each summed in unit order by one lane per value,
      // then the partial sums in group order; wave 0 goes straight on to the optimiser step
      static_assert(kSub * 12 <= 64 && kUnits == 16 * kSub, "one lane per (group, value)");
      static_assert(sizeof(L.wpart) >= 4096, "the set-up phases keep 4 KiB of bitmaps in L.wpart");
      if (threadIdx.x < 64) {
        // (behind every pass with the whole workgroup waiting: the sixteen unit totals of a lane are read in one go -- the
        //  compiler had paired every read with a wait -- and the lane's place is worked out here each time: kept across
        //  the pass loop it lived in scratch memory, one reload at memory latency per pass)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int grp = (ln * 43) >> 9, j = ln - 12 * grp;    // lane / 12, lane % 12 for lanes 0..63; lanes kSub*12..63: nothing to add
        double part = 0.0;
        if (ln < kSub * 12) {
          double uv[16];
#pragma unroll
          for (int v = 0; v < 16; ++v) uv[v] = L.wpart[(16 * grp + v) * 12 + j];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int v = 0; v < 16; ++v) part += uv[v];
        }
        double total = part;
#pragma unroll
        for (int g = 1; g < kSub; ++g) total += __shfl(part, j + 12 * g);
        if (lane < 12) L.tot[lane] = total;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (kProf && prof) { tt1 = wall_clock64(); }
        if (lane == 0) advance(L.S, L.P, L.M, L.tot, tr, trace_cap, trace_rows ? trace_rows + b : nullptr);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int need_tf = L.S.need_tf;                       // (uniform: one LDS word)
        if (need_tf && lane < 2) trial_transforms(L.S, L.P, lane, need_tf);
      }
      // meanwhile another wave fetches the number of registered helpers for the next pass
      // (round 4: fetched a pass EARLIER -- while the pass computes, off this stretch -- the kernel is 5.5 % slower: a helper
      //  that has just registered must be counted in at once)
      if (threadIdx.x == 64 && allow_helpers) L.sflag[1] = (int)rd32_fresh(&C->ready);
      __syncthreads();
      if (kProf && prof) {
        const u64 te = wall_clock64();
        if (threadIdx.x >= 64) tt1 = te;          // (only wave 0 stamps the end of the summation)
        t_eval += tt1 - tt0;
        t_adv += te - tt1;
        if (pass_h > 0) { a_pro += ts1 - tt0; a_own += ts2 - ts1; a_wait += ts3 - ts2; a_comb += tt1 - ts3; a_adv += te - tt1; a_n += 1; }
      }
    }
    // ---- result record; close the scan ----
    if (threadIdx.x == 0) {
      const AlignState &S = L.S;
      const Tf32 T = S.T;
      ndt_result R_;
      R_.pose[0] = (double)T.tx; R_.pose[1] = (double)T.ty; R_.pose[2] = yaw_from_T(T.c, T.s);
      R_.T00 = T.c; R_.T10 = T.s; R_.T03 = T.tx; R_.T13 = T.ty;
      R_.fitness = DBL_MAX;                 // written by fitness_reduce_kernel, queued behind this kernel
      R_.score = S.score;
      R_.trans_prob = n > 0 ? S.score / (double)n : 0.0;
      R_.H[0] = S.H[0]; R_.H[1] = S.H[1]; R_.H[2] = S.H[2];
      R_.H[3] = S.H[1]; R_.H[4] = S.H[3]; R_.H[5] = S.H[4];
      R_.H[6] = S.H[2]; R_.H[7] = S.H[4]; R_.H[8] = S.H[5];
      R_.p[0] = S.p[0]; R_.p[1] = S.p[1]; R_.p[2] = S.p[2];
      R_.iters = S.iters; R_.evals = S.evals;
      R_.ref_evals = S.ref_evals + 1;       // + the getHessian pass (src/PoseEstimator.cpp:56)
      R_.converged = S.converged;
      R_.status = aborted ? NDT_E_HIP : (n > 0 ? NDT_OK : NDT_E_ARG);
      R_.flags = n > 0 ? ((L.RG.nspill > 0 ? NDT_FLAG_WINDOW_SPILL : 0) | (L.clipped ? NDT_FLAG_REGION_CLIPPED : 0) |
                          (pts == scan ? NDT_FLAG_UNSORTED : 0)) : 0;      // (an empty scan has no window: nothing left over from the scan before it)
      R_.kbar = (S.evals > 0 && n > 0) ? S.pairs / ((double)S.evals * (double)n) : 0.0;
      results[b] = R_;
      if (allow_helpers) {
        st64(&C->ticket, (u64)kEpochDone << 32);
        const float spp = n > 0 ? (float)(S.score / (double)n) : 0.f;
        if (S.converged && spp > 0.f) __hip_atomic_fetch_max(&hdr->best_spp, __float_as_uint(spp), NDT_RLX, NDT_AGENT);   // (positive floats order like their bits)
        __hip_atomic_fetch_add(&hdr->done, 1u, NDT_RLX, NDT_AGENT);
      }
      if (kProf && prof) {
        prof[8 * b + 0] = t_eval; prof[8 * b + 1] = t_adv | (t_fit << 32) | ((u64)(L.RG.nspill > 0) << 63); prof[8 * b + 2] = (t_first_shared << 32) | (t_scan0 & 0xFFFFFFFFull);
        prof[8 * b + 3] = (unsigned long long)S.evals | ((u64)n_shared << 16) | ((u64)n_helped << 32);
        prof[8 * b + 6] = t_wait; prof[8 * b + 7] = wall_clock64() - t_start;
        u64 *p2 = prof + 8 * (size_t)B + 8 * (size_t)b;
        p2[0] = a_n; p2[1] = a_pro; p2[2] = a_own; p2[3] = a_wait; p2[4] = a_comb; p2[5] = a_adv;
      }
    }
      // next scan of the batch, if any: from the queue; once that is used up, a scan nobody has claimed (see "Claims")
    __syncthreads();
    if (threadIdx.x == 0) { L.sflag[3] = (int)gridDim.x + (int)__hip_atomic_fetch_add(&hdr->next, 1u, NDT_RLX, NDT_AGENT); L.steal = INT_MAX; }
    __syncthreads();
    b = L.sflag[3];
    if (b >= B && allow_helpers && !aborted) {
      const int first = min(B, (int)gridDim.x);
      for (int k = threadIdx.x; k < first; k += kBlock)
        if (ld32(&ctl[k].claimed) == 0u) atomicMin(&L.steal, k);
      __syncthreads();
      if (threadIdx.x == 0) {
        int got = B;
        for (int tries = 0; tries < 4 && L.steal != INT_MAX && got == B; ++tries) {   // lost a race: the next free one, if any
          u32 expect = 0u;
          if (__hip_atomic_compare_exchange_strong(&ctl[L.steal].claimed, &expect, 1u, NDT_RLX, NDT_RLX, NDT_AGENT)) got = L.steal;
          else {
            int nx = INT_MAX;
            for (int k = L.steal + 1; k < first; ++k) if (ld32(&ctl[k].claimed) == 0u) { nx = k; break; }
            L.steal = nx;
          }
        }
        L.sflag[3] = got;
      }
      __syncthreads();
      b = L.sflag[3];
      preclaimed = b;
    }
  }

  // ============================================ helper ============================================
  if (!allow_helpers || aborted) return;
  u64 idle_ticks = 400;
  for (unsigned rounds = 0; rounds < 0x40000000u; ++rounds) {
    // ---- find an unfinished scan that still has room for a helper ----
    __syncthreads();
    if (threadIdx.x == 0) { L.sflag[0] = INT_MAX; L.sflag[3] = 0; L.sflag[2] = 0; }
    __syncthreads();
    const int start = (int)((blockIdx.x * 97u) % (unsigned)B);
    // helpers per scan: as many as the unfinished scans leave workgroups for (the last stragglers get
    // up to kMaxHelpers, a unit each per wave)
    const int unfinished = max(1, B - (int)ld32(&hdr->done));
    const int room = min(allow_helpers, max(min(allow_helpers, kBaseHelpers), (int)gridDim.x / unfinished - 1));
    const float best_spp = __uint_as_float(ld32(&hdr->best_spp));
    for (int k = threadIdx.x; k < B; k += kBlock) {
      int b = start + k; if (b >= B) b -= B;
      const u32 ep = (u32)(rd64_fresh(&ctl[b].ticket) >> 32);
      if (ep == 0u) { L.sflag[2] = 1; continue; }          // not open yet (its owner is setting up, or a workgroup that runs out of work will claim it): may need help later
      if (ep == kEpochDone) continue;
      const u32 h = rd32_fresh(&ctl[b].helpers);
      if (h >= (u32)room) continue;
      // Which scan has most left to do?  Its score per point says: against the best a FINISHED scan of this launch reached,
      // 1.00 / 0.97 / 0.8 / 0.7 / 0.5 / 0.4 of it go with 1 / 2 / 3 / 4 / 5 / 6-7 passes still to run (bench workload,
      // tools/pass_counts.py: the score alone explains 65-78 % of the variance of what is left; the passes run so far --
      // the rule until round 4 -- nothing once the repeated passes were gone).  In quarter passes; every attached helper
      // counts like NDT_HELPER_PENALTY passes fewer, an owner on this XCD like NDT_XCD_BONUS more; then nearest.
      int need4;
#ifndef NDT_NO_NEED_POLICY
      const float spp = __uint_as_float(ld32(&ctl[b].spp));
      if (best_spp > 0.f) need4 = (int)(4.f * fminf(fmaxf(1.f + (float)NDT_NEED_SLOPE * (1.f - spp / best_spp), 0.f), 12.f));
      else
#endif
        need4 = 4 * (int)min(ld32(&ctl[b].passes), 12u);
      const int score = need4 - 4 * NDT_HELPER_PENALTY * (int)h +
                        ((NDT_XCD_BONUS && ld32(&ctl[b].owner_xcd) == 1u + (blockIdx.x & 7u)) ? 4 * NDT_XCD_BONUS : 0);
      atomicMin(&L.sflag[0], (int)(((u32)(1024 - score) << 20) | (u32)k));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int code = -1;                                       // -1: nothing joinable right now
      if (ld32(&hdr->done) >= (u32)B || ld32(&hdr->abort)) code = -2;                // -2: leave
      // nothing to join, every open scan already has the most helpers a scan can get, none is still to
      // open: no work can come any more -- leave, so that the CU is free for whatever is queued next
      else if (L.sflag[0] == INT_MAX && L.sflag[2] == 0 && room >= allow_helpers) code = -2;
      else if (L.sflag[0] != INT_MAX) {
        int b = start + (L.sflag[0] & 0xFFFFF); if (b >= B) b -= B;
        const u32 h = __hip_atomic_fetch_add(&ctl[b].helpers, 1u, NDT_RLX, NDT_AGENT);
        if (h >= (u32)room) {
          __hip_atomic_fetch_sub(&ctl[b].helpers, 1u, NDT_RLX, NDT_AGENT);   // lost the race: look again
        } else {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // geometry + ordered copy of the owner
          drain_vmem();
          if (kProf && prof && h == 0) prof[8 * b + 4] = wall_clock64() - t_start;
          code = b;
        }
      }
      L.sflag[3] = code;
      if (code == -1) {                                    // back off: 4 us, doubling up to NDT_IDLE_MAX ticks
        const u64 t0 = wall_clock64();
        while (wall_clock64() - t0 < idle_ticks) __builtin_amdgcn_s_sleep(64);
        if (idle_ticks < NDT_IDLE_MAX) idle_ticks *= 2;
      } else {
        idle_ticks = 400;
      }
    }
    __syncthreads();
    const int vb = L.sflag[3];
    if (vb == -2) break;
    if (vb < 0) continue;
    // ---- attached to scan vb: stage its window, register, then serve its epochs until it is done ----
    ScanCtl *C = ctl + vb;
    const u64 o0 = shared_scan ? offsets[0] : offsets[vb];
    const u64 o1 = shared_scan ? offsets[1] : offsets[vb + 1];
    const int n = (int)(o1 - o0);
    if (threadIdx.x == 0) L.sflag[1] = (int)C->use_sorted;
    if (threadIdx.x == 0) {
      Region r; r.x0 = C->region[0]; r.y0 = C->region[1]; r.rw = C->region[2]; r.rh = C->region[3];
      r.cap = C->region[4]; r.nspill = C->region[5];
      L.RG = r;
    }
    {
      unsigned *wmap = reinterpret_cast<unsigned *>(L.wpart);
      const unsigned *gw = wantmap + (size_t)vb * (kRegionCells / 32);
      for (int i = threadIdx.x; i < kRegionCells / 32; i += kBlock) wmap[i] = gw[i];
    }
    __syncthreads();
    const float2 *pts = L.sflag[1] ? (shared_scan ? sorted + (size_t)vb * (size_t)n : sorted + o0)
                                   : (reinterpret_cast<const float2 *>(scans) + o0);
    fill_window(L.M, L, pool);
    if (threadIdx.x == 0) { L.pts = pts; L.npts = n; }
    if (kProf && prof && threadIdx.x == 0 && prof[8 * vb + 5] == 0) prof[8 * vb + 5] = wall_clock64() - t_start;
    u64 *vtot = utot + (size_t)vb * kUnits * kUnitWords;
    // register: from now on this workgroup does nothing but watch the scan's epoch word
    if (threadIdx.x == 0) L.hrank = (int)__hip_atomic_fetch_add(&C->ready, 1u, NDT_RLX, NDT_AGENT);
    __syncthreads();
    const int rank = L.hrank;
    u32 last_ep = 0;
    for (unsigned turns = 0; turns < 0x40000000u; ++turns) {         // counted (tools/repro/ticket2.hip)
      if (wave == 0) {
        // wave 0 polls line 0 of the scan's control block: lane 0 the epoch word, lanes 1..12 the pose halves.  A new
        // epoch that counts this helper in is taken once every pose word carries its tag as well.
        u64 word = 0, mine = 0;
        unsigned polls = 0;
        const u64 w0 = wall_clock64();
#if NDT_POLL2H
        u64 mine_next = 0;
        if (lane <= kPoseWords) mine_next = ld64(&C->ticket + lane);   // two polls in flight: half the time between looks
#endif
        for (unsigned it = 0; it < 0x40000000u; ++it) {
#if NDT_POLL2H
          mine = mine_next;
          if (lane <= kPoseWords) mine_next = ld64(&C->ticket + lane);
#else
          if (lane <= kPoseWords) mine = ld64(&C->ticket + lane);
#endif
          word = wave_bcast64(mine);
          const u32 ep = (u32)(word >> 32);
          if (ep != last_ep && ep != 0u) {
            if (ep == kEpochDone || rank >= (int)((word >> 16) & 0xFFu)) break;
            if (__builtin_amdgcn_ballot_w64(lane <= kPoseWords && (u32)(mine >> 32) != ep) == 0ull) break;
          }
          if (watchdog(hdr, w0, polls)) { word = (u64)kEpochDone << 32; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if ((u32)(word >> 32) != kEpochDone && rank < (int)((word >> 16) & 0xFFu) && lane >= 1 && lane <= kPoseWords)
          reinterpret_cast<u32 *>(&L.PP)[lane - 1] = (u32)mine;      // T.c, T.s, T.tx, T.ty, then the halves of cj, sj, ch, sh
        if (lane == 0) { L.hword = word; L.jnext = 0; }
        if (kProf && prof && lane == 0 && rank < 6 && rank < (int)((word >> 16) & 0xFFu) && (u32)(word >> 32) - 2u < (unsigned)kProfPasses)
          prof[32 * (size_t)B + ((size_t)vb * kProfPasses + ((u32)(word >> 32) - 2u)) * 16 + 4 + 2 * rank] = wall_clock64();
      }
      __syncthreads();
      const u64 word = L.hword;
      const u32 ep = (u32)(word >> 32);
      if (ep == kEpochDone) break;
      last_ep = ep;
      const int h = (int)((word >> 16) & 0xFFu), ubeg = (int)(word & 0xFFu), uend = (int)((word >> 8) & 0xFFu);
      if (rank < h) {
        // this workgroup's units ubeg + rank+1 + j*(h+1), handed to its waves from an LDS counter
        pass_units<SSE, INCL, CHK>(ubeg + (rank + 1), h + 1, uend, vtot, ep, 0);
      }
      __syncthreads();                                       // L.jnext / L.PP are rewritten by wave 0 in the next turn
      if (kProf && prof && threadIdx.x == 0 && rank < 6 && rank < h && ep - 2u < (unsigned)kProfPasses)
        prof[32 * (size_t)B + ((size_t)vb * kProfPasses + (ep - 2u)) * 16 + 5 + 2 * rank] = wall_clock64();
    }
  }
}

// One derivative pass at an explicit pose (tests / profiling): grid-stride over points,
// one partial record per workgroup, summed on the host in block order.
template <bool SSE, bool INCL>
__global__ void __launch_bounds__(256)
ndt_eval_kernel(MapView M, double snap, int libm_f32, const float *__restrict__ scan, size_t stride, int n,
                double p0, double p1, double p2, double *__restrict__ partial /* grid x kAcc */) {
  __shared__ double sred[(4 + 1) * kAcc];
  __shared__ double etab[64];
  __shared__ unsigned short no_slot[16];
  __shared__ CellEntry no_ent[1];
  if (threadIdx.x < 64) etab[threadIdx.x] = c_exp2_tab[threadIdx.x];
  if (threadIdx.x < 16) no_slot[threadIdx.x] = 0;
  if (threadIdx.x == 0) { no_ent[0].cent = make_float2(INFINITY, INFINITY); no_ent[0].mx = no_ent[0].my = 0; no_ent[0].i00 = no_ent[0].i01 = no_ent[0].i11 = 0; }
  __syncthreads();
  double p[3] = {p0, p1, p2};
  Tf32 T = tf_from_p(p, libm_f32);
  double cj, sj;
  angle_cs(snap, p2, cj, sj);
  Acc A = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0u};
  Window W;
  W.R = Region{0, 0, 0, 0, 0, 0}; W.slot = no_slot; W.ent = no_ent;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float2 pt = load_pt(scan, stride, i);
    eval_point<SSE, INCL>(M, W, etab, T, pt.x, pt.y, cj, sj, cj, sj, A);
  }
  block_reduce_acc(A, sred, sred + 4 * kAcc);
  if (threadIdx.x < kAcc) partial[blockIdx.x * kAcc + threadIdx.x] = sred[4 * kAcc + threadIdx.x];
}

__global__ void __launch_bounds__(256)
ndt_fitness_kernel(MapView M, const float *__restrict__ scan, size_t stride, int n, Tf32 T,
                   double *__restrict__ partial /* grid x 2 */) {
  __shared__ double sred[(4 + 1) * 2];
  double fsum = 0.0, fcnt = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float2 pt = load_pt(scan, stride, i);
    float qx, qy;
    tf_apply(T, M.transform_sse, pt.x, pt.y, qx, qy);
    if (!finite2(qx, qy)) continue;
    float best = nearest_sq(M, qx, qy);
    if (best < INFINITY) { fsum += (double)best; fcnt += 1.0; }
  }
  block_reduce2(fsum, fcnt, sred, sred + 4 * 2);
  if (threadIdx.x < 2) partial[blockIdx.x * 2 + threadIdx.x] = sred[4 * 2 + threadIdx.x];
}
