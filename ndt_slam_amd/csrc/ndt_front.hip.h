// ndt_front.hip.h -- rows f1-f3: source pre-filter, prediction / fusion around the match, neighbour removal.
// Part of libndt_mi355x.so: included by ndt_mi355x.hip inside its anonymous namespace (one translation
// unit; the order of the includes matters).  Not a standalone header.

// ------------------------------------------------------------------------------------------
// f1: source pre-filter = pcl::ApproximateVoxelGrid::filter on a z = 0 cloud
// (src/PoseEstimator.cpp:6-10).  The filter is a sequential machine: 512
// direct-mapped slots, a point either joins the voxel its slot holds or flushes that voxel's
// centroid to the output and takes the slot; what is left is flushed in slot order at the end.
// One wave per scan replays it 64 points at a time: lanes whose points hash to different slots
// update them at once, lanes sharing a slot take turns in point order (the float32 sums of a slot
// are therefore added in cloud order), and the flushes of a step are written in point order.
// Output = the reference's output, bit for bit and in the same order.
// ------------------------------------------------------------------------------------------
constexpr int kPfSlots = 512;
struct __attribute__((aligned(8))) PfSlot { int ix, iy, cnt; float cx, cy; int pad; };   // 24 B: one b128 + one b64 LDS read
#ifndef NDT_PF_SORTED
#define NDT_PF_SORTED 1   // 0: every scan through the step-by-step replay (A/B)
#endif
#ifndef NDT_PF_AHEAD
#define NDT_PF_AHEAD 8
#endif
#ifndef NDT_PF_WAVE_BITS
#define NDT_PF_WAVE_BITS 3
#endif
constexpr int kPfWaveBits = NDT_PF_WAVE_BITS, kPfWaves = 1 << kPfWaveBits, kPfQueue = 128, kPfMaxPoints = 1 << 18;   // bitmap: 32 KiB of LDS
__global__ void __launch_bounds__(64 * kPfWaves)
prefilter_mw_kernel(const float *__restrict__ xy, size_t stride, const unsigned long long *__restrict__ offsets, int B,
                    float leaf, float2 *__restrict__ sparse /* at the raw offsets */,
                    float2 *__restrict__ tmp /* at the raw offsets: dense result */, unsigned *__restrict__ counts,
                    int skip_up_to /* scans of at most this many points are another kernel's */) {
  __shared__ PfSlot slot[kPfSlots];
  __shared__ float2 qpt[kPfWaves][kPfQueue];
  __shared__ int qidx[kPfWaves][kPfQueue];
  __shared__ unsigned fbits[kPfMaxPoints / 32];
  __shared__ int wsum[kPfWaves + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const float inv = 1.0f / leaf;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const unsigned long long o0 = offsets[b];
    const int n = (int)(offsets[b + 1] - o0);
    if (n <= skip_up_to) continue;                           // (uniform over the workgroup)
    __syncthreads();
    for (int h = threadIdx.x; h < kPfSlots; h += 64 * kPfWaves) { PfSlot z; z.ix = 0; z.iy = 0; z.cnt = 0; z.cx = 0.f; z.cy = 0.f; z.pad = 0; slot[h] = z; }
    const int nwords = (min(n, kPfMaxPoints) + 31) / 32;
    for (int i = threadIdx.x; i < nwords; i += 64 * kPfWaves) fbits[i] = 0u;
    __syncthreads();
    int nout = 0;
    if (n <= kPfMaxPoints) {
      // replay of one step: `cnt` queued points of this wave's slots, in point order
      auto replay = [&](int cnt) {
        const bool active = lane < cnt;
        const float2 p = active ? qpt[w][lane] : make_float2(0.f, 0.f);
        const int i = active ? qidx[w][lane] : 0;
        const int ix = (int)floorf(p.x * inv), iy = (int)floorf(p.y * inv);
        const unsigned h = ((unsigned)ix * 7171u + (unsigned)iy * 3079u) & (unsigned)(kPfSlots - 1);
        unsigned long long peers = __ballot(active);
#pragma unroll
        for (int bit = kPfWaveBits; bit < 9; ++bit) {     // the low bits are this wave's number
          const unsigned long long one = __ballot(active && ((h >> bit) & 1u));
          peers &= ((h >> bit) & 1u) ? one : ~one;
        }
        const int rank = __builtin_popcountll(peers & lt);
        for (int turn = 0; turn < 64; ++turn) {
          if (!__ballot(active && rank >= turn)) break;
          if (active && rank == turn) {
            PfSlot e = slot[h];
            if (e.cnt && (ix != e.ix || iy != e.iy)) {
              sparse[o0 + (unsigned long long)i] = make_float2(e.cx / (float)e.cnt, e.cy / (float)e.cnt);
              atomicOr(&fbits[i >> 5], 1u << (i & 31));
              e.cnt = 0; e.cx = 0.f; e.cy = 0.f;
            }
            e.ix = ix; e.iy = iy; e.cnt += 1; e.cx += p.x; e.cy += p.y;
            slot[h] = e;
          }
          __builtin_amdgcn_wave_barrier();
        }
      };
      int qn = 0;
      // the read-hash-queue pass over the whole scan: kPfAhead steps of 64 points are in flight (one step ahead left
      // every step waiting for its load: 0.6 us a step, most of the kernel)
      constexpr int kPfAhead = NDT_PF_AHEAD;
      float2 pnext[kPfAhead];
#pragma unroll
      for (int u = 0; u < kPfAhead; ++u) {
        const int i = 64 * u + lane;
        pnext[u] = i < n ? load_pt(xy, stride, (size_t)o0 + (size_t)i) : make_float2(0.f, 0.f);
      }
      for (int base0 = 0; base0 < n; base0 += 64 * kPfAhead) {
        float2 pcur[kPfAhead];
#pragma unroll
        for (int u = 0; u < kPfAhead; ++u) pcur[u] = pnext[u];
#pragma unroll
        for (int u = 0; u < kPfAhead; ++u) {
          const int i = base0 + 64 * (kPfAhead + u) + lane;
          pnext[u] = i < n ? load_pt(xy, stride, (size_t)o0 + (size_t)i) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kPfAhead; ++u) {
          const int base = base0 + 64 * u;
          if (base >= n) break;
          const int i = base + lane;
          const float2 p = pcur[u];
          const int ix = (int)floorf(p.x * inv), iy = (int)floorf(p.y * inv);
          const unsigned h = ((unsigned)ix * 7171u + (unsigned)iy * 3079u) & (unsigned)(kPfSlots - 1);
          const bool mine = i < n && (int)(h & (unsigned)(kPfWaves - 1)) == w;
          const unsigned long long mb = __ballot(mine);
          if (mine) { const int pos = qn + __builtin_popcountll(mb & lt); qpt[w][pos] = p; qidx[w][pos] = i; }
          qn += __builtin_popcountll(mb);
          __builtin_amdgcn_wave_barrier();
          if (qn >= 64) {
            replay(64);
            const int rest = qn - 64;                           // < 64: move the tail to the front
            float2 tp = make_float2(0.f, 0.f); int ti = 0;
            if (lane < rest) { tp = qpt[w][64 + lane]; ti = qidx[w][64 + lane]; }
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) { qpt[w][lane] = tp; qidx[w][lane] = ti; }
            __builtin_amdgcn_wave_barrier();
            qn = rest;
          }
        }
      }
      if (qn > 0) replay(qn);
      __syncthreads();
      // flushes in the order of the points that caused them: prefix sum over the bitmap
      int mine_cnt = 0;
      const int per = (nwords + 64 * kPfWaves - 1) / (64 * kPfWaves);
      const int w0 = min((int)threadIdx.x * per, nwords), w1 = min(w0 + per, nwords);
      for (int k = w0; k < w1; ++k) mine_cnt += __builtin_popcount(fbits[k]);
      int incl = mine_cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
      if (lane == 63) wsum[w] = incl;
      __syncthreads();
      int off = incl - mine_cnt;
      for (int k = 0; k < w; ++k) off += wsum[k];
      int total = 0;
      for (int k = 0; k < kPfWaves; ++k) total += wsum[k];
      for (int k = w0; k < w1; ++k) {
        unsigned bits = fbits[k];
        while (bits) {
          const int bpos = __builtin_ctz(bits); bits &= bits - 1;
          tmp[o0 + (unsigned long long)off] = sparse[o0 + (unsigned long long)(k * 32 + bpos)];
          ++off;
        }
      }
      nout = total;
      __syncthreads();
    } else if (w == 0) {
      // a scan too long for the bitmap: one wave, the plain replay
      float2 pnext = make_float2(0.f, 0.f);
      if (lane < n) pnext = load_pt(xy, stride, (size_t)o0 + (size_t)lane);
      for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool active = i < n;
        const float2 p = pnext;
        if (i + 64 < n) pnext = load_pt(xy, stride, (size_t)o0 + (size_t)(i + 64));
        const int ix = (int)floorf(p.x * inv), iy = (int)floorf(p.y * inv);
        const unsigned h = ((unsigned)ix * 7171u + (unsigned)iy * 3079u) & (unsigned)(kPfSlots - 1);
        unsigned long long peers = __ballot(active);
#pragma unroll
        for (int bit = 0; bit < 9; ++bit) {
          const unsigned long long one = __ballot(active && ((h >> bit) & 1u));
          peers &= ((h >> bit) & 1u) ? one : ~one;
        }
        const int rank = __builtin_popcountll(peers & lt);
        bool flushed = false; float fx = 0.f, fy = 0.f;
        for (int turn = 0; turn < 64; ++turn) {
          if (!__ballot(active && rank >= turn)) break;
          if (active && rank == turn) {
            PfSlot e = slot[h];
            if (e.cnt && (ix != e.ix || iy != e.iy)) { flushed = true; fx = e.cx / (float)e.cnt; fy = e.cy / (float)e.cnt; e.cnt = 0; e.cx = 0.f; e.cy = 0.f; }
            e.ix = ix; e.iy = iy; e.cnt += 1; e.cx += p.x; e.cy += p.y;
            slot[h] = e;
          }
          __builtin_amdgcn_wave_barrier();
        }
        const unsigned long long fb = __ballot(flushed);
        if (flushed) tmp[o0 + (unsigned long long)(nout + __builtin_popcountll(fb & lt))] = make_float2(fx, fy);
        nout += __builtin_popcountll(fb);
      }
    }
    __syncthreads();
    if (w == 0) {                                            // what is left, in slot order
      if (n > kPfMaxPoints) nout = __builtin_amdgcn_readfirstlane(nout);
      for (int h0 = 0; h0 < kPfSlots; h0 += 64) {
        const PfSlot e = slot[h0 + lane];
        const unsigned long long fb = __ballot(e.cnt > 0);
        if (e.cnt > 0) tmp[o0 + (unsigned long long)(nout + __builtin_popcountll(fb & lt))] =
            make_float2(e.cx / (float)e.cnt, e.cy / (float)e.cnt);
        nout += __builtin_popcountll(fb);
      }
      if (lane == 0) counts[b] = (unsigned)nout;
    }
  }
}

// The same filter for scans of up to kPfSortMax points, organised by SLOT instead of by step.  The 512 slots are
// independent state machines, and what a slot does depends only on the points that hash to it, in cloud order: so the
// points are first ordered by slot -- a stable counting sort of their numbers in LDS: every wave counts and later places
// the points of its own contiguous part of the scan, step by step in cloud order, a lane's turn among equal slots inside
// a step from ballots -- and then thread h walks the points of slot h with the slot's state in registers, one after the
// other, its loads in flight ahead of it.  No lane ever waits for its turn on a slot, and every point is looked at by
// one lane per pass instead of by each of eight waves.  The scan goes through LDS in tiles of kPfTile points (read once,
// coalesced); a flush is parked in LDS in the place of the point that caused it and marked in the tile's bitmap; a
// prefix sum over the bitmap later, the flushes of the tile -- the r-th marked point is flush number r: the order the
// sequential filter emits them in -- are divided by their counts and stored.  The slots left over follow in slot
// order -- the reference's output, bit for bit (tests/test_gpu_prefilter.py).
constexpr int kPfSortMax = 65535;      // the count of a voxel's points travels as 16 bits (fcount): a scan must not have more
constexpr int kPfSortThreads = 512, kPfSortWaves = kPfSortThreads / 64;
constexpr int kPfTile = 8192;                                    // points of a scan staged in LDS at a time
constexpr int kPfStepsPerWave = kPfTile / 64 / kPfSortWaves;      // steps of 64 points a wave counts and places in a tile
__global__ void __launch_bounds__(kPfSortThreads)
prefilter_sorted_kernel(const float *__restrict__ xy, size_t stride, const unsigned long long *__restrict__ offsets, int B,
                        float leaf, float2 *__restrict__ tmp /* at the raw offsets: dense result */,
                        unsigned *__restrict__ counts) {
  __shared__ float2 pts[kPfTile];                               // the tile of the scan being worked on
  __shared__ unsigned short order[kPfTile];                     // its point numbers (in the tile), ordered by slot, cloud order inside a slot
  __shared__ unsigned short wcount[kPfSortWaves][kPfSlots];      // count: points of wave w's part in slot h; then: where they go
  __shared__ int sbase[kPfSlots + 1];                           // first position of slot h in `order`
  __shared__ PfSlot slot[kPfSlots];
  __shared__ unsigned short fcount[kPfTile];                    // flush parked at point i of the tile: points of the voxel it closes
  __shared__ unsigned fbits[kPfTile / 32];                      // point i of the tile causes a flush
  __shared__ unsigned short fpre[kPfTile / 32];                 // marked points of the tile in front of word k
  __shared__ unsigned long long same[kPfSortWaves][kPfSlots];   // per wave: lanes of the current step that hash to slot h
  __shared__ int wsum[kPfSortWaves + 1];
  static_assert(kPfSortThreads == kPfSlots, "one thread per slot");
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const float inv = 1.0f / leaf;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const unsigned long long o0 = offsets[b];
    const int n = (int)(offsets[b + 1] - o0);
    if (n > kPfSortMax || !NDT_PF_SORTED) continue;            // prefilter_mw_kernel's (uniform over the workgroup)
    __syncthreads();
    for (int i = threadIdx.x; i < kPfSortWaves * kPfSlots; i += kPfSortThreads) (&same[0][0])[i] = 0ull;
    auto slot_of = [&](float2 p) {
      const int ix = (int)floorf(p.x * inv), iy = (int)floorf(p.y * inv);
      return ((unsigned)ix * 7171u + (unsigned)iy * 3079u) & (unsigned)(kPfSlots - 1);
    };
    // the lanes of a step that share a slot: every lane ORs its bit into the wave's mask word of its slot (an LDS atomic;
    // lanes of one slot take turns in the LDS unit, a few of them in a scan a LiDAR made), reads the word back, and the
    // first lane of every group clears it again.  (Nine ballots over the bits of the slot number -- what the step-by-step
    // kernel does -- cost 70 instructions a step, most of them waiting on each other: 35 us a pass against 18.)
    auto peers_of = [&](bool active, unsigned h) {
      unsigned long long *word = &same[w][h];
      if (active) atomicOr(word, 1ull << lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned long long peers = active ? *word : 0ull;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (active && (peers & lt) == 0ull) *word = 0ull;
      return peers;
    };
    // slot h, this thread: its state lives in registers over the whole scan
    const int h = threadIdx.x;
    int six = 0, siy = 0, cnt = 0; float cx = 0.f, cy = 0.f;
    int nout = 0;                                              // flushes written so far (uniform)
    constexpr int kPfPerThread = kPfTile / kPfSortThreads;      // points of a tile a thread brings in
    float2 ahead[kPfPerThread];                                // the next tile, in flight while this one is worked on
#pragma unroll
    for (int u = 0; u < kPfPerThread; ++u) {
      const int i = (int)threadIdx.x + u * kPfSortThreads;
      ahead[u] = i < n ? load_pt(xy, stride, (size_t)o0 + (size_t)i) : make_float2(0.f, 0.f);
    }
    for (int base = 0; base < n; base += kPfTile) {
      const int m = min(kPfTile, n - base);
      __syncthreads();                                         // the previous tile has been walked
      // ---- the tile into LDS: the only time the scan is read (coalesced); everything below works from LDS
#pragma unroll
      for (int u = 0; u < kPfPerThread; ++u) {
        const int i = (int)threadIdx.x + u * kPfSortThreads;
        if (i < m) pts[i] = ahead[u];
      }
#pragma unroll
      for (int u = 0; u < kPfPerThread; ++u) {
        const int i = base + kPfTile + (int)threadIdx.x + u * kPfSortThreads;
        ahead[u] = i < n ? load_pt(xy, stride, (size_t)o0 + (size_t)i) : make_float2(0.f, 0.f);
      }
      for (int i = threadIdx.x; i < kPfSortWaves * kPfSlots; i += kPfSortThreads) (&wcount[0][0])[i] = 0;
      for (int i = threadIdx.x; i < kPfTile / 32; i += kPfSortThreads) fbits[i] = 0u;
      __syncthreads();
      // the part of the tile this wave counts and places: whole steps of 64 points
      const int steps = (m + 63) / 64, spw = (steps + kPfSortWaves - 1) / kPfSortWaves;
      const int s0 = min(w * spw, steps), s1 = min(s0 + spw, steps);
      // ---- count: how many points of this wave's part fall into every slot.  What a lane learns about its point here --
      // slot, turn among the lanes of the step with the same slot, size of that group -- is kept in a register per step
      // for the placing pass (which then needs no second look at the mask words)
      unsigned memo[kPfStepsPerWave];
#pragma unroll
      for (int j = 0; j < kPfStepsPerWave; ++j) {
        const int sg = s0 + j;
        memo[j] = 0xFFFFFFFFu;
        if (sg < s1) {                                         // (uniform over the wave)
          const int i = sg * 64 + lane;
          const bool active = i < m;
          const unsigned hh = slot_of(active ? pts[i] : make_float2(0.f, 0.f));
          const unsigned long long peers = peers_of(active, hh);
          const unsigned turn = (unsigned)__builtin_popcountll(peers & lt), size = (unsigned)__builtin_popcountll(peers);
          if (active && turn == 0u) wcount[w][hh] = (unsigned short)(wcount[w][hh] + size);   // first of its group
          if (active) memo[j] = hh | (turn << 9) | (size << 16);
          __builtin_amdgcn_wave_barrier();
        }
      }
      __syncthreads();
      // ---- slot h: its total in the tile, the offsets of the waves' shares inside it, the exclusive scan of the totals
      {
        int run = 0;
#pragma unroll
        for (int k = 0; k < kPfSortWaves; ++k) { const int c = wcount[k][h]; wcount[k][h] = (unsigned short)run; run += c; }
        int incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int sb = incl - run;
        for (int k = 0; k < w; ++k) sb += wsum[k];
        sbase[h] = sb;
        if (h == kPfSlots - 1) sbase[kPfSlots] = sb + run;
      }
      __syncthreads();
      // ---- place: every point's number to its place (slot base + the wave's offset in the slot + turn in the step)
#pragma unroll
      for (int j = 0; j < kPfStepsPerWave; ++j) {
        const int sg = s0 + j;
        if (sg < s1) {                                         // (uniform over the wave)
          const bool active = memo[j] != 0xFFFFFFFFu;
          const unsigned hh = memo[j] & 511u, turn = (memo[j] >> 9) & 127u, size = memo[j] >> 16;
          int at = 0;
          if (active) at = sbase[hh] + wcount[w][hh];
          __builtin_amdgcn_wave_barrier();                     // all lanes of the group have read the wave's offset
          if (active) {
            order[at + (int)turn] = (unsigned short)(sg * 64 + lane);
            if (turn == 0u) wcount[w][hh] = (unsigned short)(wcount[w][hh] + size);
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      __syncthreads();
      // ---- slot h, one thread: its points of the tile in cloud order, from LDS.  A flush is parked in the place of the
      // point that caused it (the point has been read: its place is free) -- the sums, and the count beside it; the
      // division is the copy's, a thread per run of flushes there, one lane after the other here
      for (int k = sbase[h]; k < sbase[h + 1]; ++k) {
        const int li = order[k];
        const float2 p = pts[li];
        const int ix = (int)floorf(p.x * inv), iy = (int)floorf(p.y * inv);
        if (cnt && (ix != six || iy != siy)) {
          pts[li] = make_float2(cx, cy);
          fcount[li] = (unsigned short)cnt;
          atomicOr(&fbits[li >> 5], 1u << (li & 31));
          cnt = 0; cx = 0.f; cy = 0.f;
        }
        six = ix; siy = iy; cnt += 1; cx += p.x; cy += p.y;
      }
      __syncthreads();
      // ---- the flushes of the tile in the order of the points that caused them (every flush of this tile comes before
      // every flush of the next): prefix sum over the tile's bitmap, then thread t divides and stores the flushes of word t
      {
        const int twords = (m + 31) / 32;                      // <= 256
        const int mine = (int)threadIdx.x < twords ? __builtin_popcount(fbits[threadIdx.x]) : 0;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int r = nout + incl - mine;
        for (int k = 0; k < w; ++k) r += wsum[k];
        int tile_total = 0;
        for (int k = 0; k < kPfSortWaves; ++k) tile_total += wsum[k];
        if ((int)threadIdx.x < twords) {
          unsigned bits = fbits[threadIdx.x];
          while (bits) {
            const int li = (int)threadIdx.x * 32 + __builtin_ctz(bits); bits &= bits - 1u;
            const float2 sum = pts[li]; const float c = (float)fcount[li];
            tmp[o0 + (unsigned long long)r] = make_float2(sum.x / c, sum.y / c);
            ++r;
          }
        }
        nout += tile_total;
      }
    }
    { PfSlot z; z.ix = six; z.iy = siy; z.cnt = cnt; z.cx = cx; z.cy = cy; z.pad = 0; slot[h] = z; }
    __syncthreads();
    if (w == 0) {                                              // what is left, in slot order
      for (int h0 = 0; h0 < kPfSlots; h0 += 64) {
        const PfSlot e = slot[h0 + lane];
        const unsigned long long fb = __ballot(e.cnt > 0);
        if (e.cnt > 0) tmp[o0 + (unsigned long long)(nout + __builtin_popcountll(fb & lt))] =
            make_float2(e.cx / (float)e.cnt, e.cy / (float)e.cnt);
        nout += __builtin_popcountll(fb);
      }
      if (lane == 0) counts[b] = (unsigned)nout;
    }
  }
}

// offsets of the filtered scans: exclusive scan of the counts (one workgroup)
__global__ void __launch_bounds__(1024)
prefilter_offsets_kernel(const unsigned *__restrict__ counts, int B, unsigned long long *__restrict__ out_offsets) {
  __shared__ unsigned long long sh[1024];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 1024) {
    const int i = base + threadIdx.x;
    const unsigned long long v = i < B ? counts[i] : 0ull;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const unsigned long long t = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0ull;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < B) out_offsets[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) out_offsets[B] = carry;
}

// filtered points from their raw offsets to the packed output
__global__ void __launch_bounds__(256)
prefilter_pack_kernel(const float2 *__restrict__ tmp, const unsigned long long *__restrict__ raw_offsets,
                      const unsigned long long *__restrict__ out_offsets, int B, float2 *__restrict__ out) {
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const unsigned long long r0 = raw_offsets[b], q0 = out_offsets[b];
    const unsigned long long n = out_offsets[b + 1] - q0;
    for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < n;
         j += (unsigned long long)gridDim.x * blockDim.x)
      out[q0 + j] = tmp[r0 + j];
  }
}


// ------------------------------------------------------------------------------------------
// f2: the steps either side of the match for a batch -- odometry prediction (Pose2D::calMotion +
// calPredPose, src/Pose2D.cpp:5-37, chained as in src/ScanMatcher.cpp:27-32) and, after the
// match, cost / NDT covariance (src/PoseEstimator.cpp:43-64), the accept test
// (src/ScanMatcher.cpp:50) and the EKF fusion or the odometry covariance alone
// (src/PoseFuser.cpp:3-61).  One lane per match, fp64, the oracle's expression order.
// Poses are (tx, ty, th) with th in degrees (include/ndt_slam/Pose2D.h:14).
// ------------------------------------------------------------------------------------------
struct FuseParams { double coe_ndt_cov, coe_vel, coe_omega, del_time, score_thre; };
__device__ __forceinline__ double f2_deg2rad(double x) { return x * M_PI / 180; }
__device__ __forceinline__ double f2_rad2deg(double x) { return x * 180 / M_PI; }
__device__ __forceinline__ double f2_add_angle(double a1, double a2) {
  double sum = a1 + a2;
  if (sum < -180) sum += 360; else if (sum >= 180) sum -= 360;
  return sum;
}
__device__ __forceinline__ double f2_sub_angle(double a1, double a2) {
  double dif = a1 - a2;
  if (dif < -180) dif += 360; else if (dif >= 180) dif -= 360;
  return dif;
}
__device__ __forceinline__ void f2_inv3(const double m[9], double out[9]) {   // Eigen's fixed 3x3 inverse
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c10 = m[2] * m[7] - m[1] * m[8];
  const double c20 = m[1] * m[5] - m[2] * m[4];
  const double det = c00 * m[0] + c10 * m[3] + c20 * m[6];
  const double id = 1.0 / det;
  out[0] = c00 * id; out[1] = c10 * id; out[2] = c20 * id;
  out[3] = (m[5] * m[6] - m[3] * m[8]) * id;
  out[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  out[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  out[6] = (m[3] * m[7] - m[4] * m[6]) * id;
  out[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  out[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
__device__ __forceinline__ void f2_mul3(const double a[9], const double b[9], double o[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = a[3 * i] * b[j];
      s += a[3 * i + 1] * b[3 + j];
      s += a[3 * i + 2] * b[6 + j];
      o[3 * i + j] = s;
    }
}
__device__ __forceinline__ void f2_odo_cov(const double motion[3], const double last[3], const double last_cov[9],
                                           const FuseParams &p, double cov[9]) {
  const double dt = p.del_time;
  const double v = sqrt(motion[0] * motion[0] + motion[1] * motion[1]) / dt;
  const double omega = f2_deg2rad(motion[2] / dt);
  const double m00 = p.coe_vel * v * v, m11 = p.coe_omega * omega * omega;
  const double a = f2_deg2rad(last[2]), c = cos(a), s = sin(a);
  const double F[9] = {1, 0, -v * dt * s, 0, 1, v * dt * c, 0, 0, 1};
  const double Ft[9] = {1, 0, 0, 0, 1, 0, F[2], F[5], 1};
  double t[9], flf[9];
  f2_mul3(F, last_cov, t); f2_mul3(t, Ft, flf);
  const double a0 = dt * c, a1 = dt * s;
  const double ama[9] = {a0 * m00 * a0, a0 * m00 * a1, 0, a1 * m00 * a0, a1 * m00 * a1, 0, 0, 0, dt * m11 * dt};
#pragma unroll
  for (int i = 0; i < 9; ++i) cov[i] = flf[i] + ama[i];
}

__global__ void __launch_bounds__(256)
predict_kernel(const double *__restrict__ odo_cur, const double *__restrict__ odo_prev,
               const double *__restrict__ last_pose, int B, double *__restrict__ motion_out,
               double *__restrict__ pred_out, double *__restrict__ init_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double *cur = odo_cur + 3 * b, *prev = odo_prev + 3 * b, *last = last_pose + 3 * b;
  const double ap = f2_deg2rad(prev[2]), cp = cos(ap), sp = sin(ap);
  const double dx = cur[0] - prev[0], dy = cur[1] - prev[1];
  double motion[3];
  motion[0] = cp * dx + sp * dy;
  motion[1] = -sp * dx + cp * dy;
  motion[2] = f2_sub_angle(cur[2], prev[2]);
  const double al = f2_deg2rad(last[2]), cl = cos(al), sl = sin(al);
  double pred[3];
  pred[0] = cl * motion[0] + -sl * motion[1] + last[0];
  pred[1] = sl * motion[0] + cl * motion[1] + last[1];
  pred[2] = f2_add_angle(last[2], motion[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) { motion_out[3 * b + i] = motion[i]; pred_out[3 * b + i] = pred[i]; }
  if (init_out) {                               // the guess ndt_align takes (src/PoseEstimator.cpp:22-24)
    init_out[3 * b] = pred[0]; init_out[3 * b + 1] = pred[1]; init_out[3 * b + 2] = f2_deg2rad(pred[2]);
  }
}

__global__ void __launch_bounds__(256)
fuse_kernel(const ndt_result *__restrict__ res, const double *__restrict__ pred_pose,
            const double *__restrict__ odo_motion, const double *__restrict__ last_pose,
            const double *__restrict__ last_cov, int B, FuseParams p, double *__restrict__ fused_out,
            double *__restrict__ cov_out, int *__restrict__ successful_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const ndt_result r = res[b];
  double pred[3], motion[3], last[3], lc[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) { pred[i] = pred_pose[3 * b + i]; motion[i] = odo_motion[3 * b + i]; last[i] = last_pose[3 * b + i]; }
#pragma unroll
  for (int i = 0; i < 9; ++i) lc[i] = last_cov[9 * b + i];
  const double est[3] = {r.pose[0], r.pose[1], f2_rad2deg(r.pose[2])};
  const double cost = (r.status == NDT_OK && r.converged) ? r.fitness : 10000000.0;
  const int successful = cost <= p.score_thre;
  double fused[3], cov[9];
  if (!successful) {
    f2_odo_cov(motion, last, lc, p, cov);
    fused[0] = pred[0]; fused[1] = pred[1]; fused[2] = pred[2];
  } else {
    double nh[9], Q[9], ch[9], sum[9], inv[9], K[9], imk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) nh[i] = -r.H[i];
    f2_inv3(nh, Q);
#pragma unroll
    for (int i = 0; i < 9; ++i) Q[i] *= p.coe_ndt_cov;
    f2_odo_cov(motion, last, lc, p, ch);
#pragma unroll
    for (int i = 0; i < 9; ++i) sum[i] = Q[i] + ch[i];
    f2_inv3(sum, inv);
    f2_mul3(ch, inv, K);
#pragma unroll
    for (int i = 0; i < 9; ++i) imk[i] = ((i % 4 == 0) ? 1.0 : 0.0) - K[i];
    f2_mul3(imk, ch, cov);
    const double zh[3] = {est[0] - pred[0], est[1] - pred[1], f2_deg2rad(f2_sub_angle(est[2], pred[2]))};
    const double mu_hat[3] = {pred[0], pred[1], f2_deg2rad(pred[2])};
    double mu[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double s = K[3 * i] * zh[0];
      s += K[3 * i + 1] * zh[1];
      s += K[3 * i + 2] * zh[2];
      mu[i] = s + mu_hat[i];
    }
    fused[0] = mu[0]; fused[1] = mu[1]; fused[2] = f2_rad2deg(mu[2]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) fused_out[3 * b + i] = fused[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) cov_out[9 * b + i] = cov[i];
  if (successful_out) successful_out[b] = successful;
}


// ------------------------------------------------------------------------------------------
// f3 (the all-pairs step on its own): PCFilter::remove_neighborPoint (include/ndt_slam/PCFilter.h:29-56) -- keep the points
// of `base` that have no point of `list` closer than thre_neighbor, in input order.  The
// reference tests every pair (O(n*m) on the CPU, the largest cost outside NDT when moving
// objects are removed, src/PointCloudMap.cpp:15-39); here one lane per base point walks the list
// through LDS tiles.  The distance is PCLUtil::distance_points' float32 expression
// (include/ndt_slam/PCLUtil.h:21-23) compared with the double threshold (through rn_cutoff), so the
// kept set is identical; a ballot prefix keeps the order.
// ------------------------------------------------------------------------------------------
constexpr int kRnBlock = 256, kRnTile = 1024;
// The reference's test is (double)sqrtf(d2) < thre with d2 the float32 squared distance.  sqrtf is correctly
// rounded and monotone, so the test equals d2 < cut with cut the smallest float whose square root reaches
// thre; rn_cutoff finds it on the host (bisection over the float bit patterns), and the device needs no sqrt.
inline float rn_cutoff(double thre) {
  if (!(thre > 0.0)) return 0.0f;                                   // nothing is closer than a non-positive bound
  if (!((double)sqrtf(FLT_MAX) >= thre)) return INFINITY;           // every finite distance is
  unsigned lo = 0u, hi = 0x7f7fffffu;                               // float bits: (double)sqrtf(hi) >= thre holds
  while (lo < hi) {
    const unsigned mid = lo + (hi - lo) / 2;
    float f; memcpy(&f, &mid, 4);
    if ((double)sqrtf(f) >= thre) hi = mid; else lo = mid + 1;
  }
  float f; memcpy(&f, &lo, 4);
  return f;
}
__device__ __forceinline__ bool rn_near(float2 p, float2 q, float cut) {
  const float dx = p.x - q.x, dy = p.y - q.y;
  const float d2 = dx * dx + dy * dy;            // (+ dz*dz with dz = 0 adds nothing)
  return d2 < cut;
}
__global__ void __launch_bounds__(kRnBlock)
remove_neighbors_flag_kernel(const float *__restrict__ base, size_t bstride, int nb, const float *__restrict__ list,
                             size_t lstride, int nl, float cut, unsigned char *__restrict__ keep,
                             int *__restrict__ block_count) {
  __shared__ float2 tile[kRnTile];
  __shared__ int wsum[kRnBlock / 64];
  const int i = blockIdx.x * kRnBlock + threadIdx.x;
  float2 p = make_float2(0.f, 0.f);
  if (i < nb) p = load_pt(base, bstride, (size_t)i);
  bool flag = i < nb;
  for (int t0 = 0; t0 < nl; t0 += kRnTile) {
    const int m = min(kRnTile, nl - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += kRnBlock) tile[j] = load_pt(list, lstride, (size_t)(t0 + j));
    __syncthreads();
    if (flag) {                                   // (the reference keeps testing; the outcome is the same)
      bool near = false;
      for (int j = 0; j < m; ++j) near = near || rn_near(p, tile[j], cut);
      flag = !near;
    }
  }
  if (i < nb) keep[i] = flag ? 1 : 0;
  const unsigned long long b = __ballot(flag);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __builtin_popcountll(b);
  __syncthreads();
  if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < kRnBlock / 64; ++w) s += wsum[w]; block_count[blockIdx.x] = s; }
}
// exclusive scan of the block counts (one workgroup), total to *n_out
__global__ void __launch_bounds__(1024)
remove_neighbors_scan_kernel(int *__restrict__ block_count, int nblocks, unsigned long long *__restrict__ n_out) {
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? block_count[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) block_count[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = (unsigned long long)carry;
}
__global__ void __launch_bounds__(kRnBlock)
remove_neighbors_pack_kernel(const float *__restrict__ base, size_t bstride, int nb, const unsigned char *__restrict__ keep,
                             const int *__restrict__ block_off, float2 *__restrict__ out) {
  __shared__ int wsum[kRnBlock / 64];
  const int i = blockIdx.x * kRnBlock + threadIdx.x;
  const bool flag = i < nb && keep[i];
  const unsigned long long b = __ballot(flag);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = __builtin_popcountll(b);
  __syncthreads();
  int off = block_off[blockIdx.x];
  for (int w = 0; w < wv; ++w) off += wsum[w];
  if (flag) out[off + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = load_pt(base, bstride, (size_t)i);
}
