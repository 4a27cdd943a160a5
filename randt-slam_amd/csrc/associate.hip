// Association kernel for gfx950 (compiled with -ffp-contract=off, see cell_math.h).
//
// Replaces the data-association half of Matcher::addNDTFactor (src/ndt_registration/ndt_matcher.cpp:200-215,
// 249-253) together with Map::getClosestCells / Map::getAdjacentIndizes
// (src/ndt_representation/ndt_map.cpp:101-175) and Cell::mahalanobisSquaredIntensity
// (src/ndt_representation/ndt_cell.cpp:172-176).
//
// One workgroup (256 threads) per (scan, submap) pair -- per FOUR pairs, one after the other, when batches share the chip; per
// (pair, chunk) for a lone batch (launch_associate below) --, phases per chunk of 64 moving cells so that every
// global-memory round trip is taken once by all lanes together instead of once per cell, and so that the per-cell
// logic runs one THREAD per cell (64 cells per wave-instruction) instead of one wavefront per cell:
//   P0  a ring-major table of the (2R+1)^2 window offsets is built; optionally (RANDT_ASSOC_STAGE_GRID=1) the
//       submap's dense int32 index grid (40 KB for the 100x100 indoor map) is staged into LDS with 16-byte
//       loads -- by default it is gathered from L2, which is as fast and leaves the LDS to co-running kernels;
//   P1  one thread per moving cell: 48-byte record from HBM/L2, fp32 transform by the initial guess
//       (reference operation order), centre slot; query cells parked in LDS;
//   P1b one wavefront per moving cell, lanes over the 64 innermost window slots (radii 0..3; a window row is one
//       contiguous run of the index grid): every cell's gather is in flight at once, the slot contents land in LDS;
//   P2a one thread per cell: occupied / in-range bit masks of its 64 slots, the reference's termination radius
//       from popcounts of mask prefixes (the window is enumerated ring-major, so "radius <= r" is a prefix), the
//       occupied slots of the final window copied out as the candidate list;
//   P2b cells not settled within radius 3 (sparse neighbourhoods) and maps narrower than the window: one
//       wavefront per cell, ballots over the outer rings (the round-1 path);
//   P3  one thread per (cell, candidate): gather the fixed cell's 48-byte record (L2), fp32
//       Mahalanobis / Euclidean distance in Eigen's operation order;
//   P4  one thread per cell: insertion of sortable 64-bit (distance, compact index) keys into a k-entry sorted
//       list = the order std::sort produces on std::pair<double,size_t>.
#include "cell_math.h"

using namespace randt_dev;

#ifndef ASSOC_BLOCK
#define ASSOC_BLOCK 256
#endif
#define ASSOC_WAVES (ASSOC_BLOCK / 64)
#define ASSOC_MAX_R 7    // window <= 15x15 = 225 slots = 4 lane passes
#define ASSOC_PASSES 4
#ifndef ASSOC_CH
#define ASSOC_CH 64      // moving cells per chunk of the widest instantiation; the kernel is a template on the chunk size CH (16 / 32 / 64):
#endif                   // the LDS of a workgroup is proportional to it (36 KB at 64, 18 KB at 32), and what a co-running batch pays for is LDS x time
#define ASSOC_CAND 64    // candidates per cell: <= (k-1) + 8R = 63 for k <= 8, R <= 7
#define ASSOC_CS 65      // LDS stride of a cell's candidate / slot row (odd: thread-per-cell accesses are conflict-free)
#define ASSOC_QS 11      // LDS stride of a query record (odd => conflict-free)
#define ASSOC_WIDE_MAX_R 15   // the WIDE instantiation: window <= 31 x 31 (max_neighbour_dist / resolution <= 16), k <= 16
#define ASSOC_WIDE_CAND 144   // (k - 1) + 8 R <= 15 + 120
#define ASSOC_WIDE_CS 145
#define ASSOC_WIDE_WT 1024    // window table entries (>= 31^2 = 961)

#ifdef RANDT_TIMING
__device__ long long g_randt_assoc_timing[16];
#define ASSOC_TICK(slot)                                                                  \
  do {                                                                                    \
    __syncthreads();                                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0 && c0 == 0) g_randt_assoc_timing[slot] = wall_clock64(); \
  } while (0)
extern "C" int randt_debug_assoc_timing(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_randt_assoc_timing), sizeof(long long) * 16);
}
#else
#define ASSOC_TICK(slot) do {} while (0)
#endif

namespace {

// duplicate test of Map::getAdjacentIndizes' std::find (ndt_map.cpp:169): slot (i, j) of the window
// of radius r repeats an EARLIER (x-offset-major) window entry iff some t >= 1 has
// i - t*size_x >= -r and j + t <= r.  Never true when size_x > 2r.
__device__ __forceinline__ bool window_dup(int i, int j, int r, int size_x) {
  for (int t = 1; i - t * size_x >= -r; ++t)
    if (j + t <= r) return true;
  return false;
}

// ring-major enumeration of window offsets: w = 0 is the centre, ring r >= 1 occupies
// w in [(2r-1)^2, (2r+1)^2).  Order inside a ring is irrelevant (results are sorted afterwards).
__device__ __forceinline__ void ring_offset(int w, int& i, int& j) {
  if (w == 0) {
    i = j = 0;
    return;
  }
  int r = 1;
  while ((2 * r + 1) * (2 * r + 1) <= w) ++r;
  const int o = w - (2 * r - 1) * (2 * r - 1);
  const int side = 2 * r, edge = o / side, pos = o % side;
  switch (edge) {
    case 0: i = -r + pos; j = -r; break;
    case 1: i = r; j = -r + pos; break;
    case 2: i = r - pos; j = r; break;
    default: i = -r; j = r - pos; break;
  }
}

template <int CTRL>
__device__ __forceinline__ unsigned long long min_u64_dpp(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
  const unsigned long long o = ((unsigned long long)hi << 32) | lo;
  return o < v ? o : v;
}

__device__ __forceinline__ unsigned long long prefix_mask(int n) {
  return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
}

// TP ("throughput placement"): the batch shares the chip with other batches' solves.  Left alone the kernel takes 131 registers
// (three wavefronts per SIMD of its own).  While the solve held 136 x 3 per SIMD it was capped at 96 (22 spilled dwords) to fit
// beside three solves (+1.7 %); since the solve runs at 128 x 4 everything on the chip is placed in 128-register slots and this
// kernel is held to exactly that (no spills): 11.38 -> 11.76 M registrations/s.  A lone batch keeps the registers it likes.
#ifndef RANDT_ASSOC_TP_WPE
#define RANDT_ASSOC_TP_WPE 4
#endif
template <bool STAGE_GRID, int CH, bool TP, bool WIDE = false>
__global__ __launch_bounds__(ASSOC_BLOCK) __attribute__((amdgpu_waves_per_eu(TP ? RANDT_ASSOC_TP_WPE : 1, TP ? RANDT_ASSOC_TP_WPE : 8))) void k_associate(MapView fixed, const int32_t* __restrict__ fixed_idx,
                                                           MapView moving, int moving_first,
                                                           const int32_t* __restrict__ moving_idx,
                                                           const double* __restrict__ guess4, int k, int metric_mahal,
                                                           int transform_full, int32_t* __restrict__ corr, int ch /* cells per chunk <= CH */, int n_pairs_total, int ppw) {
  constexpr int CH_LOG2 = CH > 64 ? 7 : (CH > 32 ? 6 : (CH > 16 ? 5 : 4));
  static_assert(CH % ASSOC_WAVES == 0 && CH >= 16 && CH <= 64, "chunk size (P2a runs one thread per cell on ONE wavefront)");
  // WIDE: configurations beyond every shipped one -- window radius up to ASSOC_WIDE_MAX_R (a 0.25 m map with the same 4 m window:
  // radius 15, 31 x 31 slots) and up to 16 neighbours: longer candidate rows ((k - 1) + 8 R <= 143), a window table of 33^2
  // entries, a 16-entry top-k list, and a ring-by-ring walk of the outer window (P2b).  A correctness path, not a tuned one.
  constexpr int CS = WIDE ? ASSOC_WIDE_CS : ASSOC_CS, CAND = WIDE ? ASSOC_WIDE_CAND : ASSOC_CAND, KMAX = WIDE ? 16 : 8;
  constexpr int WT = WIDE ? ASSOC_WIDE_WT : 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // This kernel is a chain of L2 round trips with a few instructions between them.  Sharing a SIMD with solve wavefronts
  // (16 batches in flight: three fp64-bound wavefronts per SIMD that are older, and the issue arbiter prefers older ones) it
  // used to wait for the issue port at every step of that chain while holding 36 KB of LDS.  Raised issue priority lets it
  // through -- it needs few slots, the solves hardly notice: 9.67 -> 10.54 M registrations/s together with the same line in
  // k_ndt_build (profiles/experiments/r03_issue_priority.md).
  __builtin_amdgcn_s_setprio(RANDT_LATENCY_KERNEL_PRIO);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // TP: a workgroup walks `ppw` pairs one after the other -- in the pipelined region fewer, longer-lived workgroups are cheaper than
  // many short ones (4 per workgroup: +2 %; spreading a pair's chunks over workgroups, the lone-batch placement: -3 .. -7 %).  A
  // compile-time single trip otherwise, so the lone-batch instantiations keep their code.
  const int n_walk = TP ? ppw : 1;
  for (int pp = 0; pp < n_walk; ++pp) {
  const int pair = TP ? (int)blockIdx.x * ppw + pp : (int)blockIdx.x;
  if (TP && pair >= n_pairs_total) break;
  if (TP) __syncthreads();  // the previous pair's LDS is free
  const int fmap = fixed_idx ? fixed_idx[pair] : 0;
  const int mmap = moving_idx ? moving_idx[pair] : moving_first + pair;
  const int n_slots = fixed.n_slots;
  const int32_t* ggrid = fixed.grid + (size_t)fmap * n_slots;
  const randt_cell* fcells = fixed.cells + (size_t)fmap * fixed.cap;
  const randt_cell* mcells = moving.cells + (size_t)mmap * moving.cap;
  int32_t* out = corr + (size_t)pair * moving.cap * k;
  int M = moving.counts[mmap];
  M = M > moving.cap ? moving.cap : M;

  // ---- LDS carve (4-byte types; the 64-bit mask array is rounded up to an 8-byte boundary)
  int32_t* lgrid = reinterpret_cast<int32_t*>(smem);
  const int grid_words = STAGE_GRID ? ((n_slots + 3) & ~3) : 0;
  int32_t* wtab = lgrid + grid_words;                       // [256] packed (i+128) << 8 | (j+128)
  float* qrec = reinterpret_cast<float*>(wtab + WT);       // [CH][QS]: mean3, cov6, centre(bits), pad
  int32_t* clen = reinterpret_cast<int32_t*>(qrec + CH * ASSOC_QS + 1);  // [CH]
  int32_t* cpref = clen + CH;                         // [CH + 1]
  int32_t* cand = cpref + CH + 4;                     // [CH][CS]
  float* cdist = reinterpret_cast<float*>(cand + CH * CS);  // [CH][CS]
  int32_t* p0tab = reinterpret_cast<int32_t*>(cdist);       // [CH][CS] window slots 0..63 of every cell (dead before P3 writes cdist)
  // [CH][2] occupied / in-range bits; CH * CS is odd, so the word offset is rounded up to an even one (8-byte aligned ds_read_b64)
  unsigned long long* pmask = reinterpret_cast<unsigned long long*>(
      smem + ((static_cast<size_t>(reinterpret_cast<char*>(cdist + CH * CS) - smem) + 7) & ~static_cast<size_t>(7)));
  int32_t* ulist = reinterpret_cast<int32_t*>(pmask + 2 * CH);                                // [CH + 1] cells left to P2b, count last

  const int R = fixed.rmax - 1 > 0 ? fixed.rmax - 1 : 0;  // last radius the reference evaluates
  const int side = 2 * R + 1, nwin = side * side;
  const bool need_dup = fixed.size_x <= 2 * R;

  // ---- P0: stage the index grid, build the window table
  if (STAGE_GRID) {
    if (((size_t)ggrid & 15) == 0) {
      const int n4 = n_slots >> 2;
      const int4* g4 = reinterpret_cast<const int4*>(ggrid);
      int4* l4 = reinterpret_cast<int4*>(lgrid);
      for (int i = tid; i < n4; i += ASSOC_BLOCK) l4[i] = g4[i];
      for (int i = (n4 << 2) + tid; i < n_slots; i += ASSOC_BLOCK) lgrid[i] = ggrid[i];
    } else {
      for (int i = tid; i < n_slots; i += ASSOC_BLOCK) lgrid[i] = ggrid[i];
    }
  }
  for (int w = tid; w < WT; w += ASSOC_BLOCK) {
    int i = 0, j = 0;
    if (w < nwin) ring_offset(w, i, j);
    wtab[w] = ((i + 128) << 8) | (j + 128);
  }
  const int32_t* grid = STAGE_GRID ? lgrid : ggrid;

  float aff[4];
  pose_to_affine_f(guess4 + 4 * (size_t)pair, aff);

  // chunks are independent (a moving cell's correspondences depend on nothing but the pair's pose): a small batch spreads
  // the chunks of a pair over gridDim.y workgroups (a lone pair of 75 cells = two chunks used to cost 37 us of latency in
  // front of every fixed-lag window solve)
  for (int c0 = (int)blockIdx.y * ch; c0 < M; c0 += (int)gridDim.y * ch) {
    const int nch = M - c0 < ch ? M - c0 : ch;
    __syncthreads();  // previous chunk fully consumed (also orders P0 before first use)
    ASSOC_TICK(0);

    // ---- P1: transform the query cells of this chunk
    if (tid < nch) {
      randt_cell q = load_cell(mcells + c0 + tid);
      if (transform_full) {
        cell_transform(q, aff);  // Cell::transformCell (ndt_matcher.cpp:207-209)
      } else {
        // initial_guess.cast<float>() * mean.xy (ndt_matcher.cpp:213,250)
        const float x = q.mean[0], y = q.mean[1];
        q.mean[0] = (aff[0] * x - aff[1] * y) + aff[2];
        q.mean[1] = (aff[1] * x + aff[0] * y) + aff[3];
      }
      float* o = qrec + tid * ASSOC_QS;
      o[0] = q.mean[0]; o[1] = q.mean[1]; o[2] = q.mean[2];
#pragma unroll
      for (int e = 0; e < 6; ++e) o[3 + e] = q.cov[e];
      o[9] = __uint_as_float(coord_to_index(fixed, q.mean[0], q.mean[1]));
    }
    __syncthreads();

    // ---- P1b: slots 0..63 of every cell's window (one wavefront per cell; all gathers of the chunk in flight together)
    {
      const int lane_i = (wtab[lane] >> 8) - 128, lane_j = (wtab[lane] & 255) - 128;
      constexpr int PER_WAVE = CH / ASSOC_WAVES;
      int32_t civ[PER_WAVE];
#pragma unroll
      for (int t = 0; t < PER_WAVE; ++t) {  // every request of this wavefront's cells first ...
        const int cc = wave + t * ASSOC_WAVES;
        int32_t ci = -2;  // -2: not a valid window slot, -1: empty slot
        if (cc < nch && lane < nwin) {
          const uint32_t ctr = __float_as_uint(qrec[cc * ASSOC_QS + 9]);
          const uint32_t ni = ctr + (uint32_t)lane_i + (uint32_t)lane_j * (uint32_t)fixed.size_x;
          if (ni < (uint32_t)n_slots) ci = grid[ni];
        }
        civ[t] = ci;
      }
#pragma unroll
      for (int t = 0; t < PER_WAVE; ++t) {  // ... then the slot contents and their occupied / in-range bit masks into LDS
        const int cc = wave + t * ASSOC_WAVES;
        if (cc < nch) {  // wave-uniform
          int32_t ci = civ[t];
          if (ci < -1 && lane < nwin) {
            const uint32_t ctr = __float_as_uint(qrec[cc * ASSOC_QS + 9]);
            const uint32_t ni = ctr + (uint32_t)lane_i + (uint32_t)lane_j * (uint32_t)fixed.size_x;
            ci = ni < (uint32_t)n_slots ? -1 : -2;  // a stored index below -1 counts as an empty slot
          }
          p0tab[cc * CS + lane] = ci;
          const unsigned long long o = __ballot(ci >= 0), v = __ballot(ci >= -1);
          if (lane == 0) {
            pmask[2 * cc] = o;
            pmask[2 * cc + 1] = v;
          }
        }
      }
    }
    __syncthreads();
    ASSOC_TICK(1);
    // ---- P2a: one thread per cell.  Reference loop (ndt_map.cpp:101-152): evaluate radius 0, 1, ... until enough
    // targets (nt >= k) or enough adjacent slots (nadj >= n_slots) were seen or the radius reaches rmax.  Ring-major
    // enumeration makes "everything up to radius r" the first (2r+1)^2 slots, so nt / nadj are popcounts of mask prefixes.
    if (wave == 0) {
      const int c = lane;
      int len = -1;  // -1: not settled here (P2b)
      if (c < nch && !need_dup) {
        const int32_t* row = p0tab + c * CS;
        const unsigned long long occ = pmask[2 * c], val = pmask[2 * c + 1];
        const int r_hi = R < 3 ? R : 3;  // radii completely inside the 64 slots
        int rstar = -1;
        for (int r = 0; r <= r_hi && rstar < 0; ++r) {
          const unsigned long long pm = prefix_mask((2 * r + 1) * (2 * r + 1));
          const int nt = __popcll(occ & pm), nadj = __popcll(val & pm);
          if (!(nt < k && nadj < n_slots)) rstar = r;
        }
        if (rstar < 0 && r_hi == R) rstar = R;  // "if (r >= rmax) break" after the last radius
        if (rstar >= 0) {
          unsigned long long m = occ & prefix_mask((2 * rstar + 1) * (2 * rstar + 1));
          len = 0;
          while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            cand[c * CS + len] = row[b];
            ++len;
          }
        }
      }
      if (c < nch) clen[c] = len;
      // the cells left open, compacted so that P2b spreads them evenly over the wavefronts
      const bool open = c < nch && len < 0;
      const unsigned long long om = __ballot(open);
      if (open) ulist[__popcll(om & prefix_mask(lane))] = c;
      if (lane == 0) ulist[CH] = __popcll(om);
    }
    __syncthreads();
    // ---- P2b: the cells P2a left open, one wavefront per cell (outer rings / wrapping windows)
    const int n_open = ulist[CH];
    for (int u = wave; u < n_open; u += ASSOC_WAVES) {
      const int c = ulist[u];
      const uint32_t center = __float_as_uint(qrec[c * ASSOC_QS + 9]);
      const int32_t ci_pass0 = p0tab[c * CS + lane];
      if (WIDE && !need_dup) {
        // radii 0 .. 3 (the first 49 slots) are known not to end the search (P2a); from radius 4 on ring by ring, as the
        // reference's loop does: collect the ring, then test "enough targets or every slot seen", then "r >= rmax"
        const unsigned long long pm49 = prefix_mask(49);
        const unsigned long long o0 = __ballot(ci_pass0 >= 0) & pm49, v0 = __ballot(ci_pass0 >= -1) & pm49;
        int base = 0, nt = __popcll(o0), nadj = __popcll(v0);
        if ((o0 >> lane) & 1ull) cand[c * CS + __popcll(o0 & prefix_mask(lane))] = ci_pass0;
        base = nt;
        for (int r = 4; r <= R; ++r) {
          const int w_end = (2 * r + 1) * (2 * r + 1);
          for (int w0 = (2 * r - 1) * (2 * r - 1); w0 < w_end; w0 += 64) {
            const int w = w0 + lane;
            int32_t ci = -2;
            if (w < w_end) {
              const int packed = wtab[w];
              const uint32_t ni = center + (uint32_t)((packed >> 8) - 128) + (uint32_t)((packed & 255) - 128) * (uint32_t)fixed.size_x;
              if (ni < (uint32_t)n_slots) {
                ci = grid[ni];
                if (ci < -1) ci = -1;
              }
            }
            const unsigned long long o = __ballot(ci >= 0), v = __ballot(ci >= -1);
            const int pos = base + __popcll(o & prefix_mask(lane));
            if (ci >= 0 && pos < CAND) cand[c * CS + pos] = ci;
            base += __popcll(o);
            nt += __popcll(o);
            nadj += __popcll(v);
          }
          if (!(nt < k && nadj < n_slots)) break;
        }
        if (lane == 0) clen[c] = base < CAND ? base : CAND;
        continue;
      }
      // the outer rings' slots, all requested before the first one is looked at (one L2 round trip per cell, not per pass)
      int32_t ci_outer[ASSOC_PASSES];
#pragma unroll
      for (int p = 1; p < ASSOC_PASSES; ++p) {
        const int r_first = p == 1 ? 4 : (p == 2 ? 6 : 7);
        const int w = p * 64 + lane;
        int32_t ci = -2;
        if (!need_dup && r_first <= R && w < nwin) {
          const int packed = wtab[w];
          const uint32_t ni = center + (uint32_t)((packed >> 8) - 128) + (uint32_t)((packed & 255) - 128) * (uint32_t)fixed.size_x;
          if (ni < (uint32_t)n_slots) {
            ci = grid[ni];
            if (ci < -1) ci = -1;
          }
        }
        ci_outer[p] = ci;
      }
      int32_t cidx[ASSOC_PASSES];
      unsigned long long occ[ASSOC_PASSES], val[ASSOC_PASSES];
      int rstar = -1;
      if (!need_dup) {
        // radius of lane r: prefix length (2r+1)^2; lanes >= rmax never win
        const int need_l = (2 * lane + 1) * (2 * lane + 1);
        int nt_l = 0, nadj_l = 0;
        int loaded = 0;
#pragma unroll
        for (int p = 0; p < ASSOC_PASSES; ++p) {
          // passes are fetched lazily: pass 0 covers radii 0..3, pass 1 radii 4..5, pass 2 radius 6, pass 3 radius 7
          const int r_first = p == 0 ? 0 : (p == 1 ? 4 : (p == 2 ? 6 : 7));
          if (rstar < 0 && r_first <= R) {
            const int w = p * 64 + lane;
            const int32_t ci = p == 0 ? ci_pass0 : ci_outer[p];
            (void)w;
            cidx[p] = ci;
            occ[p] = __ballot(ci >= 0);
            val[p] = __ballot(ci >= -1);
            loaded = p + 1;
            const unsigned long long pm = prefix_mask(need_l - 64 * p);
            nt_l += __popcll(occ[p] & pm);
            nadj_l += __popcll(val[p] & pm);
            // radii completely inside the passes fetched so far
            const int r_last = p == 0 ? 3 : (p == 1 ? 5 : (p == 2 ? 6 : 7));
            const int r_hi = r_last < R ? r_last : R;
            const bool stop = lane <= r_hi && !(nt_l < k && nadj_l < n_slots);
            const unsigned long long sm = __ballot(stop);
            if (sm) rstar = __ffsll((long long)sm) - 1;
            else if (r_hi == R) rstar = R;  // "if (r >= rmax) break" after the last radius
          }
        }
        // candidates of the final window -> cand[c][0 .. nt)
        int base = 0;
        const int need = (2 * rstar + 1) * (2 * rstar + 1);
#pragma unroll
        for (int p = 0; p < ASSOC_PASSES; ++p) {
          if (p < loaded) {
            const bool in = cidx[p] >= 0 && (p * 64 + lane) < need;
            const unsigned long long m = __ballot(in);
            const int pos = base + __popcll(m & prefix_mask(lane));
            if (in && pos < CAND) cand[c * CS + pos] = cidx[p];
            base += __popcll(m);
          }
        }
        if (lane == 0) clen[c] = base < CAND ? base : CAND;
        continue;
      }
      // tiny maps (size_x <= 2R): the window wraps onto itself and the reference drops repeated entries
      // (std::find) -- radius by radius, as written there
      int wi[ASSOC_PASSES], wj[ASSOC_PASSES];
      int loaded = 0;  // passes evaluated so far
      int nt = 0, nadj = 0, radius = 0;
      // while (targets.size() < n && adjacent.size() < n_cells_) {...; r++; if (r >= rmax) break;}
      while (nt < k && nadj < n_slots) {
        const int need = (2 * radius + 1) * (2 * radius + 1);  // window entries of this radius
#pragma unroll
        for (int p = 0; p < ASSOC_PASSES; ++p) {
          if (p == loaded && need > 64 * p) {
            const int w = p * 64 + lane;
            int32_t ci = -2;
            int i = 0, j = 0;
            if (w < nwin) {
              const int packed = wtab[w];
              i = (packed >> 8) - 128;
              j = (packed & 255) - 128;
              const uint32_t ni = center + (uint32_t)i + (uint32_t)j * (uint32_t)fixed.size_x;
              if (ni < (uint32_t)n_slots) {
                ci = grid[ni];
                if (ci < -1) ci = -1;
              }
            }
            cidx[p] = ci;
            wi[p] = i;
            wj[p] = j;
            loaded = p + 1;
          }
        }
        nt = 0;
        nadj = 0;
#pragma unroll
        for (int p = 0; p < ASSOC_PASSES; ++p) {
          if (p < loaded) {
            const bool rep = window_dup(wi[p], wj[p], radius, fixed.size_x);
            const unsigned long long o = __ballot(cidx[p] >= 0 && !rep);
            const unsigned long long v = __ballot(cidx[p] >= -1 && !rep);
            const unsigned long long pm = prefix_mask(need - 64 * p);
            nt += __popcll(o & pm);
            nadj += __popcll(v & pm);
          }
        }
        rstar = radius;
        ++radius;
        if (radius >= fixed.rmax) break;
      }
      int base = 0;
      if (rstar >= 0) {
        const int need = (2 * rstar + 1) * (2 * rstar + 1);
#pragma unroll
        for (int p = 0; p < ASSOC_PASSES; ++p) {
          if (p < loaded) {
            const bool in = cidx[p] >= 0 && (p * 64 + lane) < need && !window_dup(wi[p], wj[p], rstar, fixed.size_x);
            const unsigned long long m = __ballot(in);
            const int pos = base + __popcll(m & prefix_mask(lane));
            if (in && pos < CAND) cand[c * CS + pos] = cidx[p];
            base += __popcll(m);
          }
        }
      }
      if (lane == 0) clen[c] = base < CAND ? base : CAND;
    }
    __syncthreads();

    ASSOC_TICK(2);
    // exclusive prefix of clen over the chunk (one wavefront)
    if (wave == 0) {
      int carry = 0;
#pragma unroll
      for (int h = 0; h < (CH + 63) / 64; ++h) {
        const int e = h * 64 + lane;
        const int v = e < nch ? clen[e] : 0;
        const int incl = wave_inclusive_scan(v);
        if (e < CH) cpref[e] = carry + incl - v;
        carry += __builtin_amdgcn_readlane(incl, 63);
      }
      if (lane == 0) cpref[CH] = carry;
    }
    __syncthreads();
    const int total = cpref[CH];

    ASSOC_TICK(3);
    // ---- P3: one thread per (cell, candidate): gather + fp32 distance
    for (int p = tid; p < total; p += ASSOC_BLOCK) {
      int lo = 0, hi = nch;  // largest c with cpref[c] <= p
#pragma unroll
      for (int s = 0; s < CH_LOG2 + 1; ++s) {
        const int mid = (lo + hi) >> 1;
        if (hi - lo > 1) {
          if (cpref[mid] <= p) lo = mid; else hi = mid;
        }
      }
      const int c = lo, jj = p - cpref[lo];
      const int32_t fi_raw = cand[c * CS + jj];
      const int32_t fi = fi_raw < fixed.cap ? fi_raw : fixed.cap - 1;
      const randt_cell f = load_cell(fcells + fi);
      const float* o = qrec + c * ASSOC_QS;
      float d;
      if (metric_mahal) {
        randt_cell q;
        q.mean[0] = o[0]; q.mean[1] = o[1]; q.mean[2] = o[2];
#pragma unroll
        for (int e = 0; e < 6; ++e) q.cov[e] = o[3 + e];
        d = mahalanobis3f(q, f);
      } else {
        const float dx = o[0] - f.mean[0], dy = o[1] - f.mean[1];
        d = sqrtf(dx * dx + dy * dy);
      }
      cdist[c * CS + jj] = d;
    }
    __syncthreads();

    ASSOC_TICK(4);
    // ---- P4: top-k per cell by (dist, idx), one thread per cell.  A candidate is ONE sortable 64-bit key (order-preserving
    // image of the float distance, then the compact index); the k smallest keys are kept in a sorted register list by
    // compare-and-swap insertion = the first k entries of std::sort on std::pair<double,size_t>.
    if (tid < nch) {
      const int c = tid;
      constexpr unsigned long long EMPTY = ~0ull;
      const int n = clen[c];
      unsigned long long best[KMAX];
#pragma unroll
      for (int s2 = 0; s2 < KMAX; ++s2) best[s2] = EMPTY;
      for (int jj = 0; jj < n; ++jj) {
        const uint32_t u = __float_as_uint(cdist[c * CS + jj] + 0.0f);  // -0 -> +0
        const uint32_t ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // numeric order as unsigned order
        unsigned long long key = ((unsigned long long)ord << 32) | (uint32_t)cand[c * CS + jj];
#pragma unroll
        for (int s2 = 0; s2 < KMAX; ++s2) {
          if (s2 < k) {  // uniform
            if (key == best[s2]) key = EMPTY;  // a repeated (distance, index) pair counts once
            const unsigned long long lo = key < best[s2] ? key : best[s2];
            key = key < best[s2] ? best[s2] : key;
            best[s2] = lo;
          }
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < KMAX; ++s2)
        if (s2 < k) out[(size_t)(c0 + c) * k + s2] = best[s2] == EMPTY ? -1 : (int32_t)(uint32_t)best[s2];
    }
    ASSOC_TICK(5);
  }
  }
}

size_t assoc_lds_bytes(int n_slots, bool stage, int CH = ASSOC_CH, bool wide = false) {
  const int cs = wide ? ASSOC_WIDE_CS : ASSOC_CS, wt = wide ? ASSOC_WIDE_WT : 256;
  size_t words = (stage ? ((n_slots + 3) & ~3) : 0) + wt + (CH * ASSOC_QS + 1) + CH + (CH + 4) +
                 2 * CH * cs + 2 /* pmask alignment */ + 4 * CH + CH + 4;
  return words * 4 + 64;
}

template <bool STAGE, int CH, bool TP = false, bool WIDE = false>
int launch_associate_cfg(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving, int moving_first,
                         int n_pairs, const double* d_guess4, int k, int full, int32_t* d_corr, const int32_t* d_moving_idx, bool spread,
                         bool shared = false) {
  const size_t lds = assoc_lds_bytes(fixed.n_slots, STAGE, CH, WIDE);
  int split = 1;
  if (spread) {
    split = (moving.cap + CH - 1) / CH;
    if (split > 64) split = 64;
    if (split < 1) split = 1;
  }
  RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_associate<STAGE, CH, TP, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // pairs per workgroup: the walk is for callers that keep several batches in flight (or a batch that fills the chip several
  // times over); a lone batch of a few hundred pairs more than the split geometry takes stays one pair per workgroup
  const bool walk = TP && (n_pairs >= 8 * ctx->n_cus || shared);
  const int ppw = walk ? (ctx->assoc_tp_ppw > 0 ? ctx->assoc_tp_ppw : 1) : 1;
  hipLaunchKernelGGL((k_associate<STAGE, CH, TP, WIDE>), dim3((n_pairs + ppw - 1) / ppw, split), dim3(ASSOC_BLOCK), lds, ctx->stream, fixed, d_fixed_idx, moving,
                     moving_first, d_moving_idx, d_guess4, k, full, full, d_corr, CH, n_pairs, ppw);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

}  // namespace

#ifndef RANDT_ASSOC_SMALL_CH
#define RANDT_ASSOC_SMALL_CH 16
#endif

int launch_associate(randt_ctx* ctx, const MapView& fixed, const int32_t* d_fixed_idx, const MapView& moving,
                     int moving_first, int n_pairs, const double* d_guess4, int k, int lookup_mahalanobis,
                     int use_intensity, int32_t* d_corr, const int32_t* d_moving_idx) {
  if (n_pairs <= 0) return RANDT_OK;
  if (!fixed.grid) return randt_set_error(ctx, RANDT_ERR_INVALID, "fixed maps need an index grid", hipSuccess);
  if (fixed.rmax - 1 > ASSOC_WIDE_MAX_R)
    return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "max_neighbour_dist/resolution > 16 not supported by the association kernel", hipSuccess);
  if (k > 16) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "n_neighbours > 16 not supported by the association kernel", hipSuccess);
  const int full = (use_intensity && lookup_mahalanobis) ? 1 : 0;
  if (fixed.rmax - 1 > ASSOC_MAX_R || k > 8) {
    // beyond every shipped configuration (ndt_map.cpp:117 and ndt_matcher.cpp:210 take any value): the WIDE instantiation,
    // one workgroup per (pair, 32-cell chunk); maps narrower than such a window (the reference's std::find de-duplication) stay refused
    if (fixed.size_x <= 2 * (fixed.rmax - 1) && fixed.rmax - 1 > ASSOC_MAX_R)
      return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "maps narrower than a window of radius > 7 are not supported by the association kernel", hipSuccess);
    return launch_associate_cfg<false, 32, false, true>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_guess4, k, full, d_corr, d_moving_idx, true);
  }
  const bool stage = ctx->assoc_stage_grid && assoc_lds_bytes(fixed.n_slots, true) <= (size_t)ctx->lds_limit / 2;
  // Chunk size and placement.  A handful of pairs (the terms of a fixed-lag window, one loop-closure candidate) and a lone
  // batch of up to two pairs per CU (the size at which the solve takes its split geometry) are all latency: one workgroup per
  // (pair, chunk), 16-cell chunks for the handful (the wavefront-per-cell phases of a chunk are serial rounds).  Batches
  // that share the chip with other batches (RANDT_SOLVE_THROUGHPUT, or more than two pairs per CU) are charged for LDS x
  // time: one workgroup per pair walking RANDT_ASSOC_TP_CH-cell chunks.
  // (does this batch have the device to itself?  RANDT_SOLVE_AUTO looks at the process's other contexts: randt_internal.h)
  const bool shared = n_pairs > 64 && randt_throughput_placement(ctx);
  randt_note_enqueue(ctx);
  const bool lone = !shared && n_pairs <= 2 * ctx->n_cus;
  const bool spread = n_pairs <= 64 || lone;
  // lone batches: as many (pair, chunk) workgroups as stay resident in one round -- 16-cell chunks up to 128 pairs (64 pairs:
  // 17.9 -> 15.0 us), 32-cell ones above (512 pairs: 24.8 -> 23.5 us; 16-cell chunks would need a second round there: 36.5)
  const int chunk = spread ? (n_pairs <= 128 ? RANDT_ASSOC_SMALL_CH : 32) : ctx->assoc_tp_ch;
#define RANDT_ASSOC_GO(ST, CC) \
  return launch_associate_cfg<ST, CC>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_guess4, k, full, d_corr, d_moving_idx, spread)
  if (stage) {
    if (chunk <= 16) RANDT_ASSOC_GO(true, 16);
    if (chunk <= 32) RANDT_ASSOC_GO(true, 32);
    RANDT_ASSOC_GO(true, 64);
  }
  if (chunk <= 16) RANDT_ASSOC_GO(false, 16);
  if (chunk <= 32) RANDT_ASSOC_GO(false, 32);
  if (chunk <= 48) RANDT_ASSOC_GO(false, 48);
  if (!spread)
    return launch_associate_cfg<false, 64, true>(ctx, fixed, d_fixed_idx, moving, moving_first, n_pairs, d_guess4, k, full, d_corr, d_moving_idx, spread, shared);
  RANDT_ASSOC_GO(false, 64);
#undef RANDT_ASSOC_GO
}
