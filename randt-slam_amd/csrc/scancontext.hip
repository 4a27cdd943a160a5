// Scan Context loop-closure candidate search for gfx950 -- SURVEY row f-4 (compiled with -ffp-contract=off).
//
// Replaces SCManager (src/local_fuser/Scancontext/Scancontext.cpp of the reference; parameters
// src/ndt_slam/ndt_slam.cpp:515-552), the stage that hands loop-closure candidates to
// Matcher::estimateLoopConstraint (local_fuser.cpp:323-335):
//   k_sc_make     makeScancontext + makeRingkey / makeSectorkey (:156-237), one workgroup per keyframe scan:
//                 bin index and value of every point into LDS, a stable counting sort of the points by bin, then ONE
//                 THREAD PER BIN folds its points in input order, so every bin sum runs in the reference's sequential
//                 order (bit-reproducible, including the quirk that a touched bin starts at NO_POINT = -1000);
//   k_sc_knn      detectLoopClosureID's candidate search (:261-300), one workgroup per query node: float ring-key
//                 distances to the searchable part of the database, k rounds of (distance, index) arg-min = the k nearest keys;
//   k_sc_detect   the rest of detectLoopClosureID (:300-341), one workgroup per (query, candidate rank): sector-key
//                 alignment, the column-shift search of the cosine distance and the odometry term
//                 (distanceBtnScanContext, :115-152); k_sc_pick takes the first smallest distance.
// Every reduction is a per-thread left-to-right loop in the oracle's order.
#include "randt_internal.h"

#include <math.h>

#pragma clang fp contract(off)

#define SC_BLOCK 256
#define SC_MAX_SECTOR 128
#define SC_MAX_RING 64
#define SC_MAX_CAND 32
#define SC_TERM_CAP 4096  // (shift, column) cosine terms of one candidate kept in LDS (32 KB): (2 radius + 1) x num_sector <= this

namespace {

struct ScParams {
  int R, S, exclude_recent, n_cand, radius;
  double max_radius, dist_thresh, assumed_drift, odom_eps, odom_weight, intensity_factor;
};

// xy2theta (Scancontext.cpp:24-37); the arctangent is evaluated in double and rounded to float (= the correctly
// rounded float arctangent, identical on host and device)
__device__ __forceinline__ float xy2theta(float x, float y) {
  const double k = 180 / M_PI;
  if (x >= 0 && y >= 0) return (float)(k * (float)atan((double)(y / x)));
  if (x < 0 && y >= 0) return (float)(180 - (k * (float)atan((double)(y / (-x)))));
  if (x < 0 && y < 0) return (float)(180 + (k * (float)atan((double)(y / x))));
  if (x >= 0 && y < 0) return (float)(360 - (k * (float)atan((double)((-y) / x))));
  return 0.0f;
}

__global__ __launch_bounds__(SC_BLOCK) void k_sc_make(const float* __restrict__ pts, int pitch, const int32_t* __restrict__ n_pts_arr,
                                                      int stride, int ioff, ScParams P, double* __restrict__ desc,
                                                      double* __restrict__ ring_key, double* __restrict__ sector_key) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* pz = reinterpret_cast<float*>(smem);                     // [pitch]
  uint16_t* pbin = reinterpret_cast<uint16_t*>(pz + pitch);      // [pitch]
  const int scan = blockIdx.x, tid = threadIdx.x;
  int n = n_pts_arr ? n_pts_arr[scan] : pitch;
  n = n < 0 ? 0 : (n > pitch ? pitch : n);
  const float* sp = pts + (size_t)scan * pitch * stride;
  const int R = P.R, S = P.S, nb = R * S;
  for (int i = tid; i < n; i += SC_BLOCK) {
    const float x = sp[(size_t)i * stride], y = sp[(size_t)i * stride + 1];
    const float z = (float)(sp[(size_t)i * stride + ioff] * P.intensity_factor);
    const float azim_range = sqrtf(x * x + y * y);
    const float azim_angle = xy2theta(x, y);
    int bin = 0xffff;
    if (!(azim_range > P.max_radius)) {
      int ring = (int)ceil((azim_range / P.max_radius) * R);
      ring = ring < R ? ring : R;
      ring = ring > 1 ? ring : 1;
      int sect = (int)ceil((azim_angle / 360.0) * S);
      sect = sect < S ? sect : S;
      sect = sect > 1 ? sect : 1;
      bin = (sect - 1) * R + (ring - 1);
    }
    pz[i] = z;
    pbin[i] = (uint16_t)bin;
  }
  // Every bin sum must run in the points' input order (the reference's loop, bit for bit).  Round 3 gave a thread four bins and
  // let it walk all n points (a 2000-step chain: 100 us for one scan); adding in input order 64 points at a time through LDS is
  // no better, because consecutive points of a radar sweep share a bin (91 us).  So the points are SORTED by bin first -- a
  // stable counting sort on integers -- and a thread then folds only its own bins' points, in order:
  //   ranks   a wavefront walks the points 64 at a time; the lanes of a step that share a bin find each other with one ballot
  //           per distinct bin, the group's first lane takes (and advances) the bin's counter in LDS, every lane's rank is that
  //           base + its position in the group.  Steps run in program order, so ranks follow the input order; the four
  //           wavefronts take the bins with (bin mod 4) == their number: disjoint counters, four independent chains;
  //   scan    exclusive prefix of the counters -> first[bin];
  //   place   zs[first[bin] + rank] = the point's value;
  //   fold    one thread per bin: NO_POINT + z + z + ... over its slice of `zs`.
  uint16_t* prank = pbin + pitch;                      // [pitch]
  float* zs = reinterpret_cast<float*>(smem + (((size_t)pitch * 8 + 15) & ~(size_t)15));  // [pitch] the values in bin order
  int* cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(zs) + (((size_t)pitch * 4 + 15) & ~(size_t)15));  // [nb]
  int* first = cnt + nb;                               // [nb]
  double* dl = reinterpret_cast<double*>(first + nb);  // [nb] the finished descriptor
  __shared__ int wsum[SC_BLOCK / 64];
  for (int b = tid; b < nb; b += SC_BLOCK) cnt[b] = 0;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  {
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const int b = i < n ? pbin[i] : 0xffff;
      const bool mine = b != 0xffff && (b & 3) == wave;
      // the groups of this step, one ballot per DISTINCT bin (consecutive points of a sweep share bins: a handful per step)
      unsigned long long todo = __ballot(mine), grp = 0ull;
      while (todo) {  // wave-uniform
        const int head = __ffsll((long long)todo) - 1;
        const int hb = __shfl(b, head, 64);
        const unsigned long long g = __ballot(mine && b == hb);
        if (mine && b == hb) grp = g;
        todo &= ~g;
      }
      const int leader = mine ? __ffsll((long long)grp) - 1 : lane;
      int base = 0;
      if (mine && lane == leader) base = atomicAdd(&cnt[b], __popcll(grp));  // all groups' counters in one LDS instruction
      base = __shfl(base, leader, 64);
      if (mine) prank[i] = (uint16_t)(base + __popcll(grp & lt));
    }
  }
  __syncthreads();
  {  // exclusive prefix over the bins: a contiguous chunk per thread, wave scan of the chunk sums, four wave totals
    const int chunk = (nb + SC_BLOCK - 1) / SC_BLOCK;
    const int b0 = tid * chunk, b1 = (b0 + chunk) < nb ? (b0 + chunk) : nb;
    int local = 0;
    for (int b = b0; b < b1; ++b) local += cnt[b];
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w2 = 0; w2 < wave; ++w2) run += wsum[w2];
    for (int b = b0; b < b1; ++b) {
      first[b] = run;
      run += cnt[b];
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += SC_BLOCK) {
    const int b = pbin[i];
    if (b != 0xffff) zs[first[b] + prank[i]] = pz[i];
  }
  __syncthreads();
  double* d = desc + (size_t)scan * nb;
  for (int b = tid; b < nb; b += SC_BLOCK) {
    double a = -1000;  // NO_POINT; values are ADDED to it (:187)
    const int k0 = first[b], k1 = k0 + cnt[b];
#pragma unroll 4
    for (int k = k0; k < k1; ++k) a += (double)zs[k];  // (the reads do not depend on the sum: only the additions are a chain)
    a = a == -1000 ? 0.0 : a;
    d[b] = a;
    dl[b] = a;  // the keys below read the descriptor from LDS (from global memory each of their 20 .. 45 steps was an L2 round trip)
  }
  __syncthreads();
  if (tid < R) {  // rowwise mean
    double a = 0;
    for (int s = 0; s < S; ++s) a += dl[(size_t)s * R + tid];
    ring_key[(size_t)scan * R + tid] = a / S;
  }
  if (tid >= 64 && tid - 64 < S) {  // columnwise mean (second wavefront onwards)
    const int s = tid - 64;
    double a = 0;
    for (int r = 0; r < R; ++r) a += dl[(size_t)s * R + r];
    sector_key[(size_t)scan * S + s] = a / R;
  }
}

// lexicographic (value, index) block arg-min over 256 threads; result broadcast.  scratch: 2 x 4 words.
__device__ __forceinline__ void block_argmin(double& v, int& idx, double* sv, int* si) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_xor(v, off, 64);
    const int oi = __shfl_xor(idx, off, 64);
    if (ov < v || (ov == v && oi < idx)) {
      v = ov;
      idx = oi;
    }
  }
  __syncthreads();
  if (lane == 0) {
    sv[wave] = v;
    si[wave] = idx;
  }
  __syncthreads();
  v = sv[0];
  idx = si[0];
#pragma unroll
  for (int w = 1; w < SC_BLOCK / 64; ++w)
    if (sv[w] < v || (sv[w] == v && si[w] < idx)) {
      v = sv[w];
      idx = si[w];
    }
}

// what a (query, candidate) workgroup leaves for k_sc_pick
struct ScCandidate {
  double cd;           // distanceBtnScanContext + the odometry term
  int32_t shift, idx;  // argmin shift, database index (-1: this query has fewer candidates)
};

// The candidate search of a query, once (every (query, rank) workgroup used to repeat it on a distance row of its
// own -- n_queries x n_candidates x n_db floats of workspace): the squared float distances of the ring keys (nanoflann L2,
// accumulated in float) and the k nearest keys in ascending (distance, index) order.  One workgroup per query; cand_out
// [q][c] = database index of rank c, -1 beyond the query's candidates (or for the early return, :274-278).
__global__ __launch_bounds__(SC_BLOCK) void k_sc_knn(ScParams P, const double* __restrict__ ring_keys, int n_db, const int32_t* __restrict__ query_ids,
                                                     float* __restrict__ d2ws, int d2pitch, int32_t* __restrict__ cand_out) {
  __shared__ double sv[4];
  __shared__ int si[4];
  const int q = blockIdx.x, tid = threadIdx.x;
  const int node = query_ids ? query_ids[q] : q;
  const int R = P.R;
  float* d2 = d2ws + (size_t)q * d2pitch;
  int32_t* cand = cand_out + (size_t)q * P.n_cand;
  if (node < P.exclude_recent + 1 || node >= n_db) {
    for (int c = tid; c < P.n_cand; c += SC_BLOCK) cand[c] = -1;
    return;
  }
  const int n_search = node + 1 - P.exclude_recent;
  for (int i = tid; i < n_search; i += SC_BLOCK) {
    float acc = 0.0f;
    for (int r = 0; r < R; ++r) {
      const float a = (float)ring_keys[(size_t)node * R + r], b = (float)ring_keys[(size_t)i * R + r];
      const float d = a - b;
      acc += d * d;
    }
    d2[i] = acc;
  }
  __syncthreads();
  const int kk = P.n_cand < n_search ? P.n_cand : n_search;
  for (int c = 0; c < P.n_cand; ++c) {
    if (c >= kk) {  // uniform
      if (tid == 0) cand[c] = -1;
      continue;
    }
    double bv = 1e300;
    int bi = 0x7fffffff;
    for (int i = tid; i < n_search; i += SC_BLOCK) {
      const float v = d2[i];
      if (v >= 0.0f && ((double)v < bv || ((double)v == bv && i < bi))) {
        bv = (double)v;
        bi = i;
      }
    }
    block_argmin(bv, bi, sv, si);
    if (tid == 0) {
      cand[c] = bi;
      d2[bi] = -1.0f;
    }
    __syncthreads();
  }
}

// One workgroup per (query, candidate rank): the candidates of a query are independent of each other until the final
// "smallest distance, first one wins" (k_sc_pick), and each is a chain of short phases (~8 us) -- ten of them in a row were
// most of a query's 106 us.  The candidates themselves come from k_sc_knn.
__global__ __launch_bounds__(SC_BLOCK) void k_sc_detect(ScParams P, const double* __restrict__ desc, const double* __restrict__ ring_keys,
                                                        const double* __restrict__ pos, const double* __restrict__ dist, int n_db,
                                                        const int32_t* __restrict__ query_ids, const int32_t* __restrict__ cand_in,
                                                        ScCandidate* __restrict__ records, int staged) {
  // staged: the query's and the current candidate's descriptor (S x R doubles each) are copied into LDS with coalesced loads;
  // every dot product below then reads LDS instead of walking two global columns element by element (a candidate cost 12 us of
  // L2 round trips that way).  Descriptors too large for it stay in global memory.
  extern __shared__ __attribute__((aligned(16))) char sc_dyn[];
  __shared__ double k1[SC_MAX_SECTOR], k2[SC_MAX_SECTOR];
  __shared__ double n1[SC_MAX_SECTOR], n2[SC_MAX_SECTOR];  // column norms of the query / the candidate (shift-independent)
  __shared__ double term[SC_TERM_CAP];
  __shared__ double sv[4];
  __shared__ int si[4];
  const int q = blockIdx.x, c_only = blockIdx.y, tid = threadIdx.x;
  const int node = query_ids ? query_ids[q] : q;
  const int R = P.R, S = P.S;
  ScCandidate* rec = records + (size_t)q * gridDim.y + c_only;
  const int ci = cand_in[(size_t)q * gridDim.y + c_only];  // uniform
  if (ci < 0) {  // the early return (:274-278) or a rank beyond the query's candidates: k_sc_pick says so
    if (tid == 0) rec->idx = -1;
    return;
  }
  // ---- pairwise distances (distanceBtnScanContext) in candidate order
  const double* sc1 = desc + (size_t)node * R * S;
  double* s1 = reinterpret_cast<double*>(sc_dyn);
  double* s2 = s1 + (size_t)R * S;
  if (staged) {
    for (int i = tid; i < R * S; i += SC_BLOCK) s1[i] = sc1[i];
    __syncthreads();
    sc1 = s1;
  }
  // The column-shift search evaluates, per shift, sum over the columns of dot / (|a| |b|): the norms do not depend on the shift
  // and the (shift, column) dot products not on each other, so they are formed once / by all 256 threads; only each shift's sum
  // over the columns stays one thread's left-to-right loop (the reference's order) -- same numbers, ~30x less serial work.
  const int n_shift = 2 * P.radius + 1;
  const bool spread_shifts = n_shift <= S && n_shift * S <= SC_TERM_CAP;
  if (tid < S) {
    double a = 0, na = 0;
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
      const double v = sc1[(size_t)tid * R + r];
      a += v;
      na += v * v;
    }
    k1[tid] = a / R;
    n1[tid] = sqrt(na);
  }
  {
    const double* sc2 = desc + (size_t)ci * R * S;
    __syncthreads();
    if (staged) {
      for (int i = tid; i < R * S; i += SC_BLOCK) s2[i] = sc2[i];
      __syncthreads();
      sc2 = s2;
    }
    if (tid < S) {
      double b = 0, nb = 0;
#pragma unroll 4
      for (int r = 0; r < R; ++r) {
        const double v = sc2[(size_t)tid * R + r];
        b += v;
        nb += v * v;
      }
      k2[tid] = b / R;
      n2[tid] = sqrt(nb);
    }
    __syncthreads();
    // fastAlignUsingVkey: thread = shift, first minimal shift wins
    double nv = 1e300;
    int ns = 0x7fffffff;
    if (tid < S) {
      double nn = 0;
#pragma unroll 4
      for (int s = 0; s < S; ++s) {
        int j = s - tid;
        j = j < 0 ? j + S : j;
        const double d = k1[s] - k2[j];
        nn += d * d;
      }
      nv = sqrt(nn);
      ns = tid;
      if (!(nv < 10000000)) {  // "cur_diff_norm < min" never true: shift 0 stays
        nv = 1e300;
        ns = 0x7fffffff;
      }
    }
    block_argmin(nv, ns, sv, si);
    const int argmin_vkey = ns == 0x7fffffff ? 0 : ns;
    // column-shift search: the 2 radius + 1 shifts around it, ascending shift value = thread order
    double dv = 1e300;
    int ds = 0x7fffffff;
    if (spread_shifts) {
      // every (shift slot, column) term by all threads; slot i holds shift argmin - radius + i (mod S)
      for (int p = tid; p < n_shift * S; p += SC_BLOCK) {
        const int slot = p / S, col = p - slot * S;
        int sh = argmin_vkey - P.radius + slot;
        sh = sh < 0 ? sh + S : (sh >= S ? sh - S : sh);
        int j = col - sh;
        j = j < 0 ? j + S : j;
        const double* a = sc1 + (size_t)col * R;
        const double* b = sc2 + (size_t)j * R;
        double dot = 0;
#pragma unroll 4
        for (int r = 0; r < R; ++r) dot += a[r] * b[r];
        const double na = n1[col], nb2 = n2[j];
        term[p] = (na == 0 || nb2 == 0) ? 1e300 : dot / (na * nb2);  // 1e300: the column is skipped
      }
      __syncthreads();
      if (tid < S) {
        int delta = tid - argmin_vkey;
        delta = delta < 0 ? delta + S : delta;
        const bool in = delta <= P.radius || S - delta <= P.radius;
        if (in) {
          const int slot = delta <= P.radius ? P.radius + delta : P.radius - (S - delta);
          int n_eff = 0;
          double sum = 0;
#pragma unroll 4
          for (int col = 0; col < S; ++col) {
            const double t = term[slot * S + col];
            if (t == 1e300) continue;
            sum = sum + t;
            n_eff = n_eff + 1;
          }
          const double d = 1.0 - sum / n_eff;
          if (d < 10000000) {  // NaN (no effective column) never wins, like "cur < min"
            dv = d;
            ds = tid;
          }
        }
      }
    } else if (tid < S) {
      // is shift `tid` in the search space {argmin + ii mod S, |ii| <= radius}?
      int delta = tid - argmin_vkey;
      delta = delta < 0 ? delta + S : delta;
      const bool in = delta <= P.radius || S - delta <= P.radius;
      if (in) {
        int n_eff = 0;
        double sum = 0;
        for (int col = 0; col < S; ++col) {
          int j = col - tid;
          j = j < 0 ? j + S : j;
          const double* a = sc1 + (size_t)col * R;
          const double* b = sc2 + (size_t)j * R;
          double na = 0, nb2 = 0, dot = 0;
          for (int r = 0; r < R; ++r) {
            na += a[r] * a[r];
            nb2 += b[r] * b[r];
            dot += a[r] * b[r];
          }
          na = sqrt(na);
          nb2 = sqrt(nb2);
          if (na == 0 || nb2 == 0) continue;
          sum = sum + dot / (na * nb2);
          n_eff = n_eff + 1;
        }
        const double d = 1.0 - sum / n_eff;
        if (d < 10000000) {  // NaN (no effective column) never wins, like "cur < min"
          dv = d;
          ds = tid;
        }
      }
    }
    block_argmin(dv, ds, sv, si);
    const int argmin_shift = ds == 0x7fffffff ? 0 : ds;
    const double min_sc = ds == 0x7fffffff ? 10000000 : dv;
    const double dx = pos[2 * (size_t)ci] - pos[2 * (size_t)node], dy = pos[2 * (size_t)ci + 1] - pos[2 * (size_t)node + 1];
    double t_err = sqrt(dx * dx + dy * dy) - P.odom_eps;
    t_err = (t_err > 0.0 ? t_err : 0.0) / (dist[ci] - dist[node]);
    const double odom_dist = 1 - exp(-(t_err * t_err) / (2 * P.assumed_drift * P.assumed_drift));
    const double cd = min_sc + odom_dist * R * P.odom_weight;
    if (tid == 0) {
      rec->cd = cd;
      rec->shift = argmin_shift;
      rec->idx = ci;
    }
  }
}

// the candidates of a query in rank order: the smallest distance wins, the first one on ties (:318-327)
__global__ __launch_bounds__(64) void k_sc_pick(ScParams P, int n_queries, int n_cand, const ScCandidate* __restrict__ records,
                                                int32_t* __restrict__ loop_id, float* __restrict__ yaw, double* __restrict__ min_dist_out) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= n_queries) return;
  double min_d = 10000000;
  int nn_align = 0, nn_idx = 0;
  for (int c = 0; c < n_cand; ++c) {
    const ScCandidate r = records[(size_t)q * n_cand + c];
    if (r.idx < 0) break;  // (ranks beyond the query's candidates, or the early return)
    if (r.cd < min_d) {
      min_d = r.cd;
      nn_align = r.shift;
      nn_idx = r.idx;
    }
  }
  loop_id[q] = min_d < P.dist_thresh ? nn_idx : -1;
  yaw[q] = (float)((float)(nn_align * (360.0 / (double)P.S)) * M_PI / 180.0);
  if (min_dist_out) min_dist_out[q] = min_d;
}

ScParams to_dev(const randt_sc_params* p) {
  ScParams P;
  P.R = p->num_ring;
  P.S = p->num_sector;
  P.exclude_recent = p->num_exclude_recent;
  P.n_cand = p->num_candidates;
  P.radius = (int)round(0.5 * p->search_ratio * p->num_sector);
  P.max_radius = p->max_radius;
  P.dist_thresh = p->dist_thresh;
  P.assumed_drift = p->assumed_drift;
  P.odom_eps = p->odom_eps;
  P.odom_weight = p->odom_weight;
  P.intensity_factor = p->intensity_factor;
  return P;
}

}  // namespace

// records | candidate indices | one row of ring-key distances per query
size_t sc_detect_ws_bytes(int n_queries, int n_db, int n_cand) {
  const size_t q = n_queries > 0 ? n_queries : 1, n = n_db > 0 ? n_db : 1;
  const size_t c = n_cand < 1 ? 1 : (n_cand > SC_MAX_CAND ? SC_MAX_CAND : n_cand);  // (launch_sc_detect refuses counts outside 1 .. SC_MAX_CAND)
  return ((q * c * sizeof(ScCandidate) + 255) & ~(size_t)255) + ((q * c * sizeof(int32_t) + 255) & ~(size_t)255) + sizeof(float) * q * n + 256;
}

int launch_sc_make(randt_ctx* ctx, const float* d_points, int n_scans, int pitch, const int32_t* d_n_points, int stride, int ioff,
                   const randt_sc_params* p, double* d_desc, double* d_ring_keys, double* d_sector_keys) {
  if (n_scans <= 0) return RANDT_OK;
  if (p->num_ring < 1 || p->num_ring > SC_MAX_RING || p->num_sector < 1 || p->num_sector > SC_MAX_SECTOR)
    return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "scan context: num_ring <= 64 and num_sector <= 128", hipSuccess);
  const size_t lds = (((size_t)pitch * 8 + 15) & ~(size_t)15) + (((size_t)pitch * 4 + 15) & ~(size_t)15) + (size_t)p->num_ring * p->num_sector * 16 + 16;  // per point: value, bin, rank, sorted value; per bin: count, first, sum
  if (lds > (size_t)ctx->lds_limit) return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "scan too large for the scan-context kernel", hipSuccess);
  RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_make), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_sc_make, dim3(n_scans), dim3(SC_BLOCK), lds, ctx->stream, d_points, pitch, d_n_points, stride, ioff, to_dev(p),
                     d_desc, d_ring_keys, d_sector_keys);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}

int launch_sc_detect(randt_ctx* ctx, const randt_sc_params* p, const double* d_desc, const double* d_ring_keys, const double* d_pos,
                     const double* d_dist, int n_db, const int32_t* d_query_ids, int n_queries, float* d_ws, int32_t* d_loop_id,
                     float* d_yaw, double* d_min_dist) {
  if (n_queries <= 0) return RANDT_OK;
  if (p->num_ring < 1 || p->num_ring > SC_MAX_RING || p->num_sector < 1 || p->num_sector > SC_MAX_SECTOR || p->num_candidates < 1 ||
      p->num_candidates > SC_MAX_CAND)
    return randt_set_error(ctx, RANDT_ERR_UNSUPPORTED, "scan context: num_ring <= 64, num_sector <= 128, num_candidates <= 32", hipSuccess);
  size_t lds = 2 * sizeof(double) * (size_t)p->num_ring * p->num_sector;
  const int staged = lds + 40 * 1024 <= (size_t)ctx->lds_limit ? 1 : 0;  // (the kernel's static arrays take 37 KB)
  if (!staged) lds = 0;
  RANDT_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_detect), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // workspace (sc_detect_ws_bytes): the records, the candidate indices, one row of ring-key distances per query
  ScCandidate* records = reinterpret_cast<ScCandidate*>(d_ws);
  char* after = reinterpret_cast<char*>(d_ws) + (((size_t)n_queries * p->num_candidates * sizeof(ScCandidate) + 255) & ~(size_t)255);
  int32_t* cand = reinterpret_cast<int32_t*>(after);
  float* d2ws = reinterpret_cast<float*>(after + (((size_t)n_queries * p->num_candidates * sizeof(int32_t) + 255) & ~(size_t)255));
  hipLaunchKernelGGL(k_sc_knn, dim3(n_queries), dim3(SC_BLOCK), 0, ctx->stream, to_dev(p), d_ring_keys, n_db, d_query_ids, d2ws, n_db, cand);
  hipLaunchKernelGGL(k_sc_detect, dim3(n_queries, p->num_candidates), dim3(SC_BLOCK), lds, ctx->stream, to_dev(p), d_desc, d_ring_keys, d_pos, d_dist,
                     n_db, d_query_ids, cand, records, staged);
  hipLaunchKernelGGL(k_sc_pick, dim3((n_queries + 63) / 64), dim3(64), 0, ctx->stream, to_dev(p), n_queries, p->num_candidates, records, d_loop_id,
                     d_yaw, d_min_dist);
  RANDT_HIP_CHECK(ctx, hipGetLastError());
  return RANDT_OK;
}
