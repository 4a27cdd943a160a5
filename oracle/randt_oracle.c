/*
 * randt_oracle.c -- CPU restatement of the RaNDT-SLAM NDT scan-matching hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see randt_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math [-fopenmp] (no FMA contraction: the reference
 * is built RelWithDebInfo for generic x86-64, CMakeLists.txt:4, so its fp32 cell statistics are
 * plain IEEE mul/add).
 *
 * Citations are relative to /root/reference/ros/ndt_radar_slam/.
 */
#define _GNU_SOURCE
#include "randt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================================================ small helpers =============== */

/* static_cast<unsigned int>(double) as gcc/x86-64 compiles it (cvttsd2si to 64 bit, low 32 bits);
 * used by Map::coordinateToIndex (ndt_map.h:87-90).  Out-of-range / NaN -> 0 (SPEC DECISION: UB
 * in the reference). */
static uint32_t trunc_to_u32(double v) {
  if (!(v > -9.0e18 && v < 9.0e18)) return 0u;
  return (uint32_t)(int64_t)v;
}

/* static_cast<int>(float) with the same guard (grid.cpp:11). */
static int32_t trunc_to_i32(float v) {
  if (!(v > -2.0e9f && v < 2.0e9f)) return 0;
  return (int32_t)v;
}

/* TIMING ONLY (bench.py's cpu_baseline legs): residual blocks of ONE problem evaluated by several OpenMP threads, what
 * Ceres does with options.num_threads = hardware_concurrency() (ndt_matcher.cpp:376,461).  The cost is then a reduction
 * over threads, i.e. summed in another order than the sequential evaluation every parity test uses (default 1). */
/* how often the sqrt(0) guard of orc_ndt_residual (SPEC DECISION 4) has fired since the last reset: the reference has no such
   guard (autodiff of sqrt(0) is NaN there), so every count is an evaluation where the two would part (tests/test_oracle_eigen_bound.py) */
static long long g_sqrt_zero_count = 0;
long long orc_sqrt_zero_count(int reset) {
  const long long n = g_sqrt_zero_count;
  if (reset) g_sqrt_zero_count = 0;
  return n;
}
static int g_eval_threads = 1;
void orc_set_eval_threads(int n) { g_eval_threads = n > 1 ? n : 1; }

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ============================================================ map ========================= */

/* Map::initialize, ndt_map.cpp:7-21 */
orc_map* orc_map_create(int size_x, int size_y, double res, double center_x, double center_y,
                        double max_neighbour_dist, int min_points, int cap) {
  orc_map* m = (orc_map*)calloc(1, sizeof(orc_map));
  if (!m) return NULL;
  m->size_x = size_x;
  m->size_y = size_y;
  m->res = res;
  m->offset_x = -(double)(uint32_t)size_x / 2.0 * res + center_x;
  m->offset_y = -(double)(uint32_t)size_y / 2.0 * res + center_y;
  m->max_neighbour_dist = max_neighbour_dist;
  m->min_points = min_points;
  m->cap = cap;
  m->cells = (orc_cell*)calloc((size_t)cap, sizeof(orc_cell));
  m->grid = (int32_t*)malloc(sizeof(int32_t) * (size_t)size_x * (size_t)size_y);
  if (!m->cells || !m->grid) {
    orc_map_destroy(m);
    return NULL;
  }
  orc_map_clear(m);
  return m;
}

void orc_map_destroy(orc_map* m) {
  if (!m) return;
  free(m->cells);
  free(m->grid);
  free(m);
}

void orc_map_clear(orc_map* m) {
  size_t n = (size_t)m->size_x * (size_t)m->size_y;
  for (size_t i = 0; i < n; ++i) m->grid[i] = -1;
  m->n_cells = 0;
  m->n_dropped = 0;
}

void orc_map_copy(orc_map* dst, const orc_map* src) {
  size_t n = (size_t)src->size_x * (size_t)src->size_y;
  dst->size_x = src->size_x;
  dst->size_y = src->size_y;
  dst->res = src->res;
  dst->offset_x = src->offset_x;
  dst->offset_y = src->offset_y;
  dst->max_neighbour_dist = src->max_neighbour_dist;
  dst->min_points = src->min_points;
  dst->n_cells = src->n_cells < dst->cap ? src->n_cells : dst->cap;
  dst->n_dropped = src->n_dropped;
  memcpy(dst->cells, src->cells, sizeof(orc_cell) * (size_t)dst->n_cells);
  memcpy(dst->grid, src->grid, sizeof(int32_t) * n);
}

/* Map::coordinateToIndex + getIndex, ndt_map.h:87-90,181-184 (all unsigned 32-bit arithmetic). */
uint32_t orc_map_coord_to_index(const orc_map* m, float x, float y) {
  uint32_t mx = trunc_to_u32(((double)x - m->offset_x) / m->res);
  uint32_t my = trunc_to_u32(((double)y - m->offset_y) / m->res);
  return my * (uint32_t)m->size_x + mx;
}

/* ============================================================ clustering ================== */

/* Grid::cluster, grid.cpp:7-14 */
void orc_grid_labels(const float* pts, int n, int stride, int ioff, int n_clusters, float max_range,
                     int32_t* labels) {
  (void)ioff;
  int row_size = (int)sqrt((double)n_clusters);
  float resolution = max_range * 2 / (float)row_size;
  for (int i = 0; i < n; ++i) {
    float x = pts[(size_t)i * stride + 0];
    float y = pts[(size_t)i * stride + 1];
    labels[i] = trunc_to_i32(x / resolution) + row_size * trunc_to_i32(y / resolution);
  }
}

/* ============================================================ cell statistics ============= */

/* SPEC DECISION: closed-form fp32 symmetric 2x2 eigen-decomposition standing in for
 * Eigen::SelfAdjointEigenSolver<Matrix2f> (iterative, ndt_cell.cpp:104).  Eigenvalues ascending,
 * V = [v0 v1] column eigenvectors (orthonormal up to rounding). */
static void sym_eig2f(float a, float b, float d, float* l0, float* l1, float V[4] /* v00 v01 v10 v11 row-major */) {
  float t = 0.5f * (a + d);
  float h = 0.5f * (a - d);
  float r = sqrtf(h * h + b * b);
  *l0 = t - r;
  *l1 = t + r;
  float vx, vy; /* eigenvector of l1 */
  if (b == 0.0f) {
    if (a <= d) { vx = 0.0f; vy = 1.0f; } else { vx = 1.0f; vy = 0.0f; }
  } else {
    if (h >= 0.0f) { vx = h + r; vy = b; } else { vx = b; vy = r - h; }
    float nrm = sqrtf(vx * vx + vy * vy);
    vx = vx / nrm;
    vy = vy / nrm;
  }
  /* v0 = perpendicular */
  V[0] = vy;  V[1] = vx;
  V[2] = -vx; V[3] = vy;
}

/* Regularisation, ndt_cell.cpp:102-112 */
static void cell_regularize(orc_cell* c) {
  float l0, l1, V[4];
  sym_eig2f(c->cov[0], c->cov[1], c->cov[3], &l0, &l1, V);
  l0 = fmaxf(l0, 0.001f * l1);
  /* eig_vectors.inverse(): Eigen 2x2 inverse = adjugate * (1/det) */
  float det = V[0] * V[3] - V[1] * V[2];
  float invdet = 1.0f / det;
  float I00 = V[3] * invdet, I01 = -V[1] * invdet;
  float I10 = -V[2] * invdet, I11 = V[0] * invdet;
  /* (V * Lambda) * Vinv, left to right */
  float T00 = V[0] * l0, T01 = V[1] * l1;
  float T10 = V[2] * l0, T11 = V[3] * l1;
  c->cov[0] = T00 * I00 + T01 * I10;
  c->cov[1] = T00 * I01 + T01 * I11; /* SPEC DECISION: symmetric storage keeps element (0,1) */
  c->cov[3] = T10 * I01 + T11 * I11;
  c->cov[5] = (float)((double)c->cov[5] + 0.000001); /* new_cov_(2,2) += 0.000001 (double literal) */
}

int orc_cell_from_points_pndt(orc_cell* c, const float* pts, const int32_t* idx, int k, int stride, int ioff, int min_points,
                              const float* polar, const float* beam);

/* Cell::addPointCloud + updateCell (first-fill branch), ndt_cell.cpp:25-65,90-94,102-112.
 * idx may be NULL (points 0..k-1).  Returns 1 if accepted. */
int orc_cell_from_points(orc_cell* c, const float* pts, const int32_t* idx, int k, int stride,
                         int ioff, int min_points) {
  return orc_cell_from_points_pndt(c, pts, idx, k, stride, ioff, min_points, NULL, NULL);
}

/* The same with the pNDT branch of updateCell (ndt_cell.cpp:67-82, use_pndt = true; false in every shipped configuration,
 * "hard to tune, better not touch it"): every point adds J * beam_cov * J^T, J = d(x, y, i) / d(angle, range, i) at its
 * polar coordinates; the mean of those is added to the sample covariance and the regularisation of :102-112 is skipped.
 * polar (NULL = plain NDT): (angle, range) per point, indexed like pts; beam: row-major 3x3 (NDTCellParameters::beam_cov).
 * SPEC DECISIONS: Eigen's 3x3 float products as ((a0 b0 + a1 b1) + a2 b2) with (J * S) evaluated first, like transformCell;
 * std::sin / std::cos of the float angle are taken in double and rounded once (glibc's sinf is an IFUNC whose FMA and
 * non-FMA variants differ in the last bit, so the reference itself is not reproducible across CPUs here); the six upper
 * entries are kept. */
int orc_cell_from_points_pndt(orc_cell* c, const float* pts, const int32_t* idx, int k, int stride,
                              int ioff, int min_points, const float* polar, const float* beam) {
  memset(c, 0, sizeof(*c));
  if (!((long long)k > (long long)min_points)) return 0; /* n_points_(0) + size > min_points_per_cell_ */
  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
  double maxi = 0.0;
  for (int j = 0; j < k; ++j) {
    const float* p = pts + (size_t)(idx ? idx[j] : j) * stride;
    m0 += p[0];
    m1 += p[1];
    m2 += p[ioff];
    if ((double)p[ioff] > maxi) maxi = (double)p[ioff];
  }
  float nf = (float)(double)(uint32_t)k; /* /= static_cast<double>(n) -> Scalar(float) */
  m0 = m0 / nf;
  m1 = m1 / nf;
  m2 = m2 / nf;
  float c00 = 0.f, c11 = 0.f, c22 = 0.f, c01 = 0.f, c02 = 0.f, c12 = 0.f;
  for (int j = 0; j < k; ++j) {
    const float* p = pts + (size_t)(idx ? idx[j] : j) * stride;
    float d0 = p[0] - m0, d1 = p[1] - m1, d2 = p[ioff] - m2;
    c00 += (d0 * d0);
    c11 += (d1 * d1);
    c22 += (d2 * d2);
    c01 += (d0 * d1);
    c02 += (d0 * d2);
    c12 += (d1 * d2);
  }
  c->mean[0] = m0;
  c->mean[1] = m1;
  c->mean[2] = m2;
  c->cov[0] = c00 / nf;
  c->cov[1] = c01 / nf;
  c->cov[2] = c02 / nf;
  c->cov[3] = c11 / nf;
  c->cov[4] = c12 / nf;
  c->cov[5] = c22 / nf;
  c->n = (uint32_t)k;
  c->max_intensity = (float)maxi;
  if (polar && beam) {
    float acc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    for (int j = 0; j < k; ++j) {
      const size_t id = (size_t)(idx ? idx[j] : j);
      const float a = polar[2 * id], r = polar[2 * id + 1];
      const float sn = (float)sin((double)a), cs = (float)cos((double)a);
      const float J[3][3] = {{-r * sn, cs, 0.f}, {r * cs, sn, 0.f}, {0.f, 0.f, 1.f}};
      float JS[3][3];
      for (int i = 0; i < 3; ++i)
        for (int q = 0; q < 3; ++q) JS[i][q] = (J[i][0] * beam[0 * 3 + q] + J[i][1] * beam[1 * 3 + q]) + J[i][2] * beam[2 * 3 + q];
      for (int i = 0; i < 3; ++i)
        for (int q = i; q < 3; ++q) acc[i][q] = acc[i][q] + ((JS[i][0] * J[q][0] + JS[i][1] * J[q][1]) + JS[i][2] * J[q][2]);
    }
    c->cov[0] = c->cov[0] + acc[0][0] / nf;
    c->cov[1] = c->cov[1] + acc[0][1] / nf;
    c->cov[2] = c->cov[2] + acc[0][2] / nf;
    c->cov[3] = c->cov[3] + acc[1][1] / nf;
    c->cov[4] = c->cov[4] + acc[1][2] / nf;
    c->cov[5] = c->cov[5] + acc[2][2] / nf;
    return 1; /* :102: no regularisation with use_pndt */
  }
  cell_regularize(c); /* use_pndt = false in every shipped config */
  return 1;
}

/* Cell::addPointCloud + updateCell on a cell that may already hold a distribution (ndt_cell.cpp:25-34,36-114): batch
 * statistics as in orc_cell_from_points (unregularised), then the recursive update of :84-89 when n_points_ > 0 -- the
 * same formula as operator+= -- and the regularisation of :102-112 on the result.  Returns 1 if the points were taken
 * (n_points_ + size > min_points_per_cell_), 0 if the cell is left untouched. */
int orc_cell_update(orc_cell* c, const float* pts, int k, int stride, int ioff, int min_points) {
  if (!((long long)c->n + (long long)k > (long long)min_points) || k <= 0) return 0;
  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
  double maxi = (double)c->max_intensity;
  for (int j = 0; j < k; ++j) {
    const float* p = pts + (size_t)j * stride;
    m0 += p[0];
    m1 += p[1];
    m2 += p[ioff];
    if ((double)p[ioff] > maxi) maxi = (double)p[ioff];
  }
  float nf = (float)(double)(uint32_t)k;
  m0 = m0 / nf;
  m1 = m1 / nf;
  m2 = m2 / nf;
  float c00 = 0.f, c11 = 0.f, c22 = 0.f, c01 = 0.f, c02 = 0.f, c12 = 0.f;
  for (int j = 0; j < k; ++j) {
    const float* p = pts + (size_t)j * stride;
    float d0 = p[0] - m0, d1 = p[1] - m1, d2 = p[ioff] - m2;
    c00 += (d0 * d0);
    c11 += (d1 * d1);
    c22 += (d2 * d2);
    c01 += (d0 * d1);
    c02 += (d0 * d2);
    c12 += (d1 * d2);
  }
  orc_cell add;
  memset(&add, 0, sizeof(add));
  add.mean[0] = m0;
  add.mean[1] = m1;
  add.mean[2] = m2;
  add.cov[0] = c00 / nf;
  add.cov[1] = c01 / nf;
  add.cov[2] = c02 / nf;
  add.cov[3] = c11 / nf;
  add.cov[4] = c12 / nf;
  add.cov[5] = c22 / nf;
  add.n = (uint32_t)k;
  if (c->n > 0) {
    orc_cell_merge(c, &add); /* :84-89, unsigned (n*m)/(n+m) like operator+= */
  } else {
    *c = add;
  }
  c->max_intensity = (float)maxi;
  cell_regularize(c);
  return 1;
}

/* Cell::operator+=, ndt_cell.h:133-142 (note the integer division (n*m)/(n+m)). */
void orc_cell_merge(orc_cell* dst, const orc_cell* src) {
  uint32_t n = dst->n;
  uint64_t m = src->n; /* size_t m_n_points */
  float wn = (float)(uint32_t)(n - 1u);
  float wm = (float)(uint64_t)(m - 1u);
  float wk = (float)(uint64_t)(((uint64_t)n * m) / ((uint64_t)n + m));
  float d[3] = {dst->mean[0] - src->mean[0], dst->mean[1] - src->mean[1], dst->mean[2] - src->mean[2]};
  static const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
  float nc[6];
  for (int e = 0; e < 6; ++e) nc[e] = (wn * dst->cov[e] + wm * src->cov[e]) + wk * (d[ii[e]] * d[jj[e]]);
  float fn = (float)n, fm = (float)m, fnm = (float)((uint64_t)n + m);
  for (int e = 0; e < 3; ++e) dst->mean[e] = ((dst->mean[e] * fn) + (src->mean[e] * fm)) / fnm;
  dst->n = (uint32_t)(n + m);
  float den = (float)(uint32_t)(dst->n - 1u);
  for (int e = 0; e < 6; ++e) dst->cov[e] = nc[e] / den;
  if (src->max_intensity > dst->max_intensity) dst->max_intensity = src->max_intensity; /* not in the reference (viz only) */
}

/* Sophus::SE2d::cast<float>() (SO2 ctor normalises the complex) then .matrix() -> Eigen::Affine2f.
 * aff = [c, s, tx, ty] in float. */
void orc_pose_to_affine_f(const double pose4[4], float aff[4]) {
  float c = (float)pose4[0], s = (float)pose4[1];
  float len = sqrtf(c * c + s * s);
  aff[0] = c / len;
  aff[1] = s / len;
  aff[2] = (float)pose4[2];
  aff[3] = (float)pose4[3];
}

/* Cell::transformCell, ndt_cell.cpp:117-123.  SPEC DECISION: Eigen's Affine3f::rotation() goes
 * through a JacobiSVD of the linear part; here the rotation is taken as exactly
 * blockdiag([[c,-s],[s,c]], 1).  Products evaluated left to right, k = 0,1,2. */
void orc_cell_transform(orc_cell* cl, const float aff[4]) {
  float c = aff[0], s = aff[1];
  float x = cl->mean[0], y = cl->mean[1];
  cl->mean[0] = (c * x + (-s) * y) + aff[2];
  cl->mean[1] = (s * x + c * y) + aff[3];
  /* full symmetric 3x3 */
  float S[3][3] = {{cl->cov[0], cl->cov[1], cl->cov[2]}, {cl->cov[1], cl->cov[3], cl->cov[4]}, {cl->cov[2], cl->cov[4], cl->cov[5]}};
  float R[3][3] = {{c, -s, 0.f}, {s, c, 0.f}, {0.f, 0.f, 1.f}};
  float T[3][3], O[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i][j] = (R[i][0] * S[0][j] + R[i][1] * S[1][j]) + R[i][2] * S[2][j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O[i][j] = (T[i][0] * R[j][0] + T[i][1] * R[j][1]) + T[i][2] * R[j][2];
  cl->cov[0] = O[0][0];
  cl->cov[1] = O[0][1];
  cl->cov[2] = O[0][2];
  cl->cov[3] = O[1][1];
  cl->cov[4] = O[1][2];
  cl->cov[5] = O[2][2];
}

void orc_map_transform(orc_map* m, const float aff[4]) {
  for (int i = 0; i < m->n_cells; ++i) orc_cell_transform(&m->cells[i], aff);
}

/* ============================================================ NDT build =================== */

typedef struct { int32_t label; int32_t idx; } lbl_idx;
static int cmp_lbl_idx(const void* a, const void* b) {
  const lbl_idx* x = (const lbl_idx*)a;
  const lbl_idx* y = (const lbl_idx*)b;
  if (x->label != y->label) return x->label < y->label ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

int orc_ndt_build_pndt(orc_map* m, const float* pts, int n, int stride, int ioff, int n_clusters, float max_range,
                       const float* polar, const float* beam);

/* processScan's cluster + labelClouds (radar_preprocessor.cpp:34-37,151-169): clusters in ascending
 * label order, points inside a cluster in input order; then addClusters -> insertCluster
 * (ndt_hierarchical_map.cpp:28-33, ndt_map.cpp:238-245). */
int orc_ndt_build(orc_map* m, const float* pts, int n, int stride, int ioff, int n_clusters,
                  float max_range) {
  return orc_ndt_build_pndt(m, pts, n, stride, ioff, n_clusters, max_range, NULL, NULL);
}

/* polar / beam: see orc_cell_from_points_pndt (labelClouds hands every cluster its polar points in the same order,
 * radar_preprocessor.cpp:164-167). */
int orc_ndt_build_pndt(orc_map* m, const float* pts, int n, int stride, int ioff, int n_clusters,
                       float max_range, const float* polar, const float* beam) {
  if (n <= 0) return 0;
  int32_t* labels = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  lbl_idx* li = (lbl_idx*)malloc(sizeof(lbl_idx) * (size_t)n);
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  orc_grid_labels(pts, n, stride, ioff, n_clusters, max_range, labels);
  for (int i = 0; i < n; ++i) { li[i].label = labels[i]; li[i].idx = i; }
  qsort(li, (size_t)n, sizeof(lbl_idx), cmp_lbl_idx);
  for (int i = 0; i < n; ++i) idx[i] = li[i].idx;
  uint32_t n_slots = (uint32_t)m->size_x * (uint32_t)m->size_y;
  int start = 0;
  while (start < n) {
    int end = start + 1;
    while (end < n && li[end].label == li[start].label) ++end;
    orc_cell c;
    if (orc_cell_from_points_pndt(&c, pts, idx + start, end - start, stride, ioff, m->min_points, polar, beam)) {
      uint32_t slot = orc_map_coord_to_index(m, c.mean[0], c.mean[1]);
      if (slot < n_slots && m->n_cells < m->cap) {
        m->grid[slot] = m->n_cells; /* later cluster overwrites the slot, both cells stay (quirk A.7-5) */
        m->cells[m->n_cells++] = c;
      } else {
        m->n_dropped++; /* reference: std::vector::at throws (ndt_map.cpp:242) */
      }
    }
    start = end;
  }
  free(labels);
  free(li);
  free(idx);
  return m->n_cells;
}

/* Map::mergeMapCell, ndt_map.cpp:191-207 */
void orc_map_merge(orc_map* fixed, const orc_map* moving) {
  uint32_t n_slots = (uint32_t)fixed->size_x * (uint32_t)fixed->size_y;
  for (int i = 0; i < moving->n_cells; ++i) {
    const orc_cell* mc = &moving->cells[i];
    uint32_t slot = orc_map_coord_to_index(fixed, mc->mean[0], mc->mean[1]);
    if (slot < n_slots) {
      int32_t index = fixed->grid[slot];
      if (index >= 0) {
        orc_cell_merge(&fixed->cells[index], mc);
      } else if (fixed->n_cells < fixed->cap) {
        fixed->cells[fixed->n_cells] = *mc;
        fixed->grid[slot] = fixed->n_cells++;
      }
    }
  }
}

/* ============================================================ association ================= */
static float mahalanobis3f(const orc_cell* q, const orc_cell* f);

/* Cell::mahalanobisSquared (ndt_cell.cpp:158-162, Eigen 2x2 inverse = adjugate * (1/det)) and
 * Cell::mahalanobisSquaredIntensity (:165-169) of `self` against `subtrahend`, fp32 like the reference, returned as double. */
double orc_cell_mahalanobis(const orc_cell* self, const orc_cell* subtrahend, int use_intensity) {
  if (use_intensity) return (double)mahalanobis3f(self, subtrahend);
  float s00 = subtrahend->cov[0] + self->cov[0], s01 = subtrahend->cov[1] + self->cov[1], s11 = subtrahend->cov[3] + self->cov[3];
  float mu0 = subtrahend->mean[0] - self->mean[0], mu1 = subtrahend->mean[1] - self->mean[1];
  float det = s00 * s11 - s01 * s01;
  float invdet = 1.0f / det;
  float i00 = s11 * invdet, i01 = -s01 * invdet, i10 = -s01 * invdet, i11 = s00 * invdet;
  float r0 = mu0 * i00 + mu1 * i10, r1 = mu0 * i01 + mu1 * i11;
  return (double)(r0 * mu0 + r1 * mu1);
}

/* Eigen 3.3 Matrix3f::inverse() (cofactor formulation) followed by mu^T * inv * mu, all fp32;
 * Cell::mahalanobisSquaredIntensity, ndt_cell.cpp:172-176.  q = query (transformed moving) cell,
 * f = candidate fixed cell.  mu = f.mean - q.mean, summed = f.cov + q.cov. */
static float mahalanobis3f(const orc_cell* q, const orc_cell* f) {
  float S[3][3];
  S[0][0] = f->cov[0] + q->cov[0];
  S[0][1] = S[1][0] = f->cov[1] + q->cov[1];
  S[0][2] = S[2][0] = f->cov[2] + q->cov[2];
  S[1][1] = f->cov[3] + q->cov[3];
  S[1][2] = S[2][1] = f->cov[4] + q->cov[4];
  S[2][2] = f->cov[5] + q->cov[5];
  float mu[3] = {f->mean[0] - q->mean[0], f->mean[1] - q->mean[1], f->mean[2] - q->mean[2]};
  /* cofactor_3x3<i,j> = m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1) */
#define COF(i, j) (S[((i) + 1) % 3][((j) + 1) % 3] * S[((i) + 2) % 3][((j) + 2) % 3] - S[((i) + 1) % 3][((j) + 2) % 3] * S[((i) + 2) % 3][((j) + 1) % 3])
  float c0 = COF(0, 0), c1 = COF(1, 0), c2 = COF(2, 0);
  float det = (c0 * S[0][0] + c1 * S[1][0]) + c2 * S[2][0];
  float invdet = 1.0f / det;
  float inv[3][3];
  inv[0][0] = c0 * invdet;
  inv[0][1] = c1 * invdet;
  inv[0][2] = c2 * invdet;
  inv[1][0] = COF(0, 1) * invdet;
  inv[1][1] = COF(1, 1) * invdet;
  inv[1][2] = COF(2, 1) * invdet;
  inv[2][0] = COF(0, 2) * invdet;
  inv[2][1] = COF(1, 2) * invdet;
  inv[2][2] = COF(2, 2) * invdet;
#undef COF
  float row[3];
  for (int j = 0; j < 3; ++j) row[j] = (mu[0] * inv[0][j] + mu[1] * inv[1][j]) + mu[2] * inv[2][j];
  return (row[0] * mu[0] + row[1] * mu[1]) + row[2] * mu[2];
}

typedef struct { double dist; uint64_t idx; } target_t;
static int cmp_target(const void* a, const void* b) {
  const target_t* x = (const target_t*)a;
  const target_t* y = (const target_t*)b;
  if (x->dist < y->dist) return -1;
  if (y->dist < x->dist) return 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* Map::getAdjacentIndizes, ndt_map.cpp:163-175: x-offset outer, y-offset inner, unsigned wrap,
 * no row-wrap guard, de-duplicated (first occurrence kept). Returns count. */
static int adjacent_indices(const orc_map* m, uint32_t index, int r, uint32_t* out) {
  uint32_t n_slots = (uint32_t)m->size_x * (uint32_t)m->size_y;
  int cnt = 0;
  for (int i = -r; i <= r; ++i) {
    for (int j = -r; j <= r; ++j) {
      uint32_t ni = index + (uint32_t)i + (uint32_t)j * (uint32_t)m->size_x;
      if (ni < n_slots) {
        int dup = 0;
        for (int t = 0; t < cnt; ++t)
          if (out[t] == ni) { dup = 1; break; }
        if (!dup) out[cnt++] = ni;
      }
    }
  }
  return cnt;
}

/* Map::getClosestCells (both overloads), ndt_map.cpp:101-151.  qcell: the transformed query cell
 * (mean used for the centre slot; full cell for the distribution metric).  metric 1 = Mahalanobis
 * 3-D fp32, 0 = Euclidean 2-D fp32.  Writes up to k compact indices, returns count. */
static int closest_cells(const orc_map* m, const orc_cell* qcell, int k, int metric, int32_t* out) {
  uint32_t n_slots = (uint32_t)m->size_x * (uint32_t)m->size_y;
  uint32_t center = orc_map_coord_to_index(m, qcell->mean[0], qcell->mean[1]);
  int rmax = (int)(m->max_neighbour_dist / m->res);
  int radius = 0;
  int wcap = (2 * (rmax > 0 ? rmax : 1) + 1);
  wcap = wcap * wcap + 8;
  uint32_t* adj = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)wcap);
  target_t* targets = (target_t*)malloc(sizeof(target_t) * (size_t)wcap);
  int nt = 0, nadj = 0;
  while ((uint64_t)nt < (uint64_t)(int64_t)k && (uint32_t)nadj < n_slots) {
    nt = 0;
    nadj = adjacent_indices(m, center, radius, adj);
    for (int i = 0; i < nadj; ++i) {
      int32_t ci = m->grid[adj[i]];
      if (ci >= 0) {
        const orc_cell* fc = &m->cells[ci];
        double dist;
        if (metric) {
          dist = (double)mahalanobis3f(qcell, fc);
        } else {
          float dx = qcell->mean[0] - fc->mean[0], dy = qcell->mean[1] - fc->mean[1];
          dist = (double)sqrtf(dx * dx + dy * dy);
        }
        targets[nt].dist = dist;
        targets[nt].idx = (uint64_t)ci;
        ++nt;
      }
    }
    ++radius;
    if (radius >= rmax) break;
  }
  qsort(targets, (size_t)nt, sizeof(target_t), cmp_target);
  int cnt = k < nt ? k : nt;
  if (cnt < 0) cnt = 0;
  for (int i = 0; i < cnt; ++i) out[i] = (int32_t)targets[i].idx;
  free(adj);
  free(targets);
  return cnt;
}

/* Association part of Matcher::addNDTFactor, ndt_matcher.cpp:200-215,249-253. */
int orc_associate(const orc_map* fixed, const orc_map* moving, const double pose4[4], int k,
                  int lookup_mahalanobis, int use_intensity, int32_t* corr) {
  float aff[4];
  orc_pose_to_affine_f(pose4, aff);
  int total = 0;
  for (int i = 0; i < moving->n_cells; ++i) {
    int32_t* out = corr + (size_t)i * k;
    for (int j = 0; j < k; ++j) out[j] = -1;
    orc_cell q = moving->cells[i];
    int cnt;
    if (use_intensity && lookup_mahalanobis) {
      orc_cell_transform(&q, aff);
      cnt = closest_cells(fixed, &q, k, 1, out);
    } else {
      /* query_vector = initial_guess.cast<float>() * mean.xy  (SO2f * p + t) */
      float x = q.mean[0], y = q.mean[1];
      q.mean[0] = (aff[0] * x - aff[1] * y) + aff[2];
      q.mean[1] = (aff[1] * x + aff[0] * y) + aff[3];
      cnt = closest_cells(fixed, &q, k, 0, out);
    }
    total += cnt;
  }
  return total;
}

/* ============================================================ loss ======================== */

/* BarronLoss ctor (ceres_loss_functions.h:27-35) + Evaluate (ceres_loss_functions.cpp:19-39),
 * then ceres::ScaledLoss (multiplies rho[0..2] by weight). */
void orc_barron_scaled(double s, double scale_a, double alpha, double mu, double weight, double rho[3]) {
  const double a_ = alpha;
  const double b_ = mu * scale_a * scale_a;
  const double c_ = 1 / b_;
  const double factor_ = fabs(a_ - 2.0);
  const double exponent_ = 0.5 * a_;
  const double pre_factor_ = b_ * factor_ / a_;
  const double times_s_ = 2 * c_ / factor_;
  if (a_ >= 2.0) {
    rho[0] = s;
    rho[1] = 1;
    rho[2] = 0;
  } else if (fabs(a_) <= 0.05) {
    const double sum = 1.0 + s * c_;
    const double inv = 1.0 / sum;
    rho[0] = b_ * log(sum);
    rho[1] = inv > DBL_MIN ? inv : DBL_MIN;
    rho[2] = -c_ * (inv * inv);
  } else {
    const double to_exp = s * times_s_ + 1.0;
    rho[0] = pre_factor_ * (pow(to_exp, exponent_) - 1.);
    rho[1] = pre_factor_ * exponent_ * pow(to_exp, exponent_ - 1.) * times_s_;
    rho[2] = pre_factor_ * exponent_ * (exponent_ - 1) * pow(to_exp, exponent_ - 2.) * times_s_ * times_s_;
  }
  rho[0] *= weight;
  rho[1] *= weight;
  rho[2] *= weight;
}

/* ============================================================ SE(2) (Sophus 1.22.10) ====== */

static void so2_normalize(double* c, double* s) {
  double len = sqrt((*c) * (*c) + (*s) * (*s));
  *c = *c / len;
  *s = *s / len;
}

/* SO2Base::operator*: complex product, first-order renormalisation, then the (real,imag) ctor's
 * normalize(). */
static void so2_mul(double ar, double ai, double br, double bi, double* rr, double* ri) {
  double re = ar * br - ai * bi;
  double im = ar * bi + ai * br;
  double sq = re * re + im * im;
  if (sq != 1.0) {
    double scale = 2.0 / (1.0 + sq);
    re *= scale;
    im *= scale;
  }
  so2_normalize(&re, &im);
  *rr = re;
  *ri = im;
}

void orc_se2_exp(const double xi[3], double out[4]) {
  double theta = xi[2];
  double c = cos(theta), s = sin(theta);
  so2_normalize(&c, &s);
  double sbt, omcbt;
  if (fabs(theta) < 1e-10) {
    double tsq = theta * theta;
    sbt = 1.0 - (1.0 / 6.0) * tsq;
    omcbt = 0.5 * theta - (1.0 / 24.0) * theta * tsq;
  } else {
    sbt = s / theta;
    omcbt = (1.0 - c) / theta;
  }
  out[0] = c;
  out[1] = s;
  out[2] = sbt * xi[0] - omcbt * xi[1];
  out[3] = omcbt * xi[0] + sbt * xi[1];
}

void orc_se2_log(const double p[4], double xi[3]) {
  double theta = atan2(p[1], p[0]);
  double half = 0.5 * theta;
  double rm1 = p[0] - 1.0;
  double hbt;
  if (fabs(rm1) < 1e-10) {
    hbt = 1.0 - (1.0 / 12) * theta * theta;
  } else {
    hbt = -(half * p[1]) / rm1;
  }
  xi[0] = hbt * p[2] + half * p[3];
  xi[1] = -half * p[2] + hbt * p[3];
  xi[2] = theta;
}

void orc_se2_mul(const double a[4], const double b[4], double out[4]) {
  double rr, ri;
  so2_mul(a[0], a[1], b[0], b[1], &rr, &ri);
  double tx = a[2] + (a[0] * b[2] - a[1] * b[3]);
  double ty = a[3] + (a[1] * b[2] + a[0] * b[3]);
  out[0] = rr;
  out[1] = ri;
  out[2] = tx;
  out[3] = ty;
}

void orc_se2_inv(const double a[4], double out[4]) {
  double c = a[0], s = -a[1];
  double tx = -a[2], ty = -a[3];
  out[0] = c;
  out[1] = s;
  out[2] = c * tx - s * ty;
  out[3] = s * tx + c * ty;
}

/* ============================================================ residual ==================== */

/* NDTFrameToMap{,Intensity}FactorResidual{,SE2} (ceres_residuals.h:421-552) with the Jacobian that
 * Ceres autodiff (+ Sophus::Manifold<SE2>::PlusJacobian for MANIFOLD) produces -- SURVEY A.2.
 * theta = atan2(s, c) of the stored complex; R built from cos/sin(theta) (Eigen::AngleAxis /
 * Rotation2D); intensity (3rd coordinate) is neither rotated nor translated. */
double orc_ndt_residual(int d, int parameterization, const double* pose4, const double* mm,
                        const double* mc, const double* fm, const double* fc, double* jac) {
  const double cp = pose4[0], sp = pose4[1];
  const double theta = atan2(sp, cp);
  const double c = cos(theta), s = sin(theta);
  double R[3][3] = {{c, -s, 0}, {s, c, 0}, {0, 0, 1}};
  double dv[3] = {0, 0, 0}, C[3][3] = {{0}}, RS[3][3] = {{0}};
  /* RS = R * Sm ; C = RS * R^T + Sf */
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      double a = 0;
      for (int k = 0; k < d; ++k) a += R[i][k] * mc[k * d + j];
      RS[i][j] = a;
    }
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      double a = 0;
      for (int k = 0; k < d; ++k) a += RS[i][k] * R[j][k];
      C[i][j] = a + fc[i * d + j];
    }
  double t[3] = {pose4[2], pose4[3], 0};
  for (int i = 0; i < d; ++i) {
    double a = 0;
    for (int k = 0; k < d; ++k) a += R[i][k] * mm[k];
    dv[i] = a + t[i] - fm[i];
  }
  double q[3] = {0, 0, 0};
  if (d == 3) {
    double c00 = C[1][1] * C[2][2] - C[1][2] * C[2][1];
    double c01 = C[1][2] * C[2][0] - C[1][0] * C[2][2];
    double c02 = C[1][0] * C[2][1] - C[1][1] * C[2][0];
    double det = C[0][0] * c00 + C[0][1] * c01 + C[0][2] * c02;
    double id = 1.0 / det;
    double inv[3][3];
    inv[0][0] = c00 * id;
    inv[0][1] = (C[0][2] * C[2][1] - C[0][1] * C[2][2]) * id;
    inv[0][2] = (C[0][1] * C[1][2] - C[0][2] * C[1][1]) * id;
    inv[1][0] = c01 * id;
    inv[1][1] = (C[0][0] * C[2][2] - C[0][2] * C[2][0]) * id;
    inv[1][2] = (C[0][2] * C[1][0] - C[0][0] * C[1][2]) * id;
    inv[2][0] = c02 * id;
    inv[2][1] = (C[0][1] * C[2][0] - C[0][0] * C[2][1]) * id;
    inv[2][2] = (C[0][0] * C[1][1] - C[0][1] * C[1][0]) * id;
    for (int i = 0; i < 3; ++i) q[i] = inv[i][0] * dv[0] + inv[i][1] * dv[1] + inv[i][2] * dv[2];
  } else {
    double det = C[0][0] * C[1][1] - C[0][1] * C[1][0];
    double id = 1.0 / det;
    q[0] = (C[1][1] * dv[0] - C[0][1] * dv[1]) * id;
    q[1] = (-C[1][0] * dv[0] + C[0][0] * dv[1]) * id;
  }
  double ssq = dv[0] * q[0] + dv[1] * q[1] + dv[2] * q[2];
  double r = sqrt(ssq);
  if (jac) {
    int nj = parameterization == ORC_PARAM_AMBIENT4 ? 4 : 3;
    if (!(r > 0.0)) {
      /* SPEC DECISION: autodiff of sqrt(0) is inf/NaN in the reference (ceres_residuals.h:545);
       * guard with a zero Jacobian row. */
#ifdef _OPENMP
#pragma omp atomic
#endif
      g_sqrt_zero_count += 1;
      for (int i = 0; i < nj; ++i) jac[i] = 0.0;
      return r;
    }
    /* u = R^T q ; dr/dtheta = (u^T J m - u^T J Sm u) / r with J the so(2) generator */
    double u[3] = {c * q[0] + s * q[1], -s * q[0] + c * q[1], q[2]};
    double Su0 = 0, Su1 = 0;
    for (int k = 0; k < d; ++k) {
      Su0 += mc[0 * d + k] * u[k];
      Su1 += mc[1 * d + k] * u[k];
    }
    double dtheta = ((u[1] * mm[0] - u[0] * mm[1]) - (u[1] * Su0 - u[0] * Su1)) / r;
    if (parameterization == ORC_PARAM_MANIFOLD) {
      /* ambient row [dr/dc, dr/ds, q0/r, q1/r] times PlusJacobian [[0,0,-s],[0,0,c],[c,-s,0],[s,c,0]]
       * built from the STORED complex (cp, sp) */
      double n2 = cp * cp + sp * sp;
      double dc = dtheta * (-sp / n2), ds = dtheta * (cp / n2);
      double dtx = q[0] / r, dty = q[1] / r;
      jac[0] = dtx * cp + dty * sp;
      jac[1] = -dtx * sp + dty * cp;
      jac[2] = dc * (-sp) + ds * cp;
    } else if (parameterization == ORC_PARAM_AMBIENT4) {
      double n2 = cp * cp + sp * sp;
      jac[0] = dtheta * (-sp / n2);
      jac[1] = dtheta * (cp / n2);
      jac[2] = q[0] / r;
      jac[3] = q[1] / r;
    } else if (parameterization == ORC_PARAM_ANALYTIC) {
      /* NDTFrameToMap{,Intensity}FactorResidualAnalytic (ceres_residuals.h:207-305), AS WRITTEN: the second term of the rotation
       * Jacobian is  d^T C^-1 R Sm (R J) C^-1 d  (:245, :295) where  -d^T C^-1 (R J) Sm R^T C^-1 d  would be correct -- equal only
       * at theta = 0.  With u = R^T q and w = R J q:  (u^T J m + (Sm u)^T w) / r.  Reproduced because
       * use_analytic_expressions_for_optimization: true makes Ceres iterate with exactly this row. */
      const double w0 = c * (-q[1]) - s * q[0], w1 = s * (-q[1]) + c * q[0];
      jac[0] = q[0] / r;
      jac[1] = q[1] / r;
      jac[2] = ((u[1] * mm[0] - u[0] * mm[1]) + (Su0 * w0 + Su1 * w1)) / r;
    } else {
      jac[0] = q[0] / r;
      jac[1] = q[1] / r;
      jac[2] = dtheta;
    }
  }
  return r;
}

/* ============================================================ generic LM (Ceres 2.1.0) ==== */

#define ORC_MAX_TANGENT 160 /* windows of up to 15 optimised states: 9 S + 5 tangent dimensions */

typedef struct orc_problem {
  int n_ambient, n_tangent, n_res;
  void* user;
  /* cost = sum 1/2 rho(s); residuals/jac are the loss-corrected ones (corrector.cc). */
  int (*eval)(void* user, const double* x, double* cost, double* residuals, double* jac);
  void (*plus)(void* user, const double* x, const double* delta, double* x_plus);
} orc_problem;

/* Householder QR least squares min |A y - b| (Eigen HouseholderQR semantics, unblocked).
 * A is m x n column-major (destroyed), b length m (destroyed), y length n. */
static int qr_least_squares(double* A, double* b, int m, int n, double* y) {
  for (int k = 0; k < n; ++k) {
    double* col = A + (size_t)k * m;
    double c0 = col[k];
    double tail = 0.0;
    for (int i = k + 1; i < m; ++i) tail += col[i] * col[i];
    double tau, beta;
    if (tail <= DBL_MIN) {
      tau = 0.0;
      beta = c0;
      for (int i = k + 1; i < m; ++i) col[i] = 0.0;
    } else {
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= 0.0) beta = -beta;
      double den = c0 - beta;
      for (int i = k + 1; i < m; ++i) col[i] /= den;
      tau = (beta - c0) / beta;
    }
    col[k] = beta;
    if (tau != 0.0) {
      /* apply H = I - tau v v^T (v = [1; essential]) to remaining columns and b */
      for (int j = k + 1; j <= n; ++j) {
        double* t = (j < n) ? (A + (size_t)j * m) : b;
        double dot = t[k];
        for (int i = k + 1; i < m; ++i) dot += col[i] * t[i];
        dot *= tau;
        t[k] -= dot;
        for (int i = k + 1; i < m; ++i) t[i] -= dot * col[i];
      }
    }
  }
  for (int k = n - 1; k >= 0; --k) {
    double acc = b[k];
    for (int j = k + 1; j < n; ++j) acc -= A[(size_t)j * m + k] * y[j];
    double dk = A[(size_t)k * m + k];
    if (dk == 0.0) return 0;
    y[k] = acc / dk;
  }
  return 1;
}

/* Cholesky solve of the n x n SPD system (row-major H, destroyed). */
static int chol_solve(double* H, double* g, int n, double* y) {
  for (int j = 0; j < n; ++j) {
    double d = H[j * n + j];
    for (int k = 0; k < j; ++k) d -= H[j * n + k] * H[j * n + k];
    if (!(d > 0.0)) return 0;
    d = sqrt(d);
    H[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double a = H[i * n + j];
      for (int k = 0; k < j; ++k) a -= H[i * n + k] * H[j * n + k];
      H[i * n + j] = a / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double a = g[i];
    for (int k = 0; k < i; ++k) a -= H[i * n + k] * y[k];
    y[i] = a / H[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double a = y[i];
    for (int k = i + 1; k < n; ++k) a -= H[k * n + i] * y[k];
    y[i] = a / H[i * n + i];
  }
  return 1;
}

static void trace_push(orc_solve_stats* st, double cost, double radius, int flag) {
  if (st && st->trace_len < ORC_TRACE_MAX) {
    st->trace_cost[st->trace_len] = cost;
    st->trace_radius[st->trace_len] = radius;
    st->trace_flag[st->trace_len] = flag;
    st->trace_len++;
  }
}

/* TrustRegionMinimizer::Minimize with LevenbergMarquardtStrategy + DENSE_QR, monotonic steps,
 * jacobi scaling, no bounds, no inner iterations (Ceres 2.1.0).
 * x (ambient) is updated to the best accepted point; returns termination type. */
static int lm_minimize(const orc_problem* P, const orc_matcher_params* opt, double* x_user,
                       orc_solve_stats* st, double* final_cost_out) {
  const int na = P->n_ambient, nt = P->n_tangent, nr = P->n_res;
  double* residuals = (double*)malloc(sizeof(double) * (size_t)(nr + nt));
  double* jac = (double*)malloc(sizeof(double) * (size_t)nr * nt);      /* row-major nr x nt, Jacobi-scaled after eval */
  double* work = (double*)malloc(sizeof(double) * (size_t)(nr + nt) * nt); /* column-major stacked [J; D] */
  double* rhs = (double*)malloc(sizeof(double) * (size_t)(nr + nt));
  double* model = (double*)malloc(sizeof(double) * (size_t)(nr > 0 ? nr : 1));
  double x[8 * ORC_MAX_TANGENT], cand[8 * ORC_MAX_TANGENT], pg[8 * ORC_MAX_TANGENT];
  double gradient[ORC_MAX_TANGENT], scaling[ORC_MAX_TANGENT], diagonal[ORC_MAX_TANGENT], lmdiag[ORC_MAX_TANGENT];
  double step[ORC_MAX_TANGENT], delta[ORC_MAX_TANGENT], neg[ORC_MAX_TANGENT];
  int term = ORC_TERM_FAILURE;
  memcpy(x, x_user, sizeof(double) * na);

  double x_cost = 0, x_norm = 0, cand_cost = 0;
  double radius = opt->initial_radius, decrease_factor = 2.0;
  int reuse_diagonal = 0;
  int num_invalid = 0;
  double minimum_cost = DBL_MAX; /* best accepted cost -> user parameters */
  double summary_min_cost;       /* SetSummaryFinalCost: min over all logged iteration costs */
  double grad_max_norm = 0;
  int iteration = 0;
  int step_successful;

  /* ---- IterationZero -> EvaluateGradientAndJacobian */
#define EVAL_GRAD_JAC(FIRST)                                                                    \
  do {                                                                                          \
    if (!P->eval(P->user, x, &x_cost, residuals, jac)) { term = ORC_TERM_FAILURE; goto done; }  \
    if (st) st->n_jac_evals++;                                                                  \
    for (int j = 0; j < nt; ++j) {                                                              \
      double g = 0;                                                                             \
      for (int i = 0; i < nr; ++i) g += jac[(size_t)i * nt + j] * residuals[i];                 \
      gradient[j] = g;                                                                          \
    }                                                                                           \
    if (FIRST) {                                                                                \
      for (int j = 0; j < nt; ++j) {                                                            \
        double sq = 0;                                                                          \
        for (int i = 0; i < nr; ++i) sq += jac[(size_t)i * nt + j] * jac[(size_t)i * nt + j];   \
        scaling[j] = 1.0 / (1.0 + sqrt(sq));                                                    \
      }                                                                                         \
    }                                                                                           \
    for (int i = 0; i < nr; ++i)                                                                \
      for (int j = 0; j < nt; ++j) jac[(size_t)i * nt + j] *= scaling[j];                       \
    for (int j = 0; j < nt; ++j) neg[j] = -gradient[j];                                         \
    P->plus(P->user, x, neg, pg);                                                               \
    grad_max_norm = 0;                                                                          \
    for (int i = 0; i < na; ++i) {                                                              \
      double a = fabs(x[i] - pg[i]);                                                            \
      if (a > grad_max_norm) grad_max_norm = a;                                                 \
    }                                                                                           \
  } while (0)

  x_norm = 0;
  for (int i = 0; i < na; ++i) x_norm += x[i] * x[i];
  x_norm = sqrt(x_norm);
  EVAL_GRAD_JAC(1);
  if (st && st->n_solves == 0) st->initial_cost = x_cost;
  summary_min_cost = x_cost;
  step_successful = 1;
  trace_push(st, x_cost, radius, 0);
  if (st) st->n_iterations++;

  for (;;) {
    /* ---- FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (step_successful && x_cost < minimum_cost) {
      minimum_cost = x_cost;
      memcpy(x_user, x, sizeof(double) * na);
    }
    if (iteration >= opt->max_iterations) { term = ORC_TERM_NO_CONVERGENCE; break; }
    if (step_successful && grad_max_norm <= opt->gradient_tolerance) { term = ORC_TERM_CONVERGENCE_GRADIENT; break; }
    if (radius <= opt->min_radius) { term = ORC_TERM_CONVERGENCE_RADIUS; break; }
    ++iteration;
    if (st) st->n_iterations++;

    /* ---- ComputeTrustRegionStep: LevenbergMarquardtStrategy::ComputeStep */
    if (!reuse_diagonal) {
      for (int j = 0; j < nt; ++j) {
        double sq = 0;
        for (int i = 0; i < nr; ++i) sq += jac[(size_t)i * nt + j] * jac[(size_t)i * nt + j];
        diagonal[j] = fmin(fmax(sq, opt->min_lm_diagonal), opt->max_lm_diagonal);
      }
    }
    for (int j = 0; j < nt; ++j) lmdiag[j] = sqrt(diagonal[j] / radius);
    int solved;
    if (opt->linear_solver == ORC_LINSOLVE_QR) {
      const int m = nr + nt;
      for (int j = 0; j < nt; ++j) {
        for (int i = 0; i < nr; ++i) work[(size_t)j * m + i] = jac[(size_t)i * nt + j];
        for (int i = 0; i < nt; ++i) work[(size_t)j * m + nr + i] = (i == j) ? lmdiag[j] : 0.0;
      }
      memcpy(rhs, residuals, sizeof(double) * nr);
      for (int i = 0; i < nt; ++i) rhs[nr + i] = 0.0;
      solved = qr_least_squares(work, rhs, m, nt, step);
    } else {
      double H[ORC_MAX_TANGENT * ORC_MAX_TANGENT], g[ORC_MAX_TANGENT];
      for (int a = 0; a < nt; ++a) {
        for (int b = 0; b < nt; ++b) {
          double h = 0;
          for (int i = 0; i < nr; ++i) h += jac[(size_t)i * nt + a] * jac[(size_t)i * nt + b];
          H[a * nt + b] = h + (a == b ? lmdiag[a] * lmdiag[a] : 0.0);
        }
        double gg = 0;
        for (int i = 0; i < nr; ++i) gg += jac[(size_t)i * nt + a] * residuals[i];
        g[a] = gg;
      }
      solved = chol_solve(H, g, nt, step);
    }
    for (int j = 0; j < nt; ++j) {
      if (!isfinite(step[j])) solved = 0;
      step[j] = -step[j];
    }
    reuse_diagonal = 1;
    int step_valid = 0;
    double model_cost_change = 0;
    if (solved) {
      /* model_cost_change = -(J step)^T (r + J step / 2) */
      double acc = 0;
      for (int i = 0; i < nr; ++i) {
        double mr = 0;
        for (int j = 0; j < nt; ++j) mr += jac[(size_t)i * nt + j] * step[j];
        model[i] = mr;
        acc += mr * (residuals[i] + mr / 2.0);
      }
      model_cost_change = -acc;
      step_valid = model_cost_change > 0.0;
    }
    if (!step_valid) {
      /* ---- HandleInvalidStep */
      if (++num_invalid >= opt->max_consecutive_invalid_steps) { term = ORC_TERM_FAILURE; break; }
      radius = radius / decrease_factor; /* StepIsInvalid -> StepRejected(0) */
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
      step_successful = 0;
      if (x_cost < summary_min_cost) summary_min_cost = x_cost;
      trace_push(st, x_cost, radius, 3);
      continue;
    }
    num_invalid = 0;
    for (int j = 0; j < nt; ++j) delta[j] = step[j] * scaling[j];

    /* ---- ComputeCandidatePointAndEvaluateCost */
    P->plus(P->user, x, delta, cand);
    if (!P->eval(P->user, cand, &cand_cost, NULL, NULL)) cand_cost = DBL_MAX;
    if (st) st->n_cost_evals++;

    /* ---- ParameterToleranceReached */
    double step_norm = 0;
    for (int i = 0; i < na; ++i) step_norm += (x[i] - cand[i]) * (x[i] - cand[i]);
    step_norm = sqrt(step_norm);
    if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { term = ORC_TERM_CONVERGENCE_PARAMETER; break; }
    /* ---- FunctionToleranceReached */
    double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= opt->function_tolerance * x_cost) { term = ORC_TERM_CONVERGENCE_FUNCTION; break; }

    /* ---- IsStepSuccessful (monotonic TrustRegionStepEvaluator) */
    double relative_decrease = (cand_cost >= DBL_MAX) ? -DBL_MAX : cost_change / model_cost_change;
    if (relative_decrease > opt->min_relative_decrease) {
      /* ---- HandleSuccessfulStep */
      memcpy(x, cand, sizeof(double) * na);
      x_norm = 0;
      for (int i = 0; i < na; ++i) x_norm += x[i] * x[i];
      x_norm = sqrt(x_norm);
      EVAL_GRAD_JAC(0);
      step_successful = 1;
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3));
      radius = fmin(opt->max_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = 0;
      if (x_cost < summary_min_cost) summary_min_cost = x_cost;
      trace_push(st, x_cost, radius, 1);
    } else {
      step_successful = 0;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
      if (cand_cost < summary_min_cost) summary_min_cost = cand_cost;
      trace_push(st, cand_cost, radius, 2);
    }
  }
#undef EVAL_GRAD_JAC
done:
  if (final_cost_out) *final_cost_out = (term == ORC_TERM_FAILURE && minimum_cost == DBL_MAX) ? x_cost : summary_min_cost;
  free(residuals);
  free(jac);
  free(work);
  free(rhs);
  free(model);
  return term;
}

/* ============================================================ pair problem ================ */

typedef struct pair_user {
  int d, parameterization, n;
  const double* mm; /* n x d   */
  const double* mc; /* n x d*d */
  const double* fm;
  const double* fc;
  int apply_loss;
  double loss_a, loss_alpha, loss_mu, loss_w;
} pair_user;

static void x_to_pose4(const pair_user* u, const double* x, double p4[4]) {
  if (u->parameterization == ORC_PARAM_VECTOR || u->parameterization == ORC_PARAM_ANALYTIC) {
    /* NormalizeAngle(rot) then cos/sin: residual only depends on theta mod 2pi */
    p4[0] = cos(x[2]);
    p4[1] = sin(x[2]);
    p4[2] = x[0];
    p4[3] = x[1];
  } else {
    memcpy(p4, x, sizeof(double) * 4);
  }
}

/* ResidualBlock::Evaluate + Corrector (Ceres 2.1.0 residual_block.cc, corrector.cc). */
/* one residual block of the pair problem; returns 0 on a non-finite residual */
static int pair_eval_block(const pair_user* u, const double* p4, int i, int nj, double* total, double* residuals, double* jac) {
  double j[4];
  int d = u->d;
  double r = orc_ndt_residual(d, u->parameterization, p4, u->mm + (size_t)i * d, u->mc + (size_t)i * d * d,
                              u->fm + (size_t)i * d, u->fc + (size_t)i * d * d, jac ? j : NULL);
  if (!isfinite(r)) return 0;
  double sq = r * r;
  if (!u->apply_loss) {
    *total += 0.5 * sq;
    if (residuals) residuals[i] = r;
    if (jac) for (int a = 0; a < nj; ++a) jac[(size_t)i * nj + a] = j[a];
    return 1;
  }
  double rho[3];
  orc_barron_scaled(sq, u->loss_a, u->loss_alpha, u->loss_mu, u->loss_w, rho);
  *total += 0.5 * rho[0];
  if (residuals || jac) {
    double sqrt_rho1 = sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq == 0.0 || rho[2] <= 0.0) {
      residual_scaling = sqrt_rho1;
      alpha_sq_norm = 0.0;
    } else {
      const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
      const double alpha = 1.0 - sqrt(D);
      residual_scaling = sqrt_rho1 / (1 - alpha);
      alpha_sq_norm = alpha / sq;
    }
    if (jac) {
      /* CorrectJacobian: J = sqrt_rho1 * (J - alpha_sq_norm * r r^T J); one residual row */
      for (int a = 0; a < nj; ++a) {
        double ja = j[a];
        if (alpha_sq_norm != 0.0) ja = ja - alpha_sq_norm * r * (r * ja);
        jac[(size_t)i * nj + a] = sqrt_rho1 * ja;
      }
    }
    if (residuals) residuals[i] = residual_scaling * r;
  }
  return 1;
}

/* ResidualBlock::Evaluate + Corrector (Ceres 2.1.0 residual_block.cc, corrector.cc). */
static int pair_eval(void* user, const double* x, double* cost, double* residuals, double* jac) {
  const pair_user* u = (const pair_user*)user;
  const int nj = u->parameterization == ORC_PARAM_AMBIENT4 ? 4 : 3;
  double p4[4];
  x_to_pose4(u, x, p4);
  double total = 0;
#ifdef _OPENMP
  if (g_eval_threads > 1) { /* timing only: see orc_set_eval_threads */
    int bad = 0;
#pragma omp parallel for schedule(static) num_threads(g_eval_threads) reduction(+ : total) reduction(| : bad)
    for (int i = 0; i < u->n; ++i)
      if (!pair_eval_block(u, p4, i, nj, &total, residuals, jac)) bad |= 1;
    if (bad) return 0;
    *cost = total;
    return 1;
  }
#endif
  for (int i = 0; i < u->n; ++i)
    if (!pair_eval_block(u, p4, i, nj, &total, residuals, jac)) return 0;
  *cost = total;
  return 1;
}

static void pair_plus(void* user, const double* x, const double* delta, double* xp) {
  const pair_user* u = (const pair_user*)user;
  if (u->parameterization == ORC_PARAM_MANIFOLD) {
    double e[4];
    orc_se2_exp(delta, e);
    orc_se2_mul(x, e, xp); /* Sophus::Manifold<SE2>::Plus: T * exp(delta) */
  } else if (u->parameterization == ORC_PARAM_AMBIENT4) {
    for (int i = 0; i < 4; ++i) xp[i] = x[i] + delta[i];
  } else {
    for (int i = 0; i < 3; ++i) xp[i] = x[i] + delta[i];
  }
}

void orc_matcher_params_default(orc_matcher_params* p) {
  memset(p, 0, sizeof(*p));
  /* config/parameters_indoor.yaml:24-39, base yaml :44-48; loop closure values :7,9 */
  p->loss_scale = 1.5;
  p->mu_scale = 1.5;
  p->loss_alpha = -2.0;
  p->loss_weight = 1.0;
  p->gnc_divisor = 1.3;
  p->gnc_steps = 2;
  p->max_iterations = 200;
  p->n_neighbours = 4;
  p->lookup_mahalanobis = 1;
  p->use_intensity = 1;
  p->parameterization = ORC_PARAM_AMBIENT4;
  p->linear_solver = ORC_LINSOLVE_QR;
  p->max_consecutive_invalid_steps = 5;
  p->function_tolerance = 1e-6;
  p->gradient_tolerance = 1e-10;
  p->parameter_tolerance = 1e-8;
  p->initial_radius = 1e4;
  p->max_radius = 1e16;
  p->min_radius = 1e-32;
  p->min_relative_decrease = 1e-3;
  p->min_lm_diagonal = 1e-6;
  p->max_lm_diagonal = 1e32;
}

static void cell_cov_full(const orc_cell* c, int d, double* out) {
  if (d == 3) {
    out[0] = c->cov[0]; out[1] = c->cov[1]; out[2] = c->cov[2];
    out[3] = c->cov[1]; out[4] = c->cov[3]; out[5] = c->cov[4];
    out[6] = c->cov[2]; out[7] = c->cov[4]; out[8] = c->cov[5];
  } else {
    out[0] = c->cov[0]; out[1] = c->cov[1];
    out[2] = c->cov[1]; out[3] = c->cov[3];
  }
}

/* GNC loop + Solve of Matcher::estimateLoopConstraint (ndt_matcher.cpp:466-492); the same loop
 * serves estimateTransformCeres' NDT-only special case (:382-397) via loss_weight / mu_scale. */
int orc_solve_pair(const orc_map* fixed, const orc_map* moving, const int32_t* corr, int k,
                   const orc_matcher_params* p, double pose4[4], orc_solve_stats* st) {
  const int d = p->use_intensity ? 3 : 2;
  int n = 0;
  for (int i = 0; i < moving->n_cells; ++i)
    for (int j = 0; j < k; ++j)
      if (corr[(size_t)i * k + j] >= 0) ++n;
  if (st) {
    memset(st, 0, sizeof(*st));
    st->n_residuals = n;
  }
  if (n == 0) return 1; /* "WARNING: NO RESIDUALS ADDED!" (ndt_matcher.cpp:454-456): pose unchanged */
  double* mm = (double*)malloc(sizeof(double) * (size_t)n * d);
  double* mc = (double*)malloc(sizeof(double) * (size_t)n * d * d);
  double* fm = (double*)malloc(sizeof(double) * (size_t)n * d);
  double* fc = (double*)malloc(sizeof(double) * (size_t)n * d * d);
  int r = 0;
  for (int i = 0; i < moving->n_cells; ++i)
    for (int j = 0; j < k; ++j) {
      int32_t ci = corr[(size_t)i * k + j];
      if (ci < 0) continue;
      for (int e = 0; e < d; ++e) {
        mm[(size_t)r * d + e] = (double)moving->cells[i].mean[e];
        fm[(size_t)r * d + e] = (double)fixed->cells[ci].mean[e];
      }
      cell_cov_full(&moving->cells[i], d, mc + (size_t)r * d * d);
      cell_cov_full(&fixed->cells[ci], d, fc + (size_t)r * d * d);
      ++r;
    }
  pair_user u;
  u.d = d;
  u.parameterization = p->parameterization;
  u.n = n;
  u.mm = mm; u.mc = mc; u.fm = fm; u.fc = fc;
  u.loss_a = p->loss_scale;
  u.loss_alpha = p->loss_alpha;
  u.loss_w = p->loss_weight;
  u.loss_mu = 1.0;
  u.apply_loss = 0;

  orc_problem P;
  P.user = &u;
  P.eval = pair_eval;
  P.plus = pair_plus;
  P.n_res = n;
  const int vec_like = p->parameterization == ORC_PARAM_VECTOR || p->parameterization == ORC_PARAM_ANALYTIC; /* (pos, rot) blocks */
  P.n_ambient = vec_like ? 3 : 4;
  P.n_tangent = (p->parameterization == ORC_PARAM_AMBIENT4) ? 4 : 3;

  double x[4];
  if (vec_like) {
    double xi[3];
    orc_se2_log(pose4, xi); /* two_representation_state.rot = trans.log()(2) (ndt_matcher.cpp:438-439) */
    x[0] = pose4[2]; x[1] = pose4[3]; x[2] = xi[2];
  } else {
    memcpy(x, pose4, sizeof(double) * 4);
  }

  /* raw residuals at the initial point (apply_loss_function = false) */
  double* raw = (double*)malloc(sizeof(double) * (size_t)n);
  double c0;
  int ok = pair_eval(&u, x, &c0, raw, NULL);
  double max_res = raw[0];
  for (int i = 1; i < n; ++i) if (raw[i] > max_res) max_res = raw[i];
  free(raw);
  double gnc_mu = 2.0 * pow(max_res, 2) / pow(p->mu_scale, 2);
  gnc_mu = fmin(gnc_mu, pow(p->gnc_divisor, (double)(p->gnc_steps - 1)));
  if (st) {
    st->max_raw_residual = max_res;
    st->mu0 = gnc_mu;
  }
  int term = ORC_TERM_FAILURE;
  double final_cost = 0;
  u.apply_loss = 1;
  if (ok) {
    do {
      gnc_mu = fmax(gnc_mu, 1.0);
      u.loss_mu = gnc_mu;
      term = lm_minimize(&P, p, x, st, &final_cost);
      if (st) st->n_solves++;
      gnc_mu /= p->gnc_divisor;
    } while (gnc_mu > 1.0 / sqrt(p->gnc_divisor));
  }
  if (st) {
    st->termination = term;
    st->final_cost = final_cost;
  }
  if (vec_like) {
    /* Sophus::SE2d(rot, pos) (ndt_matcher.cpp:486) */
    double c = cos(x[2]), s = sin(x[2]);
    so2_normalize(&c, &s);
    pose4[0] = c; pose4[1] = s; pose4[2] = x[0]; pose4[3] = x[1];
  } else {
    memcpy(pose4, x, sizeof(double) * 4);
  }
  free(mm); free(mc); free(fm); free(fc);
  return ok ? 0 : 2;
}

int orc_register_pair(const orc_map* fixed, const orc_map* moving, const orc_matcher_params* p,
                      double pose4[4], double* cost_out, orc_solve_stats* st) {
  const int k = p->n_neighbours;
  int32_t* corr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(moving->n_cells > 0 ? moving->n_cells : 1) * k);
  orc_solve_stats local;
  orc_solve_stats* s = st ? st : &local;
  orc_associate(fixed, moving, pose4, k, p->lookup_mahalanobis, p->use_intensity, corr);
  int rc = orc_solve_pair(fixed, moving, corr, k, p, pose4, s);
  if (cost_out) *cost_out = s->n_residuals > 0 ? s->final_cost / (double)s->n_residuals : 0.0;
  free(corr);
  return rc;
}

int orc_register_batch(int B, const float* pts, int n, int stride, int ioff, int n_clusters,
                       float max_range, orc_map* const* fixed_maps, const int32_t* fixed_idx,
                       const orc_matcher_params* p, const double* guess4, double* pose4_out,
                       double* cost_out, int32_t* iters_out, int n_threads, int32_t* stats_out) {
  int fail = 0;
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : fail)
#else
  (void)n_threads;
#endif
  for (int b = 0; b < B; ++b) {
    const orc_map* fx = fixed_maps[fixed_idx[b]];
    orc_map* scan = orc_map_create(fx->size_x, fx->size_y, fx->res, 0.0, 0.0, fx->max_neighbour_dist,
                                   fx->min_points, n / (fx->min_points + 1 > 0 ? fx->min_points + 1 : 1) + 8);
    orc_solve_stats* st = (orc_solve_stats*)malloc(sizeof(orc_solve_stats));
    orc_ndt_build(scan, pts + (size_t)b * n * stride, n, stride, ioff, n_clusters, max_range);
    double p4[4];
    memcpy(p4, guess4 + (size_t)b * 4, sizeof(p4));
    double cost = 0;
    int rc = orc_register_pair(fx, scan, p, p4, &cost, st);
    if (rc) fail++;
    memcpy(pose4_out + (size_t)b * 4, p4, sizeof(p4));
    if (cost_out) cost_out[b] = cost;
    if (iters_out) iters_out[b] = st->n_iterations;
    if (stats_out) { /* [B][4]: residual blocks, GNC solves, termination of the last solve, passes (cost + Jacobian evaluations) */
      stats_out[4 * b + 0] = st->n_residuals;
      stats_out[4 * b + 1] = st->n_solves;
      stats_out[4 * b + 2] = st->termination;
      stats_out[4 * b + 3] = st->n_jac_evals + st->n_cost_evals;
    }
    free(st);
    orc_map_destroy(scan);
  }
  return fail;
}

/* ============================================================ fixed-lag window (a16/a17) == */

/* predictSE2 (ceres_residuals.h:62-83): dt clamped to >= 0.2; screw uses 0.5*dt*a (sic). */
static void predict_se2(const double pose[4], const double v[2], double w, const double a[2], double raw_dt,
                        double pose_out[4], double v_out[2], double* w_out, double a_out[2]) {
  const double dt = raw_dt > 0.2 ? raw_dt : 0.2;
  double screw[3] = {v[0] * dt + 0.5 * dt * a[0], v[1] * dt + 0.5 * dt * a[1], w * dt};
  double e[4];
  orc_se2_exp(screw, e);
  orc_se2_mul(pose, e, pose_out);
  v_out[0] = v[0] + dt * a[0];
  v_out[1] = v[1] + dt * a[1];
  *w_out = w;
  a_out[0] = a[0];
  a_out[1] = a[1];
}

/* Matcher::predictTransform (ndt_matcher.cpp:22-59), optimize_on_manifold branch */
void orc_predict_state(const orc_state* last, double stamp, orc_state* next) {
  const double zero[2] = {0.0, 0.0}; /* last_state.lin_acc = Zero (:26) */
  memset(next, 0, sizeof(*next));
  predict_se2(last->pose, last->lin_vel, last->rot_vel, zero, stamp - last->stamp, next->pose, next->lin_vel,
              &next->rot_vel, next->lin_acc);
  next->pos[0] = next->pose[2];
  next->pos[1] = next->pose[3];
  next->rot = atan2(next->pose[1], next->pose[0]); /* pose.log()(2) */
  next->imu_bias = 0.0;                            /* X_next_.imu_bias stays at Matcher::initialize's 0 (:15) */
  next->stamp = stamp;
}

/* V(w) = [[a,-b],[b,a]], a = sin w / w, b = (1-cos w)/w and derivatives; Sophus' small-angle branch */
static void se2_V(double w, double* a, double* b, double* da, double* db) {
  if (fabs(w) < 1e-10) {
    *a = 1.0 - w * w / 6.0;
    *b = 0.5 * w - w * w * w / 24.0;
    *da = -w / 3.0;
    *db = 0.5 - w * w / 8.0;
  } else {
    const double s = sin(w), c = cos(w);
    *a = s / w;
    *b = (1.0 - c) / w;
    *da = (w * c - s) / (w * w);
    *db = (w * s - (1.0 - c)) / (w * w);
  }
}

/* MotionModelFactorSE2 (ceres_residuals.h:621-679) with the Jacobian Ceres autodiff x
 * Sophus::Manifold<SE2>::PlusJacobian yields (right perturbations X <- X exp(delta)). */
void orc_motion_residual(const orc_state* x0, const orc_state* x1, const double* sqrtI, double* r8, double* J) {
  const double raw_dt = x1->stamp - x0->stamp;
  const double dt = raw_dt > 0.2 ? raw_dt : 0.2;
  double pred[4], vp[2], wp_, ap[2];
  predict_se2(x0->pose, x0->lin_vel, x0->rot_vel, x0->lin_acc, raw_dt, pred, vp, &wp_, ap);
  double pinv[4], E[4], lg[3];
  orc_se2_inv(pred, pinv);
  orc_se2_mul(pinv, x1->pose, E);
  orc_se2_log(E, lg);
  double r[8] = {lg[0], lg[1], lg[2], x1->lin_vel[0] - vp[0], x1->lin_vel[1] - vp[1], x1->rot_vel - wp_,
                 x1->lin_acc[0] - ap[0], x1->lin_acc[1] - ap[1]};
  double Jr[8][16];
  memset(Jr, 0, sizeof(Jr));
  if (J) {
    /* E = (R_E, t_E), phi = angle(E); r_xy = Vinv(phi) t_E, r_phi = phi */
    const double phi = lg[2];
    const double cE = E[0], sE = E[1], tEx = E[2], tEy = E[3];
    double h, dh; /* Vinv = [[h, phi/2], [-phi/2, h]] */
    if (fabs(E[0] - 1.0) < 1e-10) {
      h = 1.0 - phi * phi / 12.0;
      dh = -phi / 6.0;
    } else {
      const double half = 0.5 * phi, sh = sin(half), ch = cos(half);
      h = half * ch / sh;
      dh = 0.5 * ch / sh - 0.5 * half / (sh * sh);
    }
    const double Vi[2][2] = {{h, 0.5 * phi}, {-0.5 * phi, h}};
    /* d(Vinv)/dphi * t_E */
    const double dVt[2] = {dh * tEx + 0.5 * tEy, -0.5 * tEx + dh * tEy};
    /* ---- X1 = X1 exp(d1): dphi = dw1, dt_E = R_E dv1 */
    {
      const double RE[2][2] = {{cE, -sE}, {sE, cE}};
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) Jr[i][8 + j] = Vi[i][0] * RE[0][j] + Vi[i][1] * RE[1][j];
      Jr[0][8 + 2] = dVt[0];
      Jr[1][8 + 2] = dVt[1];
      Jr[2][8 + 2] = 1.0;
    }
    /* ---- pred = X0 exp(xi): dtheta_p and dt_p in terms of (dv0', dw0') of X0 and dxi */
    const double xi[3] = {x0->lin_vel[0] * dt + 0.5 * dt * x0->lin_acc[0], x0->lin_vel[1] * dt + 0.5 * dt * x0->lin_acc[1],
                          x0->rot_vel * dt};
    double a, b, da, db;
    se2_V(xi[2], &a, &b, &da, &db);
    const double c0 = x0->pose[0], s0 = x0->pose[1];
    const double Vxi[2] = {a * xi[0] - b * xi[1], b * xi[0] + a * xi[1]};   /* V(xi_w) xi_v */
    const double dVxi[2] = {da * xi[0] - db * xi[1], db * xi[0] + da * xi[1]}; /* V'(xi_w) xi_v */
    /* columns of dt_p (world frame) and dtheta_p for the 6 generators: X0 tangent (vx, vy, w) and xi (vx, vy, w) */
    double dtp[6][2], dth[6];
    /* X0 vx, vy: dt_p = R0 e_k */
    dtp[0][0] = c0;  dtp[0][1] = s0;  dth[0] = 0;
    dtp[1][0] = -s0; dtp[1][1] = c0;  dth[1] = 0;
    /* X0 w: dt_p = R0 J V xi_v, dtheta_p = 1 */
    dtp[2][0] = c0 * (-Vxi[1]) - s0 * Vxi[0];
    dtp[2][1] = s0 * (-Vxi[1]) + c0 * Vxi[0];
    dth[2] = 1;
    /* xi vx, vy: dt_p = R0 V e_k */
    dtp[3][0] = c0 * a - s0 * b;     dtp[3][1] = s0 * a + c0 * b;  dth[3] = 0;
    dtp[4][0] = c0 * (-b) - s0 * a;  dtp[4][1] = s0 * (-b) + c0 * a; dth[4] = 0;
    /* xi w: dt_p = R0 V' xi_v, dtheta_p = 1 */
    dtp[5][0] = c0 * dVxi[0] - s0 * dVxi[1];
    dtp[5][1] = s0 * dVxi[0] + c0 * dVxi[1];
    dth[5] = 1;
    /* dphi = -dtheta_p ; dt_E = -R_p^T dt_p - J t_E dtheta_p ; dr_xy = Vinv dt_E + dVt dphi */
    const double cp = pred[0], sp = pred[1];
    double G[6][3];
    for (int g = 0; g < 6; ++g) {
      const double ex = -(cp * dtp[g][0] + sp * dtp[g][1]) - (-tEy) * dth[g];
      const double ey = -(-sp * dtp[g][0] + cp * dtp[g][1]) - (tEx) * dth[g];
      const double dphi = -dth[g];
      G[g][0] = Vi[0][0] * ex + Vi[0][1] * ey + dVt[0] * dphi;
      G[g][1] = Vi[1][0] * ex + Vi[1][1] * ey + dVt[1] * dphi;
      G[g][2] = dphi;
    }
    for (int i = 0; i < 3; ++i) {
      Jr[i][0] = G[0][i];              /* X0 vx */
      Jr[i][1] = G[1][i];              /* X0 vy */
      Jr[i][2] = G[2][i];              /* X0 w  */
      Jr[i][3] = G[3][i] * dt;         /* v0x: xi_vx = dt v0x + .. */
      Jr[i][4] = G[4][i] * dt;         /* v0y */
      Jr[i][5] = G[5][i] * dt;         /* w0:  xi_w = dt w0 */
      Jr[i][6] = G[3][i] * 0.5 * dt;   /* a0x */
      Jr[i][7] = G[4][i] * 0.5 * dt;   /* a0y */
    }
    /* rows 3..7 */
    Jr[3][3] = -1; Jr[3][6] = -dt; Jr[3][8 + 3] = 1;
    Jr[4][4] = -1; Jr[4][7] = -dt; Jr[4][8 + 4] = 1;
    Jr[5][5] = -1; Jr[5][8 + 5] = 1;
    Jr[6][6] = -1; Jr[6][8 + 6] = 1;
    Jr[7][7] = -1; Jr[7][8 + 7] = 1;
  }
  /* residuals_map.applyOnTheLeft(sqrtI_) */
  for (int i = 0; i < 8; ++i) {
    double acc = 0;
    for (int k = 0; k < 8; ++k) acc += sqrtI[i * 8 + k] * r[k];
    r8[i] = acc;
  }
  if (J)
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 16; ++j) {
        double acc = 0;
        for (int k = 0; k < 8; ++k) acc += sqrtI[i * 8 + k] * Jr[k][j];
        J[i * 16 + j] = acc;
      }
}

/* RotationalResidualSE2 (ceres_residuals.h:338-370) */
void orc_imu_residual(const orc_state* x0, const orc_state* x1, double imu_rot, double weight, double bias_weight,
                      double* r2, double* J) {
  const double raw_dt = x1->stamp - x0->stamp; /* dt_ is NOT clamped here (ndt_matcher.cpp:147) */
  double screw[3] = {0.0, 0.0, x1->imu_bias * raw_dt};
  double e[4], M1[4], inv0[4], E[4], lg[3];
  orc_se2_exp(screw, e);
  orc_se2_mul(x1->pose, e, M1);
  orc_se2_inv(x0->pose, inv0);
  orc_se2_mul(inv0, M1, E);
  orc_se2_log(E, lg);
  r2[0] = weight * (imu_rot - lg[2]);
  r2[1] = bias_weight * (x1->imu_bias - x0->imu_bias);
  if (J) {
    memset(J, 0, sizeof(double) * 16);
    J[0 * 8 + 2] = weight;           /* d/dw0: phi = th1 + b1 dt - th0 */
    J[0 * 8 + 3 + 2] = -weight;      /* d/dw1 */
    J[0 * 8 + 7] = -weight * raw_dt; /* d/db1 */
    J[1 * 8 + 6] = -bias_weight;     /* d/db0 */
    J[1 * 8 + 7] = bias_weight;      /* d/db1 */
  }
}

/* NormalizeAngle (include/ndt_registration/state_manifold.h:17-23): into [-pi, pi) */
static double normalize_angle(double a) { return a - 2.0 * M_PI * floor((a + M_PI) / (2.0 * M_PI)); }

/* predict, (pos, rot) form (ceres_residuals.h:25-55 template / :91-123 double): mid-point heading, dt clamped to >= 0.2 */
static void predict_vec(const double pos[2], double rot, const double v[2], double w, const double a[2], double raw_dt,
                        double pos_out[2], double* rot_out, double v_out[2], double* w_out, double a_out[2], double* cy_out,
                        double* sy_out, double delta_rot[2]) {
  const double dt = raw_dt > 0.2 ? raw_dt : 0.2;
  double new_rot = rot;
  const double mid = normalize_angle(rot + 0.5 * dt * w);
  new_rot += dt * w;
  new_rot = normalize_angle(new_rot);
  const double sy = sin(mid), cy = cos(mid);
  const double delta_x = v[0] * dt + 0.5 * a[0] * dt * dt;
  const double delta_y = v[1] * dt + 0.5 * a[1] * dt * dt;
  const double dxr = cy * delta_x - sy * delta_y, dyr = sy * delta_x + cy * delta_y;
  pos_out[0] = pos[0] + dxr;
  pos_out[1] = pos[1] + dyr;
  *rot_out = new_rot;
  v_out[0] = v[0] + dt * a[0];
  v_out[1] = v[1] + dt * a[1];
  *w_out = w;
  a_out[0] = a[0];
  a_out[1] = a[1];
  if (cy_out) *cy_out = cy;
  if (sy_out) *sy_out = sy;
  if (delta_rot) {
    delta_rot[0] = dxr;
    delta_rot[1] = dyr;
  }
}

/* Matcher::predictTransform, the branch taken when optimize_on_manifold is false (ndt_matcher.cpp:27-41) */
void orc_predict_state_vec(const orc_state* last, double stamp, orc_state* next) {
  const double zero[2] = {0.0, 0.0}; /* last_state.lin_acc = Zero (:26) */
  memset(next, 0, sizeof(*next));
  predict_vec(last->pos, last->rot, last->lin_vel, last->rot_vel, zero, stamp - last->stamp, next->pos, &next->rot, next->lin_vel,
              &next->rot_vel, next->lin_acc, NULL, NULL, NULL);
  next->pose[0] = cos(next->rot); /* Sophus::SE2d(rot, pos) (:41) */
  next->pose[1] = sin(next->rot);
  next->pose[2] = next->pos[0];
  next->pose[3] = next->pos[1];
  next->imu_bias = 0.0;
  next->stamp = stamp;
}

/* MotionModelFactor (ceres_residuals.h:554-619), parameter blocks pos(2) rot(1) lin_vel(2) rot_vel(1) lin_acc(2) of both
 * states; J: 8 x 16, columns [X0: pos2 rot1 v2 w1 a2 | X1: the same] -- what Ceres' autodiff yields (NormalizeAngle has
 * derivative 1: the floor() term is piecewise constant). */
void orc_motion_residual_vec(const orc_state* x0, const orc_state* x1, const double* sqrtI, double* r8, double* J) {
  const double raw_dt = x1->stamp - x0->stamp;
  const double dt = raw_dt > 0.2 ? raw_dt : 0.2;
  double pp[2], rp, vp[2], wp_, ap[2], cy, sy, dr[2];
  predict_vec(x0->pos, x0->rot, x0->lin_vel, x0->rot_vel, x0->lin_acc, raw_dt, pp, &rp, vp, &wp_, ap, &cy, &sy, dr);
  const double r[8] = {x1->pos[0] - pp[0], x1->pos[1] - pp[1], normalize_angle(x1->rot - rp), x1->lin_vel[0] - vp[0],
                       x1->lin_vel[1] - vp[1], x1->rot_vel - wp_, x1->lin_acc[0] - ap[0], x1->lin_acc[1] - ap[1]};
  double Jr[8][16];
  memset(Jr, 0, sizeof(Jr));
  if (J) {
    const double h = 0.5 * dt * dt;
    /* position rows: -d pos_pred */
    Jr[0][0] = -1; Jr[1][1] = -1;
    Jr[0][2] = dr[1];             Jr[1][2] = -dr[0];              /* d/d rot0: -(d R / d theta) delta = -(-dyr, dxr) */
    Jr[0][3] = -cy * dt;          Jr[0][4] = sy * dt;
    Jr[1][3] = -sy * dt;          Jr[1][4] = -cy * dt;
    Jr[0][5] = dr[1] * 0.5 * dt;  Jr[1][5] = -dr[0] * 0.5 * dt;   /* heading at rot0 + dt w0 / 2 */
    Jr[0][6] = -cy * h;           Jr[0][7] = sy * h;
    Jr[1][6] = -sy * h;           Jr[1][7] = -cy * h;
    Jr[0][8] = 1; Jr[1][9] = 1;
    /* rotation row */
    Jr[2][2] = -1; Jr[2][5] = -dt; Jr[2][10] = 1;
    /* velocity / acceleration rows */
    Jr[3][3] = -1; Jr[3][6] = -dt; Jr[3][11] = 1;
    Jr[4][4] = -1; Jr[4][7] = -dt; Jr[4][12] = 1;
    Jr[5][5] = -1; Jr[5][13] = 1;
    Jr[6][6] = -1; Jr[6][14] = 1;
    Jr[7][7] = -1; Jr[7][15] = 1;
  }
  for (int i = 0; i < 8; ++i) {
    double acc = 0;
    for (int k = 0; k < 8; ++k) acc += sqrtI[i * 8 + k] * r[k];
    r8[i] = acc;
  }
  if (J)
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 16; ++j) {
        double acc = 0;
        for (int k = 0; k < 8; ++k) acc += sqrtI[i * 8 + k] * Jr[k][j];
        J[i * 16 + j] = acc;
      }
}

/* RotationalResidual (ceres_residuals.h:307-336): blocks rot0, rot1, bias0, bias1; J: 2 x 8, columns as orc_imu_residual
 * ([X0 pos2 rot1 | X1 pos2 rot1 | b0 | b1]). */
void orc_imu_residual_vec(const orc_state* x0, const orc_state* x1, double imu_rot, double weight, double bias_weight, double* r2,
                          double* J) {
  const double raw_dt = x1->stamp - x0->stamp; /* dt_ is NOT clamped here (ndt_matcher.cpp:147) */
  r2[0] = weight * (imu_rot - normalize_angle(x1->rot - x0->rot + x1->imu_bias * raw_dt));
  r2[1] = bias_weight * (x1->imu_bias - x0->imu_bias);
  if (J) {
    memset(J, 0, sizeof(double) * 16);
    J[0 * 8 + 2] = weight;
    J[0 * 8 + 3 + 2] = -weight;
    J[0 * 8 + 7] = -weight * raw_dt;
    J[1 * 8 + 6] = -bias_weight;
    J[1 * 8 + 7] = bias_weight;
  }
}

typedef struct win_user {
  int vec;                      /* (pos[2], rot) parameter blocks instead of the SE(2) manifold (optimize_on_manifold: false) */
  int analytic;                 /* ... with the reference's hand-written NDT Jacobian (use_analytic_expressions_for_optimization) */
  int S, F, d, k;               /* states 0..S (0 = oldest, pose constant), fixed maps */
  int const_vel, use_imu;
  orc_state base[16];           /* constant parts (stamps, oldest pose, constant blocks) */
  const double* imu;
  const double* sqrtI;
  double weight_imu, weight_imu_bias;
  /* NDT residuals, grouped by state j = 1..S */
  int n_ndt;
  int* ndt_state;
  double *mm, *mc, *fm, *fc;
  int apply_loss;
  double loss_a, loss_alpha, loss_mu, loss_w;
  /* layout */
  int off_amb[16][5], off_tan[16][5]; /* block offsets per state: pose, v, w, a, b ; -1 = constant */
  int n_amb, n_tan, n_res;
} win_user;

static void win_layout(win_user* u) {
  int a = 0, t = 0;
  for (int j = 0; j <= u->S; ++j) {
    /* AddParameterBlock order: pose, lin_vel, rot_vel, lin_acc, [imu_bias] (ndt_matcher.cpp:290-320,146-181) */
    if (j == 0) { u->off_amb[j][0] = u->off_tan[j][0] = -1; } else { u->off_amb[j][0] = a; u->off_tan[j][0] = t; a += u->vec ? 3 : 4; t += 3; }
    u->off_amb[j][1] = a; u->off_tan[j][1] = t; a += 2; t += 2;
    u->off_amb[j][2] = a; u->off_tan[j][2] = t; a += 1; t += 1;
    if (u->const_vel) { u->off_amb[j][3] = u->off_tan[j][3] = -1; } else { u->off_amb[j][3] = a; u->off_tan[j][3] = t; a += 2; t += 2; }
    if (u->use_imu && j > 0) { u->off_amb[j][4] = a; u->off_tan[j][4] = t; a += 1; t += 1; } else { u->off_amb[j][4] = u->off_tan[j][4] = -1; }
  }
  u->n_amb = a;
  u->n_tan = t;
}

static void win_unpack(const win_user* u, const double* x, orc_state* st) {
  for (int j = 0; j <= u->S; ++j) {
    st[j] = u->base[j];
    if (u->off_amb[j][0] >= 0) {
      if (u->vec) {
        st[j].pos[0] = x[u->off_amb[j][0]];
        st[j].pos[1] = x[u->off_amb[j][0] + 1];
        st[j].rot = x[u->off_amb[j][0] + 2];
      } else {
        memcpy(st[j].pose, x + u->off_amb[j][0], sizeof(double) * 4);
      }
    }
    memcpy(st[j].lin_vel, x + u->off_amb[j][1], sizeof(double) * 2);
    st[j].rot_vel = x[u->off_amb[j][2]];
    if (u->off_amb[j][3] >= 0) memcpy(st[j].lin_acc, x + u->off_amb[j][3], sizeof(double) * 2);
    if (u->off_amb[j][4] >= 0) st[j].imu_bias = x[u->off_amb[j][4]];
  }
}

static void win_pack(const win_user* u, const orc_state* st, double* x) {
  for (int j = 0; j <= u->S; ++j) {
    if (u->off_amb[j][0] >= 0) {
      if (u->vec) {
        x[u->off_amb[j][0]] = st[j].pos[0];
        x[u->off_amb[j][0] + 1] = st[j].pos[1];
        x[u->off_amb[j][0] + 2] = st[j].rot;
      } else {
        memcpy(x + u->off_amb[j][0], st[j].pose, sizeof(double) * 4);
      }
    }
    memcpy(x + u->off_amb[j][1], st[j].lin_vel, sizeof(double) * 2);
    x[u->off_amb[j][2]] = st[j].rot_vel;
    if (u->off_amb[j][3] >= 0) memcpy(x + u->off_amb[j][3], st[j].lin_acc, sizeof(double) * 2);
    if (u->off_amb[j][4] >= 0) x[u->off_amb[j][4]] = st[j].imu_bias;
  }
}

static void win_plus(void* user, const double* x, const double* delta, double* xp) {
  const win_user* u = (const win_user*)user;
  for (int j = 0; j <= u->S; ++j) {
    if (u->off_amb[j][0] >= 0 && u->vec) {
      for (int e = 0; e < 3; ++e) xp[u->off_amb[j][0] + e] = x[u->off_amb[j][0] + e] + delta[u->off_tan[j][0] + e];
    } else if (u->off_amb[j][0] >= 0) {
      double e[4];
      orc_se2_exp(delta + u->off_tan[j][0], e);
      orc_se2_mul(x + u->off_amb[j][0], e, xp + u->off_amb[j][0]);
    }
    static const int sz[5] = {0, 2, 1, 2, 1};
    for (int b = 1; b < 5; ++b)
      if (u->off_amb[j][b] >= 0)
        for (int e = 0; e < sz[b]; ++e) xp[u->off_amb[j][b] + e] = x[u->off_amb[j][b] + e] + delta[u->off_tan[j][b] + e];
  }
}

/* scatter a factor's local Jacobian columns (blocks of state j: pose3 v2 w1 a2 [b1]) into the dense row */
static void win_scatter(const win_user* u, double* row, int j, int blk, const double* vals, int n) {
  const int off = u->off_tan[j][blk];
  if (off < 0) return;
  for (int e = 0; e < n; ++e) row[off + e] += vals[e];
}

/* Residual order = AddResidualBlock order (ndt_matcher.cpp:355-368): for i = S..1 (state j = S+1-i ... ),
 * i.e. oldest pair first: motion(j-1,j), [imu(j-1,j)], NDT(j, f = 0..F-1). */
static int win_eval(void* user, const double* x, double* cost, double* residuals, double* jac) {
  win_user* u = (win_user*)user;
  orc_state st[16];
  win_unpack(u, x, st);
  const int nt = u->n_tan;
  double total = 0;
  int row = 0, ndt_at = 0;
  if (jac) memset(jac, 0, sizeof(double) * (size_t)u->n_res * nt);
  for (int j = 1; j <= u->S; ++j) {
    double r8[8], J[8 * 16];
    if (u->vec) orc_motion_residual_vec(&st[j - 1], &st[j], u->sqrtI, r8, jac ? J : NULL);
    else orc_motion_residual(&st[j - 1], &st[j], u->sqrtI, r8, jac ? J : NULL);
    for (int i = 0; i < 8; ++i) {
      total += 0.5 * r8[i] * r8[i];
      if (residuals) residuals[row + i] = r8[i];
      if (jac) {
        double* rw = jac + (size_t)(row + i) * nt;
        win_scatter(u, rw, j - 1, 0, J + i * 16 + 0, 3);
        win_scatter(u, rw, j - 1, 1, J + i * 16 + 3, 2);
        win_scatter(u, rw, j - 1, 2, J + i * 16 + 5, 1);
        win_scatter(u, rw, j - 1, 3, J + i * 16 + 6, 2);
        win_scatter(u, rw, j, 0, J + i * 16 + 8, 3);
        win_scatter(u, rw, j, 1, J + i * 16 + 11, 2);
        win_scatter(u, rw, j, 2, J + i * 16 + 13, 1);
        win_scatter(u, rw, j, 3, J + i * 16 + 14, 2);
      }
    }
    row += 8;
    if (u->use_imu) {
      double r2[2], J2[16];
      if (u->vec) orc_imu_residual_vec(&st[j - 1], &st[j], u->imu[j - 1], u->weight_imu, u->weight_imu_bias, r2, jac ? J2 : NULL);
      else orc_imu_residual(&st[j - 1], &st[j], u->imu[j - 1], u->weight_imu, u->weight_imu_bias, r2, jac ? J2 : NULL);
      for (int i = 0; i < 2; ++i) {
        total += 0.5 * r2[i] * r2[i];
        if (residuals) residuals[row + i] = r2[i];
        if (jac) {
          double* rw = jac + (size_t)(row + i) * nt;
          win_scatter(u, rw, j - 1, 0, J2 + i * 8 + 0, 3);
          win_scatter(u, rw, j, 0, J2 + i * 8 + 3, 3);
          win_scatter(u, rw, j - 1, 4, J2 + i * 8 + 6, 1);
          win_scatter(u, rw, j, 4, J2 + i * 8 + 7, 1);
        }
      }
      row += 2;
    }
    int ndt_end = ndt_at;
    while (ndt_end < u->n_ndt && u->ndt_state[ndt_end] == j) ++ndt_end;
    /* vector form: NDTFrameToMap{,Intensity}FactorResidual on (pos, rot) (ceres_residuals.h:421-451, 486-518) */
    const double pv[4] = {cos(st[j].rot), sin(st[j].rot), st[j].pos[0], st[j].pos[1]};
    int bad = 0;
    /* one NDT residual block of state j: the statement of the sequential loop and of the timing-only OpenMP loop below */
#define WIN_NDT_BLOCK(TOTAL, BAD)                                                                                          \
  do {                                                                                                                     \
        const int d = u->d;                                                                                                \
        const int rw_i = row + (at - ndt_at);                                                                              \
        double jl[4];                                                                                                      \
        double r = orc_ndt_residual(d, u->vec ? (u->analytic ? ORC_PARAM_ANALYTIC : ORC_PARAM_VECTOR) : ORC_PARAM_MANIFOLD, u->vec ? pv : st[j].pose, u->mm + (size_t)at * d,\
                                    u->mc + (size_t)at * d * d, u->fm + (size_t)at * d, u->fc + (size_t)at * d * d, jac ? jl : NULL);\
        if (!isfinite(r)) {                                                                                                \
          (BAD) |= 1;                                                                                                      \
          break;                                                                                                           \
        }                                                                                                                  \
        const double sq = r * r;                                                                                           \
        double rs = 1.0, js = 1.0;                                                                                         \
        if (u->apply_loss) {                                                                                               \
          double rho[3];                                                                                                   \
          orc_barron_scaled(sq, u->loss_a, u->loss_alpha, u->loss_mu, u->loss_w, rho);                                     \
          (TOTAL) += 0.5 * rho[0];                                                                                         \
          const double sqrt_rho1 = sqrt(rho[1]);                                                                           \
          if (sq == 0.0 || rho[2] <= 0.0) {                                                                                \
            rs = js = sqrt_rho1;                                                                                           \
          } else {                                                                                                         \
            const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];                                                             \
            const double alpha = 1.0 - sqrt(D);                                                                            \
            rs = sqrt_rho1 / (1 - alpha);                                                                                  \
            js = sqrt_rho1 * (1.0 - alpha);                                                                                \
          }                                                                                                                \
        } else {                                                                                                           \
          (TOTAL) += 0.5 * sq;                                                                                             \
        }                                                                                                                  \
        if (residuals) residuals[rw_i] = rs * r;                                                                           \
        if (jac) {                                                                                                         \
          double* rw = jac + (size_t)rw_i * nt;                                                                            \
          for (int e = 0; e < 3; ++e) rw[u->off_tan[j][0] + e] = js * jl[e];                                               \
        }                                                                                                                  \
  } while (0)
#ifdef _OPENMP
    if (g_eval_threads > 1) { /* timing only (orc_set_eval_threads): another summation order */
#pragma omp parallel for schedule(static) num_threads(g_eval_threads) reduction(+ : total) reduction(| : bad)
      for (int at = ndt_at; at < ndt_end; ++at) WIN_NDT_BLOCK(total, bad);
    } else
#endif
    {
      /* default: strictly sequential sums, ((total + t1) + t2) + ... -- an `if (0)` OpenMP reduction would still add a private
         partial sum to total at the end, i.e. total + (t1 + t2 + ...) */
      for (int at = ndt_at; at < ndt_end; ++at) WIN_NDT_BLOCK(total, bad);
    }
#undef WIN_NDT_BLOCK
    if (bad) return 0;
    row += ndt_end - ndt_at;
    ndt_at = ndt_end;
  }
  *cost = total;
  return 1;
}

int orc_register_window(orc_map* const* fixed, int n_fixed, orc_map* const* moving, orc_state* states, int n_states,
                        const double* imu, const orc_matcher_params* p, const orc_window_params* wp, double trans4[4],
                        orc_solve_stats* st) {
  const int S = n_states - 1;
  if (S < 1 || S > 15 || n_fixed < 1) return -1; /* (the arrays above; the reference takes any lag, ndt_matcher.cpp:343) */
  orc_solve_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof(*st));
  const int k = p->n_neighbours, d = p->use_intensity ? 3 : 2;
  /* prior for the rejection gate (ndt_matcher.cpp:339-340) */
  const double prior_t[2] = {trans4[2], trans4[3]};
  const double prior_rot = atan2(trans4[1], trans4[0]);

  win_user u;
  memset(&u, 0, sizeof(u));
  if (p->parameterization != ORC_PARAM_MANIFOLD && p->parameterization != ORC_PARAM_VECTOR && p->parameterization != ORC_PARAM_ANALYTIC) return -1;
  u.vec = p->parameterization != ORC_PARAM_MANIFOLD;
  /* use_analytic_expressions_for_optimization: MotionModelFactorAnalytic / RotationalResidualAnalytic (ceres_residuals.h:794-889,
   * 372-419) are the vector factors with hand-written -- correct -- Jacobians; only the NDT functor's rotation column differs */
  u.analytic = p->parameterization == ORC_PARAM_ANALYTIC;
  u.S = S; u.F = n_fixed; u.d = d; u.k = k;
  u.const_vel = wp->use_constant_velocity_model;
  u.use_imu = wp->use_imu && imu;
  u.imu = imu;
  u.sqrtI = wp->motion_sqrtI;
  u.weight_imu = wp->weight_imu;
  u.weight_imu_bias = wp->weight_imu_bias;
  for (int j = 0; j <= S; ++j) u.base[j] = states[j];
  win_layout(&u);

  /* addNDTFactor per state and fixed map; association at the state's own pose (ndt_matcher.cpp:363-366) */
  int n_cells = 0, cap = 0;
  for (int j = 1; j <= S; ++j) {
    n_cells += moving[j - 1]->n_cells;
    cap += moving[j - 1]->n_cells * k * n_fixed;
  }
  u.ndt_state = (int*)malloc(sizeof(int) * (size_t)(cap > 0 ? cap : 1));
  u.mm = (double*)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1) * d);
  u.mc = (double*)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1) * d * d);
  u.fm = (double*)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1) * d);
  u.fc = (double*)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1) * d * d);
  int n = 0;
  for (int j = 1; j <= S; ++j) {
    const orc_map* mv = moving[j - 1];
    int32_t* corr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(mv->n_cells > 0 ? mv->n_cells : 1) * k);
    for (int f = 0; f < n_fixed; ++f) {
      orc_associate(fixed[f], mv, states[j].pose, k, p->lookup_mahalanobis, p->use_intensity, corr);
      for (int i = 0; i < mv->n_cells; ++i)
        for (int q = 0; q < k; ++q) {
          const int32_t ci = corr[(size_t)i * k + q];
          if (ci < 0) continue;
          u.ndt_state[n] = j;
          for (int e = 0; e < d; ++e) {
            u.mm[(size_t)n * d + e] = (double)mv->cells[i].mean[e];
            u.fm[(size_t)n * d + e] = (double)fixed[f]->cells[ci].mean[e];
          }
          cell_cov_full(&mv->cells[i], d, u.mc + (size_t)n * d * d);
          cell_cov_full(&fixed[f]->cells[ci], d, u.fc + (size_t)n * d * d);
          ++n;
        }
    }
    free(corr);
  }
  u.n_ndt = n;
  u.n_res = n + S * 8 + (u.use_imu ? S * 2 : 0);
  st->n_residuals = n;
  u.loss_a = p->loss_scale;
  u.loss_alpha = p->loss_alpha;
  u.loss_w = (n_cells > 0) ? wp->ndt_weight / (double)(n_cells * k) : 0.0; /* ndt_matcher.cpp:392 */
  u.loss_mu = 1.0;

  orc_problem P;
  P.user = &u;
  P.eval = win_eval;
  P.plus = win_plus;
  P.n_res = u.n_res;
  P.n_ambient = u.n_amb;
  P.n_tangent = u.n_tan;

  double x[8 * ORC_MAX_TANGENT]; /* ambient size: <= 5 + 10 S */
  win_pack(&u, states, x);

  /* raw NDT residuals -> gnc_mu (ndt_matcher.cpp:382-389) */
  double max_res = 0.0;
  int ok = 1;
  if (n > 0) {
    double* raw = (double*)malloc(sizeof(double) * (size_t)u.n_res);
    double c0;
    u.apply_loss = 0;
    ok = win_eval(&u, x, &c0, raw, NULL);
    /* NDT rows only */
    int row = 0, at = 0;
    max_res = -DBL_MAX;
    for (int j = 1; j <= S; ++j) {
      row += 8 + (u.use_imu ? 2 : 0);
      while (at < n && u.ndt_state[at] == j) {
        if (raw[row] > max_res) max_res = raw[row];
        ++row;
        ++at;
      }
    }
    free(raw);
  }
  double gnc_mu = 2.0 * pow(max_res, 2) / pow(p->mu_scale, 2);
  gnc_mu = fmin(gnc_mu, pow(p->gnc_divisor, (double)(p->gnc_steps - 1)));
  st->max_raw_residual = max_res;
  st->mu0 = gnc_mu;
  int term = ORC_TERM_FAILURE;
  double final_cost = 0;
  u.apply_loss = 1;
  if (ok) {
    do {
      gnc_mu = fmax(gnc_mu, 1.0);
      u.loss_mu = gnc_mu;
      term = lm_minimize(&P, p, x, st, &final_cost);
      st->n_solves++;
      gnc_mu /= p->gnc_divisor;
    } while (gnc_mu > 1.0 / sqrt(p->gnc_divisor));
  }
  st->termination = term;
  st->final_cost = final_cost;
  win_unpack(&u, x, states);
  /* both representations (ndt_matcher.cpp:399-406; local_fuser.cpp:141-150 does it for the whole window) */
  for (int j = 0; j <= S; ++j) {
    if (u.vec) { /* Sophus::SE2d(rot, pos) */
      states[j].pose[0] = cos(states[j].rot);
      states[j].pose[1] = sin(states[j].rot);
      states[j].pose[2] = states[j].pos[0];
      states[j].pose[3] = states[j].pos[1];
    } else {
      states[j].pos[0] = states[j].pose[2];
      states[j].pos[1] = states[j].pose[3];
      states[j].rot = atan2(states[j].pose[1], states[j].pose[0]);
    }
  }
  int rejected = 0;
  {
    /* rejection gate (ndt_matcher.cpp:411-422) */
    orc_state* X = &states[S];
    /* (pose.so2().inverse() * SO2(prior_rotation)).log() */
    const double pc = cos(prior_rot), ps = sin(prior_rot);
    const double re = X->pose[0] * pc + X->pose[1] * ps, im = X->pose[0] * ps - X->pose[1] * pc;
    const double dth = atan2(im, re);
    if (fabs(X->pose[2] - prior_t[0]) > wp->pose_reject_translation || fabs(X->pose[3] - prior_t[1]) > wp->pose_reject_translation ||
        fabs(dth) > wp->pose_reject_rotation) {
      rejected = 1;
      memcpy(X->pos, states[S - 1].pos, sizeof(X->pos));
      memcpy(X->pose, states[S - 1].pose, sizeof(X->pose));
      X->rot = states[S - 1].rot;
      X->lin_vel[0] = X->lin_vel[1] = 0.0;
      X->rot_vel = 0.0;
      X->lin_acc[0] = X->lin_acc[1] = 0.0;
      X->imu_bias = states[S - 1].imu_bias;
    }
  }
  memcpy(trans4, states[S].pose, sizeof(double) * 4);
  free(u.ndt_state); free(u.mm); free(u.mc); free(u.fm); free(u.fc);
  return ok ? rejected : -2;
}

/* ============================================================ f-1: filterScan ============= */

/* std::hypot(float, float): glibc evaluates hypotf in double and rounds once. */
static float hypot_f(float x, float y) { return (float)sqrt((double)x * (double)x + (double)y * (double)y); }

int orc_filter_scan(const float* raw, int n, int stride, int ioff, const orc_filter_params* p, float* out_pts,
                    float* out_polar, int capacity, float* peaks, int peak_cap, int* n_peaks) {
#define PX(i) raw[(size_t)(i) * stride + 0]
#define PY(i) raw[(size_t)(i) * stride + 1]
#define PZ(i) raw[(size_t)(i) * stride + 2]
#define PI_(i) raw[(size_t)(i) * stride + ioff]
  const float min_d = p->min_range, max_d = p->max_range, min_i = p->min_intensity; /* members are double in the
      reference but hold these float-representable config values; comparisons are float vs double there */
  float current_angle = 1000;
  float max_intensity = 0;
  size_t current_max_idx = 0;
  size_t* idzs = (size_t*)malloc(sizeof(size_t) * (size_t)(n > 0 ? n : 1));
  int n_idz = 0, np = 0;
  /* radar_preprocessor.cpp:56-75 */
  for (int i = 0; i < n; ++i) {
    const float dist = hypot_f(PX(i), PY(i));
    const float angle = atan2f(PY(i), PX(i));
    const float intensity = PI_(i);
    if (fabsf(angle - current_angle) > 0.0001) {
      if ((double)current_angle < 3 * M_PI) {
        if (n_idz == 0 || idzs[n_idz - 1] != current_max_idx) {
          if (peaks && np < peak_cap) {
            peaks[3 * np + 0] = current_angle;
            peaks[3 * np + 1] = hypot_f(PX(current_max_idx), PY(current_max_idx));
            peaks[3 * np + 2] = max_intensity;
          }
          ++np;
          idzs[n_idz++] = current_max_idx;
        }
        max_intensity = 0;
      }
      current_angle = angle;
    }
    if ((double)dist > (double)min_d && (double)dist < (double)max_d && intensity > max_intensity) {
      max_intensity = intensity;
      current_max_idx = (size_t)i;
    }
  }
  if (n_peaks) *n_peaks = np;
  /* :76-119 */
  int cnt = 0;
  const float thr = p->beam_distance_increment_threshold;
  for (int q = 0; q < n_idz; ++q) {
    const long long m = (long long)idzs[q];
    long long closer, further;
    long long d = 0;
    for (;;) {
      const long long b = m - d - 1;
      if (b < 0 || b > (long long)n - 1) { closer = m - d; break; } /* SPEC DECISION: closer_idx is left uninitialised in the reference */
      const long long a = m - d;
      if (((double)(hypot_f(PX(a), PY(a)) - hypot_f(PX(b), PY(b))) > (double)thr) || (PI_(a) <= PI_(b)) ||
          ((double)hypot_f(PX(a), PY(a)) < (double)min_d)) {
        closer = a;
        break;
      }
      ++d;
    }
    d = 0;
    for (;;) {
      const long long b = m + d + 1;
      if (b < 0 || b > (long long)n - 1) { further = m + d; break; } /* SPEC DECISION: idem */
      const long long a = m + d;
      if (((double)(hypot_f(PX(a), PY(a)) - hypot_f(PX(b), PY(b))) > (double)thr) || (PI_(a) <= PI_(b)) ||
          ((double)hypot_f(PX(a), PY(a)) < (double)min_d)) {
        further = a;
        break;
      }
      ++d;
    }
    for (long long j = closer; j <= further; ++j) {
      const float dist = hypot_f(PX(j), PY(j));
      const float angle = atan2f(PY(j), PX(j));
      const float intensity = PI_(j);
      if ((double)dist > (double)min_d && (double)dist < (double)max_d && (double)intensity > (double)min_i) {
        if (cnt >= capacity) { free(idzs); return -1; }
        const float x = PX(j), y = PY(j), z = PZ(j);
        const float* T = p->sensor_to_base;
        /* pcl::transformPointCloud with an Affine3f */
        out_pts[4 * cnt + 0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
        out_pts[4 * cnt + 1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
        out_pts[4 * cnt + 2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
        out_pts[4 * cnt + 3] = intensity;
        if (out_polar) {
          out_polar[2 * cnt + 0] = angle;
          out_polar[2 * cnt + 1] = dist;
        }
        ++cnt;
      }
    }
  }
  free(idzs);
  return cnt;
#undef PX
#undef PY
#undef PZ
#undef PI_
}

/* ============================================================ f-2: CS divergence ========== */

static void cell_full3f(const orc_cell* c, float S[3][3]) {
  S[0][0] = c->cov[0]; S[0][1] = S[1][0] = c->cov[1]; S[0][2] = S[2][0] = c->cov[2];
  S[1][1] = c->cov[3]; S[1][2] = S[2][1] = c->cov[4]; S[2][2] = c->cov[5];
}
/* Eigen bruteforce_det3_helper order */
static float det3f(float m[3][3]) {
#define H3(a, b, c) (m[0][a] * (m[1][b] * m[2][c] - m[1][c] * m[2][b]))
  return H3(0, 1, 2) - H3(1, 0, 2) + H3(2, 0, 1);
#undef H3
}
/* Eigen 3.3 cofactor inverse (see mahalanobis3f) */
static void inv3f(float S[3][3], float inv[3][3]) {
#define COF(i, j) (S[((i) + 1) % 3][((j) + 1) % 3] * S[((i) + 2) % 3][((j) + 2) % 3] - S[((i) + 1) % 3][((j) + 2) % 3] * S[((i) + 2) % 3][((j) + 1) % 3])
  float c0 = COF(0, 0), c1 = COF(1, 0), c2 = COF(2, 0);
  float det = (c0 * S[0][0] + c1 * S[1][0]) + c2 * S[2][0];
  float invdet = 1.0f / det;
  inv[0][0] = c0 * invdet; inv[0][1] = c1 * invdet; inv[0][2] = c2 * invdet;
  inv[1][0] = COF(0, 1) * invdet; inv[1][1] = COF(1, 1) * invdet; inv[1][2] = COF(2, 1) * invdet;
  inv[2][0] = COF(0, 2) * invdet; inv[2][1] = COF(1, 2) * invdet; inv[2][2] = COF(2, 2) * invdet;
#undef COF
}
/* (0.5 / sqrt(pi^2 det(Sf+Sq))) * exp(-0.5 d^T (Sf+Sq)^-1 d), ndt_map.cpp:60-64 */
static double cs_pair(const orc_cell* f, const orc_cell* q) {
  float Sf[3][3], Sq[3][3], M[3][3], inv[3][3];
  cell_full3f(f, Sf);
  cell_full3f(q, Sq);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i][j] = Sf[i][j] + Sq[i][j];
  float d[3] = {f->mean[0] - q->mean[0], f->mean[1] - q->mean[1], f->mean[2] - q->mean[2]};
  inv3f(M, inv);
  float row[3];
  for (int j = 0; j < 3; ++j) row[j] = (d[0] * inv[0][j] + d[1] * inv[1][j]) + d[2] * inv[2][j];
  const double e = (double)((row[0] * d[0] + row[1] * d[1]) + row[2] * d[2]);
  return (0.5 / sqrt(M_PI * M_PI * (double)det3f(M))) * exp(-0.5 * e);
}

double orc_cs_divergence(const orc_map* fixed, const orc_map* moving, double terms[3]) {
  double interaction = 0.0, fixed_term = 0.0, moving_term = 0.0;
  for (int fi = 0; fi < fixed->n_cells; ++fi) {
    float Sf[3][3], inv[3][3];
    cell_full3f(&fixed->cells[fi], Sf);
    if ((double)det3f(Sf) < 0.00001) continue;
    for (int qi = 0; qi < moving->n_cells; ++qi) interaction += cs_pair(&fixed->cells[fi], &moving->cells[qi]);
    inv3f(Sf, inv);
    fixed_term += (double)sqrtf(det3f(inv)) / (2 * M_PI);
    for (int qi = 0; qi < fi; ++qi) fixed_term += 2 * cs_pair(&fixed->cells[fi], &fixed->cells[qi]);
  }
  for (int fi = 0; fi < moving->n_cells; ++fi) {
    float Sf[3][3], inv[3][3];
    cell_full3f(&moving->cells[fi], Sf);
    if ((double)det3f(Sf) < 0.00001) continue;
    inv3f(Sf, inv);
    moving_term += (double)sqrtf(det3f(inv)) / (2 * M_PI);
    for (int qi = 0; qi < fi; ++qi) moving_term += 2 * cs_pair(&moving->cells[fi], &moving->cells[qi]);
  }
  if (terms) {
    terms[0] = interaction;
    terms[1] = fixed_term;
    terms[2] = moving_term;
  }
  return -log(interaction) + 0.5 * log(fixed_term) + 0.5 * log(moving_term);
}

/* ============================================================ f-3: correlative search ===== */

void orc_eval_cost_batch(const orc_map* fixed, const orc_map* moving, const int32_t* corr, int k, int use_intensity,
                         double scale, double alpha, const double* poses4, int n_poses, double* cost, int* n_res) {
  const int d = use_intensity ? 3 : 2;
  int n = 0;
  for (int i = 0; i < moving->n_cells * k; ++i) n += corr[i] >= 0;
  if (n_res) *n_res = n;
  for (int p = 0; p < n_poses; ++p) {
    double total = 0.0;
    for (int i = 0; i < moving->n_cells; ++i)
      for (int j = 0; j < k; ++j) {
        const int32_t ci = corr[(size_t)i * k + j];
        if (ci < 0) continue;
        double mm[3], fm[3], mc[9], fc[9];
        for (int e = 0; e < d; ++e) {
          mm[e] = (double)moving->cells[i].mean[e];
          fm[e] = (double)fixed->cells[ci].mean[e];
        }
        cell_cov_full(&moving->cells[i], d, mc);
        cell_cov_full(&fixed->cells[ci], d, fc);
        const double r = orc_ndt_residual(d, ORC_PARAM_AMBIENT4, poses4 + 4 * (size_t)p, mm, mc, fm, fc, NULL);
        double rho[3];
        orc_barron_scaled(r * r, scale, alpha, 1.0, 1.0, rho); /* BarronLoss(scale, alpha): b = a^2 */
        total += 0.5 * rho[0];
      }
    cost[p] = total;
  }
}

typedef struct { double pose[4]; int level; } bnb_node;

static void se2_from_angle(double a, double tx, double ty, double out[4]) {
  double c = cos(a), s = sin(a);
  so2_normalize(&c, &s);
  out[0] = c; out[1] = s; out[2] = tx; out[3] = ty;
}
/* std::vector<float>(matrix.data(), ...) of the 3x3 homogeneous matrix, column-major */
static void pose_key(const double p[4], float key[9]) {
  key[0] = (float)p[0]; key[1] = (float)p[1]; key[2] = 0.f;
  key[3] = (float)(-p[1]); key[4] = (float)p[0]; key[5] = 0.f;
  key[6] = (float)p[2]; key[7] = (float)p[3]; key[8] = 1.f;
}

double orc_search_global_bnb(const orc_map* fixed, const orc_map* moving, const orc_matcher_params* p, const orc_bnb_params* bp,
                             double scale, double swl, double swa, double trans4[4], int* n_evals) {
  swl = fmin(swl, bp->csm_window_linear);
  swa = fmin(swa, bp->csm_window_angular);
  const int k = 4; /* addNDTFactor(..., 4) (ndt_matcher.cpp:520) */
  int32_t* corr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(moving->n_cells > 0 ? moving->n_cells : 1) * k);
  orc_associate(fixed, moving, trans4, k, p->lookup_mahalanobis, p->use_intensity, corr);
  const double linear_step = bp->csm_linear_step, max_range = bp->csm_max_px_accurate_range;
  const double angular_step = acos(1 - ((linear_step * linear_step) / (2 * max_range * max_range)));
  const double cost_threshold = bp->csm_cost_threshold;
  const size_t n_iter = (size_t)bp->csm_n_iter;
  const double initial_linear_step = pow(2, (double)n_iter - 1) * linear_step;
  const double initial_angular_step = angular_step;
  double min_cost = 100000.0;
  size_t cap = 1024, nq = 0, head = 0, nk = 0;
  bnb_node* q = (bnb_node*)malloc(sizeof(bnb_node) * cap);
  float* keys = (float*)malloc(sizeof(float) * 9 * cap);
#define PUSH(P4, LVL)                                                   \
  do {                                                                  \
    if (nq == cap) {                                                    \
      cap *= 2;                                                         \
      q = (bnb_node*)realloc(q, sizeof(bnb_node) * cap);                \
      keys = (float*)realloc(keys, sizeof(float) * 9 * cap);            \
    }                                                                   \
    memcpy(q[nq].pose, (P4), sizeof(double) * 4);                       \
    q[nq].level = (LVL);                                                \
    ++nq;                                                               \
  } while (0)
  for (double tx = -swl / 2.0; tx <= swl / 2.0; tx += initial_linear_step)
    for (double ty = -swl / 2.0; ty <= swl / 2.0; ty += initial_linear_step)
      for (double a = -swa / 2.0; a < swa / 2.0; a += initial_angular_step) {
        double d4[4], cur[4];
        se2_from_angle(a, tx, ty, d4);
        orc_se2_mul(trans4, d4, cur);
        PUSH(cur, 1);
        pose_key(cur, keys + 9 * nk);
        ++nk;
      }
  double best[4] = {1.0, 0.0, 0.0, 0.0}; /* Sophus::SE2d best_trans: identity if nothing qualifies */
  int evals = 0, n_res = 0;
  while (head < nq) {
    double cost;
    orc_eval_cost_batch(fixed, moving, corr, k, p->use_intensity, scale, p->loss_alpha, q[head].pose, 1, &cost, &n_res);
    ++evals;
    const double current_cost = cost / (double)n_res;
    const size_t level = (size_t)q[head].level;
    if (current_cost < cost_threshold) {
      if (current_cost < min_cost) {
        memcpy(best, q[head].pose, sizeof(best));
        min_cost = current_cost;
      }
      if (level < n_iter) {
        const double cls = pow(2.0, (double)level) * linear_step, cas = angular_step;
        for (double tx = -cls; tx <= cls; tx += cls)
          for (double ty = -cls; ty <= cls; ty += cls)
            for (double a = -cas; a <= cas; a += cas) {
              double d4[4], smp[4];
              float key[9];
              se2_from_angle(a, tx, ty, d4);
              orc_se2_mul(q[head].pose, d4, smp);
              pose_key(smp, key);
              int found = 0;
              for (size_t t = 0; t < nk && !found; ++t) found = memcmp(keys + 9 * t, key, sizeof(key)) == 0 ? 1 : 0;
              if (!found) {
                PUSH(smp, (int)level + 1); /* may realloc keys */
                memcpy(keys + 9 * nk, key, sizeof(key));
                ++nk;
              }
            }
      }
    }
    ++head;
  }
#undef PUSH
  memcpy(trans4, best, sizeof(best));
  if (n_evals) *n_evals = evals;
  free(q); free(keys); free(corr);
  return min_cost;
}

/* ================================================================ f-4: Scan Context ========= */
/* xy2theta (Scancontext.cpp:24-37): degrees in [0, 360].  SPEC DECISION: the reference's float arctangent is taken as
 * the correctly rounded one, (float)atan((double)t) -- glibc's atanf may differ from it by one ulp, which moves a
 * point to the neighbouring sector only if it lies within ~1e-7 of a sector boundary; this form is identical on
 * every platform (host libm and device). */
static float sc_atanf(float t) { return (float)atan((double)t); }
static float sc_xy2theta(float x, float y) {
  if (x >= 0 && y >= 0) return (float)((180 / M_PI) * sc_atanf(y / x));
  if (x < 0 && y >= 0) return (float)(180 - ((180 / M_PI) * sc_atanf(y / (-x))));
  if (x < 0 && y < 0) return (float)(180 + ((180 / M_PI) * sc_atanf(y / x)));
  if (x >= 0 && y < 0) return (float)(360 - ((180 / M_PI) * sc_atanf((-y) / x)));
  return 0.0f; /* NaN coordinates: the reference falls off the end of the function (undefined) */
}

void orc_sc_make(const float* pts, int n, int stride, int ioff, const orc_sc_params* p, double* desc, double* ring_key,
                 double* sector_key) {
  const int R = p->num_ring, S = p->num_sector;
  const double NO_POINT = -1000;
  for (int i = 0; i < R * S; ++i) desc[i] = NO_POINT;
  for (int i = 0; i < n; ++i) {
    const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1];
    const float z = (float)(pts[(size_t)i * stride + ioff] * p->intensity_factor); /* pt.z = intensity * factor (:171) */
    const float azim_range = sqrtf(x * x + y * y);
    const float azim_angle = sc_xy2theta(x, y);
    if (azim_range > p->max_radius) continue;
    int ring = (int)ceil((azim_range / p->max_radius) * R);
    ring = ring < R ? ring : R;
    ring = ring > 1 ? ring : 1;
    int sect = (int)ceil((azim_angle / 360.0) * S);
    sect = sect < S ? sect : S;
    sect = sect > 1 ? sect : 1;
    /* quirk (:187): the bin started at NO_POINT and the values are ADDED to it */
    desc[(size_t)(sect - 1) * R + (ring - 1)] += z;
  }
  for (int i = 0; i < R * S; ++i)
    if (desc[i] == NO_POINT) desc[i] = 0;
  for (int r = 0; r < R; ++r) { /* rowwise mean */
    double a = 0;
    for (int s = 0; s < S; ++s) a += desc[(size_t)s * R + r];
    ring_key[r] = a / S;
  }
  for (int s = 0; s < S; ++s) { /* columnwise mean */
    double a = 0;
    for (int r = 0; r < R; ++r) a += desc[(size_t)s * R + r];
    sector_key[s] = a / R;
  }
}

/* distDirectSC (:64-87) of sc1 against sc2 circularly shifted right by `shift` columns */
static double sc_dist_direct(const double* sc1, const double* sc2, int R, int S, int shift) {
  int n_eff = 0;
  double sum = 0;
  for (int col = 0; col < S; ++col) {
    const double* a = sc1 + (size_t)col * R;
    const double* b = sc2 + (size_t)((col - shift + S) % S) * R; /* shifted.col(col) = sc2.col(col - shift) */
    double na = 0, nb = 0, dot = 0;
    for (int r = 0; r < R; ++r) {
      na += a[r] * a[r];
      nb += b[r] * b[r];
      dot += a[r] * b[r];
    }
    na = sqrt(na);
    nb = sqrt(nb);
    if (na == 0 || nb == 0) continue;
    sum = sum + dot / (na * nb);
    n_eff = n_eff + 1;
  }
  return 1.0 - sum / n_eff; /* 0/0 = NaN when no column counts, like the reference */
}

static int sc_cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

double orc_sc_distance(const orc_sc_params* p, const double* sc1, const double* sc2, const double pos1[2], const double pos2[2],
                       double dist1, double dist2, int* shift_out) {
  const int R = p->num_ring, S = p->num_sector;
  double* k1 = (double*)malloc(sizeof(double) * 2 * S);
  double* k2 = k1 + S;
  for (int s = 0; s < S; ++s) {
    double a = 0, b = 0;
    for (int r = 0; r < R; ++r) {
      a += sc1[(size_t)s * R + r];
      b += sc2[(size_t)s * R + r];
    }
    k1[s] = a / R;
    k2[s] = b / R;
  }
  /* fastAlignUsingVkey (:90-112) */
  int argmin_vkey = 0;
  double min_norm = 10000000;
  for (int sh = 0; sh < S; ++sh) {
    double nn = 0;
    for (int s = 0; s < S; ++s) {
      const double d = k1[s] - k2[(s - sh + S) % S];
      nn += d * d;
    }
    nn = sqrt(nn);
    if (nn < min_norm) {
      argmin_vkey = sh;
      min_norm = nn;
    }
  }
  free(k1);
  const int radius = (int)round(0.5 * p->search_ratio * S);
  int* space = (int*)malloc(sizeof(int) * (2 * radius + 1));
  int ns = 0;
  space[ns++] = argmin_vkey;
  for (int ii = 1; ii < radius + 1; ++ii) {
    space[ns++] = (argmin_vkey + ii + S) % S;
    space[ns++] = (argmin_vkey - ii + S) % S;
  }
  qsort(space, ns, sizeof(int), sc_cmp_int);
  int argmin_shift = 0;
  double min_sc = 10000000;
  for (int i = 0; i < ns; ++i) {
    const double d = sc_dist_direct(sc1, sc2, R, S, space[i]);
    if (d < min_sc) {
      argmin_shift = space[i];
      min_sc = d;
    }
  }
  free(space);
  const double dx = pos2[0] - pos1[0], dy = pos2[1] - pos1[1];
  double t_err = sqrt(dx * dx + dy * dy) - p->odom_eps;
  t_err = (t_err > 0.0 ? t_err : 0.0) / (dist2 - dist1);
  const double odom_dist = 1 - exp(-(t_err * t_err) / (2 * p->assumed_drift * p->assumed_drift));
  if (shift_out) *shift_out = argmin_shift;
  return min_sc + odom_dist * R * p->odom_weight;
}

int orc_sc_detect(const orc_sc_params* p, const double* desc, const double* ring_keys, const double* pos, const double* dist,
                  int n_db, int node_id, float* yaw, double* min_dist_out) {
  const int R = p->num_ring, S = p->num_sector, K = p->num_candidates;
  if (yaw) *yaw = 0.0f;
  if (min_dist_out) *min_dist_out = 10000000;
  if (node_id < p->num_exclude_recent + 1 || node_id >= n_db) return -1;
  const int n_search = node_id + 1 - p->num_exclude_recent; /* keys[0 .. node_id - NUM_EXCLUDE_RECENT] */
  /* exact kNN on the float ring keys (squared L2 accumulated in float like nanoflann's L2 adaptor) */
  float* d2 = (float*)malloc(sizeof(float) * n_search);
  for (int i = 0; i < n_search; ++i) {
    float acc = 0.0f;
    for (int r = 0; r < R; ++r) {
      const float a = (float)ring_keys[(size_t)node_id * R + r], b = (float)ring_keys[(size_t)i * R + r];
      const float d = a - b;
      acc += d * d;
    }
    d2[i] = acc;
  }
  double min_d = 10000000;
  int nn_align = 0, nn_idx = 0;
  const int kk = K < n_search ? K : n_search;
  for (int c = 0; c < kk; ++c) {
    int best = -1;
    for (int i = 0; i < n_search; ++i)
      if (d2[i] >= 0.0f && (best < 0 || d2[i] < d2[best])) best = i;
    d2[best] = -1.0f; /* taken */
    int sh;
    const double d = orc_sc_distance(p, desc + (size_t)node_id * R * S, desc + (size_t)best * R * S, pos + 2 * (size_t)node_id,
                                     pos + 2 * (size_t)best, dist[node_id], dist[best], &sh);
    if (d < min_d) {
      min_d = d;
      nn_align = sh;
      nn_idx = best;
    }
  }
  free(d2);
  if (min_dist_out) *min_dist_out = min_d;
  /* deg2rad(nn_align * PC_UNIT_SECTORANGLE) in float (:336, :17-20) */
  if (yaw) *yaw = (float)((float)(nn_align * (360.0 / (double)S)) * M_PI / 180.0);
  return min_d < p->dist_thresh ? nn_idx : -1;
}

/* ============================================================ f-4: pose graph ================= */

void orc_pg_params_default(orc_pg_params* p) {
  memset(p, 0, sizeof(*p));
  p->use_robust_loss = 0;  /* every shipped config (parameters_*.yaml global_fuser.use_robust_loss: false) */
  p->loss_scale = 60.0;
  p->max_iterations = 200000; /* global_fuser.cpp:52 */
  p->max_consecutive_invalid_steps = 5;
  p->function_tolerance = 1e-6;
  p->gradient_tolerance = 1e-10;
  p->parameter_tolerance = 1e-8;
  p->initial_radius = 1e4;
  p->max_radius = 1e16;
  p->min_radius = 1e-32;
  p->min_relative_decrease = 1e-3;
  p->min_lm_diagonal = 1e-6;
  p->max_lm_diagonal = 1e32;
}

/* state_manifold.h:17-23 */
static double pg_normalize_angle(double a) {
  const double two_pi = 2.0 * M_PI;
  return a - two_pi * floor((a + M_PI) / two_pi);
}

/* pose_graph_2d_error_term.h:44-60; the autodiff Jacobian written out: d/dyaw_a of R(yaw_a)^T v = [[-s, c], [-c, -s]] v,
 * floor() has derivative zero so the angle row is (-1, +1). */
void orc_pg_edge(const double pa[3], const double pb[3], const double meas[3], const double sqi[9], double r[3], double Ja[9],
                 double Jb[9]) {
  const double c = cos(pa[2]), s = sin(pa[2]);
  const double dx = pb[0] - pa[0], dy = pb[1] - pa[1];
  double e[3], A[9], B[9];
  e[0] = (c * dx + s * dy) - meas[0];
  e[1] = (-s * dx + c * dy) - meas[1];
  e[2] = pg_normalize_angle((pb[2] - pa[2]) - meas[2]);
  /* unweighted Jacobians, row-major 3x3 */
  A[0] = -c; A[1] = -s; A[2] = -s * dx + c * dy;
  A[3] = s;  A[4] = -c; A[5] = -c * dx - s * dy;
  A[6] = 0;  A[7] = 0;  A[8] = -1.0;
  B[0] = c;  B[1] = s;  B[2] = 0;
  B[3] = -s; B[4] = c;  B[5] = 0;
  B[6] = 0;  B[7] = 0;  B[8] = 1.0;
  for (int i = 0; i < 3; ++i) {
    r[i] = sqi[i * 3 + 0] * e[0] + sqi[i * 3 + 1] * e[1] + sqi[i * 3 + 2] * e[2];
    for (int j = 0; j < 3; ++j) {
      Ja[i * 3 + j] = sqi[i * 3 + 0] * A[0 + j] + sqi[i * 3 + 1] * A[3 + j] + sqi[i * 3 + 2] * A[6 + j];
      Jb[i * 3 + j] = sqi[i * 3 + 0] * B[0 + j] + sqi[i * 3 + 1] * B[3 + j] + sqi[i * 3 + 2] * B[6 + j];
    }
  }
}

typedef struct pg_problem {
  int n_poses, n_used;
  const int32_t* ia;  /* per used edge */
  const int32_t* ib;
  const double* meas; /* [n_used][3] */
  const double* sqi;  /* [n_used][9] */
  int robust;
  double huber_a;
  const int32_t* var; /* pose -> first tangent index, or -1 (constant / not in the problem) */
  int nt;
} pg_problem;

/* cost = sum 1/2 rho(|r|^2); r / Ja / Jb are the loss-corrected ones (corrector.cc with rho'' <= 0: scale by sqrt(rho')) */
static double pg_eval(const pg_problem* P, const double* poses, double* r, double* Ja, double* Jb) {
  double cost = 0.0;
  for (int e = 0; e < P->n_used; ++e) {
    double re[3], A[9], B[9];
    orc_pg_edge(poses + 3 * P->ia[e], poses + 3 * P->ib[e], P->meas + 3 * e, P->sqi + 9 * e, re, A, B);
    const double s = re[0] * re[0] + re[1] * re[1] + re[2] * re[2];
    double rho0 = s, rho1 = 1.0;
    if (P->robust) {
      const double b = P->huber_a * P->huber_a; /* HuberLoss: b_ = a * a */
      if (s > b) {
        const double rr = sqrt(s);
        rho0 = 2.0 * P->huber_a * rr - b;
        rho1 = fmax(DBL_MIN, P->huber_a / rr);
      }
    }
    cost += 0.5 * rho0;
    if (r) {
      const double w = sqrt(rho1);
      for (int i = 0; i < 3; ++i) r[3 * e + i] = re[i] * w;
      for (int i = 0; i < 9; ++i) {
        Ja[9 * e + i] = A[i] * w;
        Jb[9 * e + i] = B[i] * w;
      }
    }
  }
  return cost;
}

int orc_pose_graph_optimize(int n_poses, double* poses, int n_edges, const int32_t* id_begin, const int32_t* id_end,
                            const double* meas, const double* sqrt_info, int max_update_index, const orc_pg_params* opt,
                            orc_pg_result* out) {
  if (n_poses <= 0 || n_edges < 0) return -1;
  int32_t* ia = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_edges + 1));
  int32_t* ib = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_edges + 1));
  double* um = (double*)malloc(sizeof(double) * 3 * (size_t)(n_edges + 1));
  double* us = (double*)malloc(sizeof(double) * 9 * (size_t)(n_edges + 1));
  int32_t* var = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_poses);
  for (int i = 0; i < n_poses; ++i) var[i] = -2; /* -2 = not referenced */
  int nu = 0, bad = 0;
  for (int e = 0; e < n_edges; ++e) {
    const int a = id_begin[e], b = id_end[e];
    if (!(a + 1 == b || b <= max_update_index)) continue; /* global_fuser.cpp:32 */
    if (a < 0 || b < 0 || a >= n_poses || b >= n_poses || a == b) { bad = 1; break; }
    ia[nu] = a; ib[nu] = b;
    memcpy(um + 3 * nu, meas + 3 * e, sizeof(double) * 3);
    memcpy(us + 9 * nu, sqrt_info + 9 * e, sizeof(double) * 9);
    var[a] = var[b] = -1;
    ++nu;
  }
  int nt = 0;
  if (!bad) {
    for (int i = 1; i < n_poses; ++i) /* pose 0 = poses.begin(): SetParameterBlockConstant (:48-49) */
      if (var[i] == -1) { var[i] = nt; nt += 3; }
    var[0] = -1;
    for (int i = 1; i < n_poses; ++i) if (var[i] == -2) var[i] = -1;
  }
  if (out) {
    memset(out, 0, sizeof(*out));
    out->n_residual_blocks = nu;
    out->n_loop_closures = n_edges + 1 - n_poses; /* :26 */
  }
  if (bad || nu == 0 || nt == 0) {
    free(ia); free(ib); free(um); free(us); free(var);
    return bad ? -1 : 0;
  }
  pg_problem P = {n_poses, nu, ia, ib, um, us, opt->use_robust_loss, opt->loss_scale, var, nt};

  double* x = (double*)malloc(sizeof(double) * 3 * (size_t)n_poses);
  double* cand = (double*)malloc(sizeof(double) * 3 * (size_t)n_poses);
  double* best = (double*)malloc(sizeof(double) * 3 * (size_t)n_poses);
  double* r = (double*)malloc(sizeof(double) * 3 * (size_t)nu);
  double* Ja = (double*)malloc(sizeof(double) * 9 * (size_t)nu);
  double* Jb = (double*)malloc(sizeof(double) * 9 * (size_t)nu);
  double* H = (double*)malloc(sizeof(double) * (size_t)nt * nt);
  double* Hw = (double*)malloc(sizeof(double) * (size_t)nt * nt);
  double* g = (double*)malloc(sizeof(double) * (size_t)nt);
  double* gw = (double*)malloc(sizeof(double) * (size_t)nt);
  double* scaling = (double*)malloc(sizeof(double) * (size_t)nt);
  double* diagonal = (double*)malloc(sizeof(double) * (size_t)nt);
  double* step = (double*)malloc(sizeof(double) * (size_t)nt);
  double* delta = (double*)malloc(sizeof(double) * (size_t)nt);
  memcpy(x, poses, sizeof(double) * 3 * (size_t)n_poses);
  memcpy(best, poses, sizeof(double) * 3 * (size_t)n_poses);

  int term = ORC_TERM_FAILURE;
  double x_cost = 0, x_norm = 0, cand_cost = 0, grad_max_norm = 0;
  double radius = opt->initial_radius, decrease_factor = 2.0;
  int reuse_diagonal = 0, num_invalid = 0, iteration = 0, step_successful = 1, n_it = 0;
  double minimum_cost = DBL_MAX, summary_min_cost;

  /* gradient, Jacobi scaling (first call), scaled J^T J (dense, tangent order = ascending pose id) */
#define PG_LINEARIZE(FIRST)                                                                               \
  do {                                                                                                    \
    x_cost = pg_eval(&P, x, r, Ja, Jb);                                                                   \
    memset(H, 0, sizeof(double) * (size_t)nt * nt);                                                       \
    memset(g, 0, sizeof(double) * (size_t)nt);                                                            \
    for (int e = 0; e < nu; ++e) {                                                                        \
      const int va = var[ia[e]], vb = var[ib[e]];                                                         \
      const double* A = Ja + 9 * e;                                                                       \
      const double* B = Jb + 9 * e;                                                                       \
      for (int i = 0; i < 3; ++i)                                                                         \
        for (int j = 0; j < 3; ++j) {                                                                     \
          double aa = 0, ab = 0, bb = 0;                                                                  \
          for (int k = 0; k < 3; ++k) {                                                                   \
            aa += A[k * 3 + i] * A[k * 3 + j];                                                            \
            ab += A[k * 3 + i] * B[k * 3 + j];                                                            \
            bb += B[k * 3 + i] * B[k * 3 + j];                                                            \
          }                                                                                               \
          if (va >= 0) H[(size_t)(va + i) * nt + va + j] += aa;                                           \
          if (vb >= 0) H[(size_t)(vb + i) * nt + vb + j] += bb;                                           \
          if (va >= 0 && vb >= 0) {                                                                       \
            H[(size_t)(va + i) * nt + vb + j] += ab;                                                      \
            H[(size_t)(vb + j) * nt + va + i] += ab;                                                      \
          }                                                                                               \
        }                                                                                                 \
      for (int i = 0; i < 3; ++i) {                                                                       \
        double ga = 0, gb = 0;                                                                            \
        for (int k = 0; k < 3; ++k) {                                                                     \
          ga += A[k * 3 + i] * r[3 * e + k];                                                              \
          gb += B[k * 3 + i] * r[3 * e + k];                                                              \
        }                                                                                                 \
        if (va >= 0) g[va + i] += ga;                                                                     \
        if (vb >= 0) g[vb + i] += gb;                                                                     \
      }                                                                                                   \
    }                                                                                                     \
    if (FIRST)                                                                                            \
      for (int j = 0; j < nt; ++j) scaling[j] = 1.0 / (1.0 + sqrt(H[(size_t)j * nt + j]));                \
    grad_max_norm = 0;                                                                                    \
    for (int j = 0; j < nt; ++j) grad_max_norm = fmax(grad_max_norm, fabs(g[j]));                         \
    for (int i = 0; i < nt; ++i)                                                                          \
      for (int j = 0; j < nt; ++j) H[(size_t)i * nt + j] *= scaling[i] * scaling[j];                      \
    for (int j = 0; j < nt; ++j) g[j] *= scaling[j];                                                      \
  } while (0)
#define PG_XNORM(V, OUT)                                                                                  \
  do {                                                                                                    \
    double acc_ = 0;                                                                                      \
    for (int i = 0; i < n_poses; ++i)                                                                     \
      if (var[i] >= 0) acc_ += V[3 * i] * V[3 * i] + V[3 * i + 1] * V[3 * i + 1] + V[3 * i + 2] * V[3 * i + 2]; \
    OUT = sqrt(acc_);                                                                                     \
  } while (0)

  PG_XNORM(x, x_norm);
  PG_LINEARIZE(1);
  if (out) out->initial_cost = x_cost;
  summary_min_cost = x_cost;
  n_it = 1;

  for (;;) {
    if (step_successful && x_cost < minimum_cost) {
      minimum_cost = x_cost;
      memcpy(best, x, sizeof(double) * 3 * (size_t)n_poses);
    }
    if (iteration >= opt->max_iterations) { term = ORC_TERM_NO_CONVERGENCE; break; }
    if (step_successful && grad_max_norm <= opt->gradient_tolerance) { term = ORC_TERM_CONVERGENCE_GRADIENT; break; }
    if (radius <= opt->min_radius) { term = ORC_TERM_CONVERGENCE_RADIUS; break; }
    ++iteration;
    ++n_it;

    if (!reuse_diagonal)
      for (int j = 0; j < nt; ++j) diagonal[j] = fmin(fmax(H[(size_t)j * nt + j], opt->min_lm_diagonal), opt->max_lm_diagonal);
    memcpy(Hw, H, sizeof(double) * (size_t)nt * nt);
    for (int j = 0; j < nt; ++j) {
      const double d = sqrt(diagonal[j] / radius);
      Hw[(size_t)j * nt + j] += d * d;
    }
    memcpy(gw, g, sizeof(double) * (size_t)nt);
    int solved = chol_solve(Hw, gw, nt, step);
    for (int j = 0; j < nt; ++j) {
      if (!isfinite(step[j])) solved = 0;
      step[j] = -step[j];
    }
    reuse_diagonal = 1;
    int step_valid = 0;
    double model_cost_change = 0;
    if (solved) {
      /* -(J step)^T (r + J step / 2) = -(step.g + step^T H step / 2) on the scaled system */
      double acc = 0;
      for (int i = 0; i < nt; ++i) {
        double hs = 0;
        for (int j = 0; j < nt; ++j) hs += H[(size_t)i * nt + j] * step[j];
        acc += step[i] * (g[i] + 0.5 * hs);
      }
      model_cost_change = -acc;
      step_valid = model_cost_change > 0.0;
    }
    if (!step_valid) {
      if (++num_invalid >= opt->max_consecutive_invalid_steps) { term = ORC_TERM_FAILURE; break; }
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
      step_successful = 0;
      if (x_cost < summary_min_cost) summary_min_cost = x_cost;
      continue;
    }
    num_invalid = 0;
    for (int j = 0; j < nt; ++j) delta[j] = step[j] * scaling[j];
    memcpy(cand, x, sizeof(double) * 3 * (size_t)n_poses);
    for (int i = 0; i < n_poses; ++i)
      if (var[i] >= 0)
        for (int k = 0; k < 3; ++k) cand[3 * i + k] = x[3 * i + k] + delta[var[i] + k];
    cand_cost = pg_eval(&P, cand, NULL, NULL, NULL);

    double step_norm = 0;
    for (int i = 0; i < n_poses; ++i)
      if (var[i] >= 0)
        for (int k = 0; k < 3; ++k) step_norm += (x[3 * i + k] - cand[3 * i + k]) * (x[3 * i + k] - cand[3 * i + k]);
    step_norm = sqrt(step_norm);
    if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { term = ORC_TERM_CONVERGENCE_PARAMETER; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= opt->function_tolerance * x_cost) { term = ORC_TERM_CONVERGENCE_FUNCTION; break; }
    const double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > opt->min_relative_decrease) {
      memcpy(x, cand, sizeof(double) * 3 * (size_t)n_poses);
      PG_XNORM(x, x_norm);
      PG_LINEARIZE(0);
      step_successful = 1;
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3));
      radius = fmin(opt->max_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = 0;
      if (x_cost < summary_min_cost) summary_min_cost = x_cost;
    } else {
      step_successful = 0;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
      if (cand_cost < summary_min_cost) summary_min_cost = cand_cost;
    }
  }
#undef PG_LINEARIZE
#undef PG_XNORM
  memcpy(poses, best, sizeof(double) * 3 * (size_t)n_poses);
  if (out) {
    out->final_cost = summary_min_cost;
    out->iterations = n_it;
    out->termination = term;
  }
  free(x); free(cand); free(best); free(r); free(Ja); free(Jb); free(H); free(Hw); free(g); free(gw);
  free(scaling); free(diagonal); free(step); free(delta);
  free(ia); free(ib); free(um); free(us); free(var);
  return 0;
}
