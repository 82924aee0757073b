/*
 * orc_odometry.c — CPU ORACLE (test infrastructure): restatement of the reference class
 * RGBDOdometry (elasticfusion/Core/src/Utils/RGBDOdometry.cpp:21-610) and of the Eigen
 * host arithmetic it performs between kernels.  See orc_track.c for the status header
 * (which parts are pinned to the reference's own kernels, and that the Eigen host arithmetic restated here is not; who may load
 * this library).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"
#include "orc_scalar.h"

#define NUM_PYRS 3

struct orc_odometry {
  int width, height;
  float cx, cy, fx, fy, distThres, angleThres;
  float sobelScale, maxDepthDeltaRGB, maxDepthRGB;
  float minGrad[NUM_PYRS];
  uint16_t* depth_tmp[NUM_PYRS];
  float *vmaps_g_prev[NUM_PYRS], *nmaps_g_prev[NUM_PYRS], *vmaps_curr[NUM_PYRS], *nmaps_curr[NUM_PYRS];
  float *lastDepth[NUM_PYRS], *nextDepth[NUM_PYRS];
  uint8_t *lastImage[NUM_PYRS], *nextImage[NUM_PYRS], *lastNextImage[NUM_PYRS];
  int16_t *nextdIdx[NUM_PYRS], *nextdIdy[NUM_PYRS];
  float* pointClouds[NUM_PYRS];
  orc_dataterm* corresImg[NUM_PYRS];
  float *vmaps_tmp, *nmaps_tmp;
  int exp_bias;   /* test hook: added to the static exponents of a call's first reductions (product: dms_odometry_debug_set "exp_bias") */
  int solve_mode; /* 1 (default): scalar section in the product's canonical operation order (orc_scalar.c); 0: the independent Eigen-like restatement below */
  int sum_mode;   /* 0: fp64 accumulation in loop order (order-dependent control), 1: canonical order-free sums (orc_canon.c) */
  int fused_rows; /* evaluate the Gauss-Newton rows with fused multiply-adds (orc_set_fused_rows), as the product's resident kernels do */
  /* Step hooks (sum_mode 0 only; NULL = the restatements in orc_track.c): same signatures as orc_so3Step / orc_computeRgbResidual /
   * orc_icpStep / orc_rgbStep.  tests/golden/make_ref_tracker_golden.py points them at the REFERENCE's own kernels
   * (oracle/_ref/libref_reduce.so) so that whole tracker calls run this host loop around the reference's device code. */
  void (*hook_so3)(const uint8_t*, const uint8_t*, const float*, const float*, const float*, int, int, float*, float*, float*);
  void (*hook_rgbres)(float, const int16_t*, const int16_t*, const float*, const float*, const uint8_t*, const uint8_t*, orc_dataterm*, float,
                      const float*, const float*, int, int, int*, int*);
  void (*hook_icp)(const float*, const float*, const float*, const float*, const float*, const float*, float, float, float, float, const float*,
                   const float*, float, float, int, int, float*, float*, float*);
  void (*hook_rgb)(const orc_dataterm*, float, const float*, float, float, const int16_t*, const int16_t*, float, int, int, float*, float*);
};

/* ---- small dense algebra (Eigen stand-ins, fp64 unless stated) ------------------------ */

/* Gauss-Jordan inverse with partial pivoting, n <= 6, row-major */
static void gj_inverse(const double* m, int n, double* out) {
  double a[36], inv[36];
  for (int i = 0; i < n * n; ++i) a[i] = m[i];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) inv[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r)
      if (fabs(a[r * n + c]) > fabs(a[p * n + c])) p = r;
    if (p != c)
      for (int j = 0; j < n; ++j) {
        double t = a[c * n + j]; a[c * n + j] = a[p * n + j]; a[p * n + j] = t;
        t = inv[c * n + j]; inv[c * n + j] = inv[p * n + j]; inv[p * n + j] = t;
      }
    const double d = a[c * n + c];
    for (int j = 0; j < n; ++j) { a[c * n + j] /= d; inv[c * n + j] /= d; }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = a[r * n + c];
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) { a[r * n + j] -= f * a[c * n + j]; inv[r * n + j] -= f * inv[c * n + j]; }
    }
  }
  for (int i = 0; i < n * n; ++i) out[i] = inv[i];
}

void orc_covariance(const double* lastA36, double* cov36) { gj_inverse(lastA36, 6, cov36); } /* RGBDOdometry.cpp:607-610 */

static void matmul_d(const double* a, const double* b, int n, double* o) {
  double t[36];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += a[i * n + k] * b[k * n + j];
      t[i * n + j] = s;
    }
  memcpy(o, t, sizeof(double) * n * n);
}

/* float 3x3 inverse by adjugate / determinant (Eigen's fixed-size 3x3 path) */
static void inv3f(const float* m, float* o) {
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const float id = 1.0f / det;
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/*
 * Symmetric solve the way Eigen's LDLT does it (`A.ldlt().solve(b)`, RGBDOdometry.cpp:371,554):
 * diagonal pivoting on the largest remaining |A_kk|, unit-lower L, D on the diagonal, zero
 * (or sub-denormal) pivots give a zero component.  Written in outer-product (right-looking)
 * form on a full symmetric working copy, in the scalar type given by `is_float`.
 */
static void sym_solve(const double* Ain, const double* bin, int n, int is_float, double* x) {
  double A[36], y[6];
  int perm[6];
#define RND(v) (is_float ? (double)(float)(v) : (v))
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
  for (int i = 0; i < n; ++i) { y[i] = bin[i]; perm[i] = i; }
  const double tiny = is_float ? (double)(1.0f / FLT_MAX) : 1.0 / DBL_MAX;
  int rank_zero = 0;
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (fabs(A[i * n + i]) > fabs(A[p * n + p])) p = i;
    if (p != k) { /* symmetric row/column exchange */
      for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
      for (int i = 0; i < n; ++i) { double t = A[i * n + k]; A[i * n + k] = A[i * n + p]; A[i * n + p] = t; }
      int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
      double ty = y[k]; y[k] = y[p]; y[p] = ty;
    }
    const double d = A[k * n + k];
    if (k == 0 && !(fabs(d) > 0.0)) { rank_zero = 1; break; }
    if (fabs(d) > 0.0) {
      for (int i = k + 1; i < n; ++i) A[i * n + k] = RND(A[i * n + k] / d); /* L column */
      for (int i = k + 1; i < n; ++i)
        for (int j = k + 1; j <= i; ++j) {
          A[i * n + j] = RND(A[i * n + j] - RND(RND(A[i * n + k] * d) * A[j * n + k]));
          A[j * n + i] = A[i * n + j];
        }
    }
  }
  if (rank_zero) {
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    return;
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) y[i] = RND(y[i] - RND(A[i * n + j] * y[j]));
  for (int i = 0; i < n; ++i) y[i] = fabs(A[i * n + i]) > tiny ? RND(y[i] / A[i * n + i]) : 0.0;
  for (int i = n - 1; i >= 0; --i)
    for (int j = i + 1; j < n; ++j) y[i] = RND(y[i] - RND(A[j * n + i] * y[j]));
  for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
#undef RND
}

/* OdometryProvider::rodrigues, OdometryProvider.h:35-71 */
static void rodrigues(const double* src, double* dst) {
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(dst, I, sizeof(I));
  double rx = src[0], ry = src[1], rz = src[2];
  const double theta = sqrt(rx * rx + ry * ry + rz * rz);
  if (theta >= DBL_EPSILON) {
    const double c = cos(theta), s = sin(theta), c1 = 1. - c;
    const double itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; k++) dst[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
  }
}

static void level_K(const orc_odometry* o, int level, double* K) { /* intr(level): types.cuh:115-119 */
  const int div = 1 << level;
  memset(K, 0, 9 * sizeof(double));
  K[0] = (double)(o->fx / div);
  K[4] = (double)(o->fy / div);
  K[2] = (double)(o->cx / div);
  K[5] = (double)(o->cy / div);
  K[8] = 1;
}

/* ---- object ---------------------------------------------------------------------------- */

orc_odometry* orc_odometry_create(int width, int height, float cx, float cy, float fx, float fy, float distThresh,
                                  float angleThresh) {
  orc_odometry* o = (orc_odometry*)calloc(1, sizeof(orc_odometry));
  o->sum_mode = 1;
  o->solve_mode = 1;
  o->fused_rows = 1; /* the tracker object of the product evaluates its rows fused (track.hip kTrackerFma) */
  o->width = width; o->height = height; o->cx = cx; o->cy = cy; o->fx = fx; o->fy = fy;
  o->distThres = distThresh > 0 ? distThresh : 0.10f;                                   /* RGBDOdometry.h:35 */
  o->angleThres = angleThresh > 0 ? angleThresh : (float)sin(20.f * 3.14159254f / 180.f); /* RGBDOdometry.h:36 */
  o->sobelScale = (float)(1.0 / pow(2.0, 3));                                           /* :34-35 */
  o->maxDepthDeltaRGB = 0.07f;                                                          /* :36 */
  o->maxDepthRGB = 6.0f;                                                                /* :37 */
  o->minGrad[0] = 5; o->minGrad[1] = 3; o->minGrad[2] = 1;                              /* :108-110 */
  for (int i = 0; i < NUM_PYRS; ++i) {
    const size_t P = (size_t)(height >> i) * (width >> i);
    o->depth_tmp[i] = (uint16_t*)calloc(P, 2);
    o->vmaps_g_prev[i] = (float*)calloc(3 * P, 4);
    o->nmaps_g_prev[i] = (float*)calloc(3 * P, 4);
    o->vmaps_curr[i] = (float*)calloc(3 * P, 4);
    o->nmaps_curr[i] = (float*)calloc(3 * P, 4);
    o->lastDepth[i] = (float*)calloc(P, 4);
    o->nextDepth[i] = (float*)calloc(P, 4);
    o->lastImage[i] = (uint8_t*)calloc(P, 1);
    o->nextImage[i] = (uint8_t*)calloc(P, 1);
    o->lastNextImage[i] = (uint8_t*)calloc(P, 1);
    o->nextdIdx[i] = (int16_t*)calloc(P, 2);
    o->nextdIdy[i] = (int16_t*)calloc(P, 2);
    o->pointClouds[i] = (float*)calloc(3 * P, 4);
    o->corresImg[i] = (orc_dataterm*)calloc(P, sizeof(orc_dataterm));
  }
  o->vmaps_tmp = (float*)calloc((size_t)width * height * 4, 4);
  o->nmaps_tmp = (float*)calloc((size_t)width * height * 4, 4);
  return o;
}

void orc_odometry_destroy(orc_odometry* o) {
  if (!o) return;
  for (int i = 0; i < NUM_PYRS; ++i) {
    free(o->depth_tmp[i]); free(o->vmaps_g_prev[i]); free(o->nmaps_g_prev[i]); free(o->vmaps_curr[i]); free(o->nmaps_curr[i]);
    free(o->lastDepth[i]); free(o->nextDepth[i]); free(o->lastImage[i]); free(o->nextImage[i]); free(o->lastNextImage[i]);
    free(o->nextdIdx[i]); free(o->nextdIdy[i]); free(o->pointClouds[i]); free(o->corresImg[i]);
  }
  free(o->vmaps_tmp); free(o->nmaps_tmp);
  free(o);
}

void* orc_odometry_buffer(orc_odometry* o, int which, int level) {
  switch (which) {
    case 0: return o->vmaps_curr[level];
    case 1: return o->nmaps_curr[level];
    case 2: return o->vmaps_g_prev[level];
    case 3: return o->nmaps_g_prev[level];
    case 4: return o->lastDepth[level];
    case 5: return o->nextDepth[level];
    case 6: return o->lastImage[level];
    case 7: return o->nextImage[level];
    case 8: return o->lastNextImage[level];
    case 9: return o->nextdIdx[level];
    case 10: return o->nextdIdy[level];
    case 11: return o->pointClouds[level];
    case 12: return o->depth_tmp[level];
    case 13: return o->corresImg[level];
  }
  return NULL;
}

/* RGBDOdometry::initICP(filteredDepth), RGBDOdometry.cpp:118-142 */
void orc_odometry_initICP_depth(orc_odometry* o, const uint16_t* filteredDepth, float depthCutoff) {
  memcpy(o->depth_tmp[0], filteredDepth, (size_t)o->width * o->height * 2);
  for (int i = 1; i < NUM_PYRS; ++i) orc_pyrDown(o->depth_tmp[i - 1], o->height >> (i - 1), o->width >> (i - 1), o->depth_tmp[i]);
  for (int i = 0; i < NUM_PYRS; ++i) {
    const int div = 1 << i;
    orc_createVMap(o->fx / div, o->fy / div, o->cx / div, o->cy / div, o->depth_tmp[i], o->height >> i, o->width >> i,
                   o->vmaps_curr[i], depthCutoff);
    orc_createNMap(o->vmaps_curr[i], o->height >> i, o->width >> i, o->nmaps_curr[i]);
  }
}

/* RGBDOdometry::initICP(predictedVertices, predictedNormals), :144-167 */
void orc_odometry_initICP_maps(orc_odometry* o, const float* verts4, const float* norms4, float depthCutoff) {
  (void)depthCutoff;
  const size_t n = (size_t)o->width * o->height * 4;
  memcpy(o->vmaps_tmp, verts4, n * 4);
  memcpy(o->nmaps_tmp, norms4, n * 4);
  orc_copyMaps(o->vmaps_tmp, o->nmaps_tmp, o->height, o->width, o->vmaps_curr[0], o->nmaps_curr[0]);
  for (int i = 1; i < NUM_PYRS; ++i) {
    orc_resizeMap(o->vmaps_curr[i - 1], o->height >> (i - 1), o->width >> (i - 1), o->vmaps_curr[i], 0);
    orc_resizeMap(o->nmaps_curr[i - 1], o->height >> (i - 1), o->width >> (i - 1), o->nmaps_curr[i], 1);
  }
}

/* RGBDOdometry::initICPModel, :169-207 */
void orc_odometry_initICPModel(orc_odometry* o, const float* verts4, const float* norms4, float depthCutoff,
                               const float* modelPose16) {
  (void)depthCutoff;
  const size_t n = (size_t)o->width * o->height * 4;
  memcpy(o->vmaps_tmp, verts4, n * 4);
  memcpy(o->nmaps_tmp, norms4, n * 4);
  orc_copyMaps(o->vmaps_tmp, o->nmaps_tmp, o->height, o->width, o->vmaps_g_prev[0], o->nmaps_g_prev[0]);
  for (int i = 1; i < NUM_PYRS; ++i) {
    orc_resizeMap(o->vmaps_g_prev[i - 1], o->height >> (i - 1), o->width >> (i - 1), o->vmaps_g_prev[i], 0);
    orc_resizeMap(o->nmaps_g_prev[i - 1], o->height >> (i - 1), o->width >> (i - 1), o->nmaps_g_prev[i], 1);
  }
  float R[9], t[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = modelPose16[i * 4 + j];
    t[i] = modelPose16[i * 4 + 3];
  }
  for (int i = 0; i < NUM_PYRS; ++i)
    orc_tranformMaps(o->vmaps_g_prev[i], o->nmaps_g_prev[i], o->height >> i, o->width >> i, R, t, o->vmaps_g_prev[i],
                     o->nmaps_g_prev[i]);
}

/* RGBDOdometry::populateRGBDData, :209-236 */
static void populate(orc_odometry* o, const uint8_t* rgba, float** destDepths, uint8_t** destImages) {
  orc_verticesToDepth(o->vmaps_tmp, o->height, o->width, destDepths[0], o->maxDepthRGB);
  for (int i = 0; i + 1 < NUM_PYRS; i++) orc_pyrDownGaussF(destDepths[i], o->height >> i, o->width >> i, destDepths[i + 1]);
  orc_imageBGRToIntensity(rgba, o->height, o->width, destImages[0]);
  for (int i = 0; i + 1 < NUM_PYRS; i++) orc_pyrDownUcharGauss(destImages[i], o->height >> i, o->width >> i, destImages[i + 1]);
}
void orc_odometry_initRGBModel(orc_odometry* o, const uint8_t* rgba) { populate(o, rgba, o->lastDepth, o->lastImage); } /* :238-242 */
void orc_odometry_initRGB(orc_odometry* o, const uint8_t* rgba) { populate(o, rgba, o->nextDepth, o->nextImage); }      /* :244-248 */
void orc_odometry_initFirstRGB(orc_odometry* o, const uint8_t* rgba) {                                                  /* :250-266 */
  orc_imageBGRToIntensity(rgba, o->height, o->width, o->lastNextImage[0]);
  for (int i = 0; i + 1 < NUM_PYRS; i++)
    orc_pyrDownUcharGauss(o->lastNextImage[i], o->height >> i, o->width >> i, o->lastNextImage[i + 1]);
}

/* ---- canonical (order-free) reductions: rows of every pixel, then orc_canon_reduce ------------------------------ */
static int canon_icp(const orc_odometry* o, int level, const float* Rcurr, const float* tcurr, const float* Rprev_inv, const float* tprev,
                     float lfx, float lfy, float lcx, float lcy, int* E, float* A, float* b, float* residual) {
  const int rows = o->height >> level, cols = o->width >> level;
  const long n = (long)rows * cols;
  float* rw = (float*)malloc((size_t)n * 7 * sizeof(float));
  unsigned char* fd = (unsigned char*)malloc((size_t)n);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const size_t p = (size_t)y * cols + x;
      fd[p] = (unsigned char)orc_icp_row(Rcurr, tcurr, o->vmaps_curr[level], o->nmaps_curr[level], Rprev_inv, tprev, lfx, lfy, lcx, lcy,
                                         o->vmaps_g_prev[level], o->nmaps_g_prev[level], o->distThres, o->angleThres, rows, cols, x, y,
                                         rw + p * 7);
    }
  float sums[29];
  const int retries = orc_canon_reduce(6, rw, fd, n, E, sums);
  free(rw);
  free(fd);
  int shift = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      const float v = sums[shift++];
      if (j == 6) b[i] = v; else A[j * 6 + i] = A[i * 6 + j] = v;
    }
  residual[0] = sums[27];
  residual[1] = sums[28];
  orc_canon_next_exponents(6, sums, E);
  return retries;
}

static int canon_rgb(const orc_odometry* o, int level, float sigma, float lfx, float lfy, int* E, float* A, float* b) {
  const int rows = o->height >> level, cols = o->width >> level;
  const long n = (long)rows * cols;
  float* rw = (float*)malloc((size_t)n * 7 * sizeof(float));
  unsigned char* fd = (unsigned char*)malloc((size_t)n);
#pragma omp parallel for schedule(static)
  for (long p = 0; p < n; ++p) {
    const orc_dataterm* c = o->corresImg[level] + p;
    orc_rgb_row(c, sigma, o->pointClouds[level], lfx, lfy, o->nextdIdx[level], o->nextdIdy[level], o->sobelScale, cols, rw + p * 7);
    fd[p] = c->valid ? 1 : 0;
  }
  float sums[29];
  const int retries = orc_canon_reduce(6, rw, fd, n, E, sums);
  free(rw);
  free(fd);
  int shift = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      const float v = sums[shift++];
      if (j == 6) b[i] = v; else A[j * 6 + i] = A[i * 6 + j] = v;
    }
  orc_canon_next_exponents(6, sums, E);
  return retries;
}

static int canon_so3(const orc_odometry* o, int level, const float* imageBasis, const float* kinv, const float* krlr, int* E, float* A,
                     float* b, float* residual) {
  const int rows = o->height >> level, cols = o->width >> level;
  const long n = (long)rows * cols;
  float* rw = (float*)malloc((size_t)n * 4 * sizeof(float));
  unsigned char* fd = (unsigned char*)malloc((size_t)n);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const size_t p = (size_t)y * cols + x;
      fd[p] = (unsigned char)orc_so3_row(o->lastNextImage[level], o->nextImage[level], imageBasis, kinv, krlr, rows, cols, x, y, rw + p * 4);
    }
  float sums[11];
  const int retries = orc_canon_reduce(3, rw, fd, n, E, sums);
  free(rw);
  free(fd);
  int shift = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 4; ++j) {
      const float v = sums[shift++];
      if (j == 3) b[i] = v; else A[j * 3 + i] = A[i * 3 + j] = v;
    }
  residual[0] = sums[9];
  residual[1] = sums[10];
  orc_canon_next_exponents(3, sums, E);
  return retries;
}

/* RGBDOdometry::getIncrementalTransformation, :268-605 */
static void track_impl(orc_odometry* o, float* trans, float* rot, int rgbOnly, float icpWeight,
                                               int pyramid, int fastOdom, int so3, int interMap, orc_track_result* res) {
  const int icp = !rgbOnly && icpWeight > 0; /* :278 */
  const int rgb = rgbOnly || icpWeight < 100; /* :279 */
  memset(res, 0, sizeof(*res));

  float Rprev[9], tprev[3], Rcurr[9], tcurr[3];
  memcpy(Rprev, rot, sizeof(Rprev));
  memcpy(tprev, trans, sizeof(tprev));
  memcpy(Rcurr, Rprev, sizeof(Rcurr));
  memcpy(tcurr, tprev, sizeof(tcurr));

  if (rgb) /* :287-293 */
    for (int i = 0; i < NUM_PYRS; i++)
      orc_computeDerivativeImages(o->nextImage[i], o->height >> i, o->width >> i, o->nextdIdx[i], o->nextdIdy[i]);

  double resultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  int E_so3[4], E_icp[7], E_rgb[7], have_E = 0; /* canonical sums: column exponents (orc_canon.c) */

  if (so3) { /* :297-385 */
    const int L = 2;
    float R_lr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double K[9], Kinv[9];
    level_K(o, L, K);
    gj_inverse(K, 3, Kinv);
    float lastError = FLT_MAX / 2, lastCount = FLT_MAX / 2;
    double lastResultR[9];
    memcpy(lastResultR, resultR, sizeof(resultR));
    for (int i = 0; i < 10; i++) {
      double t[9], H[9];
      float imageBasis[9], kinv[9], krlr[9];
      if (o->solve_mode) {
        const orc_kpre k2 = orc_kpre_of(o->fx, o->fy, o->cx, o->cy, L);
        orc_scalar_so3_params(resultR, &k2, imageBasis, kinv, krlr);
      } else {
        matmul_d(K, resultR, 3, t);
        matmul_d(t, Kinv, 3, H);
        for (int k = 0; k < 9; ++k) { imageBasis[k] = (float)H[k]; kinv[k] = (float)Kinv[k]; krlr[k] = (float)t[k]; }
      }
      float jtj[9], jtr[3], residual[2];
      if (o->sum_mode) {
        if (i == 0) {
          orc_canon_static_so3((o->height >> L) * (o->width >> L), E_so3);
          for (int c = 0; c < 4; ++c) E_so3[c] = orc_canon_clamp_e(E_so3[c] + o->exp_bias);
        }
        res->canon_retries += canon_so3(o, L, imageBasis, kinv, krlr, E_so3, jtj, jtr, residual);
      } else
      (o->hook_so3 ? o->hook_so3 : orc_so3Step)(o->lastNextImage[L], o->nextImage[L], imageBasis, kinv, krlr, o->height >> L, o->width >> L, jtj, jtr, residual);
      res->so3_iterations_run++;
      res->lastSO3Error = sqrtf(residual[0]) / residual[1];
      res->lastSO3Count = residual[1];
      if (res->lastSO3Error < lastError && lastCount == res->lastSO3Count) break; /* converged */
      else if (res->lastSO3Error > lastError + 0.001) {                           /* diverging */
        res->lastSO3Error = lastError;
        res->lastSO3Count = lastCount;
        memcpy(resultR, lastResultR, sizeof(resultR));
        break;
      }
      lastError = res->lastSO3Error;
      lastCount = res->lastSO3Count;
      memcpy(lastResultR, resultR, sizeof(resultR));
      if (o->solve_mode) {
        orc_scalar_so3_update(jtj, jtr, R_lr, resultR);
        continue;
      }
      double Ad[9], bd[3], xd[3];
      for (int k = 0; k < 9; ++k) Ad[k] = jtj[k];
      for (int k = 0; k < 3; ++k) bd[k] = jtr[k];
      sym_solve(Ad, bd, 3, 1, xd); /* Eigen::Vector3f delta = jtj.ldlt().solve(jtr) */
      double rotUpdate[9];
      rodrigues(xd, rotUpdate);
      float ru[9], nr[9];
      for (int k = 0; k < 9; ++k) ru[k] = (float)rotUpdate[k];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) nr[r * 3 + c] = ru[r * 3 + 0] * R_lr[0 * 3 + c] + ru[r * 3 + 1] * R_lr[1 * 3 + c] + ru[r * 3 + 2] * R_lr[2 * 3 + c];
      memcpy(R_lr, nr, sizeof(nr));
      for (int k = 0; k < 9; ++k) resultR[k] = R_lr[k];
    }
  }

  int iterations[NUM_PYRS];
  iterations[0] = interMap ? 50 : fastOdom ? 3 : 10; /* :387-389 */
  iterations[1] = interMap ? 50 : pyramid ? 5 : 0;
  iterations[2] = interMap ? 50 : pyramid ? 4 : 0;

  float Rprev_inv[9];
  inv3f(Rprev, Rprev_inv); /* :391 */

  double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (so3)
    for (int x = 0; x < 3; x++)
      for (int y = 0; y < 3; y++) resultRt[x * 4 + y] = resultR[x * 3 + y];

  float residual[2] = {0.f, 0.f}; /* uninitialised in the reference when icp == false (:491); fixed at 0 here */

  for (int i = NUM_PYRS - 1; i >= 0; i--) {
    const int rows = o->height >> i, cols = o->width >> i;
    if (rgb) orc_projectToPointCloud(o->lastDepth[i], rows, cols, o->pointClouds[i], o->fx, o->fy, o->cx, o->cy, i); /* :412 */
    double K[9], Kinv[9];
    level_K(o, i, K);
    gj_inverse(K, 3, Kinv);
    const int div = 1 << i;
    const float lfx = o->fx / div, lfy = o->fy / div, lcx = o->cx / div, lcy = o->cy / div;
    float lastRGBError = FLT_MAX; /* :423 */
    if (o->sum_mode && iterations[i] > 0) { /* first iteration of a level: static guess (first level of the call) or the coarser level's totals x 4 */
      if (!have_E) {
        orc_canon_static_icp(rows * cols, E_icp);
        orc_canon_static_rgb(rows * cols, lfx, rgbOnly, E_rgb);
        for (int c = 0; c < 7; ++c) {
          E_icp[c] = orc_canon_clamp_e(E_icp[c] + o->exp_bias);
          E_rgb[c] = orc_canon_clamp_e(E_rgb[c] + o->exp_bias);
        }
        have_E = 1;
      } else {
        orc_canon_level_step(6, E_icp);
        orc_canon_level_step(6, E_rgb);
      }
    }

    for (int j = 0; j < iterations[i]; j++) {
      double Rt[16], R[9], t[9], H[9];
      gj_inverse(resultRt, 4, Rt);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = Rt[r * 4 + c];
      matmul_d(K, R, 3, t);
      matmul_d(t, Kinv, 3, H);
      float krkInv[9], kt[3];
      for (int k = 0; k < 9; ++k) krkInv[k] = (float)H[k];
      for (int r = 0; r < 3; ++r) kt[r] = (float)(K[r * 3 + 0] * Rt[3] + K[r * 3 + 1] * Rt[7] + K[r * 3 + 2] * Rt[11]);
      if (o->solve_mode) {
        const orc_kpre kl = orc_kpre_of(o->fx, o->fy, o->cx, o->cy, i);
        orc_scalar_gn_params(resultRt, &kl, krkInv, kt);
      }

      int sigma = 0, rgbSize = 0;
      if (rgb)
        ((o->hook_rgbres && !o->sum_mode) ? o->hook_rgbres : orc_computeRgbResidual)((float)(pow(o->minGrad[i], 2.0) / pow(o->sobelScale, 2.0)), o->nextdIdx[i], o->nextdIdy[i],
                               o->lastDepth[i], o->nextDepth[i], o->lastImage[i], o->nextImage[i], o->corresImg[i],
                               o->maxDepthDeltaRGB, kt, krkInv, rows, cols, &sigma, &rgbSize);

      float sigmaVal = sqrt((float)sigma / rgbSize == 0 ? 1 : rgbSize); /* :464, precedence as written */
      if (rgbOnly && sqrt(sigma) / rgbSize > lastRGBError) break;        /* :466-469 */
      lastRGBError = sqrt(sigma) / rgbSize;
      res->lastRGBError = lastRGBError;
      res->lastRGBCount = rgbSize;
      if (rgbOnly) sigmaVal = -1;

      float A_icp[36], b_icp[6], A_rgbd[36], b_rgbd[6];
      memset(A_icp, 0, sizeof(A_icp)); memset(b_icp, 0, sizeof(b_icp));
      memset(A_rgbd, 0, sizeof(A_rgbd)); memset(b_rgbd, 0, sizeof(b_rgbd));
      if (icp && o->sum_mode)
        res->canon_retries += canon_icp(o, i, Rcurr, tcurr, Rprev_inv, tprev, lfx, lfy, lcx, lcy, E_icp, A_icp, b_icp, residual);
      else if (icp)
        (o->hook_icp ? o->hook_icp : orc_icpStep)(Rcurr, tcurr, o->vmaps_curr[i], o->nmaps_curr[i], Rprev_inv, tprev, lfx, lfy, lcx, lcy, o->vmaps_g_prev[i],
                    o->nmaps_g_prev[i], o->distThres, o->angleThres, rows, cols, A_icp, b_icp, residual);
      res->lastICPError = sqrtf(residual[0]) / residual[1];
      res->lastICPCount = residual[1];
      if (rgb && o->sum_mode)
        res->canon_retries += canon_rgb(o, i, sigmaVal, lfx, lfy, E_rgb, A_rgbd, b_rgbd);
      else if (rgb)
        (o->hook_rgb ? o->hook_rgb : orc_rgbStep)(o->corresImg[i], sigmaVal, o->pointClouds[i], lfx, lfy, o->nextdIdx[i], o->nextdIdy[i], o->sobelScale, rows,
                    cols, A_rgbd, b_rgbd);

      double A[36], b[6], x[6];
      if (o->solve_mode) {
        orc_scalar_gn_update(A_icp, b_icp, A_rgbd, b_rgbd, icp, rgb, icpWeight, Rprev, tprev, resultRt, A, b, Rcurr, tcurr);
        memcpy(res->lastA, A, sizeof(A));
        memcpy(res->lastb, b, sizeof(b));
        res->iterations_run[i]++;
        if (res->trace_len < 160) {
          float* tr = res->trace[res->trace_len++];
          for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) tr[r * 4 + c] = Rcurr[r * 3 + c];
            tr[r * 4 + 3] = tcurr[r];
          }
        }
        continue;
      }
      if (icp && rgb) { /* :549-555 */
        const double w = icpWeight;
        for (int k = 0; k < 36; ++k) A[k] = (double)A_rgbd[k] + w * w * (double)A_icp[k];
        for (int k = 0; k < 6; ++k) b[k] = (double)b_rgbd[k] + w * (double)b_icp[k];
      } else if (icp) {
        for (int k = 0; k < 36; ++k) A[k] = A_icp[k];
        for (int k = 0; k < 6; ++k) b[k] = b_icp[k];
      } else {
        for (int k = 0; k < 36; ++k) A[k] = A_rgbd[k];
        for (int k = 0; k < 6; ++k) b[k] = b_rgbd[k];
      }
      memcpy(res->lastA, A, sizeof(A));
      memcpy(res->lastb, b, sizeof(b));
      sym_solve(A, b, 6, 0, x);
      res->iterations_run[i]++;

      /* computeUpdateSE3, OdometryProvider.h:73-93 */
      double upd[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      double Rr[9];
      rodrigues(x + 3, Rr);
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) upd[r * 4 + c] = Rr[r * 3 + c];
        upd[r * 4 + 3] = x[r];
      }
      matmul_d(upd, resultRt, 4, resultRt);

      /* rgbOdom (Isometry3f) = float(resultRt); currentT = [Rprev|tprev] * rgbOdom.inverse(), :573-585 */
      float Ro[9], to[3], RoT[9], ti[3];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Ro[r * 3 + c] = (float)resultRt[r * 4 + c];
        to[r] = (float)resultRt[r * 4 + 3];
      }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) RoT[r * 3 + c] = Ro[c * 3 + r];
      for (int r = 0; r < 3; ++r) ti[r] = -(RoT[r * 3 + 0] * to[0] + RoT[r * 3 + 1] * to[1] + RoT[r * 3 + 2] * to[2]);
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          Rcurr[r * 3 + c] = Rprev[r * 3 + 0] * RoT[0 * 3 + c] + Rprev[r * 3 + 1] * RoT[1 * 3 + c] + Rprev[r * 3 + 2] * RoT[2 * 3 + c];
        tcurr[r] = (Rprev[r * 3 + 0] * ti[0] + Rprev[r * 3 + 1] * ti[1] + Rprev[r * 3 + 2] * ti[2]) + tprev[r];
      }
      if (res->trace_len < 160) {
        float* tr = res->trace[res->trace_len++];
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c) tr[r * 4 + c] = Rcurr[r * 3 + c];
          tr[r * 4 + 3] = tcurr[r];
        }
      }
    }
  }

  { /* :589-593 */
    const float dx = tcurr[0] - tprev[0], dy = tcurr[1] - tprev[1], dz = tcurr[2] - tprev[2];
    if (rgb && sqrtf(dx * dx + dy * dy + dz * dz) > 0.3) {
      memcpy(Rcurr, Rprev, sizeof(Rcurr));
      memcpy(tcurr, tprev, sizeof(tcurr));
      res->rejected_jump = 1;
    }
  }
  if (so3) /* :595-601 */
    for (int i = 0; i < NUM_PYRS; i++) {
      uint8_t* t = o->lastNextImage[i];
      o->lastNextImage[i] = o->nextImage[i];
      o->nextImage[i] = t;
    }
  memcpy(trans, tcurr, sizeof(tcurr));
  memcpy(rot, Rcurr, sizeof(Rcurr));
  memcpy(res->trans, tcurr, sizeof(tcurr));
  memcpy(res->rot, Rcurr, sizeof(Rcurr));
}

void orc_odometry_set_fused_rows(orc_odometry* o, int on) { o->fused_rows = on ? 1 : 0; }
void orc_odometry_set_sum_mode(orc_odometry* o, int mode) { o->sum_mode = mode ? 1 : 0; }
void orc_odometry_set_step_hooks(orc_odometry* o, void* so3, void* rgbres, void* icp, void* rgb) {
  o->hook_so3 = (void (*)(const uint8_t*, const uint8_t*, const float*, const float*, const float*, int, int, float*, float*, float*))so3;
  o->hook_rgbres = (void (*)(float, const int16_t*, const int16_t*, const float*, const float*, const uint8_t*, const uint8_t*, orc_dataterm*,
                             float, const float*, const float*, int, int, int*, int*))rgbres;
  o->hook_icp = (void (*)(const float*, const float*, const float*, const float*, const float*, const float*, float, float, float, float,
                          const float*, const float*, float, float, int, int, float*, float*, float*))icp;
  o->hook_rgb = (void (*)(const orc_dataterm*, float, const float*, float, float, const int16_t*, const int16_t*, float, int, int, float*,
                          float*))rgb;
}
void orc_odometry_set_solve_mode(orc_odometry* o, int mode) { o->solve_mode = mode ? 1 : 0; }
void orc_odometry_set_exp_bias(orc_odometry* o, int bias) { o->exp_bias = bias; }

void orc_odometry_getIncrementalTransformation(orc_odometry* o, float* trans, float* rot, int rgbOnly, float icpWeight,
                                               int pyramid, int fastOdom, int so3, int interMap, orc_track_result* res) {
  const int prev = orc_get_fused_rows();
  orc_set_fused_rows(o->fused_rows);
  track_impl(o, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3, interMap, res);
  orc_set_fused_rows(prev);
}
