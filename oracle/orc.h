/*
 * orc.h — interface of the CPU ORACLE (test infrastructure; see the header of orc_track.c).
 * Dense row-major host arrays everywhere; vertex/normal maps are 3 stacked planes.
 */
#ifndef ORC_H_
#define ORC_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference DataTerm, Cuda/types.cuh:77-83 (bool widened to int) */
typedef struct orc_dataterm {
  short zero_x, zero_y, one_x, one_y;
  float diff;
  int valid;
} orc_dataterm;

float orc_qnan(void);

/* ---- cudafuncs.cu restatements ---- */
void orc_pyrDown(const uint16_t* src, int srows, int scols, uint16_t* dst);
void orc_createVMap(float fx, float fy, float cx, float cy, const uint16_t* depth, int rows, int cols, float* vmap,
                    float depthCutoff);
void orc_createNMap(const float* vmap, int rows, int cols, float* nmap);
void orc_tranformMaps(const float* vsrc, const float* nsrc, int rows, int cols, const float* R, const float* t, float* vdst,
                      float* ndst);
void orc_copyMaps(const float* vsrc4, const float* nsrc4, int rows, int cols, float* vdst, float* ndst);
void orc_resizeMap(const float* in, int srows, int scols, float* out, int normalize);
void orc_pyrDownGaussF(const float* src, int srows, int scols, float* dst);
void orc_pyrDownUcharGauss(const uint8_t* src, int srows, int scols, uint8_t* dst);
void orc_verticesToDepth(const float* vsrc4, int rows, int cols, float* dst, float cutOff);
void orc_imageBGRToIntensity(const uint8_t* rgba, int rows, int cols, uint8_t* dst);
void orc_computeDerivativeImages(const uint8_t* src, int rows, int cols, int16_t* dx, int16_t* dy);
void orc_projectToPointCloud(const float* depth, int rows, int cols, float* cloud3, float fx, float fy, float cx, float cy,
                             int level);

/* ---- reduce.cu restatements ---- */
/* rows of the tracker with every multiply-add chain fused in source order (1) or every operation rounded (0, default):
 * see orc_track.c.  Process-wide switch; orc_odometry sets it around a call from its own `fused_rows` flag. */
void orc_set_fused_rows(int on);
int orc_get_fused_rows(void);
int orc_icp_row(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr, const float* Rprev_inv,
                const float* tprev, float fx, float fy, float cx, float cy, const float* vmap_g_prev, const float* nmap_g_prev,
                float distThres, float angleThres, int rows, int cols, int x, int y, float* row);
void orc_icpStep(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr, const float* Rprev_inv,
                 const float* tprev, float fx, float fy, float cx, float cy, const float* vmap_g_prev, const float* nmap_g_prev,
                 float distThres, float angleThres, int rows, int cols, float* A, float* b, float* residual);
void orc_computeRgbResidual(float minScale, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth,
                            const float* nextDepth, const uint8_t* lastImage, const uint8_t* nextImage, orc_dataterm* corres,
                            float maxDepthDelta, const float* kt, const float* krkinv, int rows, int cols, int* sigmaSum,
                            int* count);
void orc_rgb_row(const orc_dataterm* c, float sigma, const float* cloud3, float fx, float fy, const int16_t* dIdx,
                 const int16_t* dIdy, float sobelScale, int cols, float* row);
void orc_rgbStep(const orc_dataterm* corres, float sigma, const float* cloud3, float fx, float fy, const int16_t* dIdx,
                 const int16_t* dIdy, float sobelScale, int rows, int cols, float* A, float* b);
int orc_so3_row(const uint8_t* lastImage, const uint8_t* nextImage, const float* imageBasis, const float* kinv, const float* krlr,
                int rows, int cols, int x, int y, float* row);
void orc_so3Step(const uint8_t* lastImage, const uint8_t* nextImage, const float* imageBasis, const float* kinv, const float* krlr,
                 int rows, int cols, float* A, float* b, float* residual);

/* ---- RGBDOdometry restatement (orc_odometry.c) ---- */
typedef struct orc_odometry orc_odometry;

typedef struct orc_track_result {
  float trans[3];
  float rot[9];
  float lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
  double lastA[36];
  double lastb[6];
  int iterations_run[3];
  int so3_iterations_run;
  int rejected_jump;
  int canon_retries; /* canonical sums: reductions repeated with a coarser grid (orc_canon.c) */
  /* per-iteration pose trace (row-major 3x4 [R|t]) for up to 160 GN iterations */
  int trace_len;
  float trace[160][12];
} orc_track_result;

orc_odometry* orc_odometry_create(int width, int height, float cx, float cy, float fx, float fy, float distThresh,
                                  float angleThresh);
void orc_odometry_destroy(orc_odometry* o);
void orc_odometry_initICP_depth(orc_odometry* o, const uint16_t* filteredDepth, float depthCutoff);
void orc_odometry_initICP_maps(orc_odometry* o, const float* verts4, const float* norms4, float depthCutoff);
void orc_odometry_initICPModel(orc_odometry* o, const float* verts4, const float* norms4, float depthCutoff,
                               const float* modelPose16);
void orc_odometry_initRGB(orc_odometry* o, const uint8_t* rgba);
void orc_odometry_initRGBModel(orc_odometry* o, const uint8_t* rgba);
void orc_odometry_initFirstRGB(orc_odometry* o, const uint8_t* rgba);
void orc_odometry_set_fused_rows(orc_odometry* o, int on);
/* cross-pixel sums of the tracker object: 1 (default) = canonical order-free sums (orc_canon.c), 0 = fp64 accumulation in
 * loop / thread order (the order-dependent form, kept as the control of what summation order alone does to a pose) */
void orc_odometry_set_sum_mode(orc_odometry* o, int mode);
/* sum_mode 0 only: replace the four step restatements by functions of the same signatures (NULL = own) */
void orc_odometry_set_step_hooks(orc_odometry* o, void* so3, void* rgbres, void* icp, void* rgb);
/* scalar section between the reductions: 1 (default) = the product's canonical operation order (orc_scalar.c), 0 = the
 * independent Eigen-like restatement (pivoted LDLT in outer-product form, Rodrigues through libm, general inverses) */
void orc_odometry_set_solve_mode(orc_odometry* o, int mode);
/* canonical sums (orc_canon.c) */
void orc_odometry_set_exp_bias(orc_odometry* o, int bias); /* test hook: static exponents of a call's first reductions + bias */
int orc_canon_exp_of(float d);
int orc_canon_clamp_e(int e);
void orc_canon_next_exponents(int n, const float* sums, int* E);
void orc_canon_level_step(int n, int* E);
void orc_canon_static_icp(int npix, int* E);
void orc_canon_static_rgb(int npix, float fx_level, int rgbOnly, int* E);
void orc_canon_static_so3(int npix, int* E);
int orc_canon_reduce(int n, const float* rows, const unsigned char* found, long npix, int* E, float* sums);
void orc_odometry_getIncrementalTransformation(orc_odometry* o, float* trans, float* rot, int rgbOnly, float icpWeight,
                                               int pyramid, int fastOdom, int so3, int interMap, orc_track_result* result);
/* which: same numbering as dms_odometry_get_buffer; returns pointer to the dense host buffer */
void* orc_odometry_buffer(orc_odometry* o, int which, int level);
void orc_covariance(const double* lastA36, double* cov36);
void orc_set_threads(int n);
int orc_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
