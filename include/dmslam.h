/*
 * dmslam.h — C ABI of the MI355X-native dense tracking + surfel-fusion back end.
 *
 * This is the drop-in boundary for the hot path of robotvisionmu/DenseMonoSLAM
 * (elasticfusion/Core).  Two layers are exported from libdmslam_hip.so:
 *
 *   (B) operator layer  — one entry point per free function of the reference's
 *       Core/src/Cuda/cudafuncs.cuh:70-168 (same names, same argument meaning),
 *       plus one per GLSL program of the fusion half.
 *   (A) object layer    — opaque handles mirroring the reference classes
 *       RGBDOdometry (Core/src/Utils/RGBDOdometry.h:32-153),
 *       IndexMap (Core/src/IndexMap.h:33-205),
 *       GlobalModel (Core/src/GlobalModel.h:43-141) and the per-camera part of
 *       ElasticFusion::processFrame (Core/src/ElasticFusion.cpp:99-637).
 *
 * Conventions
 *   - plain C, no torch / Eigen / HIP types in any signature; `dms_stream` is a
 *     hipStream_t passed as void* (NULL = default stream).
 *   - every device pointer is caller-owned HBM unless the function name says
 *     `_create`; the callee never allocates persistent memory behind the
 *     caller's back (reference ownership model: DeviceArray passed by reference,
 *     RGBDOdometry.cpp:44-49).
 *   - every function returns an int status: 0 = DMS_OK, negative = error.  No
 *     exceptions, no exit() (the reference's cudaSafeCall prints and exit(0)s,
 *     Cuda/convenience.cuh:64-71).
 *   - re-entrant per (device, stream): no global state, no singletons.
 *   - matrices are row-major float (reference `mat33` = 3×float3 rows,
 *     Cuda/types.cuh:61-75); poses are 4×4 row-major float, camera-to-world.
 *   - vertex / normal maps use the reference layout: three planes (x, y, z)
 *     stacked along rows in one pitched image of (3*rows) × cols floats
 *     (RGBDOdometry.cpp:97-101).  An invalid element has NaN in plane x only
 *     (Cuda/cudafuncs.cu:125,157,181).
 */
#ifndef DMSLAM_H_
#define DMSLAM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMS_OK 0
#define DMS_ERR_INVALID_ARG (-1)
#define DMS_ERR_HIP (-2)
#define DMS_ERR_WORKSPACE (-3)
#define DMS_ERR_CAPACITY (-4)
#define DMS_ERR_STATE (-5)
#define DMS_ERR_TIMEOUT (-6) /* a resident tracker kernel gave up at a grid barrier: result invalid */
#ifndef DMS_ERR_UNSUPPORTED
#define DMS_ERR_UNSUPPORTED (-7) /* a dependency resolved at run time (zlib, RCCL) is missing, or an input flavour is not handled */
#endif
#define DMS_ERR_COMM (-9) /* an RCCL call failed (dmslam_collab.h) */

#define DMS_NUM_PYRS 3         /* RGBDOdometry.h: NUM_PYRS */
#define DMS_MAX_SENSORS 8      /* reference Vertex::MAX_SENSORS = 3 (Shaders/Vertex.cpp:49); 8 = one per GPU of the node */
#define DMS_REF_MAX_SENSORS 3  /* layout used by dms_model_download_ref (60-byte reference surfel) */
#define DMS_MAX_PARTIAL_BLOCKS 1024 /* reference MAX_THREADS partial slots (cudafuncs.cuh:56-60) */

typedef void* dms_stream; /* hipStream_t */

/* Pitched 2-D device image view (reference PtrStepSz<T>, containers/kernel_containers.hpp:49-93). */
typedef struct dms_image2d {
  void* data;
  size_t pitch; /* bytes between rows */
  int rows;
  int cols;
} dms_image2d;

typedef struct dms_float3 {
  float x, y, z;
} dms_float3;

/* row-major 3×3 (reference mat33, Cuda/types.cuh:61-75) */
typedef struct dms_mat33 {
  float m[9];
} dms_mat33;

/* reference CameraModel (Cuda/types.cuh:103-121); level l divides all four by 2^l */
typedef struct dms_camera {
  float fx, fy, cx, cy;
} dms_camera;

/* reference DataTerm (Cuda/types.cuh:77-83), 16 bytes */
typedef struct dms_dataterm {
  short zero_x, zero_y;
  short one_x, one_y;
  float diff;
  int valid; /* reference: bool + 3 bytes padding */
} dms_dataterm;

/* ------------------------------------------------------------------------- */
/* library info                                                               */
/* ------------------------------------------------------------------------- */
const char* dms_version(void);
const char* dms_last_error(void); /* thread-local text of the last failure */
int dms_device_count(int* count);
int dms_set_device(int device);

/* Device self-test of csrc/exact_arith.hpp: the map kernels' short correctly-rounded quotient / reciprocal / square-root sequences
 * against the compiler's IEEE sequences, operand by operand on the current device (default stream, synchronous).
 *   what 0: square root of every finite float >= 2^-96;  1: reciprocal of every float in [1, 4);
 *        2: a / d for a = 0 and every finite |a| >= 2^-96 whose quotient is finite and >= 2^-96 in size, for the divisor d (a camera constant)
 * mismatches: operands whose result bits differ (0 is the only acceptable answer); first_bad: the largest such operand's bits. */
int dms_exact_arith_selftest(int what, float d, unsigned long long* mismatches, unsigned* first_bad);

/* bytes of device workspace the reduction operators need (partials + result) */
size_t dms_reduce_workspace_bytes(void);

/* Simple owned device buffers for hosts without their own allocator
 * (reference DeviceArray<T>::create / release, containers/device_memory.cpp:108-143). */
int dms_device_alloc(void** ptr, size_t bytes);
int dms_device_free(void* ptr);
int dms_memcpy_h2d(void* dst, const void* src, size_t bytes, dms_stream s);
int dms_memcpy_d2h(void* dst, const void* src, size_t bytes, dms_stream s);
int dms_memcpy_d2d_async(void* dst, const void* src, size_t bytes, dms_stream s); /* stream ordered, no synchronisation */
int dms_memset(void* dst, int value, size_t bytes, dms_stream s);
/* `rows` rows of `width` bytes, pitched source to pitched destination, one launch, stream ordered; everything a multiple of 4 bytes.
 * The destination may be host memory the device can write (hipHostMalloc): how small per-tick metadata reaches the host. */
int dms_copy_rows_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, dms_stream s);
/* free / total HBM of the current device (hipMemGetInfo): sizing maps against the 288 GB, leak checks */
int dms_mem_info(size_t* free_bytes, size_t* total_bytes);
int dms_stream_sync(dms_stream s);
/* convenience for callers without a HIP runtime of their own (e.g. ctypes): a non-blocking stream */
int dms_stream_create(dms_stream* out);
int dms_stream_destroy(dms_stream s);

/* ------------------------------------------------------------------------- */
/* (B) operator layer — tracking (reference Cuda/cudafuncs.cuh)               */
/* ------------------------------------------------------------------------- */

/* reference icpStep (cudafuncs.cuh:70-79, reduce.cu:367-428).
 * A: 36 floats row-major symmetric, b: 6 floats, residual: {sum r^2, inliers}.
 * threads/blocks = 0 selects the CDNA4 launch shape.  Synchronous (like the reference). */
int dms_icpStep(const dms_mat33* Rcurr, const dms_float3* tcurr,
                const dms_image2d* vmap_curr, const dms_image2d* nmap_curr,
                const dms_mat33* Rprev_inv, const dms_float3* tprev,
                const dms_camera* intr, const dms_image2d* vmap_g_prev,
                const dms_image2d* nmap_g_prev, float distThres, float angleThres,
                void* workspace, size_t workspace_bytes, float* matrixA_host,
                float* vectorB_host, float* residual_host, int threads, int blocks,
                dms_stream stream);

/* reference rgbStep (cudafuncs.cuh:87-92, reduce.cu:643-685) */
int dms_rgbStep(const dms_image2d* corresImg, float sigma, const dms_image2d* cloud,
                float fx, float fy, const dms_image2d* dIdx, const dms_image2d* dIdy,
                float sobelScale, void* workspace, size_t workspace_bytes,
                float* matrixA_host, float* vectorB_host, int threads, int blocks,
                dms_stream stream);

/* reference so3Step (cudafuncs.cuh:94-100, reduce.cu:1054-1103); A is 3×3, b is 3 */
int dms_so3Step(const dms_image2d* lastImage, const dms_image2d* nextImage,
                const dms_mat33* imageBasis, const dms_mat33* kinv,
                const dms_mat33* krlr, void* workspace, size_t workspace_bytes,
                float* matrixA_host, float* vectorB_host, float* residual_host,
                int threads, int blocks, dms_stream stream);

/* reference computeRgbResidual (cudafuncs.cuh:102-114, reduce.cu:865-925) */
int dms_computeRgbResidual(float minScale, const dms_image2d* dIdx,
                           const dms_image2d* dIdy, const dms_image2d* lastDepth,
                           const dms_image2d* nextDepth, const dms_image2d* lastImage,
                           const dms_image2d* nextImage, dms_image2d* corresImg,
                           void* workspace, size_t workspace_bytes,
                           float maxDepthDelta, const dms_float3* kt,
                           const dms_mat33* krkinv, int* sigmaSum, int* count,
                           int threads, int blocks, dms_stream stream);

/* reference createVMap / createNMap (cudafuncs.cuh:116-122, cudafuncs.cu:106-198) */
int dms_createVMap(const dms_camera* intr, const dms_image2d* depth, dms_image2d* vmap,
                   float depthCutoff, dms_stream stream);
int dms_createNMap(const dms_image2d* vmap, dms_image2d* nmap, dms_stream stream);

/* reference tranformMaps, both overloads (cudafuncs.cuh:124-130, cudafuncs.cu:200-311) */
int dms_tranformMaps(const dms_image2d* vmap_src, const dms_image2d* nmap_src,
                     const dms_mat33* Rmat, const dms_float3* tvec,
                     dms_image2d* vmap_dst, dms_image2d* nmap_dst, dms_stream stream);
int dms_tranformVMap(const dms_image2d* vmap_src, const dms_mat33* Rmat,
                     const dms_float3* tvec, dms_image2d* vmap_dst, dms_stream stream);

/* reference copyMaps, both overloads (cudafuncs.cuh:132-138, cudafuncs.cu:313-414):
 * dense RGBA32F (4 floats / pixel, rows*cols*4 floats) -> stacked planes */
int dms_copyMaps(const float* vmap_src, const float* nmap_src, dms_image2d* vmap_dst,
                 dms_image2d* nmap_dst, dms_stream stream);
int dms_copyVMap(const float* vmap_src, dms_image2d* vmap_dst, dms_stream stream);

/* reference resizeVMap / resizeNMap (cudafuncs.cuh:140-144, cudafuncs.cu:445-521) */
int dms_resizeVMap(const dms_image2d* input, dms_image2d* output, dms_stream stream);
int dms_resizeNMap(const dms_image2d* input, dms_image2d* output, dms_stream stream);

/* reference imageBGRToIntensity (cudafuncs.cuh:146-147, cudafuncs.cu:643-669).
 * `rgba` replaces the cudaArray: dense-or-pitched RGBA8 image (4 bytes / pixel). */
int dms_imageBGRToIntensity(const dms_image2d* rgba, dms_image2d* dst, dms_stream stream);

/* reference verticesToDepth, both overloads (cudafuncs.cuh:149-153, cudafuncs.cu:597-639) */
int dms_verticesToDepth(const float* vmap_src_rgba32f, dms_image2d* dst, float cutOff,
                        dms_stream stream);
int dms_verticesToDepth2D(const dms_image2d* vmap_src, dms_image2d* dst, float cutOff,
                          dms_stream stream);

/* reference projectToPointCloud (cudafuncs.cuh:156-158, cudafuncs.cu:727-757); cloud is float3 (12 B) */
int dms_projectToPointCloud(const dms_image2d* depth, dms_image2d* cloud,
                            const dms_camera* intrinsics, int level, dms_stream stream);

/* reference pyrDown / pyrDownGaussF / pyrDownUcharGauss (cudafuncs.cuh:160-167, cudafuncs.cu:57-104,416-443,523-595) */
int dms_pyrDown(const dms_image2d* src, dms_image2d* dst, dms_stream stream);
int dms_pyrDownGaussF(const dms_image2d* src, dms_image2d* dst, dms_stream stream);
int dms_pyrDownUcharGauss(const dms_image2d* src, dms_image2d* dst, dms_stream stream);

/* reference computeDerivativeImages (cudafuncs.cuh:169-171, cudafuncs.cu:674-725) */
int dms_computeDerivativeImages(const dms_image2d* src, dms_image2d* dx, dms_image2d* dy,
                                dms_stream stream);

/* Normalised information distance of the NID key-framing gate (ElasticFusion::fuseFrame,
 * ElasticFusion.cpp:639-677): reference computeNIDImg / computeNIDDepth (cudafuncs.cuh:170-181,
 * cudafuncs.cu:1513-1650, 1794-1916).  Key-frame value per pixel = the nearer of the active and the
 * "old" prediction (NaN depth = no prediction); joint histogram against the live frame (intensity:
 * bins of 256 / num_bins grey levels; depth: bins of int(max_depth / num_bins) millimetres, depths in
 * metres, max_depth in millimetres); nid = (H_joint - MI) / H_joint.  `workspace` holds the histogram
 * (dms_nid_workspace_bytes(num_bins)); the score is written to *nid_host (the call synchronises,
 * like the reference). */
size_t dms_nid_workspace_bytes(int num_bins);
int dms_computeNIDImg(const dms_image2d* img_kf, const dms_image2d* img_kf_old, const dms_image2d* dmap_kf,
                      const dms_image2d* dmap_kf_old, const dms_image2d* img_curr, int num_bins,
                      void* workspace, size_t workspace_bytes, float* nid_host, dms_stream stream);
int dms_computeNIDDepth(const dms_image2d* dmap_kf, const dms_image2d* dmap_kf_old, const dms_image2d* dmap_curr,
                        int num_bins, float max_depth, void* workspace, size_t workspace_bytes,
                        float* nid_host, dms_stream stream);

/* ------------------------------------------------------------------------- */
/* (A) object layer — RGBDOdometry                                            */
/* ------------------------------------------------------------------------- */
typedef struct dms_odometry dms_odometry;

/* reference RGBDOdometry::RGBDOdometry (RGBDOdometry.cpp:21-111).  distThresh <= 0 and
 * angleThresh <= 0 select the reference defaults 0.10 m / sin(20 deg) (RGBDOdometry.h:35-36). */
int dms_odometry_create(dms_odometry** out, int width, int height, float cx, float cy,
                        float fx, float fy, float distThresh, float angleThresh);
int dms_odometry_destroy(dms_odometry* o);
/* Execution switches of one tracker (no reference counterpart).  They are read from the environment ONCE, when the
 * handle is created - DMS_TRACK_MODE=launches, DMS_TRACK_EARLY_EXIT=0|1, DMS_TRACK_FUSE=0|1 - and changed only here (-1 = keep):
 *   resident       1 = one resident kernel per pyramid level (default; ignored where the device cannot hold a resident kernel's blocks
 *                  at once), 0 = three launches per iteration
 *   early_exit     1 / 0 = force the resident-kernel variant that leaves a level after an iteration without any correspondence on / off
 *                  (the handle's default - on for the frame step's model-to-model tracker - or what DMS_TRACK_EARLY_EXIT chose)
 *   coarse_launch  1 = SO3 pre-alignment, level 2 and level 1 as stages of ONE resident launch where their shapes fit one grid
 *                  (round 6; same bits, slower on the MI355X than a launch per stage: default 0)
 * Every cross-pixel sum of the tracker is the order-free integer sum of csrc/canon.hpp in every execution mode: poses do not depend on
 * these switches, on the grid size or on the run. */
int dms_odometry_set_exec(dms_odometry* o, int resident, int early_exit, int coarse_launch);
/* Several trackers AT THE SAME TIME on one device (cameras on streams of their own): a resident launch needs all its blocks on the device
 * at once, so launches of different streams are chained one behind the other by default.  max_blocks > 0 caps this handle's resident
 * grids (120 of 256 compute units: level 0 runs 5 pixels per thread instead of 3); unchained != 0 is the owner's word that ALL handles that
 * may track at the same time carry caps whose sum fits the device - their launches then skip the chain and overlap.  0, 0 = the default.
 * Same bits whatever the grid (the sums are order-free).  dms_session sets this for its cameras by itself. */
int dms_odometry_set_resident_budget(dms_odometry* o, int max_blocks, int unchained);
/* DEPRECATED since round 6 (kept for callers built against rounds 1-5): fp64_sums and atomic_reduce have been ignored since round 3;
 * equals dms_odometry_set_exec(o, resident, early_exit, -1). */
int dms_odometry_set_mode(dms_odometry* o, int resident, int fp64_sums, int early_exit, int atomic_reduce);
/* launch-per-phase kernels from now on (what the frame step does after a resident kernel timed out); nothing else changes */
int dms_odometry_fall_back_to_launches(dms_odometry* o);
/* Residency.  The blocks of a resident kernel wait for each other, so all of them must be on the device at once: a grid never
 * has more blocks than the device has compute units (hipDeviceProp_t::multiProcessorCount, checked against the occupancy API
 * when the handle is created; DMS_PERSIST_MAX_BLOCKS lowers the limit); a pyramid level that does not fit with <= 4 pixels
 * per thread runs launch-per-phase.  If a resident kernel nevertheless times out at a grid-wide wait (another process holds
 * compute units), the handle switches to launch-per-phase for good (`fell_back`); dms_odometry_getIncrementalTransformation
 * repeats that call at once — same bits either way. */
int dms_odometry_get_mode(dms_odometry* o, int* resident, int* max_resident_blocks, int* fell_back);
/* Fault injection for tests: the next `calls` tracking calls behave as if a resident kernel had timed out at a
 * grid barrier (DMS_ERR_TIMEOUT from dms_odometry_fetch_result; the frame step keeps the prior pose and fuses nothing). */
int dms_odometry_inject_timeout(dms_odometry* o, int calls);
/* Test hooks by name.  "exp_bias": added to the static column exponents of a call's first reductions (csrc/canon.hpp); a
 * negative value makes their diagonal totals overflow the grid, so those reductions are repeated on a coarser one. */
int dms_odometry_debug_set(dms_odometry* o, const char* key, int value);
/* Number of reductions of the last fetched call that were repeated on a coarser grid (csrc/canon.hpp). */
int dms_odometry_canon_retries(dms_odometry* o, int* retries);
/* The tracker's scalar section (csrc/gn_scalar.hpp: 6x6 solve, exp map, pose update, projection parameters; SO3 update)
 * compiled for the HOST — the same source the kernels run, no GPU needed.  Test hooks: the CPU test suite checks that this
 * operation sequence and the oracle's restatement of it give the same bits.
 *   gn: sums_icp / sums_rgb = the 29 floats of a reduction (either may be null), resultRt (4x4 row-major, in / out),
 *       out: A (36), b (6), Rcurr (9), tcurr (3), krkinv (9), kt (3) for the camera matrix of pyramid level `next_level`
 *   so3: sums = the 11 floats of an SO3 reduction, R_lr (3x3 float) and resultR (3x3 double) in / out,
 *       out: imageBasis, kinv, krlr (9 each) of the next iteration at pyramid level 2 */
int dms_debug_scalar_gn(const float* sums_icp, const float* sums_rgb, float icpWeight, const float* Rprev, const float* tprev,
                        double* resultRt, float fx, float fy, float cx, float cy, int next_level, double* A, double* b,
                        float* Rcurr, float* tcurr, float* krkinv, float* kt);
int dms_debug_scalar_so3(const float* sums, float* R_lr, double* resultR, float fx, float fy, float cx, float cy,
                         float* imageBasis, float* kinv, float* krlr);

/* reference initICP(GPUTexture* filteredDepth, ...) (RGBDOdometry.cpp:118-142); depth = dense u16 mm */
int dms_odometry_initICP_depth(dms_odometry* o, const dms_image2d* filteredDepth_u16,
                               float depthCutoff, dms_stream s);
/* reference initICP(predictedVertices, predictedNormals, ...) (RGBDOdometry.cpp:144-167); RGBA32F dense */
int dms_odometry_initICP_maps(dms_odometry* o, const float* predictedVertices,
                              const float* predictedNormals, float depthCutoff, dms_stream s);
/* reference initICPModel (RGBDOdometry.cpp:169-207); modelPose 4×4 row-major */
int dms_odometry_initICPModel(dms_odometry* o, const float* predictedVertices,
                              const float* predictedNormals, float depthCutoff,
                              const float* modelPose, dms_stream s);
/* reference initRGB / initRGBModel / initFirstRGB (RGBDOdometry.cpp:238-266); rgba = RGBA8 */
int dms_odometry_initRGB(dms_odometry* o, const dms_image2d* rgba, dms_stream s);
int dms_odometry_initRGBModel(dms_odometry* o, const dms_image2d* rgba, dms_stream s);
int dms_odometry_initFirstRGB(dms_odometry* o, const dms_image2d* rgba, dms_stream s);
/* Frame-step form of initICPModel + initRGBModel (ElasticFusion.cpp:172-189) in four launches: the
 * source (predicted maps A or fill-in maps B, dense W x H RGBA32F / RGBA8 in HBM) is chosen by a
 * device flag (*use_b_dev != 0, or force_b_image for the image alone) and the model pose is read from a
 * row-major 4x4 in HBM, so neither decision visits the host.  Writes vmaps_g_prev / nmaps_g_prev /
 * lastDepth / lastImage exactly as the two reference calls do. */
int dms_odometry_initModelFused(dms_odometry* o, const void* vertA, const void* normA, const void* rgbaA,
                                const void* vertB, const void* normB, const void* rgbaB, const int* use_b_dev,
                                int force_b_image, const float* modelPose16_dev, dms_stream s);

/* Side outputs of getIncrementalTransformation (RGBDOdometry.h:64-72) */
typedef struct dms_track_result {
  float trans[3];
  float rot[9]; /* row-major */
  float lastICPError, lastICPCount;
  float lastRGBError, lastRGBCount;
  float lastSO3Error, lastSO3Count;
  double lastA[36]; /* row-major */
  double lastb[6];
  int iterations_run[DMS_NUM_PYRS];
  int so3_iterations_run;
  int rejected_jump; /* 1 when the 0.3 m jump gate restored the prior pose (RGBDOdometry.cpp:589-593) */
} dms_track_result;

/* reference getIncrementalTransformation (RGBDOdometry.cpp:268-605).
 * trans[3] / rot[9] are in/out.  Device-resident Gauss-Newton: one host sync at the end. */
int dms_odometry_getIncrementalTransformation(dms_odometry* o, float* trans, float* rot,
                                              int rgbOnly, float icpWeight, int pyramid,
                                              int fastOdom, int so3, int interMap,
                                              dms_track_result* result, dms_stream s);
/* Asynchronous form: enqueues the whole loop on `s`; the result lands in the
 * handle's device result block; fetch with dms_odometry_fetch_result (which syncs). */
int dms_odometry_track_async(dms_odometry* o, const float* trans, const float* rot,
                             int rgbOnly, float icpWeight, int pyramid, int fastOdom,
                             int so3, int interMap, dms_stream s);
int dms_odometry_fetch_result(dms_odometry* o, dms_track_result* result, dms_stream s);
/* reference getCovariance (RGBDOdometry.cpp:607-610): inverse of lastA, 6×6 row-major double */
int dms_odometry_getCovariance(dms_odometry* o, double* cov36);

/* pyramid accessors used by tests and by the NID stage (RGBDOdometry.h:74-86):
 * which: 0 vmaps_curr 1 nmaps_curr 2 vmaps_g_prev 3 nmaps_g_prev 4 lastDepth 5 nextDepth
 *        6 lastImage 7 nextImage 8 lastNextImage 9 nextdIdx 10 nextdIdy 11 pointClouds
 *        12 depth_tmp 13 corresImg (the fused tracker keeps 8-byte records there:
 *        short zero_x, zero_y, diff, valid) 14 nextGate (pose-independent photometric gate, u8) */
int dms_odometry_get_buffer(dms_odometry* o, int which, int level, dms_image2d* view);

/* per-kernel device time of the last tracking call, measured with HIP events on the
 * caller's stream (kernel name -> accumulated ms, launches).  names: "gn_pass1", "gn_pass2",
 * "gn_solve", "so3_pass", "so3_solve".  Only filled when profiling was enabled.  enabled = 2: only "gn_level0" is bracketed
 * (one event pair per tracking call, no in-kernel phase clocks): the dominant kernel's duration in an otherwise unperturbed run. */
int dms_odometry_set_profiling(dms_odometry* o, int enabled);
int dms_odometry_get_kernel_time(dms_odometry* o, const char* name, double* total_ms,
                                 int* launches);

#ifdef __cplusplus
}
#endif

#endif /* DMSLAM_H_ */
