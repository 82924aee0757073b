// extern "C" surface of libdmslam_hip.so (include/dmslam.h): argument checks, error text,
// memory helpers and the operator-layer wrappers.  No exceptions cross this boundary.
#include <stdarg.h>
#include <stdio.h>

#include "internal.hpp"

namespace dms {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
  return DMS_ERR_HIP;
}

}  // namespace dms

using namespace dms;

#define S(x) ((hipStream_t)(x))

extern "C" {

const char* dms_version(void) { return "densemonoslam_amd 0.1 (gfx950)"; }
const char* dms_last_error(void) { return g_err; }

int dms_device_count(int* count) {
  DMS_REQUIRE(count, "null count");
  DMS_HIP(hipGetDeviceCount(count));
  return DMS_OK;
}
int dms_set_device(int device) {
  DMS_HIP(hipSetDevice(device));
  return DMS_OK;
}

size_t dms_reduce_workspace_bytes(void) { return reduce_workspace_bytes(); }

int dms_device_alloc(void** ptr, size_t bytes) {
  DMS_REQUIRE(ptr, "null ptr");
  DMS_HIP(hipMalloc(ptr, bytes ? bytes : 1));
  return DMS_OK;
}
int dms_device_free(void* ptr) {
  if (ptr) DMS_HIP(hipFree(ptr));
  return DMS_OK;
}
int dms_stream_create(dms_stream* out) {
  DMS_REQUIRE(out, "null argument");
  hipStream_t st = nullptr;
  DMS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *out = (dms_stream)st;
  return DMS_OK;
}
int dms_stream_destroy(dms_stream s) {
  if (s) DMS_HIP(hipStreamDestroy(S(s)));
  return DMS_OK;
}
int dms_memcpy_h2d(void* dst, const void* src, size_t bytes, dms_stream s) {
  DMS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(s)));
  DMS_HIP(hipStreamSynchronize(S(s)));
  return DMS_OK;
}
int dms_memcpy_d2h(void* dst, const void* src, size_t bytes, dms_stream s) {
  DMS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(s)));
  DMS_HIP(hipStreamSynchronize(S(s)));
  return DMS_OK;
}
int dms_memcpy_d2d_async(void* dst, const void* src, size_t bytes, dms_stream s) {
  DMS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(s)));
  return DMS_OK;
}
// rows of `width` bytes from a pitched source to a pitched destination in ONE launch - the destination may be mapped host memory
// (a few KB of per-block metadata that the host reads a tick later: a 2-D copy-engine transfer costs more than it moves)
__global__ void k_copy_rows(unsigned* __restrict__ dst, size_t dpitch_w, const unsigned* __restrict__ src, size_t spitch_w, int width_w) {
  const size_t r = blockIdx.x;
  for (int i = threadIdx.x; i < width_w; i += blockDim.x) dst[r * dpitch_w + i] = src[r * spitch_w + i];
}
int dms_copy_rows_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, dms_stream s) {
  DMS_REQUIRE(dst && src && rows >= 1 && width >= 4, "bad argument");
  DMS_REQUIRE((((uintptr_t)dst | (uintptr_t)src | dpitch | spitch | width) & 3) == 0, "4-byte aligned pointers, pitches and width required");
  hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)rows), dim3(256), 0, S(s), (unsigned*)dst, dpitch / 4, (const unsigned*)src, spitch / 4, (int)(width / 4));
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}
int dms_memset(void* dst, int value, size_t bytes, dms_stream s) {
  DMS_HIP(hipMemsetAsync(dst, value, bytes, S(s)));
  return DMS_OK;
}
int dms_mem_info(size_t* free_bytes, size_t* total_bytes) {
  DMS_REQUIRE(free_bytes && total_bytes, "null argument");
  DMS_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return DMS_OK;
}
int dms_stream_sync(dms_stream s) {
  DMS_HIP(hipStreamSynchronize(S(s)));
  return DMS_OK;
}

int dms_icpStep(const dms_mat33* Rcurr, const dms_float3* tcurr, const dms_image2d* vmap_curr, const dms_image2d* nmap_curr,
                const dms_mat33* Rprev_inv, const dms_float3* tprev, const dms_camera* intr, const dms_image2d* vmap_g_prev,
                const dms_image2d* nmap_g_prev, float distThres, float angleThres, void* workspace, size_t workspace_bytes,
                float* matrixA_host, float* vectorB_host, float* residual_host, int threads, int blocks, dms_stream stream) {
  return icpStep(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, distThres, angleThres,
                 workspace, workspace_bytes, matrixA_host, vectorB_host, residual_host, threads, blocks, S(stream));
}

int dms_rgbStep(const dms_image2d* corresImg, float sigma, const dms_image2d* cloud, float fx, float fy, const dms_image2d* dIdx,
                const dms_image2d* dIdy, float sobelScale, void* workspace, size_t workspace_bytes, float* matrixA_host,
                float* vectorB_host, int threads, int blocks, dms_stream stream) {
  return rgbStep(corresImg, sigma, cloud, fx, fy, dIdx, dIdy, sobelScale, workspace, workspace_bytes, matrixA_host, vectorB_host,
                 threads, blocks, S(stream));
}

int dms_so3Step(const dms_image2d* lastImage, const dms_image2d* nextImage, const dms_mat33* imageBasis, const dms_mat33* kinv,
                const dms_mat33* krlr, void* workspace, size_t workspace_bytes, float* matrixA_host, float* vectorB_host,
                float* residual_host, int threads, int blocks, dms_stream stream) {
  return so3Step(lastImage, nextImage, imageBasis, kinv, krlr, workspace, workspace_bytes, matrixA_host, vectorB_host,
                 residual_host, threads, blocks, S(stream));
}

int dms_computeRgbResidual(float minScale, const dms_image2d* dIdx, const dms_image2d* dIdy, const dms_image2d* lastDepth,
                           const dms_image2d* nextDepth, const dms_image2d* lastImage, const dms_image2d* nextImage,
                           dms_image2d* corresImg, void* workspace, size_t workspace_bytes, float maxDepthDelta,
                           const dms_float3* kt, const dms_mat33* krkinv, int* sigmaSum, int* count, int threads, int blocks,
                           dms_stream stream) {
  return computeRgbResidual(minScale, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, corresImg, workspace,
                            workspace_bytes, maxDepthDelta, kt, krkinv, sigmaSum, count, threads, blocks, S(stream));
}

int dms_createVMap(const dms_camera* intr, const dms_image2d* depth, dms_image2d* vmap, float depthCutoff, dms_stream stream) {
  return createVMap(intr, depth, vmap, depthCutoff, S(stream));
}
int dms_createNMap(const dms_image2d* vmap, dms_image2d* nmap, dms_stream stream) { return createNMap(vmap, nmap, S(stream)); }

int dms_tranformMaps(const dms_image2d* vmap_src, const dms_image2d* nmap_src, const dms_mat33* Rmat, const dms_float3* tvec,
                     dms_image2d* vmap_dst, dms_image2d* nmap_dst, dms_stream stream) {
  DMS_REQUIRE(nmap_src && nmap_dst, "null normal map");
  return transformMaps(vmap_src, nmap_src, Rmat, tvec, vmap_dst, nmap_dst, S(stream));
}
int dms_tranformVMap(const dms_image2d* vmap_src, const dms_mat33* Rmat, const dms_float3* tvec, dms_image2d* vmap_dst,
                     dms_stream stream) {
  return transformMaps(vmap_src, nullptr, Rmat, tvec, vmap_dst, nullptr, S(stream));
}

int dms_copyMaps(const float* vmap_src, const float* nmap_src, dms_image2d* vmap_dst, dms_image2d* nmap_dst, dms_stream stream) {
  DMS_REQUIRE(nmap_src && nmap_dst, "null normal map");
  return copyMaps(vmap_src, nmap_src, vmap_dst, nmap_dst, S(stream));
}
int dms_copyVMap(const float* vmap_src, dms_image2d* vmap_dst, dms_stream stream) {
  return copyMaps(vmap_src, nullptr, vmap_dst, nullptr, S(stream));
}

int dms_resizeVMap(const dms_image2d* input, dms_image2d* output, dms_stream stream) {
  return resizeMap(input, output, false, S(stream));
}
int dms_resizeNMap(const dms_image2d* input, dms_image2d* output, dms_stream stream) {
  return resizeMap(input, output, true, S(stream));
}

int dms_imageBGRToIntensity(const dms_image2d* rgba, dms_image2d* dst, dms_stream stream) {
  return imageToIntensity(rgba, dst, S(stream));
}

int dms_verticesToDepth(const float* vmap_src_rgba32f, dms_image2d* dst, float cutOff, dms_stream stream) {
  return verticesToDepth(vmap_src_rgba32f, dst, cutOff, S(stream));
}
int dms_verticesToDepth2D(const dms_image2d* vmap_src, dms_image2d* dst, float cutOff, dms_stream stream) {
  return verticesToDepth2D(vmap_src, dst, cutOff, S(stream));
}

int dms_projectToPointCloud(const dms_image2d* depth, dms_image2d* cloud, const dms_camera* intrinsics, int level,
                            dms_stream stream) {
  return projectToPointCloud(depth, cloud, intrinsics, level, S(stream));
}

int dms_pyrDown(const dms_image2d* src, dms_image2d* dst, dms_stream stream) { return pyrDown(src, dst, S(stream)); }
int dms_pyrDownGaussF(const dms_image2d* src, dms_image2d* dst, dms_stream stream) { return pyrDownGaussF(src, dst, S(stream)); }
int dms_pyrDownUcharGauss(const dms_image2d* src, dms_image2d* dst, dms_stream stream) {
  return pyrDownUcharGauss(src, dst, S(stream));
}

int dms_computeDerivativeImages(const dms_image2d* src, dms_image2d* dx, dms_image2d* dy, dms_stream stream) {
  return derivativeImages(src, dx, dy, S(stream));
}

}  // extern "C"
