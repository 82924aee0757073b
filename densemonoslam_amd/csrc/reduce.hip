// Operator-layer reduction steps of the tracker: icpStep, rgbStep, computeRgbResidual,
// so3Step (reference Cuda/reduce.cu).  Each is a grid-stride pass of 256-thread blocks that
// accumulates per-thread running sums, reduces them per wave with a step-major DPP butterfly,
// across the 4 waves in fp64, and writes one 128-byte record per block; a single 256-thread
// block then folds the records in fp64 with batched 16-byte loads (common.hpp).  No float
// atomics: the summation order is fixed by the launch shape, so results are deterministic.
#include "pixel_ops.hpp"

namespace dms {

// workspace carving ----------------------------------------------------------------------
struct ReduceWs {
  float* partials;   // [kMaxPartialBlocks][kPartStride] one 128-byte record per block
  float* result;     // [32]
  int* ipartials;    // [2][kMaxPartialBlocks]
  int* iresult;      // [4]
};
static constexpr size_t kWsBytes = (size_t)kPartStride * kMaxPartialBlocks * 4 + 32 * 4 + 2 * (size_t)kMaxPartialBlocks * 4 + 16;

size_t reduce_workspace_bytes() { return kWsBytes; }

static inline ReduceWs carve(void* ws) {
  ReduceWs r;
  char* p = (char*)ws;
  r.partials = (float*)p;
  p += (size_t)kPartStride * kMaxPartialBlocks * 4;
  r.result = (float*)p;
  p += 32 * 4;
  r.ipartials = (int*)p;
  p += 2 * (size_t)kMaxPartialBlocks * 4;
  r.iresult = (int*)p;
  return r;
}

// final fold of the per-block records ---------------------------------------------------------
__global__ __launch_bounds__(256) void k_fold_records(const float* __restrict__ partials, int nblocks, int nv, float* __restrict__ out) {
  fold_records256(partials, nblocks, nv, out);
}

__global__ __launch_bounds__(1024) void k_fold_rows_i(const int* __restrict__ partials, int stride, int nblocks, int nv,
                                                      int* __restrict__ out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int k = wid; k < nv; k += nw) {
    int s = 0;
    for (int b = lane; b < nblocks; b += 64) s += partials[(size_t)k * stride + b];
    s = wave_sum_to_lane63_i(s);
    if (lane == 63) out[k] = s;
  }
}

// ---- icpStep (reference icpKernel, reduce.cu:235-365) -----------------------------------
__global__ __launch_bounds__(kBlock) void k_icp(IcpParams p, MapPtrs m, int N, float* __restrict__ partials, int stride) {
  float acc[kSE3];
#pragma unroll
  for (int k = 0; k < kSE3; ++k) acc[k] = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += blockDim.x * gridDim.x) {
    const int y = i / p.cols;
    const int x = i - y * p.cols;
    float row[7];
    const bool found = icp_row(p, m, x, y, row);
    accumulate_se3(acc, row, found);
  }
  block_reduce_store<kSE3>(acc, partials, stride, blockIdx.x);
}

// ---- rgbStep (reference rgbKernel, reduce.cu:544-641) -----------------------------------
__global__ __launch_bounds__(kBlock) void k_rgb(RgbStepParams p, const dms_dataterm* __restrict__ corres, const float* cloud,
                                                size_t cloud_pitch, const short* dIdx, const short* dIdy, size_t dI_pitch, int N,
                                                float* __restrict__ partials, int stride) {
  float acc[kSE3];
#pragma unroll
  for (int k = 0; k < kSE3; ++k) acc[k] = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += blockDim.x * gridDim.x) {
    const dms_dataterm c = corres[i];  // reference indexes corresImg.data[i] linearly (reduce.cu:562)
    float row[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c.valid & 0xff) rgb_row(p, c, cloud, cloud_pitch, dIdx, dIdy, dI_pitch, row);
    accumulate_se3(acc, row, (c.valid & 0xff) != 0);
  }
  block_reduce_store<kSE3>(acc, partials, stride, blockIdx.x);
}

// ---- computeRgbResidual (reference residualKernel, reduce.cu:739-863) -------------------
__global__ __launch_bounds__(kBlock) void k_rgb_residual(RgbResParams p, RgbResPtrs q, dms_dataterm* __restrict__ corres, int N,
                                                         int* __restrict__ ipartials, int stride) {
  int cnt = 0, sig = 0;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += blockDim.x * gridDim.x) {
    const int i = k / p.cols;
    const int j0 = k - i * p.cols;
    dms_dataterm c;
    int d2;
    if (rgb_residual(p, q, j0, i, c, d2)) {
      cnt += 1;
      sig += d2;
    }
    corres[k] = c;
  }
  __shared__ int lds[kBlock / kWave][2];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  cnt = wave_sum_to_lane63_i(cnt);
  sig = wave_sum_to_lane63_i(sig);
  if (lane == 63) {
    lds[wid][0] = cnt;
    lds[wid][1] = sig;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    int s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += lds[w][threadIdx.x];
    ipartials[(size_t)threadIdx.x * stride + blockIdx.x] = s;
  }
}

// ---- so3Step (reference so3Kernel, reduce.cu:927-1052) ----------------------------------
__global__ __launch_bounds__(kBlock) void k_so3(So3Params p, const unsigned char* lastImage, size_t last_pitch,
                                                const unsigned char* nextImage, size_t next_pitch, int N,
                                                float* __restrict__ partials, int stride) {
  float acc[kSO3];
#pragma unroll
  for (int k = 0; k < kSO3; ++k) acc[k] = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += blockDim.x * gridDim.x) {
    const int y = i / p.cols;
    const int x = i - y * p.cols;
    float row[4];
    const bool found = so3_row(p, lastImage, last_pitch, nextImage, next_pitch, x, y, row);
    accumulate_so3(acc, row, found);
  }
  block_reduce_store<kSO3>(acc, partials, stride, blockIdx.x);
}

// ---- host side --------------------------------------------------------------------------
static int pick_blocks(int N, int threads, int blocks) {
  (void)threads;
  if (blocks > 0) return blocks > kMaxPartialBlocks ? kMaxPartialBlocks : blocks;
  return reduce_blocks_for(N);
}

static void unpack_se3(const float* host, float* A, float* b) {
  int shift = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      const float v = host[shift++];
      if (j == 6)
        b[i] = v;
      else
        A[j * 6 + i] = A[i * 6 + j] = v;
    }
}

MapPtrs make_map_ptrs(const dms_image2d* vc, const dms_image2d* nc, const dms_image2d* vp, const dms_image2d* np) {
  MapPtrs m;
  m.vcurr = (const float*)vc->data;
  m.vcurr_pitch = vc->pitch;
  m.ncurr = (const float*)nc->data;
  m.ncurr_pitch = nc->pitch;
  m.vprev = (const float*)vp->data;
  m.vprev_pitch = vp->pitch;
  m.nprev = (const float*)np->data;
  m.nprev_pitch = np->pitch;
  return m;
}

int icpStep(const dms_mat33* Rcurr, const dms_float3* tcurr, const dms_image2d* vmap_curr, const dms_image2d* nmap_curr,
            const dms_mat33* Rprev_inv, const dms_float3* tprev, const dms_camera* intr, const dms_image2d* vmap_g_prev,
            const dms_image2d* nmap_g_prev, float distThres, float angleThres, void* workspace, size_t workspace_bytes, float* A,
            float* b, float* residual, int threads, int blocks, hipStream_t s) {
  DMS_REQUIRE(Rcurr && tcurr && vmap_curr && nmap_curr && Rprev_inv && tprev && intr && vmap_g_prev && nmap_g_prev, "null argument");
  DMS_REQUIRE(A && b && residual, "null host output");
  DMS_REQUIRE(threads == 0 || threads == kBlock, "threads must be 0 (auto) or 256");
  if (!workspace || workspace_bytes < kWsBytes) {
    set_error("icpStep: workspace too small (%zu < %zu)", workspace_bytes, kWsBytes);
    return DMS_ERR_WORKSPACE;
  }
  DMS_REQUIRE(vmap_curr->rows % 3 == 0, "vmap rows must be 3*H");
  IcpParams p;
  p.Rcurr = to_m33(Rcurr);
  p.tcurr = to_f3(tcurr);
  p.Rprev_inv = to_m33(Rprev_inv);
  p.tprev = to_f3(tprev);
  p.fx = intr->fx;
  p.fy = intr->fy;
  p.cx = intr->cx;
  p.cy = intr->cy;
  p.distThres = distThres;
  p.angleThres = angleThres;
  p.dist2Le = sqrt_le_bound(distThres);
  p.sine2Le = sqrt_lt_bound(angleThres);
  p.cols = vmap_curr->cols;
  p.rows = vmap_curr->rows / 3;
  const int N = p.cols * p.rows;
  const int nb = pick_blocks(N, threads, blocks);
  ReduceWs w = carve(workspace);
  hipLaunchKernelGGL(k_icp, dim3(nb), dim3(kBlock), 0, s, p, make_map_ptrs(vmap_curr, nmap_curr, vmap_g_prev, nmap_g_prev), N,
                     w.partials, kMaxPartialBlocks);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_fold_records, dim3(1), dim3(256), 0, s, w.partials, nb, kSE3, w.result);
  DMS_CHECK_LAUNCH();
  float host[32];
  DMS_HIP(hipMemcpyAsync(host, w.result, kSE3 * sizeof(float), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  unpack_se3(host, A, b);
  residual[0] = host[27];
  residual[1] = host[28];
  return DMS_OK;
}

int rgbStep(const dms_image2d* corresImg, float sigma, const dms_image2d* cloud, float fx, float fy, const dms_image2d* dIdx,
            const dms_image2d* dIdy, float sobelScale, void* workspace, size_t workspace_bytes, float* A, float* b, int threads,
            int blocks, hipStream_t s) {
  DMS_REQUIRE(corresImg && cloud && dIdx && dIdy && A && b, "null argument");
  DMS_REQUIRE(threads == 0 || threads == kBlock, "threads must be 0 (auto) or 256");
  DMS_REQUIRE(dIdx->pitch == dIdy->pitch, "dIdx/dIdy pitch must match");
  if (!workspace || workspace_bytes < kWsBytes) {
    set_error("rgbStep: workspace too small (%zu < %zu)", workspace_bytes, kWsBytes);
    return DMS_ERR_WORKSPACE;
  }
  RgbStepParams p;
  p.sigma = sigma;
  p.fx = fx;
  p.fy = fy;
  p.sobelScale = sobelScale;
  const int N = corresImg->cols * corresImg->rows;
  const int nb = pick_blocks(N, threads, blocks);
  ReduceWs w = carve(workspace);
  hipLaunchKernelGGL(k_rgb, dim3(nb), dim3(kBlock), 0, s, p, (const dms_dataterm*)corresImg->data, (const float*)cloud->data,
                     cloud->pitch, (const short*)dIdx->data, (const short*)dIdy->data, dIdx->pitch, N, w.partials,
                     kMaxPartialBlocks);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_fold_records, dim3(1), dim3(256), 0, s, w.partials, nb, kSE3, w.result);
  DMS_CHECK_LAUNCH();
  float host[32];
  DMS_HIP(hipMemcpyAsync(host, w.result, kSE3 * sizeof(float), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  unpack_se3(host, A, b);
  return DMS_OK;
}

RgbResPtrs make_rgbres_ptrs(const dms_image2d* dIdx, const dms_image2d* dIdy, const dms_image2d* lastDepth,
                            const dms_image2d* nextDepth, const dms_image2d* lastImage, const dms_image2d* nextImage) {
  RgbResPtrs q;
  q.dIdx = (const short*)dIdx->data;
  q.dIdy = (const short*)dIdy->data;
  q.dI_pitch = dIdx->pitch;
  q.lastDepth = (const float*)lastDepth->data;
  q.lastDepth_pitch = lastDepth->pitch;
  q.nextDepth = (const float*)nextDepth->data;
  q.nextDepth_pitch = nextDepth->pitch;
  q.lastImage = (const unsigned char*)lastImage->data;
  q.lastImage_pitch = lastImage->pitch;
  q.nextImage = (const unsigned char*)nextImage->data;
  q.nextImage_pitch = nextImage->pitch;
  q.gate = nullptr;
  q.gate_pitch = 0;
  return q;
}

int computeRgbResidual(float minScale, const dms_image2d* dIdx, const dms_image2d* dIdy, const dms_image2d* lastDepth,
                       const dms_image2d* nextDepth, const dms_image2d* lastImage, const dms_image2d* nextImage,
                       dms_image2d* corresImg, void* workspace, size_t workspace_bytes, float maxDepthDelta, const dms_float3* kt,
                       const dms_mat33* krkinv, int* sigmaSum, int* count, int threads, int blocks, hipStream_t s) {
  DMS_REQUIRE(dIdx && dIdy && lastDepth && nextDepth && lastImage && nextImage && corresImg && kt && krkinv && sigmaSum && count,
              "null argument");
  DMS_REQUIRE(threads == 0 || threads == kBlock, "threads must be 0 (auto) or 256");
  DMS_REQUIRE(dIdx->pitch == dIdy->pitch, "dIdx/dIdy pitch must match");
  if (!workspace || workspace_bytes < kWsBytes) {
    set_error("computeRgbResidual: workspace too small (%zu < %zu)", workspace_bytes, kWsBytes);
    return DMS_ERR_WORKSPACE;
  }
  RgbResParams p;
  p.minScale = minScale;
  p.maxDepthDelta = maxDepthDelta;
  p.kt = to_f3(kt);
  p.krkinv = to_m33(krkinv);
  p.cols = nextImage->cols;
  p.rows = nextImage->rows;
  const int N = p.cols * p.rows;
  const int nb = pick_blocks(N, threads, blocks);
  ReduceWs w = carve(workspace);
  hipLaunchKernelGGL(k_rgb_residual, dim3(nb), dim3(kBlock), 0, s, p,
                     make_rgbres_ptrs(dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage), (dms_dataterm*)corresImg->data, N,
                     w.ipartials, kMaxPartialBlocks);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_fold_rows_i, dim3(1), dim3(1024), 0, s, w.ipartials, kMaxPartialBlocks, nb, 2, w.iresult);
  DMS_CHECK_LAUNCH();
  int host[2];
  DMS_HIP(hipMemcpyAsync(host, w.iresult, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  *count = host[0];
  *sigmaSum = host[1];
  return DMS_OK;
}

int so3Step(const dms_image2d* lastImage, const dms_image2d* nextImage, const dms_mat33* imageBasis, const dms_mat33* kinv,
            const dms_mat33* krlr, void* workspace, size_t workspace_bytes, float* A, float* b, float* residual, int threads,
            int blocks, hipStream_t s) {
  DMS_REQUIRE(lastImage && nextImage && imageBasis && kinv && krlr && A && b && residual, "null argument");
  DMS_REQUIRE(threads == 0 || threads == kBlock, "threads must be 0 (auto) or 256");
  if (!workspace || workspace_bytes < kWsBytes) {
    set_error("so3Step: workspace too small (%zu < %zu)", workspace_bytes, kWsBytes);
    return DMS_ERR_WORKSPACE;
  }
  So3Params p;
  p.imageBasis = to_m33(imageBasis);
  p.kinv = to_m33(kinv);
  p.krlr = to_m33(krlr);
  p.cols = nextImage->cols;
  p.rows = nextImage->rows;
  const int N = p.cols * p.rows;
  const int nb = pick_blocks(N, threads, blocks);
  ReduceWs w = carve(workspace);
  hipLaunchKernelGGL(k_so3, dim3(nb), dim3(kBlock), 0, s, p, (const unsigned char*)lastImage->data, lastImage->pitch,
                     (const unsigned char*)nextImage->data, nextImage->pitch, N, w.partials, kMaxPartialBlocks);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_fold_records, dim3(1), dim3(256), 0, s, w.partials, nb, kSO3, w.result);
  DMS_CHECK_LAUNCH();
  float host[kSO3];
  DMS_HIP(hipMemcpyAsync(host, w.result, kSO3 * sizeof(float), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  int shift = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 4; ++j) {
      const float v = host[shift++];
      if (j == 3)
        b[i] = v;
      else
        A[j * 3 + i] = A[i * 3 + j] = v;
    }
  residual[0] = host[9];
  residual[1] = host[10];
  return DMS_OK;
}

}  // namespace dms
