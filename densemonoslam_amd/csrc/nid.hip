// Normalised information distance between a key frame and the live frame: the two operators behind
// the NID key-framing gate of ElasticFusion::fuseFrame (ElasticFusion.cpp:639-677), reference
// computeNIDImg / computeNIDDepth (Cuda/cudafuncs.cu:1513-1650, 1794-1916; kernels :1086-1157,
// bin rules :906-918) — SURVEY.md 8(f3).
//
// The reference builds a num_bins x num_bins float histogram with global atomicAdd(…, 1), copies it
// to the host and evaluates three entropies in scalar float loops.  Here: integer counts (exact,
// order free; the 64 x 64 intensity histogram is privatised in LDS per block, the sparse 500 x 500
// depth histogram goes to L2 atomics directly), and one 1024-thread block evaluates the marginals
// and the entropies with fp64 accumulation, so the score never leaves the device unless asked for.
// Deviations that make the result defined where the reference is not: bins are clamped to
// [0, num_bins-1] (the reference indexes out of bounds for depths beyond max_depth or for
// num_bins that do not divide 256), NaN depth -> bin 0 (CUDA's float->int conversion of NaN).
#include "common.hpp"
#include "internal.hpp"

namespace dms {

__device__ __forceinline__ int nid_bin_u8(unsigned v, int num_bins) {  // cudafuncs.cu:906-911
  const int b_w = 256 / num_bins;
  const int b = (int)v / (b_w > 0 ? b_w : 1);
  return b < num_bins ? b : num_bins - 1;
}
__device__ __forceinline__ int nid_bin_depth(float depth_mm, float max_depth, int num_bins) {  // cudafuncs.cu:913-918
  const int b_w = f2i_rz(max_depth / (float)num_bins);
  const int b = f2i_rz(depth_mm / (float)(b_w > 0 ? b_w : 1));
  return b < 0 ? 0 : (b < num_bins ? b : num_bins - 1);
}

// key-frame value of a pixel: the reference tests `dmap != __int_as_float(0x7fffffff)` (cudafuncs.cu:1099-1100, :1136-1137),
// a comparison with a NaN, which is TRUE for every value: its first branch is always the one taken, i.e. the active
// prediction where `d <= dold` holds and the old one otherwise - a NaN on either side makes the comparison false, so a pixel
// the old view does not see takes the OLD image's value (or, for the depth score, the old depth's NaN -> bin 0).  Pinned by
// the reference's own kernel run on gfx950 (tests/golden/ref_cudafuncs.npz `nid`).
__device__ __forceinline__ int kf_pick(float d, float dold) {  // 0 = active, 1 = old
  return d <= dold ? 0 : 1;
}

__global__ __launch_bounds__(256) void k_nid_hist_img(View<const unsigned char> img_kf, View<const unsigned char> img_kf_old,
                                                      View<const float> dmap_kf, View<const float> dmap_kf_old,
                                                      View<const unsigned char> img_curr, int num_bins, unsigned* __restrict__ hist) {
  extern __shared__ unsigned s_hist[];  // num_bins^2 counters when it fits, else unused
  const bool priv = num_bins * num_bins <= 4096;
  if (priv) {
    for (int k = threadIdx.x; k < num_bins * num_bins; k += blockDim.x) s_hist[k] = 0;
    __syncthreads();
  }
  const int n = img_kf.cols * img_kf.rows;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    const int y = i / img_kf.cols, x = i - y * img_kf.cols;
    const int pick = kf_pick(dmap_kf.at(y, x), dmap_kf_old.at(y, x));
    const unsigned a = pick == 0 ? img_kf.at(y, x) : img_kf_old.at(y, x);
    const int cell = nid_bin_u8(img_curr.at(y, x), num_bins) * num_bins + nid_bin_u8(a, num_bins);  // row = live, column = key frame
    if (priv)
      atomicAdd(&s_hist[cell], 1u);
    else
      atomicAdd(&hist[cell], 1u);
  }
  if (priv) {
    __syncthreads();
    for (int k = threadIdx.x; k < num_bins * num_bins; k += blockDim.x)
      if (s_hist[k]) atomicAdd(&hist[k], s_hist[k]);
  }
}

__global__ __launch_bounds__(256) void k_nid_hist_depth(View<const float> dmap_kf, View<const float> dmap_kf_old, View<const float> dmap_curr,
                                                        int num_bins, float max_depth, unsigned* __restrict__ hist) {
  const int n = dmap_kf.cols * dmap_kf.rows;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    const int y = i / dmap_kf.cols, x = i - y * dmap_kf.cols;
    const float d = dmap_kf.at(y, x), dold = dmap_kf_old.at(y, x);
    const int pick = kf_pick(d, dold);
    const float a = (pick == 0 ? d : dold) * 1000.0f;  // NaN -> bin 0 (nid_bin_depth)
    const float b = dmap_curr.at(y, x) * 1000.0f;
    atomicAdd(&hist[nid_bin_depth(b, max_depth, num_bins) * num_bins + nid_bin_depth(a, max_depth, num_bins)], 1u);
  }
}

__global__ void k_nid_clear(unsigned* hist, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) hist[i] = 0;
}

// out[0] = nid, out[1] = joint entropy, out[2] = key-frame marginal entropy, out[3] = live marginal entropy
__global__ __launch_bounds__(1024) void k_nid_entropy(const unsigned* __restrict__ hist, int num_bins, int num_points, float* __restrict__ marg,
                                                      float* __restrict__ out) {
  __shared__ double s_part[16];
  const int nb2 = num_bins * num_bins;
  const double inv = 1.0 / (double)num_points;
  float* PB = marg;             // column sums (key frame), cudafuncs.cu:1571-1578
  float* PA = marg + num_bins;  // row sums (live frame), :1580-1587
  for (int k = threadIdx.x; k < num_bins; k += blockDim.x) {
    unsigned col = 0, row = 0;
    for (int j = 0; j < num_bins; ++j) {
      col += hist[j * num_bins + k];
      row += hist[k * num_bins + j];
    }
    PB[k] = (float)((double)col * inv);
    PA[k] = (float)((double)row * inv);
  }
  __syncthreads();
  auto block_sum = [&](double v) {
    v = wave_sum_to_lane63_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_part[w];
    return t;
  };
  double j = 0., hk = 0., hc = 0.;
  for (int k = threadIdx.x; k < nb2; k += blockDim.x) {
    const float p = (float)((double)hist[k] * inv);  // histogram_host[...] /= num_points (float)
    if (p != 0.f) j += (double)p * log2((double)p);
  }
  for (int k = threadIdx.x; k < num_bins; k += blockDim.x) {
    const float pb = PB[k], pa = PA[k];
    if (pb != 0.f) hk += (double)pb * log2((double)pb);
    if (pa != 0.f) hc += (double)pa * log2((double)pa);
  }
  const double joint = -block_sum(j), kf = -block_sum(hk), cf = -block_sum(hc);
  if (threadIdx.x == 0) {
    const float jf = (float)joint, kff = (float)kf, cff = (float)cf;
    const float mi = kff + cff - jf;          // :1609-1610
    out[0] = num_points == 0 ? 1.0f : (jf - mi) / jf;  // :1548-1551, :1612
    out[1] = jf;
    out[2] = kff;
    out[3] = cff;
  }
}

size_t nid_workspace_bytes(int num_bins) { return ((size_t)num_bins * num_bins + 2 * (size_t)num_bins + 16) * 4; }

static int nid_finish(unsigned* hist, int num_bins, int num_points, float* nid_host, float* nid_dev, hipStream_t s) {
  float* marg = reinterpret_cast<float*>(hist + (size_t)num_bins * num_bins);
  float* out = marg + 2 * num_bins;
  hipLaunchKernelGGL(k_nid_entropy, dim3(1), dim3(1024), 0, s, hist, num_bins, num_points, marg, out);
  DMS_CHECK_LAUNCH();
  if (nid_dev) DMS_HIP(hipMemcpyAsync(nid_dev, out, sizeof(float), hipMemcpyDeviceToDevice, s));
  if (nid_host) {
    DMS_HIP(hipMemcpyAsync(nid_host, out, sizeof(float), hipMemcpyDeviceToHost, s));
    DMS_HIP(hipStreamSynchronize(s));
  }
  return DMS_OK;
}

int computeNIDImg(const dms_image2d* img_kf, const dms_image2d* img_kf_old, const dms_image2d* dmap_kf, const dms_image2d* dmap_kf_old,
                  const dms_image2d* img_curr, int num_bins, void* workspace, size_t workspace_bytes, float* nid_host, float* nid_dev,
                  hipStream_t s) {
  DMS_REQUIRE(img_kf && img_kf_old && dmap_kf && dmap_kf_old && img_curr && workspace, "null argument");
  DMS_REQUIRE(num_bins >= 1 && num_bins <= 256, "num_bins must be in [1, 256]");
  const int rows = img_kf->rows, cols = img_kf->cols;
  DMS_REQUIRE(img_kf_old->rows == rows && img_kf_old->cols == cols && dmap_kf->rows == rows && dmap_kf->cols == cols &&
                  dmap_kf_old->rows == rows && dmap_kf_old->cols == cols && img_curr->rows == rows && img_curr->cols == cols,
              "shape mismatch");
  if (workspace_bytes < nid_workspace_bytes(num_bins)) {
    set_error("dms_computeNIDImg: workspace of %zu bytes, %zu needed", workspace_bytes, nid_workspace_bytes(num_bins));
    return DMS_ERR_WORKSPACE;
  }
  unsigned* hist = (unsigned*)workspace;
  const int nb2 = num_bins * num_bins, n = rows * cols;
  hipLaunchKernelGGL(k_nid_clear, dim3((nb2 + 255) / 256), dim3(256), 0, s, hist, nb2);
  DMS_CHECK_LAUNCH();
  const size_t lds = nb2 <= 4096 ? (size_t)nb2 * 4 : 0;
  hipLaunchKernelGGL(k_nid_hist_img, dim3(min((n + 255) / 256, 1024)), dim3(256), lds, s, view<const unsigned char>(img_kf),
                     view<const unsigned char>(img_kf_old), view<const float>(dmap_kf), view<const float>(dmap_kf_old),
                     view<const unsigned char>(img_curr), num_bins, hist);
  DMS_CHECK_LAUNCH();
  return nid_finish(hist, num_bins, n, nid_host, nid_dev, s);
}

int computeNIDDepth(const dms_image2d* dmap_kf, const dms_image2d* dmap_kf_old, const dms_image2d* dmap_curr, int num_bins, float max_depth,
                    void* workspace, size_t workspace_bytes, float* nid_host, float* nid_dev, hipStream_t s) {
  DMS_REQUIRE(dmap_kf && dmap_kf_old && dmap_curr && workspace, "null argument");
  DMS_REQUIRE(num_bins >= 1 && num_bins <= 4096, "num_bins must be in [1, 4096]");
  const int rows = dmap_kf->rows, cols = dmap_kf->cols;
  DMS_REQUIRE(dmap_kf_old->rows == rows && dmap_kf_old->cols == cols && dmap_curr->rows == rows && dmap_curr->cols == cols, "shape mismatch");
  if (workspace_bytes < nid_workspace_bytes(num_bins)) {
    set_error("dms_computeNIDDepth: workspace of %zu bytes, %zu needed", workspace_bytes, nid_workspace_bytes(num_bins));
    return DMS_ERR_WORKSPACE;
  }
  unsigned* hist = (unsigned*)workspace;
  const int nb2 = num_bins * num_bins, n = rows * cols;
  hipLaunchKernelGGL(k_nid_clear, dim3(min((nb2 + 255) / 256, 2048)), dim3(256), 0, s, hist, nb2);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_nid_hist_depth, dim3(min((n + 255) / 256, 2048)), dim3(256), 0, s, view<const float>(dmap_kf),
                     view<const float>(dmap_kf_old), view<const float>(dmap_curr), num_bins, max_depth, hist);
  DMS_CHECK_LAUNCH();
  return nid_finish(hist, num_bins, n, nid_host, nid_dev, s);
}

}  // namespace dms

extern "C" {
size_t dms_nid_workspace_bytes(int num_bins) { return dms::nid_workspace_bytes(num_bins); }
int dms_computeNIDImg(const dms_image2d* img_kf, const dms_image2d* img_kf_old, const dms_image2d* dmap_kf, const dms_image2d* dmap_kf_old,
                      const dms_image2d* img_curr, int num_bins, void* workspace, size_t workspace_bytes, float* nid_host, dms_stream s) {
  return dms::computeNIDImg(img_kf, img_kf_old, dmap_kf, dmap_kf_old, img_curr, num_bins, workspace, workspace_bytes, nid_host, nullptr,
                            (hipStream_t)s);
}
int dms_computeNIDDepth(const dms_image2d* dmap_kf, const dms_image2d* dmap_kf_old, const dms_image2d* dmap_curr, int num_bins,
                        float max_depth, void* workspace, size_t workspace_bytes, float* nid_host, dms_stream s) {
  return dms::computeNIDDepth(dmap_kf, dmap_kf_old, dmap_curr, num_bins, max_depth, workspace, workspace_bytes, nid_host, nullptr,
                              (hipStream_t)s);
}
}
