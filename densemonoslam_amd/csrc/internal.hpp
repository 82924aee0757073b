// Internal C++ entry points shared between translation units of libdmslam_hip.so.
#pragma once
#include "common.hpp"

namespace dms {

// (thumb_vertex_off / thumb_normal_off / thumb_block_size: common.hpp)

struct TrackInitArgs;  // track_init.hpp

// prep.hip
int pyrDown(const dms_image2d* src, dms_image2d* dst, hipStream_t s);
int createVMap(const dms_camera* intr, const dms_image2d* depth, dms_image2d* vmap, float cutoff, hipStream_t s);
int createNMap(const dms_image2d* vmap, dms_image2d* nmap, hipStream_t s);
int transformMaps(const dms_image2d* vs, const dms_image2d* ns, const dms_mat33* R, const dms_float3* t, dms_image2d* vd,
                  dms_image2d* nd, hipStream_t s);
int copyMaps(const float* vsrc, const float* nsrc, dms_image2d* vd, dms_image2d* nd, hipStream_t s);
int resizeMap(const dms_image2d* in, dms_image2d* out, bool normalize, hipStream_t s);
int pyrDownGaussF(const dms_image2d* src, dms_image2d* dst, hipStream_t s);
int pyrDownUcharGauss(const dms_image2d* src, dms_image2d* dst, hipStream_t s);
int verticesToDepth(const float* vsrc, dms_image2d* dst, float cutOff, hipStream_t s);
int modelPyramidFused(const void* vA, const void* nA, const void* iA, const void* vB, const void* nB, const void* iB, const int* flag_dev,
                      int force_b_img, const float* pose16_dev, dms_image2d* vmaps, dms_image2d* nmaps, dms_image2d* depths,
                      dms_image2d* images, float cutOff, hipStream_t s, bool skip_last_step = false, const unsigned* dense_cnt = nullptr,
                      int dense_samples = 0, int* flag_out = nullptr, const TrackInitArgs* init = nullptr);
int verticesToDepth2D(const dms_image2d* vsrc, dms_image2d* dst, float cutOff, hipStream_t s);
int imageToIntensity(const dms_image2d* rgba, dms_image2d* dst, hipStream_t s);
int derivativeGate(const dms_image2d* src, dms_image2d* dx, dms_image2d* dy, dms_image2d* gate, float minScale, hipStream_t s);
int derivativeImages(const dms_image2d* src, dms_image2d* dx, dms_image2d* dy, hipStream_t s);
int projectToPointCloud(const dms_image2d* depth, dms_image2d* cloud, const dms_camera* intr, int level, hipStream_t s);

// reduce.hip
size_t reduce_workspace_bytes();
int icpStep(const dms_mat33* Rcurr, const dms_float3* tcurr, const dms_image2d* vmap_curr, const dms_image2d* nmap_curr,
            const dms_mat33* Rprev_inv, const dms_float3* tprev, const dms_camera* intr, const dms_image2d* vmap_g_prev,
            const dms_image2d* nmap_g_prev, float distThres, float angleThres, void* workspace, size_t workspace_bytes, float* A,
            float* b, float* residual, int threads, int blocks, hipStream_t s);
int rgbStep(const dms_image2d* corresImg, float sigma, const dms_image2d* cloud, float fx, float fy, const dms_image2d* dIdx,
            const dms_image2d* dIdy, float sobelScale, void* workspace, size_t workspace_bytes, float* A, float* b, int threads,
            int blocks, hipStream_t s);
int computeRgbResidual(float minScale, const dms_image2d* dIdx, const dms_image2d* dIdy, const dms_image2d* lastDepth,
                       const dms_image2d* nextDepth, const dms_image2d* lastImage, const dms_image2d* nextImage,
                       dms_image2d* corresImg, void* workspace, size_t workspace_bytes, float maxDepthDelta, const dms_float3* kt,
                       const dms_mat33* krkinv, int* sigmaSum, int* count, int threads, int blocks, hipStream_t s);
int so3Step(const dms_image2d* lastImage, const dms_image2d* nextImage, const dms_mat33* imageBasis, const dms_mat33* kinv,
            const dms_mat33* krlr, void* workspace, size_t workspace_bytes, float* A, float* b, float* residual, int threads,
            int blocks, hipStream_t s);

}  // namespace dms
