// ReferenceFrame's own IndexMap + RGBDOdometry (Core/src/ReferenceFrame.h:203-214: m_index, m_rgbd) and the second half of
// ReferenceFrame::resolveRelativeTransformationFern (ReferenceFrame.h:66-110): once Ferns::findFrame(..., interMap = true) has
// produced a recoveryPose for a camera of ANOTHER map, the owner of the queried map
//   1. predicts its map INACTIVE at recoveryPose            (m_index.combinedPredict(recoveryPose, model, depthCutoff, confidence, 0,
//                                                            timeIdx, maxTime, timeDelta, INACTIVE), :72-80)
//   2. initialises its tracker: model side = that prediction, (m_rgbd.initICPModel(oldVertexTex, oldNormalTex, depthCutoff, recoveryPose),
//      live side = the querying camera's fill-in textures      initICP(vertexTexture, normalTexture, depthCutoff), initRGBModel, initRGB,
//                                                              :82-86)
//   3. refines at full resolution                            (getIncrementalTransformation(t, r, false, 10, true, false, true, true):
//                                                            SO3 pre-alignment, 3 x 50 ICP + RGB iterations, :88-90)
//   4. relativeTransform = refined pose * currPose^-1 (:95) and accepts on the covariance diagonal, lastICPError and lastICPCount
//      against Options' covThresh / icpErrThresh / icpCountThresh (:98-110).
// Everything is composed from the library's own operators (splat prediction, the tracker object); this file adds no kernel.
//
// Two readings of the reference that are stated here because the block is compiled out there (`if (false)`, ElasticFusion.cpp:597)
// and was never run against them:
//   * :85 passes m_index.imageTex() — the ACTIVE colour target of an IndexMap that only ever renders INACTIVE (:72), i.e. a texture
//     nothing has written — where the model-to-model tracker of the same code base takes oldImageTex() (ElasticFusion.cpp:413).
//     The INACTIVE prediction's own colour image is used here (and in oracle/orc_pipeline.Session.refine).
//   * (a third one is kept literally, see step 2 below: the call order makes lastDepth the LIVE camera's depth)
//   * the tracker's lastNextImage pyramid (what the SO3 pre-alignment compares the live image with, RGBDOdometry.cpp:338) is whatever
//     the previous refinement of this reference frame left there (:595-601 swaps after every call); before the first one it is the
//     zero-filled pyramid every tracker of this library starts with (uninitialised device memory in the reference).
#include <cstring>

#include "../../include/dmslam_fusion.h"
#include "internal.hpp"
#include "smallmath.hpp"

namespace dms {
int splat_predict(dms_model* m, const dms_pose_block* pose, const dms_camera* cam, float maxDepth, float confThreshold, int time,
                  int timeIdx, int maxTime, int timeDelta, int active, unsigned long long* zbuf, dms_predict_out* out,
                  dms_image2d* depth_out, int zclean, hipStream_t s, const float* second_conf_time_maxtime = nullptr,
                  unsigned long long* zbuf2 = nullptr, int resolve_only = 0, const struct FillArgs* fill = nullptr,
                  const struct TrackInitArgs* init = nullptr);
int clear_zbuf(unsigned long long* zbuf, int n, hipStream_t s);
int model_flush_pending(dms_model* m, hipStream_t s);
}  // namespace dms

struct dms_refframe {
  int W = 0, H = 0;
  dms_camera cam{};
  dms_odometry* rgbd = nullptr;  // m_rgbd
  void* arena = nullptr;         // m_index's INACTIVE targets + z-buffer + pose block
  unsigned long long* zbuf = nullptr;
  dms_pose_block* pose = nullptr;
  dms_predict_out old{};  // oldImageTex / oldVertexTex / oldNormalTex / oldTimeTex
  int refinements = 0;
};

extern "C" {

int dms_refframe_create(dms_refframe** out, int width, int height, float cx, float cy, float fx, float fy) {
  DMS_REQUIRE(out, "null out");
  DMS_REQUIRE(width >= 16 && height >= 16, "resolution");
  dms_refframe* r = new dms_refframe();
  r->W = width;
  r->H = height;
  r->cam = dms_camera{fx, fy, cx, cy};
  int rc = dms_odometry_create(&r->rgbd, width, height, cx, cy, fx, fy, 0.f, 0.f);
  if (rc) {
    delete r;
    return rc;
  }
  const size_t N = (size_t)width * height;
  const size_t bytes = N * 8 + N * 4 + N * 16 + N * 16 + N * 2 + 256 + sizeof(dms_pose_block) + 256;
  if (hipMalloc(&r->arena, bytes) != hipSuccess) {
    dms_odometry_destroy(r->rgbd);
    delete r;
    ::dms::set_error("dms_refframe_create: out of device memory");
    return DMS_ERR_HIP;
  }
  unsigned char* p = (unsigned char*)r->arena;
  r->zbuf = (unsigned long long*)p;
  p += N * 8;
  r->old.vertex = dms_image2d{p, (size_t)width * 16, height, width};
  p += N * 16;
  r->old.normal = dms_image2d{p, (size_t)width * 16, height, width};
  p += N * 16;
  r->old.image = dms_image2d{p, (size_t)width * 4, height, width};
  p += N * 4;
  r->old.time = dms_image2d{p, (size_t)width * 2, height, width};
  p += (N * 2 + 255) / 256 * 256;
  r->pose = (dms_pose_block*)p;
  rc = ::dms::clear_zbuf(r->zbuf, (int)N, nullptr);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = DMS_ERR_HIP;
  if (rc) {
    (void)hipFree(r->arena);
    dms_odometry_destroy(r->rgbd);
    delete r;
    return rc;
  }
  *out = r;
  return DMS_OK;
}

int dms_refframe_destroy(dms_refframe* r) {
  if (!r) return DMS_OK;
  (void)hipDeviceSynchronize();
  if (r->rgbd) dms_odometry_destroy(r->rgbd);
  if (r->arena) (void)hipFree(r->arena);
  delete r;
  return DMS_OK;
}

dms_odometry* dms_refframe_odometry(dms_refframe* r) { return r ? r->rgbd : nullptr; }

int dms_refframe_get_prediction(dms_refframe* r, dms_predict_out* view) {
  DMS_REQUIRE(r && view, "null argument");
  *view = r->old;
  return DMS_OK;
}

int dms_refframe_refine(dms_refframe* r, dms_model* map, const float* recoveryPose16, const float* currPose16, const float* vertex_dev,
                        const float* normal_dev, const void* image_rgba_dev, int depthCutoff, float confidenceThreshold, int timeIdx,
                        int timeDelta, int maxTime, float covThresh, float icpErrThresh, float icpCountThresh, dms_intermap_result* out,
                        dms_stream st) {
  DMS_REQUIRE(r && map && recoveryPose16 && currPose16 && vertex_dev && normal_dev && image_rgba_dev && out, "null argument");
  DMS_REQUIRE(timeIdx >= 0 && timeIdx < DMS_MAX_SENSORS, "timeIdx out of range");
  hipStream_t s = (hipStream_t)st;
  int rc;
  memset(out, 0, sizeof(*out));
  // `const int depthCutoff` (ReferenceFrame.h:42): the caller's maxDepthProcessed arrives truncated to an integer and goes on as (float)
  const float cutoff = (float)depthCutoff;
  if ((rc = ::dms::model_flush_pending(map, s))) return rc;
  if ((rc = dms_pose_block_set(r->pose, recoveryPose16, st))) return rc;
  // 1. :72-80 (time = 0, the querying camera's time slot, maxTime = its tick)
  if ((rc = ::dms::splat_predict(map, r->pose, &r->cam, cutoff, confidenceThreshold, 0, timeIdx, maxTime, timeDelta, 0, r->zbuf, &r->old,
                                 nullptr, 1, s)))
    return rc;
  // 2. :82-86, in the reference's literal order: initICPModel, initICP, initRGBModel, initRGB.  It is NOT the order its own WARNING
  // asks for (ElasticFusion.cpp:172: "initICP* must be called before initRGB*" pairwise; the model-to-model tracker calls
  // initICPModel, initRGBModel, initICP, initRGB, :410-418): populateRGBDData takes the depth of a pyramid from vmaps_tmp, which
  // initICP has just overwritten with the LIVE vertices, so lastDepth — the model side's depth in the photometric term — is the
  // live camera's depth map under the model's colour image.  Kept as written: it is defined behaviour and it is the reference's.
  if ((rc = dms_odometry_initICPModel(r->rgbd, (const float*)r->old.vertex.data, (const float*)r->old.normal.data, cutoff, recoveryPose16, st)))
    return rc;
  if ((rc = dms_odometry_initICP_maps(r->rgbd, vertex_dev, normal_dev, cutoff, st))) return rc;
  if ((rc = dms_odometry_initRGBModel(r->rgbd, &r->old.image, st))) return rc;
  const dms_image2d live_img = {const_cast<void*>(image_rgba_dev), (size_t)r->W * 4, r->H, r->W};
  if ((rc = dms_odometry_initRGB(r->rgbd, &live_img, st))) return rc;
  // 3. :88-90
  float trans[3] = {recoveryPose16[3], recoveryPose16[7], recoveryPose16[11]}, rot[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) rot[i * 3 + j] = recoveryPose16[i * 4 + j];
  dms_track_result tr;
  if ((rc = dms_odometry_getIncrementalTransformation(r->rgbd, trans, rot, 0, 10.0f, 1, 0, 1, 1, &tr, st))) return rc;
  // 4. :92-110
  float refined[16];
  memcpy(refined, recoveryPose16, sizeof(refined));
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) refined[i * 4 + j] = rot[i * 3 + j];
    refined[i * 4 + 3] = trans[i];
  }
  memcpy(out->refinedPose, refined, sizeof(refined));
  if ((rc = dms_relative_transform(refined, currPose16, out->relativeTransform))) return rc;
  double cov[36];
  if ((rc = dms_odometry_getCovariance(r->rgbd, cov))) return rc;
  int covOk = 1;
  for (int i = 0; i < 6; ++i) {
    out->cov_diag[i] = cov[i * 6 + i];
    if (cov[i * 6 + i] > (double)covThresh) covOk = 0;  // (a NaN diagonal compares false, as in the reference)
  }
  out->cov_ok = covOk;
  out->lastICPError = tr.lastICPError;
  out->lastICPCount = tr.lastICPCount;
  out->lastRGBError = tr.lastRGBError;
  out->lastRGBCount = tr.lastRGBCount;
  for (int l = 0; l < DMS_NUM_PYRS; ++l) out->iterations_run[l] = tr.iterations_run[l];
  out->so3_iterations_run = tr.so3_iterations_run;
  out->accepted = (covOk && tr.lastICPError < icpErrThresh && tr.lastICPCount > icpCountThresh) ? 1 : 0;
  r->refinements += 1;
  return DMS_OK;
}

}  // extern "C"
