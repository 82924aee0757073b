// End-of-run exports of the reference (host side): the surfel map as a binary PLY (ElasticFusion::savePly,
// ElasticFusion.cpp:781-885) and a camera trajectory (Context::saveTrajectory, Context.h:117-156), written so that the
// files are the reference's byte for byte for the same map / poses.  The map comes off the device once, in the reference's
// 15-float record (GlobalModel::downloadMap, GlobalModel.cpp:866-896 = dms_model_download_ref).
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/dmslam_fusion.h"
#include "common.hpp"

extern "C" {

int dms_model_save_ply(dms_model* m, const char* path, float confidenceThreshold, int reference_offsets, unsigned int* written) {
  DMS_REQUIRE(m && path, "null argument");
  unsigned int count = 0;
  if (int rc = dms_model_count(m, &count, nullptr)) return rc;
  const int stride = 3 * 4 + DMS_REF_MAX_SENSORS;  // Vertex::SIZE / 4 (Vertex.cpp:49-50)
  std::vector<float> map((size_t)count * stride + 1);
  unsigned int got = 0;
  if (count)
    if (int rc = dms_model_download_ref(m, map.data(), count, &got, nullptr)) return rc;
  DMS_REQUIRE(got == count, "map changed while it was being saved");
  unsigned int valid = 0;
  for (unsigned int i = 0; i < count; ++i)
    if (map[(size_t)i * stride + 3] > confidenceThreshold) ++valid;  // :798-804
  FILE* fp = fopen(path, "wb");
  if (!fp) {
    ::dms::set_error("dms_model_save_ply: cannot open %s", path);
    return DMS_ERR_INVALID_ARG;
  }
  // :806-828 (the reference streams these pieces one after another)
  fprintf(fp,
          "ply\nformat binary_little_endian 1.0\nelement vertex %u\nproperty float x\nproperty float y\nproperty float z"
          "\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\nproperty float ny\nproperty float nz"
          "\nproperty float radius\nend_header\n",
          valid);
  const size_t total = (size_t)count * stride;
  const int noff = reference_offsets ? 18 : 8 + DMS_REF_MAX_SENSORS;  // :845-847 reads + 18 (see dmslam_fusion.h)
  std::vector<unsigned char> rec;
  rec.reserve((size_t)valid * 31);
  for (unsigned int i = 0; i < count; ++i) {
    const size_t o = (size_t)i * stride;
    if (!(map[o + 3] > confidenceThreshold)) continue;
    float nor[4];
    for (int k = 0; k < 4; ++k) nor[k] = o + noff + k < total ? map[o + noff + k] : 0.f;
    nor[0] *= -1;  // :849-851
    nor[1] *= -1;
    nor[2] *= -1;
    const int c = (int)map[o + 4];  // colour packed in a float (:862-864)
    const unsigned char rgb[3] = {(unsigned char)(c >> 16 & 0xFF), (unsigned char)(c >> 8 & 0xFF), (unsigned char)(c & 0xFF)};
    unsigned char b[31];
    memcpy(b, &map[o], 12);
    memcpy(b + 12, rgb, 3);
    memcpy(b + 15, nor, 16);
    rec.insert(rec.end(), b, b + 31);
  }
  const bool ok = rec.empty() || fwrite(rec.data(), 1, rec.size(), fp) == rec.size();
  const bool closed = fclose(fp) == 0;
  if (!ok || !closed) {
    ::dms::set_error("dms_model_save_ply: short write to %s", path);
    return DMS_ERR_INVALID_ARG;
  }
  if (written) *written = valid;
  return DMS_OK;
}

int dms_trajectory_save(const char* path, const float* poses16, size_t n) {
  DMS_REQUIRE(path && (poses16 || n == 0), "null argument");
  FILE* fp = fopen(path, "w");
  if (!fp) {
    ::dms::set_error("dms_trajectory_save: cannot open %s", path);
    return DMS_ERR_INVALID_ARG;
  }
  // Context.h:149-152: `f << rot(0,0) << " " << ... << trans(2) << " " << "\n"`; an ostream prints a float like "%g"
  for (size_t i = 0; i < n; ++i) {
    const float* p = poses16 + i * 16;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) fprintf(fp, "%g ", (double)p[r * 4 + c]);
    fputc('\n', fp);
  }
  if (fclose(fp) != 0) {
    ::dms::set_error("dms_trajectory_save: short write to %s", path);
    return DMS_ERR_INVALID_ARG;
  }
  return DMS_OK;
}

}  // extern "C"
