// Per-camera frame state kept in HBM by the frame step (fusion_frame.hip) and updated by the tracker's
// last kernel (track.hip): shared so that the post-tracking bookkeeping needs no launch of its own.
#pragma once
#include "../../include/dmslam_fusion.h"
#include "smallmath.hpp"

namespace dms {

struct FrameState {
  dms_pose_block cur;   // current pose + inverse
  float lastPose[16];   // pose at the end of the previous frame (ElasticFusion.cpp:158)
  float weighting;      // ElasticFusion.cpp:252-268
  int fill_in;          // shouldFillIn (ElasticFusion.cpp:167)
  unsigned surfels;
  int track_timeouts;   // sticky: tracker calls of this camera whose resident kernels timed out at a grid barrier (those frames keep
                        // their prior pose and fuse nothing; dms_fusion_fetch reports DMS_ERR_TIMEOUT)
  int track_range_failures;  // sticky: tracker calls that found no fixed-point range for a cross-pixel sum (TrackState::sync_timeout
                             // == 2: non-finite maps); same consequence for the frame, but nothing to do with residency
};

// Outcome of the local loop-closure candidate of one frame (ElasticFusion.cpp:427-474); the sampled
// constraints follow it in the same device buffer: 8 floats each {raw xyz, model xyz, time, 0}.
struct LoopState {
  int ok;
  int n_constraints;
  float icp_error, icp_count;
  float est_pose[16];
  double cov_diag[6];
};

// after tracking: inverse of the new pose and the velocity weight (ElasticFusion.cpp:252-268); the
// pose becomes next frame's lastPose (ElasticFusion.cpp:158).
// rodrigues2 (ElasticFusion.cpp:941-985) re-orthonormalises diffRot with an SVD first; a product
// of float rotations is orthonormal to ~1e-7, so the matrix is used as is.
// `timed_out`: the tracker's result is invalid (the caller has restored the prior pose): count it and mark the frame
// with a negative weight, which k_fuse_associate reads as "fuse nothing" (a legal weight is >= 0).
// `held`: 32 floats the caller already holds in LDS — [0..15] the new pose, [16..31] the previous frame's (the resident
// tracker's last block has just computed the one and read the other at its start: re-reading both from memory costs two
// dependent round trips at the very end of the kernel); null = read both from the state block.
__device__ inline void frame_after_track_body(FrameState* st, float weightMultiplier, int timeout_code = 0, float* held = nullptr) {
  const bool timed_out = timeout_code != 0;  // 1: a grid-wide wait gave up, 2: no fixed-point range for a sum
  // (every loop over these 16-element arrays is fully unrolled: constant indices keep them in registers — a loop left rolled
  // puts them in scratch, which the large level kernels would then allocate for every lane)
  float P[16], Lp[16], Ti[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    P[i] = held ? held[i] : st->cur.pose[i];
    Lp[i] = held ? held[16 + i] : st->lastPose[i];
  }
  sm::inv4t<float>(P, Ti);
#pragma unroll
  for (int i = 0; i < 16; ++i) st->cur.t_inv[i] = Ti[i];
  float diff[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = Ti[i * 4 + 0] * Lp[0 * 4 + j];
      s += Ti[i * 4 + 1] * Lp[1 * 4 + j];
      s += Ti[i * 4 + 2] * Lp[2 * 4 + j];
      s += Ti[i * 4 + 3] * Lp[3 * 4 + j];
      diff[i * 4 + j] = s;
    }
  const float tn = sqrtf(diff[3] * diff[3] + diff[7] * diff[7] + diff[11] * diff[11]);
  double rx = (double)diff[9] - (double)diff[6];
  double ry = (double)diff[2] - (double)diff[8];
  double rz = (double)diff[4] - (double)diff[1];
  const double sn = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = ((double)(diff[0] + diff[5] + diff[10]) - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = acos(c);
  double rn;
  if (sn < 1e-5) {
    rn = c > 0 ? 0.0 : theta;  // |r| = theta in the c <= 0 branch (unit axis scaled by theta)
  } else {
    const double vth = (1 / (2 * sn)) * theta;
    rx *= vth;
    ry *= vth;
    rz *= vth;
    rn = sqrt(rx * rx + ry * ry + rz * rz);
  }
  float weighting = fmaxf(tn, (float)rn);
  const float largest = 0.01f, minWeight = 0.5f;
  if (weighting > largest) weighting = largest;
  weighting = fmaxf(1.0f - (weighting / largest), minWeight) * weightMultiplier;
  st->weighting = timed_out ? -1.f : weighting;
  if (timeout_code == 2)
    st->track_range_failures += 1;
  else if (timed_out)
    st->track_timeouts += 1;
#pragma unroll
  for (int i = 0; i < 16; ++i) st->lastPose[i] = P[i];
}

}  // namespace dms
