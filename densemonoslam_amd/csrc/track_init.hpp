// The tracker call's set-up: prior pose -> TrackState, first projection parameters, all-reduce words of the call's resident
// kernels zeroed (RGBDOdometry.cpp:285-295 does the state part on the host).  A device body shared by the tracker's own first
// kernel (track.hip: k_track_init / k_track_init_pyr) and, on the frame step's path, by a block group of the model pyramid
// kernel (prep.hip: k_model_levels012) — the set-up does not depend on the model pyramid, so it needs no launch of its own.
#pragma once
#include "canon.hpp"
#include "gn_scalar.hpp"
#include "smallmath.hpp"

namespace dms {

struct Prior {
  float v[12];  // trans[3], rot[9] — passed by value so no staging copy can race a later call
};

// block `b` of `nb` blocks of `nt` threads (t = linear thread id)
__device__ __forceinline__ void track_init_body(int b, int nb, int t, int nt, TrackState* st, Prior prior, const float* __restrict__ prior_pose16,
                                                float fx, float fy, float cx, float cy, int so3, int first_level,
                                                unsigned long long* sync_words, int n_sync, int inject_timeout, unsigned* zero16) {
  if (zero16 && b == 0 && t < 16) zero16[t * 16] = 0u;  // (the frame step's dense counters: their reader ran before this launch)
  // all-reduce words of the resident kernels of this call: zero before any of them is launched
  // (n_sync counts 16-byte pairs; the grid shares the work, block 0 also sets up the state)
  {
    ulonglong2* w2 = reinterpret_cast<ulonglong2*>(sync_words);
    for (int i = b * nt + t; i < n_sync; i += nt * nb) w2[i] = make_ulonglong2(0ull, 0ull);
  }
  if (t != 0 || b != 0) return;
  if (prior_pose16) {  // device-resident prior (frame step): row-major 4×4 camera-to-world
    for (int i = 0; i < 3; ++i) {
      prior.v[i] = prior_pose16[i * 4 + 3];
      for (int j = 0; j < 3; ++j) prior.v[3 + i * 3 + j] = prior_pose16[i * 4 + j];
    }
  }
  for (int i = 0; i < 3; ++i) st->tprev[i] = st->tcurr[i] = prior.v[i];
  for (int i = 0; i < 9; ++i) st->Rprev[i] = st->Rcurr[i] = prior.v[3 + i];
  sm::inv3<float>(st->Rprev, st->Rprev_inv);
  for (int i = 0; i < 16; ++i) st->resultRt[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 9; ++i) {
    st->resultR[i] = st->lastResultR[i] = (i % 4 == 0) ? 1.0 : 0.0;
    st->R_lr[i] = (i % 4 == 0) ? 1.f : 0.f;
  }
  st->so3_lastError = 3.402823466e+38F / 2;
  st->so3_lastCount = 3.402823466e+38F / 2;
  st->so3_done = 0;
  st->so3_iters = 0;
  for (int l = 0; l < DMS_NUM_PYRS; ++l) {
    st->level_done[l] = 0;
    st->iters_run[l] = 0;
  }
  st->rejected_jump = 0;
  st->sync_timeout = inject_timeout;  // (0 unless a test injects the fault)
  st->have_E = 0;
  st->canon_retries = 0;
  for (int i = 0; i < 36; ++i) st->lastA[i] = 0.0;
  for (int i = 0; i < 6; ++i) st->lastb[i] = 0.0;
  if (so3) {
    sc::so3_params(st, sc::kpre_of(fx, fy, cx, cy, 2));
  } else {
    double Rt[16];
    for (int i = 0; i < 16; ++i) Rt[i] = st->resultRt[i];
    sc::gn_params(Rt, sc::kpre_of(fx, fy, cx, cy, first_level), st->krkinv, st->kt);
  }
}

// the same as a block group of another kernel: `blocks` blocks of the host kernel's shape (0 = not folded)
struct TrackInitArgs {
  TrackState* st;
  Prior prior;
  const float* prior_pose16;
  float fx, fy, cx, cy;
  int so3, first_level;
  unsigned long long* sync_words;
  int n_sync, inject_timeout;
  int blocks;
};

}  // namespace dms
