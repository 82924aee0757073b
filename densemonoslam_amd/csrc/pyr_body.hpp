// One step of the model-side depth / intensity pyramid (pyrDownGaussF + pyrDownUchar on one pixel), shared by the
// stand-alone kernel (prep.hip) and the tracker's start-of-call kernel (track.hip), which runs the pyramid's last step
// beside its own set-up instead of after a launch boundary of its own.  The bodies are live_bodies.hpp's.
#pragma once
#include "common.hpp"
#include "live_bodies.hpp"

namespace dms {

__device__ __forceinline__ void model_pyr_step_pixel(int x, int y, const View<const float>& dsrc, const View<float>& ddst,
                                                     const View<const unsigned char>& isrc, const View<unsigned char>& idst) {
  if (x >= ddst.cols || y >= ddst.rows) return;
  const live::Pitched<float> d = {dsrc.data, (unsigned)dsrc.pitch};
  const live::Pitched<unsigned char> c = {isrc.data, (unsigned)isrc.pitch};
  ddst.at(y, x) = live::float_half(d, x, y, dsrc.cols, dsrc.rows);
  idst.at(y, x) = live::u8_half(c, x, y, isrc.cols, isrc.rows);
}

}  // namespace dms
