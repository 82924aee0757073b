// One step of the model-side depth / intensity pyramid (pyrDownGaussF + pyrDownUchar on one pixel), shared by the
// stand-alone kernel (prep.hip) and the tracker's start-of-call kernel (track.hip), which runs the pyramid's last step
// beside its own set-up instead of after a launch boundary of its own.
#pragma once
#include "common.hpp"

namespace dms {

__device__ __forceinline__ float gauss25(int r, int c) {
  const float w[5] = {1.f, 4.f, 6.f, 4.f, 1.f};
  return w[r] * w[c];
}

// float depth (reference pyrDownKernelGaussF, cudafuncs.cu:416-443: NaN-skipping, integer weight count) and intensity
// (pyrDownKernelIntensityGauss, :544-573: zero-skipping) of destination pixel (x, y)
__device__ __forceinline__ void model_pyr_step_pixel(int x, int y, const View<const float>& dsrc, const View<float>& ddst,
                                                     const View<const unsigned char>& isrc, const View<unsigned char>& idst) {
  if (x >= ddst.cols || y >= ddst.rows) return;
  const int D = 5;
  const int tx = min(2 * x - D / 2 + D, dsrc.cols - 1);
  const int ty = min(2 * y - D / 2 + D, dsrc.rows - 1);
  float sum = 0.f, isum = 0.f;
  int count = 0, icount = 0;
  for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
    for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
      const float g = gauss25(ty - cy - 1, tx - cx - 1);
      const float s = dsrc.at(cy, cx);
      if (!isnan(s)) {
        sum += s * g;
        count += (int)g;
      }
      const unsigned char c = isrc.at(cy, cx);
      if (c > 0) {
        isum += (float)c * g;
        icount += (int)g;
      }
    }
  ddst.at(y, x) = sum / (float)count;
  idst.at(y, x) = (unsigned char)f2i_rz(isum / (float)icount);
}

}  // namespace dms
