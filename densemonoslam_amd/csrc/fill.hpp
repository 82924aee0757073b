// Hole fill-in of a model prediction (G10: fill_vertex.frag / fill_normal.frag / fill_rgb.frag), per pixel.
// Shared by the stand-alone kernel (fusion_pre.hip) and the prediction's resolve pass (fusion_map.hip), which fills
// the pixel it has just resolved instead of leaving that to a second launch over the same images.
#pragma once
#include "surfel.hpp"

namespace dms {

struct FillArgs {
  const float4* ex_vertex;
  const float4* ex_normal;
  const uchar4* ex_image;
  const unsigned short* depth;  // filtered, mm
  const uchar4* rgba;
  // optional: block (0,0) also copies `mirror_words` dwords (the frame's result block into its pinned
  // host mirror — the frame step's last kernel does the copy a separate blit launch would do)
  const unsigned* mirror_src;
  unsigned* mirror_dst;
  int mirror_words;
  // optional: denseEnough (ElasticFusion.cpp:84-97,166-167) from the existing image: the share of non-black pixels on
  // its W/20 x H/20 NEAREST subsample (Resize::image); *dense_flag = 0 when more than 95 % are covered, 1 (fill in)
  // otherwise.  Independent of the fill-in itself.  Stand-alone kernel: one extra block (grid row `rows_blocks`).
  int* dense_flag;
  int rows_blocks;
  // resolve pass: the threads that own a subsampled pixel (bit masks of its columns, words 0..63, and rows, words
  // 64..127: fill_sample_masks) add their non-black pixel to one of 16 counters (64 bytes apart, zero on entry); the
  // consumer of the decision sums them (dense_from_counters) — no block has to wait for the others inside the launch
  unsigned* dense_cnt;
  const unsigned* sample_mask;
  // optional (resolve pass): the frame block of collaborative mode (dms_fusion_arm_frame_block) written by the threads that own a
  // sample of the W/8 x H/8 NEAREST thumbnails of the FILLED image / vertex / normal (what k_thumbnails reads back in a launch of its
  // own): thumb_mask = column bits (words 0..63) | row bits (64..127) | samples in the words before (128..191 columns, 192..255
  // rows) - thumb_sample_masks; block 0 also copies the camera's pose and stores the tick
  unsigned char* thumb_block;
  const unsigned* thumb_mask;
  int thumb_w, thumb_h;
  const float* thumb_pose_src;
  float* thumb_pose_dst;
  int* thumb_tick_dst;
  int thumb_tick;
  float4* out_vertex;
  float4* out_normal;
  uchar4* out_image;
  int cols, rows;
  float cx, cy, ifx, ify;  // cam = (cx, cy, 1/fx, 1/fy) with float reciprocals (FillIn.cpp:120-123)
  int pass_geom, pass_rgb;
};

__device__ __forceinline__ f3 fill_vertex_at(const FillArgs& a, int sx, int sy, int x, int y) {
  // geometry.glsl:41-45 (usampler2D variant): z = texel / 1000
  const float z = (float)a.depth[(size_t)sy * a.cols + sx] / 1000.0f;
  return mk3((((float)x - a.cx) * z) * a.ifx, (((float)y - a.cy) * z) * a.ify, z);
}

// pixel (px, py) whose existing (predicted) values are sv / sn / si; ov / on / oi = what was stored
__device__ __forceinline__ void fill_pixel_out(const FillArgs& a, int px, int py, const float4& sv, const float4& sn, const uchar4& si, float4& ov,
                                               float4& on, uchar4& oi) {
  const size_t i = (size_t)py * a.cols + px;
  const float colsf = (float)a.cols, rowsf = (float)a.rows;
  const float tcx = ((float)px + 0.5f) / colsf, tcy = ((float)py + 0.5f) / rowsf;
  const int x = (int)(tcx * colsf), y = (int)(tcy * rowsf);
  {  // fill_vertex.frag:41-55
    if (sv.z == 0.f || a.pass_geom == 1) {
      const f3 v = fill_vertex_at(a, px, py, x, y);
      ov = make_float4(v.x, v.y, v.z, 1.f);
    } else {
      ov = sv;
    }
    a.out_vertex[i] = ov;
  }
  {  // fill_normal.frag:33-48 with geometry.glsl:48-58 forward differences
    if (sn.z == 0.f || a.pass_geom == 1) {
      const f3 v = fill_vertex_at(a, px, py, x, y);
      const int sxp = texel(tcx + (1.0f / colsf), colsf, a.cols);
      const int syp = texel(tcy + (1.0f / rowsf), rowsf, a.rows);
      const f3 vx = fill_vertex_at(a, sxp, py, x + 1, y);
      const f3 vy = fill_vertex_at(a, px, syp, x, y + 1);
      const f3 n = normalized3(cross3(vx - v, vy - v));
      on = make_float4(n.x, n.y, n.z, 1.f);
    } else {
      on = sn;
    }
    a.out_normal[i] = on;
  }
  {  // fill_rgb.frag:29-37: samp.x + samp.y + samp.z == 0 on normalised bytes <=> all three zero
    if ((si.x == 0 && si.y == 0 && si.z == 0) || a.pass_rgb == 1)
      oi = a.rgba[i];
    else
      oi = si;
    a.out_image[i] = oi;
  }
}
__device__ __forceinline__ void fill_pixel(const FillArgs& a, int px, int py, const float4& sv, const float4& sn, const uchar4& si) {
  float4 ov, on;
  uchar4 oi;
  fill_pixel_out(a, px, py, sv, sn, si, ov, on, oi);
}

// the frame block's thumbnails from the pixel just filled (a.thumb_block != nullptr): NEAREST as resize.frag, the sample of thumbnail
// texel (i, j) is source texel (texel((i + .5) / tw), texel((j + .5) / th)) - the thread that owns it stores it
__device__ __forceinline__ void fill_thumb(const FillArgs& a, int px, int py, const float4& ov, const float4& on, const uchar4& oi) {
  const unsigned cw = a.thumb_mask[px >> 5], rw = a.thumb_mask[64 + (py >> 5)];
  if (!((cw >> (px & 31)) & (rw >> (py & 31)) & 1u)) return;
  const int i = (int)a.thumb_mask[128 + (px >> 5)] + __popc(cw & ((1u << (px & 31)) - 1u));
  const int j = (int)a.thumb_mask[192 + (py >> 5)] + __popc(rw & ((1u << (py & 31)) - 1u));
  const size_t n = (size_t)a.thumb_w * a.thumb_h, k = (size_t)j * a.thumb_w + i;
  reinterpret_cast<uchar4*>(a.thumb_block)[k] = oi;
  reinterpret_cast<float4*>(a.thumb_block + thumb_vertex_off(n))[k] = ov;
  reinterpret_cast<float4*>(a.thumb_block + thumb_normal_off(n))[k] = on;
}

// host: masks + prefix counts of the W/8 x H/8 NEAREST subsample (cols, rows <= 2048), 256 words; false when two thumbnail texels
// would sample the same source texel (the owner-writes form needs distinct owners: sizes below 8 x 8 per sample never occur)
inline bool thumb_sample_masks(int cols, int rows, unsigned* words256) {
  for (int i = 0; i < 256; ++i) words256[i] = 0u;
  const int tw = cols / 8, th = rows / 8;
  if (tw < 1 || th < 1 || cols > 2048 || rows > 2048) return false;
  int last = -1;
  for (int i = 0; i < tw; ++i) {
    const int sx = texel(((float)i + 0.5f) / (float)tw, (float)cols, cols);
    if (sx <= last) return false;
    last = sx;
    words256[sx >> 5] |= 1u << (sx & 31);
  }
  last = -1;
  for (int j = 0; j < th; ++j) {
    const int sy = texel(((float)j + 0.5f) / (float)th, (float)rows, rows);
    if (sy <= last) return false;
    last = sy;
    words256[64 + (sy >> 5)] |= 1u << (sy & 31);
  }
  for (int w = 1; w < 64; ++w) {
    words256[128 + w] = words256[128 + w - 1] + (unsigned)__builtin_popcount(words256[w - 1]);
    words256[192 + w] = words256[192 + w - 1] + (unsigned)__builtin_popcount(words256[64 + w - 1]);
  }
  return true;
}

// host: the column / row masks of the W/20 x H/20 NEAREST subsample (cols, rows <= 2048), 128 words
inline void fill_sample_masks(int cols, int rows, unsigned* words128) {
  for (int i = 0; i < 128; ++i) words128[i] = 0u;
  const int dw = cols / 20, dh = rows / 20;
  for (int i = 0; i < dw; ++i) {
    const int sx = texel(((float)i + 0.5f) / (float)dw, (float)cols, cols);
    words128[sx >> 5] |= 1u << (sx & 31);
  }
  for (int j = 0; j < dh; ++j) {
    const int sy = texel(((float)j + 0.5f) / (float)dh, (float)rows, rows);
    words128[64 + (sy >> 5)] |= 1u << (sy & 31);
  }
}

// the denseEnough decision by one block of NT threads (t = linear thread id)
template <int NT>
__device__ __forceinline__ void fill_dense_test(const FillArgs& a, int t, int* s_sum /* [NT / 64] in LDS */) {
  const int dw = a.cols / 20, dh = a.rows / 20;
  int sum = 0;
  for (int k = t; k < dw * dh; k += NT) {
    const int i = k % dw, j = k / dw;
    const float u = ((float)i + 0.5f) / (float)dw, v = ((float)j + 0.5f) / (float)dh;
    const int sx = texel(u, (float)a.cols, a.cols), sy = texel(v, (float)a.rows, a.rows);
    const uchar4 c = a.ex_image[(size_t)sy * a.cols + sx];
    sum += (c.x > 0 && c.y > 0 && c.z > 0) ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if ((t & 63) == 0) s_sum[t >> 6] = sum;
  __syncthreads();
  if (t == 0) {
    int tot = 0;
    for (int w = 0; w < NT / 64; ++w) tot += s_sum[w];
    const bool dense = (float)tot / (float)(dh * dw) > 0.95f;
    *a.dense_flag = dense ? 0 : 1;
  }
}

// the same decision from the 16 counters of the resolve pass: 1 = fill in (not dense enough)
__device__ __forceinline__ int dense_from_counters(const unsigned* cnt, int samples) {
  unsigned tot = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) tot += cnt[k * 16];
  return ((float)(int)tot / (float)samples > 0.95f) ? 0 : 1;
}

}  // namespace dms
