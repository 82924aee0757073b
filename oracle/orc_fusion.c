/*
 * orc_fusion.c — CPU ORACLE (test infrastructure) for the surfel-map half of the hot path.
 *
 * A restatement of the reference's GLSL programs and of the OpenGL pipeline state they run under, written the way OpenGL
 * executes them (clear, then one primitive after the other in draw order through a depth-tested framebuffer; transform feedback
 * appends in order) — deliberately NOT the atomic / compaction formulation of the HIP kernels, so the two check each other.
 * (Since round 4 a draw's surfels are split into contiguous ranges that threads rasterise into z-buffers of their own, merged in
 * draw order with the same GL_LESS: the sequential result with any thread count.)  Sources followed:
 *   Core/src/Shaders/{depth_bilateral,depth_metric}.frag, vertex_feedback.{vert,geom},
 *   init_unstable.vert, index_map.{vert,frag}, splat.vert, combo_splat.frag, depth_splat.frag,
 *   data.{vert,geom,frag}, update.vert, copy_unstable.{vert,geom}, fill_{vertex,normal,rgb}.frag,
 *   resize.frag, surfels.glsl, color.glsl, geometry.glsl;
 *   Core/src/{IndexMap.cpp:146-452, GlobalModel.cpp:29-225,266-417,513-853,
 *   ElasticFusion.cpp:84-97,99-637,688-768}, Shaders/{FeedbackBuffer,FillIn,Resize,ComputePack}.cpp.
 *
 * PARITY PINNED (round 4) to the reference's own GLSL programs: oracle/ref_gl_harness.c compiles the shader files where they lie
 * under /root/reference with Mesa's GLSL compiler and runs them on llvmpipe (software OpenGL 4.5, no GPU, no X server);
 * tests/golden/ref_glsl.npz holds what they returned for every stage of a 48x36 case (depth filter, metric depth, first-frame
 * surfels, index map, the three splat predictions, fuse data + update pass, clean plain / with a deformation graph / isFern,
 * fill-in vertex / normal / image, resize, map merge, graph sampling); tests/test_ref_gl_pin_cpu.py holds this file to it and
 * tests/test_ref_gl_live_cpu.py runs both live at two more sizes.  Bar: every decision identical (which surfel, which pixel, kept
 * or removed, merged or new), integer fields exact, floats within a few ulp (llvmpipe's exp / acos / rsqrt are its own), the plain
 * clean, the splat images, the map merge and the graph samples bit for bit; index-map ids may differ only for surfels within
 * 1/64 pixel of a pixel boundary (GL leaves sub-pixel snapping to the implementation).  Not driven: Resize::time (a float sampler
 * on an integer texture: undefined GL).  What GL leaves to the implementation is fixed here by these rules:
 *   R1 NEAREST fetch: texel = clamp(floor(u * n)) with the product in fp32;
 *   R2 point of size 1 owns pixel (floor(xw), floor(yw)), xw = (x_ndc + 1) * (W/2) in fp32;
 *   R3 depth buffer: 24-bit, value round(zw * (2^24-1)), cleared to 2^24-1, test GL_LESS, so
 *      equal depth keeps the earlier primitive;
 *   R4 point sprite of size s >= 1 covers the pixels whose centres lie in [c - s/2, c + s/2);
 *      s < 1 is rasterised as 1;
 *   R5 exp / acos = orc_detmath.h;  normalize(v) = v * (1 / sqrt(v.v));  pow(x, 2) = x * x.
 * See tests/ for the properties that anchor the oracle (brute-force nearest-surfel search,
 * conservation of confidence mass, idempotence of clean without updates).
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"
#include "orc_detmath.h"
#include "orc_fusion.h"

typedef struct { float x, y, z; } v3;
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { /* R5 */
  const float rn = 1.0f / sqrtf(vdot(a, a));
  return V3(a.x * rn, a.y * rn, a.z * rn);
}
/* mat4 * vec4(p, 1) and mat3(M) * v, row-major storage */
static inline v3 m4_point(const float* M, v3 p) {
  return V3(((M[0] * p.x + M[1] * p.y) + M[2] * p.z) + M[3], ((M[4] * p.x + M[5] * p.y) + M[6] * p.z) + M[7],
            ((M[8] * p.x + M[9] * p.y) + M[10] * p.z) + M[11]);
}
static inline v3 m4_dir(const float* M, v3 v) {
  return V3((M[0] * v.x + M[1] * v.y) + M[2] * v.z, (M[4] * v.x + M[5] * v.y) + M[6] * v.z, (M[8] * v.x + M[9] * v.y) + M[10] * v.z);
}
static inline int f2i_rz(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)v;
}
static inline int f2i_rn(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)rintf(v);
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* R1 */
static inline int texel(float u, int n) {
  int i = (int)floorf(u * (float)n);
  if (i < 0) i = 0;
  if (i > n - 1) i = n - 1;
  return i;
}
/* R3 */
static inline unsigned depth24(float zw) {
  if (!(zw >= 0.f) || zw > 1.f) return 0xFFFFFFFFu;
  return (unsigned)llrint((double)zw * 16777215.0);
}

/* general 4x4 float inverse (Eigen Matrix4f::inverse(), e.g. IndexMap.cpp:166): Gauss-Jordan
 * would round differently from the cofactor form the product uses; the pose inverse is an
 * *input* of every kernel compared here, so the same cofactor expansion is restated. */
void orc_inv4f(const float* m, float* o) {
  float inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const float id = 1.0f / det;
  for (int i = 0; i < 16; ++i) o[i] = inv[i] * id;
}

/* ---- surfels.glsl / color.glsl ------------------------------------------------------------ */
static float getRadius(float depth, float norm_z, float cam_z, float cam_w) { /* surfels.glsl:19-34 */
  const float meanFocal = ((1.0f / fabsf(cam_z)) + (1.0f / fabsf(cam_w))) / 2.0f;
  const float sqrt2 = 1.41421356237f;
  const float radius = (depth / meanFocal) * sqrt2;
  float radius_n = radius;
  radius_n = radius_n / fabsf(norm_z);
  radius_n = fminf(2.0f * radius, radius_n);
  return radius_n;
}
static float confidence(float x, float y, float cx, float cy, float weighting) { /* surfels.glsl:36-46 */
  const float maxRadDist = 400;
  const float twoSigmaSquared = 0.72f;
  const float px = x - cx, py = y - cy;
  const float radialDist = sqrtf(px * px + py * py) / maxRadDist;
  return orc_expf((-(radialDist * radialDist) / twoSigmaSquared)) * weighting;
}
static float encodeColor(float cx, float cy, float cz) { /* color.glsl:19-25 */
  int rgb = (int)roundf(cx * 255.0f);
  rgb = (rgb << 8) + (int)roundf(cy * 255.0f);
  rgb = (rgb << 8) + (int)roundf(cz * 255.0f);
  return (float)rgb;
}
static v3 decodeColor(float c) { /* color.glsl:27-34 */
  const int ci = (int)c;
  return V3((float)((ci >> 16) & 0xFF) / 255.0f, (float)((ci >> 8) & 0xFF) / 255.0f, (float)(ci & 0xFF) / 255.0f);
}
/* an RGBA8 texel sampled as normalised floats and re-encoded: the bytes come back exactly */
static float encodeColorBytes(const uint8_t* c) { return (float)(int)((((unsigned)c[0] << 8) + c[1]) * 256u + c[2]); }

/* uv buffer of the reference (GlobalModel.cpp:100-108 / FeedbackBuffer.cpp:38-46) */
static float uv_coord(int i, int n) { return (float)((double)((float)i / (float)n) + 1.0 / (double)(2 * (float)n)); }

/* ---- G1 / G2 --------------------------------------------------------------------------------- */
void orc_depth_bilateral(const uint16_t* src, int rows, int cols, float maxD, uint16_t* dst) { /* depth_bilateral.frag:30-75 */
  const float colsf = (float)cols, rowsf = (float)rows;
  const unsigned gate = (unsigned)f2i_rz(maxD * 1000.0f);
  /* (every output pixel is a function of the input image only: rows are independent, the thread count changes no bit;
   * this filter was 60 % of the oracle's frame time on one thread) */
#pragma omp parallel for schedule(static)
  for (int py = 0; py < rows; ++py)
    for (int px = 0; px < cols; ++px) {
      const unsigned value = src[(size_t)py * cols + px];
      if (value > gate || value < 300U) {
        dst[(size_t)py * cols + px] = 0;
        continue;
      }
      const float tcx = ((float)px + 0.5f) / colsf, tcy = ((float)py + 0.5f) / rowsf; /* fragment centre */
      const int x = (int)(tcx * colsf), y = (int)(tcy * rowsf);
      const float sigma_space2_inv_half = 0.024691358f, sigma_color2_inv_half = 0.000555556f;
      const int R = 6, D = R * 2 + 1;
      const int tx = imin(x - D / 2 + D, cols), ty = imin(y - D / 2 + D, rows);
      float sum1 = 0, sum2 = 0;
      for (int cy = imax(y - D / 2, 0); cy < ty; ++cy)
        for (int cx = imax(x - D / 2, 0); cx < tx; ++cx) {
          const float texX = (float)cx / colsf, texY = (float)cy / rowsf;
          const unsigned tmp = src[(size_t)texel(texY, rows) * cols + texel(texX, cols)];
          const float space2 = ((float)x - (float)cx) * ((float)x - (float)cx) + ((float)y - (float)cy) * ((float)y - (float)cy);
          const float color2 = ((float)value - (float)tmp) * ((float)value - (float)tmp);
          const float weight = orc_expf(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half));
          sum1 += (float)tmp * weight;
          sum2 += weight;
        }
      dst[(size_t)py * cols + px] = (uint16_t)(unsigned)f2i_rz(roundf(sum1 / sum2));
    }
}

void orc_depth_metric(const uint16_t* src, int rows, int cols, float maxD, float* dst) { /* depth_metric.frag:28-39 */
  const unsigned gate = (unsigned)f2i_rz(maxD * 1000.0f);
  for (size_t i = 0; i < (size_t)rows * cols; ++i) {
    const unsigned value = src[i];
    dst[i] = (value > gate || value < 300U) ? 0.f : (float)value / 1000.0f;
  }
}

/* ---- geometry.glsl (float-depth variant) ------------------------------------------------------ */
static v3 getVertexF(const float* depth, int rows, int cols, float tx, float ty, float x, float y, float cx, float cy, float cz,
                     float cw) { /* :21-25 */
  const float z = depth[(size_t)texel(ty, rows) * cols + texel(tx, cols)];
  return V3((x - cx) * z * cz, (y - cy) * z * cw, z);
}
static v3 getNormalF(const float* depth, int rows, int cols, v3 vPosition, float tx, float ty, float x, float y, float cx, float cy,
                     float cz, float cw) { /* :27-39 */
  const float colsf = (float)cols, rowsf = (float)rows;
  const v3 xf = getVertexF(depth, rows, cols, tx + (1.0f / colsf), ty, x + 1, y, cx, cy, cz, cw);
  const v3 xb = getVertexF(depth, rows, cols, tx - (1.0f / colsf), ty, x - 1, y, cx, cy, cz, cw);
  const v3 yf = getVertexF(depth, rows, cols, tx, ty + (1.0f / rowsf), x, y + 1, cx, cy, cz, cw);
  const v3 yb = getVertexF(depth, rows, cols, tx, ty - (1.0f / rowsf), x, y - 1, cx, cy, cz, cw);
  const v3 del_x = V3(((xb.x + vPosition.x) / 2) - ((xf.x + vPosition.x) / 2), ((xb.y + vPosition.y) / 2) - ((xf.y + vPosition.y) / 2),
                      ((xb.z + vPosition.z) / 2) - ((xf.z + vPosition.z) / 2));
  const v3 del_y = V3(((yb.x + vPosition.x) / 2) - ((yf.x + vPosition.x) / 2), ((yb.y + vPosition.y) / 2) - ((yf.y + vPosition.y) / 2),
                      ((yb.z + vPosition.z) / 2) - ((yf.z + vPosition.z) / 2));
  return vnormalize(vcross(del_x, del_y));
}

/* ---- G3 + G4: FeedbackBuffer::compute x2 + GlobalModel::initialise ----------------------------- */
/* time slots that take part in the clean's health test: NUM_CAMERAS of the reference (Shaders/size.glsl:2) */
static int g_num_sensors = 3;
void orc_set_num_sensors(int n) { g_num_sensors = n < 1 ? 1 : (n > ORC_MAX_SENSORS ? ORC_MAX_SENSORS : n); }

typedef struct { float pos[4], col[4], times[ORC_MAX_SENSORS], nrm[4]; } fb_vertex;

/* vertex_feedback.vert:43-75 + .geom:36-49; appends to out in column-major pixel order */
static int feedback_compute(const uint8_t* rgba, const float* depth, int rows, int cols, float cx, float cy, float fx, float fy, int time,
                            int timeIdx, float maxDepth, fb_vertex* out) {
  const float cz = 1.0f / fx, cw = 1.0f / fy; /* FeedbackBuffer.cpp:93-96 */
  int n = 0;
  for (int i = 0; i < cols; i++)
    for (int j = 0; j < rows; j++) {
      const float tx = uv_coord(i, cols), ty = uv_coord(j, rows);
      const float x = tx * (float)cols, y = ty * (float)rows;
      const v3 vp = getVertexF(depth, rows, cols, tx, ty, x, y, cx, cy, cz, cw);
      const uint8_t* c = rgba + 4 * ((size_t)texel(ty, rows) * cols + texel(tx, cols));
      const v3 nl = getNormalF(depth, rows, cols, vp, tx, ty, x, y, cx, cy, cz, cw);
      const float rad = getRadius(vp.z, nl.z, cz, cw);
      const float zVal = (vp.z <= 0 || vp.z > maxDepth) ? 0 : vp.z;
      if (zVal > 0) {
        fb_vertex* o = out + n++;
        o->pos[0] = vp.x; o->pos[1] = vp.y; o->pos[2] = vp.z;
        o->pos[3] = confidence(x, y, cx, cy, 1.0f);
        o->col[0] = encodeColorBytes(c);
        o->col[1] = 0;
        o->col[2] = (float)c[2] / 255.0f; /* vColor.z is the blue channel here; init_unstable overwrites it */
        o->col[3] = (float)time;
        for (int s = 0; s < ORC_MAX_SENSORS; ++s) o->times[s] = s == timeIdx ? (float)time : -3.f;
        o->nrm[0] = nl.x; o->nrm[1] = nl.y; o->nrm[2] = nl.z; o->nrm[3] = rad;
      }
    }
  return n;
}

int orc_model_initialise(const uint8_t* rgba, const float* depth_metric, const float* depth_metric_filtered, int rows, int cols,
                         float cx, float cy, float fx, float fy, int time, int timeIdx, float maxDepth, orc_surfel* out, int cap) {
  fb_vertex* raw = (fb_vertex*)malloc(sizeof(fb_vertex) * (size_t)rows * cols);
  fb_vertex* fil = (fb_vertex*)malloc(sizeof(fb_vertex) * (size_t)rows * cols);
  const int nr = feedback_compute(rgba, depth_metric, rows, cols, cx, cy, fx, fy, time, timeIdx, maxDepth, raw);
  const int nf = feedback_compute(rgba, depth_metric_filtered, rows, cols, cx, cy, fx, fy, time, timeIdx, maxDepth, fil);
  /* GlobalModel::initialise draws rawFeedback.fid vertices; position / colour / times stream from the
   * RAW buffer, normal+radius from the FILTERED buffer, element by element (GlobalModel.cpp:355-392).
   * The two buffers line up only when both depth maps are valid at the same pixels; the caller
   * (tests) checks nr == nf, which holds because both metric maps share one gate. */
  int n = nr < cap ? nr : cap;
  for (int i = 0; i < n; ++i) { /* init_unstable.vert:34-46 */
    orc_surfel* s = out + i;
    const fb_vertex* f = (i < nf) ? fil + i : raw + i;
    memcpy(s->pos, raw[i].pos, sizeof(s->pos));
    s->col[0] = raw[i].col[0]; s->col[1] = 0; s->col[2] = 1; s->col[3] = raw[i].col[3];
    memcpy(s->nrm, f->nrm, sizeof(s->nrm));
    memcpy(s->times, raw[i].times, sizeof(s->times));
  }
  free(raw);
  free(fil);
  return nr == nf ? n : -n - 1; /* negative: the reference would pair mismatched streams */
}

/* ---- G5: IndexMap::predictIndices -------------------------------------------------------------- */
typedef struct {
  float cx, cy, fx, fy, colsf, rowsf, maxDepth;
  int cols, rows;
} proj_t;

/* NDC of the shaders (index_map.vert:56-57, splat.vert:37-42) + viewport transform; GL clips a
 * point by its centre */
static int project_window(const proj_t* a, v3 p, float* xw, float* yw, float* zw) {
  const float xn = ((((a->fx * p.x) / p.z) + a->cx) - (a->colsf * 0.5f)) / (a->colsf * 0.5f);
  const float yn = ((((a->fy * p.y) / p.z) + a->cy) - (a->rowsf * 0.5f)) / (a->rowsf * 0.5f);
  const float zn = p.z / a->maxDepth;
  if (!(xn >= -1.f && xn <= 1.f && yn >= -1.f && yn <= 1.f && zn >= -1.f && zn <= 1.f)) return 0;
  *xw = (xn + 1.f) * (a->colsf * 0.5f);
  *yw = (yn + 1.f) * (a->rowsf * 0.5f);
  *zw = zn * 0.5f + 0.5f;
  return 1;
}

/* The draws below are replayed in two steps that give the sequential result bit for bit with any number of threads: (1) every
 * thread rasterises a contiguous range of the surfels, in order, into a z-buffer of its own that remembers the winning surfel
 * (GL_LESS: the first of equal depths stays); the buffers are merged in thread order with the same strict comparison, so the winner
 * of a pixel is the first surfel of minimal depth, as in one sequential pass; (2) every pixel writes the outputs of its winner,
 * recomputed with the arithmetic that found it. */
static int raster_threads(void) {
  int t = 1;
#ifdef _OPENMP
  t = omp_get_max_threads();
#endif
  return t < 1 ? 1 : (t > 16 ? 16 : t);
}

void orc_index_map(const orc_surfel* model, int M, const float* pose16, float cx, float cy, float fx, float fy, int rows, int cols, int time,
                   int timeIdx, float maxDepth, int timeDelta, uint32_t* index, float* vertConf, float* colorTime, float* normRad) {
  const size_t N = (size_t)rows * cols;
  float t_inv[16];
  orc_inv4f(pose16, t_inv);
  const int T = raster_threads();
  unsigned* zbs = (unsigned*)malloc(N * sizeof(unsigned) * T);
  int* ids = (int*)malloc(N * sizeof(int) * T);
  proj_t a = {cx, cy, fx, fy, (float)cols, (float)rows, maxDepth, cols, rows};
#pragma omp parallel for schedule(static, 1) num_threads(T)
  for (int t = 0; t < T; ++t) {
    unsigned* zb = zbs + (size_t)t * N;
    int* id = ids + (size_t)t * N;
    for (size_t i = 0; i < N; ++i) {
      zb[i] = 0xFFFFFFu; /* glClear depth = 1.0 */
      id[i] = -1;
    }
    const int lo = (int)((long long)M * t / T), hi = (int)((long long)M * (t + 1) / T);
    for (int i = lo; i < hi; ++i) { /* glDrawTransformFeedback: one point per surfel, in order */
      const orc_surfel* s = model + i;
      const v3 ph = m4_point(t_inv, V3(s->pos[0], s->pos[1], s->pos[2]));
      const float vt = s->times[timeIdx];
      if (ph.z > maxDepth || ph.z < 0 || (vt != -3 && time - vt > timeDelta)) continue; /* x = y = -10: clipped */
      float xw, yw, zw;
      if (!project_window(&a, ph, &xw, &yw, &zw)) continue;
      const int px = (int)floorf(xw), py = (int)floorf(yw); /* R2 */
      if (px < 0 || py < 0 || px >= cols || py >= rows) continue;
      const unsigned d = depth24(zw);
      const size_t q = (size_t)py * cols + px;
      if (!(d < zb[q])) continue; /* GL_LESS */
      zb[q] = d;
      id[q] = i;
    }
  }
#pragma omp parallel for schedule(static) num_threads(T)
  for (long long qq = 0; qq < (long long)N; ++qq) {
    const size_t q = (size_t)qq;
    unsigned best = 0xFFFFFFu;
    int win = -1;
    for (int t = 0; t < T; ++t)
      if (zbs[(size_t)t * N + q] < best) {
        best = zbs[(size_t)t * N + q];
        win = ids[(size_t)t * N + q];
      }
    float* vc = vertConf + 4 * q;
    float* ct = colorTime + 4 * q;
    float* nr = normRad + 4 * q;
    if (win < 0) {
      index[q] = 0;
      vc[0] = vc[1] = vc[2] = vc[3] = 0;
      ct[0] = ct[1] = ct[2] = ct[3] = 0;
      nr[0] = nr[1] = nr[2] = nr[3] = 0;
      continue;
    }
    const orc_surfel* s = model + win;
    const v3 ph = m4_point(t_inv, V3(s->pos[0], s->pos[1], s->pos[2]));
    const v3 nh = vnormalize(m4_dir(t_inv, V3(s->nrm[0], s->nrm[1], s->nrm[2])));
    index[q] = (uint32_t)win; /* vertexId = gl_VertexID */
    vc[0] = ph.x; vc[1] = ph.y; vc[2] = ph.z; vc[3] = s->pos[3];
    ct[0] = s->col[0]; ct[1] = s->col[1]; ct[2] = s->col[2]; ct[3] = s->times[timeIdx];
    nr[0] = nh.x; nr[1] = nh.y; nr[2] = nh.z; nr[3] = s->nrm[3];
  }
  free(zbs);
  free(ids);
}

/* ---- G6 / G6': IndexMap::combinedPredict / synthesizeDepth -------------------------------------- */
static v3 projectPointImage(const proj_t* a, v3 p) { /* splat.vert:44-49 */
  return V3(((a->fx * p.x) / p.z) + a->cx, ((a->fy * p.y) / p.z) + a->cy, p.z);
}

static void sprite_range(float c, float size, int n, int* lo, int* hi) { /* R4 */
  const float sz = fmaxf(size, 1.0f);
  const float a = c - sz * 0.5f, b = c + sz * 0.5f;
  *lo = (int)ceilf(a - 0.5f);
  *hi = (int)ceilf(b - 0.5f) - 1;
  if (*lo < 0) *lo = 0;
  if (*hi > n - 1) *hi = n - 1;
}

/* vertex stage of one surfel (splat.vert:57-94); 0 = clipped / culled */
typedef struct { v3 ph, nrm; float rad, conf, xw, yw, size; } splat_v;
static int splat_vertex(const proj_t* a, const float* t_inv, const orc_surfel* s, float maxDepth, float confThreshold, int time, int timeIdx,
                        int maxTime, int timeDelta, int actv, splat_v* o) {
  const v3 ph = m4_point(t_inv, V3(s->pos[0], s->pos[1], s->pos[2]));
  const float vt = s->times[timeIdx], conf = s->pos[3];
  if (!(!actv && vt == -3) &&
      (ph.z > maxDepth || ph.z < 0 || conf < confThreshold || (actv && vt == -3) || (vt != -3 && time - vt > timeDelta) || vt > maxTime))
    return 0; /* gl_Position = 1000: clipped */
  float zw0;
  if (!project_window(a, ph, &o->xw, &o->yw, &zw0)) return 0;
  const v3 nrm = vnormalize(m4_dir(t_inv, V3(s->nrm[0], s->nrm[1], s->nrm[2])));
  const float rad = s->nrm[3];
  const v3 x1n = vnormalize(V3(nrm.y - nrm.z, -nrm.x, nrm.x));
  const v3 x1 = V3(x1n.x * rad * 1.41421356f, x1n.y * rad * 1.41421356f, x1n.z * rad * 1.41421356f);
  const v3 y1 = vcross(nrm, x1);
  const v3 p1 = projectPointImage(a, vadd(ph, x1)), p2 = projectPointImage(a, vadd(ph, y1));
  const v3 p3 = projectPointImage(a, vsub(ph, y1)), p4 = projectPointImage(a, vsub(ph, x1));
  const float xmin = fminf(p1.x, fminf(p2.x, fminf(p3.x, p4.x))), xmax = fmaxf(p1.x, fmaxf(p2.x, fmaxf(p3.x, p4.x)));
  const float ymin = fminf(p1.y, fminf(p2.y, fminf(p3.y, p4.y))), ymax = fmaxf(p1.y, fmaxf(p2.y, fmaxf(p3.y, p4.y)));
  const float size = fmaxf(0, fmaxf(fabsf(xmax - xmin), fabsf(ymax - ymin))); /* gl_PointSize */
  if (!(size == size)) return 0;
  o->ph = ph;
  o->nrm = nrm;
  o->rad = rad;
  o->conf = conf;
  o->size = size;
  return 1;
}
/* fragment stage at (px, py) (combo_splat.frag:35-60 / depth_splat.frag:29-46); 0 = discard */
static int splat_fragment(const splat_v* v, float cx, float cy, float fx, float fy, float maxDepth, int px, int py, v3* corrected, unsigned* d) {
  const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
  const v3 l = vnormalize(V3((fcx - cx) / fx, (fcy - cy) / fy, 1.0f));
  const float k = vdot(v->ph, v->nrm) / vdot(l, v->nrm);
  *corrected = V3(k * l.x, k * l.y, k * l.z);
  const float sqrRad = v->rad * v->rad;
  const v3 diff = vsub(*corrected, v->ph);
  if (vdot(diff, diff) > sqrRad) return 0; /* discard */
  const float zw = (corrected->z / (2 * maxDepth)) + 0.5f; /* gl_FragDepth */
  *d = depth24(zw);
  return 1;
}

void orc_splat_predict(const orc_surfel* model, int M, const float* pose16, float cx, float cy, float fx, float fy, int rows, int cols,
                       float maxDepth, float confThreshold, int time, int timeIdx, int maxTime, int timeDelta, int actv, uint8_t* image,
                       float* vertex, float* normal, uint16_t* timeImg, float* depthOnly) {
  const size_t N = (size_t)rows * cols;
  float t_inv[16];
  orc_inv4f(pose16, t_inv);
  const int T = raster_threads(); /* (two steps, as orc_index_map) */
  unsigned* zbs = (unsigned*)malloc(N * sizeof(unsigned) * T);
  int* ids = (int*)malloc(N * sizeof(int) * T);
  proj_t a = {cx, cy, fx, fy, (float)cols, (float)rows, maxDepth, cols, rows};
#pragma omp parallel for schedule(static, 1) num_threads(T)
  for (int t = 0; t < T; ++t) {
    unsigned* zb = zbs + (size_t)t * N;
    int* id = ids + (size_t)t * N;
    for (size_t i = 0; i < N; ++i) {
      zb[i] = 0xFFFFFFu;
      id[i] = -1;
    }
    const int lo = (int)((long long)M * t / T), hi = (int)((long long)M * (t + 1) / T);
    for (int i = lo; i < hi; ++i) {
      splat_v v;
      if (!splat_vertex(&a, t_inv, model + i, maxDepth, confThreshold, time, timeIdx, maxTime, timeDelta, actv, &v)) continue;
      int x0, x1i, y0, y1i;
      sprite_range(v.xw, v.size, cols, &x0, &x1i);
      sprite_range(v.yw, v.size, rows, &y0, &y1i);
      for (int py = y0; py <= y1i; ++py)
        for (int px = x0; px <= x1i; ++px) {
          v3 corrected;
          unsigned d;
          if (!splat_fragment(&v, cx, cy, fx, fy, maxDepth, px, py, &corrected, &d)) continue;
          const size_t q = (size_t)py * cols + px;
          if (!(d < zb[q])) continue;
          zb[q] = d;
          id[q] = i;
        }
    }
  }
#pragma omp parallel for schedule(static) num_threads(T)
  for (long long qq = 0; qq < (long long)N; ++qq) {
    const size_t q = (size_t)qq;
    unsigned best = 0xFFFFFFu;
    int win = -1;
    for (int t = 0; t < T; ++t)
      if (zbs[(size_t)t * N + q] < best) {
        best = zbs[(size_t)t * N + q];
        win = ids[(size_t)t * N + q];
      }
    if (win < 0) {
      if (depthOnly) {
        depthOnly[q] = 0;
      } else {
        memset(image + 4 * q, 0, 4);
        memset(vertex + 4 * q, 0, 16);
        memset(normal + 4 * q, 0, 16);
        timeImg[q] = 0;
      }
      continue;
    }
    const orc_surfel* s = model + win;
    const int px = (int)(q % (size_t)cols), py = (int)(q / (size_t)cols);
    splat_v v;
    v3 corrected;
    unsigned d;
    (void)splat_vertex(&a, t_inv, s, maxDepth, confThreshold, time, timeIdx, maxTime, timeDelta, actv, &v);
    (void)splat_fragment(&v, cx, cy, fx, fy, maxDepth, px, py, &corrected, &d);
    if (depthOnly) {
      depthOnly[q] = corrected.z;
      continue;
    }
    const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
    const v3 rgb = decodeColor(s->col[0]);
    uint8_t* im = image + 4 * q; /* RGBA8 target: round(c * 255) */
    im[0] = (uint8_t)f2i_rn(rgb.x * 255.0f); im[1] = (uint8_t)f2i_rn(rgb.y * 255.0f); im[2] = (uint8_t)f2i_rn(rgb.z * 255.0f); im[3] = 255;
    const float z = corrected.z;
    float* vv = vertex + 4 * q;
    vv[0] = (fcx - cx) * z * (1.f / fx); vv[1] = (fcy - cy) * z * (1.f / fy); vv[2] = z; vv[3] = v.conf;
    float* n = normal + 4 * q;
    n[0] = v.nrm.x; n[1] = v.nrm.y; n[2] = v.nrm.z; n[3] = v.rad;
    const float tz = s->col[2];
    unsigned tv = tz > 0 ? (unsigned)f2i_rz(tz) : 0u;
    if (tv > 65535u) tv = 65535u;
    timeImg[q] = (uint16_t)tv;
  }
  free(zbs);
  free(ids);
}

/* ---- G7 + G8: GlobalModel::fuse ------------------------------------------------------------------ */
static float angleBetween(v3 a, v3 b) { return orc_acosf(vdot(a, b) / (vlength(a) * vlength(b))); } /* data.vert:66-69 */

int orc_model_fuse(orc_surfel* model, int M, const float* pose16, int time, int timeIdx, const uint8_t* rgba, const float* dr,
                   const float* drf, const uint32_t* index, const float* vertConf, const float* normRad, int rows, int cols, float cx,
                   float cy, float fx, float fy, float maxDepth, float weighting, orc_surfel* newUnstable, int* nNew) {
  const float colsf = (float)cols, rowsf = (float)rows;
  const float cz = (float)(1.0 / (double)fx), cw = (float)(1.0 / (double)fy); /* GlobalModel.cpp:546-550 */
  const float timef = (float)time;
  /* the three 5700^2 "update" textures addressed by surfel id + their depth buffer
   * (GlobalModel.cpp:45-51, 524-531): cleared each call; one slot per surfel is enough */
  float* upPos = (float*)calloc((size_t)(M > 0 ? M : 1) * 4, 4);
  float* upCol = (float*)calloc((size_t)(M > 0 ? M : 1) * 4, 4);
  float* upNrm = (float*)calloc((size_t)(M > 0 ? M : 1) * 4, 4);
  unsigned char* upDepthWritten = (unsigned char*)calloc((size_t)(M > 0 ? M : 1), 1);
  int nn = 0, merged = 0;
  /* the vertex stage of every pixel (independent of the others) in parallel into scratch records; what depends on the draw order -
   * the feedback buffer's sequence and "the first one stays" of the raster - is the sequential walk that follows */
  const size_t NP = (size_t)rows * cols;
  orc_surfel* ev = (orc_surfel*)malloc(NP * sizeof(orc_surfel));
  unsigned* ebest = (unsigned*)malloc(NP * sizeof(unsigned));
  unsigned char* ekind = (unsigned char*)calloc(NP, 1); /* 0: nothing emitted, 1: update, 2: new unstable */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < cols; i++)     /* uv buffer order: column-major (GlobalModel.cpp:100-108) */
    for (int j = 0; j < rows; j++) { /* data.vert:76-192 */
      const float tx = uv_coord(i, cols), ty = uv_coord(j, rows);
      const float x = tx * colsf, y = ty * rowsf;
      const v3 vPosLocal = getVertexF(dr, rows, cols, tx, ty, x, y, cx, cy, cz, cw);
      int updateId = 0;
      unsigned best = 0U;
      float colw = 0;
      int ok = ((int)x % 2 == (int)timef % 2) && ((int)y % 2 == (int)timef % 2);
      if (ok) { /* checkNeighbours, :45-64 */
        ok = !(dr[(size_t)texel(ty, rows) * cols + texel(tx - (1.0f / colsf), cols)] == 0) &&
             !(dr[(size_t)texel(ty - (1.0f / rowsf), rows) * cols + texel(tx, cols)] == 0) &&
             !(dr[(size_t)texel(ty, rows) * cols + texel(tx + (1.0f / colsf), cols)] == 0) &&
             !(dr[(size_t)texel(ty + (1.0f / rowsf), rows) * cols + texel(tx, cols)] == 0);
      }
      if (!(ok && vPosLocal.z > 0 && vPosLocal.z <= maxDepth)) continue; /* nothing emitted (data.geom:41) */
      const v3 vPos = m4_point(pose16, vPosLocal);
      const v3 vPos_f = getVertexF(drf, rows, cols, tx, ty, x, y, cx, cy, cz, cw);
      const uint8_t* c = rgba + 4 * ((size_t)texel(ty, rows) * cols + texel(tx, cols));
      const v3 vNormLocal = getNormalF(drf, rows, cols, vPos_f, tx, ty, x, y, cx, cy, cz, cw);
      const v3 nG = m4_dir(pose16, vNormLocal);
      const float rad = getRadius(vPos_f.z, vNormLocal.z, cz, cw);
      const float conf = confidence(x, y, cx, cy, weighting);
      {
        int counter = 0;
        const float scale = 1.0f; /* IndexMap::FACTOR */
        const float indexXStep = (1.0f / (colsf * scale)) * 0.5f;
        const float indexYStep = (1.0f / (rowsf * scale)) * 0.5f;
        float bestDist = 1000;
        const float windowMultiplier = 2;
        const float xl = (x - cx) * cz, yl = (y - cy) * cw;
        const float lambda = sqrtf(xl * xl + yl * yl + 1);
        const v3 ray = V3(xl, yl, 1);
        for (float wi = tx - (scale * indexXStep * windowMultiplier); wi < tx + (scale * indexXStep * windowMultiplier); wi += indexXStep)
          for (float wj = ty - (scale * indexYStep * windowMultiplier); wj < ty + (scale * indexYStep * windowMultiplier); wj += indexYStep) {
            const size_t q = (size_t)texel(wj, rows) * cols + texel(wi, cols);
            const unsigned current = index[q];
            if (current > 0U) {
              const float* vc = vertConf + 4 * q;
              if (fabsf((vc[2] * lambda) - (vPosLocal.z * lambda)) < 0.05f) {
                const float dist = vlength(vcross(ray, V3(vc[0], vc[1], vc[2]))) / vlength(ray);
                const float* nr = normRad + 4 * q;
                if (dist < bestDist && (fabsf(nr[2]) < 0.75f || fabsf(angleBetween(V3(nr[0], nr[1], nr[2]), vNormLocal)) < 0.5f)) {
                  counter++;
                  bestDist = dist;
                  best = current;
                }
              }
            }
          }
        if (counter > 0) { updateId = 1; colw = -1; } else { updateId = 2; colw = -2; }
      }
      const size_t slot = (size_t)i * rows + j;
      orc_surfel* e = ev + slot;
      e->pos[0] = vPos.x; e->pos[1] = vPos.y; e->pos[2] = vPos.z; e->pos[3] = conf;
      e->col[0] = encodeColorBytes(c); e->col[1] = 0; e->col[2] = timef; e->col[3] = colw;
      e->nrm[0] = nG.x; e->nrm[1] = nG.y; e->nrm[2] = nG.z; e->nrm[3] = rad;
      for (int s = 0; s < ORC_MAX_SENSORS; ++s) e->times[s] = s == timeIdx ? colw : -3.f;
      ebest[slot] = best;
      ekind[slot] = (unsigned char)updateId;
    }
  for (size_t slot = 0; slot < NP; ++slot) { /* the draw order: column-major */
    if (!ekind[slot]) continue;
    /* data.geom:38-59: both kinds are captured by transform feedback into newUnstableVbo */
    orc_surfel* e = newUnstable + nn++;
    *e = ev[slot];
    const unsigned best = ebest[slot];
    /* data.frag + raster: a size-1 point at texel `best`, z = 0, GL_LESS: the first one stays */
    if (ekind[slot] == 1 && best < (unsigned)M && !upDepthWritten[best]) {
      upDepthWritten[best] = 1;
      memcpy(upPos + 4 * best, e->pos, 16);
      memcpy(upCol + 4 * best, e->col, 16);
      memcpy(upNrm + 4 * best, e->nrm, 16);
    }
  }
  free(ev);
  free(ebest);
  free(ekind);
  *nNew = nn;
  /* update.vert:42-104 over every surfel (each one on its own) */
#pragma omp parallel for schedule(static) reduction(+ : merged)
  for (int id = 0; id < M; ++id) {
    const float* newColor = upCol + 4 * id;
    if (newColor[3] == -1) {
      orc_surfel* s = model + id;
      const float* newPos = upPos + 4 * id;
      const float* newNorm = upNrm + 4 * id;
      const float c_k = s->pos[3], a = newPos[3];
      if (newNorm[3] < (1.0f + 0.5f) * s->nrm[3]) {
        const float ws = c_k + a;
        const v3 oldCol = decodeColor(s->col[0]), newCol = decodeColor(newColor[0]);
        s->pos[0] = ((c_k * s->pos[0]) + (a * newPos[0])) / ws;
        s->pos[1] = ((c_k * s->pos[1]) + (a * newPos[1])) / ws;
        s->pos[2] = ((c_k * s->pos[2]) + (a * newPos[2])) / ws;
        s->pos[3] = ws;
        s->col[0] = encodeColor(((c_k * oldCol.x) + (a * newCol.x)) / ws, ((c_k * oldCol.y) + (a * newCol.y)) / ws,
                                ((c_k * oldCol.z) + (a * newCol.z)) / ws);
        v3 n = V3(((c_k * s->nrm[0]) + (a * newNorm[0])) / ws, ((c_k * s->nrm[1]) + (a * newNorm[1])) / ws,
                  ((c_k * s->nrm[2]) + (a * newNorm[2])) / ws);
        s->nrm[3] = ((c_k * s->nrm[3]) + (a * newNorm[3])) / ws;
        n = vnormalize(n);
        s->nrm[0] = n.x; s->nrm[1] = n.y; s->nrm[2] = n.z;
      } else {
        s->pos[3] = c_k + a;
      }
      s->times[timeIdx] = (float)time;
      merged++;
    }
  }
  free(upPos); free(upCol); free(upNrm); free(upDepthWritten);
  return merged;
}

/* ---- G9: GlobalModel::clean ------------------------------------------------------------------------ */
static void inv3f(const float* m, float* o) { /* GLSL inverse(mat3) */
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const float id = 1.0f / det;
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* copy_unstable.vert:53-352 for one vertex; returns `test` */
static int copy_unstable_vertex(orc_surfel* v, const float* t_inv, float cx, float cy, float fx, float fy, int rows, int cols, int time,
                                int timeIdx, float confThreshold, const uint32_t* index, const float* vertConf, const float* colorTime,
                                const float* nodes, int nNodes, const float* depthSynth, float maxDepth, int timeDelta, int isFern) {
  int test = 1;
  const float colsf = (float)cols, rowsf = (float)rows;
  v3 localPos = m4_point(t_inv, V3(v->pos[0], v->pos[1], v->pos[2]));
  float x = ((fx * localPos.x) / localPos.z) + cx;
  float y = ((fy * localPos.y) / localPos.z) + cy;
  const v3 localNorm = vnormalize(m4_dir(t_inv, V3(v->nrm[0], v->nrm[1], v->nrm[2])));
  const float scale = 1.0f;
  const float indexXStep = (1.0f / (colsf * scale)) * 0.5f;
  const float indexYStep = (1.0f / (rowsf * scale)) * 0.5f;
  const float windowMultiplier = 2;
  int count = 0, zCount = 0;
  if (time - v->times[timeIdx] < timeDelta && localPos.z > 0 && x > 0 && y > 0 && x < colsf && y < rowsf) {
    for (float i = x / colsf - (scale * indexXStep * windowMultiplier); i < x / colsf + (scale * indexXStep * windowMultiplier); i += indexXStep)
      for (float j = y / rowsf - (scale * indexYStep * windowMultiplier); j < y / rowsf + (scale * indexYStep * windowMultiplier); j += indexYStep) {
        const size_t q = (size_t)texel(j, rows) * cols + texel(i, cols);
        const unsigned current = index[q];
        if (current > 0U) {
          const float* vc = vertConf + 4 * q;
          const float* ct = colorTime + 4 * q;
          const float dx = vc[0] - localPos.x, dy = vc[1] - localPos.y;
          if (ct[2] < v->col[2] && vc[3] > confThreshold && vc[2] > localPos.z && vc[2] - localPos.z < 0.01f &&
              sqrtf(dx * dx + dy * dy) < v->nrm[3] * 1.4f)
            count++;
          if (ct[3] == time && vc[3] > confThreshold && vc[2] > localPos.z && vc[2] - localPos.z > 0.01f && fabsf(localNorm.z) > 0.85f)
            zCount++;
        }
      }
  }
  if (count > 8 || zCount > 4) test = 0;
  if (v->times[timeIdx] == -2) { /* new unstable point */
    v->col[3] = (float)time;
    v->times[timeIdx] = (float)time;
  }
  /* copy_unstable.vert:137-150: the loop runs over vTimes.length() = NUM_CAMERAS (3, Shaders/size.glsl:2) */
  int unHealthy = 0;
  for (int i = 0; i < g_num_sensors; i++)
    if (v->times[i] == -1 || ((time - v->times[i]) > 20 && v->pos[3] < confThreshold)) unHealthy++;
  if (unHealthy == g_num_sensors) test = 0;
  if (v->times[timeIdx] > 0 && time - v->times[timeIdx] > timeDelta) test = 1;

  if (test == 1 && nNodes > 0 && v->col[2] != time) { /* :161-351 */
    const int k = 4, lookBack = 20;
    int nearNodes[20];
    float nearDists[20];
    for (int i = 0; i < lookBack; i++) { nearNodes[i] = -1; nearDists[i] = 16777216.0f; }
    const int poseTime = (int)v->col[2];
    int foundIndex = 0, imin_ = 0, imax_ = nNodes - 1, imid = (imin_ + imax_) / 2;
    while (imax_ >= imin_) {
      imid = (imin_ + imax_) / 2;
      const int nodeTime = (int)nodes[imid * 16 + 15];
      if (nodeTime < poseTime) imin_ = imid + 1;
      else if (nodeTime > poseTime) imax_ = imid - 1;
      else break;
    }
    imin_ = imin(imin_, nNodes - 1);
    const int nodeMin = (int)nodes[imin_ * 16 + 15];
    const int nodeMid = (int)nodes[imid * 16 + 15];
    const int nodeMax = imax_ < 0 ? (int)nodes[0] : (int)nodes[imax_ * 16 + 15]; /* negative coordinate clamps to texel 0 */
    if (abs(nodeMin - poseTime) <= abs(nodeMid - poseTime) && abs(nodeMin - poseTime) <= abs(nodeMax - poseTime)) foundIndex = imin_;
    else if (abs(nodeMid - poseTime) <= abs(nodeMin - poseTime) && abs(nodeMid - poseTime) <= abs(nodeMax - poseTime)) foundIndex = imid;
    else foundIndex = imax_;
    if (foundIndex == nNodes) foundIndex = nNodes - 1;
    if (foundIndex < 0) foundIndex = 0; /* the j-loops below would not run; keep a defined index */
    const v3 P = V3(v->pos[0], v->pos[1], v->pos[2]);
    int nearNodeIndex = 0, distanceBack = 0;
    for (int j = foundIndex; j >= 0; j--) {
      const v3 d = vsub(P, V3(nodes[j * 16], nodes[j * 16 + 1], nodes[j * 16 + 2]));
      nearNodes[nearNodeIndex] = j;
      nearDists[nearNodeIndex] = sqrtf(vdot(d, d));
      nearNodeIndex++;
      if (++distanceBack == lookBack / 2) break;
    }
    for (int j = foundIndex + 1; j < nNodes; j++) {
      const v3 d = vsub(P, V3(nodes[j * 16], nodes[j * 16 + 1], nodes[j * 16 + 2]));
      nearNodes[nearNodeIndex] = j;
      nearDists[nearNodeIndex] = sqrtf(vdot(d, d));
      nearNodeIndex++;
      if (++distanceBack == lookBack) break;
    }
    for (int i = 0; i < lookBack - 1; ++i)
      for (int j = i + 1; j < lookBack; ++j)
        if (nearDists[j] < nearDists[i]) {
          const float t = nearDists[i]; nearDists[i] = nearDists[j]; nearDists[j] = t;
          const int t2 = nearNodes[i]; nearNodes[i] = nearNodes[j]; nearNodes[j] = t2;
        }
    const float dMax = nearDists[k];
    float nodeWeights[4], weightSum = 0;
    for (int j = 0; j < k; j++) {
      const int nj = imax(nearNodes[j], 0);
      const v3 d = vsub(P, V3(nodes[nj * 16], nodes[nj * 16 + 1], nodes[nj * 16 + 2]));
      const float w1 = 1.0f - (sqrtf(vdot(d, d)) / dMax);
      nodeWeights[j] = w1 * w1;
      weightSum += nodeWeights[j];
    }
    for (int j = 0; j < k; j++) nodeWeights[j] /= weightSum;
    v3 newPos = V3(0, 0, 0), newNorm = V3(0, 0, 0);
    const v3 Nn = V3(v->nrm[0], v->nrm[1], v->nrm[2]);
    for (int i = 0; i < k; i++) {
      const float* q = nodes + imax(nearNodes[i], 0) * 16;
      const v3 g = V3(q[0], q[1], q[2]);
      const float R[9] = {q[3], q[6], q[9], q[4], q[7], q[10], q[5], q[8], q[11]}; /* mat3(column0, column1, column2) */
      const v3 tr = V3(q[12], q[13], q[14]);
      const v3 d = vsub(P, g);
      const v3 Rd = V3((R[0] * d.x + R[1] * d.y) + R[2] * d.z, (R[3] * d.x + R[4] * d.y) + R[5] * d.z, (R[6] * d.x + R[7] * d.y) + R[8] * d.z);
      const v3 cand = vadd(vadd(Rd, g), tr);
      newPos = vadd(newPos, V3(nodeWeights[i] * cand.x, nodeWeights[i] * cand.y, nodeWeights[i] * cand.z));
      float Ri[9];
      inv3f(R, Ri);
      const v3 nn = V3((Ri[0] * Nn.x + Ri[3] * Nn.y) + Ri[6] * Nn.z, (Ri[1] * Nn.x + Ri[4] * Nn.y) + Ri[7] * Nn.z,
                       (Ri[2] * Nn.x + Ri[5] * Nn.y) + Ri[8] * Nn.z); /* transpose(inverse(R)) * n */
      newNorm = vadd(newNorm, V3(nodeWeights[i] * nn.x, nodeWeights[i] * nn.y, nodeWeights[i] * nn.z));
    }
    v->pos[0] = newPos.x; v->pos[1] = newPos.y; v->pos[2] = newPos.z;
    const v3 nn = vnormalize(newNorm);
    v->nrm[0] = nn.x; v->nrm[1] = nn.y; v->nrm[2] = nn.z;
    if (v->pos[3] > confThreshold && isFern == 0) {
      localPos = m4_point(t_inv, V3(v->pos[0], v->pos[1], v->pos[2]));
      x = ((fx * localPos.x) / localPos.z) + cx;
      y = ((fy * localPos.y) / localPos.z) + cy;
      if (localPos.z > 0 && localPos.z < maxDepth && x > 0 && y > 0 && x < colsf && y < rowsf) {
        const float currentDepth = depthSynth ? depthSynth[(size_t)texel(y / rowsf, rows) * cols + texel(x / colsf, cols)] : 0.f;
        if (currentDepth > 0.0f && localPos.z < currentDepth + 0.1f) {
          v->col[3] = (float)time;
          v->times[timeIdx] = (float)time;
        }
      }
    }
  }
  return test;
}

int orc_model_clean(const orc_surfel* model, int M, const orc_surfel* newUnstable, int nNew, const float* pose16, int time, int timeIdx,
                    const uint32_t* index, const float* vertConf, const float* colorTime, const float* depthSynth, int rows, int cols,
                    float cx, float cy, float fx, float fy, float confThreshold, const float* nodes, int nNodes, int timeDelta,
                    float maxDepth, int isFern, orc_surfel* out, int cap) {
  float t_inv[16];
  orc_inv4f(pose16, t_inv);
  int n = 0;
  for (int pass = 0; pass < 2; ++pass) { /* glDrawTransformFeedback(model) then (newUnstableFid), GlobalModel.cpp:799,823 */
    const orc_surfel* src = pass == 0 ? model : newUnstable;
    const int cnt = pass == 0 ? M : nNew;
    /* every vertex is a function of its own record and of the (read-only) index-map images: the vertex stage runs over all of them
     * in parallel into a scratch copy, the geometry stage's stream-out order is the sequential compaction that follows - the
     * thread count changes no bit */
    orc_surfel* tmp = (orc_surfel*)malloc((size_t)(cnt > 0 ? cnt : 1) * sizeof(orc_surfel));
    unsigned char* keep = (unsigned char*)malloc((size_t)(cnt > 0 ? cnt : 1));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < cnt; ++i) {
      tmp[i] = src[i];
      keep[i] = copy_unstable_vertex(&tmp[i], t_inv, cx, cy, fx, fy, rows, cols, time, timeIdx, confThreshold, index, vertConf, colorTime, nodes,
                                     nNodes, depthSynth, maxDepth, timeDelta, isFern) > 0;
    }
    for (int i = 0; i < cnt; ++i)
      if (keep[i] && n < cap) out[n++] = tmp[i]; /* copy_unstable.geom:36-50 */
    free(tmp);
    free(keep);
  }
  return n;
}

/* ---- G10 / G11 ---------------------------------------------------------------------------------------- */
static v3 fillVertex(const uint16_t* depth, int rows, int cols, float tx, float ty, int x, int y, float cx, float cy, float cz, float cw) {
  const float z = (float)depth[(size_t)texel(ty, rows) * cols + texel(tx, cols)] / 1000.0f; /* geometry.glsl:41-45 */
  return V3((x - cx) * z * cz, (y - cy) * z * cw, z);
}

void orc_fill_in(const float* exVertex, const float* exNormal, const uint8_t* exImage, const uint16_t* depth, const uint8_t* rgba, int rows,
                 int cols, float cx, float cy, float fx, float fy, int passGeom, int passRgb, float* outVertex, float* outNormal,
                 uint8_t* outImage) {
  const float colsf = (float)cols, rowsf = (float)rows;
  const float cz = 1.0f / fx, cw = 1.0f / fy; /* FillIn.cpp:120-123 */
  for (int py = 0; py < rows; ++py)
    for (int px = 0; px < cols; ++px) {
      const size_t i = (size_t)py * cols + px;
      const float tx = ((float)px + 0.5f) / colsf, ty = ((float)py + 0.5f) / rowsf;
      const int x = (int)(tx * colsf), y = (int)(ty * rowsf);
      { /* fill_vertex.frag:41-55 */
        const float* samp = exVertex + 4 * i;
        if (samp[2] == 0 || passGeom == 1) {
          const v3 v = fillVertex(depth, rows, cols, tx, ty, x, y, cx, cy, cz, cw);
          outVertex[4 * i] = v.x; outVertex[4 * i + 1] = v.y; outVertex[4 * i + 2] = v.z; outVertex[4 * i + 3] = 1;
        } else {
          memcpy(outVertex + 4 * i, samp, 16);
        }
      }
      { /* fill_normal.frag:33-48 */
        const float* samp = exNormal + 4 * i;
        if (samp[2] == 0 || passGeom == 1) {
          const v3 v = fillVertex(depth, rows, cols, tx, ty, x, y, cx, cy, cz, cw);
          const v3 vx = fillVertex(depth, rows, cols, tx + (1.0f / colsf), ty, x + 1, y, cx, cy, cz, cw);
          const v3 vy = fillVertex(depth, rows, cols, tx, ty + (1.0f / rowsf), x, y + 1, cx, cy, cz, cw);
          const v3 n = vnormalize(vcross(vsub(vx, v), vsub(vy, v)));
          outNormal[4 * i] = n.x; outNormal[4 * i + 1] = n.y; outNormal[4 * i + 2] = n.z; outNormal[4 * i + 3] = 1;
        } else {
          memcpy(outNormal + 4 * i, samp, 16);
        }
      }
      { /* fill_rgb.frag:29-37 */
        const uint8_t* samp = exImage + 4 * i;
        if ((samp[0] == 0 && samp[1] == 0 && samp[2] == 0) || passRgb == 1) memcpy(outImage + 4 * i, rgba + 4 * i, 4);
        else memcpy(outImage + 4 * i, samp, 4);
      }
    }
}

void orc_resize_nn(const void* src, int srows, int scols, void* dst, int drows, int dcols, int elem) { /* resize.frag */
  for (int j = 0; j < drows; ++j)
    for (int i = 0; i < dcols; ++i) {
      const float u = ((float)i + 0.5f) / (float)dcols, v = ((float)j + 0.5f) / (float)drows;
      const size_t s = (size_t)texel(v, srows) * scols + texel(u, scols);
      memcpy((char*)dst + ((size_t)j * dcols + i) * elem, (const char*)src + s * elem, elem);
    }
}

/* ElasticFusion::denseEnough over Resize::image (ElasticFusion.cpp:84-97,166-167) */
int orc_dense_enough(const uint8_t* image_rgba, int rows, int cols) {
  const int dw = cols / 20, dh = rows / 20;
  int sum = 0;
  for (int j = 0; j < dh; ++j)
    for (int i = 0; i < dw; ++i) {
      const float u = ((float)i + 0.5f) / (float)dw, v = ((float)j + 0.5f) / (float)dh;
      const uint8_t* c = image_rgba + 4 * ((size_t)texel(v, rows) * cols + texel(u, cols));
      sum += c[0] > 0 && c[1] > 0 && c[2] > 0;
    }
  return (float)sum / (float)(dh * dw) > 0.95f;
}

/* velocity weight, ElasticFusion.cpp:252-268 (+ rodrigues2 :941-985 without the SVD re-orthonormalisation) */
float orc_velocity_weight(const float* currPose16, const float* lastPose16, float weightMultiplier) {
  float inv[16], diff[16];
  orc_inv4f(currPose16, inv);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = inv[i * 4 + 0] * lastPose16[0 * 4 + j];
      s += inv[i * 4 + 1] * lastPose16[1 * 4 + j];
      s += inv[i * 4 + 2] * lastPose16[2 * 4 + j];
      s += inv[i * 4 + 3] * lastPose16[3 * 4 + j];
      diff[i * 4 + j] = s;
    }
  const float tn = sqrtf(diff[3] * diff[3] + diff[7] * diff[7] + diff[11] * diff[11]);
  double rx = (double)diff[9] - (double)diff[6], ry = (double)diff[2] - (double)diff[8], rz = (double)diff[4] - (double)diff[1];
  const double sn = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = ((double)(diff[0] + diff[5] + diff[10]) - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  const double theta = acos(c);
  double rn;
  if (sn < 1e-5) {
    rn = c > 0 ? 0.0 : theta;
  } else {
    const double vth = (1 / (2 * sn)) * theta;
    rx *= vth; ry *= vth; rz *= vth;
    rn = sqrt(rx * rx + ry * ry + rz * rz);
  }
  float weighting = fmaxf(tn, (float)rn);
  const float largest = 0.01f, minWeight = 0.5f;
  if (weighting > largest) weighting = largest;
  return fmaxf(1.0f - (weighting / largest), minWeight) * weightMultiplier;
}
