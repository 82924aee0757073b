// Surfel map: storage, bootstrap from the first frame (G3+G4), index map (G5), splat
// prediction (G6, G6').
//
// The reference renders the map through OpenGL (one point per surfel, depth-tested MRT).
// Here every "draw" is two HBM passes:
//   pass 1 (one thread per surfel): cull, project, and fight for each covered pixel with a
//          64-bit atomicMin on key = depth24 << 32 | surfel id — GL_LESS on a 24-bit depth
//          buffer with first-drawn-wins ties is exactly "smallest key";
//   pass 2 (one thread per pixel): decode the winner and recompute its attributes.
// Pass 1 streams only the position/normal planes; the colour plane is touched for winners only.
#include <vector>

#include "exact_arith.hpp"
#include "scan.hpp"
#include "smallmath.hpp"
#include "surfel.hpp"
#include "fill.hpp"
#include "track_init.hpp"

namespace dms {
int model_flush_pending(dms_model* m, hipStream_t s);  // fusion_fuse.hip: applies a fuse's deferred update pass

// ---------------------------------------------------------------------------------------
// storage
// ---------------------------------------------------------------------------------------
static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct ModelCarver {
  char* base = nullptr;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = up256(off);
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

static void model_layout(dms_model* m, ModelCarver& c) {
  for (int b = 0; b < 2; ++b) {
    m->buf[b].pos = c.take<float4>(m->cap);
    m->buf[b].col = c.take<float4>(m->cap);
    m->buf[b].nrm = c.take<float4>(m->cap);
    m->buf[b].times = c.take<float>(m->cap * DMS_MAX_SENSORS);
  }
  m->d_count = c.take<unsigned>(8);
  m->d_count_alt = m->d_count ? m->d_count + 4 : nullptr;
  m->slot_pos = c.take<float4>(m->slots);
  m->slot_col = c.take<float4>(m->slots);
  m->slot_nrm = c.take<float4>(m->slots);
  m->slot_best = c.take<unsigned>(m->slots);
  m->slot_flag = c.take<unsigned char>(m->slots);
  m->winner = c.take<unsigned>(m->cap);
  const size_t total = m->cap + (size_t)m->width * m->height;  // clean: cap + slots; bootstrap: W*H
  m->keep = c.take<unsigned char>(total);
  m->block_count = c.take<unsigned>(total / kScanChunk + 2);
  m->block_offset = c.take<unsigned>(total / kScanChunk + 2);
  m->clean_first = c.take<unsigned>(4);
  m->nodes = c.take<float>((size_t)m->max_nodes * 16);
}

__global__ void k_fill_u32(unsigned* p, size_t n, unsigned v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)blockDim.x * gridDim.x) p[i] = v;
}

}  // namespace dms

using namespace dms;

extern "C" {

int dms_model_create(dms_model** out, size_t capacity, int width, int height) {
  DMS_REQUIRE(out && width >= 16 && height >= 16, "bad argument");
  dms_model* m = new dms_model();
  m->cap = capacity ? capacity : (size_t)5700 * 5700;  // GlobalModel::MAX_VERTICES
  m->width = width;
  m->height = height;
  m->slot_h = (height + 1) / 2;
  m->slots = ((width + 1) / 2) * m->slot_h;
  ModelCarver sz;
  model_layout(m, sz);
  m->arena_bytes = up256(sz.off);
  hipError_t e = hipMalloc((void**)&m->arena, m->arena_bytes);
  if (e != hipSuccess) {
    delete m;
    return hip_fail(e, "hipMalloc(model)", __FILE__, __LINE__);
  }
  ModelCarver c;
  c.base = m->arena;
  model_layout(m, c);
  e = hipHostMalloc((void**)&m->h_count, 8 * sizeof(unsigned), hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipFree(m->arena);
    delete m;
    return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
  }
  DMS_HIP(hipMemset(m->d_count, 0, 8 * sizeof(unsigned)));
  hipLaunchKernelGGL(k_fill_u32, dim3(1024), dim3(256), 0, 0, m->winner, m->cap, kEmptyWinner);
  DMS_CHECK_LAUNCH();
  for (int b = 0; b < 2; ++b) {  // every time plane of both buffers: -3.0f, "never seen by this sensor" (surfel.hpp, live_planes)
    hipLaunchKernelGGL(k_fill_u32, dim3(2048), dim3(256), 0, 0, (unsigned*)m->buf[b].times, m->cap * DMS_MAX_SENSORS, 0xC0400000u);
    DMS_CHECK_LAUNCH();
  }
  DMS_HIP(hipDeviceSynchronize());
  m->count_upper = 0;
  if (const char* smin = getenv("DMS_CLEAN_SUFFIX_MIN")) m->clean_suffix_min = (size_t)atoll(smin);  // once, at creation
  *out = m;
  return DMS_OK;
}

int dms_model_set_num_sensors(dms_model* m, int num_sensors) {
  DMS_REQUIRE(m, "null argument");
  DMS_REQUIRE(num_sensors >= 1 && num_sensors <= DMS_MAX_SENSORS, "num_sensors out of range");
  m->num_sensors = num_sensors;
  return DMS_OK;
}

int dms_model_set_clean_suffix_min(dms_model* m, size_t surfels) {
  DMS_REQUIRE(m, "null argument");
  m->clean_suffix_min = surfels;
  return DMS_OK;
}

int dms_model_destroy(dms_model* m) {
  if (!m) return DMS_OK;
  if (m->arena) (void)hipFree(m->arena);
  if (m->h_count) (void)hipHostFree(m->h_count);
  delete m;
  return DMS_OK;
}

size_t dms_model_capacity(dms_model* m) { return m ? m->cap : 0; }
size_t dms_model_count_bound(dms_model* m) { return m ? m->count_upper : 0; }

int dms_model_count(dms_model* m, unsigned int* count, dms_stream s) {
  DMS_REQUIRE(m && count, "null argument");
  DMS_HIP(hipMemcpyAsync(m->h_count, m->d_count, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)s));
  DMS_HIP(hipStreamSynchronize((hipStream_t)s));
  *count = m->h_count[0];
  m->count_upper = *count;
  return DMS_OK;
}

static int model_download_planes(dms_model* m, unsigned n, std::vector<float4>& pos, std::vector<float4>& col, std::vector<float4>& nrm,
                                 std::vector<float>& times, hipStream_t s) {
  const SurfelPlanes& b = m->buf[m->cur];
  pos.resize(n);
  col.resize(n);
  nrm.resize(n);
  times.resize((size_t)n * DMS_MAX_SENSORS);
  if (n == 0) return DMS_OK;
  DMS_HIP(hipMemcpyAsync(pos.data(), b.pos, n * sizeof(float4), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipMemcpyAsync(col.data(), b.col, n * sizeof(float4), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipMemcpyAsync(nrm.data(), b.nrm, n * sizeof(float4), hipMemcpyDeviceToHost, s));
  for (int k = 0; k < DMS_MAX_SENSORS; ++k)
    DMS_HIP(hipMemcpyAsync(times.data() + (size_t)k * n, b.times + (size_t)k * m->cap, n * sizeof(float), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  return DMS_OK;
}


static int model_download_any(dms_model* m, float* host, unsigned max_count, unsigned* count, int nsens, bool ref_layout,
                              hipStream_t s) {
  DMS_REQUIRE(m && host && count, "null argument");
  unsigned n = 0;
  int rc = model_flush_pending(m, s);
  if (!rc) rc = dms_model_count(m, &n, s);
  if (rc) return rc;
  if (n > max_count) n = max_count;
  std::vector<float4> pos, col, nrm;
  std::vector<float> times;
  if ((rc = model_download_planes(m, n, pos, col, nrm, times, s))) return rc;
  const int stride = 12 + nsens;
  for (unsigned i = 0; i < n; ++i) {
    float* r = host + (size_t)i * stride;
    r[0] = pos[i].x; r[1] = pos[i].y; r[2] = pos[i].z; r[3] = pos[i].w;
    r[4] = col[i].x; r[5] = col[i].y; r[6] = col[i].z; r[7] = col[i].w;
    if (ref_layout) {  // Shaders/Vertex.cpp:21-50: times sit between colour and normal
      for (int k = 0; k < nsens; ++k) r[8 + k] = times[(size_t)k * n + i];
      r[8 + nsens] = nrm[i].x; r[9 + nsens] = nrm[i].y; r[10 + nsens] = nrm[i].z; r[11 + nsens] = nrm[i].w;
    } else {
      r[8] = nrm[i].x; r[9] = nrm[i].y; r[10] = nrm[i].z; r[11] = nrm[i].w;
      for (int k = 0; k < nsens; ++k) r[12 + k] = times[(size_t)k * n + i];
    }
  }
  *count = n;
  return DMS_OK;
}

static int model_upload_any(dms_model* m, const float* host, unsigned n, int nsens, bool ref_layout, hipStream_t s) {
  DMS_REQUIRE(m && (host || n == 0), "null argument");
  m->pending_update = false;  // (the map is replaced: a deferred update of the old one has nothing left to update)
  if (n > m->cap) {
    set_error("dms_model_upload: %u surfels exceed capacity %zu", n, m->cap);
    return DMS_ERR_CAPACITY;
  }
  std::vector<float4> pos(n), col(n), nrm(n);
  std::vector<float> times((size_t)n * DMS_MAX_SENSORS, -3.0f);
  const int stride = 12 + nsens;
  for (unsigned i = 0; i < n; ++i) {
    const float* r = host + (size_t)i * stride;
    pos[i] = make_float4(r[0], r[1], r[2], r[3]);
    col[i] = make_float4(r[4], r[5], r[6], r[7]);
    if (ref_layout) {
      for (int k = 0; k < nsens; ++k) times[(size_t)k * n + i] = r[8 + k];
      nrm[i] = make_float4(r[8 + nsens], r[9 + nsens], r[10 + nsens], r[11 + nsens]);
    } else {
      nrm[i] = make_float4(r[8], r[9], r[10], r[11]);
      for (int k = 0; k < nsens; ++k) times[(size_t)k * n + i] = r[12 + k];
    }
  }
  for (int k = DMS_MAX_SENSORS - 1; k >= m->live_planes; --k) {  // the planes the caller's records carry times in (surfel.hpp, live_planes)
    bool any = false;
    for (unsigned i = 0; i < n && !any; ++i) any = !(times[(size_t)k * n + i] == -3.0f);
    if (any) {
      m->live_planes = k + 1;
      break;
    }
  }
  const SurfelPlanes& b = m->buf[m->cur];
  if (n) {
    DMS_HIP(hipMemcpyAsync(b.pos, pos.data(), n * sizeof(float4), hipMemcpyHostToDevice, s));
    DMS_HIP(hipMemcpyAsync(b.col, col.data(), n * sizeof(float4), hipMemcpyHostToDevice, s));
    DMS_HIP(hipMemcpyAsync(b.nrm, nrm.data(), n * sizeof(float4), hipMemcpyHostToDevice, s));
    for (int k = 0; k < DMS_MAX_SENSORS; ++k)
      DMS_HIP(hipMemcpyAsync(b.times + (size_t)k * m->cap, times.data() + (size_t)k * n, n * sizeof(float), hipMemcpyHostToDevice, s));
  }
  unsigned cnt = n;
  DMS_HIP(hipMemcpyAsync(m->d_count, &cnt, sizeof(unsigned), hipMemcpyHostToDevice, s));
  DMS_HIP(hipStreamSynchronize(s));
  m->count_upper = n;
  m->count_hold = 3;  // the map was replaced from outside: older frame results no longer bound it
  m->version += 1;
  return DMS_OK;
}

int dms_model_download_ref(dms_model* m, float* host, unsigned int max_count, unsigned int* count, dms_stream s) {
  return model_download_any(m, host, max_count, count, DMS_REF_MAX_SENSORS, true, (hipStream_t)s);
}
int dms_model_upload_ref(dms_model* m, const float* host, unsigned int count, dms_stream s) {
  return model_upload_any(m, host, count, DMS_REF_MAX_SENSORS, true, (hipStream_t)s);
}
int dms_model_download(dms_model* m, float* host, unsigned int max_count, unsigned int* count, dms_stream s) {
  return model_download_any(m, host, max_count, count, DMS_MAX_SENSORS, false, (hipStream_t)s);
}
int dms_model_upload(dms_model* m, const float* host, unsigned int count, dms_stream s) {
  return model_upload_any(m, host, count, DMS_MAX_SENSORS, false, (hipStream_t)s);
}

int dms_pose_block_set(dms_pose_block* dev, const float* pose16, dms_stream s) {
  DMS_REQUIRE(dev && pose16, "null argument");
  dms_pose_block h;
  memcpy(h.pose, pose16, sizeof(h.pose));
  // Eigen Matrix4f::inverse() (IndexMap.cpp:166 etc.): general 4×4 inverse in float
  sm::inv4t<float>(h.pose, h.t_inv);
  DMS_HIP(hipMemcpyAsync(dev, &h, sizeof(h), hipMemcpyHostToDevice, (hipStream_t)s));
  DMS_HIP(hipStreamSynchronize((hipStream_t)s));
  return DMS_OK;
}

}  // extern "C"

namespace dms {

// ---------------------------------------------------------------------------------------
// G3 + G4: first-frame surfels (vertex_feedback.{vert,geom}, init_unstable.vert)
// ---------------------------------------------------------------------------------------
struct BootArgs {
  const uchar4* rgba;
  const float* depth_raw;       // metric
  const float* depth_filtered;  // metric, filtered
  int cols, rows;
  float cx, cy, ifx, ify;  // cam = (cx, cy, 1/fx, 1/fy), float reciprocals (FeedbackBuffer.cpp:93-96)
  int time, timeIdx;
  float maxDepth;
};

// geometry.glsl:19-39: vertex / central-difference normal on a float depth map.
// texcoords are the uv-buffer values; x = tx * cols, y = ty * rows.
__device__ __forceinline__ f3 fb_vertex(const float* depth, int cols, int sx, int sy, float x, float y, float cx, float cy, float ifx,
                                        float ify) {
  const float z = depth[(size_t)sy * cols + sx];
  return mk3(((x - cx) * z) * ifx, ((y - cy) * z) * ify, z);
}

__device__ __forceinline__ f3 fb_normal(const float* depth, int cols, int rows, const f3& vPosition, float tx, float ty, float x, float y,
                                        float cx, float cy, float ifx, float ify) {
  const float colsf = (float)cols, rowsf = (float)rows;
  const int sx = texel(tx, colsf, cols), sy = texel(ty, rowsf, rows);
  const int sxf = texel(tx + (1.0f / colsf), colsf, cols), sxb = texel(tx - (1.0f / colsf), colsf, cols);
  const int syf = texel(ty + (1.0f / rowsf), rowsf, rows), syb = texel(ty - (1.0f / rowsf), rowsf, rows);
  const f3 xf = fb_vertex(depth, cols, sxf, sy, x + 1.f, y, cx, cy, ifx, ify);
  const f3 xb = fb_vertex(depth, cols, sxb, sy, x - 1.f, y, cx, cy, ifx, ify);
  const f3 yf = fb_vertex(depth, cols, sx, syf, x, y + 1.f, cx, cy, ifx, ify);
  const f3 yb = fb_vertex(depth, cols, sx, syb, x, y - 1.f, cx, cy, ifx, ify);
  const f3 del_x = mk3(((xb.x + vPosition.x) / 2.f) - ((xf.x + vPosition.x) / 2.f), ((xb.y + vPosition.y) / 2.f) - ((xf.y + vPosition.y) / 2.f),
                       ((xb.z + vPosition.z) / 2.f) - ((xf.z + vPosition.z) / 2.f));
  const f3 del_y = mk3(((yb.x + vPosition.x) / 2.f) - ((yf.x + vPosition.x) / 2.f), ((yb.y + vPosition.y) / 2.f) - ((yf.y + vPosition.y) / 2.f),
                       ((yb.z + vPosition.z) / 2.f) - ((yf.z + vPosition.z) / 2.f));
  return normalized3(cross3(del_x, del_y));
}

// element e = column-major pixel (x = e / rows, y = e % rows): GlobalModel.cpp:100-108 order
__global__ __launch_bounds__(256) void k_boot_flags(BootArgs a, unsigned char* __restrict__ keep, unsigned* __restrict__ block_count) {
  const int n = a.cols * a.rows;
  const int base = blockIdx.x * kScanChunk;
  int cnt = 0;
  for (int k = 0; k < kScanChunk / 256; ++k) {
    const int e = base + k * 256 + threadIdx.x;
    unsigned char f = 0;
    if (e < n) {
      const int px = e / a.rows, py = e - px * a.rows;
      const float z = a.depth_raw[(size_t)py * a.cols + px];
      // vertex_feedback.vert:55-62 + .geom:38: emitted iff 0 < z <= maxDepth
      f = (z > 0.f && !(z > a.maxDepth)) ? 1 : 0;
      keep[e] = f;
    }
    cnt += f;
  }
  const unsigned tot = block_sum_u32(cnt);
  if (threadIdx.x == 0) block_count[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_boot_scatter(BootArgs a, const unsigned char* __restrict__ keep,
                                                      const unsigned* __restrict__ block_offset, SurfelPlanes out, size_t cap) {
  const int n = a.cols * a.rows;
  const int base = blockIdx.x * kScanChunk;
  unsigned running = block_offset[blockIdx.x];
  for (int k = 0; k < kScanChunk / 256; ++k) {
    const int e = base + k * 256 + threadIdx.x;
    const bool f = (e < n) && keep[e];
    unsigned total;
    const unsigned rank = block_exclusive_rank(f, total);
    if (f) {
      const size_t dst = running + rank;
      if (dst < cap) {
        const int px = e / a.rows, py = e - px * a.rows;
        const float tx = uv_coord(px, a.cols), ty = uv_coord(py, a.rows);
        const float x = tx * (float)a.cols, y = ty * (float)a.rows;
        // RAW feedback: position, colour, confidence (GlobalModel.cpp:355-370)
        const f3 vr = fb_vertex(a.depth_raw, a.cols, px, py, x, y, a.cx, a.cy, a.ifx, a.ify);
        // FILTERED feedback: normal + radius (GlobalModel.cpp:372-378)
        const f3 vf = fb_vertex(a.depth_filtered, a.cols, px, py, x, y, a.cx, a.cy, a.ifx, a.ify);
        const f3 nf = fb_normal(a.depth_filtered, a.cols, a.rows, vf, tx, ty, x, y, a.cx, a.cy, a.ifx, a.ify);
        const float rad = surfel_radius(vf.z, nf.z, a.ifx, a.ify);
        const float conf = surfel_confidence(x, y, a.cx, a.cy, 1.0f);
        const uchar4 c = a.rgba[(size_t)py * a.cols + px];
        out.pos[dst] = make_float4(vr.x, vr.y, vr.z, conf);
        // init_unstable.vert:36-39: colour kept, y = 0, z (init time) = 1, w = time stamp
        out.col[dst] = make_float4(encode_color_bytes(c.x, c.y, c.z), 0.f, 1.f, (float)a.time);
        out.nrm[dst] = make_float4(nf.x, nf.y, nf.z, rad);
        for (int s = 0; s < DMS_MAX_SENSORS; ++s) out.times[(size_t)s * cap + dst] = (s == a.timeIdx) ? (float)a.time : -3.f;
      }
    }
    running += total;
  }
}

int model_initialise(dms_model* m, const dms_image2d* rgba, const dms_image2d* dm, const dms_image2d* dmf, const dms_camera* cam, int time,
                     int timeIdx, float maxDepth, hipStream_t s) {
  DMS_REQUIRE(m && rgba && dm && dmf && cam, "null argument");
  DMS_REQUIRE(rgba->cols == m->width && rgba->rows == m->height && dm->cols == m->width && dmf->cols == m->width, "shape mismatch");
  DMS_REQUIRE(timeIdx >= 0 && timeIdx < DMS_MAX_SENSORS, "timeIdx out of range");
  BootArgs a;
  a.rgba = (const uchar4*)rgba->data;
  a.depth_raw = (const float*)dm->data;
  a.depth_filtered = (const float*)dmf->data;
  a.cols = m->width;
  a.rows = m->height;
  a.cx = cam->cx;
  a.cy = cam->cy;
  a.ifx = 1.0f / cam->fx;
  a.ify = 1.0f / cam->fy;
  a.time = time;
  a.timeIdx = timeIdx;
  a.maxDepth = maxDepth;
  const int n = a.cols * a.rows;
  const int nb = (n + kScanChunk - 1) / kScanChunk;
  if (m->live_planes < timeIdx + 1) m->live_planes = timeIdx + 1;
  hipLaunchKernelGGL(k_boot_flags, dim3(nb), dim3(256), 0, s, a, m->keep, m->block_count);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, m->block_count, m->block_offset, nb, m->d_count, (unsigned)m->cap, (unsigned*)nullptr);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_boot_scatter, dim3(nb), dim3(256), 0, s, a, m->keep, m->block_offset, m->buf[m->cur], m->cap);
  DMS_CHECK_LAUNCH();
  m->count_upper = (size_t)n < m->cap ? (size_t)n : m->cap;
  m->count_hold = 3;
  m->version += 1;
  return DMS_OK;
}

// ---------------------------------------------------------------------------------------
// G5: index map (index_map.vert:41-67, index_map.frag:31-37)
// ---------------------------------------------------------------------------------------
struct ProjArgs {
  const dms_pose_block* pose;
  float cx, cy, fx, fy;
  float colsf, rowsf;
  int cols, rows;
  float maxDepth;
  int time, timeIdx, timeDelta;
  // splat only
  float confThreshold;
  int maxTime;
  int actv;
  // index map only: 1 = images stored column-major (pixel (x, y) at x * rows + y).  The fusion
  // consumers walk the image by columns (the reference's draw order, GlobalModel.cpp:100-108),
  // so this is the layout that makes their 16-byte gathers coalesce.
  int transposed;
  int xcd;  // XCD-aware block order of the per-pixel passes (common.hpp xcd_block)
};

__global__ void k_clear_zbuf(unsigned long long* z, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) z[i] = kZClear;
}

int clear_zbuf(unsigned long long* zbuf, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_clear_zbuf, dim3(min((n + 255) / 256, 2048)), dim3(256), 0, s, zbuf, n);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

// window position of a camera-frame point through the shader's NDC arithmetic
// (index_map.vert:56-57 / splat.vert:37-42) and the viewport transform xw = (x_ndc + 1) * (W/2).
__device__ __forceinline__ bool project_window(const ProjArgs& a, const f3& p, float& xw, float& yw, float& zw) {
  const float xn = ((((a.fx * p.x) / p.z) + a.cx) - (a.colsf * 0.5f)) / (a.colsf * 0.5f);
  const float yn = ((((a.fy * p.y) / p.z) + a.cy) - (a.rowsf * 0.5f)) / (a.rowsf * 0.5f);
  const float zn = p.z / a.maxDepth;
  // GL clips a point by its centre against -w <= x,y,z <= w (w = 1)
  if (!(xn >= -1.f && xn <= 1.f && yn >= -1.f && yn <= 1.f && zn >= -1.f && zn <= 1.f)) return false;
  xw = (xn + 1.f) * (a.colsf * 0.5f);
  yw = (yn + 1.f) * (a.rowsf * 0.5f);
  zw = zn * 0.5f + 0.5f;
  return true;
}

// APPLY: the update pass of the preceding fuse (model_fuse with defer_update) has not run: a surfel that won a measurement
// (winner[i] = its slot) is updated here, by the same function, before it is projected — the index map after a fuse reads
// every surfel anyway, so the separate pass over the slots and its launch are saved.
struct PendingUpdate {
  const float4 *slot_pos, *slot_col, *slot_nrm;
  unsigned* winner;
  int time, timeIdx;
};
template <bool APPLY>
__global__ __launch_bounds__(256) void k_index_project(ProjArgs a, SurfelPlanes sp, size_t cap, const unsigned* __restrict__ d_count,
                                                       unsigned long long* __restrict__ zbuf, PendingUpdate u) {
  const unsigned M = d_count[0];
  const float* Tinv = a.pose->t_inv;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += blockDim.x * gridDim.x) {
    if (APPLY) {
      const unsigned w = u.winner[i];
      if (w != kEmptyWinner) {
        u.winner[i] = kEmptyWinner;  // re-arm for the next frame
        fuse_update_apply(i, w, u.slot_pos, u.slot_col, u.slot_nrm, sp, cap, u.time, u.timeIdx);
      }
    }
    const float4 pc = sp.pos[i];
    const f3 ph = xform_point(Tinv, mk3(pc.x, pc.y, pc.z));
    const float vt = sp.times[(size_t)a.timeIdx * cap + i];
    if (ph.z > a.maxDepth || ph.z < 0.f || (vt != -3.f && (float)a.time - vt > (float)a.timeDelta)) continue;
    float xw, yw, zw;
    if (!project_window(a, ph, xw, yw, zw)) continue;
    const int px = (int)floorf(xw), py = (int)floorf(yw);
    if (px < 0 || py < 0 || px >= a.cols || py >= a.rows) continue;
    const unsigned d = depth24(zw);
    if (d >= 0xFFFFFFu) continue;  // GL_LESS against the cleared depth 1.0
    const unsigned long long key = ((unsigned long long)d << 32) | (unsigned long long)i;
    unsigned long long* cell = zbuf + (a.transposed ? (size_t)px * a.rows + py : (size_t)py * a.cols + px);
    if (key < *cell) atomicMin(cell, key);
  }
}

__global__ __launch_bounds__(256) void k_index_resolve(ProjArgs a, SurfelPlanes sp, size_t cap, unsigned long long* __restrict__ zbuf,
                                                       unsigned* __restrict__ index, float4* __restrict__ vertConf,
                                                       float4* __restrict__ colorTime, float4* __restrict__ normRad, int clear_after) {
  // p runs over the storage order of the images (row-major, or column-major when a.transposed):
  // the z-buffer uses the same order, so reads and writes are coalesced either way
  const int n = a.cols * a.rows;
  const float* Tinv = a.pose->t_inv;
  for (int p = xcd_block(blockIdx.x, gridDim.x, a.xcd) * blockDim.x + threadIdx.x; p < n; p += blockDim.x * gridDim.x) {
    const unsigned long long key = zbuf[p];
    if (clear_after) zbuf[p] = kZClear;  // hand the z-buffer back empty: the next draw needs no clear launch
    if ((unsigned)(key >> 32) >= 0xFFFFFFu) {  // cleared colour (glClearColor 0)
      index[p] = 0;
      vertConf[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      colorTime[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      normRad[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    const unsigned i = (unsigned)(key & 0xFFFFFFFFull);
    const float4 pc = sp.pos[i], cc = sp.col[i], nr = sp.nrm[i];
    const f3 ph = xform_point(Tinv, mk3(pc.x, pc.y, pc.z));
    const f3 nh = normalized3(xform_dir(Tinv, mk3(nr.x, nr.y, nr.z)));
    index[p] = i;
    vertConf[p] = make_float4(ph.x, ph.y, ph.z, pc.w);
    colorTime[p] = make_float4(cc.x, cc.y, cc.z, sp.times[(size_t)a.timeIdx * cap + i]);
    normRad[p] = make_float4(nh.x, nh.y, nh.z, nr.w);
  }
}

// column-major -> row-major copy of an image with `elem`-byte pixels (inspection path only)
template <typename T>
__global__ void k_untranspose(const T* __restrict__ src, T* __restrict__ dst, int cols, int rows) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  dst[(size_t)y * cols + x] = src[(size_t)x * rows + y];
}

int untranspose(const void* src, void* dst, int cols, int rows, int elem, hipStream_t s) {
  dim3 b(32, 8), g((cols + 31) / 32, (rows + 7) / 8);
  if (elem == 4)
    hipLaunchKernelGGL(k_untranspose<unsigned>, g, b, 0, s, (const unsigned*)src, (unsigned*)dst, cols, rows);
  else if (elem == 16)
    hipLaunchKernelGGL(k_untranspose<float4>, g, b, 0, s, (const float4*)src, (float4*)dst, cols, rows);
  else
    DMS_REQUIRE(false, "elem must be 4 or 16");
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

static int surfel_grid(size_t upper) {
  size_t b = (upper + 255) / 256;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

static bool dense_img(const dms_image2d& im, size_t elem, int w, int h) {
  return im.data && im.cols == w && im.rows == h && im.pitch == (size_t)w * elem;
}

static void fill_proj(ProjArgs& a, const dms_model* m, const dms_pose_block* pose, const dms_camera* cam, float maxDepth, int time,
                      int timeIdx, int timeDelta) {
  a.pose = pose;
  a.cx = cam->cx;
  a.cy = cam->cy;
  a.fx = cam->fx;
  a.fy = cam->fy;
  a.cols = m->width;
  a.rows = m->height;
  a.colsf = (float)m->width;
  a.rowsf = (float)m->height;
  a.maxDepth = maxDepth;
  a.time = time;
  a.timeIdx = timeIdx;
  a.timeDelta = timeDelta;
  a.confThreshold = 0.f;
  a.maxTime = 0;
  a.actv = 0;
  a.transposed = 0;
  a.xcd = xcd_remap_enabled();
}

int index_map(dms_model* m, const dms_pose_block* pose, const dms_camera* cam, int time, int timeIdx, float maxDepth, int timeDelta,
              unsigned long long* zbuf, dms_indexmap_out* out, int transposed, int zclean, hipStream_t s) {
  // zclean: the caller's z-buffer is empty on entry and must be handed back empty (the resolve pass
  // clears what it reads), so a chain of draws needs no clear launches
  DMS_REQUIRE(m && pose && cam && zbuf && out, "null argument");
  DMS_REQUIRE(timeIdx >= 0 && timeIdx < DMS_MAX_SENSORS, "timeIdx out of range");
  const int W = m->width, H = m->height;
  DMS_REQUIRE(dense_img(out->index, 4, W, H) && dense_img(out->vertConf, 16, W, H) && dense_img(out->colorTime, 16, W, H) &&
                  dense_img(out->normRad, 16, W, H),
              "index-map targets must be dense W×H");
  ProjArgs a;
  fill_proj(a, m, pose, cam, maxDepth, time, timeIdx, timeDelta);
  a.transposed = transposed ? 1 : 0;
  const int n = W * H;
  if (!zclean) {
    hipLaunchKernelGGL(k_clear_zbuf, dim3(min((n + 255) / 256, 2048)), dim3(256), 0, s, zbuf, n);
    DMS_CHECK_LAUNCH();
  }
  PendingUpdate u = {m->slot_pos, m->slot_col, m->slot_nrm, m->winner, m->pending_time, m->pending_timeIdx};
  if (m->pending_update) {
    m->pending_update = false;
    hipLaunchKernelGGL(k_index_project<true>, dim3(surfel_grid(m->count_upper)), dim3(256), 0, s, a, m->buf[m->cur], m->cap, m->d_count, zbuf, u);
  } else {
    hipLaunchKernelGGL(k_index_project<false>, dim3(surfel_grid(m->count_upper)), dim3(256), 0, s, a, m->buf[m->cur], m->cap, m->d_count, zbuf, u);
  }
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_index_resolve, dim3(min((n + 255) / 256, 2048)), dim3(256), 0, s, a, m->buf[m->cur], m->cap, zbuf,
                     (unsigned*)out->index.data, (float4*)out->vertConf.data, (float4*)out->colorTime.data, (float4*)out->normRad.data,
                     zclean);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

// ---------------------------------------------------------------------------------------
// Map merge: GlobalModel::consume (GlobalModel.cpp:898-993) + consume.vert — the
// consuming map keeps its surfels (identity transform) and appends the other map's surfels moved by
// the relative transform: pos = T * (x, y, z, 1) with the confidence kept, normal = mat3(T) * n with
// the radius kept, colour and all per-sensor times unchanged.  The reference re-streams both maps
// through transform feedback into the other buffer; appending in place moves only the consumed
// map.  Sources: another model on this device, or a packed device buffer of 20-float records
// (pos4 col4 nrm4 times8, the dms_model_download layout) as received from another rank.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void consume_store(const SurfelPlanes& dst, size_t cap, size_t at, const float* T, const float4 p, const float4 c,
                                              const float4 n, const float (&tm)[DMS_MAX_SENSORS]) {
  // GLSL `transform * vec4(v, 1)`: row i = ((T_i0 x + T_i1 y) + T_i2 z) + T_i3 (row-major T here)
  const f3 q = xform_point(T, mk3(p.x, p.y, p.z));
  const f3 m = xform_dir(T, mk3(n.x, n.y, n.z));
  dst.pos[at] = make_float4(q.x, q.y, q.z, p.w);
  dst.col[at] = c;
  dst.nrm[at] = make_float4(m.x, m.y, m.z, n.w);
#pragma unroll
  for (int k = 0; k < DMS_MAX_SENSORS; ++k) dst.times[(size_t)k * cap + at] = tm[k];
}

struct Pose16v {
  float v[16];
};

// The consuming map's own records go through consume.vert too, with the identity (GlobalModel.cpp:903-949): every finite value
// comes out as it went in, but a component that is -0 can come out +0 (-0 + 0 * y = +0 for y > 0) — found by running the reference's
// program (tests/test_ref_gl_pin_*.py, stage `consumed`).  One pass over position and normal, in place; the rest is copied as is.
__global__ __launch_bounds__(256) void k_consume_identity(SurfelPlanes dst, const unsigned* __restrict__ dcount) {
  const unsigned n = dcount[0];
  const float I[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    const float4 p = dst.pos[i], nr = dst.nrm[i];
    const f3 q = xform_point(I, mk3(p.x, p.y, p.z));
    const f3 m = xform_dir(I, mk3(nr.x, nr.y, nr.z));
    dst.pos[i] = make_float4(q.x, q.y, q.z, p.w);
    dst.nrm[i] = make_float4(m.x, m.y, m.z, nr.w);
  }
}

__global__ __launch_bounds__(256) void k_consume_model(SurfelPlanes dst, size_t dcap, const unsigned* __restrict__ dcount, unsigned* __restrict__ dcount_new,
                                                       SurfelPlanes src, size_t scap, const unsigned* __restrict__ scount, Pose16v T) {
  const unsigned base = dcount[0], ns = scount[0];
  const unsigned room = base < dcap ? (unsigned)(dcap - base) : 0u;
  const unsigned n = ns < room ? ns : room;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    float tm[DMS_MAX_SENSORS];
#pragma unroll
    for (int k = 0; k < DMS_MAX_SENSORS; ++k) tm[k] = src.times[(size_t)k * scap + i];
    consume_store(dst, dcap, (size_t)base + i, T.v, src.pos[i], src.col[i], src.nrm[i], tm);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) dcount_new[0] = base + n;
}

__global__ __launch_bounds__(256) void k_consume_records(SurfelPlanes dst, size_t dcap, const unsigned* __restrict__ dcount, unsigned* __restrict__ dcount_new,
                                                         const float* __restrict__ rec, unsigned ns, Pose16v T) {
  const unsigned base = dcount[0];
  const unsigned room = base < dcap ? (unsigned)(dcap - base) : 0u;
  const unsigned n = ns < room ? ns : room;
  constexpr int stride = 12 + DMS_MAX_SENSORS;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    const float4* r = reinterpret_cast<const float4*>(rec + (size_t)i * stride);  // 80-byte records: 16-byte aligned
    float tm[DMS_MAX_SENSORS];
#pragma unroll
    for (int k = 0; k < DMS_MAX_SENSORS; k += 4) {
      const float4 t4 = r[3 + k / 4];
      tm[k] = t4.x;
      tm[k + 1] = t4.y;
      tm[k + 2] = t4.z;
      tm[k + 3] = t4.w;
    }
    consume_store(dst, dcap, (size_t)base + i, T.v, r[0], r[1], r[2], tm);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) dcount_new[0] = base + n;
}

// pack the map into 20-float records in a device buffer (the device-side form of dms_model_download)
__global__ __launch_bounds__(256) void k_export_records(SurfelPlanes src, size_t scap, const unsigned* __restrict__ scount, unsigned max_count,
                                                        float* __restrict__ rec, unsigned* __restrict__ written) {
  const unsigned ns = scount[0];
  const unsigned n = ns < max_count ? ns : max_count;
  constexpr int stride = 12 + DMS_MAX_SENSORS;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    float4* r = reinterpret_cast<float4*>(rec + (size_t)i * stride);
    r[0] = src.pos[i];
    r[1] = src.col[i];
    r[2] = src.nrm[i];
#pragma unroll
    for (int k = 0; k < DMS_MAX_SENSORS; k += 4)
      r[3 + k / 4] = make_float4(src.times[(size_t)k * scap + i], src.times[(size_t)(k + 1) * scap + i], src.times[(size_t)(k + 2) * scap + i],
                                 src.times[(size_t)(k + 3) * scap + i]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) written[0] = n;
}

// exact device count -> host bound (synchronises `s`)
static int refresh_count(dms_model* m, hipStream_t s) {
  DMS_HIP(hipMemcpyAsync(m->h_count, m->d_count, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  m->count_upper = m->h_count[0];
  return DMS_OK;
}

static int consume_finish(dms_model* m, size_t added_upper) {
  std::swap(m->d_count, m->d_count_alt);  // the kernel wrote the new count into the other cell
  const size_t upper = m->count_upper + added_upper;
  m->count_upper = upper < m->cap ? upper : m->cap;
  m->count_hold = 3;
  m->version += 1;
  return DMS_OK;
}

// highest time plane (+ 1) in which any record carries something other than the "never seen" marker
__global__ __launch_bounds__(256) void k_records_live_planes(const float* __restrict__ rec, unsigned n, unsigned* __restrict__ out) {
  constexpr int stride = 12 + DMS_MAX_SENSORS;
  unsigned live = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)blockDim.x * gridDim.x)
    for (int k = DMS_MAX_SENSORS - 1; k >= 0; --k)
      if (!(rec[i * stride + 12 + k] == -3.0f)) {
        live = live > (unsigned)(k + 1) ? live : (unsigned)(k + 1);
        break;
      }
  if (live) atomicMax(out, live);
}

int model_consume(dms_model* dst, const dms_model* src, const float* T16, hipStream_t s) {
  DMS_REQUIRE(dst && src && T16 && dst != src, "bad argument");
  {
    int rc = model_flush_pending(dst, s);
    if (!rc) rc = model_flush_pending(const_cast<dms_model*>(src), s);
    if (rc) return rc;
  }
  if (dst->count_upper + src->count_upper > dst->cap) {
    // the host-side bounds are loose (they grow by a frame's worth of slots per clean): read the exact counts before refusing
    int rc = refresh_count(dst, s);
    if (!rc) rc = refresh_count(const_cast<dms_model*>(src), s);
    if (rc) return rc;
  }
  if (dst->count_upper + src->count_upper > dst->cap) {
    set_error("dms_model_consume: %zu + %zu surfels exceed the capacity %zu", dst->count_upper, src->count_upper, dst->cap);
    return DMS_ERR_CAPACITY;
  }
  Pose16v T;
  memcpy(T.v, T16, sizeof(T.v));
  hipLaunchKernelGGL(k_consume_identity, dim3(surfel_grid(dst->count_upper)), dim3(256), 0, s, dst->buf[dst->cur], dst->d_count);
  hipLaunchKernelGGL(k_consume_model, dim3(surfel_grid(src->count_upper)), dim3(256), 0, s, dst->buf[dst->cur], dst->cap, dst->d_count,
                     dst->d_count_alt, src->buf[src->cur], src->cap, src->d_count, T);
  DMS_CHECK_LAUNCH();
  if (dst->live_planes < src->live_planes) dst->live_planes = src->live_planes;
  return consume_finish(dst, src->count_upper);
}

int model_consume_records(dms_model* dst, const float* rec_dev, unsigned n, const float* T16, hipStream_t s) {
  DMS_REQUIRE(dst && T16 && (rec_dev || n == 0), "bad argument");
  if (int rc0 = model_flush_pending(dst, s)) return rc0;
  DMS_REQUIRE(((uintptr_t)rec_dev & 15) == 0, "record buffer must be 16-byte aligned");
  if (dst->count_upper + n > dst->cap) {
    const int rc = refresh_count(dst, s);  // loose bound: read the exact count before refusing
    if (rc) return rc;
  }
  if (dst->count_upper + n > dst->cap) {
    set_error("dms_model_consume_records: %zu + %u surfels exceed the capacity %zu", dst->count_upper, n, dst->cap);
    return DMS_ERR_CAPACITY;
  }
  Pose16v T;
  memcpy(T.v, T16, sizeof(T.v));
  hipLaunchKernelGGL(k_consume_identity, dim3(surfel_grid(dst->count_upper)), dim3(256), 0, s, dst->buf[dst->cur], dst->d_count);
  hipLaunchKernelGGL(k_consume_records, dim3(surfel_grid(n)), dim3(256), 0, s, dst->buf[dst->cur], dst->cap, dst->d_count, dst->d_count_alt,
                     rec_dev, n, T);
  DMS_CHECK_LAUNCH();
  if (n && dst->live_planes < DMS_MAX_SENSORS) {  // which time planes the records carry anything in (surfel.hpp, live_planes)
    unsigned* cell = dst->clean_first + 2;
    DMS_HIP(hipMemsetAsync(cell, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_records_live_planes, dim3(surfel_grid(n)), dim3(256), 0, s, rec_dev, n, cell);
    DMS_CHECK_LAUNCH();
    unsigned live = 0;
    DMS_HIP(hipMemcpyAsync(&live, cell, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    DMS_HIP(hipStreamSynchronize(s));
    if (dst->live_planes < (int)live) dst->live_planes = (int)live;
  }
  return consume_finish(dst, n);
}

int model_export_records(dms_model* m, float* rec_dev, unsigned max_count, unsigned* count_host, hipStream_t s) {
  DMS_REQUIRE(m && rec_dev && count_host, "null argument");
  if (int rc0 = model_flush_pending(m, s)) return rc0;
  DMS_REQUIRE(((uintptr_t)rec_dev & 15) == 0, "record buffer must be 16-byte aligned");
  hipLaunchKernelGGL(k_export_records, dim3(surfel_grid(m->count_upper)), dim3(256), 0, s, m->buf[m->cur], m->cap, m->d_count, max_count,
                     rec_dev, m->d_count_alt + 1);
  DMS_CHECK_LAUNCH();
  DMS_HIP(hipMemcpyAsync(m->h_count, m->d_count_alt + 1, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  *count_host = m->h_count[0];
  return DMS_OK;
}

// ---------------------------------------------------------------------------------------
// Deformation::sampleGraphModel, device half (Deformation.cpp:250-348; sample.vert, sample.geom):
// every sampleRate-th surfel as {pos.xyz, init time}, then ordered by init time.  The reference
// downloads the samples and std::sorts them on the host (order of equal times unspecified); here
// the ordering is a stable rank sort on device (ties keep surfel order) and only the sorted rows
// travel.  Scratch: the inactive half of the double-buffered map (free between cleans).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_graph_sample(SurfelPlanes src, const unsigned* __restrict__ d_count, int rate, float4* __restrict__ out,
                                                      unsigned* __restrict__ n_out) {
  const unsigned M = d_count[0];
  const unsigned n = (M + (unsigned)rate - 1u) / (unsigned)rate;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *n_out = n;
  if (i >= n) return;
  const size_t id = (size_t)i * (size_t)rate;  // gl_VertexID % sampleRate == 0
  const float4 p = src.pos[id];
  out[i] = make_float4(p.x, p.y, p.z, src.col[id].z);
}

__global__ __launch_bounds__(256) void k_graph_rank(const float4* __restrict__ in, const unsigned* __restrict__ n_in, float4* __restrict__ out) {
  const unsigned n = *n_in;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ float s_t[256];
  const float ti = i < n ? in[i].w : 0.f;
  unsigned rank = 0;
  for (unsigned base = 0; base < n; base += 256) {
    const unsigned j = base + threadIdx.x;
    s_t[threadIdx.x] = j < n ? in[j].w : 0.f;
    __syncthreads();
    const unsigned lim = min(256u, n - base);
    for (unsigned q = 0; q < lim; ++q) {
      const float tj = s_t[q];
      rank += (tj < ti || (tj == ti && base + q < i)) ? 1u : 0u;
    }
    __syncthreads();
  }
  if (i < n) out[rank] = in[i];
}

int model_sample_graph(dms_model* m, int sampleRate, float* rows4_host, int max_rows, int* n_host, hipStream_t s) {
  DMS_REQUIRE(m && n_host && sampleRate >= 2 && (rows4_host || max_rows == 0), "bad argument");
  if (int rc0 = model_flush_pending(m, s)) return rc0;
  const size_t n_upper = (m->count_upper + (size_t)sampleRate - 1) / (size_t)sampleRate;
  DMS_REQUIRE(2 * n_upper <= m->cap, "sample scratch");
  float4* raw = m->buf[1 - m->cur].pos;
  float4* sorted = raw + n_upper;
  const int nb = (int)((n_upper + 255) / 256);
  *n_host = 0;
  if (nb == 0) return DMS_OK;
  hipLaunchKernelGGL(k_graph_sample, dim3(nb), dim3(256), 0, s, m->buf[m->cur], m->d_count, sampleRate, raw, m->d_count_alt + 1);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_graph_rank, dim3(nb), dim3(256), 0, s, raw, m->d_count_alt + 1, sorted);
  DMS_CHECK_LAUNCH();
  DMS_HIP(hipMemcpyAsync(m->h_count, m->d_count_alt + 1, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  const unsigned n = m->h_count[0];
  *n_host = (int)n;
  const unsigned take = n < (unsigned)max_rows ? n : (unsigned)max_rows;
  if (take) DMS_HIP(hipMemcpy(rows4_host, sorted, (size_t)take * sizeof(float4), hipMemcpyDeviceToHost));
  return DMS_OK;
}

// ---------------------------------------------------------------------------------------
// G6 / G6': splat prediction (splat.vert:57-94, combo_splat.frag:35-60, depth_splat.frag:29-46)
// ---------------------------------------------------------------------------------------
struct SplatSurfel {
  f3 pos;       // camera frame
  f3 nrm;       // camera frame, normalised
  float rad, conf;
  float xw, yw;  // sprite centre in window coordinates
  float size;    // gl_PointSize
};

__device__ __forceinline__ f3 project_image(const ProjArgs& a, const f3& p) {  // splat.vert:44-49
  return mk3(((a.fx * p.x) / p.z) + a.cx, ((a.fy * p.y) / p.z) + a.cy, p.z);
}

// vertex stage: returns false when the surfel is culled / clipped
// (the normal / radius plane is read only for surfels that survive the cull and the clip: for a large map most do
// not, and the pass is bandwidth-bound there — 20 instead of 36 bytes per culled surfel)
// the cull of splat.vert for one set of (confidence threshold, time, newest time) — the part of the vertex stage that
// differs between two predictions of the same map from the same pose
__device__ __forceinline__ bool splat_culled(const ProjArgs& a, const f3& ph, float conf, float vt, float confThreshold, int time, int maxTime) {
  const bool actv = a.actv != 0;
  return !(!actv && vt == -3.f) && (ph.z > a.maxDepth || ph.z < 0.f || conf < confThreshold || (actv && vt == -3.f) ||
                                    (vt != -3.f && (float)time - vt > (float)a.timeDelta) || vt > (float)maxTime);
}

struct SplatSecond {  // second prediction sharing the project pass (same map, pose, depth range and mode)
  float confThreshold;
  int time, maxTime;
};

// `pass`: bit 0 = survives the cull of `a`, bit 1 = survives the cull of `b` (when given)
__device__ __forceinline__ bool splat_vertex(const ProjArgs& a, const float4& pc, const float4* __restrict__ nrp, float vt, SplatSurfel& o,
                                             const SplatSecond* b = nullptr, int* pass = nullptr) {
  const float* Tinv = a.pose->t_inv;
  const f3 ph = xform_point(Tinv, mk3(pc.x, pc.y, pc.z));
  int keep = splat_culled(a, ph, pc.w, vt, a.confThreshold, a.time, a.maxTime) ? 0 : 1;
  if (b) keep |= splat_culled(a, ph, pc.w, vt, b->confThreshold, b->time, b->maxTime) ? 0 : 2;
  if (pass) *pass = keep;
  if (!keep) return false;
  float zw;
  if (!project_window(a, ph, o.xw, o.yw, zw)) return false;
  const float4 nr = *nrp;
  o.pos = ph;
  o.conf = pc.w;
  o.rad = nr.w;
  o.nrm = normalized3(xform_dir(Tinv, mk3(nr.x, nr.y, nr.z)));
  const f3 x1n = normalized3(mk3(o.nrm.y - o.nrm.z, -o.nrm.x, o.nrm.x));
  const f3 x1 = mk3((x1n.x * o.rad) * 1.41421356f, (x1n.y * o.rad) * 1.41421356f, (x1n.z * o.rad) * 1.41421356f);
  const f3 y1 = cross3(o.nrm, x1);
  const f3 p1 = project_image(a, ph + x1), p2 = project_image(a, ph + y1), p3 = project_image(a, ph - y1), p4 = project_image(a, ph - x1);
  const float xmin = fminf(p1.x, fminf(p2.x, fminf(p3.x, p4.x))), xmax = fmaxf(p1.x, fmaxf(p2.x, fmaxf(p3.x, p4.x)));
  const float ymin = fminf(p1.y, fminf(p2.y, fminf(p3.y, p4.y))), ymax = fmaxf(p1.y, fmaxf(p2.y, fmaxf(p3.y, p4.y)));
  const float xDiff = fabsf(xmax - xmin), yDiff = fabsf(ymax - ymin);
  o.size = fmaxf(0.f, fmaxf(xDiff, yDiff));
  return true;
}

// fragment stage at pixel (px, py): ray/disc test.  Returns false on discard.
__device__ __forceinline__ bool splat_fragment(const ProjArgs& a, const SplatSurfel& s, int px, int py, f3& corrected, float& zw) {
  const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;  // gl_FragCoord
  const f3 l = normalized3(mk3((fcx - a.cx) / a.fx, (fcy - a.cy) / a.fy, 1.0f));
  const float k = dot3(s.pos, s.nrm) / dot3(l, s.nrm);
  corrected = mk3(k * l.x, k * l.y, k * l.z);
  const float sqrRad = s.rad * s.rad;  // pow(r, 2)
  const f3 diff = corrected - s.pos;
  if (dot3(diff, diff) > sqrRad) return false;
  zw = (corrected.z / (2.f * a.maxDepth)) + 0.5f;  // gl_FragDepth
  return true;
}

// The same fragment in fewer instructions for the project pass, whose sprite loop is bound by vector-instruction issue: four IEEE
// quotients, a square root and the 64-bit depth conversion were 70 of a fragment's 115 instructions.  exact_arith.hpp: the divisors
// fx, fy and 2 maxDepth are constants of the launch (reciprocal once, two correction steps per quotient); the ray's squared length is
// in [1, 16) for any pixel within 3.8 focal lengths of the principal point (square root without the small-operand scaling, reciprocal
// by one Newton step - both compared with the IEEE results over their whole domains on the device).  Same bits as splat_fragment:
// the accept / reject decision, the depth and `corrected` (a -0 quotient becomes +0 only where 0.5 is added next or the operand is
// an exact difference, which is never -0; NaN / infinite depths are rejected on both sides).
struct SplatRay {
  exact::Divisor fx, fy, z2;
};
__device__ __forceinline__ SplatRay splat_ray_consts(const ProjArgs& a) {
  return SplatRay{exact::divisor(a.fx), exact::divisor(a.fy), exact::divisor(2.f * a.maxDepth)};
}
__device__ __forceinline__ bool splat_fragment_lean(const SplatRay& rc, const ProjArgs& a, const SplatSurfel& s, int px, int py, f3& corrected, float& zw) {
  const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;  // gl_FragCoord
  const f3 v = mk3(exact::div(fcx - a.cx, rc.fx), exact::div(fcy - a.cy, rc.fy), 1.0f);
  const float vv = dot3(v, v);
  const float rn = vv < 16.f ? exact::rcp_1_4(exact::sqrt_normal(vv)) : 1.0f / sqrtf(vv);
  const f3 l = mk3(v.x * rn, v.y * rn, v.z * rn);
  const float k = dot3(s.pos, s.nrm) / dot3(l, s.nrm);
  corrected = mk3(k * l.x, k * l.y, k * l.z);
  const float sqrRad = s.rad * s.rad;
  const f3 diff = corrected - s.pos;
  if (dot3(diff, diff) > sqrRad) return false;
  zw = exact::div(corrected.z, rc.z2) + 0.5f;
  return true;
}

// sprite coverage: pixel centres inside [c - size/2, c + size/2); a size below 1 rasterises as 1
__device__ __forceinline__ void sprite_range(float c, float size, int n, int& lo, int& hi) {
  const float sz = fmaxf(size, 1.0f);
  const float a = c - sz * 0.5f, b = c + sz * 0.5f;
  lo = (int)ceilf(a - 0.5f);
  hi = (int)ceilf(b - 0.5f) - 1;
  if (lo < 0) lo = 0;
  if (hi > n - 1) hi = n - 1;
}

// Sprite sizes differ by up to 3x inside a wave (5x5 ... 9x9 pixels), so "one thread rasterises its
// surfel" leaves ~55 % of the lanes idle in a loop that is instruction bound.  Here a block runs the
// vertex stage for 256 surfels, parks their parameters in LDS, and then hands out sprite ROWS to
// threads (prefix sum of the row counts + binary search): a thread's trip count is one sprite width,
// and the rows of a large sprite are spread over many threads.  The winners are decided by 64-bit
// atomicMin, so the work order does not matter.
// DUAL: one pass over the map feeds two z-buffers — two predictions of the same map from the same pose that differ in
// the cull only (the frame's final prediction and the next frame's tracking prediction): the vertex stage and the sprite
// loop run once, a fragment competes in the z-buffer of every prediction its surfel survives the cull of.
template <bool DUAL>
__global__ __launch_bounds__(256) void k_splat_project(ProjArgs a, SurfelPlanes sp, size_t cap, const unsigned* __restrict__ d_count,
                                                       unsigned long long* __restrict__ zbuf, SplatSecond b2,
                                                       unsigned long long* __restrict__ zbuf2) {
  __shared__ float s_par[10][256];  // pos.xyz, nrm.xyz, rad, window x / y of the centre, squared reach (below)
  __shared__ int s_box[4][256];     // x0, width, y0, cull survivors (DUAL)
  __shared__ unsigned s_off[257];   // exclusive prefix of the row counts
  __shared__ unsigned s_w[4];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const unsigned M = d_count[0];
  const SplatRay ray = splat_ray_consts(a);
  // position and time of the next chunk are fetched while this one is scanned and rasterised: on a large map
  // most chunks are culled as a whole, and their cost would otherwise be one exposed memory round trip each
  float4 pc_next = make_float4(0.f, 0.f, 0.f, 0.f);
  float vt_next = 0.f;
  {
    const unsigned i0 = blockIdx.x * 256u + t;
    if (i0 < M) {
      pc_next = sp.pos[i0];
      vt_next = sp.times[(size_t)a.timeIdx * cap + i0];
    }
  }
  for (unsigned base = blockIdx.x * 256u; base < M; base += gridDim.x * 256u) {
    const unsigned i = base + t;
    const float4 pc = pc_next;
    const float vt = vt_next;
    {
      const unsigned in = i + gridDim.x * 256u;
      if (in < M) {
        pc_next = sp.pos[in];
        vt_next = sp.times[(size_t)a.timeIdx * cap + in];
      }
    }
    int rows_here = 0;
    if (i < M) {
      SplatSurfel s;
      int pass = 1;
      if (splat_vertex(a, pc, sp.nrm + i, vt, s, DUAL ? &b2 : nullptr, &pass) && s.size == s.size) {
        int x0, x1, y0, y1;
        sprite_range(s.xw, s.size, a.cols, x0, x1);
        sprite_range(s.yw, s.size, a.rows, y0, y1);
        // The sprite is the square around four points at sqrt(2) r along the tangent axes; the disc test passes on about
        // 40 % of its fragments.  A fragment can only pass if its ray comes within r of the centre P = Z (xc, yc, 1):
        // the distance of P from the ray along v = (x, y, 1) is Z |(xc, yc, 1) x v| / |v| >= Z rho / |v|, with
        // rho^2 = (x - xc)^2 + (y - yc)^2 — the fragment's offset from the centre in normalised image coordinates —
        // and |v|^2 <= G2 = 1 + (|xc| + h)^2 + (|yc| + h)^2 inside the sprite (half size h).  So rho > r sqrt(G2) / Z
        // rules a fragment out; with 2 % (and 0.02 pixel) added, far more than the rounding of the fragment test itself, the
        // rows and the columns of each row are cut to that circle before any fragment is evaluated.
        float reach = 3.0e18f;  // normalised units; no cut when the bound cannot be formed
        {
          const float ifx = 1.f / a.fx, ify = 1.f / a.fy;
          const float h = (0.5f * fmaxf(s.size, 1.f) + 1.f) * fmaxf(ifx, ify);
          const float xc = fabsf((s.xw - a.cx) * ifx) + h, yc = fabsf((s.yw - a.cy) * ify) + h;
          const float g2 = 1.f + xc * xc + yc * yc;
          const float rn = (s.rad * sqrtf(g2) / s.pos.z) * 1.02f + 0.02f * fmaxf(ifx, ify);
          if (s.pos.z > 0.f && rn == rn && rn < 3.0e18f) reach = rn;
          const float ry = fminf(reach * a.fy, 1.0e9f);
          const int c0 = (int)ceilf(fmaxf(s.yw - ry - 0.5f, -1.0e9f)), c1 = (int)floorf(fminf(s.yw + ry - 0.5f, 1.0e9f));  // |py + 0.5 - yw| <= ry
          y0 = max(y0, c0);
          y1 = min(y1, c1);
        }
        if (x1 >= x0 && y1 >= y0) {
          rows_here = y1 - y0 + 1;
          s_par[0][t] = s.pos.x;
          s_par[1][t] = s.pos.y;
          s_par[2][t] = s.pos.z;
          s_par[3][t] = s.nrm.x;
          s_par[4][t] = s.nrm.y;
          s_par[5][t] = s.nrm.z;
          s_par[6][t] = s.rad;
          s_par[7][t] = s.xw;
          s_par[8][t] = s.yw;
          s_par[9][t] = reach * reach;
          s_box[0][t] = x0;
          s_box[1][t] = x1 - x0 + 1;
          s_box[2][t] = y0;
          if (DUAL) s_box[3][t] = pass;
        }
      }
    }
    // block-wide exclusive scan of rows_here
    unsigned incl = (unsigned)rows_here;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    unsigned wbase = 0;
    for (int w = 0; w < wid; ++w) wbase += s_w[w];
    s_off[t] = wbase + incl - (unsigned)rows_here;
    if (t == 255) s_off[256] = wbase + incl;
    __syncthreads();
    const unsigned total = s_off[256];
    for (unsigned u = t; u < total; u += 256u) {
      // surfel j with s_off[j] <= u < s_off[j + 1]
      int lo = 0, hi = 256;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= u)
          lo = mid;
        else
          hi = mid;
      }
      const int j = lo;
      SplatSurfel s;
      s.pos = mk3(s_par[0][j], s_par[1][j], s_par[2][j]);
      s.nrm = mk3(s_par[3][j], s_par[4][j], s_par[5][j]);
      s.rad = s_par[6][j];
      const int x0 = s_box[0][j], w = s_box[1][j];
      const int py = s_box[2][j] + (int)(u - s_off[j]);
      const unsigned long long id = (unsigned long long)(base + (unsigned)j);
      const int pass = DUAL ? s_box[3][j] : 1;
      // columns of this row inside the surfel's reach (see the vertex stage)
      const float dyn = (((float)py + 0.5f) - s_par[8][j]) * (1.f / a.fy);
      const float left = s_par[9][j] - dyn * dyn;
      if (left < 0.f) continue;
      const float rx = fminf(sqrtf(left) * a.fx, 1.0e9f), xwc = s_par[7][j];
      const int xa = max(x0, (int)ceilf(fmaxf(xwc - rx - 0.5f, -1.0e9f)));  // |px + 0.5 - xw| <= rx
      const int xb = min(x0 + w - 1, (int)floorf(fminf(xwc + rx - 0.5f, 1.0e9f)));
      // The row in groups of four fragments: the z-buffer cells a group will compete for are fetched first (their addresses
      // depend on the pixel only), the ray / disc tests run while those loads are in flight, and a fragment goes to the
      // atomic only when it beats the value fetched (cells only ever decrease: a stale value can cost an atomic, never a win).
      const bool use1 = !DUAL || (pass & 1), use2 = DUAL && (pass & 2);
      for (int px0 = xa; px0 <= xb; px0 += 4) {
        unsigned long long z1[4], z2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int px = px0 + k;
          const size_t q = (size_t)px * a.rows + py;  // column-major z-buffer
          const bool in = px <= xb;
          z1[k] = (in && use1) ? zbuf[q] : 0ull;
          z2[k] = (in && use2) ? zbuf2[q] : 0ull;
        }
        // (the four ray / disc tests first, the atomics behind them: an atomic between two tests made the next test wait for it -
        // the counter that guards the prefetched cells also counts the atomics in flight)
        unsigned long long key[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int px = px0 + k;
          key[k] = ~0ull;
          f3 c;
          float zw;
          if (px <= xb && splat_fragment_lean(ray, a, s, px, py, c, zw)) {
            const unsigned d = depth24(zw);
            if (d < 0xFFFFFFu) key[k] = ((unsigned long long)d << 32) | id;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const size_t q = (size_t)(px0 + k) * a.rows + py;
          if (key[k] < z1[k]) atomicMin(zbuf + q, key[k]);
          if (DUAL && key[k] < z2[k]) atomicMin(zbuf2 + q, key[k]);
        }
      }
    }
    __syncthreads();  // the parameter tables are rewritten by the next chunk
  }
}

// FILL: the hole fill-in of the prediction (fill.hpp) for the pixel just resolved — the launch and the re-read of the
// three images that a separate fill-in pass costs are saved; the result-block mirror is done by block 0; for the
// denseEnough decision the threads that own a subsampled pixel count the non-black ones into 16 counters, which the
// consumer of the decision (the model pyramid kernel) sums.  (Taking the decision inside this launch — last block by
// ticket, the subsampled pixels written through — cost the 6 us the separate launch had cost.)
template <bool DEPTH_ONLY, bool FILL>
__global__ __launch_bounds__(256) void k_splat_resolve(ProjArgs a, SurfelPlanes sp, size_t cap, unsigned long long* __restrict__ zbuf,
                                                       uchar4* __restrict__ image, float4* __restrict__ vertex, float4* __restrict__ normal,
                                                       unsigned short* __restrict__ timeImg, float* __restrict__ depthOut, int clear_after,
                                                       FillArgs fa, TrackInitArgs ti) {
  // (round 6) the last ti.blocks blocks of the grid carry the set-up of the tracker call that follows two launches later (track_init.hpp):
  // it depends on the prior pose only, and doing it HERE - a kernel boundary before the model pyramid launch - lets that launch run
  // the tracker's SO3 stage beside the pyramid (track.hip: k_so3_model)
  const unsigned nres = gridDim.x - (FILL ? (unsigned)ti.blocks : 0u);
  if (FILL && blockIdx.x >= nres) {
    track_init_body((int)(blockIdx.x - nres), ti.blocks, (int)threadIdx.x, (int)blockDim.x, ti.st, ti.prior, ti.prior_pose16, ti.fx, ti.fy, ti.cx, ti.cy,
                    ti.so3, ti.first_level, ti.sync_words, ti.n_sync, ti.inject_timeout, nullptr);
    return;
  }
  if (FILL) {
    if (fa.mirror_words > 0 && blockIdx.x == 0 && (int)threadIdx.x < fa.mirror_words) fa.mirror_dst[threadIdx.x] = fa.mirror_src[threadIdx.x];
    if (fa.thumb_block && blockIdx.x == 0) {  // the frame block's pose and tick ride along (k_thumbnails does the same)
      if (fa.thumb_pose_dst && threadIdx.x < 16) fa.thumb_pose_dst[threadIdx.x] = fa.thumb_pose_src[threadIdx.x];
      if (fa.thumb_tick_dst && threadIdx.x == 16) *fa.thumb_tick_dst = fa.thumb_tick;
    }
  }
  // outputs are row-major (the tracker consumes them), the z-buffer is column-major: each wave takes an
  // 8 x 8 pixel tile, lanes running down the columns first, so that a z-buffer access touches 8 full
  // 64-byte lines (instead of 64 lines with a row-major thread map) and every output row of the tile
  // is one 128-byte (float4) run
  const int tiles_y = (a.rows + 7) >> 3, tiles_x = (a.cols + 7) >> 3, ntiles = tiles_x * tiles_y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves_per_block = blockDim.x >> 6;
  for (int t = xcd_block(blockIdx.x, nres, a.xcd) * waves_per_block + wave; t < ntiles; t += nres * waves_per_block) {
    const int tx = t / tiles_y, ty = t - tx * tiles_y;
    const int px = tx * 8 + (lane >> 3), py = ty * 8 + (lane & 7);
    if (px >= a.cols || py >= a.rows) continue;
    const int p = py * a.cols + px;
    const unsigned long long key = zbuf[(size_t)px * a.rows + py];
    if (clear_after) zbuf[(size_t)px * a.rows + py] = kZClear;
    bool sampled = false;
    if (FILL) {
      if (fa.dense_cnt) sampled = ((fa.sample_mask[px >> 5] >> (px & 31)) & (fa.sample_mask[64 + (py >> 5)] >> (py & 31)) & 1u) != 0u;
    }
    if ((unsigned)(key >> 32) >= 0xFFFFFFu) {
      if (DEPTH_ONLY) {
        depthOut[p] = 0.f;
      } else {
        image[p] = make_uchar4(0, 0, 0, 0);
        vertex[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        normal[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        timeImg[p] = 0;
        if (FILL) {
          float4 ov, on;
          uchar4 oi;
          fill_pixel_out(fa, px, py, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_uchar4(0, 0, 0, 0), ov, on, oi);
          if (fa.thumb_block) fill_thumb(fa, px, py, ov, on, oi);
        }
      }
      continue;
    }
    const unsigned i = (unsigned)(key & 0xFFFFFFFFull);
    SplatSurfel s;
    splat_vertex(a, sp.pos[i], sp.nrm + i, sp.times[(size_t)a.timeIdx * cap + i], s);
    f3 c;
    float zw;
    splat_fragment(a, s, px, py, c, zw);
    if (DEPTH_ONLY) {
      depthOut[p] = c.z;
      continue;
    }
    const float4 cc = sp.col[i];
    const f3 rgb = decode_color(cc.x);
    // RGBA8 render target: round(c * 255)
    const uchar4 o_img = make_uchar4((unsigned char)f2i_rn(rgb.x * 255.0f), (unsigned char)f2i_rn(rgb.y * 255.0f),
                                     (unsigned char)f2i_rn(rgb.z * 255.0f), 255);
    image[p] = o_img;
    if (FILL && sampled && o_img.x > 0 && o_img.y > 0 && o_img.z > 0) atomicAdd(fa.dense_cnt + ((px + py) & 15) * 16, 1u);
    const float z = c.z;
    const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
    const float4 o_v = make_float4(((fcx - a.cx) * z) * (1.f / a.fx), ((fcy - a.cy) * z) * (1.f / a.fy), z, s.conf);
    const float4 o_n = make_float4(s.nrm.x, s.nrm.y, s.nrm.z, s.rad);
    vertex[p] = o_v;
    normal[p] = o_n;
    // time = uint(colTime.z) into a 16-bit unsigned target
    const float tz = cc.z;
    unsigned tv = tz > 0.f ? (unsigned)f2i_rz(tz) : 0u;
    if (tv > 65535u) tv = 65535u;
    timeImg[p] = (unsigned short)tv;
    if (FILL) {
      float4 ov, on;
      uchar4 oi;
      fill_pixel_out(fa, px, py, o_v, o_n, o_img, ov, on, oi);
      if (fa.thumb_block) fill_thumb(fa, px, py, ov, on, oi);
    }
  }
}

int splat_predict(dms_model* m, const dms_pose_block* pose, const dms_camera* cam, float maxDepth, float confThreshold, int time,
                  int timeIdx, int maxTime, int timeDelta, int active, unsigned long long* zbuf, dms_predict_out* out,
                  dms_image2d* depth_out, int zclean, hipStream_t s, const float* second_conf_time_maxtime, unsigned long long* zbuf2,
                  int resolve_only, const FillArgs* fill, const TrackInitArgs* init) {
  DMS_REQUIRE(m && pose && cam && zbuf, "null argument");
  DMS_REQUIRE(!init || init->blocks == 0 || (fill && !depth_out), "the tracker set-up rides on the fused resolve + fill-in pass only");
  DMS_REQUIRE(!m->pending_update, "a deferred update pass is still pending (index_map applies it)");
  DMS_REQUIRE(timeIdx >= 0 && timeIdx < DMS_MAX_SENSORS, "timeIdx out of range");
  // (the project pass divides by fx, fy and 2 maxDepth through exact_arith.hpp: normal numbers far from the ends of the exponent range)
  DMS_REQUIRE(fabsf(cam->fx) > 1e-18f && fabsf(cam->fx) < 1e18f && fabsf(cam->fy) > 1e-18f && fabsf(cam->fy) < 1e18f && maxDepth > 1e-18f && maxDepth < 1e18f,
              "focal lengths and depth cut-off must be finite, non-zero and within 1e-18 .. 1e18");
  const int W = m->width, H = m->height;
  if (depth_out)
    DMS_REQUIRE(dense_img(*depth_out, 4, W, H), "depth target must be dense W×H f32");
  else
    DMS_REQUIRE(out && dense_img(out->image, 4, W, H) && dense_img(out->vertex, 16, W, H) && dense_img(out->normal, 16, W, H) &&
                    dense_img(out->time, 2, W, H),
                "prediction targets must be dense W×H");
  ProjArgs a;
  fill_proj(a, m, pose, cam, maxDepth, time, timeIdx, timeDelta);
  a.confThreshold = confThreshold;
  a.maxTime = maxTime;
  a.actv = active ? 1 : 0;
  const int n = W * H;
  if (!zclean) {
    hipLaunchKernelGGL(k_clear_zbuf, dim3(min((n + 255) / 256, 2048)), dim3(256), 0, s, zbuf, n);
    DMS_CHECK_LAUNCH();
  }
  SplatSecond b2 = {0.f, 0, 0};
  if (resolve_only) {
    // the z-buffer was filled by an earlier dual pass with exactly these parameters
  } else if (second_conf_time_maxtime && zbuf2) {
    b2.confThreshold = second_conf_time_maxtime[0];
    b2.time = (int)second_conf_time_maxtime[1];
    b2.maxTime = (int)second_conf_time_maxtime[2];
    hipLaunchKernelGGL(k_splat_project<true>, dim3(surfel_grid(m->count_upper)), dim3(256), 0, s, a, m->buf[m->cur], m->cap, m->d_count, zbuf, b2,
                       zbuf2);
  } else {
    hipLaunchKernelGGL(k_splat_project<false>, dim3(surfel_grid(m->count_upper)), dim3(256), 0, s, a, m->buf[m->cur], m->cap, m->d_count, zbuf,
                       b2, (unsigned long long*)nullptr);
  }
  DMS_CHECK_LAUNCH();
  const FillArgs no_fill = {};
  TrackInitArgs no_init;
  memset(&no_init, 0, sizeof(no_init));
  const dim3 rg(min((n + 255) / 256, 2048));
  if (depth_out) {
    hipLaunchKernelGGL((k_splat_resolve<true, false>), rg, dim3(256), 0, s, a, m->buf[m->cur], m->cap, zbuf, (uchar4*)nullptr,
                       (float4*)nullptr, (float4*)nullptr, (unsigned short*)nullptr, (float*)depth_out->data, zclean, no_fill, no_init);
  } else if (fill) {
    DMS_REQUIRE(fill->ex_image == (const uchar4*)out->image.data && fill->ex_vertex == (const float4*)out->vertex.data &&
                    fill->ex_normal == (const float4*)out->normal.data && fill->cols == W && fill->rows == H &&
                    (!fill->dense_cnt || (fill->sample_mask && W <= 2048 && H <= 2048)) && fill->mirror_words <= 256,
                "fill-in arguments do not describe this prediction");
    const TrackInitArgs& ti = (init && init->blocks > 0) ? *init : no_init;
    hipLaunchKernelGGL((k_splat_resolve<false, true>), dim3(rg.x + ti.blocks), dim3(256), 0, s, a, m->buf[m->cur], m->cap, zbuf, (uchar4*)out->image.data,
                       (float4*)out->vertex.data, (float4*)out->normal.data, (unsigned short*)out->time.data, (float*)nullptr, zclean, *fill, ti);
  } else {
    hipLaunchKernelGGL((k_splat_resolve<false, false>), rg, dim3(256), 0, s, a, m->buf[m->cur], m->cap, zbuf, (uchar4*)out->image.data,
                       (float4*)out->vertex.data, (float4*)out->normal.data, (unsigned short*)out->time.data, (float*)nullptr, zclean, no_fill, no_init);
  }
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

}  // namespace dms

extern "C" {
int dms_model_consume(dms_model* dst, const dms_model* src, const float* relativeTransform16, dms_stream s) {
  return dms::model_consume(dst, src, relativeTransform16, (hipStream_t)s);
}
int dms_model_consume_records(dms_model* dst, const float* records_dev, unsigned int count, const float* relativeTransform16, dms_stream s) {
  return dms::model_consume_records(dst, records_dev, count, relativeTransform16, (hipStream_t)s);
}
int dms_model_export_records(dms_model* m, float* records_dev, unsigned int max_count, unsigned int* count, dms_stream s) {
  return dms::model_export_records(m, records_dev, max_count, count, (hipStream_t)s);
}
}
