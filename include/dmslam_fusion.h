/*
 * dmslam_fusion.h — C ABI of the surfel-map half of the hot path: depth pre-filter,
 * surfel bootstrap, index map, splat prediction, fuse, clean, fill-in, resize, and the
 * per-camera frame step that chains them with the tracker of dmslam.h.
 *
 * Each entry point replaces one GLSL program (+ its C++ wrapper) of the reference; see
 * SURVEY.md §2.3 rows G1..G11 and the citations below.  The OpenGL rasteriser semantics the
 * reference relies on (point ownership, 24-bit depth test with GL_LESS, first-drawn-wins,
 * NEAREST texel fetch at computed coordinates, transform-feedback ordering) are restated as
 * explicit integer rules — DESIGN.md §"Rasteriser rules".
 *
 * Images are dense row-major device arrays:
 *   rgba8 (4 B/px) · depth u16 mm · metric depth f32 m · RGBA32F maps (16 B/px) · u32 index ·
 *   u16 time.  Poses are 4×4 row-major float, camera-to-world.
 */
#ifndef DMSLAM_FUSION_H_
#define DMSLAM_FUSION_H_

#include "dmslam.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* surfel map (reference GlobalModel, Core/src/GlobalModel.{h,cpp})            */
/* ------------------------------------------------------------------------- */
typedef struct dms_model dms_model;

/* HBM layout: structure-of-arrays of float4 planes (position+confidence, colour+times,
 * normal+radius) plus one float plane per sensor for the last-seen time, double-buffered for
 * the order-preserving compaction of `clean`.  capacity = maximum surfel count
 * (reference MAX_VERTICES = 5700^2, GlobalModel.cpp:22-24). */
int dms_model_create(dms_model** out, size_t capacity, int width, int height);
int dms_model_destroy(dms_model* m);
/* Number of time slots the clean's health test loops over (reference NUM_CAMERAS = 3, Shaders/size.glsl:2; default 3). */
int dms_model_set_num_sensors(dms_model* m, int num_sensors);
/* Map size (surfels) from which dms_model_clean compacts only the suffix behind the first removed surfel
 * (default 2^20; the environment variable DMS_CLEAN_SUFFIX_MIN is read once, at dms_model_create). */
int dms_model_set_clean_suffix_min(dms_model* m, size_t surfels);
int dms_model_count(dms_model* m, unsigned int* count, dms_stream s); /* syncs */
size_t dms_model_capacity(dms_model* m);
/* Host-side upper bound of the surfel count (launch grids are sized from it; the exact count lives on the device,
 * dms_model_count synchronises to read it). */
size_t dms_model_count_bound(dms_model* m);

/* reference surfel record (Shaders/Vertex.cpp:21-50): 15 floats =
 * pos.xyz conf | colour 0 initTime stamp | times[3] | normal.xyz radius (60 bytes).
 * download: reference GlobalModel::downloadMap (GlobalModel.cpp:866-896); sensors 0..2 only. */
int dms_model_download_ref(dms_model* m, float* host_vertices15, unsigned int max_count, unsigned int* count, dms_stream s);
int dms_model_upload_ref(dms_model* m, const float* host_vertices15, unsigned int count, dms_stream s);
/* full-width records: 12 + DMS_MAX_SENSORS floats = pos.xyz conf | colour 0 initTime stamp |
 * normal.xyz radius | times[DMS_MAX_SENSORS] */
int dms_model_download(dms_model* m, float* host_records, unsigned int max_count, unsigned int* count, dms_stream s);
int dms_model_upload(dms_model* m, const float* host_records, unsigned int count, dms_stream s);

/* ------------------------------------------------------------------------- */
/* operator layer — one per GLSL program                                       */
/* ------------------------------------------------------------------------- */

/* G1 depth_bilateral.frag:30-75 (ElasticFusion::filterDepth, ElasticFusion.cpp:759-768) */
int dms_depth_bilateral(const dms_image2d* depth_u16, dms_image2d* filtered_u16, float maxD, dms_stream s);
/* G2 depth_metric.frag:28-39 (ElasticFusion::metriciseDepth, ElasticFusion.cpp:748-757) */
int dms_depth_metric(const dms_image2d* depth_u16, dms_image2d* metric_f32, float maxD, dms_stream s);

/* G3+G4 vertex_feedback.{vert,geom} + init_unstable.vert: first-frame surfels
 * (FeedbackBuffer::compute, FeedbackBuffer.cpp:84-143; GlobalModel::initialise, GlobalModel.cpp:266-417).
 * Appends nothing: (re)initialises the map from one frame; surfels are emitted in column-major pixel order. */
int dms_model_initialise(dms_model* m, const dms_image2d* rgba, const dms_image2d* depth_metric,
                         const dms_image2d* depth_metric_filtered, const dms_camera* cam, int time, int timeIdx,
                         float maxDepth, dms_stream s);

/* Device half of Deformation::sampleGraphModel (Deformation.cpp:250-348; sample.vert / sample.geom):
 * every sampleRate-th surfel (gl_VertexID % sampleRate == 0; reference default 5000, --dgs) as
 * {pos.xyz, init time}, ordered by init time.  The reference downloads the samples and std::sorts
 * them on the host (equal times in unspecified order); here a stable sort runs on device (ties keep
 * surfel order) and only the sorted rows are copied: min(*n, max_rows) rows of 4 floats, ready for
 * Deformation::initialiseGraph.  Synchronises `s`.  Call between frames (uses the idle half of the
 * double-buffered map as scratch). */
int dms_model_sample_graph(dms_model* m, int sampleRate, float* rows4_host, int max_rows, int* n, dms_stream s);

/* Map merge: GlobalModel::consume (GlobalModel.cpp:898-993, consume.vert).  `dst` keeps its
 * surfels and appends those of `src` moved by the row-major 4x4 `relativeTransform16` (host pointer):
 * position transformed, normal rotated, confidence / radius / colour / per-sensor times unchanged.
 * DMS_ERR_CAPACITY when the sum may not fit.  The *_records forms move a map between devices or ranks:
 * export packs the map into 20-float records (pos4 col4 nrm4 times8, the dms_model_download layout) in
 * a 16-byte aligned device buffer (e.g. a tensor handed to an RCCL send), consume_records appends such
 * a buffer (e.g. just received). */
int dms_model_consume(dms_model* dst, const dms_model* src, const float* relativeTransform16, dms_stream s);
int dms_model_export_records(dms_model* m, float* records_dev, unsigned int max_count, unsigned int* count, dms_stream s);
int dms_model_consume_records(dms_model* dst, const float* records_dev, unsigned int count, const float* relativeTransform16,
                              dms_stream s);

/* index-map render target set (reference IndexMap index framebuffer, IndexMap.cpp:26-39) */
typedef struct dms_indexmap_out {
  dms_image2d index;      /* u32   */
  dms_image2d vertConf;   /* float4: camera-frame position, confidence */
  dms_image2d colorTime;  /* float4: colour, 0, initTime, times[timeIdx] */
  dms_image2d normRad;    /* float4: camera-frame normal, radius */
} dms_indexmap_out;

/* G5 index_map.{vert,frag} (IndexMap::predictIndices, IndexMap.cpp:146-217).
 * zbuf: W*H u64 scratch.  pose_dev: device pointer to dms_pose_block. */
typedef struct dms_pose_block {
  float pose[16];  /* camera-to-world */
  float t_inv[16]; /* inverse */
} dms_pose_block;
int dms_pose_block_set(dms_pose_block* dev, const float* pose16_host, dms_stream s);

int dms_index_map(dms_model* m, const dms_pose_block* pose_dev, const dms_camera* cam, int time, int timeIdx,
                  float maxDepth, int timeDelta, unsigned long long* zbuf, dms_indexmap_out* out, dms_stream s);

/* splat prediction targets (reference combined / old framebuffers, IndexMap.cpp:59-99) */
typedef struct dms_predict_out {
  dms_image2d image;  /* rgba8  */
  dms_image2d vertex; /* float4 */
  dms_image2d normal; /* float4 */
  dms_image2d time;   /* u16    */
} dms_predict_out;

/* G6 splat.vert + combo_splat.frag (IndexMap::combinedPredict, IndexMap.cpp:253-368).
 * active = 1 for IndexMap::ACTIVE, 0 for INACTIVE. */
int dms_splat_predict(dms_model* m, const dms_pose_block* pose_dev, const dms_camera* cam, float maxDepth,
                      float confThreshold, int time, int timeIdx, int maxTime, int timeDelta, int active,
                      unsigned long long* zbuf, dms_predict_out* out, dms_stream s);
/* G6' splat.vert + depth_splat.frag (IndexMap::synthesizeDepth, IndexMap.cpp:370-452) */
int dms_splat_depth(dms_model* m, const dms_pose_block* pose_dev, const dms_camera* cam, float maxDepth,
                    float confThreshold, int time, int timeIdx, int maxTime, int timeDelta,
                    unsigned long long* zbuf, dms_image2d* depth_f32, dms_stream s);

/* G7+G8 data.{vert,geom,frag} + update.vert (GlobalModel::fuse, GlobalModel.cpp:513-694).
 * weighting_dev may be NULL (then `weighting` is used); otherwise the weight is read on device. */
int dms_model_fuse(dms_model* m, const dms_pose_block* pose_dev, int time, int timeIdx, const dms_image2d* rgba,
                   const dms_image2d* depth_metric, const dms_image2d* depth_metric_filtered,
                   const dms_indexmap_out* indexmap, const dms_camera* cam, float depthCutoff, float weighting,
                   const float* weighting_dev, dms_stream s);

/* G9 copy_unstable.{vert,geom} (GlobalModel::clean, GlobalModel.cpp:696-853).
 * graph: host array of 16 floats / node (Deformation.cpp:192-201), may be NULL / 0 nodes.
 * depth_synth: float depth image used only when nodes > 0. */
int dms_model_clean(dms_model* m, const dms_pose_block* pose_dev, int time, int timeIdx,
                    const dms_indexmap_out* indexmap, const dms_image2d* depth_synth, const dms_camera* cam,
                    float confThreshold, const float* graph_host, int graph_nodes, int timeDelta, float maxDepth,
                    int isFern, dms_stream s);

/* G10 fill_{vertex,normal,rgb}.frag (FillIn::{vertex,normal,image}, Shaders/FillIn.cpp:65-190) */
int dms_fill_in(const dms_predict_out* existing, const dms_image2d* depth_filtered_u16, const dms_image2d* rgba,
                const dms_camera* cam, int passthrough_geom, int passthrough_rgb, dms_predict_out* filled, dms_stream s);

/* G11 resize.frag (Resize::{image,vertex,time}, Shaders/Resize.cpp:67-170): nearest subsample.
 * elem_bytes = 4 (rgba8), 16 (float4) or 2 (u16). */
int dms_resize_nn(const dms_image2d* src, dms_image2d* dst, int elem_bytes, dms_stream s);

/* ------------------------------------------------------------------------- */
/* object layer — one camera's frame step (ElasticFusion::processFrame)        */
/* ------------------------------------------------------------------------- */
typedef struct dms_fusion dms_fusion;

/* subset of the reference's ElasticFusion constructor arguments that reach the hot path
 * (ElasticFusion.cpp:22-73) plus the per-camera options read in processFrame */
typedef struct dms_fusion_params {
  int width, height;
  float fx, fy, cx, cy;
  int timeDelta;             /* default 200 */
  float confidence;          /* confidenceThreshold, default 10 */
  float depthCut;            /* depthCutoff (m), default 3 */
  float icpWeight;           /* icpThresh, default 10 */
  int fastOdom;              /* --fo */
  int so3;                   /* !--nso */
  int frameToFrameRGB;       /* --ftf */
  int pyramid;               /* GUI toggle, default 1 */
  int hybrid_tracking;       /* --hybrid_tracking: refine the prior with the dense tracker */
  int rgbOnly;               /* Context::rgbOnly() */
  int timeIdx;               /* Context::id(): which per-sensor time slot this camera uses */
  float maxDepthProcessed;   /* 25 (ElasticFusion.cpp:56) */
  size_t model_capacity;     /* 0 = reference MAX_VERTICES */
  int pipeline_ingest;       /* 1 (default): ingest, depth filter and the live pyramids of a frame run on an
                                internal second stream and overlap the previous frame's tracking / fusion.
                                Contract: rgb_dev / depth_dev are complete when process_frame is called and stay
                                untouched until work enqueued on `stream` after the call would run.
                                0: everything is issued on `stream`. */
  int global_predict;        /* 1: also run the post-tracking "GlobalPredict" (ElasticFusion.cpp:273), whose
                                consumers (ferns / loop closure) the reference compiles out; its images are
                                overwritten by the final predict either way.  Default 0. */
  /* NID key-framing gate of ElasticFusion::fuseFrame (ElasticFusion.cpp:639-677); 0 = off (--nkf).
   * When on, the post-tracking prediction is compared with the live frame (dms_computeNIDImg /
   * dms_computeNIDDepth at pyramid level nid_pyramid_level) and the fusion half runs only when
   * nid_depth_lambda * nid_depth + (1 - nid_depth_lambda) * nid_img > nid_threshold.  The decision is
   * taken on the host, as in the reference: the frame synchronises once mid-way (no pipelining). */
  int nid_keyframing;
  float nid_threshold;       /* 0.80 (ElasticFusion.h:73) */
  float nid_depth_lambda;    /* 0.7 */
  int nid_bins_img;          /* 64 */
  int nid_bins_depth;        /* 500 */
  int nid_pyramid_level;     /* 0 */
  /* Device half of the local loop-closure block (`closeLoops`, ElasticFusion.cpp:399-497, the branch
   * without a fern match): INACTIVE prediction of the map (IndexMap::combinedPredict, window
   * tick - timeDelta), model-to-model tracking of the ACTIVE against the INACTIVE view (a second
   * RGBDOdometry, icpWeight 10, no SO3), the acceptance test (covariance diagonal <= 8e-5,
   * lastICPCount > 15000, lastICPError < 3e-4) and the W/20 x H/20 surface-constraint sampling.
   * Everything stays on the stream; the outcome is reported in dms_frame_result.loop_* and
   * dms_fusion_get_loop_constraints.  The reference then hands the constraints to its CPU/CHOLMOD
   * deformation solver (Deformation::constrain, :481), which is the caller's: the pose is left as
   * tracked.  Implies global_predict.  This is the "full" frame step of the measurement contract
   * (tracking runs twice per frame).  Default 0 (--o). */
  int local_loop_closure;
  /* --rl: tracking-failure detection of ElasticFusion.cpp:204-244.  A frame whose lastICPError is not
   * < 1e-4 or whose pose covariance has a diagonal entry > 1e-4 is tracked but not fused; more than
   * 10 such frames in a row mark the camera lost: no fusion, no loop closure, the tick stops, fill-in
   * passes the raw frame through (:588, :706-712).  (Recovery needs the fern relocaliser, which this
   * reference compiles out, :351-393: a lost camera stays lost.)  The decision is taken on the host
   * after one mid-frame synchronisation, as in the reference.  Default 0. */
  int reloc;
  /* Number of per-surfel time slots that take part in the clean's "unhealthy for every sensor" test
   * (copy_unstable.vert:137-150 loops over vTimes.length() = NUM_CAMERAS = 3, Shaders/size.glsl:2).
   * Default 3 as in the reference; a node that merges more than three cameras into one map raises it
   * (<= DMS_MAX_SENSORS); timeIdx must be < num_sensors. */
  int num_sensors;
  /* 1 (default): the project pass of a frame's final prediction (ElasticFusion.cpp:586) also fills the z-buffer of the
   * NEXT frame's tracking prediction (:165, confidence 0.7, next tick) — the same map from the same pose, a different
   * cull — and that frame resolves it instead of projecting the map again, provided nothing changed the map in between
   * and it brings no pose prior.  Same images bit for bit; one map pass less per frame.  0: every prediction projects. */
  int share_projection;
  /* 1 (default): the resolve pass of every prediction of the frame step also fills the holes of the pixel it has just
   * resolved (FillIn::vertex / normal / image, ElasticFusion.cpp:704-712) and its last block takes the denseEnough
   * decision (:84-97) — the separate fill-in launch and its re-read of the three images are saved.  Same images bit for
   * bit.  0: predict, then dms_fill_in as its own pass. */
  int fused_fill_in;
  /* Options::hybrid_loops: the ORB-SLAM3 front end may hand over a loop closure (orbTcwOld, orbTcwNew) with a frame
   * (ElasticFusion.cpp:292-350) or through applyGlobalLoop (:1148-1240); 1 allocates the INACTIVE-view images and the
   * constraint buffer those blocks need.  Default 0. */
  int hybrid_loops;
} dms_fusion_params;

void dms_fusion_default_params(dms_fusion_params* p, int width, int height, float fx, float fy, float cx, float cy);

int dms_fusion_create(dms_fusion** out, const dms_fusion_params* p);
int dms_fusion_destroy(dms_fusion* f);

/* Per-frame outputs */
typedef struct dms_frame_result {
  float pose[16];          /* camera-to-world after tracking */
  unsigned int surfels;    /* map size after clean */
  int tick;                /* Context::tick() after the frame */
  int fused;               /* 1 when the fusion half ran */
  int fill_in;             /* shouldFillIn decision (ElasticFusion.cpp:167) */
  float weighting;         /* velocity weight (ElasticFusion.cpp:252-268) */
  float nid_score;         /* Context::nidScores().back() (0 with key-framing off) */
  dms_track_result track;  /* tracker side outputs */
  /* local loop-closure candidate (local_loop_closure = 1; zero otherwise) */
  int loop_ok;               /* acceptance test passed (ElasticFusion.cpp:441-442) */
  int loop_constraints;      /* surface constraints sampled (:446-474); 0 unless loop_ok */
  float loop_icp_error;      /* modelToModel().lastICPError */
  float loop_icp_count;      /* modelToModel().lastICPCount */
  float loop_pose[16];       /* estPose: the ACTIVE view registered onto the INACTIVE one */
  double loop_cov_diag[6];   /* diagonal of modelToModel().getCovariance() */
  int tracking_ok;           /* trackingOk of this frame (always 1 without reloc) */
  int lost;                  /* Context::lost() after this frame */
} dms_frame_result;

/* ElasticFusion::processFrame (ElasticFusion.cpp:99-637) for one camera with loop closure off
 * (--o) and NID keyframing off (--nkf): upload-free — rgb (RGB8 or RGBA8 per `rgb_channels`)
 * and depth (u16 mm) are device pointers already resident in HBM.  inPose: host 4×4 prior or
 * NULL (then the previous pose is used; the reference would dereference NULL,
 * ElasticFusion.cpp:164).  Asynchronous on `s`; dms_fusion_fetch syncs and returns the result. */
int dms_fusion_process_frame(dms_fusion* f, const void* rgb_dev, int rgb_channels, const unsigned short* depth_dev,
                             const float* inPose16, float weightMultiplier, dms_stream s);
/* Returns DMS_ERR_CAPACITY (with the result filled in) once the map has reached model_capacity: the
 * kernels stop appending there instead of asserting like the reference (GlobalModel.cpp:703).
 * Returns DMS_ERR_TIMEOUT (result filled in) when ANY frame since the previous fetch had a resident
 * tracker kernel give up at a grid barrier (possible only when its blocks were not all resident, e.g.
 * a second process on the same GPU): such a frame keeps its prior pose and fuses nothing — the map is
 * never updated from an invalid pose — and the event is counted on the device, so pipelined callers
 * that fetch once per batch still learn about it. */
int dms_fusion_fetch(dms_fusion* f, dms_frame_result* r, dms_stream s);

/* Input-buffer ordering with pipeline_ingest = 1 (the frame's ingest runs on an internal stream).
 * Contract: rgb_dev / depth_dev are complete when process_frame[_begin] is called and stay untouched
 * until that frame's ingest has run.  Two helpers make both halves enforceable:
 *   dms_fusion_inputs_ready(f, producer)  the next frame's inputs are produced by work enqueued on
 *       `producer` so far (async H2D copy, decode kernel): the ingest waits for that point.  Naming the
 *       stream the frames run on is legal but serialises the ingest behind the previous frame.
 *   dms_fusion_inputs_consumed(f, s)      blocks the host until the last enqueued frame's ingest has read
 *       its input buffers (then they may be overwritten); `s` = the frames' stream (used when
 *       pipeline_ingest = 0). */
int dms_fusion_inputs_ready(dms_fusion* f, dms_stream producer);
int dms_fusion_inputs_consumed(dms_fusion* f, dms_stream s);

/* The same frame step in two halves, for callers that close local loops: `_begin` enqueues
 * everything up to and including the loop candidate (and takes the NID decision); the host may
 * then read the candidate (dms_fusion_fetch_loop: synchronises `s`; fills pose, fill_in, weighting,
 * nid_score and the loop_* fields; constraints through dms_fusion_get_loop_constraints), run the
 * reference's Deformation::constrain on it (ElasticFusion.cpp:481) and hand the outcome to `_end`:
 * graph_host = the deformation graph `rawGraph` (graph_nodes x 16 floats, the layout of
 * dms_model_clean; 0 nodes = no deformation), newPose16 = the corrected pose (estPose, :489) or
 * NULL.  `_end` runs the fusion half as ElasticFusion.cpp:506-591 does on such a frame: the frame
 * always fuses, IndexMap::synthesizeDepth is rendered before the clean, and the clean applies the
 * graph.  dms_fusion_process_frame == _begin + _end(NULL, 0, NULL). */
int dms_fusion_process_frame_begin(dms_fusion* f, const void* rgb_dev, int rgb_channels, const unsigned short* depth_dev,
                                   const float* inPose16, float weightMultiplier, dms_stream s);
int dms_fusion_fetch_loop(dms_fusion* f, dms_frame_result* r, dms_stream s);
int dms_fusion_process_frame_end(dms_fusion* f, const float* graph_host, int graph_nodes, const float* newPose16, dms_stream s);

/* ORB-triggered global loop closure, device half (the block `if (hybrid_loops && orbTcwOld && orbTcwNew)` of
 * ElasticFusion::processFrame, ElasticFusion.cpp:292-350).  dms_fusion_set_orb_loop arms the NEXT
 * dms_fusion_process_frame_begin with the two poses (row-major 4x4 camera-to-world; NULL, NULL disarms): after tracking, that
 * frame predicts the ACTIVE view at orbTcwOld, samples its vertex map on the W/20 x H/20 grid (Resize::vertex), predicts the
 * INACTIVE view at orbTcwNew, samples its time map (Resize::time) and emits one constraint row per sample with
 * 0 < z < maxDepthProcessed: {orbTcwOld * p, orbTcwNew * p, time} = the arguments of Deformation::addConstraint (:319-323),
 * columns outer / rows inner as :303-304; then the current view is predicted again (:349).  The host reads the rows
 * (dms_fusion_get_global_loop_constraints: synchronises `s`), runs the reference's Deformation::constrain (:337, CPU/CHOLMOD,
 * the caller's) and finishes the frame with dms_fusion_process_frame_end(graph, nodes, NULL) as for a local loop. */
int dms_fusion_set_orb_loop(dms_fusion* f, const float* orbTcwOld16, const float* orbTcwNew16);
int dms_fusion_get_global_loop_constraints(dms_fusion* f, float* rows7_host, int max_rows, int* n, dms_stream s);
/* ElasticFusion::applyGlobalLoop (ElasticFusion.cpp:1148-1240) in two halves around the caller's Deformation::constrain:
 * `_begin`: ACTIVE prediction at orbTcwNew, INACTIVE at orbTcwOld, constraint rows as above but only where the INACTIVE time
 * is non-zero (:1170-1173); `_end`: predict, predictIndices, clean with the deformation graph and isFern = accepted
 * (:1222-1239).  Outside a frame (between dms_fusion_process_frame calls). */
int dms_fusion_apply_global_loop_begin(dms_fusion* f, const float* orbTcwOld16, const float* orbTcwNew16, dms_stream s);
int dms_fusion_apply_global_loop_end(dms_fusion* f, const float* graph_host, int graph_nodes, int accepted, dms_stream s);

/* After a map merge: several cameras, ONE map.  ReferenceFrame::consumeReferenceFrame (ReferenceFrame.h:121-150) appends the consumed
 * map to the consuming one, moves the consumed map's cameras into the consuming reference frame (currPose = relativeTransform *
 * currPose) and from then on every camera of that frame tracks against and fuses into the one map, each with its own time slot
 * (timeIdx = Context::id(); MainController.cpp:262-400 runs them one after the other).
 *   dms_fusion_join_map      both cameras live in this process (two contexts on one device, as in the reference): `owner`'s map
 *                            consumes `f`'s (dms_model_consume with relativeTransform), f's pose is re-based, and f's later frames
 *                            use owner's map.  f keeps its tracker state (tick, last image pyramid): the very next
 *                            dms_fusion_process_frame continues the camera.
 *   dms_fusion_import_camera the camera arrives from another rank (its map records went through dms_model_consume_records, its
 *                            key frames through dms_ferns_consume_records): `f` is a fresh context created with the camera's
 *                            timeIdx; pose16 = relativeTransform * the camera's pose, tick = its tick, last_rgb / last_depth = the
 *                            last frame it processed, from which the context rebuilds the live state that frame left behind
 *                            (the intensity pyramid the next frame's SO3 pre-alignment reads).  Nothing is fused by this call.
 * Rules: the contexts of one map are driven from ONE host thread on ONE stream, one frame at a time (what the reference does);
 * `owner` must outlive the contexts that joined it; timeIdx values differ and are < the map's num_sensors.  Both calls synchronise. */
int dms_fusion_join_map(dms_fusion* f, dms_fusion* owner, const float* relativeTransform16, dms_stream s);
int dms_fusion_import_camera(dms_fusion* f, dms_fusion* owner, const float* pose16, int tick, const void* last_rgb_dev, int rgb_channels,
                             const unsigned short* last_depth_dev, dms_stream s);
/* relativeTransform = recoveryPose * currPose.inverse() (ReferenceFrame.h:98) and c = a * b, row-major 4 x 4 floats on the host, in
 * the library's fixed operation order (so that two hosts compute the same bits) */
int dms_relative_transform(const float* recoveryPose16, const float* currPose16, float* out16);
int dms_pose_compose(const float* a16, const float* b16, float* out16);

/* The reference frame's own IndexMap + RGBDOdometry (ReferenceFrame.h:203-214: m_index, m_rgbd) and the second half of
 * ReferenceFrame::resolveRelativeTransformationFern (ReferenceFrame.h:66-110), run by the OWNER of the queried map after
 * Ferns::findFrame(..., interMap = true) (dms_ferns_find_frame[_thumbs], dmslam_ferns.h) returned a recoveryPose for a camera of
 * another map: INACTIVE prediction of `map` at recoveryPose (time 0, the querying camera's time slot `timeIdx`, maxTime = its tick,
 * :72-80), tracker initialised from that prediction (model side) and the querying camera's fill-in vertex / normal / colour textures
 * (live side: dense W x H RGBA32F, RGBA32F, RGBA8 in this device's HBM - dms_fusion_get_image 14 / 15 / 13 on the same device, one
 * point-to-point transfer from the camera's rank otherwise), getIncrementalTransformation(t, r, false, 10, true, false, true, true)
 * (:88-90: SO3 + 3 x 50 ICP + RGB iterations), relativeTransform = refined pose * currPose^-1 (:95), accepted when every covariance
 * diagonal <= covThresh, lastICPError < icpErrThresh and lastICPCount > icpCountThresh (:98-110; Options.h:91-94 defaults 1e-5, 2e-5,
 * 35000).  depthCutoff is an int as in the reference's signature (:42: maxDepthProcessed truncated).  The tracker keeps its state
 * between calls like m_rgbd does (the SO3 pre-alignment compares with the previous refinement's live image).  Two readings of the
 * compiled-out reference block are stated in csrc/refframe.hip (the model colour image; the first call's lastNextImage).
 * Synchronises (the acceptance test is host arithmetic on the tracker's result, as in the reference). */
typedef struct dms_refframe dms_refframe;
typedef struct dms_intermap_result {
  int accepted; /* the function's return value in the reference */
  int cov_ok;
  float relativeTransform[16]; /* row-major */
  float refinedPose[16];       /* recoveryPose after the refinement */
  double cov_diag[6];
  float lastICPError, lastICPCount, lastRGBError, lastRGBCount;
  int iterations_run[DMS_NUM_PYRS];
  int so3_iterations_run;
} dms_intermap_result;
int dms_refframe_create(dms_refframe** out, int width, int height, float cx, float cy, float fx, float fy);
int dms_refframe_destroy(dms_refframe* r);
int dms_refframe_refine(dms_refframe* r, dms_model* map, const float* recoveryPose16, const float* currPose16, const float* vertex_dev,
                        const float* normal_dev, const void* image_rgba_dev, int depthCutoff, float confidenceThreshold, int timeIdx,
                        int timeDelta, int maxTime, float covThresh, float icpErrThresh, float icpCountThresh, dms_intermap_result* out,
                        dms_stream s);
dms_odometry* dms_refframe_odometry(dms_refframe* r);                        /* m_rgbd (tests, profiling) */
int dms_refframe_get_prediction(dms_refframe* r, dms_predict_out* view);     /* m_index's INACTIVE targets of the last refinement */

/* ElasticFusion::predict(context, rf[, confidence]) (ElasticFusion.h:107-108, ElasticFusion.cpp:688-746): the ACTIVE model view at the
 * camera's current pose + fill-in, into the prediction / fill-in images (dms_fusion_get_image 9-15) - what the GUI loop calls for a
 * paused camera (MainController.cpp:398).  confidence < 0: the context's confidence threshold.  Outside a frame. */
int dms_fusion_predict(dms_fusion* f, float confidence, dms_stream s);

dms_model* dms_fusion_model(dms_fusion* f);
/* Device address of the camera pose (16 floats, row-major, camera-to-world) the frame step keeps in HBM: valid for the
 * life of the context, written by the tracker's last kernel — stream-ordered consumers (e.g. dms_ferns_add_frame_async)
 * read it without a host round trip. */
const float* dms_fusion_pose_device(dms_fusion* f);
dms_odometry* dms_fusion_odometry(dms_fusion* f);
/* device images owned by the context; which: 0 rgb(rgba8) 1 depth_raw 2 depth_filtered 3 depth_metric
 * 4 depth_metric_filtered 5 index 6 vertConf 7 colorTime 8 normRad 9 pred image 10 pred vertex
 * 11 pred normal 12 pred time 13 fill image 14 fill vertex 15 fill normal; with local_loop_closure
 * also the INACTIVE view: 16 old image 17 old vertex 18 old normal 19 old time */
int dms_fusion_get_image(dms_fusion* f, int which, dms_image2d* view);

/* Collaborative mode: the per-frame block a camera publishes to the other ranks — the
 * W/8 x H/8 NEAREST thumbnails of its fill-in image (RGBA8), vertex and normal maps (RGBA32F), the
 * inputs of the inter-map fern matcher (Ferns.cpp:277-423) — packed [image | vertex | normal] into
 * dms_thumb_block_bytes(W, H) bytes of device memory ((W/8)(H/8) * 36 when that pixel count is a multiple of 4), in one launch on `s`
 * (stream-ordered after the frame). */
int dms_fusion_thumbnails(dms_fusion* f, void* block_dev, dms_stream s);
/* Size and layout of that block for a W x H camera: n = (W / 8)(H / 8) pixels (integer division, as Ferns.cpp:23-24 sizes its
 * thumbnails), [RGBA8 image: n * 4 bytes, padded to a multiple of 16 | RGBA32F vertex: n * 16 | RGBA32F normal: n * 16].  For n a
 * multiple of 4 (every size whose thumbnails have an even pixel count in fours, e.g. 640 x 480) that is n * 36 bytes, packed; at
 * 1241 x 376 (155 x 47) the image section carries 12 bytes of padding. */
size_t dms_thumb_block_bytes(int width, int height);
void dms_thumb_block_offsets(int width, int height, size_t* vertex_offset, size_t* normal_offset);
/* The same launch also copies the frame's pose (16 floats, from its place in HBM) to pose16_dst_dev and writes `tick` to
 * tick_dst_dev (either may be NULL): everything of a published frame block that must be taken before the next frame
 * starts, so that the rest of the exchange (encoding, key-frame database, all-gather, search) can run on a side stream. */
int dms_fusion_frame_block(dms_fusion* f, void* block_dev, float* pose16_dst_dev, int* tick_dst_dev, int tick, dms_stream s);
/* The same block WITHOUT a launch of its own (round 6): armed before a frame, it is written by the frame step's last kernel - the
 * resolve + fill-in pass of the final prediction (ElasticFusion.cpp:586, :704-712), whose threads hold the filled image / vertex /
 * normal of their pixel: the ones that own a thumbnail sample store it, block 0 copies the pose and the tick.  Same bytes as
 * dms_fusion_frame_block after the frame.  The arming holds for the NEXT dms_fusion_process_frame / _end only;
 * dms_fusion_frame_block_written(f) tells whether that frame did write it (1) or ran a path without the fused fill-in (0: fused_fill_in
 * off, images wider than 2048, the map's first frame) - the caller then calls dms_fusion_frame_block as before. */
int dms_fusion_arm_frame_block(dms_fusion* f, void* block_dev, float* pose16_dst_dev, int* tick_dst_dev, int tick);
int dms_fusion_frame_block_written(dms_fusion* f);
/* Makes `waiter` wait for the last frame enqueued on this context - through the event that frame's enqueue recorded anyway (no marker of
 * the caller's own behind the frame: between two kernels of one stream a marker costs about 7 us on this device). */
int dms_fusion_wait_frame_done(dms_fusion* f, dms_stream waiter);

/* Surface constraints of the last fetched frame's loop candidate, in the reference's sampling
 * order (columns outer, rows inner, ElasticFusion.cpp:446-447): per row
 * {worldRawPoint xyz, worldModelPoint xyz, source time} = the arguments of
 * Deformation::addConstraint (:468-470).  Copies min(*n, max_rows) rows of 7 floats to host memory. */
int dms_fusion_get_loop_constraints(dms_fusion* f, float* rows7_host, int max_rows, int* n);
/* Run-time setters of the reference's ElasticFusion (ElasticFusion.h:168-226, ElasticFusion.cpp:1023-1043): the GUI of the
 * reference drives them every frame (MainController.cpp:760-775), and BASELINE config 2 ("single pyramid") is reachable through
 * setPyramid only.  Takes effect with the next dms_fusion_process_frame; between frames only (DMS_ERR_STATE inside a
 * begin / end pair).  (Nothing rendered ahead depends on them: the projection a frame shares with the next one's tracking
 * prediction uses that prediction's fixed 0.7 threshold, ElasticFusion.cpp:165.) */
enum {
  DMS_OPT_RGB_ONLY = 0,            /* setRgbOnly            (bool)  */
  DMS_OPT_ICP_WEIGHT = 1,          /* setIcpWeight          (float) */
  DMS_OPT_PYRAMID = 2,             /* setPyramid            (bool)  */
  DMS_OPT_FAST_ODOM = 3,           /* setFastOdom           (bool)  */
  DMS_OPT_SO3 = 4,                 /* setSo3                (bool)  */
  DMS_OPT_FRAME_TO_FRAME_RGB = 5,  /* setFrameToFrameRGB    (bool)  */
  DMS_OPT_CONFIDENCE = 6,          /* setConfidenceThreshold(float) */
  DMS_OPT_DEPTH_CUTOFF = 7,        /* setDepthCutoff        (float) */
  DMS_OPT_NID_THRESHOLD = 8,       /* nidThreshold() = v    (float) ElasticFusion.h:340 */
  DMS_OPT_NID_DEPTH_LAMBDA = 9,    /* nidDepthLambda() = v  (float) :346 */
  DMS_OPT_NID_BINS_IMG = 10,       /* setNumBinsImg         (int, <= the creation-time value's workspace) :347 */
  DMS_OPT_NID_BINS_DEPTH = 11,     /* setNumBinsDepth       (int) :353 */
  DMS_OPT_NID_PYRAMID_LEVEL = 12,  /* nidPyramidLevel() = v (int 0..2) :362 */
  DMS_OPT_COUNT = 13
};
int dms_fusion_set_option(dms_fusion* f, int option, double value);
/* dms_odometry_set_resident_budget (dmslam.h) for this context's trackers (frame-to-model and, with local loop closure, model-to-model);
 * between frames only. */
int dms_fusion_set_tracker_budget(dms_fusion* f, int max_blocks, int unchained);
/* The "late frame": a context that is driven ALONE lets the host wait for the live half of a frame (the prep stream's event) and enqueue
 * the frame behind it, instead of a barrier packet on the frame's queue (about 10 us per frame; +1.2 - 1.4 % frame rate) - while the host
 * has the slack, which the context watches.  By default a context takes itself to be alone when it is the only live dms_fusion context
 * of the process.  An owner that knows better says so: allow = 1 (this context's frames are the only work the calling thread enqueues
 * between two of its frames), 0 (several contexts are driven by turns, or the thread has other work between two frames: dms_session says
 * 0 for its cameras), -1 (back to the default).
 * Timing only; DMS_LATE_MAIN=0 / 1 forces it for every context. */
int dms_fusion_allow_late_frame(dms_fusion* f, int allow);
int dms_fusion_get_option(dms_fusion* f, int option, double* value);

/* End-of-run exports of the reference (MainController.cpp:806-807), host side, byte for byte the reference's files:
 * dms_model_save_ply = ElasticFusion::savePly (ElasticFusion.cpp:781-885) for one map: ASCII header, then per surfel with
 * confidence > confidenceThreshold {x y z (float) r g b (uchar) -nx -ny -nz radius (float)}, little endian.  The reference
 * reads the normal at float offset 18 of its 15-float record — a constant left over from MAX_SENSORS = 10 (Vertex.cpp:49,
 * 8 + 10); with 3 sensors that is the NEXT record's {confidence, colour...} and, for the last surfel, beyond the buffer.
 * reference_offsets = 0 (default use): the normal and radius of the surfel itself (offset 8 + MAX_SENSORS = 11);
 * reference_offsets = 1: the reference's offset, zeros where it leaves the buffer.  Synchronises the null stream only: call it
 * once the last frame step on the map has completed (dms_fusion_fetch, or a synchronised stream), as the reference's savePly
 * follows its glFinish (GlobalModel.cpp:867). */
int dms_model_save_ply(dms_model* m, const char* path, float confidenceThreshold, int reference_offsets, unsigned int* written);
/* Context::saveTrajectory (Context.h:117-156): one line per pose, the 3 x 4 matrix row by row with the stream's default float
 * formatting (6 significant digits) and a blank before the newline.  poses16_host: n row-major 4 x 4 camera-to-world matrices. */
int dms_trajectory_save(const char* path, const float* poses16_host, size_t n);

/* ---- per-cluster surfel buffers (GlobalModel.h:50,93-109, GlobalModel.cpp:251-277; ElasticFusion::processFrame's `cluster`
 * argument, ElasticFusion.h:98, ElasticFusion.cpp:105,138,508-515; MainController.cpp:373-377 passes the ground-truth cluster of
 * the frame) ----
 * dms_fusion_set_cluster: the id the following frames are processed under (default 0).  A frame that FUSES under an id the map
 * does not know gets surfel buffers of its own, filled from the context's feedback buffers — those of the first frame unless
 * dms_fusion_compute_feedback (Context::computeFeedbackBuffers, ElasticFusion.h:275 / Context.h:211) has refreshed them from the
 * last processed frame — and those buffers become the current ones for everything (index map, fuse, clean, predictions,
 * dms_fusion_model, counts, exports).  As in the reference nothing ever switches back: a known id, current or not, changes
 * nothing.  Not supported on a map several cameras share (DMS_ERR_INVALID_ARG from the frame).
 * dms_fusion_clusters: GlobalModel::clusters() (ids in ascending order; *n_ids = how many there are) and the current id;
 * dms_fusion_cluster_model: a cluster's map (isCluster = non-null; NULL also for the constructor's empty cluster 0 of a context
 * whose first frame ran under another id), owned by the context. */
int dms_fusion_set_cluster(dms_fusion* f, int cluster);
int dms_fusion_compute_feedback(dms_fusion* f, dms_stream stream);
int dms_fusion_clusters(dms_fusion* f, int* ids, int max_ids, int* n_ids, int* current);
dms_model* dms_fusion_cluster_model(dms_fusion* f, int cluster);

int dms_fusion_set_profiling(dms_fusion* f, int enabled);
int dms_fusion_get_kernel_time(dms_fusion* f, const char* name, double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* DMSLAM_FUSION_H_ */
