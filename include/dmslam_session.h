/* dmslam_session.h — the collaborative session behind the C ABI.
 *
 * What it replaces.  The reference serves every camera of a session from ONE loop in ONE process
 * (GUI/src/MainController.cpp:262-400: for every log reader -> frontend(name) -> processFrame), and the inter-map block of
 * ElasticFusion::processFrame (Core/src/ElasticFusion.cpp:595-632, compiled out there with `if (false)`) lets a camera query every
 * OTHER reference frame (ReferenceFrame::resolveRelativeTransformationFern, ReferenceFrame.h:34-110) and, on success, has that frame
 * consume the camera's own (consumeReferenceFrame, :121-150).  Here the cameras of a session are spread over the GPUs of a node, one
 * process per GPU; dms_session is that loop for one rank: the same object on every rank, driven in lock step, one call per tick.
 *
 * Placement.  Camera c is READ on rank c % world for the whole session (its frames arrive there) and HOSTED - its dms_fusion context
 * lives, its frames are processed - on the rank that hosts its reference frame: c % world until the frame is consumed, then the
 * consuming frame's rank.
 *
 * One tick, dms_session_step (the protocol of densemonoslam_amd/session.py, which remains as the model the CPU tests run over gloo
 * with oracle-backed stand-ins):
 *   1. forward   a camera hosted elsewhere: its frame (RGB8 + depth u16) goes point to point to the host;
 *   2. frames    every hosted camera's dms_fusion_process_frame, in camera-id order (the reference's loop);
 *   3. publish   every hosted camera's frame block (W/8 x H/8 thumbnails of its fill-in textures | camera id | tick | pose) is offered
 *                to ITS map's key-frame database (Ferns::addFrame; not while the camera is lost, ElasticFusion.cpp:588-591) and
 *                all-gathered (the one collective);
 *   4. query     owner computes: for every reference frame hosted here and every camera of another frame, Ferns::findFrame(interMap)
 *                on the gathered thumbnails; the {closest, recoveryPose} table is all-gathered;
 *   5. decide    every rank walks the same table in the reference's order (cameras by id, frames by id); a fern match is a
 *                candidate that the matched map's owner refines at full resolution (dms_refframe_refine, ReferenceFrame.h:72-110;
 *                the querying camera's three fill-in textures travel point to point first when it lives on another rank) and
 *                accepts or rejects; {accepted, relativeTransform} and every rank's status travel in one small all-gather (a failure
 *                anywhere ends the step with an error on EVERY rank).  A camera visits the other cameras' frames in camera-id order
 *                (m_contextToReferenceFrameMap, ElasticFusion.cpp:598-599: a frame that holds several cameras is visited once per
 *                camera).  First accepted candidate of a camera wins, at most one merge per frame and tick;
 *   6. merge     same rank: dms_fusion_join_map + dms_ferns_consume.  Across ranks: surfel records, key-frame records and per
 *                camera {pose, tick, last frame, pose graph} point to point; the consuming rank appends them
 *                (dms_model_consume_records, dms_ferns_consume_records) and re-creates each camera (dms_fusion_import_camera);
 *                the sender frees its copies.  Pose graphs are re-based by relativeTransform (ReferenceFrame.h:138-141).
 * Everything that crosses ranks goes through a dms_transport: device buffers, byte counts, a stream - over RCCL
 * (dms_transport_rccl, dmslam_collab.h) in production; tests plug a transport of their own (torch.distributed's gloo through
 * ctypes callbacks) so that the same compiled protocol runs with two ranks on a one-GPU box.
 * The step synchronises where the reference does (the decision needs the query's result on the host).
 */
#ifndef DMSLAM_SESSION_H_
#define DMSLAM_SESSION_H_

#include "dmslam_collab.h"
#include "dmslam_ferns.h"
#include "dmslam_fusion.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How bytes cross ranks.  Every function moves DEVICE memory of the calling rank and returns 0 on success; a call may be
 * asynchronous on `s` (RCCL) or complete on return (a host-staged test transport) - the session synchronises `s` before it reads
 * what a call delivered.  allgather: every rank contributes `bytes`, recv_dev receives world * bytes in rank order.  broadcast:
 * `bytes` at buf_dev from rank `root` to everybody (in place; not used by the session since round 6 - the refinement's result rides an
 * all-gather with every rank's status - and may be NULL). */
typedef struct dms_transport {
  void* ctx;
  int rank, world;
  int (*allgather)(void* ctx, const void* send_dev, void* recv_dev, size_t bytes, dms_stream s);
  int (*send)(void* ctx, const void* src_dev, size_t bytes, int peer, dms_stream s);
  int (*recv)(void* ctx, void* dst_dev, size_t bytes, int peer, dms_stream s);
  int (*broadcast)(void* ctx, void* buf_dev, size_t bytes, int root, dms_stream s);
} dms_transport;
/* the RCCL transport over a communicator of dmslam_collab.h (broadcast = an all-gather of the root's bytes: 68 bytes per candidate) */
int dms_transport_rccl(dms_collab* c, dms_transport* out);

typedef struct dms_session_params {
  int n_cameras;
  dms_fusion_params camera; /* template of every camera's context; timeIdx (= camera id) and num_sensors (>= n_cameras) are set by the session */
  /* the reference frames' key-frame databases: Ferns(500, Options::depth * 1000, Options::interMapPhotoThresh) (ReferenceFrame.h:17) */
  int fern_num, fern_max_depth_mm, fern_capacity;
  float fern_photo_thresh;
  unsigned int fern_seed;   /* the same table on every rank */
  float fern_threshold;     /* Options::fernThresh 0.3095 */
  int inter_map;            /* Ferns::findFrame's interMap argument (1 = the reference's; 2: dmslam_ferns.h) */
  int query_from;           /* first tick index at which cameras query other maps (0: from the start, as the reference would) */
  int full_refine;          /* 1 (default): the second half of resolveRelativeTransformationFern decides; 0: the fern match alone (rounds 3-4) */
  float cov_thresh, icp_err_thresh, icp_count_thresh; /* Options::covThresh / icpErrThresh / icpCountThresh (Options.h:91-94) */
  /* 0 (default): the inter-map queries of a tick run after ALL cameras' frames of the tick (one publish, one query table, one walk:
   * two collectives per tick).  1: the reference's ORDER - the block sits inside ElasticFusion::processFrame (ElasticFusion.cpp:595-632),
   * so camera c queries the other maps, and a successful query merges, BEFORE camera c + 1's frame of the same tick is processed
   * (camera c + 1's key-frame database does not hold its frame of this tick yet; after a merge at camera c the following cameras of the
   * tick already track against the merged map).  dms_session_step only: the cameras of a tick are serialised across ranks (publish,
   * table and walk per camera).  Restated as oracle/orc_pipeline.Session(query_inside_frame = True). */
  int query_inside_frame;
  /* 1: dms_session_step_async brackets its per-tick all-gather with a HIP event pair on the caller's stream (dms_session_exchange_time) */
  int time_exchange;
} dms_session_params;
void dms_session_default_params(dms_session_params* p, int n_cameras, int width, int height, float fx, float fy, float cx, float cy);

typedef struct dms_session dms_session;
/* t == NULL: a one-rank session (every camera in this process, as the reference runs them) */
int dms_session_create(dms_session** out, const dms_session_params* p, const dms_transport* t);
int dms_session_destroy(dms_session* s);

/* One tick.  rgb_dev[i] / depth_dev[i]: the frame (RGB8 W x H x 3, depth u16 W x H, in this device's HBM) of the i-th camera READ on
 * this rank - the cameras c with c % world == rank, ascending.  k: the tick index (log, query_from).  Every rank calls it with the
 * same k.  Returns DMS_ERR_STATE on every rank when the query failed on any of them. */
int dms_session_step(dms_session* s, int k, const void* const* rgb_dev, const unsigned short* const* depth_dev, dms_stream st);

/* The pipelined tick: the same frames, the same publish, the same all-gather - and no host synchronisation.  Nothing is fetched; the
 * descriptor half of every inter-map query (Ferns.cpp:327-342: minimum dissimilarity over the stored key frames, then blockHDAware > 0.3
 * against the frame it chose - the test that decides whether findFrame verifies at all) runs on the device for every (hosted database,
 * gathered block) pair (dms_ferns_search_blocks_hd), and its result rows travel inside the NEXT tick's blocks, so every rank learns of
 * a hit from the same bytes.  Rule: tick k runs the reference's full inter-map block (steps 4 - 6 above, on tick k's blocks, after
 * fetching every hosted camera: a woken tick IS a synchronous tick) iff the search enqueued at tick k - 3 hit for any eligible pair
 * (camera of another frame, k - 3 >= query_from, no merge at or after tick k - 3).  3 = one tick for the rows to ride the gather, two
 * for the host to read a tick that the device has certainly finished without ever waiting for the one in flight.  Between hits the host
 * only enqueues.  Pose graphs are completed from the gathered blocks two ticks late (dms_session_sync / dms_session_pose_graph bring
 * them up to date); dms_session_last_result holds the last FETCHED frame.  Requirements: camera.reloc == 0 (a lost camera is only
 * seen by a fetch), world * n_cameras <= 64, consecutive k, and the caller's frame buffers of tick k stay unchanged until the step of
 * tick k + 2 (or dms_session_sync + a stream synchronisation) has returned.  The two steps may be mixed; a synchronous step discards
 * the hits in flight.  Restated as oracle/orc_pipeline.Session(wake_latency = 3).
 * Streams: the frames of the hosted maps run on streams of the session's own (a pool of two that the maps take in turn; cameras that
 * share a map stay serial on one stream, independent maps overlap; DMS_SESSION_MAP_STREAMS=n sets the pool size, 0 puts everything on
 * `st`), the exchange on `st`, which joins those streams once per tick: synchronising `st` after a step waits for everything the step
 * enqueued. */
int dms_session_step_async(dms_session* s, int k, const void* const* rgb_dev, const unsigned short* const* depth_dev, dms_stream st);
int dms_session_sync(dms_session* s);                                   /* host state (poses, pose graphs) up to the last enqueued tick */
int dms_session_async_stats(dms_session* s, int* ticks, int* wakes);    /* pipelined ticks so far, of which woken */
/* with params.time_exchange: the device time of the pipelined ticks' all-gathers whose completion the host has already seen (the
 * event pair of tick k is read when tick k + 2 starts, or by dms_session_sync): sum in milliseconds and how many.  Includes the wait
 * for the slowest rank - it is the time the collective occupies the exchange stream, beside the next frame. */
int dms_session_exchange_time(dms_session* s, double* allgather_ms_sum, int* allgathers);

/* state, identical on every rank */
int dms_session_frame_of(dms_session* s, int* frame_of);               /* n_cameras entries: camera -> reference frame (its founding camera) */
int dms_session_host_of_frame(dms_session* s, int frame);             /* rank, or -1 when the frame has been consumed */
int dms_session_num_merges(dms_session* s);
int dms_session_get_merge(dms_session* s, int i, int* k, int* consuming_frame, int* consumed_frame, float* relativeTransform16);
int dms_session_num_refinements(dms_session* s);
int dms_session_get_refinement(dms_session* s, int i, int* k, int* camera, int* frame, int* accepted);
/* what this rank hosts */
int dms_session_hosted(dms_session* s, int* cameras, int max, int* n);   /* ascending */
dms_fusion* dms_session_camera(dms_session* s, int camera);             /* NULL when the camera is hosted elsewhere */
dms_ferns* dms_session_ferns(dms_session* s, int frame);                /* NULL when the frame is hosted elsewhere / consumed */
int dms_session_last_result(dms_session* s, int camera, dms_frame_result* r);
/* Context::relativeCons() of a hosted camera (ElasticFusion.cpp:489-492: the rows the caller's deformation solver produced for a closed
 * loop, {src xyz, target xyz}): kept with the camera, re-based by every merge (ReferenceFrame.h:133-136) and shipped with it when it
 * migrates - the "deformation-graph constraints" that cross xGMI besides the fern descriptors. */
int dms_session_add_relative_constraint(dms_session* s, int camera, const float* src3, const float* target3);
int dms_session_relative_constraints(dms_session* s, int camera, float* rows6, int max, int* n);
/* Context::poseGraph() of a hosted camera: (tick before the frame, pose after it) per processed frame, re-based by every merge */
int dms_session_pose_graph(dms_session* s, int camera, int* ticks, float* poses16, int max, int* n);

#ifdef __cplusplus
}
#endif
#endif
