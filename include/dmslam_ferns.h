/* dmslam_ferns.h — randomised-fern keyframe database and (inter-map) relocalisation, SURVEY.md 8(f1).
 *
 * Replaces `class Ferns` (elasticfusion/Core/src/Ferns.{h,cpp}): the per-map database of W/8 x H/8 keyframe
 * thumbnails encoded by 500 random 4-bit ferns, the dissimilarity search over it, and the geometric + photometric
 * verification of the best match with a thumbnail-sized RGBDOdometry.  In the reference all of it is host code over
 * glReadPixels'd thumbnails (Ferns.cpp:21-706) and its call sites are compiled out (ElasticFusion.cpp:279,589,597);
 * here the database lives in HBM, encoding and the search over all stored frames are HIP kernels, the verification
 * runs through the same tracker as the frame step, and only the accept / reject decision is taken on the host.
 *
 * Collaborative mode (one camera per GPU): every rank builds its fern table from the SAME seed, so the 500-byte code
 * vector of a frame is a descriptor all ranks can match against their own database — that descriptor (plus the
 * thumbnails the verification needs) is what travels over RCCL (densemonoslam_amd/collab.py).
 */
#ifndef DMSLAM_FERNS_H_
#define DMSLAM_FERNS_H_

#include "dmslam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dms_ferns dms_ferns;

#define DMS_FERN_BAD_CODE 255 /* Ferns::badCode (Ferns.cpp:35) */
#define DMS_FERN_MAX 512      /* ferns per table (the reference uses 500, ReferenceFrame.h:17) */

/* Ferns::Ferns(n, maxDepth, photoThresh) (Ferns.cpp:21-58).  width / height / intrinsics = the full-resolution camera
 * (the reference reads the Resolution / Intrinsics singletons); thumbnails are width/8 x height/8.  maxDepth in
 * millimetres (ReferenceFrame.h:17 passes depthCutoff * 1000).  `seed` replaces random.seed(time(0)) (Ferns.cpp:56):
 * table = mt19937(seed) drawn through uniform_int_distribution in the reference's order (x, y, r, g, b, d per fern,
 * Ferns.cpp:68-83; Lemire's nearly-divisionless mapping, as libstdc++ 11).  capacity = frames the database can hold. */
int dms_ferns_create(dms_ferns** out, int num, int maxDepth_mm, float photoThresh, int width, int height, float cx, float cy,
                     float fx, float fy, unsigned int seed, int capacity);
int dms_ferns_destroy(dms_ferns* f);
/* the conservatory: pos[num][2] = (x, y) thumbnail pixel, rgbd[num][4] = thresholds (Ferns::Fern) */
int dms_ferns_get_table(dms_ferns* f, int* pos2, int* rgbd4);
/* Without a synchronisation: the number of stored key frames and the number of key frames that were NOT stored because the
 * database had reached its capacity, as of the newest completed asynchronous add (dms_ferns_add_frame_async /
 * dms_ferns_publish_block).  The reference's database grows without bound (Ferns.cpp:235-275); a caller that sees `dropped`
 * move should have created the handle with a larger capacity. */
int dms_ferns_status(dms_ferns* f, int* stored, int* dropped);
int dms_ferns_num_frames(dms_ferns* f);
/* stored frame `id`: pose (16 floats, row-major), source time, good-code count, codes (num bytes); NULL = skip */
int dms_ferns_get_frame(dms_ferns* f, int id, float* pose16, int* srcTime, int* goodCodes, unsigned char* codes);

/* Encode a frame without touching the database: the descriptor of collaborative mode.  image = RGBA8, vertex / normal =
 * RGBA32F, full resolution, dense (the fill-in textures the reference passes, ElasticFusion.cpp:681-684).
 * codes_dev: DMS_FERN_MAX bytes in HBM (entries past num hold the bad code), good_dev: one int in HBM (NULL = skip
 * either).  Asynchronous on `s`. */
int dms_ferns_encode(dms_ferns* f, const dms_image2d* image_rgba, const dms_image2d* vertex, const dms_image2d* normal,
                     unsigned char* codes_dev, int* good_dev, dms_stream s);

/* bool Ferns::addFrame(imageTexture, vertexTexture, normalTexture, pose, srcTime, threshold) (Ferns.cpp:170-276):
 * encode, minimum dissimilarity against every stored frame, append when (minimum > threshold || empty) && goodCodes > 0.
 * Synchronises `s` (the reference returns the decision). */
int dms_ferns_add_frame(dms_ferns* f, const dms_image2d* image_rgba, const dms_image2d* vertex, const dms_image2d* normal,
                        const float* pose16, int srcTime, float threshold, int* added, dms_stream s);

/* The same without a host synchronisation (the frame loop of a pipelined caller): search, decision and commit are
 * stream ordered on `s`; the count and the per-frame metadata live on the device (dms_ferns_num_frames /
 * dms_ferns_get_frame synchronise to read them).  Frame = full-resolution textures, or a thumbnail block
 * (thumb_block_dev != NULL, textures ignored).  Pose = host matrix or, for callers whose pose never leaves HBM, a
 * device pointer to 16 floats (pose16_dev, read when the kernel runs). */
int dms_ferns_add_frame_async(dms_ferns* f, const dms_image2d* image_rgba, const dms_image2d* vertex, const dms_image2d* normal,
                              const void* thumb_block_dev, const float* pose16_host, const float* pose16_dev, int srcTime, float threshold,
                              dms_stream s);
/* descriptor of a frame that already is a thumbnail block; touches nothing of the handle's state (any stream) */
int dms_ferns_encode_thumbs(dms_ferns* f, const void* thumb_block_dev, unsigned char* codes_dev, int* good_dev, dms_stream s);

/* What a camera does with its frame block every frame in collaborative mode, in four launches: the descriptor
 * (dms_ferns_encode_thumbs: codes_dev / good_dev, inside or beside the block) and the key-frame insertion
 * (dms_ferns_add_frame_async with the pose in HBM) share one encoding pass, and the block is read where it lies instead
 * of being staged — it must stay unchanged until the work enqueued here has run. */
int dms_ferns_publish_block(dms_ferns* f, const void* thumb_block_dev, unsigned char* codes_dev, int* good_dev, const float* pose16_dev,
                            int srcTime, float threshold, dms_stream s);
/* The same; the launch also copies mirror_bytes bytes of the block from byte offset mirror_offset on - as they are AFTER the encoding
 * (codes_dev / good_dev may lie inside that range) - to `mirror`, memory the host can read (hipHostMalloc): a session whose cameras
 * all share one map has no descriptor search whose launch could carry the per-tick metadata to the host (dms_ferns_search_blocks_hd_mirror).
 * Offsets and sizes in whole dwords. */
int dms_ferns_publish_block_mirror(dms_ferns* f, const void* thumb_block_dev, unsigned char* codes_dev, int* good_dev, const float* pose16_dev,
                                   int srcTime, float threshold, void* mirror, size_t mirror_offset, size_t mirror_bytes, dms_stream s);

typedef struct dms_fern_match {
  int closest;            /* Ferns::lastClosest: accepted frame id or -1 */
  int candidate;          /* minId of the dissimilarity search (-1: none eligible) */
  float dissimilarity;    /* of the candidate */
  float blockHDAware;     /* agreement of the jointly valid codes (verification runs when > 0.3, Ferns.cpp:342) */
  float icp_error, icp_count, photo_error; /* verification outcome (Ferns.cpp:386-392) */
  float estPose[16];      /* the returned pose: identity unless the candidate was verified by the tracker */
  int n_constraints;      /* surface constraints (Ferns.cpp:396-414), rows of 8 floats {raw xyz 1 | model xyz 1} */
} dms_fern_match;

/* Eigen::Matrix4f Ferns::findFrame(constraints, currPose, vertexTexture, normalTexture, imageTexture, time, lost,
 * depthCutoff, interMap) (Ferns.cpp:277-423).  constraints: room for 8 * 64 floats (or NULL).  Synchronises `s`.
 * interMap: 0 / 1 as the reference (1: frames of any age are candidates and the verification tracker runs its pyramid with the SO3
 * pre-alignment and 50 iterations per level, RGBDOdometry.cpp:387-389); 2 (no reference counterpart, used by the collaborative
 * session where asked): candidates as for 1, verification with the intra-map settings (one level, 10 iterations) - the 3 x 50
 * point-to-plane iterations on 80 x 60 thumbnails drift along large planes. */
int dms_ferns_find_frame(dms_ferns* f, const dms_image2d* vertex, const dms_image2d* normal, const dms_image2d* image_rgba,
                         const float* currPose16, int time, int lost, int interMap, dms_fern_match* match, float* constraints,
                         dms_stream s);
/* The same query for a frame that arrives as thumbnails (collaborative mode: another camera's block, packed as
 * dms_fusion_thumbnails writes it — RGBA8 image | RGBA32F vertex | RGBA32F normal, each W/8 x H/8). */
int dms_ferns_find_frame_thumbs(dms_ferns* f, const void* thumb_block_dev, const float* currPose16, int time, int lost, int interMap,
                                dms_fern_match* match, float* constraints, dms_stream s);
/* Only the search half of the query (no verification, no host decision): best[0] = candidate id or -1,
 * best[1] = dissimilarity bits, written to HBM asynchronously.  codes_dev / good as dms_ferns_encode writes them. */
int dms_ferns_search_codes(dms_ferns* f, const unsigned char* codes_dev, const int* good_dev, int time, int interMap, int* best2_dev,
                           dms_stream s);

/* The search for a batch of descriptors in one launch: `count` blocks of `stride` bytes at blocks_dev, each holding
 * DMS_FERN_MAX code bytes at codes_offset and its good-code count (int) at good_offset — the gathered frame blocks of
 * collaborative mode; block `skip` (the caller's own, or -1) is left out.  best2_dev: count x {candidate id or -1,
 * dissimilarity bits}.  previous_out (optional, device-accessible — e.g. mapped pinned host memory): receives what
 * best2_dev held on entry, i.e. the results of the previous call, before the words are re-armed; a caller that reads
 * its results one frame late gets them on the host without a copy on the frame's stream.  With previous_out the results
 * alternate between best2_dev and a word set of the handle's (the search kernel itself hands the previous call's set over and
 * re-arms it: no launch between two searches), so best2_dev is scratch then: read the results from previous_out. */
int dms_ferns_search_blocks(dms_ferns* f, const void* blocks_dev, size_t stride, int count, int skip, size_t codes_offset,
                            size_t good_offset, int time, int interMap, int* best2_dev, int* previous_out, dms_stream s);

/* The batch search followed, in the same stream, by the test that decides whether findFrame verifies a candidate at all
 * (Ferns.cpp:340-342): hits4_dev receives count x {candidate id or -1, dissimilarity bits, codes valid in both descriptors, of those
 * equal} (16 bytes each, 16-byte aligned); blockHDAware = equal / valid, a "hit" is (double)blockHDAware > 0.3 (Ferns.cpp:346 compares the float with the double literal: 150 / 500 = 0.3f hits).  No block is skipped.  Two
 * launches (the second re-arms the handle's result words for the next call).  What the pipelined session (dms_session_step_async)
 * runs every tick in place of the synchronous query. */
int dms_ferns_search_blocks_hd(dms_ferns* f, const void* blocks_dev, size_t stride, int count, size_t codes_offset, size_t good_offset,
                               int time, int interMap, int* hits4_dev, dms_stream s);
/* The same; its second launch also copies mirror_bytes bytes from byte offset mirror_offset of every block (as they are on entry) to
 * mirror + q * mirror_bytes - memory the host can read (hipHostMalloc): the per-tick metadata of the gathered blocks reaches the host
 * without a launch of its own. */
int dms_ferns_search_blocks_hd_mirror(dms_ferns* f, const void* blocks_dev, size_t stride, int count, size_t codes_offset, size_t good_offset,
                                      int time, int interMap, int* hits4_dev, void* mirror, size_t mirror_offset, size_t mirror_bytes,
                                      dms_stream s);

/* void Ferns::consume(otherFrames, relativeTransform, threshold) (Ferns.cpp:160-168): every stored frame of `src`,
 * re-posed by relativeTransform, goes through addFrame(Frame*, threshold) of `dst` (re-encoded with dst's table).
 * added = frames accepted.  Synchronises. */
int dms_ferns_consume(dms_ferns* dst, dms_ferns* src, const float* relativeTransform16, float threshold, int* added, dms_stream s);

/* The same when the two databases live on different ranks: the stored key frames as self-contained records of
 * dms_ferns_record_bytes() bytes each ({thumbnail block | pose | srcTime}), written to / read from HBM so that they travel
 * point to point (RCCL send / recv, dms_collab_send / _recv); consume_records re-poses each frame by relativeTransform and offers it
 * to addFrame(Frame*, threshold) exactly like dms_ferns_consume.  Both handles must have been created with the same geometry. */
size_t dms_ferns_record_bytes(dms_ferns* f);
int dms_ferns_export_records(dms_ferns* f, void* records_dev, int max_count, int* count, dms_stream s);
int dms_ferns_consume_records(dms_ferns* dst, const void* records_dev, int count, const float* relativeTransform16, float threshold, int* added,
                              dms_stream s);

#ifdef __cplusplus
}
#endif
#endif
