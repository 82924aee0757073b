/* dmslam_collab.h — the exchange step of the collaborative session, over RCCL.
 *
 * The reference runs all cameras of a collaborative session in one process on one GPU and shares their state through
 * host memory (GUI/src/MainController.cpp:262-400 loops over the `ReferenceFrame`s; the compiled-out inter-map block
 * ElasticFusion.cpp:595-632 and ReferenceFrame::resolveRelativeTransformationFern, ReferenceFrame.h:34-110, read another
 * camera's fern database directly).  Here a camera and its map live on their own GPU / process, and the only data that
 * cross xGMI are
 *   - per frame: every camera's frame block (fern descriptor + W/8 x H/8 thumbnails, dms_fusion_frame_block) — an
 *     all-gather, after which dms_ferns_search_blocks matches the remote descriptors against the local database;
 *   - on a verified inter-map match: the consumed map as packed 80-byte records (dms_model_export_records ->
 *     send / recv -> dms_model_consume_records) — point to point.
 * These entry points are what a C++ front end calls for the two; they wrap an RCCL communicator (librccl.so.1, resolved at
 * run time: a single-camera process never loads it).  No torch types, device pointers + byte counts + a HIP stream; every
 * call is asynchronous on `s` and stream-ordered like the rest of the library.  densemonoslam_amd/collab.py drives the same
 * exchange over torch.distributed (the test harness and bench.py); either carrier moves the same bytes.
 *
 * Rendezvous: rank 0 calls dms_collab_unique_id and hands the 128 bytes to the other ranks by any means (MainController
 * starts its cameras from one command line: a file, an environment variable, the LCM bus); every rank then calls
 * dms_collab_create with the device it already selected (hipSetDevice).
 */
#ifndef DMSLAM_COLLAB_H_
#define DMSLAM_COLLAB_H_

#include <stddef.h>

#include "dmslam.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DMS_COLLAB_ID_BYTES 128

typedef struct dms_collab dms_collab;

/* ncclGetUniqueId; DMS_ERR_UNSUPPORTED when librccl cannot be loaded */
int dms_collab_unique_id(void* id128);
/* ncclCommInitRank on the calling thread's current device.  nranks >= 1, 0 <= rank < nranks.  Returns DMS_ERR_TIMEOUT (never hangs) when
 * the other ranks have not arrived within DMS_RCCL_INIT_TIMEOUT_S seconds (environment, default 120). */
int dms_collab_create(dms_collab** out, int rank, int nranks, const void* id128);
/* The file the RCCL entry points were resolved from ("" when none could be loaded).  A process uses ONE copy: the one that is
 * already mapped (torch's torch/lib/librccl.so under torch.distributed, a front end's own), else DMS_RCCL_PATH, else librccl.so.1. */
const char* dms_collab_library_path(void);
int dms_collab_rank(const dms_collab* c);
int dms_collab_size(const dms_collab* c);
/* every rank contributes `bytes` from send_dev; recv_dev receives nranks * bytes, rank r's block at r * bytes
 * (ncclAllGather over bytes).  The per-frame exchange: send_dev = the block dms_fusion_frame_block filled. */
int dms_collab_allgather(dms_collab* c, const void* send_dev, void* recv_dev, size_t bytes, dms_stream s);
/* point-to-point transfer of `bytes` (a packed record buffer of a map merge); the two sides must match */
int dms_collab_send(dms_collab* c, const void* src_dev, size_t bytes, int peer, dms_stream s);
int dms_collab_recv(dms_collab* c, void* dst_dev, size_t bytes, int peer, dms_stream s);
/* max over ranks of one double, in place on the device (bench.py's timing rule; ncclAllReduce) */
int dms_collab_allreduce_max_f64(dms_collab* c, double* value_dev, dms_stream s);
/* ncclCommDestroy (after the streams that carried its collectives have drained) */
int dms_collab_destroy(dms_collab* c);

#ifdef __cplusplus
}
#endif
#endif
