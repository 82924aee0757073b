/* dmslam_io.h — frame ingest formats either side of the hot path.
 *
 * Replaces, for a maintainer wiring real datasets into the back end:
 *   - eflcm::Frame, the LCM message every live / logged camera publishes
 *       (logs/rgbd/eflcm/Frame.py:27-65, logs/rgbd/lcmtypes, GUI/src/Tools/LcmHandler.h);
 *   - RawLcmLogReader (GUI/src/Tools/RawLcmLogReader.h:37-85): LCM event log -> frames;
 *   - RawLogReader / .klg (logs/rgbd/RawLogReader.cpp:30,70-110).
 * Host-side, plain C ABI, no device work: the decoded depth (u16 mm) and RGB8 buffers are what
 * dms_memcpy_h2d + dms_fusion_process_frame take.  zlib (depth) is loaded at run time; JPEG colour is
 * decoded by the library's own baseline decoder (csrc/jpeg.hpp: libjpeg's default arithmetic — slow integer
 * inverse DCT, fancy upsampling, fixed-point colour conversion — byte for byte; progressive / arithmetic-coded /
 * non-YCbCr streams return DMS_ERR_UNSUPPORTED).
 */
#ifndef DMSLAM_IO_H_
#define DMSLAM_IO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef DMS_ERR_UNSUPPORTED
#define DMS_ERR_UNSUPPORTED (-7) /* e.g. a progressive JPEG, or zlib not loadable */
#endif
#define DMS_ERR_FORMAT (-8)      /* malformed message / log */
#define DMS_EOF 1                /* readers: no more frames (not an error) */

/* eflcm::Frame with `depth` / `image` as views into the encoded message (decode) or caller
 * buffers (encode).  Wire layout (big-endian): 8-byte fingerprint 9f620b14f1894896, then
 * b trackOnly, b compressed, b last, i32 depthSize, i32 imageSize, depth bytes, image bytes,
 * i64 timestamp, i32 frameNumber, u32 len(senderName)+1, senderName, NUL (Frame.py:33-41). */
typedef struct dms_frame_msg {
  int trackOnly, compressed, last;
  int32_t depthSize, imageSize;
  const unsigned char* depth;
  const unsigned char* image;
  int64_t timestamp;
  int32_t frameNumber;
  char senderName[128];
} dms_frame_msg;

size_t dms_eflcm_frame_encoded_size(const dms_frame_msg* m);
int dms_eflcm_frame_encode(const dms_frame_msg* m, void* buf, size_t cap, size_t* written);
int dms_eflcm_frame_decode(const void* data, size_t len, dms_frame_msg* out);

/* Baseline JPEG -> W*H*3 bytes in libjpeg's R, G, B scanline order: what jpeg_read_scanlines hands
 * JPEGLoader::readData (GUI/src/Tools/JPEGLoader.h:44-95) before its per-pixel R<->B exchange.  The size
 * in the stream must equal width x height. */
int dms_jpeg_decode(const void* data, size_t len, int width, int height, unsigned char* rgb_out);

/* message -> W*H u16 depth + W*H*3 RGB8, as RawLcmLogReader::getNext does (uncompress / copy,
 * optional R<->B swap) */
int dms_frame_unpack(const dms_frame_msg* m, int width, int height, int flipColors, unsigned short* depth_out,
                     unsigned char* rgb_out);

/* LCM event log (lcm::LogFile): events of {sync 0xEDA1DA01, i64 eventnum, i64 timestamp_us,
 * i32 channel length, i32 data length, channel, data}, big-endian */
typedef struct dms_lcmlog dms_lcmlog;
int dms_lcmlog_open(dms_lcmlog** out, const char* path);
/* next event; `data` stays valid until the next call.  Returns DMS_EOF at the end. */
int dms_lcmlog_next(dms_lcmlog* h, char* channel, size_t channel_cap, const void** data, size_t* len, int64_t* timestamp_us);
int dms_lcmlog_rewind(dms_lcmlog* h);
int dms_lcmlog_close(dms_lcmlog* h);

/* .klg: i32 numFrames, then per frame i64 timestamp, i32 depthSize, i32 imageSize, depth, image
 * (little-endian; raw when the sizes equal W*H*2 / W*H*3, zlib depth / JPEG colour otherwise;
 * imageSize 0 = black image) */
typedef struct dms_klg dms_klg;
int dms_klg_open(dms_klg** out, const char* path, int width, int height, int flipColors);
int dms_klg_num_frames(dms_klg* h);
int dms_klg_next(dms_klg* h, unsigned short* depth_out, unsigned char* rgb_out, int64_t* timestamp);
int dms_klg_rewind(dms_klg* h);
int dms_klg_close(dms_klg* h);

#ifdef __cplusplus
}
#endif
#endif /* DMSLAM_IO_H_ */
