/* orc_fusion.h — interface of the fusion half of the CPU ORACLE (test infrastructure). */
#ifndef ORC_FUSION_H_
#define ORC_FUSION_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SENSORS 8 /* reference Vertex::MAX_SENSORS = 3 (Shaders/Vertex.cpp:49); widened like the product */

/* one surfel: the record of dms_model_download (12 + ORC_MAX_SENSORS floats) */
typedef struct orc_surfel {
  float pos[4];  /* x y z confidence */
  float col[4];  /* colour, 0, initTime, stamp */
  float nrm[4];  /* nx ny nz radius */
  float times[ORC_MAX_SENSORS];
} orc_surfel;

void orc_inv4f(const float* m, float* o);
void orc_depth_bilateral(const uint16_t* src, int rows, int cols, float maxD, uint16_t* dst);
void orc_depth_metric(const uint16_t* src, int rows, int cols, float maxD, float* dst);
int orc_model_initialise(const uint8_t* rgba, const float* depth_metric, const float* depth_metric_filtered, int rows, int cols,
                         float cx, float cy, float fx, float fy, int time, int timeIdx, float maxDepth, orc_surfel* out, int cap);
void orc_index_map(const orc_surfel* model, int M, const float* pose16, float cx, float cy, float fx, float fy, int rows, int cols, int time,
                   int timeIdx, float maxDepth, int timeDelta, uint32_t* index, float* vertConf, float* colorTime, float* normRad);
void orc_splat_predict(const orc_surfel* model, int M, const float* pose16, float cx, float cy, float fx, float fy, int rows, int cols,
                       float maxDepth, float confThreshold, int time, int timeIdx, int maxTime, int timeDelta, int actv, uint8_t* image,
                       float* vertex, float* normal, uint16_t* timeImg, float* depthOnly);
int orc_model_fuse(orc_surfel* model, int M, const float* pose16, int time, int timeIdx, const uint8_t* rgba, const float* dr,
                   const float* drf, const uint32_t* index, const float* vertConf, const float* normRad, int rows, int cols, float cx,
                   float cy, float fx, float fy, float maxDepth, float weighting, orc_surfel* newUnstable, int* nNew);
int orc_model_clean(const orc_surfel* model, int M, const orc_surfel* newUnstable, int nNew, const float* pose16, int time, int timeIdx,
                    const uint32_t* index, const float* vertConf, const float* colorTime, const float* depthSynth, int rows, int cols,
                    float cx, float cy, float fx, float fy, float confThreshold, const float* nodes, int nNodes, int timeDelta,
                    float maxDepth, int isFern, orc_surfel* out, int cap);
void orc_fill_in(const float* exVertex, const float* exNormal, const uint8_t* exImage, const uint16_t* depth, const uint8_t* rgba, int rows,
                 int cols, float cx, float cy, float fx, float fy, int passGeom, int passRgb, float* outVertex, float* outNormal,
                 uint8_t* outImage);
void orc_resize_nn(const void* src, int srows, int scols, void* dst, int drows, int dcols, int elem);
int orc_dense_enough(const uint8_t* image_rgba, int rows, int cols);
float orc_velocity_weight(const float* currPose16, const float* lastPose16, float weightMultiplier);
/* time slots in the clean's health test (copy_unstable.vert:137-150; reference NUM_CAMERAS = 3, the default) */
void orc_set_num_sensors(int n);

#ifdef __cplusplus
}
#endif
#endif
