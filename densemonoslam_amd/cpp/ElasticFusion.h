// Source-compatible outer API of the reference's dense back end over the C ABI of libdmslam_hip.so:
//
//   class ElasticFusion  (Core/src/ElasticFusion.h:63-389)  constructor with the reference's 22 parameters (:69-82),
//                        frontend(name) (:315), processFrame with the reference's TEN parameters (:92-100), applyGlobalLoop
//                        (:110), getGlobalModel (:122), the GUI-driven setters (:168-226: setRgbOnly, setIcpWeight, setPyramid,
//                        setFastOdom, setSo3, setFrameToFrameRGB, setConfidenceThreshold, setDepthCutoff) and the end-of-run
//                        exports savePly / saveTrajectories / saveStats / saveTimes (:280-290)
//   class Context        (Core/src/Context.h:25-378)         one camera ("a SLAM frontend"): owns the dms_fusion of that camera;
//                        id(), rgbOnly(), currPose(), tick(), lost(), numFused()
//   Resolution / Intrinsics (Core/src/Utils/Resolution.h, Intrinsics.h)  the singletons the front end fills before it
//                        constructs ElasticFusion (GUI/src/MainController.cpp:39-60)
//
// so that GUI/src/MainController.cpp:203-229 (construction, frontend(), ctx.rgbOnly()) and :373-377 (the processFrame call with
// the ORB-SLAM3 pose prior and loop-closure poses) compile against this header unchanged.  The 4x4 matrix type is a template
// parameter (anything with float operator()(row, col): Eigen::Matrix4f in the reference's build); `ElasticFusion` itself is
// the Eigen instantiation when <Eigen/Core> has been included before this header, so the header also compiles without Eigen
// (tests/test_cpp_mirror.py uses a 20-line stand-in).
//
// What stays with the caller, as in SURVEY.md §8 / DESIGN.md §8: the deformation-graph optimisation
// (Deformation::addConstraint / constrain, CPU + CHOLMOD).  It plugs in through `constrain`: called with the constraint rows the
// device produced ({source xyz, target xyz, time} = the arguments of Deformation::addConstraint), it returns the node table
// `rawGraph` (16 floats per node, Deformation.cpp:192-201) — empty = no deformation — exactly where the reference calls
// rf.globalDeformation().constrain / rf.localDeformation().constrain (ElasticFusion.cpp:337, :481).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "dmslam.hpp"

// ---- the reference's sensor singletons (same accessors) -------------------------------------------------------------
class Resolution {
 public:
  static const Resolution& getInstance(int width = 0, int height = 0) {
    static const Resolution instance(width, height);
    return instance;
  }
  const int& width() const { return imgWidth; }
  const int& height() const { return imgHeight; }
  const int& cols() const { return imgWidth; }
  const int& rows() const { return imgHeight; }
  const int& numPixels() const { return imgNumPixels; }

 private:
  Resolution(int width, int height) : imgWidth(width), imgHeight(height), imgNumPixels(width * height) {
    if (width <= 0 || height <= 0) throw std::runtime_error("Resolution::getInstance(width, height) must be called first");
  }
  const int imgWidth, imgHeight, imgNumPixels;
};

class Intrinsics {
 public:
  static const Intrinsics& getInstance(float fx = 0, float fy = 0, float cx = 0, float cy = 0) {
    static const Intrinsics instance(fx, fy, cx, cy);
    return instance;
  }
  const float& fx() const { return fx_; }
  const float& fy() const { return fy_; }
  const float& cx() const { return cx_; }
  const float& cy() const { return cy_; }

 private:
  Intrinsics(float fx, float fy, float cx, float cy) : fx_(fx), fy_(fy), cx_(cx), cy_(cy) {
    if (fx == 0 || fy == 0) throw std::runtime_error("Intrinsics::getInstance(fx, fy, cx, cy) must be called first");
  }
  const float fx_, fy_, cx_, cy_;
};

namespace dms {

// the command-line options of the reference that reach this path (Core/src/Utils/Options.h: --hybrid_tracking, --hybrid_loops,
// --i <icp weight per camera>, --d, GUI "Pyramid"); the front end sets them before it constructs ElasticFusion
struct FrontEndOptions {
  bool hybrid_tracking = false;  // refine the pose prior with the dense tracker (ElasticFusion.cpp:169)
  bool hybrid_loops = false;     // accept ORB-SLAM3 loop closures (:292)
  bool pyramid = true;
  float maxDepthProcessed = 25.f;  // ElasticFusion.cpp:56
  size_t model_capacity = 0;       // 0 = the reference's MAX_VERTICES
  static FrontEndOptions& get() {
    static FrontEndOptions o;
    return o;
  }
};

// One camera: Context (Context.h:25-378).  The device side of everything the reference keeps per Context (IndexMap, two
// RGBDOdometry, FillIn, the input textures) lives in the dms_fusion it owns.
template <class Mat4>
class ContextT {
 public:
  ContextT(const int id, const int num_bins_depth, const int num_bins_img, const std::string filename = "", const bool iclnuim = false,
           const bool reloc = false)
      : m_id(id), m_bins_depth(num_bins_depth), m_bins_img(num_bins_img), m_file(filename), m_reloc(reloc) {
    (void)iclnuim;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) m_currPose(r, c) = r == c ? 1.f : 0.f;
  }
  virtual ~ContextT() {
    if (fusion) dms_fusion_destroy(fusion);
    if (rgb_dev) dms_device_free(rgb_dev);
    if (depth_dev) dms_device_free(depth_dev);
  }
  ContextT(const ContextT&) = delete;
  const int& id() const { return m_id; }
  bool& rgbOnly() { return m_rgbOnly; }
  Mat4& currPose() { return m_currPose; }
  int& tick() { return m_tick; }
  bool& lost() { return m_lost; }
  int& numFused() { return m_numFused; }
  const std::string& filename() const { return m_file; }
  const dms_frame_result& lastResult() const { return last; }
  // one (tick, pose) per processed frame and its time stamp (ElasticFusion.cpp:571-574)
  std::vector<std::pair<unsigned long long int, Mat4>>& poseGraph() { return m_poseGraph; }
  std::vector<int64_t>& poseLogTimes() { return m_poseLogTimes; }

  // Context::saveTrajectory (Context.h:117-156): <dir><log name from its last '/'>.freiburg, one 3 x 4 matrix per line
  void saveTrajectory(std::string dir) {
    const size_t slash = m_file.find_last_of("/");
    std::string fname = dir + (slash == std::string::npos ? m_file : m_file.substr(slash)) + ".freiburg";
    std::vector<float> flat(m_poseGraph.size() * 16);
    for (size_t i = 0; i < m_poseGraph.size(); ++i)
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) flat[i * 16 + r * 4 + c] = m_poseGraph[i].second(r, c);
    check(dms_trajectory_save(fname.c_str(), flat.data(), m_poseGraph.size()), "saveTrajectory");
  }
  // Context::saveStats (Context.h:106-115) writes the front end's Stats object (ORB / depth-net bookkeeping, not on this
  // path); here: the counters this camera has, one "name value" line each, in <dir><log name>.stats
  void saveStats(std::string dir) {
    const size_t slash = m_file.find_last_of("/");
    std::string fname = dir + (slash == std::string::npos ? m_file : m_file.substr(slash + 1)) + ".stats";
    FILE* fp = fopen(fname.c_str(), "w");
    if (!fp) throw std::runtime_error("saveStats: cannot open " + fname);
    fprintf(fp, "frames %zu\nfused %d\ntick %d\nlost %d\nsurfels %u\n", m_poseGraph.size(), m_numFused, m_tick, m_lost ? 1 : 0, last.surfels);
    fclose(fp);
  }

  dms_fusion* fusion = nullptr;  // created by the first processFrame (it needs ElasticFusion's parameters)
  void* rgb_dev = nullptr;
  void* depth_dev = nullptr;
  dms_frame_result last;
  int bins_depth() const { return m_bins_depth; }
  int bins_img() const { return m_bins_img; }
  bool reloc() const { return m_reloc; }

 private:
  const int m_id, m_bins_depth, m_bins_img;
  const std::string m_file;
  const bool m_reloc;
  bool m_rgbOnly = false, m_lost = false;
  int m_tick = 1, m_numFused = 0;
  Mat4 m_currPose;
  std::vector<std::pair<unsigned long long int, Mat4>> m_poseGraph;
  std::vector<int64_t> m_poseLogTimes;
};

template <class Mat4>
class ElasticFusionT {
 public:
  typedef ContextT<Mat4> Context;
  enum SamplingScheme { NID_KEYFRAMING, NONE, UNIFORM };

  // the reference's constructor (ElasticFusion.h:69-82)
  ElasticFusionT(const int timeDelta = 200, const int countThresh = 35000, const float errThresh = 5e-05, const float covThresh = 1e-05,
                 const bool closeLoops = true, const bool iclnuim = false, const bool reloc = false, const float photoThresh = 115,
                 const float confidence = 10, const float depthCut = 3, const float icpThresh = 10, const bool fastOdom = false,
                 const float fernThresh = 0.3095, const bool so3 = true, const bool frameToFrameRGB = false, const std::string fileName = "",
                 const SamplingScheme sampling_scheme = NID_KEYFRAMING, const float nid_threshold = 0.80f, const float nidDepthLambda = 0.7f,
                 const int num_bins_depth = 500, const int num_bins_img = 64, const int m_nid_pyramid_level = 0)
      : timeDelta(timeDelta), closeLoops(closeLoops), iclnuim(iclnuim), reloc(reloc), confidence(confidence), depthCut(depthCut),
        icpThresh(icpThresh), fastOdom(fastOdom), so3(so3), frameToFrameRGB(frameToFrameRGB), scheme(sampling_scheme),
        nid_threshold(nid_threshold), nidDepthLambda(nidDepthLambda), bins_depth(num_bins_depth), bins_img(num_bins_img),
        nid_level(m_nid_pyramid_level), saveFilename(fileName) {
    (void)countThresh;  // icpCountThresh / icpErrThresh / covThresh are the thresholds of the local-loop acceptance test, fixed at the
    (void)errThresh;    // values ElasticFusion.cpp:428-442 hard-codes; photoThresh / fernThresh belong to the fern database
    (void)covThresh;    // (dms::Ferns), whose call sites this fork compiles out (:279, :589, :597)
    (void)photoThresh;
    (void)fernThresh;
  }
  virtual ~ElasticFusionT() {}

  // Deformation::constrain of the caller: (constraint rows: n x 7 floats, tick, isGlobal) -> rawGraph (16 floats per node)
  std::function<std::vector<float>(const std::vector<float>&, int, bool)> constrain;

  // "a context represents a SLAM frontend" (ElasticFusion.h:309-315)
  std::shared_ptr<Context> frontend(std::string name) {
    auto it = m_contexts.find(name);
    if (it != m_contexts.end()) return it->second;
    auto c = std::make_shared<Context>((int)m_contexts.size(), bins_depth, bins_img, name, iclnuim, reloc);
    m_contexts[name] = c;
    return c;
  }
  std::map<std::string, std::shared_ptr<Context>>& contexts() { return m_contexts; }

  /**
   * Process an rgb/depth map pair — ElasticFusion::processFrame (ElasticFusion.h:92-100, ElasticFusion.cpp:99-637)
   * @param rgb unsigned char row major order (RGB8, host)
   * @param depth unsigned short z-depth in millimeters, invalid depths are 0 (host)
   * @param timestamp only used for the output poses
   * @param inPose pose prior (the ORB-SLAM3 pose); with hybrid_tracking it is refined by the dense tracker, else taken as is
   * @param orbTcwOld, orbTcwNew the loop-closure candidate of the ORB-SLAM3 front end (hybrid_loops)
   * @param cluster sub-map of the ground-truth-clusters mode; only cluster 0 exists here (a caller with several keeps one
   *        ElasticFusion per cluster)
   * @param weightMultiplier optional full frame fusion weight
   * @param bootstrap if true, use inPose as a pose guess rather than replacement (re-assigns currPose = *inPose, :188-191)
   */
  void processFrame(const std::shared_ptr<unsigned char>& rgb, const std::shared_ptr<unsigned short>& depth, const int64_t& timestamp,
                    Context& context, const Mat4* inPose = 0, const Mat4* orbTcwOld = 0, const Mat4* orbTcwNew = 0, const int cluster = 0,
                    const float weightMultiplier = 1.f, const bool bootstrap = false) {
    (void)bootstrap;
    const int tick_before = context.tick();
    if (cluster != 0) throw std::runtime_error("processFrame: only cluster 0 is implemented (one surfel store per ElasticFusion)");
    ensure(context);
    // Context::rgbOnly() is read every frame by the reference (ElasticFusion.cpp:505): keep the device side in step
    check(dms_fusion_set_option(context.fusion, DMS_OPT_RGB_ONLY, (context.rgbOnly() || rgbOnly) ? 1.0 : 0.0), "dms_fusion_set_option");
    const int W = Resolution::getInstance().width(), H = Resolution::getInstance().height();
    check(dms_memcpy_h2d(context.rgb_dev, rgb.get(), (size_t)W * H * 3, nullptr), "upload rgb");
    check(dms_memcpy_h2d(context.depth_dev, depth.get(), (size_t)W * H * 2, nullptr), "upload depth");
    float prior[16], po[16], pn[16];
    if (inPose) flat(*inPose, prior);
    const bool orb = FrontEndOptions::get().hybrid_loops && orbTcwOld && orbTcwNew;
    if (orb) {
      flat(*orbTcwOld, po);
      flat(*orbTcwNew, pn);
      check(dms_fusion_set_orb_loop(context.fusion, po, pn), "set_orb_loop");
    }
    if (orb || closeLoops) {
      // the frame step in two halves around the caller's deformation solver (ElasticFusion.cpp:337 / :481)
      check(dms_fusion_process_frame_begin(context.fusion, context.rgb_dev, 3, (const unsigned short*)context.depth_dev, inPose ? prior : nullptr,
                                           weightMultiplier, nullptr),
            "processFrame");
      std::vector<float> rawGraph;
      float newPose[16];
      bool havePose = false;
      const int cap = (W / 20) * (H / 20);
      if (orb) {
        std::vector<float> rows((size_t)cap * 7);
        int n = 0;
        check(dms_fusion_get_global_loop_constraints(context.fusion, rows.data(), cap, &n, nullptr), "global loop constraints");
        rows.resize((size_t)n * 7);
        if (constrain && n > 0) rawGraph = constrain(rows, context.tick(), true);
      }
      if (rawGraph.empty() && closeLoops) {  // `rawGraph.size() == 0` (:399)
        dms_frame_result rl;
        check(dms_fusion_fetch_loop(context.fusion, &rl, nullptr), "fetch_loop");
        if (rl.loop_ok && constrain) {
          std::vector<float> rows((size_t)cap * 7);
          int n = 0;
          check(dms_fusion_get_loop_constraints(context.fusion, rows.data(), cap, &n), "loop constraints");
          rows.resize((size_t)n * 7);
          rawGraph = constrain(rows, context.tick(), false);
          if (!rawGraph.empty()) {  // context.currPose() = estPose (:489)
            std::memcpy(newPose, rl.loop_pose, sizeof(newPose));
            havePose = true;
          }
        }
      }
      check(dms_fusion_process_frame_end(context.fusion, rawGraph.empty() ? nullptr : rawGraph.data(), (int)(rawGraph.size() / 16),
                                         havePose ? newPose : nullptr, nullptr),
            "processFrame");
    } else {
      check(dms_fusion_process_frame(context.fusion, context.rgb_dev, 3, (const unsigned short*)context.depth_dev, inPose ? prior : nullptr,
                                     weightMultiplier, nullptr),
            "processFrame");
    }
    refresh(context);
    // :571-574 (the tick the frame was processed at; the pose after tracking / loop closure)
    context.poseGraph().push_back(std::pair<unsigned long long int, Mat4>((unsigned long long int)tick_before, context.currPose()));
    context.poseLogTimes().push_back(timestamp);
  }

  // ElasticFusion::applyGlobalLoop (ElasticFusion.h:110, ElasticFusion.cpp:1148-1240)
  void applyGlobalLoop(Context& context, Mat4& orbTcwOld, Mat4& orbTcwNew) {
    ensure(context);
    float po[16], pn[16];
    flat(orbTcwOld, po);
    flat(orbTcwNew, pn);
    check(dms_fusion_apply_global_loop_begin(context.fusion, po, pn, nullptr), "applyGlobalLoop");
    const int cap = (Resolution::getInstance().width() / 20) * (Resolution::getInstance().height() / 20);
    std::vector<float> rows((size_t)cap * 7), rawGraph;
    int n = 0;
    check(dms_fusion_get_global_loop_constraints(context.fusion, rows.data(), cap, &n, nullptr), "global loop constraints");
    rows.resize((size_t)n * 7);
    if (constrain && n > 0) rawGraph = constrain(rows, context.tick(), true);
    check(dms_fusion_apply_global_loop_end(context.fusion, rawGraph.empty() ? nullptr : rawGraph.data(), (int)(rawGraph.size() / 16),
                                           rawGraph.empty() ? 0 : 1, nullptr),
          "applyGlobalLoop");
  }

  GlobalModel getGlobalModel(Context& ctx) {
    ensure(ctx);
    return GlobalModel(dms_fusion_model(ctx.fusion));
  }
  const int& getTimeDelta() const { return timeDelta; }
  const float& getConfidenceThreshold() const { return confidence; }
  const float& getMaxDepthProcessed() const { return FrontEndOptions::get().maxDepthProcessed; }

  // the GUI-driven setters (ElasticFusion.h:168-226, ElasticFusion.cpp:1023-1043; MainController.cpp:760-775 calls them every
  // frame): they take effect with the next processFrame of every camera
  void setRgbOnly(const bool& val) { rgbOnly = val; push(DMS_OPT_RGB_ONLY, val); }
  void setIcpWeight(const float& val) { icpThresh = val; push(DMS_OPT_ICP_WEIGHT, val); }
  void setPyramid(const bool& val) { FrontEndOptions::get().pyramid = val; push(DMS_OPT_PYRAMID, val); }
  void setFastOdom(const bool& val) { fastOdom = val; push(DMS_OPT_FAST_ODOM, val); }
  void setSo3(const bool& val) { so3 = val; push(DMS_OPT_SO3, val); }
  void setFrameToFrameRGB(const bool& val) { frameToFrameRGB = val; push(DMS_OPT_FRAME_TO_FRAME_RGB, val); }
  void setConfidenceThreshold(const float& val) { confidence = val; push(DMS_OPT_CONFIDENCE, val); }
  void setDepthCutoff(const float& val) { depthCut = val; push(DMS_OPT_DEPTH_CUTOFF, val); }
  void setFernThresh(const float&) {}  // the fern relocaliser's call sites are compiled out in this fork (:279, :589)

  // End-of-run exports (MainController.cpp:806-809).
  // savePly (ElasticFusion.cpp:781-885): one file per map, <dir><fileName>.<j>.<log name>.ply, j counting from 1; the
  // reference's bytes (dms_model_save_ply; `reference_normal_offset` = its stale + 18 read, see dmslam_fusion.h)
  void savePly(std::string dir, bool reference_normal_offset = false) {
    int j = 1;
    for (auto& kv : m_contexts) {
      Context& c = *kv.second;
      if (!c.fusion) continue;
      const std::string& name = c.filename();
      const size_t slash = name.find_last_of("/");
      const std::string refName = slash == std::string::npos ? name : name.substr(slash + 1);
      const std::string filename = dir + saveFilename + "." + std::to_string(j++) + "." + refName + ".ply";
      check(dms_model_save_ply(dms_fusion_model(c.fusion), filename.c_str(), confidence, reference_normal_offset ? 1 : 0, nullptr), "savePly");
    }
  }
  void saveTrajectories(std::string dir) {  // :968-974
    for (auto& kv : m_contexts) kv.second->saveTrajectory(dir);
  }
  void saveStats(std::string dir) {  // :887-891
    for (auto& kv : m_contexts) kv.second->saveStats(dir);
  }
  // saveTimes (:893-966) dumps the reference's Stopwatch singleton (host wall-clock sections of the GUI loop, the ORB front
  // end and the depth network: none of them on this path).  Here: the device-side stage times of every camera when
  // profiling was switched on (dms_fusion_set_profiling), "camera<id><stage> total_ms launches" per line, in <dir><fileName>.timings
  void saveTimes(std::string dir) {
    const std::string fname = dir + saveFilename + ".timings";
    FILE* fp = fopen(fname.c_str(), "w");
    if (!fp) throw std::runtime_error("saveTimes: cannot open " + fname);
    static const char* stages[] = {"ingest", "preprocess", "live_pyramids", "predict", "odom_init", "track", "index_map", "fuse", "clean", "host_wait"};
    for (auto& kv : m_contexts) {
      Context& c = *kv.second;
      if (!c.fusion) continue;
      for (const char* st : stages) {
        double ms = 0.0;
        int n = 0;
        if (dms_fusion_get_kernel_time(c.fusion, st, &ms, &n) == 0) fprintf(fp, "camera%d%s %.6f %d\n", c.id(), st, ms, n);
      }
    }
    fclose(fp);
  }

 private:
  static void flat(const Mat4& m, float* o) {
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) o[r * 4 + c] = m(r, c);
  }
  void push(int option, double value) {
    for (auto& kv : m_contexts)
      if (kv.second->fusion) check(dms_fusion_set_option(kv.second->fusion, option, value), "dms_fusion_set_option");
  }
  void ensure(Context& c) {
    if (c.fusion) return;
    const Resolution& res = Resolution::getInstance();
    const Intrinsics& k = Intrinsics::getInstance();
    const FrontEndOptions& o = FrontEndOptions::get();
    dms_fusion_params p;
    dms_fusion_default_params(&p, res.width(), res.height(), k.fx(), k.fy(), k.cx(), k.cy());
    p.timeDelta = timeDelta;
    p.confidence = confidence;
    p.depthCut = depthCut;
    p.icpWeight = icpThresh;
    p.fastOdom = fastOdom;
    p.so3 = so3;
    p.frameToFrameRGB = frameToFrameRGB;
    p.pyramid = o.pyramid;
    p.hybrid_tracking = o.hybrid_tracking;
    p.hybrid_loops = o.hybrid_loops;
    p.rgbOnly = c.rgbOnly() || rgbOnly;
    p.timeIdx = c.id();
    p.maxDepthProcessed = o.maxDepthProcessed;
    p.model_capacity = o.model_capacity;
    p.local_loop_closure = closeLoops;
    p.reloc = c.reloc();
    p.nid_keyframing = scheme == NID_KEYFRAMING;
    p.nid_threshold = nid_threshold;
    p.nid_depth_lambda = nidDepthLambda;
    p.nid_bins_depth = bins_depth;
    p.nid_bins_img = bins_img;
    p.nid_pyramid_level = nid_level;
    check(dms_fusion_create(&c.fusion, &p), "dms_fusion_create");
    check(dms_device_alloc(&c.rgb_dev, (size_t)res.numPixels() * 3), "dms_device_alloc");
    check(dms_device_alloc(&c.depth_dev, (size_t)res.numPixels() * 2), "dms_device_alloc");
  }
  void refresh(Context& c) {
    check(dms_fusion_fetch(c.fusion, &c.last, nullptr), "fetch");
    for (int r = 0; r < 4; ++r)
      for (int q = 0; q < 4; ++q) c.currPose()(r, q) = c.last.pose[r * 4 + q];
    c.tick() = c.last.tick;
    c.lost() = c.last.lost != 0;
    c.numFused() += c.last.fused;
  }

  const int timeDelta;
  const bool closeLoops, iclnuim, reloc;
  float confidence, depthCut, icpThresh;  // (changed by the setters below)
  bool fastOdom, so3, frameToFrameRGB;
  bool rgbOnly = false;  // ElasticFusion::setRgbOnly (:1023): every camera tracks photometrically only and fuses nothing
  const SamplingScheme scheme;
  const float nid_threshold, nidDepthLambda;
  const int bins_depth, bins_img, nid_level;
  const std::string saveFilename;
  std::map<std::string, std::shared_ptr<Context>> m_contexts;
};

}  // namespace dms

#if defined(EIGEN_CORE_H) || defined(EIGEN_CORE_MODULE_H) || defined(DMS_WITH_EIGEN) || defined(DMS_EIGEN_MATRIX4F_DECLARED)
// the reference's names (DMS_EIGEN_MATRIX4F_DECLARED: a build without Eigen that declares its own Eigen::Matrix4f stand-in)
typedef dms::ElasticFusionT<Eigen::Matrix4f> ElasticFusion;
typedef dms::ContextT<Eigen::Matrix4f> Context;
#endif
