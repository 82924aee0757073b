"""The C++ mirror of the reference classes (densemonoslam_amd/cpp/dmslam.hpp) must compile with a
plain host compiler and link against the C ABI only (no HIP / torch headers)."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cstring>
#include "densemonoslam_amd/cpp/dmslam.hpp"
int main() {
  std::printf("%s\n", dms_version());
  // argument validation works without a GPU and reports through the error string
  int rc = dms_createNMap(nullptr, nullptr, nullptr);
  if (rc != DMS_ERR_INVALID_ARG || !std::strstr(dms_last_error(), "null")) return 2;
  try { dms::check(rc, "createNMap"); return 3; } catch (const std::runtime_error&) {}
  // the fern / index-map mirrors are part of the header: their entry points must resolve at link time
  void* fns[] = {(void*)&dms_ferns_create, (void*)&dms_ferns_find_frame, (void*)&dms_index_map, (void*)&dms_splat_depth};
  for (void* f : fns) if (!f) return 5;
  if (sizeof(dms::Ferns) == 0 || sizeof(dms::IndexMap) == 0) return 6;
  dms_fusion_params p;
  dms_fusion_default_params(&p, 640, 480, 528.f, 528.f, 320.f, 240.f);
  return (p.timeDelta == 200 && p.confidence == 10.f && p.maxDepthProcessed == 25.f) ? 0 : 4;
}
"""


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpp_mirror_compiles_links_and_runs():
    lib_dir = os.path.join(ROOT, "densemonoslam_amd")
    assert os.path.exists(os.path.join(lib_dir, "libdmslam_hip.so"))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "mirror.cpp")
        exe = os.path.join(td, "mirror")
        with open(src, "w") as f:
            f.write(SRC)
        subprocess.check_call(["g++", "-std=c++14", "-I" + ROOT, src, "-o", exe, "-L" + lib_dir, "-ldmslam_hip",
                               "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
        assert "gfx950" in out.stdout


GPU_SRC = r"""
// A C++ host written against the mirror of the reference classes only: two frames of a synthetic plane
// through ElasticFusion::processFrame, the way MainController drives the reference.
#include <cmath>
#include <cstdio>
#include <vector>
#include "densemonoslam_amd/cpp/dmslam.hpp"
int main() {
  const int W = 320, H = 240;
  int ndev = 0;
  if (dms_device_count(&ndev) != DMS_OK || ndev < 1) return 10;
  dms::ElasticFusion ef(W, H, 264.f, 264.f, 160.f, 120.f);
  std::vector<unsigned char> rgb((size_t)W * H * 3);
  std::vector<unsigned short> depth((size_t)W * H);
  void *rgb_dev = nullptr, *depth_dev = nullptr;
  dms::check(dms_device_alloc(&rgb_dev, rgb.size()), "alloc");
  dms::check(dms_device_alloc(&depth_dev, depth.size() * 2), "alloc");
  dms_frame_result r;
  for (int k = 0; k < 3; ++k) {
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        depth[i] = (unsigned short)(1500 + 2 * k + (x / 4) % 7 + 200.0 * std::sin(0.02 * x) * std::cos(0.03 * y));  // a wavy wall, mm
        rgb[3 * i + 0] = (unsigned char)(128 + 100 * std::sin(0.11 * (x + k)));
        rgb[3 * i + 1] = (unsigned char)(128 + 100 * std::sin(0.07 * y));
        rgb[3 * i + 2] = (unsigned char)(128 + 60 * std::sin(0.05 * (x + y)));
      }
    dms::check(dms_memcpy_h2d(rgb_dev, rgb.data(), rgb.size(), nullptr), "h2d");
    dms::check(dms_memcpy_h2d(depth_dev, depth.data(), depth.size() * 2, nullptr), "h2d");
    ef.processFrame((const unsigned char*)rgb_dev, (const unsigned short*)depth_dev);
    r = ef.fetch();
    std::printf("frame %d: tick %d surfels %u fused %d icp %.0f\n", k, r.tick, r.surfels, r.fused, r.track.lastICPCount);
  }
  dms_device_free(rgb_dev);
  dms_device_free(depth_dev);
  if (r.tick != 4 || r.surfels < 50000 || !r.fused) return 11;
  if (!(r.track.lastICPCount > 30000) || r.track.iterations_run[0] != 10) return 12;
  if (!(std::fabs(r.pose[3]) < 0.05f && std::fabs(r.pose[11]) < 0.05f)) return 13;  // a static camera stays put
  auto gm = ef.getGlobalModel();
  if (gm.lastCount() != r.surfels) return 14;
  return 0;
}
"""


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpp_host_runs_the_frame_step():
    lib_dir = os.path.join(ROOT, "densemonoslam_amd")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "host.cpp")
        exe = os.path.join(td, "host")
        with open(src, "w") as f:
            f.write(GPU_SRC)
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-I" + ROOT, src, "-o", exe, "-L" + lib_dir, "-ldmslam_hip",
                               "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)


ADAPTER_SRC = r"""
// The reference's OUTER API (Core/src/ElasticFusion.h:63-134, Context.h) as GUI/src/MainController.cpp:39-60, :203-229 and :373-377
// use it, against the adapter header — with a stand-in for Eigen::Matrix4f (this image has no Eigen).
#include <cmath>
#include <cstdio>
#include <vector>
namespace Eigen {
struct Matrix4f {
  float m[16];
  float& operator()(int r, int c) { return m[r * 4 + c]; }
  const float& operator()(int r, int c) const { return m[r * 4 + c]; }
  void setIdentity() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f; }
};
}  // namespace Eigen
#define DMS_EIGEN_MATRIX4F_DECLARED 1
#include "densemonoslam_amd/cpp/ElasticFusion.h"

int main(int argc, char** argv) {
  const int W = 320, H = 240;
  Resolution::getInstance(W, H);
  Intrinsics::getInstance(264.f, 264.f, 160.f, 120.f);
  dms::FrontEndOptions::get().hybrid_tracking = true;
  dms::FrontEndOptions::get().hybrid_loops = true;
  // MainController.cpp:203-214
  ElasticFusion* eFusion = new ElasticFusion(200, 35000, 5e-05, 1e-05, /*closeLoops*/ false, false, false, 115, /*confidence: the map is four frames old*/ 1, 3, 10, false, 0.3095, true,
                                             false, "model", ElasticFusion::SamplingScheme::NONE, 0.8f, 0.7f, 500, 64, 0);
  Context& ctx = *(eFusion->frontend("logs/camera0.klg"));
  ctx.rgbOnly() = false;
  // the GUI-driven setters, as MainController.cpp:760-775 calls them every frame
  eFusion->setRgbOnly(false);
  eFusion->setPyramid(true);
  eFusion->setFastOdom(false);
  eFusion->setConfidenceThreshold(1.f);
  eFusion->setDepthCutoff(3.f);
  eFusion->setIcpWeight(10.f);
  eFusion->setSo3(true);
  eFusion->setFrameToFrameRGB(false);
  if (argc < 2) {  // CPU build check: everything above compiled and linked; nothing touches the device
    delete eFusion;
    return ctx.id() == 0 ? 0 : 2;
  }
  int constrain_calls = 0, constraint_rows = 0;
  eFusion->constrain = [&](const std::vector<float>& rows, int tick, bool isGlobal) {
    constrain_calls += isGlobal ? 1 : 100;
    constraint_rows += (int)(rows.size() / 7);
    (void)tick;
    return std::vector<float>();  // Deformation::constrain returned false: no deformation
  };
  std::shared_ptr<unsigned char> rgb(new unsigned char[(size_t)W * H * 3], std::default_delete<unsigned char[]>());
  std::shared_ptr<unsigned short> depth(new unsigned short[(size_t)W * H], std::default_delete<unsigned short[]>());
  Eigen::Matrix4f first;
  first.setIdentity();
  for (int k = 0; k < 4; ++k) {
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        depth.get()[i] = (unsigned short)(1500 + 2 * k + (x / 4) % 7 + 200.0 * std::sin(0.02 * x) * std::cos(0.03 * y));
        rgb.get()[3 * i + 0] = (unsigned char)(128 + 100 * std::sin(0.11 * (x + k)));
        rgb.get()[3 * i + 1] = (unsigned char)(128 + 100 * std::sin(0.07 * y));
        rgb.get()[3 * i + 2] = (unsigned char)(128 + 60 * std::sin(0.05 * (x + y)));
      }
    Eigen::Matrix4f* currentPose = nullptr;
    Eigen::Matrix4f *orb_lc_Tcw_old = nullptr, *orb_lc_Tcw_new = nullptr;
    if (k > 0) {  // the ORB-SLAM3 pose prior (MainController.cpp:338-356): here the previous pose
      currentPose = new Eigen::Matrix4f(ctx.currPose());
    }
    if (k == 3) {  // a loop-closure candidate of the front end (:358-367)
      orb_lc_Tcw_old = new Eigen::Matrix4f(first);
      orb_lc_Tcw_new = new Eigen::Matrix4f(ctx.currPose());
    }
    const int64_t timestamp = 1000 * k;
    const int currentCluster = 0;
    const float weightMultiplier = 1.f;
    // MainController.cpp:373-377, verbatim
    eFusion->processFrame(rgb, depth, timestamp, ctx, currentPose, orb_lc_Tcw_old, orb_lc_Tcw_new, currentCluster, weightMultiplier, false);
    delete currentPose;
    delete orb_lc_Tcw_old;
    delete orb_lc_Tcw_new;
    std::printf("frame %d: tick %d surfels %u fused %d\n", k, ctx.tick(), ctx.lastResult().surfels, ctx.lastResult().fused);
  }
  if (ctx.tick() != 5 || ctx.lastResult().surfels < 50000 || ctx.numFused() != 4) return 11;
  if (!(std::fabs(ctx.currPose()(0, 3)) < 0.05f && std::fabs(ctx.currPose()(2, 3)) < 0.05f)) return 13;
  if (constrain_calls != 1 || constraint_rows < 50) return 14;  // the ORB loop closure reached the caller's solver once, with its rows
  Eigen::Matrix4f a = first, b = ctx.currPose();
  eFusion->applyGlobalLoop(ctx, a, b);
  auto gm = eFusion->getGlobalModel(ctx);
  if (gm.lastCount() == 0) return 15;
  // a run-time switch reaches the device: one pyramid level, three iterations (BASELINE config 2's tracker shape)
  eFusion->setPyramid(false);
  eFusion->setFastOdom(true);
  {
    Eigen::Matrix4f* prior = new Eigen::Matrix4f(ctx.currPose());
    eFusion->processFrame(rgb, depth, 4000, ctx, prior, nullptr, nullptr, 0, 1.f, false);
    delete prior;
  }
  if (ctx.lastResult().track.iterations_run[0] != 3 || ctx.lastResult().track.iterations_run[1] != 0) return 16;
  if (ctx.poseGraph().size() != 5 || ctx.poseLogTimes().back() != 4000 || ctx.poseGraph().back().first != 5) return 17;
  // MainController.cpp:806-809, verbatim but for the directory
  const std::string outDirectory = argv[2];
  eFusion->savePly(outDirectory);
  eFusion->saveTrajectories(outDirectory);
  eFusion->saveTimes(outDirectory);
  eFusion->saveStats(outDirectory);
  std::printf("surfels %u\n", ctx.lastResult().surfels);
  // the ground-truth-clusters mode (MainController.cpp:373-377 passes the frame's cluster; :515 walks globalModel().clusters()):
  // a frame that fuses under a new id starts buffers of its own from the feedback buffers, here refreshed the way the GUI's
  // raw-cloud view does (:476)
  {
    auto& rf = eFusion->whichReferenceFrame(ctx);
    if (rf.globalModel().clusters().size() != 1 || !rf.globalModel().isCluster(0) || rf.globalModel().isCluster(2)) return 18;
    const unsigned before = rf.globalModel().lastCount();
    ctx.computeFeedbackBuffers(eFusion->getMaxDepthProcessed());
    Eigen::Matrix4f* prior = new Eigen::Matrix4f(ctx.currPose());
    eFusion->processFrame(rgb, depth, 5000, ctx, prior, nullptr, nullptr, 2, 1.f, false);
    delete prior;
    int listed = 0;
    for (const auto& c : rf.globalModel().clusters()) listed += c == 0 || c == 2;
    if (listed != 2 || !rf.globalModel().isCluster(2)) return 19;
    const unsigned now = rf.globalModel().lastCount();  // the current cluster
    std::printf("clusters: %u surfels in cluster 0, %u in cluster 2\n", before, now);
    if (now == 0) return 20;
  }
  delete eFusion;
  return 0;
}
"""


def _build_adapter(td):
    lib_dir = os.path.join(ROOT, "densemonoslam_amd")
    src = os.path.join(td, "adapter.cpp")
    exe = os.path.join(td, "adapter")
    with open(src, "w") as f:
        f.write(ADAPTER_SRC)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I" + ROOT, src, "-o", exe, "-L" + lib_dir, "-ldmslam_hip", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_reference_outer_api_call_sites_compile_and_link():
    """ElasticFusion's 22-parameter constructor, frontend(name), Context::rgbOnly() and the TEN-parameter processFrame call of
    GUI/src/MainController.cpp:373-377 compile and link against densemonoslam_amd/cpp/ElasticFusion.h with a plain host compiler."""
    with tempfile.TemporaryDirectory() as td:
        exe = _build_adapter(td)
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_reference_outer_api_runs_frames_and_an_orb_loop_closure():
    with tempfile.TemporaryDirectory() as td:
        exe = _build_adapter(td)
        outdir = os.path.join(td, "out") + "/"
        os.makedirs(outdir)
        out = subprocess.run([exe, "run", outdir], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
        # the reference's file names (ElasticFusion.cpp:784-787, Context.h:106-121) and shapes
        files = sorted(os.listdir(outdir))
        assert files == ["camera0.klg.freiburg", "camera0.klg.stats", "model.1.camera0.klg.ply", "model.timings"], files
        ply = open(os.path.join(outdir, "model.1.camera0.klg.ply"), "rb").read()
        head, body = ply.split(b"end_header\n", 1)
        n = int(head.split(b"element vertex ")[1].split(b"\n")[0])
        surfels = int(out.stdout.strip().splitlines()[-1].split()[1])
        assert 0 < n <= surfels and len(body) == 31 * n
        traj = open(os.path.join(outdir, "camera0.klg.freiburg")).read().splitlines()
        assert len(traj) == 5 and all(len(line.split()) == 12 and line.endswith(" ") for line in traj)
        assert traj[0].split()[:4] == ["1", "0", "0", "0"]


RUNLOOP_SRC = r"""
// A headless host shaped like MainController::run (GUI/src/MainController.cpp:246-400) - several cameras served in turn, the session
// clock, frame skipping, the pose prior, a paused camera that only predicts - plus what the GUI half of that loop reads every frame
// (:406-415, :470, :736-781) and the merge the compiled-out inter-map block performs (ElasticFusion.cpp:610-627), written against
// densemonoslam_amd/cpp/ElasticFusion.h.  Every eFusion-> call of :248-400 appears here with the reference's argument list.
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>
namespace Eigen {
struct Matrix4f {
  float m[16];
  float& operator()(int r, int c) { return m[r * 4 + c]; }
  const float& operator()(int r, int c) const { return m[r * 4 + c]; }
  void setIdentity() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f; }
};
}  // namespace Eigen
#define DMS_EIGEN_MATRIX4F_DECLARED 1
#include "densemonoslam_amd/cpp/ElasticFusion.h"

// stands in for GUI/src/Tools/LogReader.h: a synthetic wavy wall seen by a camera that is `shift` pixels further right
struct LogReader {
  int W, H, shift, currentFrame = 0, frames;
  int64_t timestamp = 0;
  std::shared_ptr<unsigned char> rgb_;
  std::shared_ptr<unsigned short> depth_;
  LogReader(int W, int H, int shift, int frames) : W(W), H(H), shift(shift), frames(frames),
      rgb_(new unsigned char[(size_t)W * H * 3], std::default_delete<unsigned char[]>()),
      depth_(new unsigned short[(size_t)W * H], std::default_delete<unsigned short[]>()) {}
  bool hasMore() { return currentFrame < frames; }
  void getNext() {
    const int k = currentFrame++;
    timestamp = 1000 * k;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        const int xs = x + shift;
        depth_.get()[i] = (unsigned short)(1500 + (xs / 4) % 7 + 200.0 * std::sin(0.02 * xs) * std::cos(0.03 * y));
        rgb_.get()[3 * i + 0] = (unsigned char)(128 + 100 * std::sin(0.11 * xs));
        rgb_.get()[3 * i + 1] = (unsigned char)(128 + 100 * std::sin(0.07 * y));
        rgb_.get()[3 * i + 2] = (unsigned char)(128 + 60 * std::sin(0.05 * (xs + y)));
      }
  }
  void fastForward(int frame) { currentFrame = frame; }
  std::shared_ptr<unsigned char> rgb() { return rgb_; }
  std::shared_ptr<unsigned short> depth() { return depth_; }
};

int main(int argc, char** argv) {
  const int W = 320, H = 240;
  Resolution::getInstance(W, H);
  Intrinsics::getInstance(264.f, 264.f, 160.f, 120.f);
  dms::FrontEndOptions::get().hybrid_tracking = true;
  ElasticFusion* eFusion = new ElasticFusion(200, 35000, 5e-05, 1e-05, /*closeLoops*/ false, false, false, 115, 1, 3, 10, false, 0.3095, true, false, "model",
                                             ElasticFusion::SamplingScheme::NID_KEYFRAMING, 0.8f, 0.7f, 500, 64, 0);
  std::map<std::string, std::shared_ptr<LogReader>> logReaders;
  logReaders["logs/a.klg"] = std::make_shared<LogReader>(W, H, 0, 8);
  logReaders["logs/b.klg"] = std::make_shared<LogReader>(W, H, 6, 8);
  const bool run = argc > 1;
  const int start = 1, end = run ? 7 : 1;
  int framesToSkip = 0;
  bool paused_b = false;
  Context& activeCtx = *(eFusion->frontend("logs/a.klg"));
  if (eFusion->referenceFrames().size() != 1) return 2;
  int processed = 0, predicted = 0;
  // ---- the loop of :254-400 ----
  while (!(eFusion->getTick() == end)) {
    for (auto lr : logReaders) {
      std::shared_ptr<LogReader> logReader = lr.second;
      std::shared_ptr<Context> ctx = eFusion->frontend(lr.first);
      if (!(paused_b && lr.first == "logs/b.klg")) {
        if (logReader->hasMore() && eFusion->getTick() < end) {
          logReader->getNext();
          if (eFusion->getTick() < start) {
            eFusion->setTick(start);
            logReader->fastForward(start);
          }
          float weightMultiplier = framesToSkip + 1;
          if (framesToSkip > 0) {
            eFusion->setTick(activeCtx.tick() + framesToSkip);
            logReader->fastForward(logReader->currentFrame + framesToSkip);
            framesToSkip = 0;
          }
          Eigen::Matrix4f* currentPose = 0;
          Eigen::Matrix4f* orb_lc_Tcw_old = 0;
          Eigen::Matrix4f* orb_lc_Tcw_new = 0;
          int currentCluster = 0;
          if (ctx->tick() > 1) {  // the pose prior of the front end (:338-356): here the previous pose
            currentPose = new Eigen::Matrix4f;
            currentPose->setIdentity();
            *currentPose = ctx->currPose();
          }
          std::shared_ptr<unsigned short> depth_frame = logReader->depth();
          eFusion->processFrame(logReader->rgb(), depth_frame, logReader->timestamp, *ctx, currentPose, orb_lc_Tcw_old, orb_lc_Tcw_new, currentCluster,
                                weightMultiplier, false);
          if (currentPose) delete currentPose;
          processed++;
        }
      } else {
        eFusion->predict(*ctx, eFusion->whichReferenceFrame(*ctx));
        predicted++;
      }
    }
    if (!run) break;
    // the session clock: the GUI loop's exit test reads it, nothing in the back end advances it (ElasticFusion.cpp:35, :1050-1054)
    eFusion->setTick(eFusion->getTick() + 1);
    // ---- what the GUI half reads / writes every frame ----
    Context& active = *(eFusion->frontend("logs/a.klg"));
    const size_t numMaps = eFusion->referenceFrames().size();
    for (auto& rf : eFusion->referenceFrames())
      if (rf->contexts().empty() || rf->name.empty()) return 3;
    const float score = eFusion->lastKFScore(active), thr = eFusion->kFThreshold();
    if (!(score >= 0.f) || thr != eFusion->nidThreshold()) return 4;
    if (eFusion->getGlobalModel(active).lastCount() == 0 || eFusion->surfelCount() <= 0) return 5;
    (void)eFusion->whichReferenceFrame(active).globalDeformation().getGraph().size();
    (void)eFusion->getLocalDeformation(active).getGraph().size();
    (void)eFusion->getDeforms();
    (void)eFusion->getFernDeforms();
    (void)eFusion->numFused(active);
    eFusion->setRgbOnly(false);
    eFusion->setPyramid(true);
    eFusion->setFastOdom(false);
    eFusion->setConfidenceThreshold(1.f);
    eFusion->setDepthCutoff(3.f);
    eFusion->setIcpWeight(10.f);
    eFusion->setSo3(true);
    eFusion->setFrameToFrameRGB(false);
    eFusion->nidThreshold() = 0.0f;   // every frame is a key frame: the fusion half runs
    eFusion->nidDepthLambda() = 0.7f;
    eFusion->setNumBinsImg(32);
    eFusion->setNumBinsDepth(250);
    eFusion->nidPyramidLevel() = 1;
    if (eFusion->getTick() == 2) framesToSkip = 1;                        // the frame-skip branch (:290-294)
    if (eFusion->getTick() == 3 && numMaps == 2) {
      // the inter-map block of ElasticFusion::processFrame (ElasticFusion.cpp:596-608), call for call: camera b queries the other
      // reference frame with its fill-in textures.  The frame's key-frame database is given camera a's current frame first
      // (processFerns, compiled out at :589), so the query has a candidate and reaches the full-resolution refinement.
      Context& context = *(eFusion->frontend("logs/b.klg"));
      auto& other = eFusion->whichReferenceFrame(active);
      float poseA[16];
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) poseA[r * 4 + c] = active.currPose()(r, c);
      other.ferns().addFrame(&active.fillIn().imageTexture, &active.fillIn().vertexTexture, &active.fillIn().normalTexture, poseA, active.tick(),
                             dms::FrontEndOptions::get().fernThresh);
      std::vector<float> constraints;
      Eigen::Matrix4f relativeTransform;
      const float maxDepthProcessed = eFusion->getMaxDepthProcessed(), confidenceThreshold = eFusion->getConfidenceThreshold();
      const int timeDelta = eFusion->getTimeDelta();
      bool success = other.resolveRelativeTransformationFern(constraints, relativeTransform, context.currPose(), context.fillIn().vertexTexture,
                                                             context.fillIn().normalTexture, context.fillIn().imageTexture, context.tick(),
                                                             context.lost(), maxDepthProcessed, confidenceThreshold, context.id(), timeDelta,
                                                             context.tick());
      std::printf("inter-map query: closest %d success %d icp error %.3g count %.0f iterations %d %d %d\n", other.ferns().lastClosest, (int)success,
                  other.lastInterMap.lastICPError, other.lastInterMap.lastICPCount, other.lastInterMap.iterations_run[0],
                  other.lastInterMap.iterations_run[1], other.lastInterMap.iterations_run[2]);
      if (other.ferns().lastClosest != -1 && other.lastInterMap.iterations_run[0] != 50) return 21;  // (interMap = true: 50 per level)
      if (other.ferns().lastClosest == -1 && success) return 22;
    }
    if (eFusion->getTick() >= 4 && numMaps == 2) {
      // a verified inter-map match (ElasticFusion.cpp:610-627): a's frame consumes b's; the two cameras see the same wall 6 px apart
      Context& b = *(eFusion->frontend("logs/b.klg"));
      Eigen::Matrix4f T;
      T.setIdentity();
      T(0, 3) = 6.f * 1.5f / 264.f;  // b's camera frame expressed in a's: shift * z / fx
      const unsigned before = eFusion->getGlobalModel(active).lastCount();
      eFusion->mergeReferenceFrames(eFusion->whichReferenceFrame(active), b, T);
      if (eFusion->referenceFrames().size() != 1 || &eFusion->whichReferenceFrame(b) != &eFusion->whichReferenceFrame(active)) return 6;
      if (eFusion->getGlobalModel(active).lastCount() <= before) return 7;
      if (!(std::fabs(b.currPose()(0, 3) - T(0, 3)) < 0.02f)) return 8;
      (void)eFusion->getFerns(active).frames();  // the merged frame's key-frame database
    }
    if (eFusion->getTick() == 6) paused_b = true;                         // a paused camera only predicts (:397-399)
    std::printf("tick %d: maps %zu surfels %d a.tick %d b.tick %d score %.3f\n", eFusion->getTick(), eFusion->referenceFrames().size(), eFusion->surfelCount(),
                active.tick(), eFusion->frontend("logs/b.klg")->tick(), score);
  }
  if (run) {
    Context& a = *(eFusion->frontend("logs/a.klg"));
    Context& b = *(eFusion->frontend("logs/b.klg"));
    if (predicted != 1 || processed < 9) return 9;
    if (eFusion->contexts().size() != 2 || eFusion->referenceFrames().size() != 1) return 10;
    // after the merge both cameras kept tracking and fusing into ONE map
    if (a.numFused() < 5 || b.numFused() < 4 || !(b.lastResult().track.lastICPCount > 20000)) return 11;
    if (!(std::fabs(b.currPose()(0, 3) - 6.f * 1.5f / 264.f) < 0.03f)) return 12;
    std::printf("ok: %d frames, %d surfels in one map\n", processed, eFusion->surfelCount());
  }
  delete eFusion;
  return 0;
}
"""


def _build_runloop(td):
    lib_dir = os.path.join(ROOT, "densemonoslam_amd")
    src, exe = os.path.join(td, "runloop.cpp"), os.path.join(td, "runloop")
    with open(src, "w") as f:
        f.write(RUNLOOP_SRC)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I" + ROOT, src, "-o", exe, "-L" + lib_dir, "-ldmslam_hip", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_reference_run_loop_call_sites_compile_and_link():
    """Every eFusion-> call of MainController::run's headless part (GUI/src/MainController.cpp:248-400: frontend, getTick, setTick,
    processFrame, predict(ctx, whichReferenceFrame(ctx))) and of the per-frame GUI half (referenceFrames, lastKFScore, kFThreshold,
    getFerns, getLocalDeformation, nidThreshold(), setNumBinsImg / Depth, ...) compiles and links against the adapter header."""
    with tempfile.TemporaryDirectory() as td:
        exe = _build_runloop(td)
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_reference_run_loop_runs_two_cameras_a_merge_and_a_paused_camera():
    with tempfile.TemporaryDirectory() as td:
        exe = _build_runloop(td)
        out = subprocess.run([exe, "run"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
        assert "ok:" in out.stdout


SESSION_SRC = r"""
// A C++ host written against dms::Session only: MainController::run's camera loop for one rank (here: every camera in this
// process), through the pipelined tick.  argv[1] = a file of frames {rgb W*H*3, depth W*H*2} per tick and camera.
#include <cstdio>
#include <cstring>
#include <vector>
#include "densemonoslam_amd/cpp/Session.h"
int main(int argc, char** argv) {
  const int W = 320, H = 240, CAMS = 2;
  dms_session_params p = dms::Session::defaults(CAMS, W, H, 264.f, 264.f, 160.f, 120.f);
  if (p.n_cameras != CAMS || p.inter_map != 1 || p.full_refine != 1 || p.camera.width != W) return 2;
  void* fns[] = {(void*)&dms_session_step, (void*)&dms_session_step_async, (void*)&dms_session_sync, (void*)&dms_transport_rccl};
  for (void* f : fns) if (!f) return 3;
  if (argc < 4) return 0;  // (compile / link check)
  const int ticks = std::atoi(argv[2]);
  p.query_from = std::atoi(argv[3]);
  p.camera.model_capacity = 2000000;
  p.camera.num_sensors = 3;
  int ndev = 0;
  if (dms_device_count(&ndev) != DMS_OK || ndev < 1) return 10;
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 11;
  const size_t N = (size_t)W * H;
  std::vector<std::vector<void*>> rgb(ticks, std::vector<void*>(CAMS)), dep(ticks, std::vector<void*>(CAMS));
  std::vector<unsigned char> host(N * 3);
  for (int k = 0; k < ticks; ++k)
    for (int c = 0; c < CAMS; ++c) {  // frames resident in HBM before the loop starts
      dms::check(dms_device_alloc(&rgb[k][c], N * 3), "alloc");
      dms::check(dms_device_alloc(&dep[k][c], N * 2), "alloc");
      if (std::fread(host.data(), 1, N * 3, f) != N * 3) return 12;
      dms::check(dms_memcpy_h2d(rgb[k][c], host.data(), N * 3, nullptr), "h2d");
      if (std::fread(host.data(), 1, N * 2, f) != N * 2) return 12;
      dms::check(dms_memcpy_h2d(dep[k][c], host.data(), N * 2, nullptr), "h2d");
    }
  std::fclose(f);
  dms_stream st = nullptr;
  dms::check(dms_stream_create(&st), "stream");
  {
    dms::Session ses(p);
    const float a[3] = {0.f, 0.25f, 0.5f}, b[3] = {0.75f, 1.f, 1.25f};
    ses.addRelativeConstraint(1, a, b);
    for (int k = 0; k < ticks; ++k) {
      std::vector<const void*> r = {rgb[k][0], rgb[k][1]};
      std::vector<const unsigned short*> d = {(const unsigned short*)dep[k][0], (const unsigned short*)dep[k][1]};
      ses.stepPipelined(k, r, d, st);
    }
    ses.sync();
    int t = 0, w = 0;
    ses.pipelinedStats(&t, &w);
    std::printf("stats %d %d\n", t, w);
    for (const auto& m : ses.merges()) {
      std::printf("merge %d %d %d", m.tick, m.consumingFrame, m.consumedFrame);
      for (float v : m.relativeTransform) {
        unsigned u;
        std::memcpy(&u, &v, 4);
        std::printf(" %08x", u);
      }
      std::printf("\n");
    }
    for (const auto& r : ses.refinements()) std::printf("refinement %d %d %d %d\n", r.tick, r.camera, r.frame, (int)r.accepted);
    const std::vector<int> fo = ses.frameOf();
    std::printf("frame_of %d %d hosted %zu pg %zu %zu cons %zu\n", fo[0], fo[1], ses.hosted().size(), ses.poseGraph(0).size(), ses.poseGraph(1).size(),
                ses.relativeCons(1).size());
    const auto pg = ses.poseGraph(1);
    std::printf("last_pose");
    for (float v : pg.back().second) {
      unsigned u;
      std::memcpy(&u, &v, 4);
      std::printf(" %08x", u);
    }
    std::printf("\n");
  }
  dms_stream_destroy(st);
  std::printf("ok\n");
  return 0;
}
"""


def _build_session(td):
    lib_dir = os.path.join(ROOT, "densemonoslam_amd")
    src, exe = os.path.join(td, "session.cpp"), os.path.join(td, "session")
    with open(src, "w") as f:
        f.write(SESSION_SRC)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I" + ROOT, src, "-o", exe, "-L" + lib_dir, "-ldmslam_hip", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_session_mirror_compiles_and_links():
    """densemonoslam_amd/cpp/Session.h (MainController::run's camera loop for one rank of a node) with a plain host compiler"""
    with tempfile.TemporaryDirectory() as td:
        out = subprocess.run([_build_session(td)], capture_output=True, text=True)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpp_host_runs_a_pipelined_session_to_the_oracles_merge(orc):
    """A C++ host against dms::Session: two cameras, the pipelined tick, frames resident - the wake, the merge, the transform's bits and
    camera 1's last pose are the oracle session's on the same schedule."""
    import numpy as np

    from densemonoslam_amd import synth
    from tests.test_session_cpu import SCENARIOS, H, K, W, run_oracle_session

    sc = SCENARIOS["reference_rule"]
    ticks = sc.query_from + 6
    ref = run_oracle_session("reference_rule", ticks, relative_cons=False, wake_latency=3)
    assert (W, H) == (320, 240) and tuple(K) == (264.0, 264.0, 160.0, 120.0) and [(m[0], m[1], m[2]) for m in ref.merges] == [(sc.query_from + 3, 0, 1)]
    with tempfile.TemporaryDirectory() as td:
        exe = _build_session(td)
        path = os.path.join(td, "frames.bin")
        with open(path, "wb") as f:
            for k in range(ticks):
                fr = sc.frames(synth, k)
                for c in range(2):
                    f.write(np.ascontiguousarray(fr[c][0], np.uint8).tobytes())
                    f.write(np.ascontiguousarray(fr[c][1], np.uint16).tobytes())
        out = subprocess.run([exe, path, str(ticks), str(sc.query_from)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "ok" in out.stdout, (out.returncode, out.stdout, out.stderr)
    lines = out.stdout.splitlines()
    assert "stats %d 1" % ticks in lines
    hexes = lambda a: " ".join("%08x" % v for v in np.ascontiguousarray(a, np.float32).reshape(-1).view(np.uint32))
    k, fb, fa, T = ref.merges[0]
    assert "merge %d %d %d %s" % (k, fb, fa, hexes(T)) in lines, out.stdout
    assert ["refinement %d %d %d %d" % (r[0], r[1], r[2], int(r[3])) for r in ref.refinements] == [l for l in lines if l.startswith("refinement")]
    assert "frame_of 0 0 hosted 2 pg %d %d cons 1" % (ticks, ticks) in lines
    assert "last_pose " + hexes(ref.pose_graph[1][-1][1]) in lines
