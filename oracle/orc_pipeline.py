"""CPU ORACLE (test infrastructure): ElasticFusion::processFrame for one camera, restated over
the oracle's C functions (oracle/orc_*.c) with host arrays.

Follows elasticfusion/Core/src/ElasticFusion.cpp:99-637 with loop closure off by default (--o: the
`closeLoops` block :399-497 is skipped; local_loop_closure=True restates its device half — INACTIVE
prediction, model-to-model tracking, acceptance test, constraint sampling — and stops where the
reference hands the constraints to the CPU/CHOLMOD deformation solver), NID keyframing off by default (--nkf: fuseFrame returns
true, :639-645; nid_keyframing=True restates the gate of :646-675), tracking-failure detection off by default (--rl off: trackingOk is always true; reloc=True restates :204-244), cluster 0.
PARITY: the functions it chains are pinned to the reference's own code (orc_track.c / orc_fusion.c / orc_nid.c headers: reduce.cu
and cudafuncs.cu built for gfx950, the GLSL programs run by llvmpipe); the CHAINING — the order of the stages in processFrame, the
host decisions between them, the multi-camera session — follows ElasticFusion.cpp / MainController.cpp line by line but is
unpinned (that code needs Eigen, Pangolin and CUDA-GL interop to build); its own regression pin is tests/golden/oracle_fusion.npz.
The deformation graph is empty (it is only filled by loop closures), so clean() runs without
nodes.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

from . import orc, orc_nid


class Frame:
    pass


def _mat_vec_f32(M, v):
    """float 4x4 * (x, y, z, 1), rows 0..2: products added left to right, every operation rounded (the kernel's order)"""
    out = np.zeros(3, np.float32)
    for r in range(3):
        s = np.float32(M[r, 0] * v[0])
        s = np.float32(s + np.float32(M[r, 1] * v[1]))
        s = np.float32(s + np.float32(M[r, 2] * v[2]))
        out[r] = np.float32(s + M[r, 3])
    return out


class MapRef:
    """One reference frame's surfel map (ReferenceFrame::m_localModel): shared by every camera that has been merged into it."""

    def __init__(self):
        # GlobalModel's per-cluster buffers (GlobalModel.h:99-109): the constructor makes cluster 0 and makes it current
        # (GlobalModel.cpp:56-59); `initialise` with an unknown id adds buffers and switches to them (:266-277) — nothing ever switches
        # back; every other member (model(), fuse, clean, lastCount, downloadMap, consume) works on the current cluster
        self.clusters = {0: np.zeros(0, orc.SURFEL_DTYPE)}
        self.current = 0

    model = property(lambda self: self.clusters[self.current], lambda self, v: self.clusters.__setitem__(self.current, v))

    def isCluster(self, cluster):
        return cluster in self.clusters

    def initialise(self, surfels, cluster):
        """GlobalModel::initialise(rawFeedback, filteredFeedback, cluster, pose): `surfels` = what the init program makes of the
        feedback buffers.  A known cluster's id does NOT make it current (:350: only its buffers are looked up, and what is written
        and counted afterwards is indexed by current_cluster all the same)."""
        if cluster not in self.clusters:
            self.clusters[cluster] = np.zeros(0, orc.SURFEL_DTYPE)
            self.current = cluster
        self.clusters[self.current] = surfels


class ElasticFusion:
    model = property(lambda self: self.map.model, lambda self, v: setattr(self.map, "model", v))

    def __init__(self, width, height, K, timeDelta=200, confidence=10.0, depthCut=3.0, icpWeight=10.0, fastOdom=False, so3=True,
                 frameToFrameRGB=False, pyramid=True, hybrid_tracking=True, rgbOnly=False, timeIdx=0, maxDepthProcessed=25.0,
                 model_capacity=None, nid_keyframing=False, nid_threshold=0.80, nid_depth_lambda=0.7, nid_bins_img=64,
                 nid_bins_depth=500, nid_pyramid_level=0, local_loop_closure=False, reloc=False):
        self.W, self.H, self.K = width, height, tuple(float(v) for v in K)
        self.timeDelta, self.confidence, self.depthCut, self.icpWeight = timeDelta, confidence, depthCut, icpWeight
        self.fastOdom, self.so3, self.frameToFrameRGB, self.pyramid = fastOdom, so3, frameToFrameRGB, pyramid
        self.hybrid_tracking, self.rgbOnly, self.timeIdx = hybrid_tracking, rgbOnly, timeIdx
        self.maxDepthProcessed = maxDepthProcessed
        self.cap = model_capacity
        self.nid_keyframing, self.nid_threshold, self.nid_depth_lambda = nid_keyframing, nid_threshold, nid_depth_lambda
        self.nid_bins_img, self.nid_bins_depth, self.nid_pyramid_level = nid_bins_img, nid_bins_depth, nid_pyramid_level
        self.framesSinceLastFusion = 0
        self.nidScores = []
        fx, fy, cx, cy = self.K
        self.frameToModel = orc.Odometry(width, height, cx, cy, fx, fy)
        self.reloc, self.lost, self.trackingCount = reloc, False, 0  # --rl (:204-244)
        self.local_loop_closure = local_loop_closure
        self.modelToModel = orc.Odometry(width, height, cx, cy, fx, fy) if local_loop_closure else None  # Context.h:317-378
        self.old = None
        self.map = MapRef()
        self.currPose = np.eye(4, dtype=np.float32)
        self.tick = 1
        self.initialised = False
        self.last = Frame()

    # ElasticFusion::predict (:688-746)
    def predict(self, confidence):
        img, vtx, nrm, tim = orc.splat_predict(self.model, self.currPose, self.K, self.H, self.W, self.maxDepthProcessed, confidence,
                                               self.tick, self.timeIdx, self.tick, self.timeDelta, True)
        # passthrough = lost / lost || frameToFrameRGB (:704-712)
        fv, fn, fi = orc.fill_in(vtx, nrm, img, self.depth_filtered, self.rgba, self.K, self.lost, self.lost or self.frameToFrameRGB)
        self.pred = (img, vtx, nrm, tim)
        self.fill = (fi, fv, fn)

    # closeLoops block without a fern match (:399-497): rawGraph is empty and the camera is never lost
    def localLoop(self):
        K, H, W = self.K, self.H, self.W
        # combinedPredict(currPose, model, maxDepthProcessed, confidenceThreshold, 0, id, tick - timeDelta, timeDelta, INACTIVE) (:403-406)
        self.old = orc.splat_predict(self.model, self.currPose, K, H, W, self.maxDepthProcessed, self.confidence, 0, self.timeIdx,
                                     self.tick - self.timeDelta, self.timeDelta, False)
        oimg, ovtx, onrm, otim = self.old
        m = self.modelToModel
        m.initICPModel(ovtx, onrm, self.maxDepthProcessed, self.currPose)  # :410-412
        m.initRGBModel(oimg)  # :413
        m.initICPMaps(self.pred[1], self.pred[2], self.maxDepthProcessed)  # :415-417
        m.initRGB(self.pred[0])  # :418
        t, R, res = m.getIncrementalTransformation(self.currPose[:3, 3], self.currPose[:3, :3], False, 10.0, self.pyramid, self.fastOdom,
                                                   False)  # :424-425
        covar = orc.covariance(np.array(res.lastA))  # :427
        covOk = not any(covar[i, i] > 8e-05 for i in range(6))  # :428-435
        estPose = np.eye(4, dtype=np.float32)
        estPose[:3, 3] = t
        estPose[:3, :3] = R
        loop = Frame()
        loop.track, loop.estPose, loop.covar = res, estPose, covar
        loop.ok = bool(covOk and res.lastICPCount > 15000 and res.lastICPError < 0.0003)  # :441-442
        loop.constraints = np.zeros((0, 7), np.float32)
        if loop.ok:
            dh, dw = H // 20, W // 20
            cons = orc.resize_nn(self.pred[1], dh, dw)  # resize.vertex(vertexTex, consBuff) (:443)
            times = orc.resize_nn(otim, dh, dw)  # resize.time(oldTimeTex, timesBuff) (:444)
            rows = []
            for i in range(dw):  # column-major walk (:446-447)
                for j in range(dh):
                    p = cons[j, i]
                    if p[2] > 0 and p[2] < self.maxDepthProcessed and times[j, i] > 0:
                        ph = np.array([p[0], p[1], p[2], 1.0], np.float32)
                        raw = (self.currPose.astype(np.float32) @ ph)[:3]
                        mod = (estPose @ ph)[:3]
                        rows.append(np.concatenate([raw, mod, [np.float32(times[j, i])]]).astype(np.float32))
            if rows:
                loop.constraints = np.stack(rows)
            # Deformation::constrain (:481) is the CPU/CHOLMOD graph optimisation: not part of the
            # device path; the pose is left as tracked (the `constrain() == false` branch)
        return loop

    # The block `if (hybrid_loops && orbTcwOld && orbTcwNew)` of processFrame (:292-326, active_new = False) and the first half of
    # ElasticFusion::applyGlobalLoop (:1148-1200, active_new = True): ACTIVE prediction at one pose, INACTIVE at the other,
    # W/20 x H/20 samples -> the arguments of Deformation::addConstraint.  (Deformation::constrain itself is CPU/CHOLMOD: not restated.)
    def globalLoopConstraints(self, orbTcwOld, orbTcwNew, active_new):
        K, H, W = self.K, self.H, self.W
        old = np.asarray(orbTcwOld, np.float32).reshape(4, 4)
        new = np.asarray(orbTcwNew, np.float32).reshape(4, 4)
        act, ina = (new, old) if active_new else (old, new)
        # predict(context, rf) with currPose = the ACTIVE pose: only the vertex texture is read afterwards
        _, vtx, _, _ = orc.splat_predict(self.model, act, K, H, W, self.maxDepthProcessed, self.confidence, self.tick, self.timeIdx, self.tick,
                                         self.timeDelta, True)
        self.old = orc.splat_predict(self.model, ina, K, H, W, self.maxDepthProcessed, self.confidence, 0, self.timeIdx,
                                     self.tick - self.timeDelta, self.timeDelta, False)
        dh, dw = H // 20, W // 20
        cons = orc.resize_nn(vtx, dh, dw)  # resize.vertex(vertexTex, consBuff)
        times = orc.resize_nn(self.old[3], dh, dw)  # resize.time(oldTimeTex, timesBuff)
        rows = []
        for i in range(dw):  # columns outer (:303-304)
            for j in range(dh):
                p = cons[j, i]
                if p[2] > 0 and p[2] < self.maxDepthProcessed and (not active_new or times[j, i] > 0):
                    ph = np.array([p[0], p[1], p[2], 1.0], np.float32)
                    raw = _mat_vec_f32(old, ph)
                    mod = _mat_vec_f32(new, ph)
                    rows.append(np.concatenate([raw, mod, [np.float32(times[j, i])]]).astype(np.float32))
        return np.stack(rows) if rows else np.zeros((0, 7), np.float32)

    # second half of applyGlobalLoop (:1222-1239): predict, predictIndices, clean(rawGraph, isFern = accepted)
    def applyGlobalLoopEnd(self, rawGraph=None, accepted=False):
        td = self.timeDelta + self.framesSinceLastFusion
        self.predict(self.confidence)  # the reference's order: predict, predictIndices, clean (the images show the map before the clean)
        im = orc.index_map(self.model, self.currPose, self.K, self.H, self.W, self.tick, self.timeIdx, self.maxDepthProcessed, td)
        nodes = None if rawGraph is None or not len(rawGraph) else np.ascontiguousarray(rawGraph, np.float32).reshape(-1, 16)
        self.model = orc.model_clean(self.model, np.zeros(0, orc.SURFEL_DTYPE), self.currPose, self.tick, self.timeIdx, im[0], im[1], im[2],
                                     self.K, self.confidence, td, self.maxDepthProcessed, nodes=nodes, depthSynth=None, cap=self.cap,
                                     isFern=int(bool(accepted)))

    # ElasticFusion::fuseFrame (:639-677): the candidate key frame is the model prediction at the new
    # pose (GlobalPredict, :273); its "old" (INACTIVE) textures are never rendered with loop closure
    # off, i.e. no prediction anywhere (NaN depth, black image)
    def fuseFrame(self):
        if not self.nid_keyframing:
            self.nidScores.append(0.0)
            return True, 0.0
        img = orc.imageBGRToIntensity(self.pred[0])
        dmap = orc.verticesToDepth(self.pred[1], self.maxDepthProcessed)
        for _ in range(self.nid_pyramid_level):  # MutualInformation::nidImg (:169-174)
            img, dmap = orc.pyrDownUcharGauss(img), orc.pyrDownGaussF(dmap)
        if self.old is not None:
            # KeyFrame (KeyFrame.h:83-172) with a rendered INACTIVE view.  (The reference copies the old
            # vertex texture with the byte count of an already released array, KeyFrame.h:147-150, so what
            # its verticesToDepth reads there is undefined; the evident intent is restated.)
            old_img = orc.imageBGRToIntensity(self.old[0])
            old_d = orc.verticesToDepth(self.old[1], self.maxDepthProcessed)
            for _ in range(self.nid_pyramid_level):
                old_img, old_d = orc.pyrDownUcharGauss(old_img), orc.pyrDownGaussF(old_d)
        else:
            old_img = np.zeros_like(img)
            old_d = np.full_like(dmap, np.nan)
        L = self.nid_pyramid_level
        nid_img, _ = orc_nid.nid_img(img, old_img, dmap, old_d, self.frameToModel.buffer(7, L), self.nid_bins_img)
        nid_depth, _ = orc_nid.nid_depth(dmap, old_d, self.frameToModel.buffer(5, L), self.nid_bins_depth, self.maxDepthProcessed * 1000.0)
        score = float(np.float32(self.nid_depth_lambda) * nid_depth + (np.float32(1.0) - np.float32(self.nid_depth_lambda)) * nid_img)
        self.nidScores.append(score)
        return score > self.nid_threshold, score

    def computeFeedbackBuffers(self):
        """Context::computeFeedbackBuffers (Context.h:211-223): the surfels of the CURRENT textures at the current tick.  processFrame
        calls it on the first frame only (ElasticFusion.cpp:133); a later `initialise` of a new cluster (:508-515) therefore reads
        the first frame's buffers unless the caller (the GUI's raw-cloud view, MainController.cpp:476) has computed them again."""
        self.feedback = orc.model_initialise(self.rgba, self.depth_metric, self.depth_metric_filtered, self.K, self.tick, self.timeIdx,
                                             float(int(self.maxDepthProcessed)), self.cap)

    def processFrame(self, rgb, depth, inPose=None, weightMultiplier=1.0, deform=None, orbTcwOld=None, orbTcwNew=None, cluster=0):
        """deform(loop) stands in for Deformation::constrain (ElasticFusion.cpp:481, CPU/CHOLMOD, not
        restated): called with the loop candidate, it returns None (no deformation) or
        (rawGraph nodes n x 16, corrected pose)."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        if rgb.shape[2] == 3:
            rgb = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
        self.rgba = rgb
        depth = np.ascontiguousarray(depth, np.uint16)
        # filterDepth / metriciseDepth (:118-119, :748-768)
        # far depth cut-offs raise the static first-iteration exponents of the canonical sums (csrc/canon.hpp depth_exp_bias: the same rule)
        import math

        bias = min(100, 2 * (math.frexp(float(np.float32(self.depthCut)))[1] - 2)) if self.depthCut > 4.0 else 0
        self.frameToModel.setExpBias(bias)
        if self.modelToModel is not None:
            self.modelToModel.setExpBias(bias)
        self.depth_filtered = orc.depth_bilateral(depth, self.depthCut)
        self.depth_metric = orc.depth_metric(depth, self.depthCut)
        self.depth_metric_filtered = orc.depth_metric(self.depth_filtered, self.depthCut)
        out = Frame()
        out.fill_in = False
        out.weighting = 1.0
        out.track = None
        out.nid_score = 0.0
        out.loop = None
        out.tracking_ok = True
        fused = False
        if not self.initialised:  # first run (:132-152)
            pose = np.eye(4, dtype=np.float32) if inPose is None else np.asarray(inPose, np.float32).reshape(4, 4)
            self.currPose = pose.copy()
            # computeFeedbackBuffers(const int& maxDepthProcessed): 25.0f -> 25 (Context.h:211)
            self.computeFeedbackBuffers()
            self.map.initialise(self.feedback.copy(), cluster)
            self.frameToModel.initFirstRGB(self.rgba)
            self.initialised = True
            fused = True
        else:
            lastPose = self.currPose.copy()  # :158
            trackingOk = True
            if inPose is not None:  # :164 (the reference dereferences NULL; the previous pose is used instead)
                self.currPose = np.asarray(inPose, np.float32).reshape(4, 4).copy()
            self.predict(0.7)  # :165
            shouldFillIn = not orc.dense_enough(self.pred[0])  # :166-167
            out.fill_in = shouldFillIn
            if self.hybrid_tracking:
                vtx, nrm = (self.fill[1], self.fill[2]) if shouldFillIn else (self.pred[1], self.pred[2])
                self.frameToModel.initICPModel(vtx, nrm, self.maxDepthProcessed, self.currPose)  # :173-178
                self.frameToModel.initRGBModel(self.fill[0] if (shouldFillIn or self.frameToFrameRGB) else self.pred[0])  # :179-181
                self.frameToModel.initICP(self.depth_filtered, self.maxDepthProcessed)  # :183-184
                self.frameToModel.initRGB(self.rgba)  # :185
                t, R, res = self.frameToModel.getIncrementalTransformation(self.currPose[:3, 3], self.currPose[:3, :3], self.rgbOnly,
                                                                         self.icpWeight, self.pyramid, self.fastOdom, self.so3)
                # tracking-failure detection (:204-244); lastFrameRecovery is only set by the compiled-out fern block
                trackingOk = (not self.reloc) or bool(res.lastICPError < 1e-04)
                if self.reloc and not self.lost:
                    cov = orc.covariance(np.array(res.lastA))
                    if any(cov[i, i] > 1e-04 for i in range(6)):
                        trackingOk = False
                    if not trackingOk:
                        self.trackingCount += 1
                        if self.trackingCount > 10:
                            self.lost = True
                    else:
                        self.trackingCount = 0
                out.tracking_ok = bool(trackingOk)
                self.currPose[:3, 3] = t
                self.currPose[:3, :3] = R
                out.track = res
            weighting = orc.velocity_weight(self.currPose, lastPose, weightMultiplier)  # :252-268
            out.weighting = weighting
            self.predict(self.confidence)  # :273
            rawGraph = None
            out.global_loop = None
            if orbTcwOld is not None and orbTcwNew is not None:  # hybrid_loops (:292-350)
                out.global_loop = self.globalLoopConstraints(orbTcwOld, orbTcwNew, False)
                res = deform(out.global_loop) if deform is not None else None  # Deformation::constrain (:337): fills rawGraph, pose unchanged
                if res is not None and res[0] is not None:
                    rawGraph = np.ascontiguousarray(res[0], np.float32).reshape(-1, 16)
                self.predict(self.confidence)  # :349
            if self.local_loop_closure and not self.lost and (rawGraph is None or not len(rawGraph)):  # `rawGraph.size() == 0` (:399)
                out.loop = self.localLoop()  # :399-497
                res = deform(out.loop) if deform is not None else None
                if res is not None:
                    rawGraph = np.ascontiguousarray(res[0], np.float32).reshape(-1, 16)
                    self.currPose = np.asarray(res[1], np.float32).reshape(4, 4).copy()  # context.currPose() = estPose (:489)
            if rawGraph is not None and len(rawGraph):  # fuseFrame(context, deforming) (:641-644)
                self.nidScores.append(0.0)
                fuse, out.nid_score = True, 0.0
            else:
                fuse, out.nid_score = self.fuseFrame()  # :501
            td = self.timeDelta + self.framesSinceLastFusion  # :518,:541,:563
            if not self.rgbOnly and trackingOk and not self.lost and fuse:  # fusion (:506-564)
                if not self.map.isCluster(cluster):  # :508-515
                    self.map.initialise(self.feedback.copy(), cluster)
                im = orc.index_map(self.model, self.currPose, self.K, self.H, self.W, self.tick, self.timeIdx, self.maxDepthProcessed, td)
                self.model, newU, _ = orc.model_fuse(self.model, self.currPose, self.tick, self.timeIdx, self.rgba, self.depth_metric,
                                                     self.depth_metric_filtered, im[0], im[1], im[3], self.K, self.maxDepthProcessed,
                                                     weighting)
                im = orc.index_map(self.model, self.currPose, self.K, self.H, self.W, self.tick, self.timeIdx, self.maxDepthProcessed, td)
                self.imap = im
                dsyn = None
                if rawGraph is not None and len(rawGraph):  # synthesizeDepth (:541-553)
                    dsyn = orc.splat_predict(self.model, self.currPose, self.K, self.H, self.W, self.maxDepthProcessed, self.confidence,
                                             self.tick, self.timeIdx, self.tick - td, 65535, False, depth_only=True)
                self.model = orc.model_clean(self.model, newU, self.currPose, self.tick, self.timeIdx, im[0], im[1], im[2], self.K,
                                             self.confidence, td, self.maxDepthProcessed, nodes=rawGraph, depthSynth=dsyn, cap=self.cap)
                fused = True
            self.framesSinceLastFusion = 0 if fuse else self.framesSinceLastFusion + 1  # :567-568
        self.predict(self.confidence)  # finalPredict (:586)
        if not self.lost:
            self.tick += 1  # :588-591
        out.lost = self.lost
        out.pose = self.currPose.copy()
        out.surfels = len(self.model)
        out.fused = fused
        out.tick = self.tick
        self.last = out
        return out


def refine_inter_map(owner, o, fill, timeIdx, tick, currPose, recoveryPose, cov_thresh=1e-05, icp_err_thresh=2e-05, icp_count_thresh=35000):
    """ReferenceFrame::resolveRelativeTransformationFern after findFrame (ReferenceFrame.h:72-110).  owner: the ElasticFusion whose
    map was matched; o: the reference frame's m_rgbd (an orc.Odometry that persists between calls); fill = (image, vertex, normal),
    timeIdx, tick, currPose: the querying camera's fill-in textures, Context::id(), tick and pose."""
    from . import orc_ferns

    rec = np.asarray(recoveryPose, np.float32).reshape(4, 4).copy()
    cutoff = float(int(owner.maxDepthProcessed))  # `const int depthCutoff` (:42)
    oimg, ovtx, onrm, _ = orc.splat_predict(owner.model, rec, owner.K, owner.H, owner.W, cutoff, owner.confidence, 0, timeIdx, tick,
                                            owner.timeDelta, False)  # :72-80
    fi, fv, fn = fill
    o.initICPModel(ovtx, onrm, cutoff, rec)  # :82
    o.initICPMaps(fv, fn, cutoff)  # :83
    o.initRGBModel(oimg)  # :85 (stated reading: the INACTIVE prediction's image)
    o.initRGB(fi)  # :86
    t, R, res = o.getIncrementalTransformation(rec[:3, 3].copy(), rec[:3, :3].copy(), False, 10.0, True, False, True, interMap=True)  # :88-90
    refined = rec.copy()
    refined[:3, 3], refined[:3, :3] = t, R
    T = orc_ferns._mul44(refined, orc.inv4f(np.asarray(currPose, np.float32).reshape(4, 4)))  # :95
    covar = orc.covariance(np.array(res.lastA))  # :98
    covOk = not any(covar[i, i] > float(np.float32(cov_thresh)) for i in range(6))  # :101-108 (double against a float option)
    ok = bool(covOk and res.lastICPError < np.float32(icp_err_thresh) and res.lastICPCount > icp_count_thresh)  # :110
    return dict(accepted=ok, cov_ok=covOk, relativeTransform=T, refinedPose=refined, cov_diag=[float(covar[i, i]) for i in range(6)],
                lastICPError=float(res.lastICPError), lastICPCount=float(res.lastICPCount), track=res, old=(oimg, ovtx, onrm))


# ---- collaborative session: several cameras, maps that merge (checker of densemonoslam_amd.collab.CollabSession) -----------------
class Session:
    """The reference's multi-camera loop in ONE process (GUI/src/MainController.cpp:262-400: every camera's processFrame in turn)
    with the inter-map block of ElasticFusion::processFrame (ElasticFusion.cpp:595-632, compiled out there with `if (false)`)
    switched on, in the form DESIGN.md 7 gives it for one camera per GPU:

      per tick, for every camera in id order: processFrame; then every camera offers its frame block (W/8 x H/8 thumbnails of the
      fill-in textures, pose, tick) to the fern database of ITS map (Ferns::addFrame); then every camera queries the database of
      every OTHER map in reference-frame order (Ferns::findFrame with interMap = true on the thumbnails: search, code agreement,
      thumbnail-sized ICP + photometric check); the first verified match merges: relativeTransform = recoveryPose *
      currPose.inverse() (ReferenceFrame.h:98), the matched map consumes the querying camera's map
      (ReferenceFrame::consumeReferenceFrame, :121-150: surfels, key frames, cameras - pose, pose graph and relative constraints
      re-based), at most one merge per map and tick.

    On a fern match the owner of the matched map runs the second half of resolveRelativeTransformationFern (`refine` below,
    ReferenceFrame.h:72-110): INACTIVE prediction of its map at recoveryPose, full-resolution ICP + RGB refinement against the
    querying camera's fill-in textures, relativeTransform from the REFINED pose, acceptance on covariance / error / count.

    Differences from the compiled-out reference block, all of them: by default queries run after all cameras of the tick have been
    processed rather than inside each camera's processFrame (`query_inside_frame = True` is the reference's order); two readings of the block's undefined inputs are stated at `refine`;
    `full_refine = False` stops after the fern database's own verification (rounds 3-4)."""

    def __init__(self, n, width, height, K, fern_seed=20260929, fern_threshold=0.3095, fern_num=500, fern_max_depth_mm=3000,
                 fern_photo_thresh=115.0, inter_map=1, query_from=0, full_refine=True, cov_thresh=1e-05, icp_err_thresh=2e-05,
                 icp_count_thresh=35000, wake_latency=None, query_inside_frame=False, **opts):
        from . import orc_ferns

        # False: the queries of a tick run after all cameras' frames (the form DESIGN.md 7 gives the block for one camera per GPU).
        # True: the reference's ORDER - the block sits inside processFrame (ElasticFusion.cpp:595-632), camera c queries and merges
        # before camera c + 1's frame of the same tick is processed (dms_session_params.query_inside_frame)
        self.query_inside_frame = query_inside_frame
        assert not (query_inside_frame and wake_latency is not None)

        # None: every tick queries (the reference's loop, dms_session_step); 3: dms_session_step_async's schedule
        self.wake_latency, self.hits, self.woken, self.valid_from = wake_latency, {}, [], 0
        self.inter_map = inter_map  # Ferns::findFrame's interMap argument (2: see dmslam_ferns.h)
        self.query_from = query_from  # first tick index at which cameras query other maps (0 = from the start, as the reference would)
        self.n, self.W, self.H, self.K = n, width, height, tuple(float(v) for v in K)
        self.cams = [ElasticFusion(width, height, K, timeIdx=i, **opts) for i in range(n)]
        Kf = self.K
        mk = lambda: orc_ferns.Ferns(width, height, K, num=fern_num, maxDepth_mm=fern_max_depth_mm, photoThresh=fern_photo_thresh, seed=fern_seed,
                                     make_odometry=lambda: orc.Odometry(width // 8, height // 8, Kf[2] / 8, Kf[3] / 8, Kf[0] / 8, Kf[1] / 8))
        self.ferns = [mk() for _ in range(n)]     # one database per reference frame, indexed by the frame's first camera
        self.frame_of = list(range(n))            # m_contextToReferenceFrameMap: camera -> reference frame
        self.fern_threshold = fern_threshold
        self.pose_graph = [[] for _ in range(n)]  # Context::poseGraph(): (tick, pose) per processed frame
        self.relative_cons = [[] for _ in range(n)]  # Context::relativeCons(): rows {src xyz, target xyz} the caller's solver produced
        self.merges = []                          # (tick index, consuming frame, consumed frame, relativeTransform)
        self.matches = []
        # ReferenceFrame::m_rgbd, one per reference frame (created on first use: it keeps its state between refinements);
        # Options::covThresh / icpErrThresh / icpCountThresh (Options.h:91-94)
        self.full_refine = full_refine
        self.cov_thresh, self.icp_err_thresh, self.icp_count_thresh = cov_thresh, icp_err_thresh, icp_count_thresh
        self.rgbd = {}
        self.refinements = []                     # (tick index, camera, frame, accepted, result dict)

    def thumbs(self, cam):
        from . import orc_ferns

        fi, fv, fn = cam.fill
        th, tw = self.H // 8, self.W // 8
        return orc_ferns.resize_nearest(fi, th, tw), orc_ferns.resize_nearest(fv, th, tw), orc_ferns.resize_nearest(fn, th, tw)

    def step(self, frames, k):
        """frames[i] = (rgb, depth) of camera i; k = the tick index (for the log)."""
        from . import orc_ferns

        # time slots of the clean's health test: NUM_CAMERAS = 3 in the reference (Shaders/size.glsl:2), which a fourth camera overruns
        # (Shaders/Vertex.cpp:49); max(3, cameras) here, as dms_session_default_params.  The oracle's count is a global: set per step.
        orc.set_num_sensors(max(3, self.n))
        try:
            return self._step_inside(frames, k) if self.query_inside_frame else self._step_batched(frames, k)
        finally:
            orc.set_num_sensors(3)

    def _step_batched(self, frames, k):
        from . import orc_ferns

        outs = []
        for i, cam in enumerate(self.cams):
            tick_before = cam.tick
            outs.append(cam.processFrame(frames[i][0], frames[i][1]))
            self.pose_graph[i].append((tick_before, cam.currPose.copy()))
        blocks = [self.thumbs(cam) for cam in self.cams]
        for i, cam in enumerate(self.cams):  # the own map's database sees every frame of its cameras (processFerns sits under `if (!lost)`, ElasticFusion.cpp:588-591)
            if cam.lost:
                continue
            img, v, nrm = blocks[i]
            self.ferns[self.frame_of[i]]._add(img, v, nrm, cam.currPose.copy(), cam.tick, self.fern_threshold)
        if self.wake_latency is not None:
            # dms_session_step_async's rule (dmslam_session.h): the descriptor half of every query runs each tick; the inter-map block
            # below runs at tick k iff that half hit for any eligible pair at tick k - wake_latency and no merge has happened since
            self.hits[k] = k >= self.query_from and any(
                self.ferns[fb].searchHit(blocks[a], cam.tick, interMap=True)
                for a, cam in enumerate(self.cams) for fb in sorted(set(self.frame_of)) if fb != self.frame_of[a])
            j = k - self.wake_latency
            if not (j >= self.valid_from and self.hits.get(j, False)):
                return outs
            self.woken.append(k)
        busy = set()
        n_merges = len(self.merges)
        for a, cam in enumerate(self.cams):  # queries, in camera order
            fa = self.frame_of[a]
            if fa in busy or k < self.query_from:
                continue
            for c in range(self.n):  # m_contextToReferenceFrameMap: context ids ascending, each mapped to its frame (ElasticFusion.cpp:598-599)
                fb = self.frame_of[c]  # (a frame that holds several cameras is visited once per camera, as there)
                if fb == fa or fb in busy:
                    continue
                m = self.ferns[fb].findFrame(cam.currPose, None, None, None, cam.tick, lost=False, interMap=self.inter_map, thumbs=blocks[a])
                self.matches.append((k, a, fb, m["closest"], m["candidate"]))
                if m["closest"] < 0:
                    continue
                if self.full_refine:
                    r = self.refine(fb, a, m["estPose"])
                    self.refinements.append((k, a, fb, r["accepted"], r))
                    if not r["accepted"]:
                        continue
                    T = r["relativeTransform"]
                else:
                    T = orc_ferns._mul44(m["estPose"], orc.inv4f(cam.currPose))
                self.consume(fb, fa, T)
                self.merges.append((k, fb, fa, T.copy()))
                busy.update((fa, fb))
                break
        if len(self.merges) != n_merges:
            self.valid_from = k + 1
        return outs

    def _step_inside(self, frames, k):
        """One tick in the reference's order: MainController.cpp:262-400 serves the cameras in turn, and each processFrame ends with
        processFerns (:588-591) and the inter-map block (:595-632) - before the next camera's frame."""
        outs = []
        for a, cam in enumerate(self.cams):
            tick_before = cam.tick
            outs.append(cam.processFrame(frames[a][0], frames[a][1]))
            self.pose_graph[a].append((tick_before, cam.currPose.copy()))
            blk = self.thumbs(cam)
            if not cam.lost:
                self.ferns[self.frame_of[a]]._add(blk[0], blk[1], blk[2], cam.currPose.copy(), cam.tick, self.fern_threshold)
            if k < self.query_from:
                continue
            fa = self.frame_of[a]
            for c in range(self.n):  # for (auto& kv : m_contextToReferenceFrameMap)
                fb = self.frame_of[c]
                if fb == fa:
                    continue
                m = self.ferns[fb].findFrame(cam.currPose, None, None, None, cam.tick, lost=False, interMap=self.inter_map, thumbs=blk)
                self.matches.append((k, a, fb, m["closest"], m["candidate"]))
                if m["closest"] < 0:
                    continue
                if self.full_refine:
                    r = self.refine(fb, a, m["estPose"])
                    self.refinements.append((k, a, fb, r["accepted"], r))
                    if not r["accepted"]:
                        continue
                    T = r["relativeTransform"]
                else:
                    T = orc_ferns._mul44(m["estPose"], orc.inv4f(cam.currPose))
                self.consume(fb, fa, T)
                self.merges.append((k, fb, fa, T.copy()))
                break
        return outs

    def refine(self, fb, a, recoveryPose):
        """ReferenceFrame::resolveRelativeTransformationFern after findFrame (ReferenceFrame.h:66-110), for camera a against frame fb.
        Stated readings (the block is compiled out in the reference and was never run): :85 hands initRGBModel `m_index.imageTex()`,
        the ACTIVE colour target of an IndexMap that only renders INACTIVE - a texture nothing wrote; the INACTIVE prediction's own
        image is used (as ElasticFusion.cpp:413 does for the model-to-model tracker).  m_rgbd's lastNextImage before its first call
        is uninitialised device memory there, zeros here.  Kept literally: the call order initICPModel, initICP, initRGBModel,
        initRGB (:82-86), which makes lastDepth the LIVE camera's depth (populateRGBDData reads vmaps_tmp), and `const int
        depthCutoff` (:42), which truncates maxDepthProcessed."""
        owner = next(c for i, c in enumerate(self.cams) if self.frame_of[i] == fb)
        cam = self.cams[a]
        if fb not in self.rgbd:
            Kf = self.K
            self.rgbd[fb] = orc.Odometry(self.W, self.H, Kf[2], Kf[3], Kf[0], Kf[1])
        return refine_inter_map(owner, self.rgbd[fb], cam.fill, cam.timeIdx, cam.tick, cam.currPose, recoveryPose, self.cov_thresh,
                                self.icp_err_thresh, self.icp_count_thresh)

    def consume(self, fb, fa, T):
        """reference frame fb consumes fa (ReferenceFrame::consumeReferenceFrame)"""
        from . import orc_ferns

        owner = next(c for i, c in enumerate(self.cams) if self.frame_of[i] == fb)
        other = next(c for i, c in enumerate(self.cams) if self.frame_of[i] == fa)
        owner.map.model = orc.model_consume(owner.map.model, other.map.model, T)
        self.ferns[fb].consume(self.ferns[fa], T, self.fern_threshold)
        for i, cam in enumerate(self.cams):
            if self.frame_of[i] != fa:
                continue
            cam.currPose = orc_ferns._mul44(T, cam.currPose)
            cam.map = owner.map
            self.frame_of[i] = fb
            self.pose_graph[i] = [(t, orc_ferns._mul44(T, p)) for t, p in self.pose_graph[i]]
            self.relative_cons[i] = [np.concatenate([orc_ferns._mul4v(T, np.append(r[:3], np.float32(1)))[:3],
                                                     orc_ferns._mul4v(T, np.append(r[3:6], np.float32(1)))[:3]]).astype(np.float32)
                                     for r in self.relative_cons[i]]
