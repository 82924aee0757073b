"""Collaborative session past the merge: cameras whose maps merge, across ranks.

The reference runs every camera's processFrame in turn in one process (GUI/src/MainController.cpp:262-400); its inter-map block
(Core/src/ElasticFusion.cpp:595-632, compiled out with `if (false)`) lets a camera query every OTHER reference frame's fern
database and, on a verified match, has that reference frame consume the camera's own (ReferenceFrame::consumeReferenceFrame,
ReferenceFrame.h:121-150): surfels and key frames are appended, the consumed frame's cameras move over with their pose, pose graph and
relative constraints re-based, and from then on all of them track against and fuse into ONE map.

Here camera c is READ on rank c % world for the whole session (its frames arrive there) and is HOSTED - its Context lives, its
frames are processed - on the rank that hosts its reference frame: rank c % world until its frame is consumed, then the consuming
frame's rank.  Per tick, on every rank, `CollabSession.step`:

  1. forward   a rank whose camera is hosted elsewhere sends that camera's frame to the host, point to point (RGB8 + depth u16:
               1.5 MB at 640 x 480, nothing against one xGMI link);
  2. frames    every hosted camera's processFrame, in camera-id order (one thread, one stream: the reference's loop);
  3. publish   every hosted camera's frame block (W/8 x H/8 thumbnails of its fill-in textures | pose | tick | camera id) is offered
               to ITS map's fern database (Ferns::addFrame) and all-gathered (the one collective; slots per rank = the most cameras
               any rank hosts, 1 before the first merge);
  4. query     owner computes: for every reference frame hosted here and every camera of another frame, Ferns::findFrame with
               interMap = true on the gathered thumbnails (search, code agreement, thumbnail-sized ICP + photometric check);
               the (closest, recoveryPose) table is all-gathered (18 floats per pair);
  5. decide    every rank walks the same table in the reference's sequential order - cameras in id order, reference frames in
               id order.  A fern match is a CANDIDATE: the owner of the matched map runs the second half of
               ReferenceFrame::resolveRelativeTransformationFern (ReferenceFrame.h:72-110, dms_refframe_refine: INACTIVE prediction
               of its map at recoveryPose, full-resolution ICP + RGB refinement against the querying camera's fill-in textures,
               acceptance on covariance / error / count).  If the camera lives on another rank its three fill-in textures travel
               point to point first (36 B per pixel, 11 MB at 640 x 480 - on a candidate only); {accepted, relativeTransform} and
               every rank's status travel in one all-gather (18 floats per rank; a failure anywhere raises on every rank).  A camera
               visits the other cameras' frames in camera-id order (ElasticFusion.cpp:598-599).  First accepted candidate wins, at
               most one merge per frame and tick;
  6. merge     relativeTransform = recoveryPose * currPose.inverse() (ReferenceFrame.h:98).  Same rank: the consuming map
               consumes the other (dms_fusion_join_map), fern databases merge (dms_ferns_consume).  Across ranks: the consumed
               frame's rank sends, point to point, its surfel records, its key-frame records and per camera {pose, tick, last
               frame, pose graph, relative constraints}; the consuming rank appends them (dms_model_consume_records,
               dms_ferns_consume_records) and re-creates each camera (dms_fusion_import_camera); the sender frees its copies.

The engines behind `backend` are either the product (GpuBackend below: fusion.ElasticFusion + ferns.Ferns in HBM) or, in the CPU
tests, stand-ins with the same call surface built on the oracle - so the protocol itself (who sends what to whom, the decision
rule, the re-basing) is exercised over gloo without a GPU and compared with oracle/orc_pipeline.Session, which plays the same
session in one process.  torch.distributed carries everything (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in tests)."""
import numpy as np
import torch
import torch.distributed as dist

DMS_MAX_SENSORS = 8  # include/dmslam.h: time slots of a map = cameras it can hold (one per GPU of the node)
META_BYTES = 80  # per published block: camera id i32 | tick i32 | 2 x pad | pose 16 x f32


class CollabSession:
    def __init__(self, backend, n_cameras, width, height, rank=0, world=1, device=None, fern_threshold=0.3095, inter_map=1, query_from=0,
                 full_refine=True, cov_thresh=1e-05, icp_err_thresh=2e-05, icp_count_thresh=35000, wake_latency=None):
        self.be, self.n, self.W, self.H = backend, n_cameras, width, height
        # None: every tick queries (dms_session_step).  3: dms_session_step_async's schedule in this model - the descriptor half of every
        # query (Ferns::searchHit) runs each tick on the owner's rank, the flags are gathered, and the inter-map block runs at tick k iff
        # any eligible pair hit at tick k - 3 and no merge has happened since (include/dmslam_session.h).  The model itself stays
        # host-synchronous: what it checks is the rule, across ranks.
        self.wake_latency, self.hits, self.woken, self.valid_from = wake_latency, {}, [], 0
        self.rank, self.world = rank, world
        self.device = device or torch.device("cpu")
        self.fern_threshold = fern_threshold
        self.inter_map = inter_map  # Ferns::findFrame's interMap argument (1 = the reference's; 2: see dmslam_ferns.h)
        self.query_from = query_from  # first tick index at which cameras query other maps (0: from the start, as the reference would)
        self.frame_of = list(range(n_cameras))                    # camera -> reference frame (the id of its founding camera)
        self.host_of_frame = {c: c % world for c in range(n_cameras)}  # reference frame -> rank
        self.cams, self.ferns = {}, {}                            # hosted here: camera id -> engine, frame id -> fern database
        self.pose_graph = {c: [] for c in range(n_cameras)}       # of the cameras hosted here
        self.relative_cons = {c: [] for c in range(n_cameras)}
        self.last_frame = {}                                      # camera -> (rgb, depth) of the last processed frame (host arrays)
        if hasattr(backend, "num_sensors") and backend.num_sensors is None:
            backend.num_sensors = max(3, n_cameras)  # (before the first camera is made)
        for c in range(n_cameras):
            if c % world == rank:
                self.cams[c] = backend.make_camera(c)
                self.ferns[c] = backend.make_ferns()
        self.block_bytes = backend.block_bytes() + META_BYTES
        self.merges, self.matches = [], []
        # second half of resolveRelativeTransformationFern: one refiner (m_index + m_rgbd) per hosted reference frame, created on
        # first use; Options::covThresh / icpErrThresh / icpCountThresh (Options.h:91-94)
        self.full_refine = full_refine
        self.thresholds = (cov_thresh, icp_err_thresh, icp_count_thresh)
        self.refiners = {}
        self.refinements = []  # (tick index, camera, frame, accepted) - the same list on every rank
        limit = getattr(backend, "num_sensors", None)
        if limit is not None and n_cameras > limit:
            raise ValueError("%d cameras but the maps have %d time slots (num_sensors): a merge would fail at join / import" % (n_cameras, limit))

    # ---- helpers -------------------------------------------------------------------------------------------------------------
    def host_of_camera(self, c):
        return self.host_of_frame[self.frame_of[c]]

    def hosted(self):
        return sorted(self.cams)

    def _t(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def _send(self, arr, dst):
        t = self._t(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))
        dist.send(torch.tensor([t.numel()], dtype=torch.int64, device=self.device), dst)
        if t.numel():
            dist.send(t, dst)

    def _recv(self, src, dtype, shape=None):
        n = torch.zeros(1, dtype=torch.int64, device=self.device)
        dist.recv(n, src)
        t = torch.zeros(int(n.item()), dtype=torch.uint8, device=self.device)
        if t.numel():
            dist.recv(t, src)
        a = t.cpu().numpy().view(dtype)
        return a.reshape(shape) if shape is not None else a

    # ---- one tick ------------------------------------------------------------------------------------------------------------
    def step(self, k, my_frames):
        """my_frames: {camera id: (rgb u8 HxWx3, depth u16 HxW)} for the cameras READ on this rank.  Returns {camera: result} for
        the cameras hosted here."""
        # 1. forward the frames of cameras hosted elsewhere / receive those hosted here
        frames = {}
        for c in range(self.n):
            src, host = c % self.world, self.host_of_camera(c)
            if src == host:
                if host == self.rank:
                    frames[c] = my_frames[c]
            elif src == self.rank:
                self._send(my_frames[c][0], host)
                self._send(my_frames[c][1], host)
            elif host == self.rank:
                rgb = self._recv(src, np.uint8, (self.H, self.W, 3))
                frames[c] = (rgb, self._recv(src, np.uint16, (self.H, self.W)))
        # 2. every hosted camera's frame, in id order
        out = {}
        for c in self.hosted():
            cam = self.cams[c]
            tick_before = cam.tick()
            out[c] = cam.processFrame(frames[c][0], frames[c][1])
            self.pose_graph[c].append((tick_before, cam.pose().copy()))
            self.last_frame[c] = frames[c]
        # 3. publish: own map's database, then the all-gather
        host_counts = [sum(1 for c in range(self.n) if self.host_of_camera(c) == r) for r in range(self.world)]
        slots = max(host_counts)
        local = np.zeros((slots, self.block_bytes), np.uint8)
        for i, c in enumerate(self.hosted()):
            cam = self.cams[c]
            blk = cam.thumbnails()
            pose, tick = cam.pose(), cam.tick()
            if not cam.lost():  # processFerns sits under `if (!lost)` (ElasticFusion.cpp:588-591)
                self.ferns[self.frame_of[c]].addBlock(blk, pose, tick, self.fern_threshold)
            local[i, :len(blk)] = blk
            meta = np.zeros(META_BYTES // 4, np.float32)
            meta[:2] = np.array([c, tick], np.int32).view(np.float32)
            meta[4:20] = np.asarray(pose, np.float32).reshape(16)
            local[i, len(blk):] = meta.view(np.uint8)
        gathered = self._allgather(local.reshape(-1)).reshape(self.world, slots, self.block_bytes)
        blocks = {}
        for r in range(self.world):
            for i in range(host_counts[r]):
                raw = gathered[r, i]
                meta = raw[self.block_bytes - META_BYTES:].view(np.float32)
                c, tick = (int(v) for v in meta[:2].view(np.int32))
                blocks[c] = (raw[:self.block_bytes - META_BYTES], meta[4:20].reshape(4, 4).copy(), tick)
        if self.wake_latency is not None:
            mine = k >= self.query_from and any(self.ferns[fb].searchHit(blocks[a][0]) for fb in sorted(self.ferns) for a in range(self.n)
                                                if self.frame_of[a] != fb)
            flags = self._allgather(np.array([1 if mine else 0], np.uint8))
            self.hits[k] = bool(flags.any())
            j = k - self.wake_latency
            if not (j >= self.valid_from and self.hits.get(j, False)):
                return out
            self.woken.append(k)
        # 4. owner computes: every hosted reference frame against every camera of another frame.  A failure on one rank must not
        # leave the others waiting in the next collective: it travels in the table's last row and every rank raises together.
        table = np.zeros((self.n + 1, self.n, 18), np.float32)  # [camera a][frame fb] = valid, closest, recoveryPose; [n][0][0] = error flag
        failure = None
        try:
            for fb in sorted(self.ferns):
                for a in range(self.n):
                    if self.frame_of[a] == fb or k < self.query_from:
                        continue
                    blk, pose, tick = blocks[a]
                    closest, cand, est = self.ferns[fb].findFrameThumbs(blk, pose, tick, self.inter_map)
                    table[a, fb, 0], table[a, fb, 1] = 1.0, float(closest)
                    table[a, fb, 2:] = np.asarray(est, np.float32).reshape(16)
                    self.matches.append((k, a, fb, closest, cand))
        except Exception as e:  # noqa: BLE001 (re-raised below, on every rank)
            failure = e
            table[self.n, 0, 0] = 1.0
        del self.matches[:-4096]
        tables = self._allgather(table.reshape(-1).view(np.uint8)).view(np.float32).reshape(self.world, self.n + 1, self.n, 18)
        if tables[:, self.n, 0, 0].any():
            bad = [r for r in range(self.world) if tables[r, self.n, 0, 0]]
            raise RuntimeError("inter-map query failed on rank(s) %s at tick %d" % (bad, k)) from failure
        # 5. the same walk on every rank (oracle/orc_pipeline.Session.step: cameras in id order, frames in id order)
        decided, busy = [], set()
        for a in range(self.n):
            fa = self.frame_of[a]
            if fa in busy or k < self.query_from:
                continue
            for c in range(self.n):  # m_contextToReferenceFrameMap: context ids ascending, each mapped to its frame (ElasticFusion.cpp:598-599)
                fb = self.frame_of[c]
                if fb == fa or fb in busy:
                    continue
                e = tables[self.host_of_frame[fb], a, fb]
                if e[0] != 1.0:
                    raise RuntimeError("no verification result for camera %d against frame %d" % (a, fb))
                if e[1] < 0:
                    continue
                rec = e[2:].reshape(4, 4)
                if self.full_refine:
                    accepted, T = self._refine(k, a, fb, rec, blocks[a][1], blocks[a][2])
                    self.refinements.append((k, a, fb, accepted))
                    if not accepted:
                        continue
                else:
                    T = self.be.relative_transform(rec, blocks[a][1])
                decided.append((fb, fa, T))
                busy.update((fa, fb))
                break
        # 6. merges
        for fb, fa, T in decided:
            self._merge(k, fb, fa, T)
        if decided:
            self.valid_from = k + 1
        return out

    def _refine(self, k, a, fb, recoveryPose, currPose, tick):
        """ReferenceFrame::resolveRelativeTransformationFern's second half for camera a against frame fb, on fb's rank; every rank
        returns the same (accepted, relativeTransform) - or raises, on EVERY rank, when any rank failed on the way: each rank's status
        rides the all-gather that carries the owner's result, the owner receives before anything that can fail, the sender sends its
        three messages whatever happened before them (csrc/session.hip refine)."""
        ha, hb = self.host_of_camera(a), self.host_of_frame[fb]
        res = np.zeros(18, np.float32)  # status (0 / -1) | accepted | relativeTransform
        failure = None
        if hb == self.rank:
            textures = None
            if ha != self.rank:  # the receives first
                textures = (self._recv(ha, np.uint8, (self.H, self.W, 4)), self._recv(ha, np.float32, (self.H, self.W, 4)),
                            self._recv(ha, np.float32, (self.H, self.W, 4)))
            try:
                owner = self.cams[next(c for c in self.hosted() if self.frame_of[c] == fb)]
                if fb not in self.refiners:
                    self.refiners[fb] = self.be.make_refiner()
                if ha == self.rank:
                    r = self.refiners[fb].refineLocal(owner, self.cams[a], recoveryPose, self.thresholds)
                else:
                    r = self.refiners[fb].refineRemote(owner, textures, a, tick, currPose, recoveryPose, self.thresholds)
                res[1] = 1.0 if r[0] else 0.0
                res[2:] = np.asarray(r[1], np.float32).reshape(16)
            except Exception as e:  # noqa: BLE001 (re-raised below, on every rank)
                failure = e
        elif ha == self.rank:
            try:
                tex = self.cams[a].fillTextures()
            except Exception as e:  # noqa: BLE001
                failure = e
                tex = (np.zeros((self.H, self.W, 4), np.uint8), np.zeros((self.H, self.W, 4), np.float32), np.zeros((self.H, self.W, 4), np.float32))
            for t in tex:
                self._send(t, hb)
        if self.world > 1:
            res[0] = -1.0 if failure is not None else 0.0
            all_res = self._allgather(res.view(np.uint8)).view(np.float32).reshape(self.world, 18)
            bad = [r for r in range(self.world) if all_res[r, 0] < 0]
            if bad:
                raise RuntimeError("inter-map refinement of camera %d against frame %d failed on rank(s) %s at tick %d" % (a, fb, bad, k)) from failure
            res = all_res[hb]
        elif failure is not None:
            raise failure
        return bool(res[1] == 1.0), res[2:].reshape(4, 4).copy()

    def _allgather(self, local_u8):
        if self.world == 1:
            return np.ascontiguousarray(local_u8).reshape(1, -1).copy()
        l = self._t(np.ascontiguousarray(local_u8))
        g = torch.zeros((self.world, l.numel()), dtype=torch.uint8, device=self.device)
        try:
            dist.all_gather_into_tensor(g.view(-1), l)
        except (RuntimeError, NotImplementedError):
            dist.all_gather([g[r] for r in range(self.world)], l)
        return g.cpu().numpy()

    def _rebase(self, c, T):
        self.pose_graph[c] = [(t, self.be.pose_compose(T, p)) for t, p in self.pose_graph[c]]
        self.relative_cons[c] = [np.concatenate([self.be.transform_point(T, r[:3]), self.be.transform_point(T, r[3:6])]).astype(np.float32)
                                 for r in self.relative_cons[c]]

    def _merge(self, k, fb, fa, T):
        """reference frame fb consumes fa (ReferenceFrame::consumeReferenceFrame)"""
        hb, ha = self.host_of_frame[fb], self.host_of_frame[fa]
        moving = [c for c in range(self.n) if self.frame_of[c] == fa]
        if hb == ha:
            if hb == self.rank:
                owner = self.cams[next(c for c in self.hosted() if self.frame_of[c] == fb)]
                founder_first = sorted(moving, key=lambda c: (c != fa, c))  # the camera that owns fa's map carries it over
                for c in founder_first:
                    self.cams[c].joinMap(owner, T)
                    self._rebase(c, T)
                self.ferns[fb].consume(self.ferns.pop(fa), T, self.fern_threshold)
        elif ha == self.rank:  # the consumed side: ship everything, free the local copies
            founder = self.cams[fa]
            self._send(founder.exportMap(), hb)
            self._send(self.ferns[fa].exportRecords(), hb)
            for c in moving:
                cam = self.cams[c]
                hdr = np.zeros(20, np.float32)
                hdr[:2] = np.array([cam.tick(), len(self.pose_graph[c])], np.int32).view(np.float32)
                hdr[2:3] = np.array([len(self.relative_cons[c])], np.int32).view(np.float32)
                hdr[4:20] = cam.pose().reshape(16)
                self._send(hdr, hb)
                self._send(self.last_frame[c][0], hb)
                self._send(self.last_frame[c][1], hb)
                pg = np.zeros((len(self.pose_graph[c]), 17), np.float32)
                for i, (t, p) in enumerate(self.pose_graph[c]):
                    pg[i, 0] = np.array([t], np.int32).view(np.float32)[0]
                    pg[i, 1:] = np.asarray(p, np.float32).reshape(16)
                self._send(pg, hb)
                self._send(np.asarray(self.relative_cons[c], np.float32).reshape(-1, 6), hb)
            for c in moving:
                self.cams.pop(c).close()
                self.pose_graph[c], self.relative_cons[c] = [], []
                self.last_frame.pop(c, None)
            self.ferns.pop(fa).close()
        elif hb == self.rank:  # the consuming side
            owner = self.cams[next(c for c in self.hosted() if self.frame_of[c] == fb)]
            owner.consumeRecords(self._recv(ha, np.float32).reshape(-1, 20), T)
            self.ferns[fb].consumeRecords(self._recv(ha, np.uint8), T, self.fern_threshold)
            for c in moving:
                hdr = self._recv(ha, np.float32)
                tick, n_pg = (int(v) for v in hdr[:2].view(np.int32))
                pose = hdr[4:20].reshape(4, 4)
                rgb = self._recv(ha, np.uint8, (self.H, self.W, 3))
                depth = self._recv(ha, np.uint16, (self.H, self.W))
                pg = self._recv(ha, np.float32).reshape(-1, 17)
                rc = self._recv(ha, np.float32).reshape(-1, 6)
                cam = self.be.make_camera(c)
                cam.importCamera(owner, self.be.pose_compose(T, pose), tick, rgb, depth)
                self.cams[c] = cam
                self.pose_graph[c] = [(int(r[:1].view(np.int32)[0]), r[1:].reshape(4, 4).copy()) for r in pg]
                self.relative_cons[c] = [r.copy() for r in rc]
                self.last_frame[c] = (rgb, depth)
                self._rebase(c, T)
        for c in moving:
            self.frame_of[c] = fb
        if fa in self.refiners:  # (the consumed reference frame is erased, ElasticFusion.cpp:610-616)
            self.refiners.pop(fa).close()
        del self.host_of_frame[fa]
        self.merges.append((k, fb, fa, np.asarray(T, np.float32).copy()))

    def close(self):
        for c in list(self.cams):
            if self.frame_of[c] != c:  # joined cameras first: their owner must outlive them
                self.cams.pop(c).close()
        for c in list(self.cams):
            self.cams.pop(c).close()
        for f in list(self.ferns):
            self.ferns.pop(f).close()
        for f in list(self.refiners):
            self.refiners.pop(f).close()


# ---- the product behind the session's call surface ------------------------------------------------------------------------------
class _GpuCamera:
    def __init__(self, be, c):
        from . import fusion

        self.be, self.c = be, c
        self.ef = fusion.ElasticFusion(be.W, be.H, be.K, timeIdx=c, num_sensors=be.num_sensors, **be.opts)
        self._last = None
        self._block = torch.zeros(be.block_bytes(), dtype=torch.uint8, device=be.device)
        self._tick, self._pose = 1, np.eye(4, dtype=np.float32)

    def processFrame(self, rgb, depth):
        r = self.ef.processFrame(rgb, depth)
        self._last = r
        self._tick, self._pose = int(r.tick), np.array(r.pose, np.float32).reshape(4, 4)
        return r

    def tick(self):
        return self._tick

    def pose(self):
        return self._pose

    def lost(self):
        return bool(self._last.lost) if self._last is not None else False

    def thumbnails(self):
        self.ef.thumbnails(self._block.data_ptr(), None)
        torch.cuda.synchronize(self.be.device)
        return self._block.cpu().numpy()

    def fillTextures(self):
        """(image u8 HxWx4, vertex f32 HxWx4, normal f32 HxWx4) of the last frame's fill-in, as host arrays for the transport"""
        return self.ef.image(13), self.ef.image(14), self.ef.image(15)

    def joinMap(self, owner, T):
        self.ef.joinMap(owner.ef, T)
        self._pose = self.be.pose_compose(T, self._pose)

    def exportMap(self):
        return self.ef.globalModel().downloadMap().view(np.float32).reshape(-1, 20)

    def consumeRecords(self, rec, T):
        if len(rec):
            t = torch.from_numpy(np.ascontiguousarray(rec, np.float32)).to(self.be.device)
            self.ef.globalModel().consumeRecords(t.data_ptr(), len(rec), T)

    def importCamera(self, owner, pose, tick, rgb, depth):
        r = torch.from_numpy(np.ascontiguousarray(rgb)).to(self.be.device)
        d = torch.from_numpy(np.ascontiguousarray(depth).view(np.int16)).to(self.be.device)
        self.ef.importCamera(owner.ef, pose, tick, r.data_ptr(), 3, d.data_ptr())
        self._tick, self._pose = int(tick), np.asarray(pose, np.float32).reshape(4, 4).copy()

    def model(self):
        return self.ef.globalModel().downloadMap()

    def close(self):
        self.ef.close()


class _GpuRefiner:
    """dms_refframe (ReferenceFrame's m_index + m_rgbd) behind the session's call surface"""

    def __init__(self, be):
        from . import fusion

        self.be = be
        self.rf = fusion.ReferenceFrameRefiner(be.W, be.H, be.K)

    def _result(self, r):
        return bool(r.accepted), np.array(r.relativeTransform, np.float32).reshape(4, 4)

    def refineLocal(self, owner, cam, recoveryPose, thresholds):
        e = cam.ef
        return self._result(self.rf.refine(owner.ef, recoveryPose, cam.pose(), e.imagePtr(14), e.imagePtr(15), e.imagePtr(13), cam.c, cam.tick(),
                                           *thresholds))

    def refineRemote(self, owner, textures, timeIdx, tick, currPose, recoveryPose, thresholds):
        img, vtx, nrm = (torch.from_numpy(np.ascontiguousarray(t)).to(self.be.device) for t in textures)
        r = self.rf.refine(owner.ef, recoveryPose, currPose, vtx.data_ptr(), nrm.data_ptr(), img.data_ptr(), timeIdx, tick, *thresholds)
        torch.cuda.synchronize(self.be.device)
        return self._result(r)

    def close(self):
        self.rf.close()


class _GpuFerns:
    def __init__(self, be):
        from . import ferns

        self.be = be
        self.db = ferns.Ferns(be.W, be.H, be.K, **be.fern_opts)

    def _dev(self, blk):
        return torch.from_numpy(np.ascontiguousarray(blk)).to(self.be.device)

    def addBlock(self, blk, pose, tick, thr):
        t = self._dev(blk)
        p = torch.from_numpy(np.ascontiguousarray(pose, np.float32).reshape(16)).to(self.be.device)
        self.db.addFrameAsync(t.data_ptr(), p.data_ptr(), int(tick), thr, None)
        torch.cuda.synchronize(self.be.device)

    def findFrameThumbs(self, blk, pose, tick, inter_map=1):
        t = self._dev(blk)
        m, _ = self.db.findFrameThumbs(t.data_ptr(), np.ascontiguousarray(pose, np.float32), int(tick), False, int(inter_map), None)
        return int(m.closest), int(m.candidate), np.array(m.estPose, np.float32).reshape(4, 4)

    def searchHit(self, blk):
        """the first half of findFrame up to `blockHDAware > 0.3` (Ferns.cpp:327-342) for one thumbnail block"""
        T = self.be.block_bytes()
        t = torch.zeros(T + 1024, dtype=torch.uint8, device=self.be.device)
        t[:T] = self._dev(blk)
        rows = torch.zeros(4, dtype=torch.int32, device=self.be.device)
        self.db.encodeThumbs(t.data_ptr(), t.data_ptr() + T, t.data_ptr() + T + 512, None)
        self.db.searchBlocksHd(t.data_ptr(), T + 1024, 1, T, T + 512, 0, True, rows.data_ptr(), None)
        r = rows.cpu().numpy()
        return bool(r[0] >= 0 and float(np.float32(r[3]) / np.float32(r[2])) > 0.3)  # Ferns.cpp:346: float ratio against the double literal

    def consume(self, other, T, thr):
        self.db.consume(other.db, T, thr)

    def exportRecords(self):
        n, rb = len(self.db), self.db.recordBytes()
        buf = torch.zeros(max(n, 1) * rb, dtype=torch.uint8, device=self.be.device)
        got = self.db.exportRecords(buf.data_ptr(), n)
        return buf[:got * rb].cpu().numpy()

    def consumeRecords(self, raw, T, thr):
        rb = self.db.recordBytes()
        if len(raw):
            t = self._dev(raw)
            self.db.consumeRecords(t.data_ptr(), len(raw) // rb, T, thr)

    def __len__(self):
        return len(self.db)

    def close(self):
        self.db.close()


class GpuBackend:
    """fusion.ElasticFusion + ferns.Ferns (HBM) behind CollabSession."""

    def __init__(self, width, height, K, device, num_sensors=None, fern_opts=None, **opts):
        from . import collab

        # None: max(3, cameras of the session) - set by CollabSession, as dms_session_default_params does (the reference's NUM_CAMERAS
        # is 3, Shaders/Vertex.cpp:49; the clean's health test walks num_sensors slots, so the count is part of the result)
        self.W, self.H, self.K, self.device, self.num_sensors, self.opts = width, height, K, device, num_sensors, opts
        self.fern_opts = dict(num=500, maxDepth_mm=3000, photoThresh=115.0, seed=20260929, capacity=1024)
        self.fern_opts.update(fern_opts or {})
        self._bb = collab.thumbnail_bytes(width, height)

    def block_bytes(self):
        return self._bb

    def make_camera(self, c):
        return _GpuCamera(self, c)

    def make_ferns(self):
        return _GpuFerns(self)

    def make_refiner(self):
        return _GpuRefiner(self)

    def relative_transform(self, recoveryPose, currPose):
        from . import fusion

        return fusion.relative_transform(recoveryPose, currPose)

    def pose_compose(self, a, b):
        from . import fusion

        return fusion.pose_compose(a, b)

    def transform_point(self, T, p):
        T = np.asarray(T, np.float32).reshape(4, 4)
        o = np.zeros(3, np.float32)
        for i in range(3):  # ((T0 x + T1 y) + T2 z) + T3, every operation rounded
            s = np.float32(T[i, 0] * p[0])
            s = np.float32(s + np.float32(T[i, 1] * p[1]))
            s = np.float32(s + np.float32(T[i, 2] * p[2]))
            o[i] = np.float32(s + T[i, 3])
        return o


# ---- the session behind the C ABI (include/dmslam_session.h, csrc/session.hip) -----------------------------------------------------
import ctypes as _C  # noqa: E402


def _native():
    from . import capi, fusion

    lib = capi.lib
    if getattr(lib, "_dms_session_bound", False):
        return lib, capi, fusion

    class SessionParams(_C.Structure):
        _fields_ = [("n_cameras", _C.c_int), ("camera", fusion.FusionParams), ("fern_num", _C.c_int), ("fern_max_depth_mm", _C.c_int),
                    ("fern_capacity", _C.c_int), ("fern_photo_thresh", _C.c_float), ("fern_seed", _C.c_uint), ("fern_threshold", _C.c_float),
                    ("inter_map", _C.c_int), ("query_from", _C.c_int), ("full_refine", _C.c_int), ("cov_thresh", _C.c_float),
                    ("icp_err_thresh", _C.c_float), ("icp_count_thresh", _C.c_float), ("query_inside_frame", _C.c_int), ("time_exchange", _C.c_int)]

    FN_AG = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_size_t, _C.c_void_p)
    FN_SR = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_void_p, _C.c_size_t, _C.c_int, _C.c_void_p)

    class Transport(_C.Structure):
        _fields_ = [("ctx", _C.c_void_p), ("rank", _C.c_int), ("world", _C.c_int), ("allgather", FN_AG), ("send", FN_SR), ("recv", FN_SR),
                    ("broadcast", FN_SR)]

    P = _C.c_void_p
    lib.dms_session_default_params.argtypes = [_C.POINTER(SessionParams), _C.c_int, _C.c_int, _C.c_int, _C.c_float, _C.c_float, _C.c_float, _C.c_float]
    lib.dms_session_default_params.restype = None
    lib.dms_session_create.argtypes = [_C.POINTER(P), _C.POINTER(SessionParams), _C.POINTER(Transport)]
    lib.dms_session_destroy.argtypes = [P]
    lib.dms_session_step.argtypes = [P, _C.c_int, _C.POINTER(P), _C.POINTER(P), P]
    lib.dms_session_step_async.argtypes = [P, _C.c_int, _C.POINTER(P), _C.POINTER(P), P]
    lib.dms_session_sync.argtypes = [P]
    lib.dms_session_async_stats.argtypes = [P, _C.POINTER(_C.c_int), _C.POINTER(_C.c_int)]
    lib.dms_session_exchange_time.argtypes = [P, _C.POINTER(_C.c_double), _C.POINTER(_C.c_int)]
    lib.dms_session_frame_of.argtypes = [P, _C.POINTER(_C.c_int)]
    lib.dms_session_host_of_frame.argtypes = [P, _C.c_int]
    lib.dms_session_num_merges.argtypes = [P]
    lib.dms_session_get_merge.argtypes = [P, _C.c_int, _C.POINTER(_C.c_int), _C.POINTER(_C.c_int), _C.POINTER(_C.c_int), _C.POINTER(_C.c_float)]
    lib.dms_session_num_refinements.argtypes = [P]
    lib.dms_session_get_refinement.argtypes = [P, _C.c_int] + [_C.POINTER(_C.c_int)] * 4
    lib.dms_session_hosted.argtypes = [P, _C.POINTER(_C.c_int), _C.c_int, _C.POINTER(_C.c_int)]
    lib.dms_session_camera.argtypes = [P, _C.c_int]
    lib.dms_session_camera.restype = P
    lib.dms_session_ferns.argtypes = [P, _C.c_int]
    lib.dms_session_ferns.restype = P
    lib.dms_session_last_result.argtypes = [P, _C.c_int, _C.POINTER(fusion.FrameResult)]
    lib.dms_session_pose_graph.argtypes = [P, _C.c_int, _C.POINTER(_C.c_int), _C.POINTER(_C.c_float), _C.c_int, _C.POINTER(_C.c_int)]
    lib.dms_session_add_relative_constraint.argtypes = [P, _C.c_int, _C.POINTER(_C.c_float), _C.POINTER(_C.c_float)]
    lib.dms_session_relative_constraints.argtypes = [P, _C.c_int, _C.POINTER(_C.c_float), _C.c_int, _C.POINTER(_C.c_int)]
    lib.dms_transport_rccl.argtypes = [P, _C.POINTER(Transport)]
    lib.dms_ferns_num_frames.argtypes = [P]
    lib._dms_session_types = (SessionParams, Transport, FN_AG, FN_SR)
    lib._dms_session_bound = True
    return lib, capi, fusion


class TorchTransport:
    """A dms_transport over torch.distributed for process groups whose tensors live on the host (gloo): device bytes are staged
    through host arrays around every call.  What the tests use to run the compiled session with two ranks on a one-GPU box; on the
    node itself the transport is RCCL (dms_transport_rccl)."""

    def __init__(self, rank, world):
        lib, capi, _ = _native()
        _, Transport, FN_AG, FN_SR = lib._dms_session_types
        self.lib, self.rank, self.world = lib, rank, world

        def d2h(ptr, n):
            a = np.empty(n, np.uint8)
            capi.check(lib.dms_memcpy_d2h(a.ctypes.data_as(_C.c_void_p), _C.c_void_p(ptr), _C.c_size_t(n), None), "dms_memcpy_d2h")
            return a

        def h2d(ptr, a):
            a = np.ascontiguousarray(a)
            capi.check(lib.dms_memcpy_h2d(_C.c_void_p(ptr), a.ctypes.data_as(_C.c_void_p), _C.c_size_t(a.nbytes), None), "dms_memcpy_h2d")

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:  # noqa: BLE001 (a Python exception must not unwind through the C caller)
                    import traceback

                    traceback.print_exc()
                    self.error = e
                    return -9
            return wrapped

        def allgather(ctx, send, recv, n, st):
            lib.dms_stream_sync(st)
            l = torch.from_numpy(d2h(send, n))
            g = [torch.empty(n, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(g, l)
            h2d(recv, torch.cat(g).numpy())

        def send(ctx, src, n, peer, st):
            lib.dms_stream_sync(st)
            dist.send(torch.from_numpy(d2h(src, n)), peer)

        def recv(ctx, dst, n, peer, st):
            t = torch.empty(n, dtype=torch.uint8)
            dist.recv(t, peer)
            h2d(dst, t.numpy())

        def broadcast(ctx, buf, n, root, st):
            lib.dms_stream_sync(st)
            t = torch.from_numpy(d2h(buf, n))
            dist.broadcast(t, root)
            h2d(buf, t.numpy())

        self.error = None
        self._keep = (FN_AG(guard(allgather)), FN_SR(guard(send)), FN_SR(guard(recv)), FN_SR(guard(broadcast)))
        self.struct = Transport(None, rank, world, *self._keep)


class RcclTransport:
    """dms_transport_rccl over a communicator of include/dmslam_collab.h (collab.RcclCarrier): what the session uses on the node -
    device buffers, stream-ordered RCCL calls, nothing staged through the host."""

    def __init__(self, carrier):
        lib, capi, _ = _native()
        Transport = lib._dms_session_types[1]
        self.carrier = carrier  # (keeps the communicator alive)
        self.struct = Transport()
        capi.check(lib.dms_transport_rccl(carrier.h, _C.byref(self.struct)), "dms_transport_rccl")
        self.rank, self.world = int(self.struct.rank), int(self.struct.world)


class _NativeCamera:
    def __init__(self, s, c):
        self.s, self.c = s, c

    def model(self):
        lib, _, fusion = _native()
        h = lib.dms_session_camera(self.s.h, self.c)
        return fusion.GlobalModel(self.s.W, self.s.H, handle=lib.dms_fusion_model(h)).downloadMap()


class _NativeFerns:
    def __init__(self, s, f):
        self.s, self.f = s, f

    def __len__(self):
        lib = _native()[0]
        return int(lib.dms_ferns_num_frames(lib.dms_session_ferns(self.s.h, self.f)))


class NativeSession:
    """dms_session (include/dmslam_session.h): CollabSession's protocol compiled into the library, the product's engines behind it.
    The same read-only surface as CollabSession (merges, refinements, frame_of, hosted(), cams[c].model(), len(ferns[f]),
    pose_graph[c]), so that the session tests run either."""

    def __init__(self, width, height, K, n_cameras, rank=0, world=1, transport=None, fern_photo_thresh=115.0, fern_capacity=1024,
                 fern_threshold=0.3095, inter_map=1, query_from=0, full_refine=True, cov_thresh=1e-05, icp_err_thresh=2e-05,
                 icp_count_thresh=35000, query_inside_frame=False, time_exchange=False, **camera_opts):
        lib, capi, _ = _native()
        SessionParams = lib._dms_session_types[0]
        p = SessionParams()
        lib.dms_session_default_params(_C.byref(p), n_cameras, width, height, K[0], K[1], K[2], K[3])
        for k, v in camera_opts.items():
            if not hasattr(p.camera, k):
                raise TypeError("unknown camera option %r" % k)
            setattr(p.camera, k, v)
        p.fern_photo_thresh, p.fern_capacity, p.fern_threshold = fern_photo_thresh, fern_capacity, fern_threshold
        p.inter_map, p.query_from, p.full_refine = inter_map, query_from, 1 if full_refine else 0
        p.cov_thresh, p.icp_err_thresh, p.icp_count_thresh = cov_thresh, icp_err_thresh, icp_count_thresh
        p.query_inside_frame, p.time_exchange = 1 if query_inside_frame else 0, 1 if time_exchange else 0
        self.W, self.H, self.n, self.rank, self.world = width, height, n_cameras, rank, world
        self.transport = transport
        h = _C.c_void_p()
        capi.check(lib.dms_session_create(_C.byref(h), _C.byref(p), _C.byref(transport.struct) if transport is not None else None), "dms_session_create")
        self.h, self.lib, self.capi = h, lib, capi
        N = width * height
        self._read = [c for c in range(n_cameras) if c % world == rank]
        # three sets of frame buffers in turn: the pipelined step reads a tick's frames until the step two ticks later has returned
        self._rgb = [[capi.DeviceBuffer(N * 3) for _ in self._read] for _ in range(3)]
        self._dep = [[capi.DeviceBuffer(N * 2) for _ in self._read] for _ in range(3)]

    def step(self, k, my_frames, pipelined=False, stream=None):
        """my_frames: {camera id: (rgb u8 HxWx3, depth u16 HxW)} for the cameras READ on this rank.
        pipelined: dms_session_step_async (no host synchronisation; the full query only on a descriptor hit, three ticks late)."""
        b = int(k) % 3
        for i, c in enumerate(self._read):
            r, d = my_frames[c]
            self._rgb[b][i].upload(np.ascontiguousarray(r, np.uint8))
            self._dep[b][i].upload(np.ascontiguousarray(d, np.uint16))
        self.step_resident(k, [x.ptr for x in self._rgb[b]], [x.ptr for x in self._dep[b]], pipelined, stream)

    def step_resident(self, k, rgb_ptrs, depth_ptrs, pipelined=False, stream=None):
        """the same for frames that already lie in this device's HBM (one pointer per camera read here, ascending)"""
        P = _C.c_void_p
        rgb, dep = (P * len(self._read))(*rgb_ptrs), (P * len(self._read))(*depth_ptrs)
        fn = self.lib.dms_session_step_async if pipelined else self.lib.dms_session_step
        self.capi.check(fn(self.h, int(k), rgb, dep, P(stream) if stream else None), "dms_session_step_async" if pipelined else "dms_session_step")

    def sync(self):
        self.capi.check(self.lib.dms_session_sync(self.h), "dms_session_sync")

    def async_stats(self):
        t, w = _C.c_int(0), _C.c_int(0)
        self.capi.check(self.lib.dms_session_async_stats(self.h, _C.byref(t), _C.byref(w)))
        return {"ticks": t.value, "woken": w.value}

    def exchange_time(self):
        """(sum of the timed all-gathers' device time in ms, how many) - needs time_exchange=True"""
        ms, n = _C.c_double(0.0), _C.c_int(0)
        self.capi.check(self.lib.dms_session_exchange_time(self.h, _C.byref(ms), _C.byref(n)))
        return ms.value, n.value

    # -- the read-only surface of CollabSession ---------------------------------------------------------------------------------
    @property
    def frame_of(self):
        a = (_C.c_int * self.n)()
        self.capi.check(self.lib.dms_session_frame_of(self.h, a))
        return list(a)

    def hosted(self):
        a, n = (_C.c_int * self.n)(), _C.c_int(0)
        self.capi.check(self.lib.dms_session_hosted(self.h, a, self.n, _C.byref(n)))
        return [int(a[i]) for i in range(n.value)]

    @property
    def merges(self):
        out = []
        for i in range(self.lib.dms_session_num_merges(self.h)):
            k, fb, fa, T = _C.c_int(0), _C.c_int(0), _C.c_int(0), (_C.c_float * 16)()
            self.capi.check(self.lib.dms_session_get_merge(self.h, i, _C.byref(k), _C.byref(fb), _C.byref(fa), T))
            out.append((k.value, fb.value, fa.value, np.array(T, np.float32).reshape(4, 4)))
        return out

    @property
    def refinements(self):
        out = []
        for i in range(self.lib.dms_session_num_refinements(self.h)):
            v = [_C.c_int(0) for _ in range(4)]
            self.capi.check(self.lib.dms_session_get_refinement(self.h, i, *[_C.byref(x) for x in v]))
            out.append((v[0].value, v[1].value, v[2].value, bool(v[3].value)))
        return out

    @property
    def cams(self):
        return {c: _NativeCamera(self, c) for c in self.hosted()}

    @property
    def ferns(self):
        return {f: _NativeFerns(self, f) for f in sorted(set(self.frame_of)) if self.lib.dms_session_ferns(self.h, f)}

    @property
    def pose_graph(self):
        out = {}
        for c in self.hosted():
            n = _C.c_int(0)
            self.capi.check(self.lib.dms_session_pose_graph(self.h, c, None, None, 0, _C.byref(n)))
            t, p = (_C.c_int * max(n.value, 1))(), (_C.c_float * (16 * max(n.value, 1)))()
            self.capi.check(self.lib.dms_session_pose_graph(self.h, c, t, p, n.value, _C.byref(n)))
            P = np.array(p, np.float32).reshape(-1, 4, 4)
            out[c] = [(int(t[i]), P[i].copy()) for i in range(n.value)]
        return out

    def addRelativeConstraint(self, camera, src, target):
        a, b = (_C.c_float * 3)(*[float(v) for v in src]), (_C.c_float * 3)(*[float(v) for v in target])
        self.capi.check(self.lib.dms_session_add_relative_constraint(self.h, int(camera), a, b), "dms_session_add_relative_constraint")

    @property
    def relative_cons(self):
        out = {}
        for c in self.hosted():
            n = _C.c_int(0)
            self.capi.check(self.lib.dms_session_relative_constraints(self.h, c, None, 0, _C.byref(n)))
            r = (_C.c_float * (6 * max(n.value, 1)))()
            self.capi.check(self.lib.dms_session_relative_constraints(self.h, c, r, n.value, _C.byref(n)))
            out[c] = [np.array(r, np.float32).reshape(-1, 6)[i].copy() for i in range(n.value)]
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.dms_session_destroy(self.h)
            self.h = None
