"""Collaborative-session plumbing: one camera (and its own surfel map) per GPU / rank.

The reference runs its cameras one after the other in one process on one GPU
(GUI/src/MainController.cpp:262-400) and has its inter-map matching compiled out
(Core/src/ElasticFusion.cpp:597).  Here every rank owns the Context + ReferenceFrame of its
cameras; nothing is shared until a map merge, so the data path needs no collective.  The only
per-frame exchange is the all-gather of each camera's frame block: its fern descriptor (the 500 codes
of the shared fern table, good-code count, tick, pose: 592 bytes — what the inter-map search needs) and
its W/8 x H/8 thumbnails (image, vertex and normal maps — what the verification of a candidate needs,
Core/src/Ferns.cpp:277-423) — SURVEY.md §8(e).  `InterMapMatcher` is the consumer: every rank searches
its own fern database with every other camera's descriptor (device only, one frame behind the
all-gather) and verifies candidates with the thumbnail-sized tracker.
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests)
carries it; PyTorch is plumbing here, not the product.

A map merge (GlobalModel::consume) is the one bulk transfer: the consumed map travels
once, point to point, as packed 80-byte surfel records (`send_map` / `recv_map`): 2 M surfels =
160 MB, about 1 ms on one xGMI link; no collective is involved.
"""
import torch
import torch.distributed as dist

THUMB_BYTES_PER_PIXEL = 4 + 16 + 16  # RGBA8 image + RGBA32F vertex + RGBA32F normal


# fern descriptor appended to the thumbnails: codes[512] | goodCodes i32 | tick i32 | pose 16 x f32 | pad
DESC_CODES, DESC_GOOD, DESC_TICK, DESC_POSE, DESC_BYTES = 0, 512, 516, 520, 592


def thumbnail_bytes(width, height):
    """dms_thumb_block_bytes: n = (W // 8)(H // 8) pixels, [RGBA8 image padded to 16 bytes | RGBA32F vertex | RGBA32F normal] - n * 36 when
    n is a multiple of 4"""
    n = (width // 8) * (height // 8)
    return ((n * 4 + 15) & ~15) + n * 32


def thumbnail_offsets(width, height):
    """(vertex offset, normal offset) inside a thumbnail block"""
    n = (width // 8) * (height // 8)
    v = (n * 4 + 15) & ~15
    return v, v + n * 16


def shard_cameras(n_cameras, rank, world):
    """Camera ids owned by `rank`: round-robin, so n_cameras == world gives one camera per GPU."""
    return [c for c in range(n_cameras) if c % world == rank]


class _EventWork:
    """the two methods ThumbnailExchange uses of a torch.distributed work handle, over a stream event"""

    def __init__(self, ev):
        self.ev = ev

    def is_completed(self):
        return self.ev.query()

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class RcclCarrier:
    """ctypes mirror of include/dmslam_collab.h: the exchange step through the library's own RCCL binding — what a C++ front
    end calls — instead of torch.distributed.  `create` is collective; the 128-byte id travels over whatever rendezvous the
    caller has (here: a torch.distributed broadcast on the CPU group, or none for a single rank)."""

    def __init__(self, rank, world, unique_id):
        import ctypes as C

        from . import capi

        self._C, self._capi = C, capi
        lib = capi.lib
        lib.dms_collab_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_char_p]
        lib.dms_collab_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dms_collab_send.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        lib.dms_collab_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        lib.dms_collab_allreduce_max_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dms_collab_destroy.argtypes = [C.c_void_p]
        lib.dms_collab_rank.argtypes = [C.c_void_p]
        lib.dms_collab_size.argtypes = [C.c_void_p]
        lib.dms_collab_library_path.restype = C.c_char_p
        self.h = C.c_void_p()
        capi.check(lib.dms_collab_create(C.byref(self.h), rank, world, bytes(unique_id)), "dms_collab_create")
        self.rank, self.world = lib.dms_collab_rank(self.h), lib.dms_collab_size(self.h)

    @staticmethod
    def unique_id():
        import ctypes as C

        from . import capi

        buf = C.create_string_buffer(128)
        capi.check(capi.lib.dms_collab_unique_id(buf), "dms_collab_unique_id")
        return buf.raw

    def allgather(self, send, recv, stream=None):
        """send: (n,) uint8 device tensor; recv: (world * n,) — on `stream` (a torch stream or None = the current one)"""
        s = torch.cuda.current_stream() if stream is None else stream
        self._capi.check(self._capi.lib.dms_collab_allgather(self.h, send.data_ptr(), recv.data_ptr(), send.numel() * send.element_size(),
                                                             self._C.c_void_p(s.cuda_stream)), "dms_collab_allgather")

    def send(self, t, peer, stream=None):
        s = torch.cuda.current_stream() if stream is None else stream
        self._capi.check(self._capi.lib.dms_collab_send(self.h, t.data_ptr(), t.numel() * t.element_size(), peer, self._C.c_void_p(s.cuda_stream)),
                         "dms_collab_send")

    def recv(self, t, peer, stream=None):
        s = torch.cuda.current_stream() if stream is None else stream
        self._capi.check(self._capi.lib.dms_collab_recv(self.h, t.data_ptr(), t.numel() * t.element_size(), peer, self._C.c_void_p(s.cuda_stream)),
                         "dms_collab_recv")

    def max_f64(self, t, stream=None):
        s = torch.cuda.current_stream() if stream is None else stream
        self._capi.check(self._capi.lib.dms_collab_allreduce_max_f64(self.h, t.data_ptr(), self._C.c_void_p(s.cuda_stream)),
                         "dms_collab_allreduce_max_f64")

    def library_path(self):
        """the file the library resolved RCCL's entry points from (one copy per process: the one already mapped, if any)"""
        return (self._capi.lib.dms_collab_library_path() or b"").decode()

    def close(self):
        if self.h:
            self._capi.lib.dms_collab_destroy(self.h)
            self.h = None


def rccl_carrier_from_process_group(rank, world):
    """Collective: rank 0 draws the id, the default process group (any backend) carries its 128 bytes."""
    ident = [RcclCarrier.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(ident, src=0)
    return RcclCarrier(rank, world, ident[0])


class ThumbnailExchange:
    """Fixed-size per-frame all-gather of one camera's thumbnail block per rank.

    Synchronous use: fill `local`, call `gather()`.  Overlapped use (bench.py): `begin()` hands out
    this frame's block (two alternate, so the collective of frame t may still be in flight while frame
    t+1 fills the other one), `gather(overlap=True)` starts the all-gather without making the calling
    stream wait for it — the matcher that consumes it runs a frame later — and `finish()` waits for
    whatever is still in flight."""

    def __init__(self, world, width, height, device, extra_bytes=0, carrier=None):
        self.world = world
        self.carrier = carrier  # RcclCarrier: the all-gather through dms_collab_allgather instead of torch.distributed
        self.side = None
        self.thumb_bytes = thumbnail_bytes(width, height)
        self.nbytes = self.thumb_bytes + extra_bytes  # (thumb_bytes is a multiple of 16: the descriptor stays aligned)
        self.locals = [torch.zeros((self.nbytes,), dtype=torch.uint8, device=device) for _ in range(2)]
        self.gathereds = [torch.zeros((world, self.nbytes), dtype=torch.uint8, device=device) for _ in range(2)]
        self.work = [None, None]
        self.slot = 0
        self.local = self.locals[0]
        self.gathered = self.gathereds[0]

    def begin(self):
        w = self.work[self.slot]
        if w is not None:  # the collective that used this block two frames ago
            # normally long finished: a host-side query then avoids putting a wait on the frame's stream
            if not w.is_completed():
                w.wait()
            self.work[self.slot] = None
        self.local = self.locals[self.slot]
        return self.local

    def gather(self, overlap=False):
        g, l = self.gathereds[self.slot], self.locals[self.slot]
        self.gathered = g
        if self.world == 1 and self.carrier is None:
            g[0].copy_(l)
            return g
        if self.carrier is not None:
            # the library's collective is stream-ordered: it runs on a side stream behind the producer of the block, and the
            # event recorded after it is what begin() / finish() / the consumer wait for
            if self.side is None:
                self.side = torch.cuda.Stream()
            cur = torch.cuda.current_stream()
            self.side.wait_stream(cur)
            self.carrier.allgather(l, g.view(-1), self.side)
            ev = torch.cuda.Event()
            ev.record(self.side)
            if overlap:
                self.work[self.slot] = _EventWork(ev)
                self.slot ^= 1
            else:
                cur.wait_event(ev)
            return g
        if overlap:
            try:
                self.work[self.slot] = dist.all_gather_into_tensor(g.view(-1), l, async_op=True)
            except (RuntimeError, NotImplementedError):
                self.work[self.slot] = dist.all_gather([g[r] for r in range(self.world)], l, async_op=True)
            self.slot ^= 1
            return g
        try:
            dist.all_gather_into_tensor(g.view(-1), l)
        except (RuntimeError, NotImplementedError):  # backends without the flat form
            parts = [g[r] for r in range(self.world)]
            dist.all_gather(parts, l)
        return g

    def finish(self):
        for k in range(2):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


RECORD_FLOATS = 20  # pos4 col4 nrm4 times8: the dms_model_download / dms_model_export_records layout


def send_map(records, count, dst):
    """Point-to-point send of `count` surfel records (float32 tensor [>= count, 20]) to rank `dst`.
    The count goes first so the receiver can size its buffer; both messages use the tensor's device
    (RCCL p2p over xGMI for device tensors)."""
    hdr = torch.tensor([int(count)], dtype=torch.int64, device=records.device)
    dist.send(hdr, dst)
    if count:
        dist.send(records[:count].contiguous().view(-1), dst)


def recv_map(src, device):
    """Counterpart of send_map: returns (float32 tensor [count, 20] on `device`, count)."""
    hdr = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(hdr, src)
    count = int(hdr.item())
    rec = torch.empty((max(count, 1), RECORD_FLOATS), dtype=torch.float32, device=device)
    if count:
        dist.recv(rec[:count].view(-1), src)
    return rec, count


def merge_remote_map(model, src, relative_transform, device):
    """Receive the map rank `src` sends with send_map and append it to `model` (a fusion.GlobalModel)
    moved by the 4x4 `relative_transform` — the consuming side of ReferenceFrame::consumeReferenceFrame."""
    rec, count = recv_map(src, device)
    if count:
        torch.cuda.synchronize(device) if rec.is_cuda else None
        model.consumeRecords(rec.data_ptr(), count, relative_transform)
    return count



class InterMapMatcher:
    """Inter-map loop-closure search of collaborative mode (what ElasticFusion.cpp:595-632 does with every other map's
    fern database, compiled out there): owner computes.

    Per frame, on the frame's stream and without a host synchronisation:
      publish()  packs this camera's frame block — thumbnails of the fill-in textures, fern codes of the SHARED table,
                 tick and the pose read from HBM — offers the frame to the local database (Ferns::addFrame) and starts
                 the all-gather;
      match()    one frame later: searches the local database with every other camera's descriptor
                 (dms_ferns_search_codes; results land in a pinned array the host reads a frame later).
    Every `verify_interval` frames verify() runs the reference's full query (Ferns::findFrame with interMap = true: search,
    code agreement, 3 x 50 ICP iterations at thumbnail size, photometric check) on each remote camera's latest block; that
    call synchronises, like the reference's.  A verified match is what triggers a map merge (send_map / merge_remote_map)."""

    def __init__(self, ferns, exchange, rank, world, device, fern_threshold=0.3095, verify_interval=0, side_stream=False):
        self.ferns, self.x, self.rank, self.world, self.device = ferns, exchange, rank, world, device
        # side_stream: only the frame block itself (thumbnails, pose, tick: one launch) is taken on the frame's stream; the
        # descriptor, the key-frame database, the all-gather and the search go to a side stream.  Off by default: measured on
        # one MI355X (bench.py, DMS_BENCH_EXCHANGE=1) the third stream costs more than the ten small launches it takes off the
        # frame's stream (1856 against 1873 frames/s; 1963 without any exchange).
        self.side = torch.cuda.Stream(device) if (side_stream and device.type == "cuda") else None
        self.block_ready = torch.cuda.Event() if self.side is not None else None
        self.fern_threshold, self.verify_interval = fern_threshold, verify_interval
        self.best_dev = torch.full((world, 2), -1, dtype=torch.int32, device=device)
        # two host-visible result arrays used in turn, each with the event recorded behind the call that fills it: a result that is
        # still in flight when the next search is enqueued is counted a frame later instead of being overwritten
        mk = lambda: torch.full((world, 2), -1, dtype=torch.int32).pin_memory() if device.type == "cuda" else torch.full((world, 2), -1, dtype=torch.int32)
        self._hosts = [mk(), mk()]
        self._events = [None, None]
        self._turn = 0
        self.best_host = self._hosts[0]  # the array the most recent search call wrote (or will write) its results to
        self.prev = None  # (gathered tensor, work handle) of the previous publish
        self.frames = 0
        self.candidates = 0  # remote descriptors that found a candidate frame (from the host-visible results)
        self.verified = []   # (frame, remote rank, FernMatch) of accepted verifications

    def publish(self, ef, tick, stream):
        """`stream`: the raw handle of torch's current stream, on which the frame was enqueued"""
        blk = self.x.begin()
        base, T = blk.data_ptr(), self.x.thumb_bytes
        if hasattr(ef, "frameBlock"):  # thumbnails + pose from HBM + tick in one launch
            ef.frameBlock(base, base + T + DESC_POSE, base + T + DESC_TICK, int(tick), stream)
        else:
            ef.thumbnails(base, stream)
            blk[T + DESC_TICK:T + DESC_TICK + 4].view(torch.int32).fill_(int(tick))
            ef.exportPose(base + T + DESC_POSE, stream)  # 64 bytes, device to device, stream ordered
        if self.side is not None:
            self.block_ready.record(torch.cuda.current_stream(self.device))
            self.side.wait_event(self.block_ready)
            with torch.cuda.stream(self.side):
                return self._publish_rest(base, T, tick, self.side.cuda_stream)
        return self._publish_rest(base, T, tick, stream)

    def _publish_rest(self, base, T, tick, stream):
        if hasattr(self.ferns, "publishBlock"):  # descriptor + key-frame insertion: one encoding pass, no staging copy
            self.ferns.publishBlock(base, base + T + DESC_CODES, base + T + DESC_GOOD, base + T + DESC_POSE, int(tick), self.fern_threshold, stream)
        else:
            self.ferns.encodeThumbs(base, base + T + DESC_CODES, base + T + DESC_GOOD, stream)
            self.ferns.addFrameAsync(base, base + T + DESC_POSE, int(tick), self.fern_threshold, stream)  # (the block's copy of the pose)
        slot = self.x.slot
        g = self.x.gather(overlap=True)
        cur = (g, slot)
        prev, self.prev = self.prev, cur
        self.frames += 1
        return prev

    def match(self, prev, tick, stream):
        """search the local database with the descriptors gathered by the PREVIOUS publish"""
        if prev is None:
            return
        if self.side is not None:
            with torch.cuda.stream(self.side):
                return self._match(prev, tick, self.side.cuda_stream)
        return self._match(prev, tick, stream)

    def _match(self, prev, tick, stream):
        g, slot = prev
        w = self.x.work[slot]
        if w is not None:
            w.wait()  # (stream-side wait; the collective had a whole frame to finish)
        # Results arrive in the pinned arrays one call late (the device writes them); each array is counted once the event recorded
        # behind the call that fills it has completed, then cleared.  The array this call is about to reuse was filled two calls
        # ago: if even that is still in flight, wait for it (it is two frames old) rather than lose it.
        self._turn ^= 1
        cur = self._turn
        for k in (cur, cur ^ 1):
            ev = self._events[k]
            if ev is not None and k == cur and not ev.query():
                ev.synchronize()
            if ev is None or ev.query():
                self.candidates += int((self._hosts[k][:, 0] >= 0).sum())
                self._hosts[k].fill_(-1)
                self._events[k] = None
        self.best_host = self._hosts[cur]
        T = self.x.thumb_bytes
        if hasattr(self.ferns, "searchBlocks"):  # every remote descriptor in one launch
            # (the previous search's results go to the pinned array inside the same call: no copy engine on the frame's stream)
            self.ferns.searchBlocks(g.data_ptr(), self.x.nbytes, self.world, self.rank, T + DESC_CODES, T + DESC_GOOD, int(tick), True,
                                    self.best_dev.data_ptr(), stream, previous_out=self.best_host.data_ptr())
            if self.device.type == "cuda":
                self._events[cur] = torch.cuda.Event()
                self._events[cur].record(torch.cuda.current_stream(self.device))
            if self.verify_interval and self.frames % self.verify_interval == 0:
                self.verify(g, tick, stream)
            return
        else:
            for r in range(self.world):
                if r == self.rank:
                    continue
                p = g[r].data_ptr()
                self.ferns.searchCodes(p + T + DESC_CODES, p + T + DESC_GOOD, int(tick), True, self.best_dev[r].data_ptr(), stream)
        self.best_host.copy_(self.best_dev, non_blocking=True)
        if self.device.type == "cuda":
            self._events[cur] = torch.cuda.Event()
            self._events[cur].record(torch.cuda.current_stream(self.device))
        if self.verify_interval and self.frames % self.verify_interval == 0:
            self.verify(g, tick, stream)

    def verify(self, g, tick, stream):
        T = self.x.thumb_bytes
        for r in range(self.world):
            if r == self.rank:
                continue
            pose = g[r][T + DESC_POSE:T + DESC_POSE + 64].view(torch.float32).cpu().numpy()  # synchronises
            m, cons = self.ferns.findFrameThumbs(g[r].data_ptr(), pose, int(tick), False, True, stream)
            if m.closest >= 0:
                self.verified.append((self.frames, r, m))
