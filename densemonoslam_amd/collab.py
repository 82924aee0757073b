"""Collaborative-session plumbing: one camera (and its own surfel map) per GPU / rank.

The reference runs its cameras one after the other in one process on one GPU
(GUI/src/MainController.cpp:262-400) and has its inter-map matching compiled out
(Core/src/ElasticFusion.cpp:597).  Here every rank owns the Context + ReferenceFrame of its
cameras; nothing is shared until a map merge, so the data path needs no collective.  The only
per-frame exchange is the all-gather of each camera's W/8 x H/8 thumbnails (image, vertex and
normal maps), the inputs of the fern matcher (Core/src/Ferns.cpp:277-423) — SURVEY.md §8(e).
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests)
carries it; PyTorch is plumbing here, not the product.

A map merge (GlobalModel::consume) is the one bulk transfer: the consumed map travels
once, point to point, as packed 80-byte surfel records (`send_map` / `recv_map`): 2 M surfels =
160 MB, about 1 ms on one xGMI link; no collective is involved.
"""
import torch
import torch.distributed as dist

THUMB_BYTES_PER_PIXEL = 4 + 16 + 16  # RGBA8 image + RGBA32F vertex + RGBA32F normal


def thumbnail_bytes(width, height):
    return (width // 8) * (height // 8) * THUMB_BYTES_PER_PIXEL


def shard_cameras(n_cameras, rank, world):
    """Camera ids owned by `rank`: round-robin, so n_cameras == world gives one camera per GPU."""
    return [c for c in range(n_cameras) if c % world == rank]


class ThumbnailExchange:
    """Fixed-size per-frame all-gather of one camera's thumbnail block per rank.

    Synchronous use: fill `local`, call `gather()`.  Overlapped use (bench.py): `begin()` hands out
    this frame's block (two alternate, so the collective of frame t may still be in flight while frame
    t+1 fills the other one), `gather(overlap=True)` starts the all-gather without making the calling
    stream wait for it — the matcher that consumes it runs a frame later — and `finish()` waits for
    whatever is still in flight."""

    def __init__(self, world, width, height, device):
        self.world = world
        self.nbytes = thumbnail_bytes(width, height)
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
        if self.world == 1:
            g[0].copy_(l)
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

