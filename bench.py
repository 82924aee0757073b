#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the full ICP + RGB + surfel-fusion frame step at 640x480.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic RGB-D frame already resident in HBM:
depth bilateral + metric conversion, model prediction + fill-in, tracker pyramid
initialisation, SO3 pre-alignment + 3-level combined ICP/RGB Gauss-Newton (10/5/4 iterations),
second prediction, index map, fuse, index map, clean, final prediction
(ElasticFusion::processFrame with --o --nkf; SURVEY.md §8(d) "frame step").
Multi-GPU = the collaborative session: one camera (and its own map) per rank, weak scaling.  With
--gpus N > 1 the timed loop IS the compiled session's pipelined tick (dms_session_step_async,
include/dmslam_session.h) over the library's RCCL transport (dms_transport_rccl): per tick one
frame per rank, frame block (W/8 x H/8 thumbnails + fern descriptor), key-frame insertion, ONE
all-gather, descriptor search of every other camera's block, host mirror.
The round-3 exchange (collab.InterMapMatcher over torch.distributed) runs first as a labelled
fall-back leg: it becomes the headline only if the session loop cannot be set up or does not finish.
`--session-loop` runs the same loop at N = 1 (a one-rank RCCL communicator carries the collectives).

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
CLOCK_GHZ = 2.4        # shader clock (MI355X_MICROARCH.md)


def pmc_passes(args, W, H):
    """HBM traffic and vector-ALU activity of the level-0 tracker kernel from hardware counters of THIS build: three nested
    `rocprofv3 --kernel-trace --pmc ...` runs of this script (20 steps; FETCH_SIZE and WRITE_SIZE do not fit one pass,
    MI355X_MICROARCH.md "rocprofv3 PMC slots").  Returns None when rocprofv3 is not there or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None
    res = {}
    tmp = tempfile.mkdtemp(prefix="dms_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for name, counters, extra in (("fetch", ["FETCH_SIZE"], []), ("write", ["WRITE_SIZE"], []),
                                      ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES"], ["--no-pipeline"])):
            d = os.path.join(tmp, name)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", d, "-o", "r", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "20", "--warmup", "5", "--width", str(W), "--height", str(H), "--no-cpu-baseline",
                   "--no-kernel-pass", "--no-full-leg", "--no-config-legs", "--no-session-leg", "--no-pmc"] + extra
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
            if r.returncode != 0:
                return None
            vals, durs = {}, {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    kn = row["Kernel_Name"]
                    if "k_gn_level" not in kn:
                        continue
                    key = (row["Dispatch_Id"], kn, int(row["Grid_Size"]), int(row["Workgroup_Size"]))
                    per = vals.setdefault(row["Counter_Name"], {})
                    per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
                    durs[key] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
            if not vals:
                return None
            # level 0 = the instantiation the dominant launch runs: selected by kernel NAME (the resident ICP + RGB kernel with
            # the most pixels per thread), never by grid size alone — another configuration's level 0 has the same grid
            names = {k[1] for per in vals.values() for k in per}
            lvl0 = level0_kernel_name(names)
            if lvl0 is None:
                return None
            res["kernel"] = lvl0
            for c, per in vals.items():
                sel = [v for k, v in per.items() if k[1] == lvl0]
                res[c] = sum(sel) / len(sel)
                res["launches_" + name] = len(sel)
                k0 = next(k for k in per if k[1] == lvl0)
                res["blocks"] = k0[2] // max(1, k0[3])
                if name == "sq":
                    dd = [v for k, v in durs.items() if k[1] == lvl0]
                    res["launch_us_under_pmc"] = round(sum(dd) / len(dd), 2)
        if "FETCH_SIZE" not in res or "WRITE_SIZE" not in res:
            return None
        out = {
            # FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes: doubled
            # (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported (uncalibrated there)
            "hbm_bytes_per_launch": 2.0 * res["FETCH_SIZE"] * 1024.0 + res["WRITE_SIZE"] * 1024.0,
            "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, two nested 20-step passes of this run's build: "
                      "2 x %.0f KB fetched + %.0f KB written per level-0 launch" % (res["FETCH_SIZE"], res["WRITE_SIZE"]),
            "blocks": res.get("blocks"),
            "kernel": res.get("kernel"),
            "launches_averaged": res.get("launches_fetch"),
        }
        if "SQ_INSTS_VALU" in res and "launch_us_under_pmc" in res:
            out["valu_insts_per_launch"] = res["SQ_INSTS_VALU"]
            out["valu_active_quadcycles_per_launch"] = res["SQ_ACTIVE_INST_VALU"]
            out["launch_us_under_pmc"] = res["launch_us_under_pmc"]
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def level0_kernel_name(names):
    """The level-0 resident kernel among rocprofv3's kernel names: `k_gn_level<ICP = true, RGB = true, P, EXIT>` with the largest P
    (pixels per thread: 3 at 640x480, levels 1 / 2 run P = 1)."""
    import re

    best, best_p = None, -1
    for n in names:
        m = re.search(r"k_gn_level<\s*true,\s*true,\s*(\d+)", n)
        if m and int(m.group(1)) > best_p:
            best, best_p = n, int(m.group(1))
    return best


def algorithmic_bytes(kernel, W, H, M, level_px):
    """Algorithmic HBM bytes of ONE launch of `kernel` (SURVEY.md §8(d) contract)."""
    N0 = W * H
    if kernel == "gn_level":  # one GN iteration over a level: 62 + 14 B/px (76 * sum_l it_l * N_l)
        return 76.0 * level_px
    if kernel == "gn_pass1":  # ICP 48 B/px (4 SoA maps) + photometric count pass 14 B/px
        return 62.0 * level_px
    if kernel == "gn_pass2":  # photometric Jacobian pass 14 B/px
        return 14.0 * level_px
    if kernel == "splat_project":  # B_pred = 60 M + 38 N0, the 60 M surfel stream is this kernel
        return 60.0 * M
    if kernel == "index_project":
        return 60.0 * M
    if kernel == "clean_flags":  # B_clean = 120 M + 15 N0 split over flags (read) and scatter (write)
        return 60.0 * M + 15.0 * N0
    if kernel == "clean_scatter":
        return 60.0 * M
    if kernel == "depth_bilateral":  # u16 in, u16 out
        return 4.0 * N0
    return float("nan")


def contract_bytes(N0, M, it, rgb, n_pred, n_trk):
    """SURVEY 8(d)'s algorithmic bytes of one whole frame: B = B_pre + n_pred B_pred + n_trk (B_init + B_gn) + 2 B_idx + B_fuse + B_clean."""
    N = [N0, N0 // 4, N0 // 16]  # (W/2)(H/2), (W/4)(H/4) up to the truncation of odd sizes
    per_px = 76.0 if rgb else 48.0
    b_gn = (6.6 * N0 + 21.0 * N0 if rgb else 0.0) + per_px * sum(i * n for i, n in zip(it, N))
    return 12.0 * N0 + n_pred * (60.0 * M + 38.0 * N0) + n_trk * (255.0 * N0 + b_gn) + 2 * (60.0 * M + 52.0 * N0) + (120.0 * M + 64.0 * N0) + (120.0 * M + 15.0 * N0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks = GPUs of this node (0: WORLD_SIZE if launched by torch.distributed.run, else 1)")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-pass", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the nested rocprofv3 counter passes (HBM traffic and VALU activity of the dominant kernel)")
    ap.add_argument("--no-pipeline", action="store_true", help="issue the live-frame half on the main stream (A/B switch)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU baseline sample (0 = auto ~20 s)")
    ap.add_argument("--loop-closure", action="store_true",
                    help="time the 'full' frame step: local loop closure on (INACTIVE prediction + model-to-model tracking every frame)")
    ap.add_argument("--no-full-leg", action="store_true", help="skip the extra 'full' (loop closure on) leg of the default run")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the short legs for BASELINE configs 2 / 4, n_pred = 3 and the populated full step")
    ap.add_argument("--no-session-leg", action="store_true", help="skip the collaborative-session leg (dms_session: two cameras, a merge by the reference's rule)")
    ap.add_argument("--session-loop", action="store_true",
                    help="N = 1: time the headline frames through dms_session_step_async over a one-rank RCCL transport (the loop --gpus N > 1 times)")
    ap.add_argument("--time-delta", type=int, default=200,
                    help="active time window in frames (reference default 200); a small value with --loop-closure populates the INACTIVE "
                         "view on this cyclic stream, so the model-to-model tracker has correspondences")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks (one per GPU, RCCL) ourselves
    if args.gpus > 1 and "RANK" not in os.environ:
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus == 0:
        args.gpus = world
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d: launch with --nproc-per-node %d (or without a launcher: this script starts "
                 "the ranks itself)" % (args.gpus, world, args.gpus))
    distributed = world > 1
    if os.environ.get("DMS_BENCH_DRY") == "1":
        # rendezvous rehearsal without a GPU (tests/test_collab_cpu.py): the ranks meet over gloo, rank 0 reports the job's shape
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if distributed:
            dist.init_process_group(backend="gloo")
            ranks = [None] * world
            dist.all_gather_object(ranks, {"rank": rank, "local_rank": local_rank})
        else:
            ranks = [{"rank": 0, "local_rank": 0}]
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": ranks}))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    import torch

    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # test-only knobs: DMS_BENCH_SHARE_GPU=1 lets several ranks share the GPUs of a smaller box (rank -> device
    # modulo the device count) and DMS_BENCH_BACKEND=gloo replaces RCCL there (two ranks on one device cannot form an
    # RCCL communicator); used to rehearse the multi-rank control flow on a 1-GPU box
    if os.environ.get("DMS_BENCH_SHARE_GPU") == "1":
        local_rank = local_rank % torch.cuda.device_count()
        # processes sharing a device must not run resident (spinning) kernels side by side: the library chains them
        # within a process only; across processes they would time out at their grid barriers (DMS_ERR_TIMEOUT)
        os.environ.setdefault("DMS_TRACK_MODE", "launches")
    backend = os.environ.get("DMS_BENCH_BACKEND", "nccl") if distributed else "none"
    if distributed:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from densemonoslam_amd import capi, collab, fusion, synth

    capi.check(capi.lib.dms_set_device(local_rank), "dms_set_device")
    W, H = args.width, args.height
    K = synth.K_640 if (W, H) == (640, 480) else (synth.K_KITTI if (W, H) == (1241, 376) else (0.825 * W, 0.825 * W, W / 2.0, H / 2.0))
    n_total = args.warmup + args.steps

    # ---- synthetic stream for this rank's camera, resident in HBM before timing --------------
    n_unique = min(n_total, 32)
    rgb_t = torch.empty((n_unique, H, W, 3), dtype=torch.uint8, device=dev)
    dep_t = torch.empty((n_unique, H, W), dtype=torch.int16, device=dev)
    host_frames = []
    for k in range(n_unique):
        d, rgb, _ = synth.frame(k, cam_id=rank, width=W, height=H, K=K, noise=True)
        rgb_t[k] = torch.from_numpy(rgb)
        dep_t[k] = torch.from_numpy(d.view(np.int16))
        if rank == 0:
            host_frames.append((d, rgb))

    def frame_index(i):  # forward then backward along the trajectory: temporally coherent for any length
        period = 2 * (n_unique - 1) if n_unique > 1 else 1
        j = i % period
        return j if j < n_unique else period - j

    def make_engine(loop_closure=args.loop_closure):
        return fusion.ElasticFusion(W, H, K, model_capacity=8_000_000, pipeline_ingest=0 if args.no_pipeline else 1,
                                    local_loop_closure=1 if loop_closure else 0, timeDelta=args.time_delta,
                                    share_projection=int(os.environ.get("DMS_SHARE_PROJECTION", "1")),
                                    fused_fill_in=int(os.environ.get("DMS_FUSED_FILL_IN", "1")))

    stream = torch.cuda.current_stream().cuda_stream
    ef = make_engine()

    # thumbnails exchanged in collaborative mode: W/8 x H/8 image + vertex + normal
    tw, th = W // 8, H // 8
    import ctypes as C

    # collaborative mode (N > 1): every camera publishes its frame block — fern descriptor + thumbnails — per frame;
    # every rank keeps a fern database of its own key frames and searches it with the other cameras' descriptors
    # (DMS_BENCH_EXCHANGE=1: the per-frame exchange work at N = 1 too, the all-gather being a local copy — measures what the
    # collaborative mode adds to a frame without a second GPU)
    exchange_on = distributed or os.environ.get("DMS_BENCH_EXCHANGE") == "1"
    # DMS_BENCH_CARRIER=c: the all-gather through the library's own RCCL binding (include/dmslam_collab.h, what a C++ front end
    # calls) instead of torch.distributed's; the process group only carries the 128-byte communicator id
    carrier = None
    if exchange_on and os.environ.get("DMS_BENCH_CARRIER") == "c" and dev.type == "cuda" and (not distributed or backend == "nccl"):
        carrier = collab.rccl_carrier_from_process_group(rank, world)
    exchange = collab.ThumbnailExchange(world, W, H, dev, extra_bytes=collab.DESC_BYTES if exchange_on else 0, carrier=carrier)
    thumb = exchange.local
    matcher = None
    if exchange_on:
        from densemonoslam_amd import ferns as ferns_mod

        fern_db = ferns_mod.Ferns(W, H, K, num=500, maxDepth_mm=3000, photoThresh=115.0, seed=20260929, capacity=4096)  # same table on every rank
        matcher = collab.InterMapMatcher(fern_db, exchange, rank, world, dev, fern_threshold=0.3095,
                                         verify_interval=int(os.environ.get("DMS_VERIFY_INTERVAL", "0")),
                                         side_stream=os.environ.get("DMS_MATCHER_SIDE", "0") == "1")

    # bounded run-ahead: the host never has more than `depth` frames enqueued beyond the one the GPU
    # is working on (what a live pipeline does anyway: frame t+depth does not exist yet)
    depth = int(os.environ.get("DMS_RUNAHEAD", "0"))
    inflight = []

    def step(i, exchange_thumbnails=True):
        j = frame_index(i)
        if depth > 0:
            if len(inflight) >= depth:
                inflight.pop(0).synchronize()
        ef.processFrameAsync(rgb_t[j].data_ptr(), 3, dep_t[j].data_ptr(), None, 1.0, stream)
        if depth > 0:
            e = torch.cuda.Event()
            e.record()
            inflight.append(e)
        if exchange_on and exchange_thumbnails:
            # frame block (fill-in thumbnails + fern descriptor + pose from HBM), own key-frame database, all-gather beside
            # the next frame; then the search of the local database with the descriptors gathered one frame earlier
            prev = matcher.publish(ef, i + 1, stream)
            matcher.match(prev, i + 1, stream)

    def barrier():
        exchange.finish()  # collectives still in flight belong to the timed region
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    host_wait0 = ef.kernel_time("host_wait")[0]
    t0 = time.perf_counter()
    for i in range(args.warmup, n_total):
        step(i)
    t_enq = time.perf_counter() - t0  # host time to enqueue the timed frames (no synchronisation inside)
    host_wait_ms = ef.kernel_time("host_wait")[0] - host_wait0
    barrier()
    elapsed = time.perf_counter() - t0
    res = ef.fetch(stream)
    M = int(res.surfels)

    elapsed = collab.max_over_ranks(elapsed, dev)  # the slowest rank defines the job time
    M_total = int(collab.sum_over_ranks(M, dev))

    # ---- the headline loop of --gpus N > 1 (and of --session-loop): dms_session_step_async over dms_transport_rccl -----------------
    # One camera per rank, this rank's frames resident in HBM, no inter-map query due inside the timed region (query_from beyond the
    # run: the cameras stay one per GPU, which is what scales weakly; the wake / refine / merge path is timed by the
    # `session_across_ranks` leg below).  Per tick and rank: the frame step, the frame block, key-frame insertion, the all-gather of
    # world blocks, the descriptor search of the hosted database against every gathered block, the host mirror.  Timed exactly like
    # the loop above: W untimed ticks, K timed ticks between barrier + synchronize, max over ranks.
    # The library's RCCL binding had never formed a communicator of more than one rank before this round's first multi-GPU run, so the
    # loop is guarded: set-up failures are agreed on by all ranks (the fall-back figure above stays the headline, labelled), and a
    # watchdog prints the line with that figure if the loop does not finish.
    session_loop = distributed or args.session_loop
    sess = None
    sess_transport = [None]  # (kept for the session_across_ranks leg)
    headline_loop = "dms_fusion_process_frame (one camera, no session)" if not distributed else (
        "fall-back: collab.InterMapMatcher over torch.distributed (round-3 exchange)")
    fallback = None
    if session_loop and not args.loop_closure:
        import threading

        from densemonoslam_amd import session as session_mod

        fallback = {"value": world * args.steps / elapsed, "unit": "frames/s", "ms_per_step": 1000.0 * elapsed / args.steps,
                    "what": "the same frames with the round-3 exchange (collab.InterMapMatcher: frame block, key-frame insertion, all-gather over "
                            "torch.distributed, descriptor search) - timed first, the headline only if the session loop below fails" if exchange_on
                            else "the bare frame step (dms_fusion_process_frame, no session)"}
        sess = {"loop": "dms_session_step_async", "error": None}
        limit_s = int(os.environ.get("DMS_BENCH_SESSION_LIMIT", "180"))
        sess_done = threading.Event()

        def headline_watchdog():
            if sess_done.wait(limit_s):
                return
            if rank == 0:
                line = {"metric": "frames/sec ICP+RGB+fusion @640x480" if (W, H) == (640, 480) else "frames/sec ICP+RGB+fusion @%dx%d" % (W, H),
                        "value": fallback["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                        "ms_per_step": fallback["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                        "data": "synthetic", "headline_loop": headline_loop,
                        "session_loop_error": "dms_session_step_async over the transport did not finish within %d s" % limit_s,
                        "config": {"workload": "TUM fr1/desk-like 640x480 full 3-level ICP+RGB tracking + surfel fusion, one camera per GPU, "
                                               "round-3 exchange (fall-back)", "resolution": [W, H], "cameras_per_gpu": 1}}
                try:
                    import ctypes
                    ctypes.CDLL(None).fflush(None)
                except Exception:  # noqa: BLE001
                    pass
                print(json.dumps(line))
                sys.stdout.flush()
            os._exit(0)

        threading.Thread(target=headline_watchdog, daemon=True).start()
        tr, err = None, None
        try:
            tmode = os.environ.get("DMS_BENCH_SESSION_TRANSPORT", "rccl" if (not distributed or backend == "nccl") else "torch")
            if tmode == "rccl":
                tr = session_mod.RcclTransport(collab.rccl_carrier_from_process_group(rank, world))
            elif tmode == "torch":
                tr = session_mod.TorchTransport(rank, world)
            # ("none": a one-rank session without a transport - nothing is gathered, the search reads the blocks where they were packed)
        except Exception as e:  # noqa: BLE001 (agreed on below: every rank must take the same branch)
            err = "%s: %s" % (type(e).__name__, e)
        ok = 0.0 if err else 1.0
        if distributed:
            flag = torch.tensor([ok], dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = float(flag.item())
        if ok < 1.0:
            sess["error"] = err or "the transport could not be set up on another rank"
        else:
            st_s = capi.create_stream()
            ns = session_mod.NativeSession(W, H, K, world, rank=rank, world=world, transport=tr, query_from=1 << 30, model_capacity=8_000_000,
                                           time_exchange=tr is not None)

            def tick(i):
                j = frame_index(i)
                ns.step_resident(i, [rgb_t[j].data_ptr()], [dep_t[j].data_ptr()], pipelined=True, stream=st_s)

            def sbarrier():
                ns.sync()
                capi.lib.dms_stream_sync(st_s)
                if distributed:
                    dist.barrier()
                torch.cuda.synchronize()

            for i in range(args.warmup):
                tick(i)
            sbarrier()
            ag0 = ns.exchange_time()
            t0 = time.perf_counter()
            for i in range(args.warmup, n_total):
                tick(i)
            t_enq_s = time.perf_counter() - t0
            sbarrier()
            el_s = time.perf_counter() - t0
            ag1 = ns.exchange_time()
            rs = fusion.FrameResult()
            capi.check(fusion.lib.dms_fusion_fetch(capi.lib.dms_session_camera(ns.h, rank), C.byref(rs), st_s), "dms_fusion_fetch")
            el_s = collab.max_over_ranks(el_s, dev)
            ag_ms = (ag1[0] - ag0[0]) / max(1, ag1[1] - ag0[1]) if tr is not None else 0.0
            sess.update(elapsed=el_s, t_enq=t_enq_s, surfels=int(rs.surfels), surfels_total=int(collab.sum_over_ranks(int(rs.surfels), dev)),
                        allgather_ms_per_frame=collab.max_over_ranks(ag_ms, dev), allgather_ms_per_frame_rank0=ag_ms, allgathers_timed=ag1[1] - ag0[1],
                        transport={"rccl": "dms_transport_rccl (the library's RCCL binding, include/dmslam_collab.h)",
                                   "torch": "session.TorchTransport (torch.distributed %s through ctypes callbacks: rehearsal only)" % backend,
                                   "none": "none (one-rank session: nothing gathered)"}[tmode],
                        rccl_ranks=int(capi.lib.dms_collab_size(tr.carrier.h)) if tmode == "rccl" else 0,
                        rccl_library=(capi.lib.dms_collab_library_path() or b"").decode() if tmode == "rccl" else None,
                        block_bytes=int(collab.thumbnail_bytes(W, H)))
            ns.close()
            capi.destroy_stream(st_s)
            sess_transport[0] = tr
            headline_loop = "dms_session_step_async over %s" % ("dms_transport_rccl" if tmode == "rccl" else tmode)
            elapsed, t_enq, M, M_total = el_s, t_enq_s, sess["surfels"], sess["surfels_total"]
        sess_done.set()
    rank_devices = [{"rank": rank, "device": local_rank, "name": torch.cuda.get_device_name(dev)}]
    if distributed:
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_devices[0])
        rank_devices = gathered
        if dist.get_world_size() != args.gpus:
            sys.exit("bench.py: process group of %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))
    share_projection = int(os.environ.get("DMS_SHARE_PROJECTION", "1"))

    fps = world * args.steps / elapsed
    out = {
        "metric": "frames/sec ICP+RGB+fusion @640x480" if (W, H) == (640, 480) else "frames/sec ICP+RGB+fusion @%dx%d" % (W, H),
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        # ranks of the communicator the timed loop's collectives ran on: the LIBRARY's (dms_collab_size) when the session loop is the
        # headline, torch.distributed's for the fall-back loop
        "rccl_ranks": (sess["rccl_ranks"] if sess and not sess["error"] and sess.get("rccl_ranks") else
                       (dist.get_world_size() if (distributed and backend == "nccl") else (0 if distributed else 1))),
        "rccl_library": sess.get("rccl_library") if sess and not sess["error"] else None,
        "headline_loop": headline_loop,
        "allgather_ms_per_frame": sess.get("allgather_ms_per_frame") if sess and not sess["error"] else None,
        "session_loop": None if not sess else {k: v for k, v in sess.items() if k not in ("elapsed", "t_enq")},
        "fallback_exchange_loop": fallback,
        "backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend,
        "exchange_carrier": "dms_collab_allgather (library's RCCL binding)" if carrier is not None else ("torch.distributed" if distributed else "none"),
        "rank_devices": rank_devices,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "host_enqueue_ms_per_step": round(1000.0 * t_enq / args.steps, 4),
        "host_blocked_ms_per_step": round(host_wait_ms / args.steps, 4) if not (sess and not sess["error"]) else None,  # of which: waiting for frame t-2 (0 = host-bound)
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "TUM fr1/desk-like 640x480 full 3-level ICP+RGB tracking (SO3 + {10,5,4} GN iterations) + surfel index-map fusion; "
                        "synthetic box-room RGB-D stream, %s, NID keyframing off (--nkf)"
                        % ("local loop closure on (INACTIVE prediction + model-to-model tracking every frame: the 'full' step)"
                           if args.loop_closure else "loop closure off (--o)") if (W, H) == (640, 480)
                        else "same frame step at %dx%d%s" % (W, H, ", local loop closure on" if args.loop_closure else ""),
            "resolution": [W, H],
            "cameras_per_gpu": 1,
            # model predictions (60 M + 38 N0 bytes each in SURVEY 8(d)'s contract) run per frame: the tracking prediction and the
            # final one; the reference's post-tracking "GlobalPredict" is dead work in this fork (off: `global_predict`), and with
            # share_projection the final prediction's project pass also serves the next frame's tracking prediction
            "n_pred": 2,
            "n_pred_project_passes": 1 if share_projection else 2,
            "time_delta": args.time_delta,
            "loop_icp_count_last_frame": float(res.loop_icp_count) if args.loop_closure else None,
            "surfels_per_map": M,
            "surfels_total": M_total,
            "session": None if not (sess and not sess["error"]) else (
                "timed loop = dms_session_step_async (include/dmslam_session.h), one camera per rank over %s: per tick the frame step, the frame "
                "block (W/8xH/8 thumbnails + 1616-byte tail: fern codes, pose, tick, hit rows), key-frame insertion into the own map's database, ONE "
                "all-gather of %d x %d bytes, the descriptor search of the hosted database against every gathered block, the host mirror; no inter-map "
                "verification due inside the timed region (query_from beyond the run), so every camera keeps its own map and GPU"
                % (sess["transport"], world, sess["block_bytes"] + 1616)),
            "exchange": ("fall-back leg: all-gather of one %d-byte frame block per camera per frame (592-byte fern descriptor + W/8xH/8 thumbnails), "
                         "every rank searches its fern database (%d key frames on rank 0) with the other cameras' descriptors: "
                         "%d remote descriptors found a candidate on rank 0, %d verified"
                         % (thumb.numel(), len(fern_db), matcher.candidates, len(matcher.verified))) if exchange_on else "none (1 camera)",
        },
    }

    Bw = contract_bytes(W * H, M, (10, 5, 4), True, 2, 2 if args.loop_closure else 1)
    out["whole_frame"] = {"contract_bytes_per_frame": Bw, "hbm_frac": Bw * (fps / world) / 8e12,
                          "what": "SURVEY 8(d)'s algorithmic bytes of one frame (this run's surfel count, the 2 predictions actually run) x frames/s per GPU / 8 TB/s"}

    # ---- "full" frame step (report both): the same stream with local loop closure on ----
    if rank == 0 and not distributed and not args.loop_closure and not args.no_full_leg:
        ef_main = ef
        ef = make_engine(True)
        nfull = min(args.steps, 100)
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.warmup, args.warmup + nfull):
            step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        rf = ef.fetch(stream)
        out["full_step"] = {
            "value": nfull / el,
            "unit": "frames/s",
            "ms_per_step": 1000.0 * el / nfull,
            "steps": nfull,
            "what": "same frame step with local loop closure on: + INACTIVE prediction, second (model-to-model) 3-level tracker pass, "
                    "acceptance test and constraint sampling on device (ElasticFusion.cpp:399-474).  On this stream nothing is older than the "
                    "200-frame window, so the INACTIVE view is (nearly) empty, as in any run without a revisit: a tracker level that finds no "
                    "correspondence of either kind ends after that iteration (the remaining ones would repeat it bit for bit). "
                    "`--loop-closure --time-delta 8` measures the same step with a populated view",
            "last_loop_icp_count": float(rf.loop_icp_count),
        }
        ef.close()
        ef = ef_main

    # ---- the other single-GPU configurations of BASELINE.json / SURVEY 8(d), short legs of their own (not in `value`) ----------
    # Each: a fresh engine, `warmup` untimed frames, then `n` frames timed like the main region (enqueue, one synchronise);
    # `hbm_frac_whole_frame` = SURVEY 8(d)'s contract bytes per frame B x frames/s / 8 TB/s with the leg's own surfel count.
    if rank == 0 and not distributed and not args.loop_closure and not args.no_config_legs and (W, H) == (640, 480):
        def leg(opts, what, it, rgb, n_pred, n_trk, Wl=W, Hl=H, Kl=K, n=50):
            if (Wl, Hl) == (W, H):
                rl, dl, nu = rgb_t, dep_t, n_unique
            else:
                nu = 16
                rl = torch.empty((nu, Hl, Wl, 3), dtype=torch.uint8, device=dev)
                dl = torch.empty((nu, Hl, Wl), dtype=torch.int16, device=dev)
                for k in range(nu):
                    d, rgbk, _ = synth.frame(k, cam_id=0, width=Wl, height=Hl, K=Kl, noise=True)
                    rl[k] = torch.from_numpy(rgbk)
                    dl[k] = torch.from_numpy(d.view(np.int16))
            eng = fusion.ElasticFusion(Wl, Hl, Kl, model_capacity=8_000_000, pipeline_ingest=0 if args.no_pipeline else 1, **opts)

            def fi(i):
                period = 2 * (nu - 1)
                j = i % period
                return j if j < nu else period - j

            for i in range(args.warmup):
                eng.processFrameAsync(rl[fi(i)].data_ptr(), 3, dl[fi(i)].data_ptr(), None, 1.0, stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.warmup, args.warmup + n):
                eng.processFrameAsync(rl[fi(i)].data_ptr(), 3, dl[fi(i)].data_ptr(), None, 1.0, stream)
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            r = eng.fetch(stream)
            eng.close()
            Ml, fpsl = int(r.surfels), n / el
            B = contract_bytes(Wl * Hl, Ml, it, rgb, n_pred, n_trk)
            return {"value": fpsl, "unit": "frames/s", "ms_per_step": 1000.0 * el / n, "steps": n, "warmup": args.warmup, "resolution": [Wl, Hl],
                    "surfels": Ml, "contract_bytes_per_frame": B, "hbm_frac_whole_frame": B * fpsl / 8e12, "what": what,
                    **({"last_loop_icp_count": float(r.loop_icp_count)} if opts.get("local_loop_closure") else {})}

        out["configs"] = {
            "C2": leg(dict(icpWeight=100.0, fastOdom=1, pyramid=0, so3=0), "BASELINE config 2: ICP-only odometry (icpWeight 100 => no photometric term, "
                      "RGBDOdometry.cpp:279), --fo, single pyramid level, no SO3: 3 point-to-plane iterations at 640x480, then the same fusion",
                      (3, 0, 0), False, 2, 1),
            "C3_n_pred3": leg(dict(global_predict=1, share_projection=0), "BASELINE config 3 with the reference's three model predictions per frame "
                              "(post-tracking 'GlobalPredict' on, every prediction projects the map itself)", (10, 5, 4), True, 3, 1),
            "full_step_populated": leg(dict(local_loop_closure=1, timeDelta=8), "the 'full' step (local loop closure: INACTIVE prediction + model-to-model "
                                       "tracker) with an 8-frame active window, so the INACTIVE view is populated on this back-and-forth stream",
                                       (10, 5, 4), True, 3, 2),
            "C4": leg(dict(depthCut=40.0), "BASELINE config 4 geometry: 1241x376, KITTI intrinsics, 40 m depth cut-off, full 3-level ICP+RGB tracking "
                      "+ fusion (synthetic depth in place of the absent depth network)", (10, 5, 4), True, 2, 1, 1241, 376, synth.K_KITTI),
        }

        # the frame rate in the REFERENCE's shape (three model predictions per frame, each projecting the map itself:
        # ElasticFusion.cpp:165,273,586), at top level beside `value` (whose timed region runs the two predictions that are not dead with --o)
        out["value_ref_shape"] = {"value": out["configs"]["C3_n_pred3"]["value"], "unit": "frames/s", "steps": out["configs"]["C3_n_pred3"]["steps"],
                                  "surfels": out["configs"]["C3_n_pred3"]["surfels"], "what": "configs.C3_n_pred3: n_pred = 3, no shared projection"}

    # ---- the collaborative session behind the boundary (dms_session, include/dmslam_session.h), BASELINE config 5 in the small ----------
    # Two cameras of one session on this GPU at 640 x 480 in the cluttered-corner scene, camera 1 eight frames ahead of camera 0 on the
    # same path: independent maps, per-tick publish / query (Ferns::findFrame with interMap = 1 on each other's thumbnails), at tick 6
    # the reference's rule verifies, the matched map's owner refines at full resolution (ReferenceFrame.h:72-110) and consumes the
    # other map; from then on both cameras track against and fuse into ONE map.  Timed per tick with the host in the loop (the session
    # synchronises where the reference does): before the merge, the merge tick itself, after it.  Not part of `value`.
    if rank == 0 and not distributed and not args.loop_closure and not args.no_session_leg and (W, H) == (640, 480):
        from densemonoslam_amd import session as session_mod

        n_ticks, q_from, off = 22, 6, 8
        sframes = [[synth.frame(k + o, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)[:2] for o in (0, off)] for k in range(n_ticks)]
        ns = session_mod.NativeSession(W, H, K, 2, query_from=q_from, model_capacity=8_000_000)
        tick_ms = []
        for k in range(n_ticks):
            fr = {c: (sframes[k][c][1], sframes[k][c][0]) for c in range(2)}
            t1 = time.perf_counter()
            ns.step(k, fr)
            torch.cuda.synchronize()
            tick_ms.append(1000.0 * (time.perf_counter() - t1))
        mg, rf = ns.merges, ns.refinements
        k_m = mg[0][0] if mg else None
        pre = tick_ms[1:k_m] if k_m else tick_ms[1:]
        post = tick_ms[k_m + 1:] if k_m is not None else []
        out["session"] = {
            "what": "dms_session (the session's protocol compiled into the library): 2 cameras on this GPU, %dx%d, cluttered-corner scene; per tick both "
                    "cameras' frame steps, key-frame insertion, thumbnail exchange and the inter-map query (interMap = 1) of each camera against the "
                    "other map; host in the loop (upload of both frames included)" % (W, H),
            "merges": [(m[0], m[1], m[2]) for m in mg],
            "refinements": rf,
            "ms_per_tick_before_merge": round(float(np.median(pre)), 3) if pre else None,  # (medians: five and fifteen ticks, each timed with the host in the loop)
            "frames_per_s_before_merge": round(2000.0 / float(np.median(pre)), 1) if pre else None,
            "ms_merge_tick": round(tick_ms[k_m], 3) if k_m is not None else None,
            "ms_per_tick_after_merge": round(float(np.median(post)), 3) if post else None,
            "frames_per_s_after_merge": round(2000.0 / float(np.median(post)), 1) if post else None,
            "ms_per_tick": [round(t, 3) for t in tick_ms],
            "surfels_after": int(len(ns.cams[mg[0][1]].model())) if mg else None,
        }
        ns.close()

        # The same session through dms_session_step_async: frames resident in HBM, no host synchronisation on the frame's path (pose
        # graphs arrive from the gathered blocks two ticks late), the reference's full query only at the tick a descriptor hit wakes
        # (three ticks after the search that hit).  Timed as a whole per phase - the host only enqueues, so a per-tick clock would
        # measure the enqueue - with a stream synchronisation at each phase boundary.
        from densemonoslam_amd import capi as capi_mod

        N = W * H
        dev_frames = []
        for k in range(n_ticks):
            row = []
            for c in range(2):
                br, bd = capi_mod.DeviceBuffer(N * 3), capi_mod.DeviceBuffer(N * 2)
                br.upload(np.ascontiguousarray(sframes[k][c][1], np.uint8))
                bd.upload(np.ascontiguousarray(sframes[k][c][0], np.uint16))
                row.append((br, bd))
            dev_frames.append(row)
        st = capi_mod.create_stream()

        def pipelined_pass(bounds):
            """the whole session through the pipelined step; returns (merges, stats, ms of every [bounds[i], bounds[i + 1]) tick range)"""
            ns = session_mod.NativeSession(W, H, K, 2, query_from=q_from, model_capacity=8_000_000)
            ms = []
            for k0, k1 in zip(bounds[:-1], bounds[1:]):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in range(k0, k1):
                    ns.step_resident(k, [dev_frames[k][c][0].ptr for c in range(2)], [dev_frames[k][c][1].ptr for c in range(2)], pipelined=True, stream=st)
                ns.sync()
                capi_mod.lib.dms_stream_sync(st)
                torch.cuda.synchronize()
                ms.append(1000.0 * (time.perf_counter() - t1))
            res = (ns.merges, ns.async_stats(), ms)
            ns.close()
            return res

        # one camera, the headline loop's own frames (same K timed steps after W warm-up ticks) through the pipelined tick: what the
        # session's per-tick exchange costs a GPU that serves ONE camera (without another rank there is no other camera's block to
        # search: frame block, key-frame insertion and host mirror only)
        ns1 = session_mod.NativeSession(W, H, K, 1, query_from=1 << 30, model_capacity=8_000_000)
        for i in range(args.warmup):
            ns1.step_resident(i, [rgb_t[frame_index(i)].data_ptr()], [dep_t[frame_index(i)].data_ptr()], pipelined=True, stream=st)
        capi_mod.lib.dms_stream_sync(st)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.warmup, n_total):
            ns1.step_resident(i, [rgb_t[frame_index(i)].data_ptr()], [dep_t[frame_index(i)].data_ptr()], pipelined=True, stream=st)
        ns1.sync()
        capi_mod.lib.dms_stream_sync(st)
        torch.cuda.synchronize()
        el1 = time.perf_counter() - t1
        ns1.close()
        out["session"]["one_camera_steady_state"] = {
            "what": "the headline loop's frames through dms_session_step_async with one camera and no other rank",
            "frames_per_s": round(args.steps / el1, 1), "of_value": round(args.steps / el1 / fps, 3)}
        # the figure the --gpus N > 1 headline loop is to be compared with at N = 1 (python bench.py --session-loop times it as `value`)
        out["value_session_loop"] = {"value": args.steps / el1, "unit": "frames/s", "of_value": round(args.steps / el1 / fps, 3),
                                     "what": "session.one_camera_steady_state: the same frames through dms_session_step_async (one camera, no transport)"}

        mg2, _, _ = pipelined_pass([0, n_ticks])  # (where the schedule merges: the timed pass puts its phase boundaries there)
        if mg2:
            km = mg2[0][0]
            mg2, stats2, ms2 = pipelined_pass([0, 1, km, km + 1, n_ticks])
            out["session"]["pipelined"] = {
                "what": "dms_session_step_async on the same frames, resident in HBM; phases timed whole (stream synchronised at the boundaries: "
                        "bootstrap tick | ticks before the woken tick | the woken tick: fetch, full query, refinement, merge | ticks after it)",
                "merges": [(m[0], m[1], m[2]) for m in mg2], "stats": stats2,
                "ms_per_tick_before_merge": round(ms2[1] / (km - 1), 3),
                "frames_per_s_before_merge": round(2000.0 * (km - 1) / ms2[1], 1),
                "ms_merge_tick": round(ms2[2], 3),
                "ms_per_tick_after_merge": round(ms2[3] / (n_ticks - km - 1), 3),
                "frames_per_s_after_merge": round(2000.0 * (n_ticks - km - 1) / ms2[3], 1),
            }
        capi_mod.destroy_stream(st)

    # The same session ACROSS ranks (with --gpus N; DMS_BENCH_SESSION=0 skips it, =1 also runs it over gloo): one camera per rank, camera r eight frames ahead of
    # camera r - 1 on the corner path, RCCL transport (dms_transport_rccl) - or gloo through ctypes callbacks in the one-GPU rehearsal.
    # Cameras migrate to the consuming rank as their maps merge; rank 0 reports the per-tick times and the merge log.  Not part of
    # `value`; every rank takes part (the session's collectives).
    # It runs AFTER the headline loop and under a watchdog: the library's own RCCL binding (dms_transport_rccl) has only ever formed a
    # one-rank communicator on the one-GPU boxes this was developed on, so if this leg does not finish (or fails on any rank) rank 0
    # prints the line with the figures measured so far and an error entry, and every rank exits.
    def session_across_ranks():
        from densemonoslam_amd import session as session_mod

        n_ticks, q_from, off = 16, 6, 8
        # (the headline loop's transport - ONE communicator of the library's per process - when that loop ran)
        tr = sess_transport[0] if sess_transport[0] is not None else (
            session_mod.RcclTransport(collab.rccl_carrier_from_process_group(rank, world)) if backend == "nccl" else session_mod.TorchTransport(rank, world))
        ns = session_mod.NativeSession(W, H, K, world, rank=rank, world=world, transport=tr, query_from=q_from, model_capacity=8_000_000)
        tick_ms = []
        for k in range(n_ticks):
            d, rgbk, _ = synth.frame(k + off * rank, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
            dist.barrier()
            t1 = time.perf_counter()
            ns.step(k, {rank: (rgbk, d)})
            torch.cuda.synchronize()
            tick_ms.append(1000.0 * (time.perf_counter() - t1))
        if rank == 0:
            out["session_across_ranks"] = {"cameras": world, "transport": "rccl" if backend == "nccl" else backend, "ms_per_tick": [round(t, 3) for t in tick_ms],
                                           "merges": [(m[0], m[1], m[2]) for m in ns.merges], "refinements": ns.refinements, "frame_of": ns.frame_of,
                                           "hosted_on_rank0": ns.hosted()}
        ns.close()
        # ... and through the pipelined step (dms_session_step_async), frames resident: the phase before the first woken tick is the
        # steady state of N independent cameras with the exchange running (what scales weakly); the phase after the last merge is one
        # rank serving every camera while the others only forward frames.
        from densemonoslam_amd import capi as capi_mod

        n_ticks = 14 + 4 * world
        Npx = W * H
        mine = []
        for k in range(n_ticks):
            d, rgbk, _ = synth.frame(k + off * rank, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
            br, bd = capi_mod.DeviceBuffer(Npx * 3), capi_mod.DeviceBuffer(Npx * 2)
            br.upload(np.ascontiguousarray(rgbk, np.uint8))
            bd.upload(np.ascontiguousarray(d, np.uint16))
            mine.append((br, bd))
        st = capi_mod.create_stream()

        def pipelined_pass(bounds):
            ns = session_mod.NativeSession(W, H, K, world, rank=rank, world=world, transport=tr, query_from=q_from, model_capacity=8_000_000)
            ms = []
            for k0, k1 in zip(bounds[:-1], bounds[1:]):
                torch.cuda.synchronize()
                dist.barrier()
                t1 = time.perf_counter()
                for k in range(k0, k1):
                    ns.step_resident(k, [mine[k][0].ptr], [mine[k][1].ptr], pipelined=True, stream=st)
                ns.sync()
                capi_mod.lib.dms_stream_sync(st)
                torch.cuda.synchronize()
                dist.barrier()
                ms.append(1000.0 * (time.perf_counter() - t1))
            res = (ns.merges, ns.async_stats(), ms, ns.frame_of)
            ns.close()
            return res

        # ... and the headline's own stream (this rank's camera in the box room, the same K timed steps after W warm-up ticks) through the
        # pipelined session with no query ever due: N independent cameras with the session's exchange running - the figure to put
        # beside `value`
        ns = session_mod.NativeSession(W, H, K, world, rank=rank, world=world, transport=tr, query_from=1 << 30, model_capacity=8_000_000)

        def tick(i):
            j = frame_index(i)
            ns.step_resident(i, [rgb_t[j].data_ptr()], [dep_t[j].data_ptr()], pipelined=True, stream=st)

        for i in range(args.warmup):
            tick(i)
        capi_mod.lib.dms_stream_sync(st)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for i in range(args.warmup, n_total):
            tick(i)
        ns.sync()
        capi_mod.lib.dms_stream_sync(st)
        torch.cuda.synchronize()
        dist.barrier()
        el = collab.max_over_ranks(time.perf_counter() - t1, dev)
        if rank == 0:
            out["session_across_ranks"]["steady_state"] = {
                "what": "the headline loop's frames (%d timed steps per rank after %d warm-up ticks) through dms_session_step_async over the "
                        "transport, no query due: per tick one frame per rank, frame block, key-frame insertion, all-gather, descriptor search of "
                        "every other camera's block, host mirror" % (args.steps, args.warmup),
                "frames_per_s": round(world * args.steps / el, 1), "ms_per_tick": round(1000.0 * el / args.steps, 4),
                "of_value": round(world * args.steps / el / fps, 3)}
        ns.close()

        mg2, _, _, _ = pipelined_pass([0, n_ticks])
        if mg2 and mg2[-1][0] + 1 < n_ticks and q_from + 3 > 1:
            first_wake, last = q_from + 3, mg2[-1][0]
            mg2, stats2, ms2, fo2 = pipelined_pass([0, 1, first_wake, last + 1, n_ticks])
            if rank == 0:
                out["session_across_ranks"]["pipelined"] = {
                    "merges": [(m[0], m[1], m[2]) for m in mg2], "stats": stats2, "frame_of": fo2,
                    "ms_per_tick_before_first_wake": round(ms2[1] / (first_wake - 1), 3),
                    "frames_per_s_before_first_wake": round(1000.0 * world * (first_wake - 1) / ms2[1], 1),
                    "ms_wake_and_merge_phase": round(ms2[2], 3), "ticks_in_that_phase": last + 1 - first_wake,
                    "ms_per_tick_after_last_merge": round(ms2[3] / (n_ticks - last - 1), 3),
                    "frames_per_s_after_last_merge": round(1000.0 * world * (n_ticks - last - 1) / ms2[3], 1),
                }
        capi_mod.destroy_stream(st)

    sess_env = os.environ.get("DMS_BENCH_SESSION", "")
    if distributed and (W, H) == (640, 480) and not args.loop_closure and (sess_env == "1" or (sess_env != "0" and backend == "nccl")):
        import threading

        limit_s = int(os.environ.get("DMS_BENCH_SESSION_LIMIT", "180"))

        def give_up():
            if rank == 0:
                out.setdefault("session_across_ranks", {})["error"] = out.get("session_across_ranks", {}).get("error") or (
                    "the leg did not finish within %d s; the figures above it were measured before it started" % limit_s)
                print(json.dumps(out))
                sys.stdout.flush()
            os._exit(0)

        timer = threading.Timer(limit_s, give_up)
        timer.daemon = True
        timer.start()
        try:
            session_across_ranks()
        except Exception as e:  # noqa: BLE001 (the other ranks may be waiting in a collective: nobody goes on to the final barrier)
            import traceback

            traceback.print_exc()
            if rank == 0:
                out.setdefault("session_across_ranks", {})["error"] = "%s: %s" % (type(e).__name__, e)
            threading.Event().wait()  # (until the watchdog ends the process)
        timer.cancel()

    # ---- per-kernel timing with HIP events on the launch stream (own passes, not in `value`) ------
    if rank == 0 and not args.no_kernel_pass:
        od = fusion.lib.dms_fusion_odometry(ef.h)
        nprof = min(args.steps, 20)
        next_frame = [n_total]

        def kernel_pass(per_frame_fetch):
            """nprof frames with an event pair around every launch.  per_frame_fetch = False: the frames are enqueued as in the timed
            region (two frames in flight, the live half of frame t + 1 beside frame t) and drained once at the end;
            True: the host waits for every frame (each kernel alone on the device)."""
            ef.set_profiling(True)
            capi.check(capi.lib.dms_odometry_set_profiling(C.c_void_p(od), 1))
            for i in range(next_frame[0], next_frame[0] + nprof):
                step(i, exchange_thumbnails=False)  # rank 0 only: no collective in this pass
                if per_frame_fetch:
                    ef.fetch(stream)
            next_frame[0] += nprof
            ef.fetch(stream)
            stages, kern, phases = {}, {}, {}
            for n in ["ingest", "preprocess", "live_pyramids", "predict", "fill_in", "odom_init", "track", "predict_old", "loop_init", "loop_track",
                      "index_map", "fuse", "clean", "initialise"]:
                ms, cnt = ef.kernel_time(n)
                if cnt:
                    stages[n] = {"ms_per_frame": ms / nprof, "launches_per_frame": cnt / nprof}
            for n in ("track_coarse", "so3_model", "so3_level", "gn_level0", "gn_level1", "gn_level2", "gn_pass1", "gn_pass2", "gn_solve", "so3_pass", "track_init", "track_finalize"):
                ms, cnt = C.c_double(0), C.c_int(0)
                capi.check(capi.lib.dms_odometry_get_kernel_time(C.c_void_p(od), n.encode(), C.byref(ms), C.byref(cnt)))
                if cnt.value:
                    kern[n] = {"ms_per_frame": ms.value / nprof, "avg_us": 1000.0 * ms.value / cnt.value, "launches_per_frame": cnt.value / nprof}
            phase_names = ["setup", "pass1", "count_pair_arrival", "icp_sum_and_pair_wait", "pass2_and_rgb_sum", "totals_wait", "unused", "solve",
                           "writeback", "clock_overhead"]
            for lvl in range(3):
                row = {}
                for i, n in enumerate(phase_names):
                    ms, cnt = C.c_double(0), C.c_int(0)
                    capi.check(capi.lib.dms_odometry_get_kernel_time(C.c_void_p(od), ("phase:%d" % (lvl * 16 + i)).encode(), C.byref(ms), C.byref(cnt)))
                    if n != "unused":
                        row[n] = round(1000.0 * ms.value / nprof, 2)
                phases["L%d" % lvl] = row
            return stages, kern, phases

        def level0_pass():
            """n0 frames enqueued exactly as in the timed region, with ONE event pair per frame: around the dominant kernel's launch.
            (With a pair around every launch of both streams the prep stream's kernels shift against the tracker's and the level-0
            launch waits for compute units it never waits for in the timed region.)  Returns mean, median, max and how many launches
            took more than 1.5 x the median (a resident grid that started incomplete waits for another stream's blocks to retire)."""
            n0 = min(args.steps, 100)
            ef.set_profiling(False)
            capi.check(capi.lib.dms_odometry_set_profiling(C.c_void_p(od), 2))
            for i in range(next_frame[0], next_frame[0] + n0):
                step(i, exchange_thumbnails=False)
            next_frame[0] += n0
            ef.fetch(stream)

            def q(name):
                ms, cnt = C.c_double(0), C.c_int(0)
                capi.check(capi.lib.dms_odometry_get_kernel_time(C.c_void_p(od), name, C.byref(ms), C.byref(cnt)))
                return ms.value, cnt.value

            tot, cnt = q(b"gn_level0")
            med, _ = q(b"gn_level0:median")
            mx, _ = q(b"gn_level0:max")
            extra, slow = q(b"gn_level0:slow")
            capi.check(capi.lib.dms_odometry_set_profiling(C.c_void_p(od), 0))
            if not cnt:
                return None, None
            return 1000.0 * tot / cnt, {"launches": cnt, "median_us": round(1000.0 * med, 2), "max_us": round(1000.0 * mx, 2),
                                        "launches_over_1p5_median": slow, "their_excess_us_per_launch_overall": round(1000.0 * extra / cnt, 2)}

        us_level0_as_timed, level0_spread = level0_pass()
        stages_pipe, kern_pipe, _ = kernel_pass(False)
        stages, kern, phases = kernel_pass(True)
        if sum(sum(r.values()) for r in phases.values()) > 0:
            out["gn_level_phase_us_per_frame"] = phases  # in-kernel clock of block 0, summed over the level's iterations (isolated pass)
        out["stage_ms_per_frame"] = {k: round(v["ms_per_frame"], 4) for k, v in stages.items()}
        out["stage_ms_per_frame_pipelined"] = {k: round(v["ms_per_frame"], 4) for k, v in stages_pipe.items()}  # (frames enqueued as in the timed region)
        out["tracker_kernels"] = {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in kern.items()}
        out["tracker_kernels_pipelined"] = {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in kern_pipe.items()}
        # dominant kernel: the resident Gauss-Newton kernel of pyramid level 0 (10 iterations in one
        # launch); in DMS_TRACK_MODE=launches the per-iteration pass-1 kernel instead
        px = [W * H, (W // 2) * (H // 2), (W // 4) * (H // 4)]
        its = [10, 5, 4]
        # what a plain device-to-device copy reaches on this box, next to the spec peak
        try:
            a_ = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
            b_ = torch.empty_like(a_)
            b_.copy_(a_)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b_.copy_(a_)
            e1.record()
            torch.cuda.synchronize()
            measured_copy = 2.0 * a_.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9  # read + write
            del a_, b_
        except Exception:
            measured_copy = None
        pmc = None
        if "gn_level0" in kern and not args.no_pmc and not distributed:
            pmc = pmc_passes(args, W, H)
        if "gn_level0" in kern:
            bytes_per_launch = algorithmic_bytes("gn_level", W, H, M, px[0]) * its[0]
            us_all = kern_pipe.get("gn_level0", kern["gn_level0"])["avg_us"]
            us_pipe = us_level0_as_timed or us_all
            us_iso = kern["gn_level0"]["avg_us"]
            achieved = bytes_per_launch / (us_pipe * 1e-6) / 1e9
            out["roofline"] = {
                "bound": "hbm",
                "kernel": "k_gn_level<ICP,RGB,P> (pyramid level 0: all 10 Gauss-Newton iterations in one resident launch)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None if not pmc else pmc.get("hbm_bytes_per_launch"),
                "bytes_per_launch": bytes_per_launch,
                "avg_launch_us": us_pipe,
                "launch_spread": level0_spread,
                "avg_launch_us_every_launch_bracketed": us_all,
                "avg_launch_us_source": "one HIP event pair per frame around this kernel's launch, on its stream, frames enqueued as in the timed "
                                        "region (pipelined: the next frame's live half runs beside this frame) and nothing else bracketed; "
                                        "`frac` uses this duration.  avg_launch_us_every_launch_bracketed: the same pass with a pair around "
                                        "every launch of both streams, which shifts the prep stream's kernels against the tracker's",
                "avg_launch_us_isolated": us_iso,
                "frac_isolated": bytes_per_launch / (us_iso * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "measured_copy_GBps": None if measured_copy is None else round(measured_copy, 1),
                "note": "algorithmic bytes = 76 B/px/iteration (SURVEY 8d) x %d px x %d iterations.  The level's maps are read from HBM once and "
                        "stay in registers / L2 for the ten iterations (`traffic`), so this launch cannot be HBM-bound: its time is 20 grid-wide "
                        "integer all-reduces, 10 one-lane 6x6 solves and its per-CU arithmetic — see `issue_roof` (DESIGN.md 6)" % (px[0], its[0]),
            }
            if pmc:
                out["roofline"]["traffic_source"] = pmc.get("source")
                out["roofline"]["traffic_kernel"] = pmc.get("kernel")
                # the figure must reproduce from profiles/: compare with the committed stand-alone PMC record of the same kernel
                # (scripts/collect_profiles.sh writes it) and say so loudly when it does not
                exp_path = os.path.join(ROOT, "profiles", "level0_pmc_expected.json")
                if (W, H) == (640, 480) and os.path.exists(exp_path):
                    exp = json.load(open(exp_path))
                    dev_ = pmc["hbm_bytes_per_launch"] / exp["hbm_bytes_per_launch"] - 1.0
                    out["roofline"]["traffic_vs_profiles"] = {"expected": exp["hbm_bytes_per_launch"], "source": exp.get("source"), "deviation": round(dev_, 4),
                                                              "consistent": abs(dev_) <= 0.10}
                    if abs(dev_) > 0.10:
                        print("bench.py: WARNING: roofline.traffic %.2f MB deviates %.0f %% from profiles/ (%s: %.2f MB)"
                              % (pmc["hbm_bytes_per_launch"] / 1e6, 100 * dev_, exp.get("source"), exp["hbm_bytes_per_launch"] / 1e6), file=sys.stderr)
                if pmc.get("valu_insts_per_launch"):
                    # second roof of the same kernel: vector-instruction issue.  One VALU per SIMD; ACTIVE_INST_VALU counts the
                    # quad-cycles in which a wave executed a vector instruction, summed over the launch's waves.
                    simds = pmc["blocks"] * 4
                    busy_us = pmc["valu_active_quadcycles_per_launch"] * 4.0 / simds / (CLOCK_GHZ * 1e3)
                    out["roofline"]["issue_roof"] = {
                        "valu_insts_per_launch": pmc["valu_insts_per_launch"],
                        "valu_busy_us_per_simd": round(busy_us, 2),
                        "issue_frac": round(busy_us / pmc["launch_us_under_pmc"], 3),
                        "simds_used": simds,
                        "launch_us_under_pmc": pmc["launch_us_under_pmc"],
                        "what": "SQ_ACTIVE_INST_VALU x 4 cycles / (blocks x 4 SIMDs) / %.1f GHz = time a SIMD's vector ALU was busy; "
                                "issue_frac = that / the launch duration in the same (single-stream) run" % CLOCK_GHZ,
                    }
        elif "gn_pass1" in kern:
            bytes_per_frame = sum(algorithmic_bytes("gn_pass1", W, H, M, p) * n for p, n in zip(px, its))
            launches = sum(its)
            avg_s = kern["gn_pass1"]["avg_us"] * 1e-6
            achieved = (bytes_per_frame / launches) / avg_s / 1e9
            out["roofline"] = {
                "bound": "hbm",
                "kernel": "k_gn_pass1<ICP,RGB> (average over the 10+5+4 launches of the 3-level pyramid)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "bytes_per_launch": bytes_per_frame / launches,
                "avg_launch_us": kern["gn_pass1"]["avg_us"],
            }

    # ---- CPU baseline: the oracle (a port of the reference algorithm) on the host cores -----------
    if rank == 0 and not distributed and not args.no_cpu_baseline:  # reported at N = 1 only
        from oracle import orc_pipeline  # checker / baseline only

        from oracle import orc as _orc

        def cpu_model():
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        return line.split(":", 1)[1].strip()
            except OSError:
                pass
            return "unknown"

        def run_oracle(threads, budget_s, max_frames):
            n = _orc.set_threads(threads)
            o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=8_000_000)
            tcpu, done = 0.0, 0
            for k in range(max_frames):
                d, rgb = host_frames[frame_index(k)]  # the GPU legs' order: forward, then backward along the trajectory
                t1 = time.perf_counter()
                o.processFrame(rgb, d)
                dt = time.perf_counter() - t1
                if k >= 2:  # frame 0 is the bootstrap frame, frame 1 the first tracked one (cold caches): not steady-state steps
                    tcpu += dt
                    done += 1
                if tcpu > budget_s:
                    break
            return n, (done / tcpu if tcpu > 0 else None), done

        # All host cores = the cores this process may use: the GPU boxes report 256 hardware threads and run the container
        # under a CPU quota (cgroup cpu.max: 16 CPUs on the boxes seen) - threads beyond the quota only add barrier time
        # (profiles/r05_cpu_baseline_threads.jsonl: 4.6 / 4.8 / 3.4 / 1.8 / 0.07 frames/s at 16 / 32 / 64 / 128 / 256 threads).
        nproc = os.cpu_count() or 1
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
        except (OSError, ValueError):
            try:  # cgroup v1
                q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0 and per > 0:
                    quota = q / per
            except (OSError, ValueError):
                pass
        usable = max(1, min(nproc, int(quota))) if quota else nproc
        cores, fps_all, n_all = run_oracle(int(os.environ.get("DMS_CPU_THREADS", min(32, usable))), 60.0, args.cpu_frames or 102)
        _, fps_one, n_one = run_oracle(1, 8.0, args.cpu_frames or 102)
        out["cpu_baseline"] = {
            "value": fps_all,
            "unit": "frames/s",
            "cores": cores,
            "kind": "port",
            "one_core": {"value": fps_one, "unit": "frames/s", "cores": 1, "frames": n_one},
            "host": {"nproc": nproc, "cpu_quota_cores": quota, "usable_cores": usable, "cpu_model": cpu_model()},
            "sample": "%d steady-state frames (after the bootstrap frame and the first tracked one) of the same synthetic stream, %dx%d, run by the "
                      "oracle/ C restatement of the reference algorithm with %d of the host's %d hardware threads: OpenMP covers the per-pixel loops "
                      "of the tracker, the depth filter and the vertex / rasterisation stages of the surfel-map passes (per-thread z-buffers over "
                      "contiguous surfel ranges, merged in draw order with GL_LESS, so the bits are the sequential draws'); on one thread stay "
                      "the parts the draw order defines (the fuse's feedback sequence, the clean's compaction) and the Python glue's array copies.  "
                      "These are all the cores this container may use: the host reports %d hardware threads under a cgroup CPU quota of %s CPUs "
                      "(host.cpu_quota_cores), and threads beyond the quota only add barrier time - measured on such a host "
                      "(profiles/r05_cpu_baseline_threads.jsonl): 0.71 / 3.1 / 4.6 / 4.8 / 3.4 / 1.8 / 0.07 frames/s at 1 / 8 / 16 / 32 / 64 / 128 / 256 "
                      "threads; sixteen independent 16-thread cameras at once deliver 3.5 - 5.6 frames/s in aggregate (profiles/r05_cpu_sixteen_replicas.txt, DESIGN.md 6).  ~100 frames keep "
                      "the default run inside its time budget.  A restatement written to be checked against, not tuned: it says nothing about kernel "
                      "quality; one_core: the same with 1 thread for ~8 s.  Non-target" % (n_all, W, H, cores, nproc, nproc, ("%g" % quota) if quota else "no"),
        }

    if rank == 0:
        # (RCCL prints its version banner through C stdio, which - redirected to a file - is flushed at exit, BEHIND this line:
        # flush it first so that the JSON line is the last line of stdout)
        try:
            C.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(out))
        sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
