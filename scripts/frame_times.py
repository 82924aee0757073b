"""Per-frame completion times of a short pipelined run (the driver's bench form: 5 warm-up + 20 timed frames):
where the time of a 20-frame run goes.  usage: python scripts/frame_times.py [warmup] [steps]"""
import sys
import time

import numpy as np
import torch

from densemonoslam_amd import fusion, synth

W, H, K = 640, 480, (528.0, 528.0, 320.0, 240.0)
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(warm + steps)]
rgb_t = [torch.from_numpy(np.ascontiguousarray(f[1])).to(dev) for f in frames]
dep_t = [torch.from_numpy(np.ascontiguousarray(f[0])).to(dev) for f in frames]
ef = fusion.ElasticFusion(W, H, K, model_capacity=8_000_000)
stream = torch.cuda.current_stream().cuda_stream
for i in range(warm):
    ef.processFrameAsync(rgb_t[i].data_ptr(), 3, dep_t[i].data_ptr(), None, 1.0, stream)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = []
t0 = time.perf_counter()
ev[0].record()
for i in range(steps):
    ef.processFrameAsync(rgb_t[warm + i].data_ptr(), 3, dep_t[warm + i].data_ptr(), None, 1.0, stream)
    ev[i + 1].record()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
total = time.perf_counter() - t0
gpu = [ev[0].elapsed_time(ev[i + 1]) for i in range(steps)]
print("total %.3f ms for %d frames = %.1f frames/s" % (total * 1e3, steps, steps / total))
prev = 0.0
for i in range(steps):
    print("frame %2d: enqueued by %.3f ms, done at %.3f ms (+%.3f)" % (i, host[i] * 1e3, gpu[i], gpu[i] - prev))
    prev = gpu[i]
