"""Four cameras on one MI355X, more ticks than the tests run: the compiled session (synchronous and pipelined tick) against the one-process
oracle session, 12 - 14 ticks past the last merge with all four cameras tracking against and fusing into ONE map.
usage (GPU box): python scripts/session_four_long.py [ticks=32]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from densemonoslam_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402
from tests.test_session_cpu import FOUR_OFFSETS, SCENARIOS, check_four, frames_at, run_oracle_session_n  # noqa: E402
from tests.test_session_gpu import _make_session, _result_of  # noqa: E402

orc.set_threads(16)
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = SCENARIOS["reference_rule"]
for pipelined in (False, True):
    t0 = time.time()
    ref = run_oracle_session_n(FOUR_OFFSETS, ticks, **({"wake_latency": 3} if pipelined else {}))
    t1 = time.time()
    s = _make_session("native", sc, 4, capacity=6_000_000)
    for k in range(ticks):
        s.step(k, frames_at(synth, k, FOUR_OFFSETS), **({"pipelined": True} if pipelined else {}))
    res = _result_of(s, 0, pipelined)
    check_four(ref, {0: res}, 1, ticks)
    fb = ref.frame_of[0]
    print("%s tick, %d ticks: merges %s, final map %d surfels, %d key frames - map, four trajectories, transforms, constraints identical to the oracle session "
          "(oracle %.0f s, product %.0f s incl. frame synthesis)" % ("pipelined" if pipelined else "synchronous", ticks, [(m[0], m[1], m[2]) for m in ref.merges],
                                                                    len(ref.cams[fb].model), len(ref.ferns[fb].frames), t1 - t0, time.time() - t1), flush=True)
    s.close()
