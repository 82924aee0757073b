"""The restatement's pyramid / preparation operators and NID scores (oracle/orc_track.c, oracle/orc_nid.py)
against what the REFERENCE's own Cuda/cudafuncs.cu returned on an MI355X (tests/golden/ref_cudafuncs.npz, recorded by
tests/golden/make_ref_cudafuncs_golden.py from oracle/_ref/libref_cudafuncs.so).  This is what makes the oracle for those
rows a pinned one: every operator bit for bit, the two that call rsqrtf to its instruction's accuracy (tests/ref_cases_cf.py)."""
import os

import numpy as np
import pytest

from tests import ref_cases_cf as cf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cudafuncs.npz")


def load_case(case):
    z = np.load(GOLDEN)
    return {k[len(case) + 1:]: z[k] for k in z.files if k.startswith(case + "_")}


@pytest.mark.parametrize("case", ["small", "full"])
def test_restatement_equals_the_references_cudafuncs(orc, case):
    fx = load_case(case)
    pair = np.load(os.path.join(os.path.dirname(GOLDEN), "gputest_pair.npz"))
    inp = cf.inputs(case, pair, orc)
    for k, v in cf.input_hashes(inp).items():
        assert str(v) == str(fx[k]), "input %s is not what the reference saw" % k
    fed = case == "small"
    out = cf.chain(cf.OrcOps(orc), inp, feed=fx if fed else None)
    kinds = {}
    for name, got in out.items():
        if name == "nid":
            continue
        kinds[name] = cf.compare(name, got, fx, cf.tol_of(name, fed))
    # everything the tolerance does not cover is the same bits
    for name, kind in kinds.items():
        if cf.tol_of(name, fed) == 0.0:
            assert kind == "exact", name
    assert len(kinds) >= 40
    # NID: integer histograms on the device, then the reference's host loop (float accumulators fed with double terms,
    # cudafuncs.cu:1556-1612) restated in oracle/orc_nid.c: the same bits for all ten scores
    assert out["nid"].tobytes() == np.asarray(fx["nid"], np.float32).tobytes(), (out["nid"], fx["nid"])


def test_fixture_is_the_references():
    z = np.load(GOLDEN)
    assert "cudafuncs.cu" in str(z["meta"]) and "MI355X" in str(z["meta"])
    # the small case holds complete arrays (operators compared in isolation), the full case hashes + samples
    assert "small_vmap_L0" in z.files and "full_vmap_L0__sha" in z.files
