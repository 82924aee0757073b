"""The cluster rules of GlobalModel as the oracle restates them (oracle/orc_pipeline.MapRef; GlobalModel.cpp:56-59, :251-277, :350,
:398): cheap CPU checks of the bookkeeping the GPU test compares the product with."""
import numpy as np


def test_cluster_bookkeeping(orc):
    from oracle import orc_pipeline

    m = orc_pipeline.MapRef()
    assert m.isCluster(0) and not m.isCluster(1) and m.current == 0 and len(m.model) == 0
    a = np.zeros(3, orc.SURFEL_DTYPE)
    m.initialise(a, 0)  # the first frame under the constructor's id: same buffers
    assert sorted(m.clusters) == [0] and len(m.model) == 3
    b = np.zeros(5, orc.SURFEL_DTYPE)
    m.initialise(b, 4)  # unknown id: new buffers, current
    assert sorted(m.clusters) == [0, 4] and m.current == 4 and len(m.model) == 5 and len(m.clusters[0]) == 3
    m.model = np.zeros(6, orc.SURFEL_DTYPE)  # fuse / clean write the current cluster
    assert len(m.clusters[4]) == 6 and len(m.clusters[0]) == 3
    m.initialise(np.zeros(1, orc.SURFEL_DTYPE), 0)  # a known id does not become current (:350): current's buffers are rewritten
    assert m.current == 4 and len(m.clusters[4]) == 1 and len(m.clusters[0]) == 3
