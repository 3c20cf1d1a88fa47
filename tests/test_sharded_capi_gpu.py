"""vdb_sharded_search_batch_device (include/vdb_index.h) through the C-ABI with a one-rank RCCL
communicator: shard search + ncclAllGather + merge must equal the shard's own search (a merge of one
list is the list) bit for bit, and the torch.distributed spelling of the same step.  More ranks need
more GPUs: the exchange itself is covered on CPU by tests/test_sharded_gloo.py and on a node by
bench.py --gpus N."""
import numpy as np
import pytest

from tests.helpers import build_pair
from vectordb_amd import datasets, sharding
from vectordb_amd import index as vi

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_equals_own_search(orc):
    import torch
    x = datasets.clustered(6000, 32, 95)
    q = datasets.clustered(300, 32, 96)
    labels = np.arange(6000, dtype=np.uint64) + 50_000
    h, o = build_pair(orc, x, vi.L2, 16, 100, labels=labels, n_threads=4)
    h.to_device(0)
    h.setefSearch(40)
    o.set_ef(40)
    comm = sharding.Communicator(sharding.Communicator.unique_id(), 1, 0, 0)
    assert (comm.n_ranks, comm.rank) == (1, 0)
    dq = torch.from_numpy(q).cuda()
    for k, nq in ((10, 300), (100, 64)):           # the second call is smaller in nq and larger in k: buffers are reused / regrown
        h.setefSearch(max(40, k))
        o.set_ef(max(40, k))
        d0, l0 = h.search_batch(dq[:nq], k)
        d1, l1 = comm.sharded_search(h, dq[:nq], k)
        md, ml = vi.merge_topk(d0.unsqueeze(0), l0.unsqueeze(0))
        torch.cuda.synchronize()
        for d, l in ((d1, l1), (md, ml)):
            np.testing.assert_array_equal(l.cpu().numpy(), l0.cpu().numpy())
            np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), d0.cpu().numpy().view(np.uint32))
        ref = o.search(q[:nq], k)
        np.testing.assert_array_equal(l1.cpu().numpy().view(np.uint64), ref["label"].astype(np.uint64))
    with pytest.raises(vi.VdbError):               # n_ranks * k beyond what merge_topk holds in LDS
        comm.sharded_search(h, dq[:4], 6000)
