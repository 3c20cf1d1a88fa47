"""Base-set sharding over the GPUs of one node (no counterpart in the reference,
which has no distributed code at all).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Shard ``s`` of ``G``
holds the contiguous rows ``[s*ceil(N/G), min(N, (s+1)*ceil(N/G)))`` as a complete, independent
HNSW graph (graph seed ``100 + s``) whose labels are the global row ids.  Every rank searches the
same query batch on its shard; the per-rank ``[nq, k]`` result lists are exchanged with ONE
all-gather per tensor (12 bytes per entry: f32 distance + u64 label; 1.2 MB per rank at
nq=10 000, k=10 — latency-bound, not bandwidth-bound, on 7 x 153 GB/s xGMI links) and merged on
every rank by the ``merge_topk`` kernel under the total order (dist, label).
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, world: int, rank: int) -> tuple[int, int]:
    """Rows [lo, hi) of shard `rank`."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def build_shard(base: np.ndarray, world: int, rank: int, metric: int, M: int = 16, ef_construction: int = 200,
                n_threads: int = 1, device_build: int | None = None):
    """HierarchicalNSW over this rank's rows, labels = global row ids, seed 100 + rank.
    ``device_build`` = a device ordinal: layer 0 is linked on that GPU (index.add_items_device)."""
    from . import index as vi

    lo, hi = shard_range(base.shape[0], world, rank)
    dim = base.shape[1]
    space = vi.L2Space(dim) if metric == vi.L2 else vi.InnerProductSpace(dim)
    h = vi.HierarchicalNSW(space, max(hi - lo, 1), M, ef_construction, 100 + rank)
    if hi > lo:
        labels = np.arange(lo, hi, dtype=np.uint64)
        if device_build is None:
            h.add_items(base[lo:hi], labels, n_threads=n_threads)
        else:
            h.add_items_device(base[lo:hi], labels, n_threads=n_threads, device=device_build)
    return h


class Communicator:
    """vdb_comm (include/vdb_index.h): an RCCL communicator over the ranks that hold the shards, made
    from a 128-byte id that rank 0 draws and the launcher ships to the others.  ``Communicator.from_torch``
    ships it through an initialised torch.distributed group (any backend: it is 128 bytes)."""

    def __init__(self, unique_id: bytes, n_ranks: int, rank: int, device: int):
        import ctypes as C

        from . import _lib
        self._l = _lib.lib()
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _lib.check(self._l.vdb_comm_create(C.byref(self._h), buf, n_ranks, rank, device))
        self.n_ranks, self.rank, self.device = n_ranks, rank, device

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().vdb_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch(cls, device: int) -> "Communicator":
        import torch.distributed as td
        box = [cls.unique_id() if td.get_rank() == 0 else None]
        td.broadcast_object_list(box, src=0)
        return cls(box[0], td.get_world_size(), td.get_rank(), device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._l.vdb_comm_free(h)
            self._h = None

    def sharded_search(self, shard_index, queries, k: int):
        """vdb_sharded_search_batch_device: shard search + grouped ncclAllGather + merge, one C call,
        everything on torch's current stream.  ``queries``: the SAME CUDA batch on every rank."""
        import torch

        from . import _lib
        q = queries.contiguous()
        nq = q.shape[0]
        od = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        ol = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        stream = torch.cuda.current_stream(q.device).cuda_stream
        _lib.check(self._l.vdb_sharded_search_batch_device(shard_index._h, self._h, q.data_ptr(), nq, k, od.data_ptr(),
                                                           ol.data_ptr(), stream))
        return od, ol


def all_gather_results(dist, label):
    """[nq, k] per rank -> [world, nq, k] on every rank (torch tensors; RCCL on GPU tensors,
    gloo on CPU tensors in the tests).  One collective per tensor.  This is the torch.distributed spelling
    of the exchange step, kept for the CPU (gloo) tests, for bench.py's per-stage timing and for callers who
    already live in torch; ``Communicator.sharded_search`` is the C-ABI one."""
    import torch
    import torch.distributed as td

    if not (td.is_available() and td.is_initialized()):   # one shard: the "gathered" lists are the rank's own
        return dist.unsqueeze(0), label.unsqueeze(0)
    world = td.get_world_size()
    nq = dist.shape[0]
    if dist.is_cuda and td.get_backend() == "gloo":
        # gloo is the CPU backend: device tensors go through the host (only the one-GPU rehearsal of
        # bench.py and the CPU tests get here; on a GPU node the backend is nccl = RCCL)
        gd, gl = all_gather_results(dist.cpu(), label.cpu())
        return gd.to(dist.device), gl.to(label.device)
    gd = torch.empty((world * nq,) + tuple(dist.shape[1:]), dtype=dist.dtype, device=dist.device)
    gl = torch.empty((world * nq,) + tuple(label.shape[1:]), dtype=label.dtype, device=label.device)
    td.all_gather_into_tensor(gd, dist.contiguous())   # concatenation along dim 0 == [world][nq][k]
    td.all_gather_into_tensor(gl, label.contiguous())
    return gd.view((world,) + tuple(dist.shape)), gl.view((world,) + tuple(label.shape))


def sharded_search(shard_index, queries, k: int, events=None):
    """Search this rank's shard, exchange, merge on the GPU.  `queries` is a CUDA tensor holding
    the SAME batch on every rank.  Returns (dist, label) [nq, k], identical on every rank.
    `events`: four torch.cuda.Event recorded on the current stream around the three stages
    (search | all-gather | merge), for bench.py's split timing."""
    from . import index as vi

    if events is not None:
        events[0].record()
    d, l = shard_index.search_batch(queries, k)
    if events is not None:
        events[1].record()
    gd, gl = all_gather_results(d, l)
    if events is not None:
        events[2].record()
    out = vi.merge_topk(gd, gl)
    if events is not None:
        events[3].record()
    return out
