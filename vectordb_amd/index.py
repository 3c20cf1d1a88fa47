"""Host-side mirror of the reference's index interface over the C-ABI.

Mirrors ``AlgorithmInterface<float>`` (reference src/hnsw/hnsw_scratch/hnswlib.hpp:71-85)
and the public surface of ``HierarchicalNSW`` (hnsw_core.hpp:21, :110-123, :207-256, :709-782,
:1107) and ``BruteforceSearch`` (bruteforce.hpp:14): same method names, argument meaning and
error behaviour (the reference throws std::runtime_error; here ``VdbError``).  The batched
entry points are the real ones — a single-query ``searchKnn`` is a batch of one.

Graph building runs on the host with the reference's algorithm; every search runs the
gfx950 kernels through ``libvdb_amd.so``.  Nothing here computes a distance on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import BF_BF16, BF_BF16_RERANK, BF_FP32, IP, L2, SUM_SEQ, SUM_TREE64, VdbError, check  # noqa: F401

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def _as_f32(a, dim: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2 or a.shape[1] != dim:
        raise ValueError(f"expected [n, {dim}] float32, got {a.shape}")
    return a


def _is_torch_cuda(x) -> bool:
    return hasattr(x, "is_cuda") and bool(x.is_cuda)


class _Space:
    """SpaceInterface (hnswlib.hpp:54-61): data size and dimension of the float space."""

    metric = L2

    def __init__(self, dim: int):
        self.dim = int(dim)

    def get_data_size(self) -> int:
        return self.dim * 4

    def get_dim(self) -> int:
        return self.dim


class L2Space(_Space):  # space_l2.hpp:16-37
    metric = L2


class InnerProductSpace(_Space):  # space_ip.hpp:20-41
    metric = IP


class HierarchicalNSW:
    """HierarchicalNSW<float> (hnsw_core.hpp:21)."""

    def __init__(self, space: _Space, capacity: int = 0, M: int = 16, ef_construction: int = 200,
                 random_seed: int = 100, reuse_deleted: bool = False, *, location: str | None = None,
                 stream_to_device: int | None = None):
        """``location``: the load constructor.  With ``stream_to_device`` = a device ordinal the file's level-0 block is
        streamed to that GPU without a host copy (vdb_index_load_to_device): the index searches, but cannot be edited or saved."""
        self._l = _lib.lib()
        self.space = space
        self._h = C.c_void_p()
        if location is not None and stream_to_device is not None:
            check(self._l.vdb_index_load_to_device(C.byref(self._h), location.encode(), space.metric, space.dim, int(stream_to_device)))
        elif location is not None:  # load ctor, hnsw_core.hpp:110-116
            check(self._l.vdb_index_load(C.byref(self._h), location.encode(), space.metric, space.dim, capacity))
        else:
            check(self._l.vdb_index_create(C.byref(self._h), space.metric, space.dim, capacity, M, ef_construction,
                                           random_seed))
        self._device = stream_to_device if location is not None else None
        if reuse_deleted:  # hnsw_core.hpp:114, :123
            check(self._l.vdb_index_set_reuse_deleted(self._h, 1))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._l.vdb_index_free(h)
            self._h = None

    # ---- building (host, reference algorithm) -------------------------------------
    def addPoint(self, datapoint, label: int, replace_deleted: bool = False) -> None:
        """hnsw_core.hpp:804-842: inserts, updates the element when the label exists (:1001-1016), or - with
        ``replace_deleted`` - takes the place of a deleted element (:815-841)."""
        v = _as_f32(datapoint, self.space.dim)
        check(self._l.vdb_index_add_point(self._h, v.ctypes.data_as(_f32p), int(label), int(bool(replace_deleted))))

    def add_items(self, vecs, labels=None, n_threads: int = 1) -> None:
        v = _as_f32(vecs, self.space.dim)
        lp = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, dtype=np.uint64)
            if labels.shape != (v.shape[0],):
                raise ValueError("labels must be [n]")
            lp = labels.ctypes.data_as(_u64p)
        check(self._l.vdb_index_add_batch(self._h, v.ctypes.data_as(_f32p), lp, v.shape[0], n_threads))

    def add_items_device(self, vecs, labels=None, n_threads: int = 1, device: int = 0) -> None:
        """add_items with layer 0 built on the GPU (include/vdb_index.h, vdb_index_add_batch_device);
        the index is left uploaded to ``device``."""
        v = _as_f32(vecs, self.space.dim)
        lab = None if labels is None else np.ascontiguousarray(labels, np.uint64)
        if lab is not None and lab.shape != (v.shape[0],):
            raise ValueError("one label per vector")
        check(self._l.vdb_index_add_batch_device(self._h, v.ctypes.data_as(_f32p), None if lab is None else lab.ctypes.data_as(_u64p),
                                                 v.shape[0], n_threads, device))
        self._device = device

    def set_build_sum_order(self, order: int) -> None:
        check(self._l.vdb_index_set_build_sum_order(self._h, order))

    def saveIndex(self, location: str) -> None:
        check(self._l.vdb_index_save(self._h, location.encode()))

    def setefSearch(self, ef: int) -> None:
        check(self._l.vdb_index_set_ef(self._h, ef))

    def set_spill_mode(self, mode: int) -> None:
        """-1 adaptive (default), 0 never, 1 always: second-level visited tables in HBM (vdb_index_set_spill_mode)."""
        check(self._l.vdb_index_set_spill_mode(self._h, mode))

    def markDelete(self, label: int) -> None:
        check(self._l.vdb_index_mark_delete(self._h, label))

    def unmarkDelete(self, label: int) -> None:
        check(self._l.vdb_index_unmark_delete(self._h, label))

    def getDataByLabel(self, label: int) -> np.ndarray:  # hnsw_core.hpp:683-703
        out = np.empty(self.space.dim, np.float32)
        check(self._l.vdb_index_get_data_by_label(self._h, int(label), out.ctypes.data_as(_f32p)))
        return out

    def resizeIndex(self, new_capacity: int) -> None:
        check(self._l.vdb_index_resize(self._h, new_capacity))

    def checkIntegrity(self):
        mn, mx = C.c_int(0), C.c_int(0)
        rc = self._l.vdb_index_check_integrity(self._h, C.byref(mn), C.byref(mx))
        return rc, mn.value, mx.value

    def getCapacity(self) -> int:
        return self._l.vdb_index_capacity(self._h)

    def getElementCount(self) -> int:
        return self._l.vdb_index_count(self._h)

    def getDeletedCount(self) -> int:
        return self._l.vdb_index_deleted_count(self._h)

    max_level = property(lambda s: s._l.vdb_index_max_level(s._h))
    entry = property(lambda s: s._l.vdb_index_entry(s._h))
    ef = property(lambda s: s._l.vdb_index_ef(s._h))
    M = property(lambda s: s._l.vdb_index_M(s._h))

    def level_of(self, i: int) -> int:
        return self._l.vdb_index_level_of(self._h, i)

    def links(self, i: int, level: int) -> list[int]:
        buf = (C.c_uint32 * 256)()
        n = self._l.vdb_index_links(self._h, i, level, buf, 256)
        return list(buf[:max(n, 0)])

    # ---- device ---------------------------------------------------------------------
    def to_device(self, device: int = 0) -> "HierarchicalNSW":
        check(self._l.vdb_index_to_device(self._h, device))
        self._device = device
        return self

    def device_bytes(self) -> int:
        return self._l.vdb_index_device_bytes(self._h)

    # ---- searching (gfx950 kernels) ----------------------------------------------
    def labels(self) -> np.ndarray:
        """External labels in internal-id order (getExternalLabel, hnsw_core.hpp:219-221)."""
        out = np.empty(self.getElementCount(), np.uint64)
        check(self._l.vdb_index_labels(self._h, out.ctypes.data_as(_u64p), out.size))
        return out

    def filter_bitmap(self, filter) -> np.ndarray:
        """A BaseFilterFunctor's verdicts (hnswlib.hpp:10-23) as the bitmap over internal ids the C-ABI
        takes.  ``filter``: a callable label -> bool, or a boolean array indexed by internal id."""
        if callable(filter):
            ok = np.fromiter((bool(filter(int(l))) for l in self.labels()), bool, self.getElementCount())
        else:
            ok = np.asarray(filter, bool)
            if ok.shape != (self.getElementCount(),):
                raise ValueError("boolean filter must have one entry per element")
        return np.packbits(ok, bitorder="little").tobytes().ljust((ok.size + 31) // 32 * 4, b"\0")

    def search_batch(self, queries, k: int, stats: bool = False, filter=None):
        """SearchKNNCloserFirst for every row of `queries`.

        numpy in -> numpy out (PCIe-inclusive); torch CUDA tensor in -> torch CUDA tensors
        out, enqueued on torch's current stream with no host synchronisation.
        Returns (dist [nq,k] f32 ascending, label [nq,k] u64/int64) and, with ``stats``,
        a [nq,4] uint32 array {ndis, hops layer 0, hops upper, found}.
        ``filter``: None, a callable label -> bool, a boolean array over internal ids, or (device
        path) an int32 CUDA tensor holding the bitmap of ``filter_bitmap``.
        """
        if _is_torch_cuda(queries):
            import torch

            q = queries.contiguous()
            if q.dtype != torch.float32 or q.dim() != 2 or q.shape[1] != self.space.dim:
                raise ValueError(f"expected [nq, {self.space.dim}] float32 CUDA tensor")
            nq = q.shape[0]
            dist = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            label = torch.empty((nq, k), dtype=torch.int64, device=q.device)  # same bits as u64
            st = torch.empty((nq, 4), dtype=torch.int32, device=q.device) if stats else None
            stream = torch.cuda.current_stream(q.device).cuda_stream
            allow = None
            if filter is not None:
                if _is_torch_cuda(filter):
                    allow = filter.contiguous()
                    if allow.dtype != torch.int32 or allow.numel() < (self.getElementCount() + 31) // 32:
                        raise ValueError("device filter must be an int32 bitmap with ceil(count / 32) words")
                else:
                    allow = torch.frombuffer(bytearray(self.filter_bitmap(filter)), dtype=torch.int32).to(q.device)
            check(self._l.vdb_search_batch_device_filtered(self._h, q.data_ptr(), nq, k,
                                                           allow.data_ptr() if allow is not None else None,
                                                           dist.data_ptr(), label.data_ptr(),
                                                           st.data_ptr() if stats else None, stream))
            return (dist, label, st) if stats else (dist, label)
        q = _as_f32(queries, self.space.dim)
        nq = q.shape[0]
        dist = np.empty((nq, k), np.float32)
        label = np.empty((nq, k), np.uint64)
        st = np.empty((nq, 4), np.uint32) if stats else None
        allow = np.frombuffer(self.filter_bitmap(filter), np.uint32) if filter is not None else None
        check(self._l.vdb_search_batch_filtered(self._h, q.ctypes.data_as(_f32p), nq, k,
                                                allow.ctypes.data_as(_u32p) if allow is not None else None,
                                                dist.ctypes.data_as(_f32p), label.ctypes.data_as(_u64p),
                                                st.ctypes.data_as(_u32p) if stats else None))
        return (dist, label, st) if stats else (dist, label)

    def search_stop_condition_closest_batch(self, queries, epsilon: float, min_num_candidates: int, max_num_candidates: int,
                                            filter=None):
        """searchStopConditionClosest (hnsw_core.hpp:1162-1213) with the epsilon-radius stop condition, for every row of
        `queries` (numpy, host path).  Returns (dist [nq, max], label [nq, max], count [nq], stats [nq, 4]): row q holds
        count[q] results within `epsilon`, closest first, padded with +inf / UINT64_MAX."""
        q = _as_f32(queries, self.space.dim)
        nq, k = q.shape[0], int(max_num_candidates)
        dist = np.empty((nq, k), np.float32)
        label = np.empty((nq, k), np.uint64)
        st = np.empty((nq, 4), np.uint32)
        allow = np.frombuffer(self.filter_bitmap(filter), np.uint32) if filter is not None else None
        check(self._l.vdb_search_epsilon_batch(self._h, q.ctypes.data_as(_f32p), nq, float(epsilon), int(min_num_candidates), k,
                                               allow.ctypes.data_as(_u32p) if allow is not None else None,
                                               dist.ctypes.data_as(_f32p), label.ctypes.data_as(_u64p), st.ctypes.data_as(_u32p)))
        return dist, label, st[:, 3].copy(), st

    def search_batch_noalloc(self, queries, dist, label, stats=None) -> None:
        """vdb_search_batch_device_noalloc: CUDA tensors in, results into the caller's CUDA tensors
        (dist [nq,k] f32, label [nq,k] int64, stats nullable [nq,4] int32); nothing is allocated."""
        nq, k = dist.shape
        if queries.shape != (nq, self.space.dim) or label.shape != (nq, k) or not (queries.is_contiguous() and dist.is_contiguous() and label.is_contiguous()):
            raise ValueError("search_batch_noalloc: contiguous [nq, dim] queries, [nq, k] dist and label")
        import torch

        stream = torch.cuda.current_stream(queries.device).cuda_stream
        check(self._l.vdb_search_batch_device_noalloc(self._h, queries.data_ptr(), nq, k, dist.data_ptr(), label.data_ptr(),
                                                      stats.data_ptr() if stats is not None else None, stream))

    def SearchKNNCloserFirst(self, query, k: int, filter=None):  # hnswlib.hpp:87-99
        d, l = self.search_batch(np.asarray(query, np.float32).reshape(1, -1), k, filter=filter)
        keep = l[0] != np.uint64(0xFFFFFFFFFFFFFFFF)
        return [(float(a), int(b)) for a, b in zip(d[0][keep], l[0][keep])]

    def searchKnn(self, query, k: int, filter=None):  # hnsw_core.hpp:1107; farthest first, like draining the heap
        return list(reversed(self.SearchKNNCloserFirst(query, k, filter)))

    SearchKNN = searchKnn  # the interface's spelling, hnswlib.hpp:75


class BruteforceSearch:
    """BruteforceSearch<float> (bruteforce.hpp:14)."""

    def __init__(self, space: _Space, capacity: int = 0, *, location: str | None = None):
        self._l = _lib.lib()
        self.space = space
        self._h = C.c_void_p()
        if location is not None:
            check(self._l.vdb_bf_load(C.byref(self._h), location.encode(), space.metric, space.dim))
        else:
            check(self._l.vdb_bf_create(C.byref(self._h), space.metric, space.dim, capacity))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._l.vdb_bf_free(h)
            self._h = None

    def addPoint(self, datapoint, label: int, replace_deleted: bool = False) -> None:
        self.add_items(np.asarray(datapoint, np.float32).reshape(1, -1), np.array([label], np.uint64))

    def add_items(self, vecs, labels=None) -> None:
        v = _as_f32(vecs, self.space.dim)
        lp = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, dtype=np.uint64)
            lp = labels.ctypes.data_as(_u64p)
        check(self._l.vdb_bf_add_batch(self._h, v.ctypes.data_as(_f32p), lp, v.shape[0]))

    def removePoint(self, label: int) -> None:
        check(self._l.vdb_bf_remove(self._h, label))

    def saveIndex(self, location: str) -> None:
        check(self._l.vdb_bf_save(self._h, location.encode()))

    def getElementCount(self) -> int:
        return self._l.vdb_bf_count(self._h)

    def to_device(self, device: int = 0) -> "BruteforceSearch":
        check(self._l.vdb_bf_to_device(self._h, device))
        return self

    def labels(self) -> np.ndarray:
        """Labels of the stored rows in storage order (vdb_bf_labels)."""
        out = np.empty(self.getElementCount(), np.uint64)
        check(self._l.vdb_bf_labels(self._h, out.ctypes.data_as(_u64p), out.size))
        return out

    def filter_bitmap(self, filter) -> np.ndarray:
        """A BaseFilterFunctor's verdicts as the bitmap over rows the C-ABI takes: ``filter`` is a callable
        label -> bool or a boolean array over the rows in storage order."""
        n = self.getElementCount()
        ok = np.fromiter((bool(filter(int(l))) for l in self.labels()), bool, n) if callable(filter) else np.asarray(filter, bool)
        if ok.shape != (n,):
            raise ValueError("boolean filter must have one entry per row")
        return np.frombuffer(np.packbits(ok, bitorder="little").tobytes().ljust((n + 31) // 32 * 4, b"\0"), np.uint32)

    def search_batch(self, queries, k: int, precision: int = BF_FP32, filter=None):
        """``filter`` (callable, boolean array over rows, or on the device path an int32 CUDA bitmap): the
        reference's filtered scan, always exact (vdb_bf_search_batch[_device]_filtered)."""
        if _is_torch_cuda(queries):
            import torch

            q = queries.contiguous()
            nq = q.shape[0]
            dist = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            label = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            stream = torch.cuda.current_stream(q.device).cuda_stream
            if filter is not None:
                allow = filter if _is_torch_cuda(filter) else torch.from_numpy(self.filter_bitmap(filter).view(np.int32).copy()).to(q.device)
                check(self._l.vdb_bf_search_batch_device_filtered(self._h, q.data_ptr(), nq, k, allow.data_ptr(), dist.data_ptr(),
                                                                  label.data_ptr(), stream))
                return dist, label
            check(self._l.vdb_bf_search_batch_device(self._h, q.data_ptr(), nq, k, precision, dist.data_ptr(),
                                                     label.data_ptr(), stream))
            return dist, label
        q = _as_f32(queries, self.space.dim)
        nq = q.shape[0]
        dist = np.empty((nq, k), np.float32)
        label = np.empty((nq, k), np.uint64)
        if filter is not None:
            allow = self.filter_bitmap(filter)
            check(self._l.vdb_bf_search_batch_filtered(self._h, q.ctypes.data_as(_f32p), nq, k, allow.ctypes.data_as(_u32p),
                                                       dist.ctypes.data_as(_f32p), label.ctypes.data_as(_u64p)))
            return dist, label
        check(self._l.vdb_bf_search_batch(self._h, q.ctypes.data_as(_f32p), nq, k, precision,
                                          dist.ctypes.data_as(_f32p), label.ctypes.data_as(_u64p)))
        return dist, label

    def SearchKNNCloserFirst(self, query, k: int, filter=None):
        d, l = self.search_batch(np.asarray(query, np.float32).reshape(1, -1), k, filter=filter)
        keep = l[0] != np.uint64(0xFFFFFFFFFFFFFFFF)
        return [(float(a), int(b)) for a, b in zip(d[0][keep], l[0][keep])]

    def SearchKNN(self, query, k: int, filter=None):
        return list(reversed(self.SearchKNNCloserFirst(query, k, filter)))


def merge_topk(dist, label):
    """Merge per-shard result lists [G, nq, k] (torch CUDA tensors, each row ascending) into the
    k smallest (dist, label) per query, [nq, k], on the device (vdb_merge_topk_device)."""
    import torch

    G, nq, k = dist.shape
    dist = dist.contiguous()
    label = label.contiguous()
    od = torch.empty((nq, k), dtype=torch.float32, device=dist.device)
    ol = torch.empty((nq, k), dtype=torch.int64, device=dist.device)
    stream = torch.cuda.current_stream(dist.device).cuda_stream
    check(_lib.lib().vdb_merge_topk_device(dist.data_ptr(), label.data_ptr(), G, nq, k, od.data_ptr(), ol.data_ptr(),
                                           stream))
    return od, ol
