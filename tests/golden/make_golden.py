#!/usr/bin/env python3
"""Generates tests/golden/*.npz and hnsw_256x16.bin with the CPU oracle (oracle/).

The reference holds NO golden vectors, tests or fixtures for its HNSW / brute-force path
 and its index classes do not compile, so these fixtures pin the oracle
against ITSELF over time (a regression net), not against the reference: parity stays
"unpinned by the reference" except for the fragments checked in test_oracle_ref_fragments.py.

    python tests/golden/make_golden.py        # rewrites the fixtures
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402


def inputs():
    rng = np.random.default_rng(20250530)
    x = rng.random((256, 16), dtype=np.float32)
    q = rng.random((32, 16), dtype=np.float32)
    return x, q


def main():
    orc.build()
    x, q = inputs()
    out = {"x": x, "q": q}
    for mode, tag in ((orc.SEQ, "seq"), (orc.TREE64, "tree64")):
        o = orc.OracleHNSW(orc.L2, 16, 256, 4, 32, 100, mode=mode)
        o.add(x)
        if mode == orc.TREE64:
            o.save(os.path.join(HERE, "hnsw_256x16.bin"))
        out[f"{tag}_levels"] = np.array([o.level_of(i) for i in range(256)], np.int32)
        out[f"{tag}_entry"] = np.array([o.entry, o.max_level], np.int64)
        for ef in (8, 16, 64):
            for k in (1, 10):
                o.set_ef(ef)
                r = o.search(q, k)
                for name in ("dist", "label", "ndis", "nhops0", "nhops_upper"):
                    out[f"{tag}_ef{ef}_k{k}_{name}"] = r[name]
        b = orc.OracleBruteforce(orc.L2, 16, 256, mode=mode)
        b.add(x)
        r = b.search(q, 10)
        out[f"{tag}_bf_dist"], out[f"{tag}_bf_label"] = r["dist"], r["label"]
    # non-bare path (hnsw_core.hpp:1141-1148): every 5th element deleted, functor label % 3 == 1
    o = orc.OracleHNSW(orc.L2, 16, 256, 4, 32, 100, mode=orc.TREE64)
    o.add(x)
    for lab in range(0, 256, 5):
        o.mark_delete(lab)
    for ef, k in ((16, 5), (64, 10)):
        o.set_ef(ef)
        for fm, tagf in ((0, "del"), (3, "delmod3")):
            r = o.search(q, k, filter_mod=fm, filter_keep=1)
            for name in ("dist", "label", "count", "ndis", "nhops0", "nhops_upper"):
                out[f"nonbare_{tagf}_ef{ef}_k{k}_{name}"] = r[name]
    # inner product, ragged dim
    rng = np.random.default_rng(7)
    xi = rng.standard_normal((200, 24)).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True)
    qi = rng.standard_normal((16, 24)).astype(np.float32)
    o = orc.OracleHNSW(orc.IP, 24, 200, 8, 40, 100, mode=orc.TREE64)
    o.add(xi)
    o.set_ef(20)
    r = o.search(qi, 5)
    out.update(ip_x=xi, ip_q=qi, ip_dist=r["dist"], ip_label=r["label"])
    # duplicate vectors: exact ties, labels descending
    base = np.random.default_rng(5).random((20, 8), dtype=np.float32)
    xt = np.concatenate([base] * 6)
    lt = (1000 - np.arange(xt.shape[0])).astype(np.uint64)
    o = orc.OracleHNSW(orc.L2, 8, xt.shape[0], 4, 16, 100, mode=orc.TREE64)
    o.add(xt, lt)
    o.set_ef(12)
    r = o.search(base, 8)
    out.update(tie_x=xt, tie_labels=lt, tie_q=base, tie_dist=r["dist"], tie_label=r["label"])
    b = orc.OracleBruteforce(orc.L2, 8, xt.shape[0], mode=orc.TREE64)
    b.add(xt, lt)
    r = b.search(base, 8)
    out.update(tie_bf_dist=r["dist"], tie_bf_label=r["label"])
    np.savez_compressed(os.path.join(HERE, "hnsw_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
