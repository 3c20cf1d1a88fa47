"""Seeded synthetic inputs for the configurations of BASELINE.json.

There is no network, so SIFT/GIST/Deep are replaced by generators of the same shape:

* ``uniform``   — i.i.d. U[0,1) (config 1, plumbing).
* ``clustered`` — "SIFT-like": Gaussian cluster centres in a low-dimensional latent,
  lifted by a fixed random matrix, plus small isotropic noise, shifted and clipped to be
  non-negative (configs 2, 3, 5).  Optionally L2-normalised rows for the inner-product
  configuration ("Deep-like", config 4).

numpy's PCG64 ``default_rng(seed)`` stream is part of the definition: the same seed gives
the same floats on every machine with this numpy.
"""
from __future__ import annotations

import numpy as np


def uniform(n: int, dim: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).random((n, dim), dtype=np.float32)


class ClusteredLaw:
    """One distribution; base and query sets are separate draws from it."""

    def __init__(self, dim: int, latent: int = 32, n_clusters: int = 1024, law_seed: int = 7,
                 cluster_std: float = 0.35, noise_std: float = 0.05, normalize: bool = False):
        rng = np.random.default_rng(law_seed)
        self.dim, self.latent, self.n_clusters = dim, latent, n_clusters
        self.centres = rng.standard_normal((n_clusters, latent), dtype=np.float32)
        self.lift = (rng.standard_normal((latent, dim), dtype=np.float32) / np.sqrt(latent)).astype(np.float32)
        self.cluster_std, self.noise_std, self.normalize = cluster_std, noise_std, normalize

    def sample(self, n: int, seed: int, chunk: int = 1 << 18) -> np.ndarray:
        rng = np.random.default_rng(seed)
        out = np.empty((n, self.dim), np.float32)
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            which = rng.integers(0, self.n_clusters, m)
            z = self.centres[which] + self.cluster_std * rng.standard_normal((m, self.latent), dtype=np.float32)
            x = z @ self.lift + self.noise_std * rng.standard_normal((m, self.dim), dtype=np.float32)
            x += 2.0
            np.maximum(x, 0.0, out=x)
            if self.normalize:
                x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
            out[lo:lo + m] = x
        return out


def clustered(n: int, dim: int, seed: int, latent: int = 32, normalize: bool = False) -> np.ndarray:
    return ClusteredLaw(dim, latent=latent, normalize=normalize).sample(n, seed)


def recall_at_k(found_labels: np.ndarray, true_labels: np.ndarray) -> float:
    """Mean over queries of |found ∩ true| / k  (both [nq, k])."""
    nq, k = true_labels.shape
    hits = 0
    for f, t in zip(found_labels, true_labels):
        hits += len(set(f.tolist()) & set(t.tolist()))
    return hits / float(nq * k)
