"""Portable synthetic inputs for the S-table / hyper-parameter sweep path.

Discount grids and (n, t) occupancy groups are generated with splitmix64 so that bench.py, the
tests and the golden-fixture generator see the same bytes on every machine (glibc's drand48 is
deliberately not used for data).  Shapes follow what `samplea`/`sampleb` take
(reference lib/psample.h:104-117): ragged n[i][k] (uint32), t[i][k] (uint16), per-restaurant
totals T[i], N[i] and concentrations bpar[i]; here the ragged arrays are flat CSR
(`K[i]` pairs for restaurant i, back to back).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SEED = 0x5EEDB001
_MASK = (1 << 64) - 1


def splitmix64(n: int, seed: int = SEED) -> np.ndarray:
    """n successive splitmix64 outputs (uint64) for the given seed."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def unit(n: int, seed: int = SEED) -> np.ndarray:
    """n doubles in [0,1) from the top 53 bits of splitmix64."""
    return (splitmix64(n, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def discount_grid(D: int = 64) -> np.ndarray:
    """a_d = 0.05 + 0.90 (d+0.5)/D, d=0..D-1: the open interval (0.05, 0.95)."""
    d = np.arange(D, dtype=np.float64)
    return 0.05 + 0.90 * (d + 0.5) / D


@dataclass
class Groups:
    I: int
    K: np.ndarray  # int32[I]   pairs per restaurant
    n: np.ndarray  # uint32[sum K]
    t: np.ndarray  # uint16[sum K]
    T: np.ndarray  # uint32[I]  sum_k t
    N: np.ndarray  # uint32[I]  sum_k n
    bpar: np.ndarray  # float64[I]
    shape: float = 1.1
    scale: float = 20.0

    @property
    def pairs(self) -> int:
        return int(self.n.shape[0])


def groups(I: int, K: int, n_max: int = 4000, profile: str = "wide", seed: int = SEED,
           bpar: float = 10.0) -> Groups:
    """I restaurants x K pairs each.  n = 2 + floor(u1 (n_max-2)); t = 1 + floor(u2 n) ("wide",
    forces M ~ N) or 1 + floor(u2 sqrt(n)) ("realistic")."""
    G = I * K
    u = unit(2 * G, seed)
    u1, u2 = u[0::2], u[1::2]
    n = (2 + np.floor(u1 * (n_max - 2))).astype(np.uint32)
    if profile == "wide":
        t = 1 + np.floor(u2 * n)
    elif profile == "realistic":
        t = 1 + np.floor(u2 * np.sqrt(n.astype(np.float64)))
    else:
        raise ValueError(profile)
    t = np.minimum(t, n).astype(np.uint16)
    Kv = np.full(I, K, dtype=np.int32)
    T = t.reshape(I, K).astype(np.uint64).sum(axis=1).astype(np.uint32)
    N = n.reshape(I, K).astype(np.uint64).sum(axis=1).astype(np.uint32)
    return Groups(I=I, K=Kv, n=n, t=t, T=T, N=N, bpar=np.full(I, bpar, dtype=np.float64))


def cells(N: int, M: int) -> int:
    """Stored cells of an S table with bounds (N, M): sum_{n=3..N} min(n-2, M-1)."""
    if N < 3:
        return 0
    if N <= M + 1:
        k = N - 2
        return k * (k + 1) // 2
    return (M - 1) * M // 2 + (N - M - 1) * (M - 1)
