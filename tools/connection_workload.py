"""A copy-constraint workload of any size for the reference's connection PIL (tests/golden/starky_data/connection.pil.json:
`{a, b, c} connect {S1, S2, S3}`, starkjs/connection/connection.pil): the grand product of `calculate_Z`
(stark_gen.rs:653-666) then runs over 2^nbits rows -- far past one scan block.  Input generation only.

Cells (row j, column i) carry the identifier k_i * w^j with k_0 = 1, k_1 = k, k_2 = k^2 (helper.rs:16-23,
starkinfo_Z.rs:273-423).  Every cell gets a value class; sigma rotates the cells of a class, S_i[j] = id(sigma(j, i)), and the
trace holds one value per class, so the argument closes."""
import copy
import json
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
P = 0xFFFFFFFF00000001
K = 12275445934081160404


def pil(nbits):
    d = json.load(open(ROOT / "tests" / "golden" / "starky_data" / "connection.pil.json"))
    d = copy.deepcopy(d)
    for r in d["references"].values():
        r["polDeg"] = 1 << nbits
    return d


def make(nbits, root_of_unity, seed=7, n_classes=None):
    """-> (const [N][4] = L1, S1, S2, S3 ; cm [N][3] = a, b, c), flat uint64.  root_of_unity = MG.0[nbits]"""
    N = 1 << nbits
    rng = np.random.default_rng(seed)
    n_classes = n_classes or max(2, N // 2)
    cls = rng.integers(0, n_classes, size=3 * N)                       # cell = 3 * row + column
    order = np.argsort(cls, kind="stable")
    sc = cls[order]
    first = np.r_[True, sc[1:] != sc[:-1]]                             # group starts in sorted order
    start = np.maximum.accumulate(np.where(first, np.arange(3 * N), 0))
    nxt = np.r_[order[1:], order[:1]]
    last = np.r_[first[1:], True]
    succ = np.where(last, order[start], nxt)
    sigma = np.empty(3 * N, np.int64); sigma[order] = succ
    w = np.empty(N, dtype=object); v = 1
    for j in range(N):
        w[j] = v; v = v * root_of_unity % P
    ks = [1, K, K * K % P]
    ident = np.empty(3 * N, dtype=object)
    for i in range(3):
        ident[i::3] = (w * ks[i]) % P
    const = np.zeros((N, 4), np.uint64)
    const[0, 0] = 1
    const[:, 1:] = ident[sigma].astype(np.uint64).reshape(N, 3)
    val = (cls.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) % np.uint64(P)   # wraps mod 2^64 first: still one value per class
    return const.reshape(-1), val.reshape(-1).astype(np.uint64)
