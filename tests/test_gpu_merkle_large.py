"""Parity at the sizes only the prover's real workload reaches (BASELINE config "2^24": the extended stage-3 section is a
2^25-row x 36-word matrix, 9 GiB, word offsets past 2^30): sampled leaves and their whole paths against the oracle's
recomputation (merklehash.rs:293-346 merkelize, :64-76 / :430-438 group proofs, linearhash.rs:79-145)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


SEED = 0x25A36


def _rows(lo, hi, width):
    """rows lo..hi-1 of the test matrix as zk_dev_fill_splitmix writes it: word i = splitmix64(SEED + i), minus p when >= p"""
    M = (1 << 64) - 1
    out = []
    for i in range(lo * width, hi * width):
        z = (SEED + i + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M; z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M; z ^= z >> 31
        out.append(z - P if z >= P else z)
    return np.array(out, dtype=np.uint64)


@pytest.mark.parametrize("log_h,width", [(25, 36)])
def test_sampled_leaves_and_paths_of_a_9_gib_tree(zk, orc, log_h, width):
    assert zk.lib().zk_device_count() >= 1, "no GPU visible (the product has no CPU fallback)"
    zk.init(0)
    h = 1 << log_h
    d = zk.DevArray(h * width)
    assert zk.lib().zk_dev_fill_splitmix(d.ptr, h * width, SEED, None) == 0          # 9 GiB born in HBM: no host copy of the matrix exists
    head = np.zeros(3 * width, np.uint64); assert zk.lib().zk_dev_download(head.ctypes.data, d.ptr, head.nbytes) == 0
    assert np.array_equal(head, _rows(0, 3, width))                                  # the host restatement of the generator is the device's
    t = zk.MerkleTreeGL(); t.merkelize_dev(d.ptr, width, h)
    root = [int(v) for v in t.root()]
    rng = np.random.default_rng(25)
    idx = sorted({0, 1, h - 1, h - 2, h // 2, h // 2 - 1, (1 << 31) // (8 * width) + 1, (1 << 32) // (8 * width) + 3} | {int(v) for v in rng.integers(0, h, 64)})
    assert len(idx) >= 64
    for i in idx:
        row, path = t.get_group_proof(i)
        want_row = _rows(i, i + 1, width)
        assert np.array_equal(np.asarray(row, np.uint64), want_row), i                        # the row the tree opens is the row that was committed
        assert path.reshape(-1, 4).shape[0] == log_h
        leaf = orc.linearhash(want_row)                                                       # oracle: the leaf digest ...
        got = orc.root_from_proof(want_row, np.asarray(path, np.uint64).reshape(-1), i)         # ... and the walk to the root over the device's siblings
        assert [int(v) for v in got] == root, i
        # the sibling at level 0 is itself a leaf digest: recompute it from the neighbouring row
        sib = orc.linearhash(_rows(i ^ 1, (i ^ 1) + 1, width))
        assert [int(v) for v in np.asarray(path, np.uint64).reshape(-1, 4)[0]] == [int(v) for v in sib], i
        assert [int(v) for v in leaf] != [int(v) for v in sib]
    # the batched opening (what a proof uses) returns the same words as the single ones
    many = t.get_group_proofs(idx[:16])
    for i, (row, path) in zip(idx[:16], many):
        r1, p1 = t.get_group_proof(i)
        assert np.array_equal(row, r1) and np.array_equal(np.asarray(path), np.asarray(p1))
    t.free(); d.free()
