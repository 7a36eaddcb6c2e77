"""Parity tests proper: the HIP path, called through the C ABI (libzkgpu.so), against the CPU
oracle on the same seeded inputs, against the reference's golden vectors, and -- at BASELINE's
full size -- through size-independent properties.  Bit-exact everywhere (integer field)."""
import ctypes as C
import pathlib
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
GOLD = pathlib.Path(__file__).resolve().parent / "golden"


def _v(x):
    if isinstance(x, str):
        return (int(x, 0) if not x.startswith("-") else (P - int(x[1:], 0))) % P
    return x % P


def _rand(rng, n):
    x = rng.integers(0, P, size=n, dtype=np.uint64)
    # sprinkle edge values
    if n >= 8:
        x[:4] = [0, 1, P - 1, 0xFFFFFFFF]
    return x


@pytest.fixture(scope="module", autouse=True)
def _gpu(zk):
    assert zk.lib().zk_device_count() >= 1, "no GPU visible: the HIP path cannot run (no CPU fallback)"
    zk.init(0)


# ---- NTT ------------------------------------------------------------------------------------
@pytest.mark.parametrize("nbits,n_pols", [(0, 1), (1, 3), (2, 1), (3, 5), (4, 1), (4, 7), (5, 2), (6, 1),
                                          (7, 19), (8, 1), (8, 16), (9, 1), (10, 2), (11, 3), (12, 1),
                                          (13, 17), (15, 1), (16, 4), (17, 1), (18, 2), (20, 1)])
def test_ntt_matches_oracle(zk, orc, nbits, n_pols):
    rng = np.random.default_rng(1000 + nbits * 37 + n_pols)
    x = _rand(rng, (1 << nbits) * n_pols)
    for inverse in (False, True):
        got = zk.ifft(x, n_pols, nbits) if inverse else zk.fft(x, n_pols, nbits)
        exp = orc.ntt(x, n_pols, nbits, inverse)
        assert np.array_equal(got, exp), f"nbits={nbits} n_pols={n_pols} inverse={inverse}"


def test_ntt_rejects_aliasing_and_bad_sizes(zk):
    lib = zk.lib()
    a = np.zeros(16, np.uint64)
    p = a.ctypes.data_as(C.c_void_p)
    assert lib.zk_gl_ntt(p, p, 1, 4, 0) != 0 and b"alias" in lib.zk_last_error()
    assert lib.zk_gl_ntt(p, p, 1, 33, 0) != 0
    assert lib.zk_gl_ntt(p, p, 0, 4, 0) == 0          # zero columns: nothing to do


def test_ntt_2p24_properties_and_oracle(zk, orc):
    """BASELINE config 2: one column, n = 2^24, x_i = splitmix64(seed, i) mod p."""
    import oracle_lib
    nbits = 24
    x = oracle_lib.splitmix64_stream(0x9E3779B97F4A7C15, 1 << nbits)
    X = zk.fft(x, 1, nbits)
    assert np.array_equal(zk.ifft(X, 1, nbits), x)                 # inverse(forward(x)) == x
    assert np.array_equal(X, orc.ntt(x, 1, nbits))                 # forward == CPU restatement
    # linearity: NTT(x + y) == NTT(x) + NTT(y)
    y = oracle_lib.splitmix64_stream(12345, 1 << nbits)
    s = ((x.astype(object) + y.astype(object)) % P).astype(np.uint64)
    Y = zk.fft(y, 1, nbits)
    S = zk.fft(s, 1, nbits)
    assert np.array_equal(S, ((X.astype(object) + Y.astype(object)) % P).astype(np.uint64))
    # X[0] is the plain sum of the inputs
    assert int(X[0]) == int(sum(int(v) for v in x.reshape(-1, 4096).astype(object).sum(axis=1)) % P)


# ---- LDE ------------------------------------------------------------------------------------
@pytest.mark.parametrize("nbits,ext,n_pols", [(0, 1, 1), (1, 2, 2), (2, 3, 3), (3, 4, 1), (3, 5, 2), (4, 5, 1),
                                              (5, 7, 3), (8, 9, 19), (10, 11, 2), (12, 13, 18), (12, 14, 1),
                                              (15, 16, 12), (16, 17, 1), (18, 19, 3),
                                              (16, 18, 5), (17, 20, 2), (13, 16, 7), (20, 21, 4), (9, 12, 36)])   # blow-up 4 / 8 at size, short passes with shift twiddles
def test_lde_matches_oracle(zk, orc, nbits, ext, n_pols):
    rng = np.random.default_rng(5000 + nbits * 31 + ext * 7 + n_pols)
    x = _rand(rng, (1 << nbits) * n_pols)
    assert np.array_equal(zk.interpolate(x, n_pols, nbits, ext), orc.lde(x, n_pols, nbits, ext))


def test_lde_empty_is_noop(zk):
    assert zk.lib().zk_gl_lde(None, 0, 4, None, 5) == 0             # fft_p.rs:262-264


def test_const_root_fib_gl_on_device(zk, golden):
    """stark_setup.rs:100-116: LDE 2^10 -> 2^11 of fib.const.gl, then Merkle root."""
    k = golden["const_root_fib_gl"]
    const = np.fromfile(GOLD / k["const"], dtype="<u8")
    ext = zk.interpolate(const, k["n_pols"], k["nbits"], k["nbits_ext"])
    t = zk.MerkleTreeGL(); t.merkelize(ext, k["n_pols"], 1 << k["nbits_ext"])
    assert [int(v) for v in t.root()] == k["root"]


# ---- Poseidon / LinearHash ------------------------------------------------------------------
def test_poseidon_kats(zk, golden):
    for k in golden["poseidon"]:
        got = zk.poseidon_hash([_v(x) for x in k["in"]], [_v(x) for x in k["cap"]], 4)
        assert [int(v) for v in got] == [_v(x) for x in k["out"]]


def test_poseidon_matches_oracle_and_rejects_bad_lengths(zk, orc):
    rng = np.random.default_rng(3)
    for _ in range(20):
        i, c = _rand(rng, 8), rng.integers(0, P, size=4, dtype=np.uint64)
        assert np.array_equal(zk.poseidon_hash(i, c, 12), orc.poseidon(i, c, 12))
    with pytest.raises(zk.ZkError):
        zk.poseidon_hash([1] * 7, [0] * 4)
    with pytest.raises(zk.ZkError):
        zk.poseidon_hash([1] * 8, [0] * 3)
    with pytest.raises(zk.ZkError):
        zk.poseidon_hash([1] * 8, [0] * 4, 13)


def test_linearhash_kats_and_all_widths(zk, orc, golden):
    for k in golden["linearhash"]:
        lo, hi = k["range"]
        assert [int(v) for v in zk.linearhash(list(range(lo, hi)))] == k["out"]
    rng = np.random.default_rng(4)
    for w in list(range(1, 70)) + [100, 128, 129, 600]:
        v = rng.integers(0, P, size=w, dtype=np.uint64)
        assert np.array_equal(zk.linearhash(v), orc.linearhash(v)), w


# ---- Merkle ---------------------------------------------------------------------------------
def _merkle_input(h, w):
    i = np.arange(h, dtype=np.uint64)[:, None]; j = np.arange(w, dtype=np.uint64)[None, :]
    return (i + 1000 * j).astype(np.uint64).reshape(-1)


def test_merkle_root_kats(zk, golden):
    for k in golden["merkle_root"]:
        t = zk.MerkleTreeGL(); t.merkelize(_merkle_input(k["height"], k["width"]), k["width"], k["height"])
        assert [int(v) for v in t.root()] == k["root"]


def test_merkle_shapes_around_every_kernel_switch(zk, orc):
    """node for node against the oracle across the heights where the builder changes kernels -- the one-launch tree top (64 children),
    the 16-lanes-per-permutation levels, a wave per leaf row below 2^12 rows (its batches on the wave's four groups), 16 lanes per
    leaf row below 2^14 rows, batches of a row side by side up to 2^18 rows -- and
    the widths where LinearHash changes shape (one batch, a short last batch, two sponge steps of digests)"""
    rng = np.random.default_rng(7)
    for height in [2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 257, 4095, 4096, 16383, 16385, 65537, 262144, 262145]:
        for width in [1, 4, 5, 8, 9, 16, 17, 33, 37, 100]:
            if height * width > 12_000_000:
                continue
            buff = rng.integers(0, P, size=height * width, dtype=np.uint64)
            t = zk.MerkleTreeGL(); t.merkelize(buff, width, height)
            assert np.array_equal(t.nodes(), orc.merkelize(buff, width, height)), (height, width)
            t.free()


@pytest.mark.parametrize("height,width", [(1, 1), (1, 9), (2, 3), (3, 5), (33, 6), (255, 2), (256, 9), (1000, 19),
                                          (4096, 12), (1 << 15, 18), (70001, 4), (2, 0), (1024, 0), (33, 0),
                                          (16383, 9), (16384, 9), (20000, 37), (300, 193), (8, 768)])   # leaf hashing: 16 lanes per row below 2^14 rows
def test_merkle_nodes_and_proofs_match_oracle(zk, orc, height, width):
    rng = np.random.default_rng(height * 13 + width)
    buff = rng.integers(0, P, size=height * width, dtype=np.uint64)
    t = zk.MerkleTreeGL(); t.merkelize(buff, width, height)
    exp = orc.merkelize(buff, width, height)
    assert np.array_equal(t.nodes(), exp)
    assert np.array_equal(t.elements(), buff)                            # to_extend (merklehash.rs:260-265)
    for idx in sorted({0, height - 1, height // 2, min(3, height - 1)}):
        row, path = t.get_group_proof(idx)
        assert np.array_equal(row, buff[idx * width:(idx + 1) * width])
        assert np.array_equal(path.reshape(-1), orc.merkle_proof(exp, height, idx))
        if height > 1:   # height 1: the reference's root() is nodes[last] = the zero pad digest (merklehash.rs:455-457)
            assert np.array_equal(orc.root_from_proof(row, path.reshape(-1), idx), t.root())
    with pytest.raises(zk.ZkError, match="access invalid node"):     # merklehash.rs:431-433
        t.get_group_proof(height)
    # every opening of a query list in one round trip (zk_merkle_group_proofs): the same answers, repeats and any order allowed
    qs = [int(v) for v in rng.integers(0, height, size=11)] + [height - 1, 0, 0]
    for (row, path), idx in zip(t.get_group_proofs(qs), qs):
        r1, p1 = t.get_group_proof(idx)
        assert np.array_equal(row, r1) and np.array_equal(path, p1)
    assert t.get_group_proofs([]) == []
    with pytest.raises(zk.ZkError, match="access invalid node"):
        t.get_group_proofs([0, height])
