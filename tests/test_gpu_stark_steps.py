"""Parity of the prover glue kernels (transcript, FRI fold/transposition, x/(x-xi), LEv, evals,
Q split, domain tables) -- HIP through the C ABI vs the CPU oracle, bit exact."""
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module", autouse=True)
def _gpu(zk):
    assert zk.lib().zk_device_count() >= 1, "no GPU visible (the product has no CPU fallback)"
    zk.init(0)


def _f3(rng, n):
    return rng.integers(0, P, size=3 * n, dtype=np.uint64)


# ---- transcript (transcript.rs) -------------------------------------------------------------
def test_transcript_matches_oracle_sequence(zk, orc):
    rng = np.random.default_rng(11)
    t, o = zk.TranscriptGL(), orc.transcript()
    for step, n in enumerate([1, 3, 4, 8, 9, 17, 2, 24, 5]):
        v = rng.integers(0, P, size=n, dtype=np.uint64)
        t.put(v); o.put(v)
        if step % 2 == 0:
            assert np.array_equal(t.get_field(), o.get_field())
        if step % 3 == 0:
            assert t.get_fields1() == o.get1()
    assert np.array_equal(t.get_permutations(8, 11), o.get_permutations(8, 11))
    assert np.array_equal(t.get_permutations(64, 21), o.get_permutations(64, 21))
    assert np.array_equal(t.get_permutations(9, 7), o.get_permutations(9, 7))      # 63 bits exactly
    assert np.array_equal(t.get_field(), o.get_field())


def test_transcript_device_side_io(zk, orc):
    rng = np.random.default_rng(12)
    root = rng.integers(0, P, size=4, dtype=np.uint64)
    t, o = zk.TranscriptGL(), orc.transcript()
    d_root = zk.DevArray.from_host(root)
    t.put_dev(d_root); o.put(root)
    d_ch = zk.DevArray(3)
    t.get_field_dev(d_ch)
    assert np.array_equal(d_ch.to_host(), o.get_field())


# ---- FRI (fri.rs) ---------------------------------------------------------------------------
@pytest.mark.parametrize("pol_bits,step_bits", [(5, 5), (3, 2), (6, 2), (11, 7), (7, 3), (12, 6), (13, 12), (16, 11), (18, 13), (17, 7), (10, 3), (12, 1), (11, 0), (19, 8)])
def test_fri_fold_matches_oracle(zk, orc, pol_bits, step_bits):
    rng = np.random.default_rng(pol_bits * 100 + step_bits)
    pol = _f3(rng, 1 << pol_bits)
    sx = _f3(rng, 1)
    shift_inv = pow(pow(49, P - 2, P), 1 << 3, P)          # as after 3 halvings (fri.rs:147-150)
    got = zk.fri_fold(zk.DevArray.from_host(pol), pol_bits, step_bits, zk.DevArray.from_host(sx), shift_inv).to_host()
    assert np.array_equal(got, orc.fri_fold(pol, pol_bits, step_bits, sx, shift_inv))


@pytest.mark.parametrize("n_bits,tbits", [(4, 2), (7, 3), (11, 7), (15, 11), (3, 0), (5, 5)])
def test_fri_transpose(zk, orc, n_bits, tbits):
    rng = np.random.default_rng(n_bits * 7 + tbits)
    pol = _f3(rng, 1 << n_bits)
    got = zk.fri_transpose(zk.DevArray.from_host(pol), 1 << n_bits, tbits).to_host()
    assert np.array_equal(got, orc.fri_transpose(pol, 1 << n_bits, tbits))


# ---- stark_gen glue -------------------------------------------------------------------------
def test_domain_tables(zk, orc):
    for nbits in (0, 1, 5, 10, 16):
        w = orc.root(nbits)
        got = zk.x_table(nbits, 49).to_host()
        exp, c = [], 49
        for _ in range(min(1 << nbits, 64)):
            exp.append(c); c = c * w % P
        assert [int(v) for v in got[:len(exp)]] == exp
        assert int(got[-1]) == 49 * pow(w, (1 << nbits) - 1, P) % P
    for nbits, ext in ((10, 1), (15, 2), (20, 3), (5, 0)):
        assert np.array_equal(zk.zh_inv(nbits, ext).to_host(), orc.zh_inv(nbits, ext))


@pytest.mark.parametrize("nbits_ext", [1, 4, 11, 15])
def test_xdivxsub(zk, orc, nbits_ext):
    rng = np.random.default_rng(70 + nbits_ext)
    xi = _f3(rng, 1)
    d_xi = zk.DevArray.from_host(xi)
    assert np.array_equal(zk.xdivxsub(d_xi, 1, nbits_ext).to_host(), orc.xdivxsub(xi, nbits_ext))
    w = orc.root(nbits_ext - 1)                                    # w*xi with w = MG[nBits]
    wxi = np.array([int(v) * w % P for v in xi], np.uint64)
    assert np.array_equal(zk.xdivxsub(d_xi, w, nbits_ext).to_host(), orc.xdivxsub(wxi, nbits_ext))


@pytest.mark.parametrize("nbits", [0, 2, 5, 10, 14])
def test_lev_tables(zk, orc, nbits):
    rng = np.random.default_rng(90 + nbits)
    xi = _f3(rng, 1)
    d_xi = zk.DevArray.from_host(xi)
    for prime in (False, True):
        assert np.array_equal(zk.lev(d_xi, nbits, prime).to_host(), orc.lev(xi, nbits, prime))


def test_evals(zk, orc):
    rng = np.random.default_rng(123)
    nbits, ext = 10, 1
    nx = 1 << (nbits + ext)
    cm1 = rng.integers(0, P, size=nx * 5, dtype=np.uint64)     # width 5: cols 0,1 dim1, col 2..4 one dim-3 pol
    const = rng.integers(0, P, size=nx * 2, dtype=np.uint64)
    xi = _f3(rng, 1)
    L, Lp = orc.lev(xi, nbits, False), orc.lev(xi, nbits, True)
    d_cm1, d_const = zk.DevArray.from_host(cm1), zk.DevArray.from_host(const)
    descs = [(d_cm1, 5, 0, 1, False), (d_cm1, 5, 1, 1, True), (d_cm1, 5, 2, 3, False), (d_const, 2, 1, 1, True),
             (d_const, 2, 0, 1, False), (d_cm1, 5, 2, 3, True)]
    got = zk.evals(descs, nbits, ext, zk.DevArray.from_host(L), zk.DevArray.from_host(Lp)).to_host().reshape(-1, 3)
    hosts = {id(d_cm1): cm1, id(d_const): const}
    for e, (b, w, off, dim, prime) in enumerate(descs):
        exp = orc.eval_dot(hosts[id(b)], w, off, dim, nbits, ext, Lp if prime else L)
        assert np.array_equal(got[e], exp), e


@pytest.mark.parametrize("nbits,ext,q_dim,q_deg", [(4, 1, 1, 1), (6, 1, 3, 2), (10, 1, 3, 2), (8, 2, 3, 3), (8, 2, 1, 4)])
def test_qsplit(zk, orc, nbits, ext, q_dim, q_deg):
    rng = np.random.default_rng(nbits * 11 + q_dim)
    qq1 = rng.integers(0, P, size=(1 << (nbits + ext)) * q_dim, dtype=np.uint64)
    got = zk.qsplit(zk.DevArray.from_host(qq1), nbits, nbits + ext, q_dim, q_deg).to_host()
    assert np.array_equal(got, orc.qsplit(qq1, nbits, nbits + ext, q_dim, q_deg))


@pytest.mark.parametrize("n,n_distinct", [(1, 1), (2, 1), (8, 3), (1000, 17), (1 << 12, 1 << 12), (1 << 16, 300), (100003, 5000)])
def test_calculate_h1h2_matches_reference(zk, n, n_distinct):
    """zk_stark_calculate_h1h2_dev against the restated calculate_H1H2 (stark_gen.rs:624-651; oracle/stark_prover.py): tables with
    repeated rows (the LAST index of a value is its place), values looked up many times or never, dim-1 and dim-3 operands"""
    import importlib
    sys.path.insert(0, str(ROOT / "oracle"))
    import stark_prover as SP
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    zk.init(0)
    rng = np.random.default_rng(n * 7 + n_distinct)
    vals = rng.integers(0, P, size=(n_distinct, 3), dtype=np.uint64)
    if n % 2 == 0:
        vals[:, 1:] = 0                                                  # base-field operands as get_pol pads them
    t = vals[rng.integers(0, n_distinct, size=n)]                        # table rows: repeats unless n_distinct >= n
    f = t[rng.integers(0, n, size=n)]                                    # every looked-up value is in the table
    if n > 4:
        f[: n // 3] = t[0]                                               # one value carries a third of the lookups
    h1, h2 = stark.calculate_h1h2_dev(zk.DevArray.from_host(f.reshape(-1)), zk.DevArray.from_host(t.reshape(-1)))
    e1, e2 = SP.calculate_h1h2([tuple(int(v) for v in r) for r in f], [tuple(int(v) for v in r) for r in t])
    assert np.array_equal(h1.to_host().reshape(-1, 3), np.array(e1, dtype=np.uint64).reshape(-1, 3))
    assert np.array_equal(h2.to_host().reshape(-1, 3), np.array(e2, dtype=np.uint64).reshape(-1, 3))
    if n >= 8:                                                           # the first missing value is the one reported
        bad = f.copy(); bad[5] = [P - 1, P - 2, 7]; bad[n - 1] = [P - 3, 1, 1]
        with pytest.raises(zk.ZkError, match="Number not included: %d" % (P - 1)):
            stark.calculate_h1h2_dev(zk.DevArray.from_host(bad.reshape(-1)), zk.DevArray.from_host(t.reshape(-1)))
