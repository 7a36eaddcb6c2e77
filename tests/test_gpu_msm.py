"""Parity of the BN254 G1 multi-scalar multiplication (csrc/msm.hip, through zk_msm_g1_bn254) against
the CPU oracle's Pippenger (oracle/ec.c) and the closed form  sum s_i [k_i]G == [sum s_i k_i mod r]G.
Points compare as affine Montgomery limbs, bit exact (the affine form of a group element is unique)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


@pytest.fixture(scope="module", autouse=True)
def _gpu(zk):
    assert zk.lib().zk_device_count() >= 1, "no GPU visible (the product has no CPU fallback)"
    zk.init(0)


def words(x, n=4):
    return np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(n)], np.uint64)


def rand_scalars(rng, n):
    raw = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 60) - 1)          # < 2^252 < r : canonical by construction
    return raw.reshape(-1)


def scalar_ints(s):
    s = s.reshape(-1, 4)
    return [sum(int(v) << (64 * j) for j, v in enumerate(row)) for row in s]


@pytest.mark.parametrize("n", [1, 2, 3, 33, 64, 65, 1000, 4097])
def test_msm_matches_oracle_small(zk, orc, n):
    rng = np.random.default_rng(100 + n)
    bases = orc.bn254_make_bases(n, 3, 5)
    scal = rand_scalars(rng, n)
    got, inf = zk.msm_g1_bn254(bases, scal)
    exp, einf = orc.bn254_msm(bases, scal, 8)
    assert inf == einf and np.array_equal(got, exp)


def test_msm_edge_scalars(zk, orc):
    """0, 1, r-1, 2^k boundaries of the 16-bit windows, all-ones windows."""
    vals = [0, 1, R - 1, R - 2, 2**16 - 1, 2**16, 2**253 % R, (1 << 240) - 1, 0xFFFF << 48, 1 << 255 - 2]
    vals = [v % R for v in vals]
    n = len(vals)
    bases = orc.bn254_make_bases(n, 7, 11)
    scal = np.concatenate([words(v) for v in vals])
    got, inf = zk.msm_g1_bn254(bases, scal)
    k = sum(v * (7 + 11 * i) for i, v in enumerate(vals)) % R
    exp, einf = orc.bn254_scalar_mul(orc.bn254_generator(), words(k))
    assert inf == einf and np.array_equal(got, exp)


def test_msm_endomorphism_split_edge_scalars(zk, orc):
    """Sums of >= 4096 points go through k P = k1 P + k2 phi(P) (csrc/msm_impl.hip.h, glv_split_kernel): scalars at the edges of the
    split -- 0, 1, r - 1, lambda and its neighbours, 2^127, 2^128 and neighbours, the lattice vectors, values whose halves
    change sign -- among random ones, against the closed form and against the oracle's Pippenger."""
    LAM = 4407920970296243842393367215006156084916469457145843978461
    A2, NB1, B2 = 147946756881789319010696353538189108491, 147946756881789319000765030803803410728, 9931322734385697763
    edge = [0, 1, 2, R - 1, R - 2, LAM, LAM - 1, LAM + 1, R - LAM, (LAM * LAM) % R, 1 << 127, (1 << 127) - 1, 1 << 128, (1 << 128) - 1, (1 << 128) + 1,
            A2, NB1, B2, A2 + 1, NB1 - 1, R // 2, R // 2 + 1, R // 3, (1 << 253), (1 << 253) - 1, 0xFFFF << 112, 0xFFFF << 128, (B2 * LAM) % R, (A2 * LAM) % R]
    n = 4096
    rng = np.random.default_rng(77)
    vals = edge + [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n - len(edge))]
    bases = orc.bn254_make_bases(n, 7, 11)
    scal = np.concatenate([words(v) for v in vals])
    got, inf = zk.msm_g1_bn254(bases, scal)
    k = sum(v * (7 + 11 * i) for i, v in enumerate(vals)) % R
    exp, einf = orc.bn254_scalar_mul(orc.bn254_generator(), words(k))
    assert inf == einf and np.array_equal(got, exp)
    exp2, einf2 = orc.bn254_msm(bases, scal, 8)
    assert inf == einf2 and np.array_equal(got, exp2)
    # s P + (r - s) P = 0 whatever the signs of the four halves: the second half of the sum repeats the bases of the first
    half = n // 2
    b = bases.reshape(n, -1).copy(); b[half:] = b[:half]
    sc = np.concatenate([words(v) for v in vals[:half]] + [words((R - v) % R) for v in vals[:half]])
    _, inf0 = zk.msm_g1_bn254(b.reshape(-1), sc)
    assert inf0


def test_msm_all_zero_scalars_is_infinity(zk, orc):
    bases = orc.bn254_make_bases(100, 1, 1)
    _, inf = zk.msm_g1_bn254(bases, np.zeros(400, np.uint64))
    assert inf


def test_msm_empty_is_infinity(zk):
    _, inf = zk.msm_g1_bn254(np.zeros(0, np.uint64), np.zeros(0, np.uint64))
    assert inf


def test_msm_cancellation_and_duplicates(zk, orc):
    one = orc.bn254_make_bases(1, 5, 1)
    # P*9 + P*(r-9) = infinity : exercises P + (-P) in the window combination
    _, inf = zk.msm_g1_bn254(np.concatenate([one, one]), np.concatenate([words(9), words(R - 9)]))
    assert inf
    # the same base 300 times with the same scalar: every add in a bucket is a doubling or P+2P...
    n = 300
    bases = np.tile(one, n)
    scal = np.tile(words(0x0123456789abcdef0123456789abcdef), n)
    got, inf = zk.msm_g1_bn254(bases, scal)
    exp, einf = orc.bn254_scalar_mul(orc.bn254_generator(), words(5 * n * 0x0123456789abcdef0123456789abcdef % R))
    assert inf == einf and np.array_equal(got, exp)


def test_msm_every_base_the_same_point_with_window_boundary_scalars(zk, orc):
    """The sum the round-6 fuzz campaign stumbled over (seed 671: 169 bases made with step 0).  The failure was the GENERATOR's -- the oracle's
    make_bases took [0]G, the point at infinity, for a finite step and produced garbage bases; fixed in oracle/ec_impl.h -- but the shape is
    worth keeping: every base is [748]G, the scalars are random 252-bit words with 0, 1, r - 1, 2^(16 k) - 1 and 2^(16 k) every seventh place,
    so every bucket addition is a doubling or meets its own multiple.  Oracle, closed form and device must agree."""
    n, a = 169, 748
    bases = orc.bn254_make_bases(n, a, 0)
    assert all(np.array_equal(bases[8 * i:8 * i + 8], bases[:8]) for i in range(n))      # b = 0: [a]G every time
    rng = np.random.default_rng(671)
    raw = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); raw[:, 3] &= np.uint64((1 << 60) - 1)
    for k in range(0, n, 7):
        v = [0, 1, R - 1, (1 << (16 * int(rng.integers(1, 15)))) - 1, 1 << (16 * int(rng.integers(1, 15)))][int(rng.integers(0, 5))]
        raw[k] = [(v >> (64 * j)) & (2**64 - 1) for j in range(4)]
    vals = [sum(int(raw[i, j]) << (64 * j) for j in range(4)) for i in range(n)]
    got, inf = zk.msm_g1_bn254(bases, raw.reshape(-1))
    exp, einf = orc.bn254_msm(bases, raw.reshape(-1), 8)
    closed, cinf = orc.bn254_scalar_mul(orc.bn254_generator(), words(a * sum(vals) % R))
    assert not inf and inf == einf == cinf and np.array_equal(got, exp) and np.array_equal(got, closed)


def test_msm_same_bucket_pressure(zk, orc):
    """All scalars share their window digits except the lowest: one bucket per window gets n points."""
    n = 5000
    rng = np.random.default_rng(5)
    base_s = int.from_bytes(rng.bytes(31), "little") & ~0xFFFF
    vals = [base_s + int(v) for v in rng.integers(0, 4, size=n)]
    bases = orc.bn254_make_bases(n, 2, 3)
    scal = np.concatenate([words(v) for v in vals])
    got, inf = zk.msm_g1_bn254(bases, scal)
    k = sum(v * (2 + 3 * i) for i, v in enumerate(vals)) % R
    exp, einf = orc.bn254_scalar_mul(orc.bn254_generator(), words(k))
    assert inf == einf and np.array_equal(got, exp)


@pytest.mark.parametrize("logn", [16, 20])
def test_msm_large_closed_form(zk, orc, logn):
    n = 1 << logn
    rng = np.random.default_rng(logn)
    bases = orc.bn254_make_bases(n, 3, 5)
    scal = rand_scalars(rng, n)
    got, inf = zk.msm_g1_bn254(bases, scal)
    ks = 3 + 5 * np.arange(n, dtype=object)
    k = int(sum(s * int(ki) for s, ki in zip(scalar_ints(scal), ks)) % R)
    exp, einf = orc.bn254_scalar_mul(orc.bn254_generator(), words(k))
    assert inf == einf and np.array_equal(got, exp)
    if logn == 16:                                   # and the oracle's own Pippenger, another window size
        exp2, _ = orc.bn254_msm(bases, scal, 13)
        assert np.array_equal(got, exp2)


def test_msm_linearity(zk, orc):
    """MSM(b, s1) + MSM(b, s2) == MSM(b, s1 + s2 mod r), checked through one more 2-point MSM."""
    n = 2048
    rng = np.random.default_rng(77)
    bases = orc.bn254_make_bases(n, 9, 2)
    s1, s2 = rand_scalars(rng, n), rand_scalars(rng, n)
    s12 = np.concatenate([words((a + b) % R) for a, b in zip(scalar_ints(s1), scalar_ints(s2))])
    p1, i1 = zk.msm_g1_bn254(bases, s1)
    p2, i2 = zk.msm_g1_bn254(bases, s2)
    p12, i12 = zk.msm_g1_bn254(bases, s12)
    assert not (i1 or i2 or i12)
    both, inf = zk.msm_g1_bn254(np.concatenate([p1, p2]), np.concatenate([words(1), words(1)]))
    assert not inf and np.array_equal(both, p12)


def test_generator_multiples_match_oracle(zk, orc):
    rng = np.random.default_rng(3)
    k = rng.integers(1, 2**64, size=200, dtype=np.uint64)
    k[:4] = [1, 2, 2**64 - 1, 2**63]
    bases = zk.g1_bn254_mul_generator(zk.DevArray.from_host(k)).to_host().reshape(-1, 8)
    g = orc.bn254_generator()
    for i in (0, 1, 2, 3, 57, 199):
        exp, _ = orc.bn254_scalar_mul(g, words(int(k[i])))
        assert np.array_equal(bases[i], exp)
        assert orc.bn254_on_curve(bases[i])


@pytest.mark.parametrize("n", [1 << 22, (1 << 23) + 3])
def test_msm_full_size_closed_form(zk, orc, n):
    """BASELINE config 4: n = 2^22, bases [k_i]G generated on the device, uniform scalars below r; and 2^23 + 3 points -- 2^24 + 6 (point,
    window-half) pairs behind the endomorphism: the sizes whose pairs are sorted chunk by chunk (2^22 pairs each, round 4) where ONE sort
    with device-scope atomics used to run."""
    rng = np.random.default_rng(22)
    k = rng.integers(1, 2**64, size=n, dtype=np.uint64)
    scal = rand_scalars(rng, n)
    d_bases = zk.g1_bn254_mul_generator(zk.DevArray.from_host(k))
    out = zk.msm_g1_bn254_dev(d_bases, zk.DevArray.from_host(scal), n).to_host()
    s4 = scal.reshape(-1, 4).astype(object)
    sv = s4[:, 0] + (s4[:, 1] << 64) + (s4[:, 2] << 128) + (s4[:, 3] << 192)
    kk = int((sv * k.astype(object)).sum() % R)
    exp, einf = orc.bn254_scalar_mul(orc.bn254_generator(), words(kk))
    assert (int(out[8]) & 0xFFFFFFFF) == int(einf) and np.array_equal(out[:8], exp)


# ---- BLS12-381 G1 (zk_msm_g1_bls12_381) ---------------------------------------------------------------
R_BLS = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def rand_scalars_255(rng, n):
    raw = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 62) - 1)          # < 2^254 < r : canonical by construction
    return raw.reshape(-1)


@pytest.mark.parametrize("n", [1, 2, 33, 65, 1000, 4097])
def test_bls_msm_matches_oracle_small(zk, orc, n):
    cv = orc.curve("bls12_381")
    rng = np.random.default_rng(381 + n)
    bases = cv.make_bases(n, 3, 5)
    scal = rand_scalars_255(rng, n)
    got, inf = zk.msm_g1(bases, scal, "bls12_381")
    exp, einf = cv.msm(bases, scal, 8)
    assert inf == einf and np.array_equal(got, exp)


def test_bls_msm_edge_scalars_and_cancellation(zk, orc):
    cv = orc.curve("bls12_381")
    vals = [0, 1, R_BLS - 1, R_BLS - 2, 2**16 - 1, 2**16, (1 << 254) % R_BLS, (1 << 240) - 1, 0xFFFF << 48, 0xFFFF << 239]
    vals = [v % R_BLS for v in vals]
    bases = cv.make_bases(len(vals), 7, 11)
    got, inf = zk.msm_g1(bases, np.concatenate([words(v) for v in vals]), "bls12_381")
    k = sum(v * (7 + 11 * i) for i, v in enumerate(vals)) % R_BLS
    exp, einf = cv.scalar_mul(cv.generator(), words(k))
    assert inf == einf and np.array_equal(got, exp)
    one = cv.make_bases(1, 5, 1)
    assert zk.msm_g1(np.concatenate([one, one]), np.concatenate([words(9), words(R_BLS - 9)]), "bls12_381")[1]
    assert zk.msm_g1(cv.make_bases(50, 1, 1), np.zeros(200, np.uint64), "bls12_381")[1]
    assert zk.msm_g1(np.zeros(0, np.uint64), np.zeros(0, np.uint64), "bls12_381")[1]
    n = 300                                                     # the same base 300 times: doublings inside buckets
    got, inf = zk.msm_g1(np.tile(one, n), np.tile(words(0x0123456789abcdef0123456789abcdef), n), "bls12_381")
    exp, einf = cv.scalar_mul(cv.generator(), words(5 * n * 0x0123456789abcdef0123456789abcdef % R_BLS))
    assert inf == einf and np.array_equal(got, exp)


def test_bls_msm_endomorphism_split_edge_scalars(zk, orc):
    """BLS12-381 G1 sums of >= 4096 points: k = k1 + k2 lambda by division (lambda = z^2 - 1).  Scalars around multiples of lambda,
    around r, and above r (brought below it before the split) among random ones; closed form, the oracle, and s P + (r - s) P = 0."""
    cv = orc.curve("bls12_381")
    LAM = 0xac45a4010001a40200000000ffffffff
    edge = [0, 1, LAM - 1, LAM, LAM + 1, 2 * LAM - 1, 2 * LAM, LAM * LAM, LAM * LAM + LAM, R_BLS - 1, R_BLS - 2, R_BLS - LAM, R_BLS // 2,
            (1 << 128) - 1, 1 << 128, (1 << 128) + 1, 1 << 127, (LAM + 1) * LAM - 1, 0xFFFF << 112, 0xFFFF << 128, (1 << 254) - 1,
            R_BLS, R_BLS + 1, R_BLS + LAM, (1 << 256) - 1, 2 * R_BLS, 2 * R_BLS + 5]                       # the last six are not canonical
    n = 4096
    rng = np.random.default_rng(381)
    vals = edge + [int.from_bytes(rng.bytes(32), "little") % R_BLS for _ in range(n - len(edge))]
    bases = cv.make_bases(n, 7, 11)
    scal = np.concatenate([words(v) for v in vals])
    got, inf = zk.msm_g1(bases, scal, "bls12_381")
    k = sum(v * (7 + 11 * i) for i, v in enumerate(vals)) % R_BLS
    exp, einf = cv.scalar_mul(cv.generator(), words(k))
    assert inf == einf and np.array_equal(got, exp)
    canon = np.concatenate([words(v % R_BLS) for v in vals])
    exp2, einf2 = cv.msm(bases, canon, 8)
    assert inf == einf2 and np.array_equal(got, exp2)
    half = n // 2
    b = bases.reshape(n, -1).copy(); b[half:] = b[:half]
    sc = np.concatenate([words(v % R_BLS) for v in vals[:half]] + [words((R_BLS - v) % R_BLS) for v in vals[:half]])
    assert zk.msm_g1(b.reshape(-1), sc, "bls12_381")[1]


def test_bls_generator_multiples_match_oracle(zk, orc):
    cv = orc.curve("bls12_381")
    rng = np.random.default_rng(4)
    k = rng.integers(1, 2**64, size=100, dtype=np.uint64)
    k[:4] = [1, 2, 2**64 - 1, 2**63]
    bases = zk.g1_mul_generator(zk.DevArray.from_host(k), "bls12_381").to_host().reshape(-1, 12)
    for i in (0, 1, 2, 3, 57, 99):
        exp, _ = cv.scalar_mul(cv.generator(), words(int(k[i])))
        assert np.array_equal(bases[i], exp) and cv.on_curve(bases[i])


@pytest.mark.parametrize("logn", [16, 20, 22])
def test_bls_msm_large_closed_form(zk, orc, logn):
    cv = orc.curve("bls12_381")
    n = 1 << logn
    rng = np.random.default_rng(1000 + logn)
    k = rng.integers(1, 2**64, size=n, dtype=np.uint64)
    scal = rand_scalars_255(rng, n)
    d_bases = zk.g1_mul_generator(zk.DevArray.from_host(k), "bls12_381")
    out = zk.msm_g1_dev(d_bases, zk.DevArray.from_host(scal), n, "bls12_381").to_host()
    s4 = scal.reshape(-1, 4).astype(object)
    sv = s4[:, 0] + (s4[:, 1] << 64) + (s4[:, 2] << 128) + (s4[:, 3] << 192)
    kk = int((sv * k.astype(object)).sum() % R_BLS)
    exp, einf = cv.scalar_mul(cv.generator(), words(kk))
    assert (int(out[12]) & 0xFFFFFFFF) == int(einf) and np.array_equal(out[:12], exp)
    if logn == 16:                                              # and the oracle's own Pippenger on the same bases
        exp2, _ = cv.msm(d_bases.to_host(), scal, 13)
        assert np.array_equal(out[:12], exp2)


# ---- G2: zk_msm_g2_bn254 / zk_msm_g2_bls12_381 (Fq2 coordinates) ------------------------------------------
@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_g2_msm_matches_oracle_and_closed_form(zk, orc, curve):
    cv = orc.curve(curve, g2=True)
    rng = np.random.default_rng(len(curve))
    for n in (1, 2, 33, 300):
        bases = cv.make_bases(n, 3, 5)
        scal = rand_scalars(rng, n)
        if n >= 33:
            scal.reshape(-1, 4)[0] = 0; scal.reshape(-1, 4)[1] = words(cv.r - 1); scal.reshape(-1, 4)[2] = words(1)
        got, inf = zk.msm_g1(bases, scal, curve, group="g2")
        exp, einf = cv.msm(bases, scal, 8)
        assert inf == einf and np.array_equal(got, exp), n
    one = cv.make_bases(1, 5, 1)
    assert zk.msm_g1(np.concatenate([one, one]), np.concatenate([words(9), words(cv.r - 9)]), curve, group="g2")[1]   # P - P
    assert zk.msm_g1(np.zeros(0, np.uint64), np.zeros(0, np.uint64), curve, group="g2")[1]
    n = 200                                                                     # the same base 200 times: doublings in buckets
    got, inf = zk.msm_g1(np.tile(one, n), np.tile(words(0x0123456789abcdef0123456789abcdef), n), curve, group="g2")
    exp, einf = cv.scalar_mul(cv.generator(), words(5 * n * 0x0123456789abcdef0123456789abcdef % cv.r))
    assert inf == einf and np.array_equal(got, exp)


@pytest.mark.parametrize("curve,logn", [("bn254", 16), ("bn254", 20), ("bls12_381", 16)])
def test_g2_msm_large_closed_form(zk, orc, curve, logn):
    cv = orc.curve(curve, g2=True)
    n = 1 << logn
    rng = np.random.default_rng(2000 + logn)
    k = rng.integers(1, 2**64, size=n, dtype=np.uint64)
    scal = rand_scalars(rng, n)
    d_bases = zk.g1_mul_generator(zk.DevArray.from_host(k), curve, group="g2")
    hb = d_bases.to_host().reshape(n, -1)
    for i in (0, 1, n - 1):                                                     # the device-made bases are the claimed multiples
        exp, _ = cv.scalar_mul(cv.generator(), words(int(k[i])))
        assert np.array_equal(hb[i], exp) and cv.on_curve(hb[i])
    out = zk.msm_g1_dev(d_bases, zk.DevArray.from_host(scal), n, curve, group="g2").to_host()
    s4 = scal.reshape(-1, 4).astype(object)
    sv = s4[:, 0] + (s4[:, 1] << 64) + (s4[:, 2] << 128) + (s4[:, 3] << 192)
    kk = int((sv * k.astype(object)).sum() % cv.r)
    exp, einf = cv.scalar_mul(cv.generator(), words(kk))
    assert (int(out[cv.pw]) & 0xFFFFFFFF) == int(einf) and np.array_equal(out[:cv.pw], exp)


@pytest.mark.parametrize("curve,group", [("bn254", "g1"), ("bn254", "g2"), ("bls12_381", "g1"), ("bls12_381", "g2")])
def test_window_table_sums_equal_plain_sums(zk, orc, curve, group):
    """zk_msm_*_table_*: the table of 2^(16 w) P_i gives bit-identical sums, for the whole array and for sub-ranges"""
    n = 3000
    rng = np.random.default_rng(77)
    k = rng.integers(1, 2**64, size=n, dtype=np.uint64)
    d_bases = zk.g1_mul_generator(zk.DevArray.from_host(k), curve, group=group)
    scal = rand_scalars(rng, n).reshape(n, 4)
    scal[5] = 0; scal[6] = [1, 0, 0, 0]; scal[7] = [0, 0, 0, 1 << 59]            # zero, one, a top-window-only scalar
    tab = zk.MsmTable(d_bases, n, curve, group)
    pw = zk._CURVES[curve] * (4 if group == "g2" else 2)
    bases = d_bases.to_host().reshape(n, pw)
    for off, cnt in ((0, n), (0, 1), (7, 1), (100, 1500), (n - 3, 3)):
        d_s = zk.DevArray.from_host(scal[off:off + cnt].reshape(-1))
        got = tab.msm(d_s, cnt, off).to_host()
        exp = zk.msm_g1_dev(zk.DevArray.from_host(bases[off:off + cnt].reshape(-1)), d_s, cnt, curve, group=group).to_host()
        assert np.array_equal(got, exp), (off, cnt)


def test_window_table_cancellation_and_duplicates(zk, orc):
    """the same corner cases through the table path: P*9 + P*(r-9) = infinity across the merged windows and bit partials,
    and one base repeated 300 times (every addition inside a bucket is a doubling or P + 2P ...)"""
    one = orc.bn254_make_bases(1, 5, 1)
    tab = zk.MsmTable(zk.DevArray.from_host(np.concatenate([one, one])), 2, "bn254", "g1")
    out = tab.msm(zk.DevArray.from_host(np.concatenate([words(9), words(R - 9)]))).to_host()
    assert int(out[8]) & 0xFFFFFFFF == 1                                   # infinity flag
    n = 300
    tab = zk.MsmTable(zk.DevArray.from_host(np.tile(one, n)), n, "bn254", "g1")
    k = 0x0123456789abcdef0123456789abcdef
    out = tab.msm(zk.DevArray.from_host(np.tile(words(k), n))).to_host()
    exp, einf = orc.bn254_scalar_mul(orc.bn254_generator(), words(5 * n * k % R))
    assert not einf and int(out[8]) & 0xFFFFFFFF == 0 and np.array_equal(out[:8], exp)
