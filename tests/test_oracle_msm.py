"""Pins the oracle's BN254 G1 arithmetic and Pippenger MSM (oracle/ec.c) by algebra: the reference
holds no MSM vector (the arithmetic lives in bellman_ce, outside /root/reference), so
parity is anchored on alt_bn128's standard constants and on
    sum_i s_i * [k_i]G  ==  [sum_i s_i * k_i mod r] G ."""
import numpy as np

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def words(x, n=4):
    return np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(n)], np.uint64)


def to_int(w):
    return sum(int(v) << (64 * i) for i, v in enumerate(w))


def affine_ints(orc, p):
    return to_int(orc.fq_from_mont(p[:4])), to_int(orc.fq_from_mont(p[4:]))


def rand_scalars(rng, n):
    return [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]


def test_generator_and_group_law(orc):
    g = orc.bn254_generator()
    assert affine_ints(orc, g) == (1, 2) and orc.bn254_on_curve(g)
    # [2]G of alt_bn128 (EIP-196 test vector)
    p2, inf = orc.bn254_scalar_mul(g, words(2))
    assert not inf and affine_ints(orc, p2) == (
        1368015179489954701390400359078579693043519447331113978918064868415326638035,
        9918110051302171585080402603319702774565515993150576347155970296011118125764)
    # [r]G = infinity, [r-1]G = -G
    assert orc.bn254_scalar_mul(g, words(R))[1]
    pm, _ = orc.bn254_scalar_mul(g, words(R - 1))
    assert affine_ints(orc, pm) == (1, Q - 2)
    # distributivity on a random pair
    a, b = 0x1234567890abcdef1122334455667788, 0xfedcba9876543210
    pa, _ = orc.bn254_scalar_mul(g, words(a)); pab, _ = orc.bn254_scalar_mul(pa, words(b))
    direct, _ = orc.bn254_scalar_mul(g, words(a * b % R))
    assert np.array_equal(pab, direct)


def test_make_bases_are_the_claimed_multiples(orc):
    g = orc.bn254_generator()
    bases = orc.bn254_make_bases(50, 7, 11).reshape(-1, 8)
    for i in (0, 1, 2, 17, 49):
        assert orc.bn254_on_curve(bases[i])
        exp, _ = orc.bn254_scalar_mul(g, words(7 + 11 * i))
        assert np.array_equal(bases[i], exp)


def test_msm_matches_closed_form(orc):
    rng = np.random.default_rng(42)
    g = orc.bn254_generator()
    for n, c in ((1, 4), (2, 8), (33, 5), (500, 8), (3000, 11)):
        a, b = 3, 5
        bases = orc.bn254_make_bases(n, a, b)
        s = rand_scalars(rng, n)
        if n >= 33:
            s[0] = 0; s[1] = R - 1; s[2] = 1                      # edge scalars
        scal = np.concatenate([words(v) for v in s])
        got, inf = orc.bn254_msm(bases, scal, c)
        k = sum(si * (a + b * i) for i, si in enumerate(s)) % R
        exp, einf = orc.bn254_scalar_mul(g, words(k))
        assert inf == einf and (inf or np.array_equal(got, exp)), (n, c)


def test_msm_cancellation_gives_infinity(orc):
    one = orc.bn254_make_bases(1, 5, 1)
    bases = np.concatenate([one, one])                            # the same point twice
    scal = np.concatenate([words(9), words(R - 9)])
    assert orc.bn254_msm(bases, scal, 6)[1]


# ---- BLS12-381 G1 (oracle/ec_bls12_381.c) -----------------------------------------------------------------
BLS_Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS_GX = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
BLS_GY = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1


def _py_affine_mul(k, P, q):
    """textbook affine double-and-add over Python integers: an implementation independent of the C oracle"""
    def add(A, B):
        if A is None: return B
        if B is None: return A
        (x1, y1), (x2, y2) = A, B
        if x1 == x2:
            if (y1 + y2) % q == 0: return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, q) % q
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, q) % q
        x3 = (lam * lam - x1 - x2) % q
        return x3, (lam * (x1 - x3) - y1) % q
    acc = None
    while k:
        if k & 1: acc = add(acc, P)
        P = add(P, P); k >>= 1
    return acc


def test_bls12_381_generator_and_group_law(orc):
    cv = orc.curve("bls12_381")
    g = cv.generator()
    assert cv.affine_ints(g) == (BLS_GX, BLS_GY) and cv.on_curve(g)
    assert (BLS_GY * BLS_GY - BLS_GX ** 3 - 4) % BLS_Q == 0                # y^2 = x^3 + 4
    assert cv.scalar_mul(g, words(cv.r))[1]                                # [r]G = infinity
    for k in (2, 3, 0xdeadbeefcafebabe1234567, cv.r - 1):
        got, inf = cv.scalar_mul(g, words(k))
        assert not inf and cv.affine_ints(got) == _py_affine_mul(k, (BLS_GX, BLS_GY), BLS_Q)


def test_bls12_381_msm_matches_closed_form(orc):
    cv = orc.curve("bls12_381")
    rng = np.random.default_rng(381)
    g = cv.generator()
    for n, c in ((1, 4), (2, 8), (33, 5), (500, 8), (2000, 11)):
        a, b = 3, 5
        bases = cv.make_bases(n, a, b)
        s = [int.from_bytes(rng.bytes(32), "little") % cv.r for _ in range(n)]
        if n >= 33:
            s[0] = 0; s[1] = cv.r - 1; s[2] = 1
        scal = np.concatenate([words(v) for v in s])
        got, inf = cv.msm(bases, scal, c)
        k = sum(si * (a + b * i) for i, si in enumerate(s)) % cv.r
        exp, einf = cv.scalar_mul(g, words(k))
        assert inf == einf and (inf or np.array_equal(got, exp)), (n, c)
    one = cv.make_bases(1, 5, 1)
    assert cv.msm(np.concatenate([one, one]), np.concatenate([words(9), words(cv.r - 9)]), 6)[1]


# ---- G2 (oracle/ec_impl.h, second half): the twist over Fq2 = Fq[u]/(u^2 + 1) --------------------------------
def _f2mul(a, b, q):
    return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)


def _py_affine_mul_f2(k, P, q):
    """textbook affine double-and-add over Fq2 with Python integers (independent of the C oracle)"""
    inv = lambda a: (lambda n: (a[0] * n % q, -a[1] * n % q))(pow((a[0] * a[0] + a[1] * a[1]) % q, -1, q))
    sub = lambda a, b: ((a[0] - b[0]) % q, (a[1] - b[1]) % q)
    def add(A, B):
        if A is None: return B
        if B is None: return A
        (x1, y1), (x2, y2) = A, B
        if x1 == x2:
            if ((y1[0] + y2[0]) % q, (y1[1] + y2[1]) % q) == (0, 0): return None
            xx = _f2mul(x1, x1, q)
            lam = _f2mul(((3 * xx[0]) % q, (3 * xx[1]) % q), inv(((2 * y1[0]) % q, (2 * y1[1]) % q)), q)
        else:
            lam = _f2mul(sub(y2, y1), inv(sub(x2, x1)), q)
        x3 = sub(sub(_f2mul(lam, lam, q), x1), x2)
        return x3, sub(_f2mul(lam, sub(x1, x3), q), y1)
    acc = None
    while k:
        if k & 1: acc = add(acc, P)
        P = add(P, P); k >>= 1
    return acc


import pytest


@pytest.mark.parametrize("name,q", [("bn254", Q), ("bls12_381", BLS_Q)])
def test_g2_generator_group_law_and_msm(orc, name, q):
    cv = orc.curve(name, g2=True)
    g = cv.generator()
    gi = cv.affine_ints(g)
    assert cv.on_curve(g) and cv.scalar_mul(g, words(cv.r))[1]                 # on the twist, order r
    for k in (2, 3, 0xdeadbeefcafebabe1234567):
        got, inf = cv.scalar_mul(g, words(k))
        ex, ey = _py_affine_mul_f2(k, ((gi[0], gi[1]), (gi[2], gi[3])), q)
        assert not inf and cv.affine_ints(got) == (ex[0], ex[1], ey[0], ey[1])
    rng = np.random.default_rng(2)
    for n, c in ((1, 4), (33, 5), (300, 8)):
        a, b = 3, 5
        bases = cv.make_bases(n, a, b)
        s = [int.from_bytes(rng.bytes(32), "little") % cv.r for _ in range(n)]
        scal = np.concatenate([words(v) for v in s])
        got, inf = cv.msm(bases, scal, c)
        k = sum(si * (a + b * i) for i, si in enumerate(s)) % cv.r
        exp, einf = cv.scalar_mul(g, words(k))
        assert inf == einf and np.array_equal(got, exp), (n, c)
