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
