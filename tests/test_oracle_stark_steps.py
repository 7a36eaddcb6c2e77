"""CPU self-checks of the oracle's prover glue (oracle/stark_steps.c) against the mathematical
definitions the reference code implements (no GPU)."""
import numpy as np

P = 0xFFFFFFFF00000001


def test_fri_fold_is_evaluation_of_interpolant(orc):
    """Definition check of the oracle itself: pol2[g] = P_g(special_x / (shift * w^g))."""
    rng = np.random.default_rng(5)
    pol_bits, step_bits = 4, 2
    n2, nx = 1 << step_bits, 1 << (pol_bits - step_bits)
    pol = rng.integers(0, P, size=1 << pol_bits, dtype=np.uint64)           # base-field polynomial values
    pol3 = np.zeros(3 << pol_bits, np.uint64); pol3[0::3] = pol
    sx = np.array([12345, 0, 0], np.uint64)
    shift = 49; shift_inv = pow(shift, P - 2, P)
    out = orc.fri_fold(pol3, pol_bits, step_bits, sx, shift_inv)
    w, wx = orc.root(pol_bits), orc.root(pol_bits - step_bits)
    for g in range(n2):
        vals = [int(pol[i * n2 + g]) for i in range(nx)]
        y = 12345 * pow(shift * pow(w, g, P) % P, P - 2, P) % P
        # Lagrange interpolation over the points wx^i
        acc = 0
        for i in range(nx):
            num, den = 1, 1
            for j in range(nx):
                if j != i:
                    num = num * ((y - pow(wx, j, P)) % P) % P
                    den = den * ((pow(wx, i, P) - pow(wx, j, P)) % P) % P
            acc = (acc + vals[i] * num % P * pow(den, P - 2, P)) % P
        assert int(out[3 * g]) == acc and int(out[3 * g + 1]) == 0 and int(out[3 * g + 2]) == 0



def test_f3_batch_inverse_and_ntt_roundtrip(orc):
    rng = np.random.default_rng(1)
    v = rng.integers(0, P, size=3 * 37, dtype=np.uint64)
    inv = orc.f3_batch_inverse(v)                                   # polutils.rs:35-53
    for i in range(37):
        assert [int(x) for x in orc.f3_mul(v[3 * i:3 * i + 3], inv[3 * i:3 * i + 3])] == [1, 0, 0]
    x = rng.integers(0, P, size=3 * 64, dtype=np.uint64)
    assert np.array_equal(orc.f3_ntt(orc.f3_ntt(x, 6), 6, inverse=True), x)
    # F3G transform == per-limb base-field transform (roots are base-field elements)
    assert np.array_equal(orc.f3_ntt(x, 6), orc.ntt(x, 3, 6))


def test_xdivxsub_and_lev_definitions(orc):
    xi = np.array([5, 6, 7], np.uint64)
    nbits_ext = 4
    out = orc.xdivxsub(xi, nbits_ext).reshape(-1, 3)
    w = orc.root(nbits_ext)
    for k in range(1 << nbits_ext):
        x = 49 * pow(w, k, P) % P
        den = np.array([(x - 5) % P, P - 6, P - 7], np.uint64)
        back = orc.f3_mul(out[k], den)                               # x/(x-xi) * (x-xi) == x
        assert [int(v) for v in back] == [x, 0, 0]
    nbits = 3
    L = orc.lev(xi, nbits, False).reshape(-1, 3)                    # stark_gen.rs:416-430
    xis = orc.f3_mul(xi, [pow(49, P - 2, P), 0, 0])
    pw = [np.array([1, 0, 0], np.uint64)]
    for _ in range(7):
        pw.append(orc.f3_mul(pw[-1], xis))
    assert np.array_equal(orc.f3_ntt(np.concatenate(pw), nbits, inverse=True).reshape(-1, 3), L)


def test_calculate_z_closes_for_permuted_columns(orc):
    rng = np.random.default_rng(3)
    n = 64
    a = rng.integers(0, P, size=n, dtype=np.uint64)
    b = rng.permutation(a)
    gamma = 987654321
    num = np.zeros(3 * n, np.uint64); den = np.zeros(3 * n, np.uint64)
    num[0::3] = (a.astype(object) + gamma) % P
    den[0::3] = (b.astype(object) + gamma) % P
    z, ok = orc.calculate_z(num, den)                               # stark_gen.rs:653-666
    assert ok and [int(v) for v in z[:3]] == [1, 0, 0]
    den[0] = (int(den[0]) + 1) % P
    assert not orc.calculate_z(num, den)[1]
