"""Derives and checks the constants of csrc/msm.hip's endomorphism split for BN254 G1 (y^2 = x^3 + 3):
beta (a cube root of unity in Fq) and lambda (in Fr) with (beta x, y) = [lambda](x, y); a short basis (a1, b1), (a2, b2) of the
lattice {(x, y): x + y lambda = 0 mod r} by the extended Euclid walk of Gallant-Lambert-Vanstone; g1 = floor(2^256 b2 / r),
g2 = floor(-2^256 b1 / r); the bound |k1|, |k2| < 2^128 of the split the device computes with them.  Prints the 32-bit words
the header of bn254::g1 in msm.hip carries.  (BLS12-381 G1 needs no lattice: lambda = z^2 - 1 < 2^128 and lambda^2 + lambda + 1 = 0
mod r, so k = k1 + k2 lambda is a plain division; its beta is x([lambda]G) / x(G), checked the same way when it was derived.)
python tools/glv_constants.py"""
import math
import random

q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
r = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def cube_root_of_unity(p):
    g = 2
    while pow(g, (p - 1) // 3, p) == 1:
        g += 1
    return pow(g, (p - 1) // 3, p)


def add(P, Q):
    if P is None: return Q
    if Q is None: return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if (y1 + y2) % q == 0: return None
        l = 3 * x1 * x1 * pow(2 * y1, -1, q) % q
    else:
        l = (y2 - y1) * pow(x2 - x1, -1, q) % q
    x3 = (l * l - x1 - x2) % q
    return (x3, (l * (x1 - x3) - y1) % q)


def mul(k, P):
    R = None
    while k:
        if k & 1: R = add(R, P)
        P = add(P, P); k >>= 1
    return R


def main():
    G = (1, 2)
    b0, l0 = cube_root_of_unity(q), cube_root_of_unity(r)
    pairs = [(b, l) for b in (b0, b0 * b0 % q) for l in (l0, l0 * l0 % r) if mul(l, G) == (b * G[0] % q, G[1])]
    beta, lam = min(pairs, key=lambda bl: bl[1])                       # the pair with the smaller lambda
    rs = [(r, 1, 0), (lam, 0, 1)]
    while rs[-1][0]:
        (r0, s0, t0), (r1, s1, t1) = rs[-2], rs[-1]
        rs.append((r0 - r0 // r1 * r1, s0 - r0 // r1 * s1, t0 - r0 // r1 * t1))
    i = next(i for i, x in enumerate(rs) if x[0] < math.isqrt(r))
    a1, b1 = rs[i][0], -rs[i][2]
    (rl, _, tl), (rl2, _, tl2) = rs[i - 1], rs[i + 1]
    a2, b2 = (rl, -tl) if rl * rl + tl * tl <= rl2 * rl2 + tl2 * tl2 else (rl2, -tl2)
    assert (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0 and a1 * b2 - a2 * b1 == r and b1 < 0 < b2
    g1, g2 = (b2 << 256) // r, ((-b1) << 256) // r
    worst = 0
    for k in [0, 1, 2, r - 1, r - 2, lam, lam - 1, lam + 1, r // 2, 1 << 253, 1 << 128, (1 << 128) - 1, a2, -b1, b2, (1 << 256) - 1] + \
             [random.randrange(r) for _ in range(200000)]:
        c1, c2 = (k * g1) >> 256, (k * g2) >> 256
        k1, k2 = k - c1 * a1 - c2 * a2, -c1 * b1 - c2 * b2
        assert (k1 + k2 * lam - k) % r == 0
        worst = max(worst, abs(k1).bit_length(), abs(k2).bit_length())
    assert worst <= 128
    words = lambda v, n: ", ".join("0x%08xu" % ((v >> (32 * j)) & 0xFFFFFFFF) for j in range(n))
    print("beta   =", beta); print("lambda =", lam)
    print("a1 = b2 =", a1, " -b1 =", -b1, " a2 =", a2, " longest half:", worst, "bits")
    print("#define GLV_BETA_STD", words(beta * (1 << 256) % q, 8))
    print("#define GLV_G1", words(g1, 3)); print("#define GLV_G2", words(g2, 5))
    print("#define GLV_A1", words(a1, 2)); print("#define GLV_A2", words(a2, 4))
    print("#define GLV_NB1", words(-b1, 4)); print("#define GLV_B2", words(b2, 2))


if __name__ == "__main__":
    main()
