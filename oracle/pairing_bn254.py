"""ORACLE -- TEST INFRASTRUCTURE ONLY.  The alt_bn128 (BN254) optimal ate pairing and the Groth16 verification equation in
plain Python integers -- slow (seconds per pairing) and independent of everything else in this repository.  It follows the
published construction (Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), D-type twist with xi = 9 + u, loop count 6t + 2 with
t = 4965661367192848881, the two Frobenius corrections, final exponent (q^12 - 1)/r), i.e. what bellman_ce's
`verify_proof` computes through pairing_ce (groth16/src/groth16.rs:59-66, 98-105 -> third-party).  Pinned by bilinearity
and by the reference's own key/proof fixture where that carries a recoverable public input (tests/test_oracle_pairing.py)."""

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE_LOOP = 29793968203157093288
LOG_ATE = 63
MOD_COEFFS = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]      # w^12 = 18 w^6 - 82


def inv(a, n=Q):
    return pow(a % n, n - 2, n)


class F12:
    """element of Fq[w]/(w^12 - 18 w^6 + 82) as 12 coefficients"""
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [int(x) % Q for x in c] + [0] * (12 - len(c))

    @staticmethod
    def one(): return F12([1])
    @staticmethod
    def zero(): return F12([0])
    def __eq__(self, o): return self.c == (o.c if isinstance(o, F12) else F12([o]).c)
    def __add__(self, o): o = o if isinstance(o, F12) else F12([o]); return F12([a + b for a, b in zip(self.c, o.c)])
    def __sub__(self, o): o = o if isinstance(o, F12) else F12([o]); return F12([a - b for a, b in zip(self.c, o.c)])
    def __neg__(self): return F12([-a for a in self.c])
    def __mul__(self, o):
        if not isinstance(o, F12): return F12([a * o for a in self.c])
        t = [0] * 23
        for i, a in enumerate(self.c):
            if a:
                for j, b in enumerate(o.c):
                    t[i + j] += a * b
        for k in range(22, 11, -1):                          # w^k = 18 w^(k-6) - 82 w^(k-12)
            top = t[k]
            if top:
                t[k - 6] += 18 * top; t[k - 12] -= 82 * top
        return F12(t[:12])
    __rmul__ = __mul__
    def __pow__(self, e):
        r, b = F12.one(), self
        while e:
            if e & 1: r = r * b
            b = b * b; e >>= 1
        return r
    def inverse(self):
        """extended Euclid over Fq[w] against the modulus polynomial"""
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = self.c + [0], [82, 0, 0, 0, 0, 0, Q - 18, 0, 0, 0, 0, 0, 1]
        deg = lambda p: max([i for i, v in enumerate(p) if v] + [0])
        def poly_div(a, b):
            a = list(a); o = [0] * len(a); db = deg(b)
            for i in range(deg(a) - db, -1, -1):
                o[i] = a[db + i] * inv(b[db]) % Q
                for k in range(db + 1): a[k + i] = (a[k + i] - o[i] * b[k]) % Q
            return o
        while deg(low):
            r = poly_div(high, low) + [0] * 13
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] -= lm[i] * r[j]; new[i + j] -= low[i] * r[j]
            nm = [x % Q for x in nm]; new = [x % Q for x in new]
            lm, low, hm, high = nm, new, lm, low
        return F12([x * inv(low[0]) for x in lm[:12]])
    def __truediv__(self, o): return self * (o.inverse() if isinstance(o, F12) else inv(o))


W = F12([0, 1])


def embed_g1(p):
    return (F12([p[0]]), F12([p[1]]))


def twist_g2(p):
    """(x.c0, x.c1, y.c0, y.c1) on y^2 = x^3 + 3/(9+u) over Fq2 -> the isomorphic point over Fq12 on y^2 = x^3 + 3:
    u = w^6 - 9, then x -> x w^2, y -> y w^3"""
    x0, x1, y0, y1 = p
    nx = F12([x0 - 9 * x1, 0, 0, 0, 0, 0, x1]); ny = F12([y0 - 9 * y1, 0, 0, 0, 0, 0, y1])
    return (nx * W * W, ny * W * W * W)


def pt_double(p):
    x, y = p
    m = (x * x * 3) / (y * 2)
    nx = m * m - x * 2
    return (nx, m * (x - nx) - y)


def pt_add(p, q):
    if p is None: return q
    if q is None: return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2: return pt_double(p) if y1 == y2 else None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def linefunc(p1, p2, t):
    (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
    if not x1 == x2: m = (y2 - y1) / (x2 - x1)
    elif y1 == y2: m = (x1 * x1 * 3) / (y1 * 2)
    else: return xt - x1
    return m * (xt - x1) - (yt - y1)


FINAL_EXP = (Q ** 12 - 1) // R


def miller_loop(q, p):
    """q: twisted G2 point, p: embedded G1 point -> unreduced pairing value (before the final exponentiation)"""
    r_, f = q, F12.one()
    for i in range(LOG_ATE, -1, -1):
        f = f * f * linefunc(r_, r_, p); r_ = pt_double(r_)
        if ATE_LOOP & (1 << i):
            f = f * linefunc(r_, q, p); r_ = pt_add(r_, q)
    q1 = (q[0] ** Q, q[1] ** Q)
    nq2 = (q1[0] ** Q, -(q1[1] ** Q))
    f = f * linefunc(r_, q1, p); r_ = pt_add(r_, q1)
    return f * linefunc(r_, nq2, p)


def pairing(g2, g1):
    """e(g1, g2) with g1 = (x, y) ints, g2 = (x.c0, x.c1, y.c0, y.c1) ints; None = infinity"""
    if g1 is None or g2 is None: return F12.one()
    return miller_loop(twist_g2(g2), embed_g1(g1)) ** FINAL_EXP


def g1_add(p, q):
    """affine alt_bn128 G1 over ints (None = infinity)"""
    if p is None: return q
    if q is None: return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if (y1 + y2) % Q == 0: return None
        m = 3 * x1 * x1 * inv(2 * y1) % Q
    else:
        m = (y2 - y1) * inv(x2 - x1) % Q
    x3 = (m * m - x1 - x2) % Q
    return (x3, (m * (x1 - x3) - y1) % Q)


def g1_mul(p, k):
    acc = None
    while k:
        if k & 1: acc = g1_add(acc, p)
        p = g1_add(p, p); k >>= 1
    return acc


def groth16_verify(vk, proof, public_inputs):
    """bellman's verify_proof: e(A, B) == e(alpha, beta) e(sum_i x_i IC_i, gamma) e(C, delta), x_0 = 1.
    vk: dict alpha_g1, beta_g2, gamma_g2, delta_g2, ic (int tuples); proof: dict a, b, c"""
    if len(public_inputs) + 1 != len(vk["ic"]): return False
    acc = vk["ic"][0]
    for x, p in zip(public_inputs, vk["ic"][1:]):
        acc = g1_add(acc, g1_mul(p, x % R))
    lhs = pairing(proof["b"], proof["a"])
    rhs = pairing(vk["beta_g2"], vk["alpha_g1"]) * pairing(vk["gamma_g2"], acc) * pairing(vk["delta_g2"], proof["c"])
    return lhs == rhs
