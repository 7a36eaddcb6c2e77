"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Optimal ate pairings of alt_bn128 (BN254) and BLS12-381 and the Groth16 verification
equation in plain Python integers -- slow (seconds per pairing) and independent of everything else in this repository.
Published constructions: Fq12 = Fq[w]/(w^12 - c6 w^6 - c0); BN254: w^12 = 18 w^6 - 82, D-type twist (xi = 9 + u), loop
count 6t + 2 with the two Frobenius corrections; BLS12-381: w^12 = 2 w^6 - 2, M-type twist (xi = 1 + u), loop count |x| =
0xd201000000010000 (the sign of x only inverts the pairing value, the same on both sides of an equation); final exponent
(q^12 - 1)/r.  This is what bellman_ce's `verify_proof` computes through pairing_ce (groth16/src/groth16.rs:59-66, 98-105
-> third-party).  Pinned by bilinearity and by the reference's own fixtures (tests/test_oracle_pairing.py)."""


def _inv(a, n):
    return pow(a % n, n - 2, n)


class Curve:
    def __init__(self, name, q, r, c6, c0, xi0, ate_loop, bn_corrections, twist_divides):
        self.name, self.Q, self.R = name, q, r
        self.c6, self.c0, self.xi0 = c6, c0, xi0            # w^12 = c6 w^6 + c0 ; u = w^6 - xi0
        self.ate_loop, self.log_ate = ate_loop, ate_loop.bit_length() - 1
        self.bn_corrections, self.twist_divides = bn_corrections, twist_divides
        self.final_exp = (q ** 12 - 1) // r
        C = self

        class F12:
            __slots__ = ("c",)

            def __init__(self, c):
                self.c = [int(x) % C.Q for x in c] + [0] * (12 - len(c))
            @staticmethod
            def one(): return F12([1])
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
                for k in range(22, 11, -1):
                    top = t[k]
                    if top:
                        t[k - 6] += C.c6 * top; t[k - 12] += C.c0 * top
                return F12(t[:12])
            __rmul__ = __mul__
            def __pow__(self, e):
                r_, b = F12.one(), self
                while e:
                    if e & 1: r_ = r_ * b
                    b = b * b; e >>= 1
                return r_
            def inverse(self):
                """extended Euclid over Fq[w] against the modulus polynomial"""
                q = C.Q
                lm, hm = [1] + [0] * 12, [0] * 13
                low, high = self.c + [0], [(-C.c0) % q, 0, 0, 0, 0, 0, (-C.c6) % q, 0, 0, 0, 0, 0, 1]
                deg = lambda p: max([i for i, v in enumerate(p) if v] + [0])
                def poly_div(a, b):
                    a = list(a); o = [0] * len(a); db = deg(b)
                    for i in range(deg(a) - db, -1, -1):
                        o[i] = a[db + i] * _inv(b[db], q) % q
                        for k in range(db + 1): a[k + i] = (a[k + i] - o[i] * b[k]) % q
                    return o
                while deg(low):
                    r_ = poly_div(high, low) + [0] * 13
                    nm, new = list(hm), list(high)
                    for i in range(13):
                        for j in range(13 - i):
                            nm[i + j] -= lm[i] * r_[j]; new[i + j] -= low[i] * r_[j]
                    nm = [x % q for x in nm]; new = [x % q for x in new]
                    lm, low, hm, high = nm, new, lm, low
                return F12([x * _inv(low[0], q) for x in lm[:12]])
            def __truediv__(self, o): return self * (o.inverse() if isinstance(o, F12) else _inv(o, C.Q))
        self.F12 = F12
        self.W = F12([0, 1])

    # ---- curve arithmetic over Fq12 (affine) ----
    @staticmethod
    def _double(p):
        x, y = p
        m = (x * x * 3) / (y * 2)
        nx = m * m - x * 2
        return (nx, m * (x - nx) - y)
    def _add(self, p, q):
        (x1, y1), (x2, y2) = p, q
        if x1 == x2: return self._double(p) if y1 == y2 else None
        m = (y2 - y1) / (x2 - x1)
        nx = m * m - x1 - x2
        return (nx, m * (x1 - nx) - y1)
    @staticmethod
    def _line(p1, p2, t):
        (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
        if not x1 == x2: m = (y2 - y1) / (x2 - x1)
        elif y1 == y2: m = (x1 * x1 * 3) / (y1 * 2)
        else: return xt - x1
        return m * (xt - x1) - (yt - y1)

    def twist(self, p):
        """(x.c0, x.c1, y.c0, y.c1) over Fq2 = Fq[u]/(u^2 + 1) -> the isomorphic point over Fq12 on the G1 curve equation:
        u = w^6 - xi0, then (x w^2, y w^3) for a D-type twist, (x / w^2, y / w^3) for an M-type one"""
        F12, W = self.F12, self.W
        x0, x1, y0, y1 = p
        nx = F12([x0 - self.xi0 * x1, 0, 0, 0, 0, 0, x1]); ny = F12([y0 - self.xi0 * y1, 0, 0, 0, 0, 0, y1])
        w2, w3 = W * W, W * W * W
        return (nx / w2, ny / w3) if self.twist_divides else (nx * w2, ny * w3)

    def miller_loop(self, q, p):
        r_, f = q, self.F12.one()
        for i in range(self.log_ate - 1, -1, -1):
            f = f * f * self._line(r_, r_, p); r_ = self._double(r_)
            if self.ate_loop & (1 << i):
                f = f * self._line(r_, q, p); r_ = self._add(r_, q)
        if self.bn_corrections:
            q1 = (q[0] ** self.Q, q[1] ** self.Q)
            nq2 = (q1[0] ** self.Q, -(q1[1] ** self.Q))
            f = f * self._line(r_, q1, p); r_ = self._add(r_, q1)
            f = f * self._line(r_, nq2, p)
        return f

    def pairing(self, g2, g1):
        """g1 = (x, y) ints, g2 = (x.c0, x.c1, y.c0, y.c1) ints; None = infinity"""
        if g1 is None or g2 is None: return self.F12.one()
        return self.miller_loop(self.twist(g2), (self.F12([g1[0]]), self.F12([g1[1]]))) ** self.final_exp

    # ---- G1 over ints ----
    def g1_add(self, p, q):
        Q = self.Q
        if p is None: return q
        if q is None: return p
        (x1, y1), (x2, y2) = p, q
        if x1 == x2:
            if (y1 + y2) % Q == 0: return None
            m = 3 * x1 * x1 * _inv(2 * y1, Q) % Q
        else:
            m = (y2 - y1) * _inv(x2 - x1, Q) % Q
        x3 = (m * m - x1 - x2) % Q
        return (x3, (m * (x1 - x3) - y1) % Q)
    def g1_mul(self, p, k):
        acc = None
        while k:
            if k & 1: acc = self.g1_add(acc, p)
            p = self.g1_add(p, p); k >>= 1
        return acc

    def groth16_verify(self, vk, proof, public_inputs):
        """bellman's verify_proof: e(A, B) == e(alpha, beta) e(sum_i x_i IC_i, gamma) e(C, delta), x_0 = 1.
        vk: dict alpha_g1, beta_g2, gamma_g2, delta_g2, ic (int tuples); proof: dict a, b, c"""
        if len(public_inputs) + 1 != len(vk["ic"]): return False
        acc = vk["ic"][0]
        for x, p in zip(public_inputs, vk["ic"][1:]):
            acc = self.g1_add(acc, self.g1_mul(p, x % self.R))
        lhs = self.pairing(proof["b"], proof["a"])
        rhs = self.pairing(vk["beta_g2"], vk["alpha_g1"]) * self.pairing(vk["gamma_g2"], acc) * self.pairing(vk["delta_g2"], proof["c"])
        return lhs == rhs


BN254 = Curve("bn254", 21888242871839275222246405745257275088696311157297823662689037894645226208583,
              21888242871839275222246405745257275088548364400416034343698204186575808495617, 18, -82, 9,
              29793968203157093288, True, False)
BLS12_381 = Curve("bls12_381", 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
                  0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, 2, -2, 1,
                  0xd201000000010000, False, True)
