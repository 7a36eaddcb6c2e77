"""Pins the Groth16 oracle (oracle/groth16_impl.h, oracle/groth16.py): SURVEY.md 8(f)-2.  The reference reaches
the prover only through bellman_ce (not in the tree) and holds no proving key / witness / quotient vector, so
the restatement is pinned by algebra and by the reference's binary/JSON key fixtures (json_utils.rs:351-429)."""
import json, pathlib, random, sys
import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import groth16 as G  # noqa: E402
GOLD = ROOT / "tests" / "golden" / "groth16"
CURVES = ("bn254", "bls12_381")


@pytest.fixture(scope="module")
def g16(orc):
    return {cv: G.Groth16Oracle(orc, cv) for cv in CURVES}


@pytest.mark.parametrize("cv", CURVES)
def test_transform_matches_the_definition(g16, cv):
    g = g16[cv]; r = g.r; rng = random.Random(3)
    for log_n in (0, 1, 3, 5):
        n = 1 << log_n; x = [rng.randrange(r) for _ in range(n)]; w = g.omega(log_n)
        assert pow(w, n, r) == 1 and (n == 1 or pow(w, n // 2, r) == r - 1)
        xm = g.to_mont(g.fr_array(x))
        assert g.fr_ints(g.from_mont(g.ntt(xm))) == [sum(x[j] * pow(w, i * j, r) for j in range(n)) % r for i in range(n)]
        assert g.fr_ints(g.from_mont(g.ntt(xm, coset=True))) == [sum(x[j] * pow(7, j, r) * pow(w, i * j, r) for j in range(n)) % r for i in range(n)]
        assert g.fr_ints(g.from_mont(g.ntt(g.ntt(xm), inverse=True))) == x
        assert g.fr_ints(g.from_mont(g.ntt(g.ntt(xm, coset=True), inverse=True, coset=True))) == x


@pytest.mark.parametrize("cv", CURVES)
def test_quotient_is_the_polynomial_division(g16, cv):
    """h = (A B - C) / (X^n - 1) checked by evaluating both sides at a random point"""
    g = g16[cv]; r = g.r; rng = random.Random(9); log_n = 5; n = 1 << log_n
    a = [rng.randrange(r) for _ in range(n)]; b = [rng.randrange(r) for _ in range(n)]
    c = [x * y % r for x, y in zip(a, b)]                       # satisfied rows: A B - C vanishes on the domain
    M = lambda v: g.to_mont(g.fr_array(v))
    h = g.fr_ints(g.from_mont(g.quotient(M(a), M(b), M(c))))
    assert h[n - 1] == 0                                        # degree <= n - 2: bellman drops this coefficient
    coef = lambda ev: g.fr_ints(g.from_mont(g.ntt(M(ev), inverse=True)))
    ca, cb, cc = coef(a), coef(b), coef(c)
    x = rng.randrange(r); at = lambda p: sum(v * pow(x, i, r) for i, v in enumerate(p)) % r
    assert (at(ca) * at(cb) - at(cc)) % r == at(h) * (pow(x, n, r) - 1) % r


@pytest.mark.parametrize("cv,stem", [("bn254", "verification_key"), ("bls12_381", "verification_key_bls12381")])
def test_key_encoding_matches_the_reference_fixtures(g16, cv, stem):
    """pairing_ce's uncompressed point encoding, as read back by json_utils.rs:351-429 from the same files"""
    g = g16[cv]
    b = (GOLD / (stem + ".bin")).read_bytes()
    vk, used = g.vk_from_bytes(b)
    assert used == len(b)
    j = json.loads((GOLD / (stem + ".json")).read_text())
    num = lambda s: int(s, 16) if s.startswith("0x") else int(s)
    j1 = lambda d: (num(d["x"]), num(d["y"])); j2 = lambda d: (num(d["x"][0]), num(d["x"][1]), num(d["y"][0]), num(d["y"][1]))
    assert g.g1.affine_ints(vk["alpha_g1"]) == j1(j["vk_alpha_1"]) and g.g1.affine_ints(vk["beta_g1"]) == j1(j["vk_beta_1"])
    assert g.g1.affine_ints(vk["delta_g1"]) == j1(j["vk_delta_1"])
    assert g.g2.affine_ints(vk["beta_g2"]) == j2(j["vk_beta_2"]) and g.g2.affine_ints(vk["gamma_g2"]) == j2(j["vk_gamma_2"])
    assert g.g2.affine_ints(vk["delta_g2"]) == j2(j["vk_delta_2"])
    assert [g.g1.affine_ints(p) for p in vk["ic"]] == [j1(d) for d in j["IC"]]
    assert all(g.g1.on_curve(p) for p in [vk["alpha_g1"], vk["beta_g1"], vk["delta_g1"]] + vk["ic"])
    assert all(g.g2.on_curve(p) for p in (vk["beta_g2"], vk["gamma_g2"], vk["delta_g2"]))
    # and the encoder is the decoder's inverse
    enc = b"".join([g.enc_point(g.g1, vk["alpha_g1"]), g.enc_point(g.g1, vk["beta_g1"]), g.enc_point(g.g2, vk["beta_g2"]), g.enc_point(g.g2, vk["gamma_g2"]),
                    g.enc_point(g.g1, vk["delta_g1"]), g.enc_point(g.g2, vk["delta_g2"])])
    assert enc == b[:len(enc)]


@pytest.mark.parametrize("cv", CURVES)
def test_proof_is_the_valid_one(g16, cv):
    """the prover restatement (transforms + density-indexed multi-scalar sums + blinding) lands on the unique
    proof the verification equation admits for (witness, r, s), computed independently from the trapdoor"""
    g = g16[cv]; r = g.r; rng = random.Random(11)
    r1cs, wit = G.synthetic_r1cs(r, 40, seed=5)
    P = g.setup(r1cs, *[rng.randrange(1, r) for _ in range(5)])
    assert sum(p is None for p in P["l"]) == 1                 # the unused wire
    rr, ss = rng.randrange(r), rng.randrange(r)
    pr, ex = g.prove(P, wit, rr, ss), g.expected_proof(P, wit, rr, ss)
    for k in "abc":
        assert np.array_equal(pr[k], ex[k]), k
    bad = list(wit); bad[-2] = (bad[-2] + 1) % r               # an unsatisfying witness must not verify
    assert not np.array_equal(g.prove(P, bad, rr, ss)["c"], g.expected_proof(P, bad, rr, ss)["c"])
    assert json.loads(g.proof_json(pr))["curve"] == g.c["json"]


def test_r1cs_fixture_header_parses():
    """the reference's own .r1cs (groth16/test-vectors/mycircuit_bls12381.r1cs) through the writer's inverse"""
    import struct
    b = (GOLD / "mycircuit_bls12381.r1cs").read_bytes()
    assert b[:4] == b"r1cs" and struct.unpack("<II", b[4:12]) == (1, 3)
    o = 12; secs = {}
    for _ in range(3):
        t, n = struct.unpack("<IQ", b[o:o + 12]); secs[t] = b[o + 12:o + 12 + n]; o += 12 + n
    fs = struct.unpack("<I", secs[1][:4])[0]
    assert fs == 32 and int.from_bytes(secs[1][4:36], "little") == G.CURVES["bls12_381"]["r"]
    n_wires, n_out, n_in, n_prv, _labels, n_cons = struct.unpack("<IIIIQI", secs[1][36:])
    assert (n_wires, n_out + n_in + n_prv, n_cons) == (4, 3, 1)
