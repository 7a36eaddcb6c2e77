"""Pins oracle/pairing_bn254.py (pure-Python optimal ate pairing + Groth16 verification equation): bilinearity on both
sides, the group order, and -- the anchor -- the reference's own fixture: groth16/test-vectors/proof.json verifies
against groth16/test-vectors/verification_key.json, with the public input recovered as 33 (an a * b = c circuit: 3 * 11).
The oracle's Groth16 prover restatement is then judged by this verifier."""
import json, pathlib, random, sys
import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import pairing_bn254 as PB  # noqa: E402
import pairing as PG  # noqa: E402
import groth16 as G  # noqa: E402
GOLD = ROOT / "tests" / "golden" / "groth16"
G1 = (1, 2)
G2 = (10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634,
      8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531)


def _fixture():
    vk = json.loads((GOLD / "verification_key.json").read_text()); pr = json.loads((GOLD / "proof.json").read_text())
    g1 = lambda d: (int(d["x"]), int(d["y"])); g2 = lambda d: (int(d["x"][0]), int(d["x"][1]), int(d["y"][0]), int(d["y"][1]))
    return (dict(alpha_g1=g1(vk["vk_alpha_1"]), beta_g2=g2(vk["vk_beta_2"]), gamma_g2=g2(vk["vk_gamma_2"]), delta_g2=g2(vk["vk_delta_2"]), ic=[g1(p) for p in vk["IC"]]),
            dict(a=g1(pr["pi_a"]), b=g2(pr["pi_b"]), c=g1(pr["pi_c"])))


def test_pairing_is_bilinear_and_of_order_r(orc):
    g = G.Groth16Oracle(orc, "bn254")
    assert g.g2.affine_ints(g.g2.generator()) == G2                     # the same generator, the same coordinate order
    e = PB.pairing(G2, G1)
    assert not e == PB.F12.one() and e ** PB.R == PB.F12.one()
    assert PB.pairing(G2, PB.g1_mul(G1, 5)) == e ** 5
    q3 = g.g2.affine_ints(g.mul(g.g2, g.g2.generator(), 3))             # [3]G2 from the C oracle
    assert PB.pairing(q3, G1) == e ** 3


def test_reference_proof_fixture_verifies():
    vk, pr = _fixture()
    assert PB.groth16_verify(vk, pr, [33])
    assert not PB.groth16_verify(vk, pr, [34])
    bad = dict(pr, c=PB.g1_add(pr["c"], G1))
    assert not PB.groth16_verify(vk, bad, [33])


def test_oracle_prover_is_accepted_by_the_verifier(orc):
    """the restated create_proof against the restated generate_parameters, judged by the pinned verifier"""
    g = G.Groth16Oracle(orc, "bn254"); rng = random.Random(21)
    r1cs, wit = G.synthetic_r1cs(g.r, 12, seed=8)
    P = g.setup(r1cs, *[rng.randrange(1, g.r) for _ in range(5)])
    pr = g.prove(P, wit, rng.randrange(g.r), rng.randrange(g.r))
    vk, proof, pub = G.verifier_inputs(g, P, pr, wit)
    assert PB.groth16_verify(vk, proof, pub)
    assert not PB.groth16_verify(vk, proof, [pub[0] + 1] + pub[1:])


def test_bls12_381_pairing_is_bilinear_and_accepts_the_oracle_prover(orc):
    """the BLS12-381 instance of the same code (other constants, M-type twist, no Frobenius corrections): no reference proof
    verifies it directly -- the recursion-gnark fixture carries a gnark commitment, and the reference's own test of it only
    asserts is_ok() -- so it is pinned by bilinearity against the C oracle's independent G2 arithmetic"""
    C = PG.BLS12_381
    g = G.Groth16Oracle(orc, "bls12_381"); rng = random.Random(23)
    G1 = g.g1.affine_ints(g.g1.generator()); G2 = g.g2.affine_ints(g.g2.generator())
    e = C.pairing(G2, G1)
    assert not e == C.F12.one() and e ** C.R == C.F12.one()
    assert C.pairing(G2, C.g1_mul(G1, 5)) == e ** 5
    assert C.pairing(g.g2.affine_ints(g.mul(g.g2, g.g2.generator(), 3)), G1) == e ** 3
    r1cs, wit = G.synthetic_r1cs(g.r, 12, seed=8)
    P = g.setup(r1cs, *[rng.randrange(1, g.r) for _ in range(5)])
    vk, proof, pub = G.verifier_inputs(g, P, g.prove(P, wit, rng.randrange(g.r), rng.randrange(g.r)), wit)
    assert C.groth16_verify(vk, proof, pub)
    assert not C.groth16_verify(vk, proof, [pub[0] + 1] + pub[1:])
