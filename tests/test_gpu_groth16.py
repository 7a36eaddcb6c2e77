"""Parity of the Groth16 device path (csrc/groth16.hip, frntt_impl.hip.h, groth16_impl.hip.h; through the C ABI) against
the oracle (oracle/groth16_impl.h, oracle/groth16.py) -- bit exact: field elements as Montgomery limbs, proof
points as affine coordinates.  SURVEY.md 8(f)-2."""
import importlib, json, pathlib, random, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import groth16 as G  # noqa: E402
CURVES = (("bn254", "BN128"), ("bls12_381", "BLS12381"))


@pytest.fixture(scope="module", autouse=True)
def _gpu(zk):
    assert zk.lib().zk_device_count() >= 1, "no GPU visible (the product has no CPU fallback)"
    zk.init(0)


@pytest.fixture(scope="module")
def g16(orc):
    return {cv: G.Groth16Oracle(orc, cv) for cv, _ in CURVES}


@pytest.fixture(scope="module")
def dev(zk):
    return importlib.import_module("eigen_zkvm_amd.groth16")


def rand_fr(g, rng, n):
    return g.to_mont(g.fr_array([rng.randrange(g.r) for _ in range(n)]))


@pytest.mark.parametrize("cv,tag", CURVES)
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 12])
def test_transforms_match_oracle(g16, dev, cv, tag, log_n):
    """every radix plan (remainder pass of 2 / 4 first, then radix 8), all four EvaluationDomain transforms"""
    g = g16[cv]; rng = random.Random(1000 + log_n)
    x = rand_fr(g, rng, 1 << log_n)
    for inverse in (False, True):
        for coset in (False, True):
            got = dev.fr_ntt(x, tag, inverse=inverse, coset=coset)
            assert np.array_equal(got, g.ntt(x, inverse=inverse, coset=coset)), (log_n, inverse, coset)


@pytest.mark.parametrize("cv,tag", CURVES)
def test_transform_edge_values_and_round_trip(zk, g16, dev, cv, tag):
    g = g16[cv]; r = g.r; log_n = 14; n = 1 << log_n; rng = random.Random(5)
    vals = [0, 1, r - 1, r - 2, 2, (r - 1) // 2] + [rng.randrange(r) for _ in range(n - 6)]
    x = g.to_mont(g.fr_array(vals))
    d = zk.DevArray.from_host(x.reshape(-1))
    dev.fr_ntt(d, tag); dev.fr_ntt(d, tag, inverse=True)
    assert np.array_equal(d.to_host().reshape(-1, 4), x)
    dev.fr_ntt(d, tag, coset=True)
    assert np.array_equal(d.to_host().reshape(-1, 4), g.ntt(x, coset=True))
    dev.fr_ntt(d, tag, inverse=True, coset=True)
    assert np.array_equal(d.to_host().reshape(-1, 4), x)
    z = np.zeros_like(x)
    assert np.array_equal(dev.fr_ntt(z, tag), z)                            # all-zero vector
    one_hot = z.copy(); one_hot[0] = g.to_mont(g.fr_array([1]))[0]
    assert np.array_equal(dev.fr_ntt(one_hot, tag), np.tile(one_hot[0], (n, 1)))   # delta -> constant


@pytest.mark.parametrize("cv,tag", CURVES)
def test_transform_is_linear_at_2_20(zk, g16, dev, cv, tag):
    """a size the oracle is not asked to match element by element: adding a one-hot vector at position p changes
    the transform by the geometric row w^(p i), and the inverse transform returns the input"""
    g = g16[cv]; log_n = 20; n = 1 << log_n
    rng = np.random.default_rng(7)
    a = np.concatenate([rng.integers(0, 2**64, size=(n, 3), dtype=np.uint64), rng.integers(0, 2**60, size=(n, 1), dtype=np.uint64)], axis=1)
    p = 12345; R = (1 << 256) % g.r                                        # rows are Montgomery residues < 2^252 < r
    e = a.copy(); e[p] = g.fr_array([(g.fr_ints(a[p:p + 1])[0] + R) % g.r])[0]
    fa = dev.fr_ntt(a, tag); fe_ = dev.fr_ntt(e, tag)
    w = g.omega(log_n)
    for i in (0, 1, 2, 77, n // 2, n - 1):
        d = (g.fr_ints(fe_[i:i + 1])[0] - g.fr_ints(fa[i:i + 1])[0]) % g.r
        assert d == pow(w, p * i, g.r) * R % g.r, i
    assert np.array_equal(dev.fr_ntt(fa, tag, inverse=True), a)


@pytest.mark.parametrize("cv,tag", CURVES)
@pytest.mark.parametrize("log_n", [1, 4, 9, 11])
def test_quotient_matches_oracle(zk, g16, dev, cv, tag, log_n):
    g = g16[cv]; rng = random.Random(31 + log_n); n = 1 << log_n
    a, b = rand_fr(g, rng, n), rand_fr(g, rng, n)
    sat = g.to_mont(g.fr_array([x * y % g.r for x, y in zip(g.fr_ints(g.from_mont(a)), g.fr_ints(g.from_mont(b)))]))
    for c in (sat, rand_fr(g, rng, n)):                                    # satisfied rows, and arbitrary ones
        da, db, dc = (zk.DevArray.from_host(v.reshape(-1)) for v in (a, b, c))
        dev.fr_quotient(da, db, dc, tag)
        assert np.array_equal(da.to_host().reshape(-1, 4), g.quotient(a, b, c))


def _case(g, n_mul, seed):
    rng = random.Random(seed)
    r1cs, wit = G.synthetic_r1cs(g.r, n_mul, seed=seed)
    P = g.setup(r1cs, *[rng.randrange(1, g.r) for _ in range(5)])
    return r1cs, wit, P, rng.randrange(g.r), rng.randrange(g.r)


@pytest.mark.parametrize("cv,tag", CURVES)
@pytest.mark.parametrize("n_mul", [6, 40, 300])
def test_proof_matches_oracle_and_the_verification_equation(zk, g16, dev, cv, tag, n_mul):
    """the device proof == the oracle's create_proof restatement == the unique valid proof for (witness, r, s)
    computed in the exponent from the trapdoor; proof.json carries the same points"""
    g = g16[cv]
    r1cs, wit, P, rr, ss = _case(g, n_mul, 100 + n_mul)
    S = dev.Groth16Setup(tag, g.r1cs_bytes(r1cs), g.params_bytes(P))
    assert (S.n_wires, S.n_inputs, S.domain_log) == (r1cs["n_wires"], P["cir"]["num_inputs"], P["cir"]["log_m"])
    w = dev.wtns_values(g.wtns_bytes(wit), tag)
    assert g.fr_ints(w) == wit
    js, pts = S.prove(w, rr, ss)
    exp = g.expected_proof(P, wit, rr, ss)
    nl = g.nl
    assert np.array_equal(pts[:2 * nl], exp["a"]) and np.array_equal(pts[2 * nl:6 * nl], exp["b"]) and np.array_equal(pts[6 * nl:], exp["c"])
    if n_mul <= 40:
        orc_pr = g.prove(P, wit, rr, ss)
        assert js == json.loads(g.proof_json(orc_pr))
        # the quotient the device fed to the `h` sum
        d_h = zk.DevArray(4 * ((1 << S.domain_log) - 1), zero=True)
        js2, _ = S.prove(zk.DevArray.from_host(w.reshape(-1)), rr, ss, d_h=d_h)
        assert js2 == js and g.fr_ints(d_h.to_host()) == orc_pr["h"]
    else:
        assert js == json.loads(g.proof_json(exp))
    # fresh r, s re-randomise the proof; the same (r, s) reproduces it
    js3, _ = S.prove(w)
    assert js3 != js and S.prove(w, rr, ss)[0] == js
    S.free()


def test_setup_and_prove_errors(zk, g16, dev):
    g = g16["bn254"]
    r1cs, wit, P, rr, ss = _case(g, 6, 3)
    rb, pb = g.r1cs_bytes(r1cs), g.params_bytes(P)
    with pytest.raises(zk.ZkError, match="unknown curve"):
        dev.Groth16Setup("BN254", rb, pb)
    with pytest.raises(zk.ZkError, match="Invalid magic number"):
        dev.Groth16Setup("BN128", b"xxxx" + rb[4:], pb)
    with pytest.raises(zk.ZkError, match="prime is not the scalar field"):
        dev.Groth16Setup("BLS12381", rb, pb)
    with pytest.raises(zk.ZkError, match="truncated"):
        dev.Groth16Setup("BN128", rb, pb[:-7])
    r2, _ = G.synthetic_r1cs(g.r, 9, seed=4)
    with pytest.raises(zk.ZkError, match="proving key"):
        dev.Groth16Setup("BN128", g.r1cs_bytes(r2), pb)                    # key of another circuit
    S = dev.Groth16Setup("BN128", rb, pb)
    w = g.fr_array(wit)
    with pytest.raises(zk.ZkError, match="wires"):
        S.prove(w[:-1], rr, ss)
    bad = w.copy(); bad[3] = g.fr_array([g.r])[0]
    with pytest.raises(zk.ZkError, match="not a canonical field element"):
        S.prove(bad, rr, ss)
    with pytest.raises(zk.ZkError, match="Invalid file header"):
        dev.wtns_values(b"wtnx" + g.wtns_bytes(wit)[4:], "BN128")
    with pytest.raises(zk.ZkError, match="invalid curve prime"):
        dev.wtns_values(g.wtns_bytes(wit), "BLS12381")
    S.free()


@pytest.mark.parametrize("cv,tag", CURVES)
def test_corner_circuits(zk, g16, dev, cv, tag):
    """shapes the readers and the density bookkeeping must survive: public inputs next to public outputs, a domain
    that is exactly full, a circuit whose only rows are bellman's input rows, sections of the .r1cs in another order"""
    import struct
    g = g16[cv]; r = g.r; rng = random.Random(77)
    td = lambda: [rng.randrange(1, r) for _ in range(5)]

    def check(r1cs, wit, rb=None):
        P = g.setup(r1cs, *td())
        S = dev.Groth16Setup(tag, rb or g.r1cs_bytes(r1cs), g.params_bytes(P))
        rr, ss = rng.randrange(r), rng.randrange(r)
        _, pts = S.prove(g.fr_array(wit), rr, ss)
        exp = g.expected_proof(P, wit, rr, ss); nl = g.nl
        assert np.array_equal(pts[:2 * nl], exp["a"]) and np.array_equal(pts[2 * nl:6 * nl], exp["b"]) and np.array_equal(pts[6 * nl:], exp["c"])
        S.free()
        return P

    # (1) one public output, two public inputs: out = (in1 + 3 in2) * prv ; aux = out * out
    w = [1, 0, 5, 7, 11, 0]
    w[1] = (w[2] + 3 * w[3]) * w[4] % r; w[5] = w[1] * w[1] % r
    r1 = dict(n_wires=6, n_pub_out=1, n_pub_in=2, n_prv_in=1,
              constraints=[([(2, 1), (3, 3)], [(4, 1)], [(1, 1)]), ([(1, 1)], [(1, 1)], [(5, 1)])])
    P = check(r1, w)
    assert P["cir"]["num_inputs"] == 4 and P["cir"]["log_m"] == 3          # 2 rows + 4 input rows -> 8
    # (2) a second generated shape (one public output, two private inputs)
    r2, w2 = G.synthetic_r1cs(r, 5, n_pub=1, n_prv=2, seed=9)
    check(r2, w2)
    # (3) no constraint at all: the domain holds the input rows only, h is empty or a single zero
    r3 = dict(n_wires=3, n_pub_out=1, n_pub_in=0, n_prv_in=1, constraints=[])
    check(r3, [1, 42, 99])
    # (4) the same file with its sections in the order 3, 2, 1 (r1cs_file.rs:206-216 indexes them by type)
    b = g.r1cs_bytes(r1); o = 12; secs = []
    for _ in range(3):
        t, n = struct.unpack("<IQ", b[o:o + 12]); secs.append(b[o:o + 12 + n]); o += 12 + n
    check(r1, w, rb=b[:12] + secs[2] + secs[1] + secs[0])


@pytest.mark.parametrize("cv,tag", CURVES)
def test_quotient_of_a_2_16_row_circuit_matches_oracle(zk, g16, dev, cv, tag):
    """the bench's synthetic circuit (tools/groth16_bench.py) at 2^16 rows: the quotient the device feeds to the `h` sum
    == the oracle's, coefficient by coefficient, and its dropped top coefficient is zero (the system is satisfied)"""
    sys.path.insert(0, str(ROOT / "tools"))
    import groth16_bench as GB
    g = g16[cv]; log_rows = 16; m = 1 << log_rows
    rb, wit, ni, n_wires = GB.make_circuit(g.r, log_rows)
    pb = GB.make_params(zk, dev, tag, ni, n_wires, log_rows, GB.density(rb, ni, n_wires))
    S = dev.Groth16Setup(tag, rb, pb)
    d_h = zk.DevArray(4 * (m - 1), zero=True)
    S.prove(zk.DevArray.from_host(wit.reshape(-1)), 5, 7, d_h=d_h)
    # the same rows for the oracle, straight from the file's fixed-shape records
    term = np.dtype([("wire", "<u4"), ("coef", "<u8", 4)])
    row = np.dtype([("na", "<u4"), ("a", term, 2), ("nb", "<u4"), ("b", term, 1), ("nc", "<u4"), ("c", term, 1)])
    rec = np.frombuffer(rb, dtype=row, count=m - ni, offset=100)
    w = g.fr_ints(wit)
    ev = lambda terms: [sum(int(t["coef"][0]) * w[int(t["wire"])] for t in r_) % g.r for r_ in terms]
    a = ev(rec["a"]) + [w[i] for i in range(ni)]; b = ev(rec["b"]) + [0] * ni; c = ev(rec["c"]) + [0] * ni
    M = lambda v: g.to_mont(g.fr_array(v))
    hq = g.from_mont(g.quotient(M(a), M(b), M(c)))
    assert g.fr_ints(hq[m - 1:]) == [0]
    assert np.array_equal(d_h.to_host().reshape(-1, 4), hq[:m - 1])
    S.free()


def test_plain_key_arrays_give_the_same_proof(zk, g16, dev, monkeypatch):
    """keys of 2^24 bases and more skip the window tables (24-bit point index): the same proof either way"""
    g = g16["bn254"]
    r1cs, wit, P, rr, ss = _case(g, 40, 140)
    rb, pb, w = g.r1cs_bytes(r1cs), g.params_bytes(P), g.fr_array(wit)
    S = dev.Groth16Setup("BN128", rb, pb); js_tab, pts_tab = S.prove(w, rr, ss); S.free()
    monkeypatch.setenv("ZK_GROTH16_NO_TABLES", "1")
    S = dev.Groth16Setup("BN128", rb, pb); js_plain, pts_plain = S.prove(w, rr, ss); S.free()
    assert js_plain == js_tab and np.array_equal(pts_plain, pts_tab)
    exp = g.expected_proof(P, wit, rr, ss)
    assert np.array_equal(pts_plain[:2 * g.nl], exp["a"]) and np.array_equal(pts_plain[6 * g.nl:], exp["c"])


@pytest.mark.parametrize("cv,tag", CURVES)
def test_device_proof_is_accepted_by_the_pairing_verifier(zk, g16, dev, cv, tag):
    """the verification equation itself, e(A, B) = e(alpha, beta) e(sum x_i IC_i, gamma) e(C, delta), evaluated by the
    pure-Python pairing (oracle/pairing.py) whose BN254 instance accepts the reference's own proof fixture
    (tests/test_oracle_pairing.py)"""
    import pairing as PG
    C = PG.BN254 if cv == "bn254" else PG.BLS12_381
    g = g16[cv]
    r1cs, wit, P, _, _ = _case(g, 12, 77)
    S = dev.Groth16Setup(tag, g.r1cs_bytes(r1cs), g.params_bytes(P))
    js, _ = S.prove(g.fr_array(wit))                                        # r, s drawn by the host mirror
    S.free()
    vk, proof, pub = G.verifier_inputs(g, P, js, wit)
    assert C.groth16_verify(vk, proof, pub)
    assert not C.groth16_verify(vk, proof, [(pub[0] + 1) % g.r] + pub[1:])


def test_wtns_fixture_of_the_reference_is_read(zk, dev):
    """test/single/witness.wtns of the reference (a snarkjs-written BN254 witness): [1, 11210000, 1121, 10000]"""
    b = (ROOT / "tests" / "golden" / "groth16" / "witness.wtns").read_bytes()
    w = dev.wtns_values(b, "BN128")
    assert w.shape == (4, 4) and w[:, 0].tolist() == [1, 11210000, 1121, 10000] and not w[:, 1:].any()


def test_reference_r1cs_fixture_end_to_end(zk, g16, dev):
    """groth16/test-vectors/mycircuit_bls12381.r1cs (c <== a * b, circom-written): the file as it is through the device
    reader, a key from the restated setup, witness 3 * 11 = 33, and the pairing verifier on the result"""
    import pairing as PG
    g = g16["bls12_381"]; rng = random.Random(3)
    rb = (ROOT / "tests" / "golden" / "groth16" / "mycircuit_bls12381.r1cs").read_bytes()
    prime, r1cs = G.read_r1cs(rb)
    assert prime == g.r and (r1cs["n_wires"], len(r1cs["constraints"])) == (4, 1)
    wit = [1, 33, 3, 11]                                                    # ONE, out c, in a, in b
    (A, B, Cc), = r1cs["constraints"]
    ev = lambda lc: sum(c * wit[j] for j, c in lc) % g.r
    assert ev(A) * ev(B) % g.r == ev(Cc)                                    # circom writes -a * b = -c
    P = g.setup(r1cs, *[rng.randrange(1, g.r) for _ in range(5)])
    S = dev.Groth16Setup("BLS12381", rb, g.params_bytes(P))
    js, _ = S.prove(g.fr_array(wit))
    S.free()
    vk, proof, pub = G.verifier_inputs(g, P, js, wit)
    assert pub == [33] and PG.BLS12_381.groth16_verify(vk, proof, pub)
    assert not PG.BLS12_381.groth16_verify(vk, proof, [34])
