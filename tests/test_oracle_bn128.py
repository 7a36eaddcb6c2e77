"""Pins the oracle's BN128-field hashing (oracle/bn128_hash.c) with the reference's own known answers:
poseidon_bn128_opt.rs:233-300, linearhash_bn128.rs:141-175, merklehash_bn128.rs:271-299."""
import numpy as np


def test_poseidon_known_answers(orc):
    h = orc.bn128()
    kat = [([1], 0x29176100eaa962bdc1fe6c654d6a3c130e96a4d1168b33848b897dc502820133),
           ([1, 2], 0x115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a),
           ([1, 2, 0, 0, 0], 0x024058dd1e168f34bac462b6fffe58fd69982807e9884c1c6148182319cee427),
           ([1, 2, 0, 0, 0, 0], 0x21e82f465e00a15965e97a44fe3c30f3bf5279d8bf37d4e65765b6c2550f42a1),
           ([3, 4, 0, 0, 0], 0x0cd93f1bab9e8c9166ef00f2a1b0e1d66d6a4145e596abe0526247747cc71214),
           ([3, 4, 0, 0, 0, 0], 0x1b1caddfc5ea47e09bb445a7447eb9694b8d1b75a97fff58e884398c6b22825a),
           ([1, 2, 3, 4, 5, 6], 0x2d1a03850084442813c8ebf094dea47538490a68b05f2239134a4cca2f6302e1),
           (list(range(16)), 0x1b733f2ff41971b23819a16bc8c16bbe13d98173358429fcc12f6f0826407a56)]
    for inp, exp in kat:
        assert h.hash_ints(inp)[0] == exp, inp


def test_linearhash_matrix_known_answer(orc):
    h = orc.bn128()
    cols = np.array([[e, e * 1000, e * 1000000] for e in range(100)], np.uint64).reshape(-1)
    assert h.from_mont(h.hash_element_matrix(cols)) == 0x29c2ac38b7b8d18b9c1b575369cb4ab930ef71ebd5e4631b3916360233a29cae


def test_linearhash_corner_case_known_answers(orc):
    """<= 4 words: the digest is the Montgomery form of the 256-bit integer (to_bn128_mont)"""
    h = orc.bn128()
    d = h.hash_element_array(np.array([6188675464075253840, 2608530331018891925], np.uint64))
    assert [int(v) for v in d] == [15714769047018385385, 14080511166848616671, 11411897157942048316, 1802287360671936077]
    d = h.hash_element_array(np.array([18440682777423237490, 1156220815552880681], np.uint64))
    assert [int(v) for v in d] == [12850950522295690944, 15045028186447136619, 11701297961637547631, 875058675367281598]


def _cols(n, n_pols):
    i, j = np.meshgrid(np.arange(n, dtype=np.uint64), np.arange(n_pols, dtype=np.uint64), indexing="ij")
    return (i + j * np.uint64(1000)).reshape(-1)


def test_merkle_root_known_answer_and_proofs(orc):
    h = orc.bn128()
    n, w = 256, 9
    rows = _cols(n, w)
    nodes = h.merkelize(rows, w, n)
    assert h.from_mont(nodes[-1]) == 2052732265221205192391066587135329070685482706470940527184785165917406935559
    for idx in (0, 3, 255):                                   # get_group_proof + verify_group_proof
        path = h.merkle_proof(nodes, n, idx)
        leaf = h.hash_element_matrix(rows[idx * w:(idx + 1) * w])    # the verifier's leaf hash (:130-138)
        assert np.array_equal(leaf, nodes[idx])                       # == the prover's hash_element_array
        assert np.array_equal(h.root_from_proof(path, leaf), nodes[-1])


def test_merkle_not_power_of_16(orc):
    h = orc.bn128()
    for n, w in ((33, 6), (17, 50), (1, 5), (16, 3)):
        rows = _cols(n, w)
        nodes = h.merkelize(rows, w, n)
        assert nodes.shape[0] == h.n_nodes(n)
        if n > 1:
            idx = n - 1
            path = h.merkle_proof(nodes, n, idx)
            assert np.array_equal(h.root_from_proof(path, nodes[idx]), nodes[-1])


def test_transcript_is_deterministic_and_sensitive(orc):
    h = orc.bn128()
    def run(vals):
        t = h.transcript()
        for v in vals:
            t.put1(v)
        t.put4(h.to_mont(123456789))
        return t.get_field(), list(t.get_permutations(8, 11)), t.get_field()
    a, b = run([1, 2, 3]), run([1, 2, 3])
    assert a == b and a != run([1, 2, 4])
    assert all(v < 0xFFFFFFFF00000001 for v in a[0]) and all(0 <= q < 2048 for q in a[1])
    # squeeze semantics: the three words of get_field are the low 192 bits of the first sponge output
    t = h.transcript(); t.put1(7)
    f = t.get_field()
    st = h.poseidon(np.concatenate([h.to_mont(7)] + [np.zeros(4, np.uint64)] * 15), np.zeros(4, np.uint64), 17)
    v = h.from_mont(st[0])
    assert f == [((v >> (64 * i)) & (2**64 - 1)) % 0xFFFFFFFF00000001 for i in range(3)]


# ---- BLS12-381 scalar field (oracle/bls12381_hash.c) ------------------------------------------------------
def test_bls12381_poseidon_known_answers(orc):
    """poseidon_bls12381_opt.rs:236-311"""
    h = orc.bls12381()
    kat = [([1], 0x164efff6c8a32ef98836c868f8c8dedcbe3068d16ba6098f282a6d185edb551f),
           ([1, 0], 0x59220c0fc5748e83c141c7bb8dae0a2bd5bbb227c778ede87296ba07960ec3d8),
           ([1, 0, 0], 0x73584296b068384db6028b55d995108518d4483ab177197274effe979b91526e),
           ([1, 2, 0, 0, 0], 0x385acd94e53a8c6f981809c2201582beceaec12250200f1e75ba93e6cf5ec736),
           ([1, 2, 0, 0, 0, 0], 0x023dd8aecc0967c0588754eebd39af39bdae2bbf4195fee1208613c909aaa29b),
           ([3, 4, 0, 0, 0], 0x19c96d726da9e3df4e5d0da19f324f7bf376dc7bf97efbf37082473f7fa24af8),
           ([3, 4, 0, 0, 0, 0], 0x0cb7b1761b9abe661847a10701c6eae7c631ff580c5b7f3ac2f8be1088d22bba),
           ([1, 2, 3, 4], 0x6f5f297b0ab0d1e7400501b9bdd4c3be2fe676b6a05deb845143b87355167a8d),
           (list(range(16)), 0x12d374bbdb8d3c1c0230b20b8fe1572f1e652a616d16e834718a982574106405)]
    for inp, exp in kat:
        assert h.hash1_ints(inp) == exp, inp


def test_bls12381_linearhash_known_answers(orc):
    """linearhash_bls12381.rs:141-193"""
    h = orc.bls12381()
    cols = np.array([[e, e * 1000, e * 1000000] for e in range(100)], np.uint64).reshape(-1)
    assert h.from_mont(h.hash_element_matrix(cols)) == 0x1aea10165e8c452045633835341291832bf7d46ace4bd6e8b1a2ddb9f257c2be
    cols = np.array([[e, e, e] for e in range(9)], np.uint64).reshape(-1)
    assert h.from_mont(h.hash_element_matrix(cols)) == 0x683f0b0c6f1a15d7715cbac061ca80f1f30a28920d32993c2f9cd307aee7bcbb
    d = h.hash_element_array(np.array([6188675464075253840, 2608530331018891925], np.uint64))
    assert [int(v) for v in d] == [664572115127318441, 16413352647427919515, 17253685441004911215, 6212100569330953807]
    d = h.hash_element_array(np.array([18440682777423237490, 1156220815552880681], np.uint64))
    assert int(d[0]) == 13796980492452026086


def test_bls12381_merkle_root_known_answer(orc):
    """merklehash_bls12381.rs:274-300"""
    h = orc.bls12381()
    n, w = 4, 3
    i, j = np.meshgrid(np.arange(n, dtype=np.uint64), np.arange(w, dtype=np.uint64), indexing="ij")
    rows = (i + j * np.uint64(10) + np.uint64(1)).reshape(-1)
    nodes = h.merkelize(rows, w, n)
    assert h.from_mont(nodes[-1]) == 32227206116237215740162377531481191838063909532381497804787245624658969614932
    path = h.merkle_proof(nodes, n, 1)
    assert np.array_equal(h.root_from_proof(path, h.hash_element_matrix(rows[3:6])), nodes[-1])
