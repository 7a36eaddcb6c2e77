"""Parity of the BN128-field hashing (csrc/poseidon_bn128.hip through the zk_bn128_* C ABI) against the CPU
oracle (oracle/bn128_hash.c, itself pinned by the reference's known answers in tests/test_oracle_bn128.py),
plus the reference's known answers straight through the GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module", autouse=True)
def _gpu(zk):
    assert zk.lib().zk_device_count() >= 1, "no GPU visible (the product has no CPU fallback)"
    zk.init(0)
    zk.bn128_init()


def test_poseidon_known_answers_on_gpu(zk, orc):
    h = orc.bn128()                                                    # only for the Montgomery conversions
    kat = [([1], 0x29176100eaa962bdc1fe6c654d6a3c130e96a4d1168b33848b897dc502820133),
           ([1, 2], 0x115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a),
           ([1, 2, 0, 0, 0], 0x024058dd1e168f34bac462b6fffe58fd69982807e9884c1c6148182319cee427),
           ([3, 4, 0, 0, 0, 0], 0x1b1caddfc5ea47e09bb445a7447eb9694b8d1b75a97fff58e884398c6b22825a),
           ([1, 2, 3, 4, 5, 6], 0x2d1a03850084442813c8ebf094dea47538490a68b05f2239134a4cca2f6302e1),
           (list(range(16)), 0x1b733f2ff41971b23819a16bc8c16bbe13d98173358429fcc12f6f0826407a56)]
    for inp, exp in kat:
        raw = np.concatenate([h.to_mont(v) for v in inp])
        assert h.from_mont(zk.bn128_poseidon(raw)[0]) == exp, inp


@pytest.mark.parametrize("n_in", list(range(1, 17)))
def test_poseidon_every_width_matches_oracle(zk, orc, n_in):
    h = orc.bn128()
    rng = np.random.default_rng(n_in)
    vals = [int.from_bytes(rng.bytes(32), "little") % h.R for _ in range(n_in + 1)]
    raw = np.concatenate([h.to_mont(v) for v in vals[1:]])
    init = h.to_mont(vals[0])
    got = zk.bn128_poseidon(raw, init, n_in + 1)
    exp = h.poseidon(raw, init, n_in + 1)
    assert np.array_equal(got, exp)


def test_poseidon_rejects_bad_lengths(zk):
    with pytest.raises(zk.ZkError):
        zk.bn128_poseidon(np.zeros(17 * 4, np.uint64))
    with pytest.raises(zk.ZkError):
        zk.bn128_poseidon(np.zeros(0, np.uint64))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 7, 47, 48, 49, 50, 96, 97, 150])
def test_linearhash_matches_oracle(zk, orc, n):
    h = orc.bn128()
    rng = np.random.default_rng(1000 + n)
    v = rng.integers(0, P, size=n, dtype=np.uint64)
    if n >= 4:
        v[:4] = [P - 1, P - 1, P - 1, P - 1]                            # 4 maximal words exceed r: the reduced branch
    assert np.array_equal(zk.bn128_linearhash(v), h.hash_element_array(v))


def test_linearhash_corner_case_known_answers(zk):
    d = zk.bn128_linearhash(np.array([6188675464075253840, 2608530331018891925], np.uint64))
    assert [int(x) for x in d] == [15714769047018385385, 14080511166848616671, 11411897157942048316, 1802287360671936077]
    d = zk.bn128_linearhash(np.array([18440682777423237490, 1156220815552880681], np.uint64))
    assert [int(x) for x in d] == [12850950522295690944, 15045028186447136619, 11701297961637547631, 875058675367281598]


def _cols(n, n_pols):
    i, j = np.meshgrid(np.arange(n, dtype=np.uint64), np.arange(n_pols, dtype=np.uint64), indexing="ij")
    return (i + j * np.uint64(1000)).reshape(-1)


def test_merkle_root_known_answer(zk, orc):
    h = orc.bn128()
    t = zk.MerkleTreeBN128(); t.merkelize(_cols(256, 9), 9, 256)
    assert h.from_mont(t.root()) == 2052732265221205192391066587135329070685482706470940527184785165917406935559


@pytest.mark.parametrize("height,width", [(1, 5), (16, 3), (17, 50), (33, 6), (256, 9), (257, 12), (4096, 20), (5000, 1), (1000, 0),
                                          (128, 3072), (3, 49), (4097, 49), (4096, 145), (2, 7),    # few wide rows: one wave per row
                                          (4097, 27), (4100, 37), (4099, 48),                       # one sponge step of 9 / 13 / 16 blocks, state in registers
                                          (4101, 96), (4098, 130), (4099, 52), (4100, 147),         # tall and wide: full steps in one kernel, the last (16 / 12 / 2 / 1 blocks) in another
                                          (45000, 3), (262144, 2), (262160, 2), (524304, 2)])       # levels of 2813 and 16384 parents (eight lanes each), 16385 and 32770 (one lane each, registers, ragged)
def test_merkle_tree_matches_oracle(zk, orc, height, width):
    h = orc.bn128()
    rng = np.random.default_rng(height * 100 + width)
    rows = rng.integers(0, P, size=height * width, dtype=np.uint64)
    t = zk.MerkleTreeBN128(); t.merkelize(rows, width, height)
    exp = h.merkelize(rows, width, height)
    assert np.array_equal(t.nodes(), exp)
    for idx in {0, height // 2, height - 1}:
        row, path = t.get_group_proof(idx)
        assert np.array_equal(row, rows[idx * width:(idx + 1) * width])
        if height > 1:
            assert np.array_equal(path, h.merkle_proof(exp, height, idx))
            assert np.array_equal(h.root_from_proof(path, exp[idx]), t.root())
    with pytest.raises(zk.ZkError):
        t.get_group_proof(height)                                       # MerkleTreeError: access invalid node
    qs = [int(v) for v in rng.integers(0, height, size=9)] + [height - 1, 0, 0]
    for (row, path), idx in zip(t.get_group_proofs(qs), qs):           # the openings of a query list in one round trip
        r1, p1 = t.get_group_proof(idx)
        assert np.array_equal(row, r1) and np.array_equal(path, p1)
    with pytest.raises(zk.ZkError):
        t.get_group_proofs([0, height])


def test_transcript_matches_oracle_sequence(zk, orc):
    h = orc.bn128()
    rng = np.random.default_rng(5)
    t, o = zk.TranscriptBN128(), h.transcript()
    for step in range(40):
        if step % 3 == 2:
            d = h.to_mont(int.from_bytes(rng.bytes(32), "little"))
            t.put([d]); o.put4(d)
        else:
            v = int(rng.integers(0, P, dtype=np.uint64))
            t.put([np.array([v], np.uint64)]); o.put1(v)
        if step % 5 == 0:
            assert t.get_field() == o.get_field()
        if step % 7 == 0:
            assert t.get_fields1() == o.get_fields1()
    assert list(t.get_permutations(8, 21)) == list(o.get_permutations(8, 21))
    assert list(t.get_permutations(64, 23)) == list(o.get_permutations(64, 23))     # spans several 253-bit fields
    assert t.get_field() == o.get_field()
    with pytest.raises(zk.ZkError):
        t.put([np.zeros(3, np.uint64)])                                 # "Invalid elements as inputs to transcript"


# ---- BLS12-381 scalar field: the twin entry points zk_bls12381_* ----------------------------------------
@pytest.fixture(scope="module")
def bls(zk):
    zk.bn128_init(field="bls12381")
    return "bls12381"


def test_bls12381_poseidon_known_answers_on_gpu(zk, orc, bls):
    """poseidon_bls12381_opt.rs:236-311: Poseidon::hash = state[1]"""
    h = orc.bls12381()
    kat = [([1], 0x164efff6c8a32ef98836c868f8c8dedcbe3068d16ba6098f282a6d185edb551f),
           ([1, 2, 0, 0, 0], 0x385acd94e53a8c6f981809c2201582beceaec12250200f1e75ba93e6cf5ec736),
           ([1, 2, 3, 4], 0x6f5f297b0ab0d1e7400501b9bdd4c3be2fe676b6a05deb845143b87355167a8d),
           (list(range(16)), 0x12d374bbdb8d3c1c0230b20b8fe1572f1e652a616d16e834718a982574106405)]
    for inp, exp in kat:
        raw = np.concatenate([h.to_mont(v) for v in inp])
        assert h.from_mont(zk.bn128_poseidon(raw, None, 2, field=bls)[1]) == exp, inp


@pytest.mark.parametrize("n_in", [1, 2, 5, 8, 15, 16])
def test_bls12381_poseidon_matches_oracle(zk, orc, bls, n_in):
    h = orc.bls12381()
    rng = np.random.default_rng(500 + n_in)
    vals = [int.from_bytes(rng.bytes(32), "little") % h.R for _ in range(n_in + 1)]
    raw = np.concatenate([h.to_mont(v) for v in vals[1:]])
    init = h.to_mont(vals[0])
    assert np.array_equal(zk.bn128_poseidon(raw, init, n_in + 1, field=bls), h.poseidon(raw, init, n_in + 1))


def test_bls12381_linearhash_and_corner_cases(zk, orc, bls):
    h = orc.bls12381()
    rng = np.random.default_rng(77)
    for n in (0, 1, 3, 4, 5, 47, 48, 49, 97):
        v = rng.integers(0, P, size=n, dtype=np.uint64)
        if n >= 4:
            v[:4] = [P - 1, P - 1, P - 1, P - 1]
        assert np.array_equal(zk.bn128_linearhash(v, field=bls), h.hash_element_array(v)), n
    d = zk.bn128_linearhash(np.array([6188675464075253840, 2608530331018891925], np.uint64), field=bls)   # linearhash_bls12381.rs:171-181
    assert [int(x) for x in d] == [664572115127318441, 16413352647427919515, 17253685441004911215, 6212100569330953807]


def test_bls12381_merkle_known_answer_and_oracle(zk, orc, bls):
    h = orc.bls12381()
    i, j = np.meshgrid(np.arange(4, dtype=np.uint64), np.arange(3, dtype=np.uint64), indexing="ij")
    t = zk.MerkleTreeBN128(field=bls); t.merkelize((i + j * np.uint64(10) + np.uint64(1)).reshape(-1), 3, 4)
    assert h.from_mont(t.root()) == 32227206116237215740162377531481191838063909532381497804787245624658969614932   # merklehash_bls12381.rs:274-300
    for height, width in ((33, 6), (257, 12), (4096, 20), (5000, 1), (4100, 37), (4099, 49), (4098, 130), (45000, 3), (524304, 2)):   # from (4100, 37) on: as in test_merkle_tree_matches_oracle
        rng = np.random.default_rng(height + width)
        rows = rng.integers(0, P, size=height * width, dtype=np.uint64)
        t = zk.MerkleTreeBN128(field=bls); t.merkelize(rows, width, height)
        exp = h.merkelize(rows, width, height)
        assert np.array_equal(t.nodes(), exp)
        row, path = t.get_group_proof(height - 1)
        assert np.array_equal(path, h.merkle_proof(exp, height, height - 1))


@pytest.mark.parametrize("field", ["bn128", "bls12381"])
def test_register_kernels_of_every_block_count_match_oracle(zk, orc, bls, field):
    """One sponge step with the state in registers, t = 3 .. 17 (rows of 5 .. 48 columns on more than 4096 rows): since round 6 the dense
    layers of these kernels run on the matrix pipe (csrc/fr_mfma.hip.h), each t with tables of its own.  Ragged heights: the idle lanes of
    the last wave shadow the last row.  Extreme words included (p - 1 everywhere in one row, zeros in another)."""
    h = orc.bn128() if field == "bn128" else orc.bls12381()
    for nb in range(2, 17):
        for width in {3 * nb - 2, 3 * nb}:
            if width <= 4:
                continue
            height = 4097 + 3 * nb
            rng = np.random.default_rng(1000 * nb + width)
            rows = rng.integers(0, P, size=height * width, dtype=np.uint64)
            rows[:width] = P - 1
            rows[width:2 * width] = 0
            t = zk.MerkleTreeBN128(field=field)
            t.merkelize(rows, width, height)
            exp = h.merkelize(rows, width, height)
            assert np.array_equal(t.nodes(), exp), (field, nb, width)


def test_bls12381_transcript_matches_oracle(zk, orc, bls):
    h = orc.bls12381()
    rng = np.random.default_rng(6)
    t, o = zk.TranscriptBN128(field=bls), h.transcript()
    for step in range(30):
        if step % 3 == 2:
            d = h.to_mont(int.from_bytes(rng.bytes(32), "little"))
            t.put([d]); o.put4(d)
        else:
            v = int(rng.integers(0, P, dtype=np.uint64))
            t.put([np.array([v], np.uint64)]); o.put1(v)
        if step % 5 == 0:
            assert t.get_field() == o.get_field()
    assert list(t.get_permutations(64, 23)) == list(o.get_permutations(64, 23))
