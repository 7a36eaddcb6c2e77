"""Parity of compressor12 exec on the device (csrc/compressor12.hip, through the C ABI) against oracle/compressor12.py:
bit exact.  SURVEY.md 8(f)-4."""
import importlib, pathlib, random, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import compressor12 as C12  # noqa: E402
P = C12.P


@pytest.fixture(scope="module")
def dev(zk):
    assert zk.lib().zk_device_count() >= 1, "no GPU visible (the product has no CPU fallback)"
    zk.init(0)
    return importlib.import_module("eigen_zkvm_amd.compressor12")


def make_case(rng, n_wit, n_adds, n_map, chain):
    """adds reading random earlier wires; `chain`: every add also reads its predecessor (depth = n_adds)"""
    adds = []
    for i in range(n_adds):
        hi = n_wit + i
        a = hi - 1 if chain and i else rng.randrange(hi)
        adds.append((a, rng.randrange(hi), rng.choice([1, P - 1, rng.randrange(P)]), rng.randrange(P)))
    s_map = [[rng.choice([0, rng.randrange(n_wit + n_adds)]) for _ in range(n_map)] for _ in range(12)]
    wit = [1] + [rng.choice([0, P - 1, rng.randrange(P)]) for _ in range(n_wit - 1)]
    return C12.write_exec(adds, s_map), wit


@pytest.mark.parametrize("n_wit,n_adds,n_map,n_rows,chain", [(5, 0, 3, 4, False), (40, 25, 16, 16, False), (300, 500, 1000, 1024, False),
                                                             (50, 200, 64, 64, True), (3000, 20000, 4000, 4096, False)])
def test_exec_matches_oracle(zk, dev, n_wit, n_adds, n_map, n_rows, chain):
    rng = random.Random(n_wit * 7 + n_adds)
    text, wit = make_case(rng, n_wit, n_adds, n_map, chain)
    E = dev.Compressor12Exec(text, n_wit)
    assert E.depth == (n_adds if chain else E.depth) and (n_adds == 0) == (E.depth == 0)
    cm = E.run(np.array(wit, dtype=np.uint64), n_rows).to_host().reshape(n_rows, 12)
    assert np.array_equal(cm, C12.exec_cm(text, wit, n_rows))
    E.free()


def test_exec_errors(zk, dev):
    rng = random.Random(1)
    text, wit = make_case(rng, 10, 5, 4, False)
    with pytest.raises(zk.ZkError, match="length does not match"):
        dev.Compressor12Exec(text[:-1] + ",7]", 10)
    with pytest.raises(zk.ZkError, match="array of integers"):
        dev.Compressor12Exec("{}", 10)
    with pytest.raises(zk.ZkError, match="does not exist yet"):
        dev.Compressor12Exec(C12.write_exec([(12, 0, 1, 1)], [[0]] * 12), 10)
    E = dev.Compressor12Exec(text, 10)
    with pytest.raises(zk.ZkError, match="witness has"):
        E.run(np.array(wit[:-1], dtype=np.uint64), 4)
    with pytest.raises(zk.ZkError, match="more rows than the trace"):
        E.run(np.array(wit, dtype=np.uint64), 2)
    bad = list(wit); bad[3] = P
    with pytest.raises(zk.ZkError, match="not a Goldilocks field element"):
        E.run(np.array(bad, dtype=np.uint64), 4)
    E.free()
