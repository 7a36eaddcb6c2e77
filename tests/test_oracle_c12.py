"""compressor12 exec oracle (oracle/compressor12.py): the reference's own test for this path, the .exec write -> read
round trip (compressor12_exec.rs:117-151), and the arithmetic on a hand-computed case."""
import pathlib, sys
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import compressor12 as C12  # noqa: E402


def test_write_and_read_exec_file():
    """compressor12_exec.rs:117-151 with its own s_map"""
    s_map = [[1, 2, 4], [2, 3, 42], [1, 1, 3], [4, 5, 2], [3, 4, 5], [1, 2, 4], [2, 3, 42], [1, 1, 3], [4, 5, 2], [3, 4, 5], [3, 4, 5], [3, 4, 5]]
    adds_len, col_len, adds, sm = C12.read_exec(C12.write_exec([], s_map))
    assert (adds_len, col_len, adds) == (0, 3, [])
    assert sm[:12] == [row[0] for row in s_map] and sm[12 * 2 + 1] == 42          # row i, column c at 12 i + c


def test_raw_coefficients_and_chained_adds():
    P = C12.P
    text = C12.write_exec([(1, 2, 3, 5), (3, 1, P - 1, 2)], [[3, 4, 0]] + [[0, 1, 2]] * 11)
    _, _, adds, _ = C12.read_exec(text)
    assert adds[2] == 3 * (1 << 64) % P                                            # the raw Montgomery word of 3
    cm = C12.exec_cm(text, [1, 10, 20], 4)
    w3 = (10 * 3 + 20 * 5) % P; w4 = (w3 * (P - 1) + 10 * 2) % P                  # the second add reads the first
    assert cm[:, 0].tolist() == [w3, w4, 0, 0] and cm[:, 1].tolist() == [0, 10, 20, 0]
    try:
        C12.exec_cm(text, [1, P, 20], 4); assert False
    except ValueError:
        pass
