"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of compressor12 exec
(recursion/src/compressor12/compressor12_exec.rs:17-125) and of the .exec file writer
(recursion/src/compressor12/compressor12_setup.rs:51-83).  The reference holds one test for this path, the
write -> read round trip of the .exec format (compressor12_exec.rs:117-151), restated in tests/test_oracle_c12.py;
the arithmetic has no known-answer vector (PARITY UNPINNED beyond the format): it is three lines of field arithmetic."""
import json
import numpy as np

P = 0xFFFFFFFF00000001
R = (1 << 64) % P            # FGL keeps a * 2^64 mod p (field_gl.rs:328-347); `.into()` hands out that raw word (:503-507)
RINV = pow(R, P - 2, P)


def write_exec(adds, s_map):
    """adds: list of (a, b, coeff_a, coeff_b) with the coefficients as field VALUES; s_map: 12 lists of equal length.
    -> the JSON text of compressor12_setup.rs:51-83 (coefficients stored as raw Montgomery words)"""
    assert len(s_map) == 12, "s_map should have 12 rows"
    n = len(s_map[0])
    buf = [len(adds), n]
    for a, b, ca, cb in adds:
        buf += [a, b, ca % P * R % P, cb % P * R % P]
    for i in range(n):
        for c in range(12):
            buf.append(s_map[c][i])
    return json.dumps(buf, separators=(",", ":"))


def read_exec(text):
    """compressor12_exec.rs:110-125 -> (adds_len, s_map_column_len, adds, s_map)"""
    buff = json.loads(text)
    adds_len, col_len = buff[0], buff[1]
    rest = buff[2:]
    assert len(rest) == adds_len * 4 + col_len * 12
    return adds_len, col_len, rest[:adds_len * 4], rest[adds_len * 4:]


def exec_cm(text, witness, n_rows):
    """compressor12_exec.rs:45-103 after the witness calculator: -> [n_rows][12] u64, the content of the .cm file"""
    adds_len, col_len, adds, s_map = read_exec(text)
    w = []
    for x in witness:
        if int(x) >= P: raise ValueError("witness value is not a field element")      # FGL::from(u64) unwraps from_repr
        w.append(int(x))
    for i in range(adds_len):
        c2, c3 = adds[4 * i + 2], adds[4 * i + 3]
        if c2 >= P or c3 >= P: raise ValueError("coefficient is not a field element")  # from_raw_repr
        w.append((w[adds[4 * i]] * (c2 * RINV % P) + w[adds[4 * i + 1]] * (c3 * RINV % P)) % P)
    cm = np.zeros((n_rows, 12), np.uint64)
    for i in range(col_len):
        for c in range(12):
            s = s_map[12 * i + c]
            cm[i, c] = w[s] if s != 0 else 0
    return cm
