"""The Rust shim's `extern "C"` block (bindings/rust/starky-hip/src/hip_ffi.rs) against include/zkgpu.h.

No Rust toolchain exists in the authoring image, so nothing ever links the crate against the library; this test is the
mechanical guard against prototype drift: every function the crate declares must exist in the header with the same
arity, the same integer widths / signedness, the same pointer constness and the same return type, and must be exported
by the built library.  (Seams: starky/src/traits.rs:24-63, fft_p.rs:242-261, prove.rs:30-160.)"""
import ctypes, pathlib, re
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
HDR = ROOT / "include" / "zkgpu.h"
RS = ROOT / "bindings" / "rust" / "starky-hip" / "src" / "hip_ffi.rs"

C_BASE = {"int": "i32", "unsigned": "u32", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int64_t": "i64",
          "size_t": "usize", "char": "char", "void": "void", "uint8_t": "u8"}
RS_BASE = {"c_int": "i32", "u32": "u32", "u64": "u64", "i32": "i32", "i64": "i64", "usize": "usize", "c_char": "char",
           "c_void": "void", "u8": "u8"}


def _strip_comments(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    return re.sub(r"//[^\n]*", " ", t)


def c_type(t):
    """'const uint64_t*' -> ('ptr', const, 'u64'); 'uint32_t' -> ('val', 'u32'); opaque handles keep their name"""
    t = t.strip()
    arr = re.search(r"\[\s*\d*\s*\]$", t)            # `uint64_t out[4]` decays to a pointer
    if arr: t = t[:arr.start()].strip() + "*"
    if t.endswith("*"):
        inner = t[:-1].strip()
        if "*" in inner:                             # a pointer to a pointer: `T* const*` (the pointee pointer is const) or `T**`
            pointee_const = inner.endswith("const")
            return ("ptr", pointee_const, c_type(inner[:-5] if pointee_const else inner))
        const = bool(re.search(r"\bconst\b", inner))
        inner = re.sub(r"\bconst\b", "", inner).strip()
        inner = re.sub(r"^struct\s+", "", inner)
        return ("ptr", const, C_BASE.get(inner, inner))
    t = re.sub(r"\bconst\b", "", t).strip()
    return ("val", C_BASE.get(t, t))


def parse_header():
    text = _strip_comments(HDR.read_text())
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(zk_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret: continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)(\b[A-Za-z_]\w*)?(\s*\[\s*\d*\s*\])?$", a)
                ty = (mm.group(1) or "").strip()
                if not ty or ty in ("const", "unsigned"):               # unnamed parameter: the whole text is the type
                    ty = a; 
                elif mm.group(3): ty = ty + "[]"
                params.append(c_type(ty))
        out[name] = (c_type(ret), params)
    return out


def rs_type(t):
    t = t.strip()
    m = re.match(r"^\*(const|mut)\s+(.+)$", t)
    if m:
        inner = m.group(2).strip()
        if inner.startswith("*"):                    # `*const *const T`
            return ("ptr", m.group(1) == "const", rs_type(inner))
        return ("ptr", m.group(1) == "const", RS_BASE.get(inner, inner))
    return ("val", RS_BASE.get(t, t))


def parse_rust():
    text = _strip_comments(RS.read_text())
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub\s+fn\s+(zk_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), " ".join(m.group(2).split()), (m.group(3) or "").strip()
        params = []
        for a in [x for x in args.split(",") if x.strip()]:
            params.append(rs_type(a.split(":", 1)[1]))
        out[name] = (rs_type(ret) if ret else ("val", "void"), params)
    return out


def test_parsers_see_the_whole_surface():
    h, r = parse_header(), parse_rust()
    assert len(h) >= 100, f"header parser found only {len(h)} prototypes"
    assert len(r) >= 30, f"rust parser found only {len(r)} declarations"
    assert h["zk_gl_ntt"] == (("val", "i32"), [("ptr", True, "u64"), ("ptr", False, "u64"), ("val", "u32"), ("val", "u32"), ("val", "i32")])


def test_every_rust_declaration_matches_the_header():
    h, r = parse_header(), parse_rust()
    problems = []
    for name, (ret, params) in r.items():
        if name not in h:
            problems.append(f"{name}: declared in hip_ffi.rs, absent from zkgpu.h"); continue
        hret, hparams = h[name]
        if ret != hret: problems.append(f"{name}: return {ret} (rust) vs {hret} (header)")
        if len(params) != len(hparams):
            problems.append(f"{name}: {len(params)} parameters (rust) vs {len(hparams)} (header)"); continue
        for i, (a, b) in enumerate(zip(params, hparams)):
            if a != b: problems.append(f"{name}: parameter {i} {a} (rust) vs {b} (header)")
    assert not problems, "\n".join(problems)


def test_every_rust_declaration_is_exported_by_the_library():
    lib = ROOT / "eigen-zkvm_amd" / "libzkgpu.so"
    if not lib.exists(): pytest.skip("libzkgpu.so not built")
    dll = ctypes.CDLL(str(lib))
    missing = [n for n in parse_rust() if not hasattr(dll, n)]
    assert not missing, missing


def test_the_guard_sees_a_drifted_prototype(tmp_path, monkeypatch):
    """negative control: a widened integer and a dropped parameter are both reported"""
    bad = RS.read_text().replace("pub fn zk_gl_lde(src: *const u64, n_pols: u32,", "pub fn zk_gl_lde(src: *const u64, n_pols: u64,")
    bad = bad.replace("pub fn zk_merkle_depth(t: *const zk_merkle_t) -> u32;", "pub fn zk_merkle_depth() -> u32;")
    f = tmp_path / "hip_ffi.rs"; f.write_text(bad)
    import sys
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "RS", f)
    with pytest.raises(AssertionError) as e:
        test_every_rust_declaration_matches_the_header()
    assert "zk_gl_lde: parameter 1" in str(e.value) and "zk_merkle_depth: 0 parameters" in str(e.value)
