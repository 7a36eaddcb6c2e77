"""The Rust shim's trait implementations (bindings/rust/starky-hip/src/{merkle,transcript}.rs) against the reference's trait
declarations (starky/src/traits.rs:24-63).

The crate cannot be compiled here (no Rust toolchain), so this is the mechanical stand-in for the first thing `rustc` would
check: every method the trait declares is implemented, with the same number of arguments, the same argument and return types
once the impl's associated types are substituted for `Self::X`, and the same receiver; and the impl declares every
associated type the trait asks for.  Reads the trait file from /root/reference (absent on the GPU box: skipped there)."""
import pathlib, re
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
TRAITS = pathlib.Path("/root/reference/starky/src/traits.rs")
SHIM = ROOT / "bindings" / "rust" / "starky-hip" / "src"

pytestmark = pytest.mark.skipif(not TRAITS.exists(), reason="reference tree not present")


def _strip(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    return re.sub(r"//[^\n]*", " ", t)


def _block(text, header_re):
    """the brace-delimited block that follows the first match of header_re"""
    m = re.search(header_re, text)
    assert m, header_re
    i = text.index("{", m.end())
    depth, j = 0, i
    while True:
        depth += {"{": 1, "}": -1}.get(text[j], 0)
        if depth == 0:
            return text[i + 1:j]
        j += 1


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "<([": depth += 1
        if ch in ">)]": depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def _norm(t):
    return re.sub(r"\s+", "", t or "()")


def _fns(block):
    """name -> (generics, receiver, [arg types], return type); nested bodies are skipped by brace depth"""
    out, depth, top = {}, 0, ""
    for ch in block:                                        # keep only depth-0 text (signatures), drop method bodies
        if ch == "{":
            depth += 1
            if depth == 1: top += "{"
            continue
        if ch == "}":
            depth -= 1
            if depth == 0: top += "}"
            continue
        if depth == 0: top += ch
    for m in re.finditer(r"\bfn\s+(\w+)\s*(<[^>]*>)?\s*\((.*?)\)\s*(?:->\s*(.*?))?\s*(?:where[^;{]*)?[;{]", top, flags=re.S):
        name, gen, args, ret = m.group(1), m.group(2) or "", _split_args(m.group(3)), m.group(4)
        recv = ""
        if args and re.fullmatch(r"&?\s*(mut\s+)?self", args[0]):
            recv, args = _norm(args[0]), args[1:]
        types = [_norm(a.split(":", 1)[1]) for a in args]
        out[name] = (_norm(gen) if gen else "", recv, types, _norm(ret))
    return out


def _assoc(block):
    return {m.group(1): _norm(m.group(2)) for m in re.finditer(r"\btype\s+(\w+)\s*=\s*([^;]+);", block)}


def _subst(t, assoc):
    return re.sub(r"Self::(\w+)", lambda m: assoc.get(m.group(1), m.group(0)), t)


@pytest.mark.parametrize("trait,file,impl", [("MerkleTree", "merkle.rs", "MerkleTreeHipGL"), ("Transcript", "transcript.rs", "TranscriptHipGL")])
def test_impl_matches_the_trait(trait, file, impl):
    tsrc = _strip(TRAITS.read_text())
    tblock = _block(tsrc, r"pub\s+trait\s+%s\b" % trait)
    want = _fns(tblock)
    want_types = set(re.findall(r"\btype\s+(\w+)\s*[:;]", tblock))
    isrc = _strip((SHIM / file).read_text())
    iblock = _block(isrc, r"impl\s+%s\s+for\s+%s\b" % (trait, impl))
    got, assoc = _fns(iblock), _assoc(iblock)
    assert want, "no methods parsed from the trait"
    assert want_types <= set(assoc), "associated types missing in the impl: %s" % sorted(want_types - set(assoc))
    assert set(got) == set(want), "methods: missing %s, extra %s" % (sorted(set(want) - set(got)), sorted(set(got) - set(want)))
    for name, (gen, recv, types, ret) in want.items():
        g_gen, g_recv, g_types, g_ret = got[name]
        assert g_recv == recv, (name, "receiver", g_recv, recv)
        assert len(g_types) == len(types), (name, "arity", g_types, types)
        norm = lambda t: _subst(t, assoc)
        assert [norm(t) for t in g_types] == [norm(t) for t in types], (name, "argument types", g_types, types)
        assert norm(g_ret) == norm(ret), (name, "return type", g_ret, ret)
        assert re.sub(r"\s", "", g_gen) == re.sub(r"\s", "", gen), (name, "generics", g_gen, gen)


def test_the_checker_sees_a_drift():
    """the parser is not vacuous: a changed argument type and a missing method are both reported"""
    trait = "pub trait T { type N: Copy; fn a(&self, x: usize) -> Self::N; fn b(&mut self, v: &[Vec<u64>]) -> Result<()>; }"
    good = "impl T for X { type N = u64; fn a(&self, x: usize) -> u64 { 0 } fn b(&mut self, v: &[Vec<u64>]) -> Result<()> { Ok(()) } }"
    bad = "impl T for X { type N = u64; fn a(&self, x: u32) -> u64 { 0 } }"
    w = _fns(_block(trait, r"pub\s+trait\s+T\b"))
    g, a = _fns(_block(good, r"impl\s+T\s+for\s+X\b")), _assoc(_block(good, r"impl\s+T\s+for\s+X\b"))
    assert set(w) == set(g) == {"a", "b"} and _subst(w["a"][3], a) == g["a"][3] and w["b"][2] == g["b"][2]
    b = _fns(_block(bad, r"impl\s+T\s+for\s+X\b"))
    assert set(b) != set(w) and b["a"][2] != w["a"][2]


def test_fft_seam_functions_match_fft_p():
    """`fft`, `ifft`, `interpolate` of the shim (src/fft.rs) carry the signatures of starky/src/fft_p.rs:242-261 -- they are
    swapped in by path, so generics, argument order and types must be the reference's"""
    ref = pathlib.Path("/root/reference/starky/src/fft_p.rs")
    want, got = _fns(_strip(ref.read_text())), _fns(_strip((SHIM / "fft.rs").read_text()))
    for name in ("fft", "ifft", "interpolate"):
        assert name in want and name in got, name
        assert got[name] == want[name], (name, got[name], want[name])


def test_stark_prove_entry_matches_prove_rs():
    """the shim's public `stark_prove` (src/prove.rs) takes what starky/src/prove.rs:30-41 takes: it is what zkit calls"""
    ref = pathlib.Path("/root/reference/starky/src/prove.rs")
    want, got = _fns(_strip(ref.read_text())), _fns(_strip((SHIM / "prove.rs").read_text()))
    assert got["stark_prove"] == want["stark_prove"], (got["stark_prove"], want["stark_prove"])


def _uses(text):
    """(crate, [path segments], item) for every `use starky::…` / `use fields::…` of a source file"""
    out = []
    for m in re.finditer(r"\buse\s+((?:starky|fields)(?:::\w+)*)::(\{[^}]*\}|\w+)(?:\s+as\s+\w+)?\s*;", text):
        segs = m.group(1).split("::")
        items = [i.strip().split(" as ")[0].strip() for i in m.group(2).strip("{}").split(",")] if m.group(2).startswith("{") else [m.group(2)]
        for it in items:
            if it:
                out.append((segs[0], segs[1:], it))
    return out


def test_every_imported_name_exists_in_the_reference_crates():
    """name resolution by hand: each `use starky::a::B` / `use fields::a::B` of the shim names a public module of that crate
    (declared `pub mod a;` in its lib.rs, file present) and a public item of that module (or a `pub use` re-export)"""
    crates = {"starky": pathlib.Path("/root/reference/starky/src"), "fields": pathlib.Path("/root/reference/fields/src")}
    seen = 0
    for f in sorted(SHIM.glob("*.rs")):
        for crate, mods, item in _uses(_strip(f.read_text())):
            src = crates[crate]
            lib = _strip((src / "lib.rs").read_text())
            if not mods:                                            # `use starky::X`: a module or a re-export of lib.rs
                assert re.search(r"pub\s+mod\s+%s\s*;" % item, lib) or re.search(r"pub\s+use\s+[^;]*\b%s\b" % item, lib), (f.name, crate, item)
                seen += 1
                continue
            assert len(mods) == 1, (f.name, mods)
            assert re.search(r"pub\s+mod\s+%s\s*;" % mods[0], lib), (f.name, "module not public", mods[0])
            mfile = src / (mods[0] + ".rs")
            assert mfile.exists(), mfile
            mtext = _strip(mfile.read_text())
            decl = r"pub\s+(?:struct|enum|trait|fn|type|const|static|mod)\s+%s\b" % item
            assert re.search(decl, mtext) or re.search(r"pub\s+use\s+[^;]*\b%s\b" % item, mtext), (f.name, crate, mods[0], item)
            seen += 1
    assert seen >= 15


def _call_args(text, callee_re):
    """the top-level argument list of the first call matching callee_re (a regex ending just before the opening parenthesis)"""
    m = re.search(callee_re + r"\s*\(", text)
    assert m, callee_re
    i, depth, j = m.end(), 1, m.end()
    while depth:
        depth += {"(": 1, ")": -1}.get(text[j], 0)
        j += 1
    return [a for a in _split_args(text[i:j - 1]) if a]


def test_calls_into_the_reference_have_the_callees_arity():
    """`prove` (src/prove.rs) calls StarkSetup::new, stark_verify and pil2circom: argument counts against the definitions
    (stark_setup.rs:27-32, stark_verify.rs:20-26, pil2circom.rs:21-28), and StarkOption is built with exactly its fields"""
    ref = pathlib.Path("/root/reference/starky/src")
    shim = _strip((SHIM / "prove.rs").read_text())
    setup_new = _fns(_block(_strip((ref / "stark_setup.rs").read_text()), r"impl<[^>]*>\s*StarkSetup<[^>]*>"))["new"]
    assert len(_call_args(shim, r"StarkSetup::<\w+>::new")) == len(setup_new[2])
    verify = _fns(_strip((ref / "stark_verify.rs").read_text()))["stark_verify"]
    assert len(_call_args(shim, r"stark_verify::stark_verify::<[^(]*>")) == len(verify[2])
    p2c_src = _strip((ref / "pil2circom.rs").read_text())
    p2c = _fns(p2c_src)["pil2circom"]
    assert len(_call_args(shim, r"pil2circom::pil2circom::<\w+>")) == len(p2c[2])
    fields = set(re.findall(r"pub\s+(\w+)\s*:", _block(p2c_src, r"pub\s+struct\s+StarkOption\b")))
    lit = re.search(r"pil2circom::StarkOption\s*\{([^}]*)\}", shim).group(1)
    used = {f.split(":")[0].strip() for f in lit.split(",") if f.strip()}
    assert used == fields, (used, fields)
