"""A C translation unit that includes include/zkgpu.h and calls the library (tests/cabi/zkgpu_cabi_test.c): catches drift
between the header and the implementation that the ctypes tables of eigen-zkvm_amd/__init__.py would hide."""
import json
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
D = ROOT / "tests" / "golden" / "starky_data"


def _build(tmp_path):
    exe = tmp_path / "zkgpu_cabi_test"
    lib_dir = ROOT / "eigen-zkvm_amd"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-Wno-pedantic", "-I", str(ROOT / "include"),
                           str(ROOT / "tests" / "cabi" / "zkgpu_cabi_test.c"), "-o", str(exe),
                           "-L", str(lib_dir), "-l:libzkgpu.so", "-Wl,-rpath," + str(lib_dir)])
    return exe


def test_c_client_compiles_links_and_loads(zk, tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([str(exe), "link"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "linked 36 entry points; p = 18446744069414584321" in out.stdout


def test_header_is_self_contained_c(tmp_path):
    """the header alone, as C99 and as C++17, with every warning on"""
    (tmp_path / "only.c").write_text('#include "zkgpu.h"\nint main(void) { return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", str(ROOT / "include"), str(tmp_path / "only.c")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I", str(ROOT / "include"), str(tmp_path / "only.c")])


@pytest.mark.gpu
def test_c_client_proves_the_fib_fixture(zk, orc, golden, tmp_path):
    import stark_prover as SP, starkinfo as SI
    ss = {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]}
    su = SP.setup(json.load(open(D / "fib.pil.json.gl")), D / "fib.const.gl", ss, orc)
    exp = SP.to_zkin(SP.stark_gen(D / "fib.cm.gl", su, ss, orc))
    (tmp_path / "prog.json").write_text(json.dumps(SI.to_json(su["starkinfo"], su["program"])))
    (tmp_path / "ss.json").write_text(json.dumps(ss))
    exe = _build(tmp_path)
    out = subprocess.run([str(exe), "run", str(tmp_path / "prog.json"), str(tmp_path / "ss.json"), str(D / "fib.const.gl"), str(D / "fib.cm.gl"),
                          str(tmp_path / "zkin.json")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    root = [int(v) for v in out.stdout.split("const_root ")[1].split("\n")[0].split()]
    assert root == golden["const_root_fib_gl"]["root"]                       # stark_setup.rs:100-116, through C
    assert json.load(open(tmp_path / "zkin.json")) == exp


@pytest.mark.gpu
def test_cli_accepts_zkit_stark_prove_flags(zk, orc, tmp_path):
    """tools/zkgpu_prove.py stark_prove with the flags of test/recursive_proof_to_snark.sh:37-40 on the reference's fixture files"""
    import stark_prover as SP, starkinfo as SI
    ss = {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]}
    su = SP.setup(json.load(open(D / "fib.pil.json")), D / "fib.const", ss, orc)
    exp = SP.to_zkin(SP.stark_gen(D / "fib.cm", su, ss, orc))
    (tmp_path / "prog.json").write_text(json.dumps(SI.to_json(su["starkinfo"], su["program"])))
    (tmp_path / "ss.json").write_text(json.dumps(ss))
    cmd = [sys.executable, str(ROOT / "tools" / "zkgpu_prove.py"), "stark_prove", "-s", str(tmp_path / "ss.json"), "-p", str(D / "fib.pil.json"),
           "--o", str(D / "fib.const"), "--m", str(D / "fib.cm"), "-c", str(tmp_path / "v.circom"), "--i", str(tmp_path / "zkin.json"), "--skip_main",
           "--program", str(tmp_path / "prog.json")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert json.load(open(tmp_path / "zkin.json")) == exp
    bad = subprocess.run(cmd[:-2] + ["--program", str(tmp_path / "ss.json")], capture_output=True, text=True)   # not a program
    assert bad.returncode == 1 and "zkgpu_prove:" in bad.stderr
