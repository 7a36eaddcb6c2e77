"""One-off fuzz of the library's verifier against the oracle's: random single-word tampering anywhere in a proof's zkin (roots, evaluations,
publics, opened values, siblings, FRI roots, last polynomial), all three hash types -- the two verifiers must give the same verdict for every
mutation (and an unmodified proof must be accepted).  python tools/fuzz_verify.py SEED MUTATIONS  (needs a GPU)"""
import copy, importlib, json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "oracle"), str(ROOT / "tools")]
import numpy as np
import eigen_zkvm_amd as zk, oracle_lib, stark_prover as SP, starkinfo as SI
zk.init(0)
stark = importlib.import_module("eigen_zkvm_amd.stark")
orc = oracle_lib.load()
D = ROOT / "tests" / "golden" / "starky_data"
P = zk.P


def leaves(node, path=()):
    """paths of every string leaf of the zkin"""
    if isinstance(node, str):
        yield path
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from leaves(v, path + (i,))
    elif isinstance(node, dict):
        for k, v in node.items():
            if k != "proverAddr":
                yield from leaves(v, path + (k,))


def run(seed, n_mut, verbose=True):
    rng = np.random.default_rng(seed)
    bad, t0, n_rej, n_acc = [], time.time(), 0, 0
    for hash_type, files in (("GL", ("plookup.pil.json.gl", "plookup.const.gl", "plookup.cm.gl")), ("BN128", ("connection.pil.json", "connection.const", "connection.cm")),
                             ("BLS12381", ("fib.pil.json", "fib.const", "fib.cm"))):
        ss = {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": hash_type, "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]}
        pil = json.load(open(D / files[0]))
        const, cm = np.fromfile(D / files[1], dtype="<u8"), np.fromfile(D / files[2], dtype="<u8")
        info, prog, _ = SI.generate(pil, ss)
        ns = stark.NativeStarkSetup(const, stark.generate_program(json.dumps(pil), json.dumps(ss)), json.dumps(ss), prover_addr="1")
        z = ns.gen(cm)
        b = orc if hash_type == "GL" else SP.BN128Backend(orc, hash_type.lower())
        croot = [int(v) for v in ns.const_root()]
        def oracle_verdict(zz):
            try:
                p = SP.from_zkin(zz) if hash_type == "GL" else SP.from_zkin_bn128(zz, b)
                return bool(SP.stark_verify(p, croot, info, prog, ss, b))
            except ValueError as e:
                assert "FRIVerifierFailed" in str(e), e
                return False
        assert ns.verify(z) is True and oracle_verdict(z) is True
        all_leaves = list(leaves(z))
        for m in range(n_mut):
            path = all_leaves[int(rng.integers(0, len(all_leaves)))]
            zz = copy.deepcopy(z)
            node = zz
            for k in path[:-1]:
                node = node[k]
            old = int(node[path[-1]])
            is_fr = hash_type != "GL" and (path[0].startswith("root") or path[0].endswith("_root") or "siblings" in path[0])
            new = (old + int(rng.integers(1, 1 << 20))) % (P if not is_fr else (1 << 250))
            if isinstance(node, dict) and isinstance(z[path[0]], str) and hash_type == "GL":
                new = new % P                                               # a one-string GL digest
            node[path[-1]] = str(new)
            if new == old:
                continue
            with stark.reference_compat_paths():                            # the reference's (lenient) path check: verdicts must equal the restated verifier's
                d = ns.verify(zz)
            o = oracle_verdict(zz)
            strict = ns.verify(zz)                                          # the library's default: every single-word tampering is rejected
            n_rej += (not d); n_acc += bool(d)
            if d != o or strict:
                bad.append((hash_type, path, d, o, strict)); print("DISAGREE", hash_type, path, "device(compat)", d, "oracle", o, "device(strict)", strict, flush=True)
        ns.free()
    if verbose:
        print("fuzz verify seed %d: %d mutations per hash type in %.0f s; reference-compatible mode: rejected %d, accepted %d (the lenient levels of 16-ary paths) = the oracle's verdicts; strict mode (the default) rejected all; disagreements: %s"
              % (seed, n_mut, time.time() - t0, n_rej, n_acc, bad), flush=True)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]))
