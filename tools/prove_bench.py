#!/usr/bin/env python3
"""Times full GL-hash STARK proofs of the synthetic wide-Fibonacci PIL on one MI355X.
usage: python tools/prove_bench.py --nbits 16 18 20 [--w 10] [--verify]
Prints one JSON line per size: setup (const LDE+Merkle, JIT compile) and stark_gen wall time."""
import argparse, json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import importlib
import eigen_zkvm_amd, synth_pil


def poseidong_main(args):
    import poseidong as PG
    zk = eigen_zkvm_amd; zk.init(0)
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    for nbits in args.nbits:
        ss = PG.stark_struct(nbits, hash_type=args.hash)
        pj = PG.program(nbits)
        const, cm = PG.consts(nbits), PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits)
        t0 = time.perf_counter()
        setup = stark.NativeStarkSetup(const, json.dumps(pj), json.dumps(ss))
        zk.lib().zk_dev_sync()
        t_setup = time.perf_counter() - t0
        d_cm = zk.DevArray.from_host(cm)
        times = []
        for _ in range(args.reps):
            t0 = time.perf_counter(); proof = setup.gen(d_cm); times.append(time.perf_counter() - t0)
        out = {"workload": "PoseidonG PIL, nBits=%d, %s hash, %d queries" % (nbits, args.hash, ss["nQueries"]),
               "setup_s": round(t_setup, 3), "setup_split": setup.setup_timing(), "stark_gen_ms": [round(t * 1e3, 1) for t in times], "root1": proof["root1"]}
        if setup.last_timing(): out["stages_ms"] = setup.last_timing()
        t0 = time.perf_counter(); out["verified_by_library"] = bool(setup.verify(proof)); out["verify_ms"] = round((time.perf_counter() - t0) * 1e3, 1)   # zk_stark_verify
        if args.verify and args.hash == "GL":
            sys.path.insert(0, str(ROOT / "oracle"))
            import stark_prover as SP, starkinfo as SI, oracle_lib
            vinfo, vprog, _ = SI.generate(PG.pil(nbits), ss)
            p = SP.from_zkin(proof)
            out["verified"] = bool(SP.stark_verify(p, p["rootC"], vinfo, vprog, ss, oracle_lib.load()))
        setup.free()
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbits", type=int, nargs="+", default=[16])
    ap.add_argument("--w", type=int, default=10)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--hash", default="GL", choices=["GL", "BN128", "BLS12381"], help="verificationHashType")
    ap.add_argument("--python-driver", action="store_true", help="eigen-zkvm_amd/stark.py step-by-step driver instead of zk_stark_gen")
    ap.add_argument("--verify", action="store_true", help="check the proof with the oracle's restated verifier")
    ap.add_argument("--pil", default="poseidong", choices=["poseidong", "widefib"], help="PoseidonG (BASELINE config 3) or the wide-Fibonacci stand-in")
    args = ap.parse_args()
    if args.pil == "poseidong":
        return poseidong_main(args)
    zk = eigen_zkvm_amd; zk.init(0)
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    for nbits in args.nbits:
        prog, ss = synth_pil.program(nbits, args.w, args.hash)
        info = prog["starkinfo"]; info["exp2pol"] = {int(k): v for k, v in info["exp2pol"].items()}
        d = {"program": prog["program"]}
        cm = synth_pil.wide_fib_trace(nbits, args.w)
        const = synth_pil.const_trace(nbits)
        t0 = time.perf_counter()
        if args.python_driver:
            setup = stark.StarkSetup(const, info, d["program"], ss)
        else:                                                              # C++ driver inside libzkgpu
            pj = json.dumps({"starkinfo": dict(info, exp2pol={str(k): v for k, v in info["exp2pol"].items()}), "program": d["program"]})
            setup = stark.NativeStarkSetup(const, pj, json.dumps(ss))
            d_cm = zk.DevArray.from_host(cm)                               # trace resident in HBM
        zk.lib().zk_dev_sync()
        t_setup = time.perf_counter() - t0
        times = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            proof = stark.stark_gen(cm, setup) if args.python_driver else setup.gen(d_cm)
            times.append(time.perf_counter() - t0)
        out = {"workload": "wide-Fibonacci PIL W=%d (%d committed cols), nBits=%d, %s hash, %d queries" % (args.w, 2 * args.w, nbits, args.hash, ss["nQueries"]),
               "setup_s": round(t_setup, 3), "stark_gen_ms": [round(t * 1e3, 1) for t in times], "root1": proof["root1"]}
        if args.verify and args.python_driver:
            sys.path.insert(0, str(ROOT / "oracle"))
            import stark_prover as SP, oracle_lib
            orc = oracle_lib.load()
            vinfo = dict(info); vinfo["ev_idx"] = {"cm": {tuple(k): v for k, v in info["ev_idx"]["cm"]}, "const_": {}}
            out["verified"] = bool(SP.stark_verify(proof, proof["rootC"], vinfo, d["program"], ss, orc))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
