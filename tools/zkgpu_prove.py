#!/usr/bin/env python3
"""zkgpu_prove -- the prover sub-commands of the reference's `eigen-zkit` on libzkgpu (last row):

  zkgpu_prove.py stark_prove -s starkStruct.json -p circuit.pil.json --o circuit.const --m circuit.cm \\
                             -c verifier.circom --i zkin.json [--norm_stage] [--skip_main] [--agg_stage] [--prover_addr A] \\
                             [--program starkinfo_program.json]
  zkgpu_prove.py groth16_prove -c BN128 --r1cs circuit.r1cs -w witness.wtns -p g16.key --public-input public_input.json --proof proof.json
  zkgpu_prove.py stark_verify -s starkStruct.json -p circuit.pil.json --o circuit.const --i zkin.json [--program FILE]
  zkgpu_prove.py join_zkin --zkin1 a.zkin.json --zkin2 b.zkin.json --zkinout out.zkin.json
  zkgpu_prove.py stark_aggregate --gpus N --num_proof 8 --workspace DIR [--workers 4] [--keep_proofs]     (starts its own N ranks;
  [torchrun --nproc-per-node N] zkgpu_prove.py stark_aggregate ...                                         or runs under torchrun)

Flags, defaults and file formats are zkit's (zkit/src/main.rs:98-123 StarkProveOpt, :199-217 Groth16ProveOpt;
starky/src/prove.rs:30-160, groth16/src/api.rs:144-205).  What differs, and why:
  * stark_prove needs the code generator's output, `{"starkinfo": StarkInfo, "program": Program}` (serde names, starkinfo.rs:27-95):
    `--program FILE`, or -- when the file is absent -- the library's own generator (zk_starkinfo_generate) if this build has one.
    The reference runs `StarkInfo::new` in process; with the Rust shim (bindings/rust/starky-hip) that is still what happens.
  * `-c/--circom`: the circom verifier text comes from `pil2circom` (template rendering, out of scope): the flag is
    accepted, the file is NOT written and a warning on stderr says so.
  * groth16_prove: `-w` takes the `.wtns` the witness calculator wrote (zkit passes the .wasm and an input.json and runs the
    calculator in process, api.rs:150-160: WASM execution is out of scope); `-i` is accepted and ignored.
  * stark_prove verifies its own proof before it writes anything, as the reference does (prove.rs:124-132; `--no_verify` skips it).
  * stark_verify: the check alone, on a zkin file (the reference exposes it only inside stark_prove).
  * stark_aggregate: test/stark_aggregation.sh:70-73 + :83-156 through eigen-zkvm_amd/aggregation.py -- NUM_PROOF recursion tasks
    sharded over the ranks (one process per GPU), their recursive1 nodes (root1 + digest of the whole proof) all-gathered, joined as
    a tree; the circuits are the synthetic ones of tools/aggregation_workload.py (the real ones are circom-compiled verifiers), so
    the join tree is a workload-shaped stand-in: its root commits to the leaves' proofs, it does not attest them -- each proof's
    validity is the per-proof self check (`each_proof_self_checked`).  Writes aggregation.json
    (every task's roots, the join tree's root, timings) into --workspace on rank 0.
Exit status 0 on success, 1 with the library's message on stderr otherwise (zkit: anyhow error -> exit 1)."""
import argparse
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def _zk():
    import eigen_zkvm_amd
    zk = eigen_zkvm_amd
    if zk.lib().zk_device_count() < 1:
        raise SystemExit("zkgpu_prove: no GPU visible (the library has no CPU fallback)")
    zk.init(0)
    return zk


def stark_prove(a):
    import importlib
    import time
    t_start = time.perf_counter()
    import numpy as np
    zk = _zk()
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    t_init = time.perf_counter()
    ss = json.load(open(a.stark_struct))
    pil = json.load(open(a.piljson))
    # polsarray.rs:137-217: headerless LE u64, row-major.  Mapped, not read: the library's upload is the one pass over the bytes
    # (np.fromfile would copy 5 GB at 2^24 rows before the first of them moves to the GPU)
    const = np.memmap(a.const_pols, dtype="<u8", mode="r")
    cm = np.memmap(a.cm_pols, dtype="<u8", mode="r")
    n = 1 << ss["nBits"]
    if const.size != n * pil["nConstants"] or cm.size != n * pil["nCommitments"]:
        raise SystemExit("zkgpu_prove: %s / %s do not hold 2^%d rows of %d / %d columns"
                         % (a.const_pols, a.cm_pols, ss["nBits"], pil["nConstants"], pil["nCommitments"]))
    if a.program:
        program_json = open(a.program).read()
    elif hasattr(stark, "generate_program"):
        program_json = stark.generate_program(json.dumps(pil), json.dumps(ss))
    else:
        raise SystemExit("zkgpu_prove: --program FILE is required (this build has no code generator)")
    t_gen = time.perf_counter()
    setup = stark.NativeStarkSetup(const, program_json, json.dumps(ss), prover_addr=a.prover_addr if ss.get("verificationHashType") != "GL" else None,
                                   self_check=not a.no_verify)                 # prove.rs:124-132: assert!(stark_verify(..)) before anything is written
    t_setup = time.perf_counter()
    zkin = setup.gen_json(cm)
    t_prove = time.perf_counter()
    with open(a.zkin, "w") as f:
        f.write(zkin)
    split = {"init_s": round(t_init - t_start, 3), "inputs_and_starkinfo_s": round(t_gen - t_init, 3), "setup_s": round(t_setup - t_gen, 3),
             "prove_s": round(t_prove - t_setup, 3), "write_s": round(time.perf_counter() - t_prove, 3), "total_s": round(time.perf_counter() - t_start, 3),
             "setup_split": setup.setup_timing(), "self_check": not a.no_verify}
    setup.free()
    # prove.rs:134-150 always renders the circom verifier into -c; pil2circom is out of scope here: say so when the flag was given
    if a.circom_file:
        print("zkgpu_prove: warning: no circom verifier is written to %s (pil2circom is out of scope of this backend; "
              "the reference's stark_prove writes it)" % a.circom_file, file=sys.stderr)
    print("zkgpu_prove: timing %s" % json.dumps(split), file=sys.stderr)        # prove.rs:95 #[time_profiler("stark_prove")] has the one number
    print("zkgpu_prove: proof of 2^%d rows %swritten to %s (rootC %s)" % (ss["nBits"], "" if a.no_verify else "verified and ", a.zkin, json.loads(zkin)["rootC"]))


def _setup_from_files(a, stark, self_check=False):
    import numpy as np
    ss = json.load(open(a.stark_struct))
    pil = json.load(open(a.piljson))
    const = np.fromfile(a.const_pols, dtype="<u8")
    program_json = open(a.program).read() if a.program else stark.generate_program(json.dumps(pil), json.dumps(ss))
    return stark.NativeStarkSetup(const, program_json, json.dumps(ss), self_check=self_check), ss


def stark_verify(a):
    import importlib
    _zk()
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    setup, ss = _setup_from_files(a, stark)
    ok = setup.verify(open(a.zkin).read())
    why = "" if ok else setup.last_reject()
    setup.free()
    if not ok:
        raise SystemExit("zkgpu_prove: %s does not verify: %s" % (a.zkin, why))
    print("zkgpu_prove: %s verifies (2^%d rows, %s hash)" % (a.zkin, ss["nBits"], ss["verificationHashType"]))


def join_zkin(a):
    """zkit join_zkin --zkin1 A --zkin2 B --zkinout OUT (zkit/src/main.rs JoinZkinExecOpt -> starky/src/zkin_join.rs:9-57); host-only"""
    import importlib
    A = importlib.import_module("eigen_zkvm_amd.aggregation")
    with open(a.zkinout, "w") as f:
        f.write(A.join_zkin_text(open(a.zkin1).read(), open(a.zkin2).read()))
    print("zkgpu_prove: %s + %s -> %s" % (a.zkin1, a.zkin2, a.zkinout))


def stark_aggregate(a):
    import importlib
    import os
    import time
    sys.path.insert(0, str(ROOT / "tools"))
    from eigen_zkvm_amd import launcher
    if a.gpus > 1 and not launcher.under_launcher():                       # --gpus N with no torchrun in front: become the launcher (before any GPU call)
        raise SystemExit(launcher.spawn_ranks([str(pathlib.Path(__file__).resolve())] + a._argv, a.gpus, json_stdout=False))
    import eigen_zkvm_amd as zk
    local_rank = int(os.environ.get("ZK_AGG_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # ZK_AGG_DEVICE: ranks sharing one GPU (tests)
    if zk.lib().zk_device_count() <= local_rank:
        raise SystemExit("zkgpu_prove: no GPU for local rank %d (the library has no CPU fallback)" % local_rank)
    zk.init(local_rank)
    A = importlib.import_module("eigen_zkvm_amd.aggregation")
    import aggregation_workload as AW
    ex = A.RootExchange.from_env(local_rank)
    n = a.num_proof
    per_rank = (n + ex.world - 1) // ex.world
    pool = AW.pool(zk, workers=max(1, min(a.workers, per_rank)), keep_proofs=a.keep_proofs, self_check=not a.no_verify)
    inputs = [pool.task_inputs(u) for u in A.shard_units(n, ex.rank, ex.world)]
    ex.barrier()
    t0 = time.perf_counter()
    res = A.aggregate(pool, inputs, n, ex)
    (dt,) = ex.max([time.perf_counter() - t0])
    seen = ex.gather([ex.rank])
    if ex.rank == 0:
        ws = pathlib.Path(a.workspace); ws.mkdir(parents=True, exist_ok=True)
        out = {"num_proof": n, "ranks": ex.world, "ranks_seen": len({r[0] for r in seen}), "seconds": round(dt, 4),
               # per proof: the library's stark_verify ran on it before it was handed out (prove.rs:124-132).  NOT a statement about the
               # join tree: recursive2 is a stand-in that takes the children's (root, proof digest) as inputs and does not re-verify them
               "each_proof_self_checked": not a.no_verify, "node": "root1 (4 words) + sha256 digest of the whole zkin (4 words)",
               "tasks": {str(u): {k: [str(w) for w in r] for k, r in zip(("fibonacci", "c12", "recursive1"), res["by_task"][u])} for u in sorted(res["by_task"])},
               "join_tree": dict(res["join_tree"], root=[str(w) for w in res["join_tree"]["root"]])}
        (ws / "aggregation.json").write_text(json.dumps(out, indent=1) + "\n")
        print("zkgpu_prove: %d tasks on %d rank(s), %d joins in %d levels, %.3f s; root %s -> %s"
              % (n, ex.world, res["join_tree"]["joins"], res["join_tree"]["levels"], dt, out["join_tree"]["root"], ws / "aggregation.json"))
    if a.keep_proofs:
        ws = pathlib.Path(a.workspace); ws.mkdir(parents=True, exist_ok=True)
        for i, (kind, z) in enumerate(pool.proofs):
            (ws / ("rank%d_%03d_%s.zkin.json" % (ex.rank, i, kind))).write_bytes(z)
        for i, (_node, text) in enumerate(pool.join_inputs):                 # what `zkit join_zkin` writes in front of every recursive2 step (children proved on this rank)
            (ws / ("rank%d_join%03d_input.zkin.json" % (ex.rank, i))).write_text(text)
    pool.free()
    ex.barrier()


def groth16_prove(a):
    import importlib
    zk = _zk()
    dev = importlib.import_module("eigen_zkvm_amd.groth16")
    r1cs, pk, wtns = (pathlib.Path(p).read_bytes() for p in (a.circuit_file, a.pk_file, a.wasm_file))
    if not wtns.startswith(b"wtns"):
        raise SystemExit("zkgpu_prove: -w must be the .wtns file of the witness calculator (running the .wasm is out of scope)")
    setup = dev.Groth16Setup(a.curve_type, r1cs, pk)
    w = dev.wtns_values(wtns, a.curve_type)
    proof, _ = setup.prove(w)
    if a.to_hex:
        raise SystemExit("zkgpu_prove: -t (hex output) is not implemented")
    json.dump(proof, open(a.proof_file, "w"))
    to_int = lambda row: sum(int(v) << (64 * i) for i, v in enumerate(row))
    json.dump([str(to_int(w[i])) for i in range(1, setup.n_inputs)], open(a.public_input_file, "w"))   # api.rs:175-177
    setup.free()
    print("zkgpu_prove: %s proof written to %s" % (a.curve_type, a.proof_file))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="zkgpu_prove", description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("stark_prove", help="Stark proving (zkit/src/main.rs:98-123)")
    s.add_argument("-s", "--stark_stuct", dest="stark_struct", default="stark_struct.json")
    s.add_argument("-p", "--piljson", default="pil.json")
    s.add_argument("-n", "--norm_stage", action="store_true")
    s.add_argument("--skip_main", action="store_true")
    s.add_argument("-a", "--agg_stage", action="store_true")
    s.add_argument("--o", dest="const_pols", default="pols.const")
    s.add_argument("--m", dest="cm_pols", default="pols.cm")
    s.add_argument("-c", "--circom", dest="circom_file", default=None, help="accepted for zkit compatibility; nothing is written (warned)")
    s.add_argument("--i", dest="zkin", default="zkin.json")
    s.add_argument("--prover_addr", default="273030697313060285579891744179749754319274977764")
    s.add_argument("--program", help='{"starkinfo", "program"} JSON of the code generator (extension, see the module text)')
    s.add_argument("--no_verify", action="store_true", help="skip the self check of prove.rs:124-132 (extension)")
    s.set_defaults(fn=stark_prove)
    v = sub.add_parser("stark_verify", help="stark_verify.rs:20-136 on a zkin file (extension: the reference runs it inside stark_prove only)")
    v.add_argument("-s", "--stark_stuct", dest="stark_struct", default="stark_struct.json")
    v.add_argument("-p", "--piljson", default="pil.json")
    v.add_argument("--o", dest="const_pols", default="pols.const")
    v.add_argument("--i", dest="zkin", default="zkin.json")
    v.add_argument("--program")
    v.set_defaults(fn=stark_verify)
    ag = sub.add_parser("stark_aggregate", help="test/stark_aggregation.sh:70-73,83-156: NUM_PROOF tasks sharded over the GPUs + the joins")
    ag.add_argument("--num_proof", type=int, default=8)
    ag.add_argument("--gpus", type=int, default=1, help="ranks to start (one process per GPU) when no launcher (torchrun) is in front")
    ag.add_argument("--workspace", default="/tmp/aggregation")
    ag.add_argument("--workers", type=int, default=4, help="provers in flight per GPU")
    ag.add_argument("--keep_proofs", action="store_true", help="write every proof's zkin into the workspace")
    ag.add_argument("--no_verify", action="store_true")
    ag.set_defaults(fn=stark_aggregate)
    j = sub.add_parser("join_zkin", help="zkin_join.rs:9-57: the input of one recursive2 step from two proofs")
    j.add_argument("--zkin1", required=True); j.add_argument("--zkin2", required=True); j.add_argument("--zkinout", required=True)
    j.set_defaults(fn=join_zkin)
    g = sub.add_parser("groth16_prove", help="Prove with groth16 (zkit/src/main.rs:199-217)")
    g.add_argument("-c", dest="curve_type", default="BN128")
    g.add_argument("--r1cs", dest="circuit_file", required=True)
    g.add_argument("-w", dest="wasm_file", required=True)
    g.add_argument("-p", dest="pk_file", default="g16.zkey")
    g.add_argument("-i", dest="input_file", default=None)
    g.add_argument("--public-input", dest="public_input_file", default="public_input.json")
    g.add_argument("--proof", dest="proof_file", default="proof.json")
    g.add_argument("-t", dest="to_hex", action="store_true")
    g.set_defaults(fn=groth16_prove)
    a = ap.parse_args(argv)
    a._argv = list(sys.argv[1:] if argv is None else argv)
    try:
        a.fn(a)
    except SystemExit:
        raise
    except Exception as e:                                                # anyhow error -> message + exit 1
        print("zkgpu_prove: %s" % e, file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
