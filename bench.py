#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X.

Workload (config.workload): BASELINE config 2, "2^24-point Goldilocks radix-2 forward+inverse NTT
on one MI355X": one column, n = 2^24, x_i = splitmix64(seed, i) mod p, resident in HBM before the
timed region.  One step = forward NTT then inverse NTT of that column (2 transforms).

  value    = Goldilocks NTT throughput, GElem/s = n_gpus * 2 * 2^24 * steps / time (whole job)
  roofline = dominant kernel (ntt_pass_kernel, 3 launches per transform): achieved =
             ALGORITHMIC bytes per launch (16 B/element/transform, SURVEY 8d, divided over the
             transform's passes) / average launch duration measured with HIP events on the launch
             stream; `pass_hbm_frac` additionally rates the bytes each pass really moves (16 B per
             element per pass) against the 8 TB/s peak.
  cpu_baseline = the CPU oracle's blocked transform (oracle/oracle.c orc_ntt_blocked: in-cache row transforms +
             transposes over all host threads, a port of the structure of fft_p.rs:174-239 -- not the scalar
             fft.rs:39-83) on the same 2^24 column, one step, timed on rank 0's host cores.

N > 1: one process per GPU, each transforming its own column (independent replicas: a single NTT
does not shard without an all-to-all that no BASELINE config needs -- DESIGN.md (e)); no data-path
collective, barrier + max-over-ranks timing only.
"""
import argparse
import ctypes as C
import json
import os
import pathlib
import sys
import time
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))   # the package (eigen_zkvm_amd.py); tests/: oracle_lib, the checker

NBITS = 24
SEED = 0x9E3779B97F4A7C15
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# Integer-ALU ceilings (profiles/r02/ubench_valu.txt, tools/ubench_valu.hip on this chip): a v_mad_u64_u32 -- the 32x32->64
# multiply-add every field product here is built from -- issues once per 4.25 cycles per SIMD at the nominal 2.4 GHz, so
# the chip retires at most 1024 SIMDs * 64 lanes * 2.4e9 / 4.25 = 37 T of them per second.
VALU_MAD_PER_S = 1024 * 64 * 2.4e9 / 4.25
# Fewest multiply-adds one Poseidon-GL permutation takes in the form the kernels use (csrc/poseidon.hip): 118 S-boxes x 4
# products x 4 + 7 dense MDS x 288 (small constants: 2 per term) + pre-sparse matrix 12 x 72 + 22 sparse rounds x (72 + 44).
POSEIDON_MADS = 118 * 16 + 7 * 288 + 12 * 72 + 22 * (72 + 44)
# A BN254 Fq product in 9 x 29-bit limbs: 81 + 81 multiply-adds (product + Montgomery reduction), csrc/fe29_impl.hip.h;
# a mixed point addition (madd-2008-s) is 8 products + 2 squares... counted as 11 products with the doubling check.
FQ_MADS = {"bn254": 2 * 9 * 9, "bls12_381": 2 * 14 * 14}
PADD_PRODUCTS = 11


def pmc_traffic(nbits):
    """roofline.traffic: HBM bytes per ntt_pass_kernel launch from the PMC counters -- rocprofv3 --pmc FETCH_SIZE and --pmc
    WRITE_SIZE in separate passes with the gfx950 fetch correction calibrated on a 1 GiB copy of the same run
    (tools/gpu_round.sh profiles -> tools/pmc_summarize.py -> profiles/rNN/pmc_hbm_traffic.json).  Counters cannot be read by the
    run that is being timed, so the number is taken from the newest committed profile and only if that profile was
    collected on the kernel sources this build has (src_sha16); otherwise null."""
    sys.path.insert(0, str(ROOT / "tools"))
    import pmc_summarize
    files = sorted(ROOT.glob("profiles/r*/pmc_hbm_traffic.json"))
    if not files or nbits != 24:
        return None, "no PMC profile for this size"
    d = json.loads(files[-1].read_text())
    rel = str(files[-1].relative_to(ROOT))
    if d.get("src_sha16") != pmc_summarize.ntt_src_sha16() or not d.get("ntt_pass_2p24"):
        return None, "%s was collected on other kernel sources" % rel
    return d["ntt_pass_2p24"]["bytes_per_launch"], rel


def _agg():
    """the product's aggregation driver (eigen-zkvm_amd/aggregation.py): sharding, the prover pool, the join tree, the root exchange"""
    import importlib
    return importlib.import_module("eigen_zkvm_amd.aggregation")


def _golden(nbits):
    """the offline oracle golden of the 2^nbits PoseidonG proof (tools/gen_golden_full.py), or None"""
    for name in ("poseidong_2p%d.json" % nbits, "poseidong_2p%d_roots.json" % nbits):
        f = ROOT / "tests" / "golden" / name
        if f.exists():
            return json.loads(f.read_text())
    return None


def one_shot_leg(zk, nbits, ss, const, cm, proof_text):
    """The reference's unit of work (starky/src/prove.rs:95-160, `zkit stark_prove`): ONE fresh process from the .const / .cm / pil.json /
    starkStruct files to a verified zkin on disk -- StarkInfo::new, StarkSetup::new, stark_gen, stark_verify, write.  Timed around the child
    process (tools/zkgpu_prove.py stark_prove, the zkit-flag mirror), with the code-object cache warm (the setup of this leg compiled the
    same programs) and the input files in the page cache (they were just written); the child's own split rides along."""
    import hashlib, shutil, subprocess, tempfile
    sys.path.insert(0, str(ROOT / "tools"))
    import poseidong as PG
    d = pathlib.Path(tempfile.mkdtemp(prefix="zk_one_shot_"))
    try:
        const.tofile(d / "c.const"); cm.tofile(d / "c.cm")
        (d / "pil.json").write_text(json.dumps(PG.pil(nbits))); (d / "ss.json").write_text(json.dumps(ss))
        # (no zk_dev_trim here: VRAM this process hands back is scrubbed by the driver before the child may have it, and the child would wait --
        # measured 1.7 s of "setup" that way; 2 x 45 GB fit side by side)
        cmd = [sys.executable, str(ROOT / "tools" / "zkgpu_prove.py"), "stark_prove", "-s", str(d / "ss.json"), "-p", str(d / "pil.json"),
               "--o", str(d / "c.const"), "--m", str(d / "c.cm"), "--i", str(d / "zkin.json")]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": "zkgpu_prove.py stark_prove exited %d: %s" % (r.returncode, r.stderr[-300:])}
        split = {}
        for l in r.stderr.splitlines():
            if l.startswith("zkgpu_prove: timing "):
                split = json.loads(l[len("zkgpu_prove: timing "):])
        same = hashlib.sha256((d / "zkin.json").read_bytes()).hexdigest() == hashlib.sha256(proof_text.encode() if isinstance(proof_text, str) else proof_text).hexdigest()
        return {"s": round(wall, 3), "what": "fresh process: tools/zkgpu_prove.py stark_prove from .const/.cm/pil.json/starkStruct files (%.1f GB, page cache) to a "
                "self-checked zkin on disk; code-object cache warm" % ((const.nbytes + cm.nbytes) / 1e9),
                "child_split": split, "zkin_equals_the_timed_proof": bool(same)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def prove_leg(zk, nbits, verify=True, cpu_baseline=False, host_trace=True, one_shot=False):
    """Third component of BASELINE's metric, "starky prove ms at 2^24 rows": one full GL-hash STARK proof
    (stark_gen.rs:193-557: LDE + Poseidon Merkle + constraint evaluation + FRI) of BASELINE's own workload, the
    PoseidonG PIL (starkjs/poseidon/poseidong.pil; compiled form tests/golden/poseidong.pil.json): 19 committed +
    18 constant columns, 36 intermediate columns (cm3), quotient in 2 x 3 columns (cm4), every 31-row slot of the trace
    hashing its own input (tools/tracegen.c).  Timed with the trace resident in HBM (`ms`) and handed over in host
    memory as `zkit stark_prove` has it after loading the .cm file (`ms_from_host_trace`).  After the clock stops the
    timed proof is checked three ways: `verified` = the product's own stark_verify (zk_stark_verify; stark_verify.rs:20-136),
    `verified_oracle` = the restated verifier of oracle/stark_prover.py, `golden` = roots / evaluations / sha256 of the whole
    zkin against the fixture the CPU oracle prover produced offline (tools/gen_golden_full.py)."""
    import importlib
    sys.path.insert(0, str(ROOT / "tools"))
    import poseidong as PG
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    ss = PG.stark_struct(nbits)
    pj = PG.program(nbits)
    info = pj["starkinfo"]
    const, cm = PG.consts(nbits), PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits)
    t0 = time.perf_counter()
    setup = stark.NativeStarkSetup(const, json.dumps(pj), json.dumps(ss))    # C++ driver inside libzkgpu
    zk.lib().zk_dev_sync()
    setup_s = time.perf_counter() - t0
    const_for_files = const if one_shot else None
    del const
    times, times_h2d = [], []
    d_cm = zk.DevArray.from_host(cm)                                        # trace resident in HBM when the clock starts
    first_stages = None
    for k in range(4):
        if k == 0:
            os.environ["ZK_STARK_TIMING"] = "quiet"                          # the first proof of a setup with the library's stage timers on: where its extra time goes
        t0 = time.perf_counter()
        proof_text = setup.gen_json(d_cm)
        times.append((time.perf_counter() - t0) * 1e3)
        if k == 0:
            first_stages = setup.last_timing(); os.environ.pop("ZK_STARK_TIMING", None)
    proof_dev = json.loads(proof_text)
    for _ in range(2 if host_trace else 0):                                # trace handed over in host memory
        t0 = time.perf_counter()
        proof = setup.gen(cm)
        times_h2d.append((time.perf_counter() - t0) * 1e3)
        assert proof_dev == proof
    n_ext = 1 << (nbits + 1)
    sN = info["map_sectionsN"]
    perms = sum((_linearhash_perms(sN[s]) + 1) * n_ext for s in ("cm1_2ns", "cm3_2ns", "cm4_2ns"))
    out = {"workload": "BASELINE config 3 PIL%s: PoseidonG (starkjs/poseidon/poseidong.pil), nBits=%d, nBitsExt=%d, "
                       "GL hash, %d queries, FRI steps %s; trace = %d Poseidon permutations, one input per 31-row slot"
                       % (" at the headline size" if nbits == 24 else "", nbits, nbits + 1, ss["nQueries"], [s["nBits"] for s in ss["steps"]], (1 << nbits) // 31),
           "columns": {"cm1": info["n_cm1"], "const": info["n_constants"], "cm2": info["n_cm2"], "cm3": info["n_cm3"],
                       "cm4_words": sN["cm4_2ns"], "q_deg": info["q_deg"], "q_dim": info["q_dim"], "evals": len(info["ev_map"])},
           "ms": round(min(times[1:]), 1), "ms_runs": [round(t, 1) for t in times], "first_run_stages_ms": first_stages, "setup_s": round(setup_s, 2),
           "poseidon_perms_per_proof": perms, "root1": proof_dev["root1"], "stand_in": False}
    if times_h2d:
        out.update({"ms_from_host_trace": round(min(times_h2d), 1), "host_trace_GB": round(cm.nbytes / 1e9, 2)})
    out["setup_split"] = setup.setup_timing()                             # StarkSetup::new: JSON / constants / step programs (code-object cache)
    # The proof is four fifths Poseidon permutations (the three trees over the extended sections): rated as a whole against
    # the same integer-ALU ceiling as the Merkle leg -- every millisecond that is not hashing lowers the fraction.
    rate = perms / (min(times[1:]) * 1e-3)
    out["roofline"] = {"bound": "int-alu", "kernel": "linearhash_rows_kernel + merkle_level_kernel inside stark_gen", "achieved": round(rate / 1e9, 3),
                       "peak": round(VALU_MAD_PER_S / POSEIDON_MADS / 1e9, 3), "unit": "Gperm/s", "frac": round(rate / (VALU_MAD_PER_S / POSEIDON_MADS), 4),
                       "model": "the proof's %d permutations / its whole time, against %d v_mad_u64_u32 per permutation at the measured issue rate" % (perms, POSEIDON_MADS)}
    if one_shot:
        out["one_shot"] = one_shot_leg(zk, nbits, ss, const_for_files, cm, proof_text)
    if verify:                                                              # after the clock has stopped
        t0 = time.perf_counter()
        out["verified"] = bool(setup.verify(proof_text))                    # the product's verifier (zk_stark_verify)
        out["verify_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        bad = dict(proof_dev, evals=[[str((int(proof_dev["evals"][0][0]) + 1) % 0xFFFFFFFF00000001)] + proof_dev["evals"][0][1:]] + proof_dev["evals"][1:])
        out["tampered_rejected"] = not setup.verify(bad)
        try:                                                                # the checker
            sys.path.insert(0, str(ROOT / "oracle"))
            import oracle_lib, stark_prover as SP, starkinfo as SI
            vinfo, vprog, _ = SI.generate(PG.pil(nbits), ss)
            p = SP.from_zkin(proof_dev)
            out["verified_oracle"] = bool(SP.stark_verify(p, [int(v) for v in setup.const_root()], vinfo, vprog, ss, oracle_lib.load()))
        except Exception as e:
            out["verified_oracle"] = "error: %s: %s" % (type(e).__name__, e)
        gold = _golden(nbits)
        if gold is not None and gold.get("starkStruct") == ss:
            sys.path.insert(0, str(ROOT / "tools"))
            from gen_golden_full import zkin_digest
            g = {"fixture": "tests/golden/poseidong_2p%d%s.json" % (nbits, "" if "zkin_digest" in gold else "_roots"),
                 "rootC_matches_golden": [str(v) for v in setup.const_root()] == gold["rootC"]}
            if "root1" in gold:
                g["root1_matches_golden"] = proof_dev["root1"] == gold["root1"]
            if "zkin_digest" in gold:
                g["zkin_sha256_matches_golden"] = zkin_digest(proof_dev) == gold["zkin_digest"]
                g["all_roots_evals_finalpol_match"] = all(proof_dev[k] == gold[k] for k in gold if k in proof_dev)
            out["golden"] = g
    if cpu_baseline:
        # the reference's CPU path beside it: the restated prover (oracle/stark_prover.py over oracle/*.c, OpenMP) on the SAME PIL at
        # 2^16 rows -- a bounded sample; rows/s compare, a 2^24-row CPU proof takes the build container an hour (tools/gen_golden_full.py)
        try:
            sys.path.insert(0, str(ROOT / "oracle"))
            import oracle_lib, stark_prover as SP
            orc = oracle_lib.load()
            nb = 16
            ss16 = PG.stark_struct(nb)
            su = SP.setup(PG.pil(nb), PG.consts(nb), ss16, orc)
            cm16 = PG.trace(nb, None, PG.FIRST_ZERO, seed=nb)
            t0 = time.perf_counter(); exp = SP.to_zkin(SP.stark_gen(cm16, su, ss16, orc)); cpu_s = time.perf_counter() - t0
            ns16 = stark.NativeStarkSetup(PG.consts(nb), json.dumps(PG.program(nb)), json.dumps(ss16))
            same = ns16.gen(cm16) == exp
            ns16.free()
            out["cpu_baseline"] = {"value": round((1 << nb) / cpu_s / 1e6, 4), "unit": "Mrows/s", "cores": orc.threads(), "kind": "port, python-driven, 2^16-row sample",
                                   "gpu_value": round((1 << nbits) / (min(times[1:]) * 1e-3) / 1e6, 2), "gpu_proof_equals_cpu_proof_on_the_sample": bool(same),
                                   "sample": "the same PIL at 2^16 rows, oracle/stark_prover.py stark_gen (C kernels with OpenMP, Python driver), %.2f s; "
                                             "the offline 2^%d-row oracle proof behind tests/golden took %s s" % (cpu_s, nbits, (_golden(nbits) or {}).get("oracle_seconds", "n/a"))}
        except Exception as e:
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    setup.free()
    return out


def _linearhash_perms(w):
    """Poseidon permutations of one LinearHash row of w words (linearhash.rs:79-145)"""
    if w <= 4:
        return 0
    bs = max(8, (w + 3) // 4)
    n_b = (w + bs - 1) // bs
    perms = sum(((min(bs, w - i * bs) + 7) // 8) if min(bs, w - i * bs) > 4 else 0 for i in range(n_b))
    return perms + ((4 * n_b + 7) // 8 if 4 * n_b > 4 else 0)


def poseidon_leg(zk, log_height, width, cpu_baseline):
    """Poseidon-GL Merkle tree (merklehash.rs:293-346) over an HBM-resident [2^log_height][width] matrix -- 73 % of a proof.
    Integer-ALU bound: `roofline` rates the permutations per second against the multiply-add issue ceiling."""
    h = 1 << log_height
    if h * width < (1 << 26):
        rng = np.random.default_rng(0x905E)
        rows = rng.integers(0, 0xFFFFFFFF00000001, size=h * width, dtype=np.uint64)
        d = zk.DevArray.from_host(rows)
    else:                                                                # the large shapes are born in HBM (splitmix64 mod p); the CPU sample is read back
        d = zk.DevArray(h * width)
        zk._check(zk.lib().zk_dev_fill_splitmix(d.ptr, h * width, 0x905E, None))
        rows = np.empty((1 << 18) * width, np.uint64)
        zk._check(zk.lib().zk_dev_download(zk._ptr(rows), d.ptr, rows.nbytes))
    times = []
    for _ in range(5):
        zk.lib().zk_dev_sync()
        t0 = time.perf_counter()
        tr = zk.MerkleTreeGL(); tr.merkelize_dev(d.ptr, width, h); zk.lib().zk_dev_sync()
        times.append(time.perf_counter() - t0)
        root = [int(v) for v in tr.root()]; tr.free()
    perms = (_linearhash_perms(width) + 1) * h
    rate = perms / min(times[1:])
    peak = VALU_MAD_PER_S / POSEIDON_MADS
    res = {"workload": "MerkleTreeGL, 2^%d rows x %d columns, HBM-resident" % (log_height, width), "ms": round(min(times[1:]) * 1e3, 2),
           "permutations": perms, "value": round(rate / 1e9, 3), "unit": "Gperm/s",
           "roofline": {"bound": "int-alu", "kernel": "linearhash_rows_kernel + merkle_level_kernel", "achieved": round(rate / 1e9, 3),
                        "peak": round(peak / 1e9, 3), "unit": "Gperm/s", "frac": round(rate / peak, 4),
                        "model": "%d v_mad_u64_u32 per permutation at the measured issue rate (profiles/r02/ubench_valu.txt)" % POSEIDON_MADS}}
    if cpu_baseline:
        import oracle_lib
        orc = oracle_lib.load()
        m = 1 << 18                                                      # bounded CPU sample: the first 2^18 rows
        t0 = time.perf_counter(); exp = orc.merkelize(rows[:m * width], width, m); cpu_s = time.perf_counter() - t0
        ts = zk.MerkleTreeGL(); ts.merkelize(rows[:m * width], width, m)
        assert np.array_equal(ts.nodes(), exp), "GPU Merkle tree != CPU oracle"
        res["cpu_baseline"] = {"value": round((_linearhash_perms(width) + 1) * m / cpu_s / 1e9, 5), "unit": "Gperm/s", "cores": orc.threads(),
                               "kind": "port", "sample": "2^18-row tree of the same matrix, oracle/oracle.c (OpenMP over rows), %.2f s" % cpu_s}
    return res


class FinalWrap:
    """BASELINE config 5, the serial tail on rank 0 after the joins (test/stark_aggregation.sh:159-210): the compressor's exec
    step (witness -> trace, compressor12_exec.rs:17-125, on the device), the final STARK -- the compressor-shaped circuit with the
    reference's own final.starkStruct.bls12381.json (2^16 rows, BLS12381 hashing, FRI steps 17 -> 7 -> 3: a 10-bit fold) -- and the
    BLS12-381 Groth16 wrap (the recursive circuits themselves need circom).  The final circuit is the layered stand-in the joins use
    (tools/aggregation_workload.py JoinCircuit: same PIL shape, linear gates in layers, so that ITS trace is what compressor12 exec
    computes from the 17-word witness vector [1, 16 primary inputs = the aggregate's node] -- no host walk over the gates inside the
    clock, as in the reference, where exec reads the witness calculator's output).
    The constructor builds what the script's FIRST_RUN builds (setups, keys, the .exec handle); `run(join_root)` is the timed part."""
    SS = {"nBits": 16, "nBitsExt": 17, "nQueries": 8, "verificationHashType": "BLS12381", "steps": [{"nBits": 17}, {"nBits": 7}, {"nBits": 3}]}

    def __init__(self, zk, log_rows=18):
        import importlib
        sys.path.insert(0, str(ROOT / "tools"))
        import groth16_bench as GB
        import aggregation_workload as AW, poseidong as PG
        self.zk, self.log_rows = zk, log_rows
        self.stark = importlib.import_module("eigen_zkvm_amd.stark")
        dev = importlib.import_module("eigen_zkvm_amd.groth16")
        c12 = importlib.import_module("eigen_zkvm_amd.compressor12")
        self.AW, self.PG = AW, PG
        # 1. + 2. the final circuit: its .exec (compressor12 exec on the device) and its setup under BLS12381 hashing
        self.jc = AW.JoinCircuit(self.SS["nBits"])
        self.E = c12.Compressor12Exec(self.jc.exec_text(), AW.JoinCircuit.N_WITNESS)
        self.setup = self.stark.NativeStarkSetup(self.jc.consts, json.dumps(PG.native_program(AW.c12_pil(self.SS["nBits"]), self.SS)), json.dumps(self.SS))
        # 3. Groth16 wrap on BLS12-381
        rb, wit, ni, n_wires = GB.make_circuit(GB.FR["BLS12381"], log_rows)
        self.S = dev.Groth16Setup("BLS12381", rb, GB.make_params(zk, dev, "BLS12381", ni, n_wires, log_rows, GB.density(rb, ni, n_wires)))
        self.d_wit = zk.DevArray.from_host(wit.reshape(-1))
        self.run([1, 2, 3, 4])                                             # warm: pool blocks, code objects

    def primary(self, join_root):
        return (([int(w) for w in join_root] if join_root else [1, 2, 3, 4]) + [0] * 16)[:16]   # the aggregate's node: root1 + proof digest

    def run(self, join_root):
        """exec -> final STARK -> Groth16, one after the other as the script runs them; -> per-stage milliseconds + the final zkin"""
        zk = self.zk
        t0 = time.perf_counter()
        d_w = zk.DevArray.from_host(self.AW.JoinCircuit.witness_vector(self.primary(join_root)))   # what the witness calculator hands to exec: 17 words
        d_cm = self.E.run(d_w, 1 << self.SS["nBits"]); zk.lib().zk_dev_sync()                     # PlonkAdds by depth + the 12-column gather, in HBM
        t1 = time.perf_counter()
        z = self.setup.gen(d_cm)
        t2 = time.perf_counter()
        self.S.prove(self.d_wit, 5, 7)
        t3 = time.perf_counter()
        return {"c12_exec_ms": round((t1 - t0) * 1e3, 2), "final_stark_bls12381_ms": round((t2 - t1) * 1e3, 1),
                "groth16_bls12381_ms": round((t3 - t2) * 1e3, 2), "ms": round((t3 - t0) * 1e3, 1)}, z

    def leg(self, join_root):
        """the timed serial tail + its check after the clock"""
        out = {"workload": "compressor12 exec of the final circuit (2^16 rows x 12 columns, depth %d) + final STARK (compressor-shaped circuit, "
                           "final.starkStruct.bls12381.json: 2^16 rows, BLS12381 hash) + Groth16 BLS12381 (2^%d rows), rank 0" % (int(self.E.depth), self.log_rows)}
        t, z = self.run(join_root)
        out.update(t); out["c12_exec_depth"] = int(self.E.depth)
        out["final_stark_root1"] = z["root1"]
        primary = self.primary(join_root)
        try:   # after the clock: the restated verifier (oracle/, stark_verify.rs:20-136 with MerkleTreeBLS12381 / TranscriptBLS12381) on the zkin text
            sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
            import stark_prover as SP, starkinfo as SI, oracle_lib
            b = SP.BN128Backend(oracle_lib.load(), "bls12381")
            vinfo, vprog, _ = SI.generate(self.AW.c12_pil(self.SS["nBits"]), self.SS)
            pz = SP.from_zkin_bn128(z, b)
            croot = [int(v) for v in self.setup.const_root()]             # the setup's own root of the constants (raw limbs), not the proof's copy
            out["final_stark_verified"] = bool(SP.stark_verify(pz, croot, vinfo, vprog, self.SS, b)) and z["publics"][:3] == [str(v) for v in primary[:3]]
            out["final_stark_verified_by_library"] = bool(self.setup.verify(z))   # the product's own (strict) stark_verify
        except Exception as e:                                              # the checker failing must not lose the measured line
            out["final_stark_verified"] = "error: %s: %s" % (type(e).__name__, e)
        return out

    def free(self):
        self.E.free(); self.setup.free(); self.S.free()


def aggregation_leg(pool, ex, n_tasks=8, make_wrap=None):
    """BASELINE config 5 through the product's driver (eigen-zkvm_amd/aggregation.py; test/stark_aggregation.sh:70-73, :83-156): a FIXED
    set of `n_tasks` independent recursion tasks, task u on rank u mod world, no collective while proving; the one exchange is the
    all-gather of the tasks' roots (3 x 32 B per task), then the joins as a tree with one all-gather per level.  This function only
    holds the clock: throughput = tasks / slowest rank's time, so it scales with the GPUs (strong scaling of the 8-task job).
    `pool` is aggregation.ProverPool over the circuits of tools/aggregation_workload.py, or a stub in the CPU tests; `ex` the RootExchange.
    `make_wrap()` -> the serial tail of rank 0 (FinalWrap; an object with .leg(root) and .run(root)).  After the per-phase clocks the whole
    job runs ONCE MORE under one clock -- tasks, root exchange, join tree, rank 0's final wrap, back to back as the script runs them:
    `end_to_end_s`, max over ranks."""
    A = _agg()
    rank, world = ex.rank, ex.world
    units = A.shard_units(n_tasks, rank, world)
    inputs = [pool.task_inputs(u) for u in units]                       # witness generation + upload: before the clock
    if inputs:                                                          # warm-up (pool, JIT modules of every worker's setups)
        pool.prove_all(inputs) if hasattr(pool, "prove_all") else pool.prove(inputs[0])
    pool.sync()
    ex.barrier()
    t0 = time.perf_counter()
    roots = pool.prove_all(inputs) if hasattr(pool, "prove_all") else [pool.prove(i) for i in inputs]   # no collective inside the clock
    pool.sync()
    dt = time.perf_counter() - t0
    (dt,) = ex.max([dt])
    lat, lat_split = None, None
    if inputs:                                                          # one task alone: the floor of the job once every rank holds one task
        t1 = time.perf_counter(); pool.prove(inputs[0]); pool.sync(); lat = time.perf_counter() - t1
        if hasattr(pool, "stage_times"):                                 # the same task once more with the library's stage timers on
            lat_split = pool.stage_times(inputs[0])
    # the other reading of "aggregate throughput": every rank proves `n_tasks` tasks of its own (weak scaling) -- what a node does with a
    # stream of recursion tasks when one small task cannot fill a GPU; at one rank it is the strong number again, measured a second time
    w_inputs = inputs if world == 1 else [pool.task_inputs(n_tasks * (rank + 1) + k) for k in range(n_tasks)]
    pool.sync(); ex.barrier()
    t1 = time.perf_counter()
    pool.prove_all(w_inputs) if hasattr(pool, "prove_all") else [pool.prove(i) for i in w_inputs]
    pool.sync()
    (wdt,) = ex.max([time.perf_counter() - t1])
    by_task = A.gather_task_roots(roots, n_tasks, ex, 3, getattr(pool, "node_words", 4))
    out = {"workload": "BASELINE config 5 (sharded part): %d recursion tasks, task u on rank u mod %d, each = %s; "
                       "witnesses resident in HBM, root all-gather only" % (n_tasks, world, getattr(pool, "description", "stub")),
           "driver": "eigen-zkvm_amd/aggregation.py (ProverPool, prove_tasks, join_tree, RootExchange)",
           "tasks": n_tasks, "tasks_per_s": round(n_tasks / dt, 3), "proofs_per_s": round(3 * n_tasks / dt, 3), "s": round(dt, 4),
           "task_latency_s": None if lat is None else round(lat, 4),
           # what 8 GPUs can make of this job at best: every rank holds one task, so the job cannot finish before one task does
           "scaling_ceiling": None if not lat else round(dt / lat, 2),
           "task_latency_split": lat_split,
           "n_gpus": world, "scaling": "strong (fixed %d tasks)" % n_tasks,
           "weak": {"scaling": "weak (%d tasks per rank)" % n_tasks, "tasks": n_tasks * world, "s": round(wdt, 4), "tasks_per_s": round(n_tasks * world / wdt, 3)},
           "distinct_roots": len({tuple(w for r in v for w in r) for v in by_task.values()}), "tasks_gathered": sorted(by_task)}
    # the join phase, timed
    if hasattr(pool, "warm_join"):
        pool.warm_join()
    ex.barrier()
    t0 = time.perf_counter()
    jt = A.join_tree(pool, [by_task[u][2] for u in sorted(by_task)], ex)
    (jdt,) = ex.max([time.perf_counter() - t0])
    jt["s"] = round(jdt, 4)
    if hasattr(pool, "join_exec_s") and jt["joins"]:  # where a join's time goes on this rank: the host exec step (witness of the joined circuit) and the proof
        mine = max(1, sum(1 for _ in A.shard_all_joins(len(by_task), rank, world)))
        jt["per_join_ms"] = {"exec_host": round(1e3 * sum(pool.join_exec_s) / mine, 2), "exec_dev": round(1e3 * sum(pool.join_exec_dev_s) / mine, 2),
                             "prove": round(1e3 * sum(pool.join_prove_s) / mine, 1)}
    out["join_tree"] = jt
    wrap = None
    if make_wrap is not None and rank == 0:                              # the serial tail lives on rank 0 (test/stark_aggregation.sh:159-210)
        try:
            wrap = make_wrap()
            out["final_wrap"] = wrap.leg(jt["root"])
        except Exception as e:                                           # never lose the bench line to the extra leg
            wrap, out["final_wrap"] = None, {"error": "%s: %s" % (type(e).__name__, e)}
    # the whole job under ONE clock: tasks -> all-gather -> joins (one all-gather per level) -> final wrap on rank 0
    ex.barrier()
    t0 = time.perf_counter()
    res = A.aggregate(pool, inputs, n_tasks, ex)
    t_sharded = time.perf_counter() - t0
    if wrap is not None:
        wrap.run(res["join_tree"]["root"])
    (e2e, t_sharded) = ex.max([time.perf_counter() - t0, t_sharded])
    out["end_to_end_s"] = round(e2e, 4)
    out["end_to_end"] = {"s": round(e2e, 4), "tasks_and_joins_s": round(t_sharded, 4), "includes_final_wrap": wrap is not None,
                         "root_equals_phase_run": res["join_tree"]["root"] == jt["root"],
                         "what": "prove_tasks + root all-gather + join_tree (%d levels, one all-gather each)%s, one clock, max over ranks"
                                 % (jt["levels"], " + rank 0's exec / final STARK / Groth16" if wrap is not None else "")}
    if wrap is not None:
        wrap.free()
    return out


def bn128_merkle_leg(zk, log_height, width, cpu_baseline):
    """SURVEY 8f-1 (first widening row): MerkleTreeBN128 over an HBM-resident [2^log_height][width] matrix -- the
    tree the final STARK of an aggregation commits to (c12a-sized: 12 columns); checked node for node against the
    CPU oracle on a bounded sample, which is also the timed CPU baseline."""
    h = 1 << log_height
    zk.bn128_init()
    if h * width < (1 << 26):
        rng = np.random.default_rng(0xB128)
        rows = rng.integers(0, 0xFFFFFFFF00000001, size=h * width, dtype=np.uint64)
        d = zk.DevArray.from_host(rows)
    else:                                                                # the large shapes are born in HBM (splitmix64 mod p); the CPU sample is read back
        d = zk.DevArray(h * width)
        zk._check(zk.lib().zk_dev_fill_splitmix(d.ptr, h * width, 0xB128, None))
        rows = np.empty((1 << 19) * width, np.uint64)
        zk._check(zk.lib().zk_dev_download(zk._ptr(rows), d.ptr, rows.nbytes))
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        tr = zk.MerkleTreeBN128(); tr.merkelize_dev(d.ptr, width, h); zk.lib().zk_dev_sync()
        times.append(time.perf_counter() - t0)
        root = tr.root(); tr.free()
    nb = (width - 1) // 3 + 1 if width > 4 else 0
    n, nodes = h, 0
    while n > 1:
        n = (n - 1) // 16 + 1; nodes += n
    res = {"workload": "MerkleTreeBN128, 2^%d rows x %d Goldilocks columns, HBM-resident" % (log_height, width),
           "ms": round(min(times[1:]) * 1e3, 2), "rows_per_s": round(h / min(times[1:]), 1),
           "permutations": {"leaf_t%d" % (nb + 1): h * ((nb + 15) // 16), "node_t17": nodes}}
    # Fr products of a Poseidon-BN128 permutation of width t (poseidon_bn128_opt.rs:98-224): 8 full rounds of t S-boxes (x^5: 3 products)
    # and a dense t x t matrix, the pre-sparse matrix, N_P(t) partial rounds of one S-box and a sparse row + column (2t - 1 products)
    n_p = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]
    fr_products = lambda t: 8 * (3 * t + t * t) + t * t + n_p[t - 2] * (3 + 2 * t - 1)
    t_leaf = min(nb, 16) + 1
    products = h * ((nb + 15) // 16) * fr_products(t_leaf) + nodes * fr_products(17)
    peak = VALU_MAD_PER_S / FQ_MADS["bn254"]
    res["roofline"] = {"bound": "int-alu", "kernel": "bn128_leaf_reg_kernel + bn128 node kernels", "achieved": round(products / min(times[1:]) / 1e9, 1),
                       "peak": round(peak / 1e9, 1), "unit": "G Fr products/s", "frac": round(products / min(times[1:]) / peak, 4),
                       "model": "%d Fr products per tree (t = %d leaves: %d each, t = 17 nodes: %d each), %d v_mad_u64_u32 per product in 29-bit limbs at the measured issue rate"
                                % (products, t_leaf, fr_products(t_leaf), fr_products(17), FQ_MADS["bn254"])}
    if cpu_baseline:
        import oracle_lib
        ob = oracle_lib.load().bn128()
        m = 1 << 19                                                      # bounded CPU sample: the first 2^19 rows
        t0 = time.perf_counter()
        exp = ob.merkelize(rows[:m * width], width, m)
        cpu_s = time.perf_counter() - t0
        ts = zk.MerkleTreeBN128(); ts.merkelize(rows[:m * width], width, m)
        assert np.array_equal(ts.nodes(), exp), "GPU BN128 tree != CPU oracle"
        res["cpu_baseline"] = {"value": round(m / cpu_s, 1), "unit": "rows/s", "cores": oracle_lib.load().threads(), "kind": "port",
                               "sample": "2^19-row tree of the same matrix, oracle/bn128_hash.c (OpenMP), %.2f s" % cpu_s}
    return res


def groth16_leg(zk, curve, log_rows, cpu_baseline):
    """one Groth16 proof of a synthetic satisfied circom-shaped circuit of 2^log_rows rows
    (tools/groth16_bench.py: ~0.7 M / 0.5 M dense A / B columns at 2^20), key and witness resident in HBM.  The key
    holds arbitrary valid points, so proofs made here do not verify; validity is tests/test_gpu_groth16.py's job."""
    import importlib
    sys.path.insert(0, str(ROOT / "tools"))
    import groth16_bench as GB
    dev = importlib.import_module("eigen_zkvm_amd.groth16")
    rb, wit, ni, n_wires = GB.make_circuit(GB.FR[curve], log_rows)
    pb = GB.make_params(zk, dev, curve, ni, n_wires, log_rows, GB.density(rb, ni, n_wires))
    S = dev.Groth16Setup(curve, rb, pb)
    d_w = zk.DevArray.from_host(wit.reshape(-1))
    d_h = zk.DevArray(4 * ((1 << log_rows) - 1), zero=True)
    S.prove(d_w, 5, 7, d_h=d_h)
    assert S.domain_log == log_rows
    ts, th = [], []
    for _ in range(5):
        t0 = time.perf_counter(); S.prove(d_w, 5, 7); ts.append(time.perf_counter() - t0)
    for _ in range(2):
        t0 = time.perf_counter(); S.prove(wit, 5, 7); th.append(time.perf_counter() - t0)
    res = {"workload": "Groth16 %s proof, synthetic circuit of 2^%d rows, %d wires; quotient (7 Fr transforms) + 3 multi-scalar sums" % (curve, log_rows, n_wires),
           "ms": round(min(ts) * 1e3, 2), "ms_from_host_witness": round(min(th) * 1e3, 2), "value": round((1 << log_rows) / min(ts) / 1e6, 2), "unit": "Mrows/s"}
    # Fq products of the three multi-scalar sums over the window tables: one mixed addition (11 products) per (point, 16-bit window)
    # pair -- g_a over the A query, g_c over h, l and the B query in G1 -- and the same over Fq2 for g_b in G2 (three Fq products per
    # Fq2 product); the seven Fr transforms and the sorts come on top, so the fraction is a lower bound of the arithmetic done.
    cn = "bn254" if curve == "BN128" else "bls12_381"
    na, nbq = GB.density(rb, ni, n_wires)
    n_win = 16
    pts_g1 = na + ((1 << log_rows) - 1) + (n_wires - ni) + nbq
    products = 11 * n_win * pts_g1 + 3 * 11 * n_win * nbq
    peak = VALU_MAD_PER_S / FQ_MADS[cn]
    res["roofline"] = {"bound": "int-alu", "kernel": "msm_accumulate_kernel over the key's window tables (G1 and G2)", "achieved": round(products / min(ts) / 1e9, 1),
                       "peak": round(peak / 1e9, 1), "unit": "G Fq products/s", "frac": round(products / min(ts) / peak, 4),
                       "model": "%d Fq products (11 per table addition, %d G1 points and %d G2 points x %d windows; an Fq2 product counted as 3), %d v_mad_u64_u32 per product"
                                % (products, pts_g1, nbq, n_win, FQ_MADS[cn])}
    if cpu_baseline:
        import oracle_lib
        sys.path.insert(0, str(ROOT / "oracle"))
        import groth16 as G
        orc = oracle_lib.load()
        g = G.Groth16Oracle(orc, GB.NAME[curve])
        lg = min(log_rows, 18); n = 1 << lg
        rng = np.random.default_rng(5)
        mk = lambda: np.concatenate([rng.integers(0, 2**64, size=(n, 3), dtype=np.uint64), rng.integers(0, 2**60, size=(n, 1), dtype=np.uint64)], axis=1)
        a, b, c = mk(), mk(), mk()
        t0 = time.perf_counter(); hq = g.quotient(a, b, c); cpu_s = time.perf_counter() - t0
        da, db, dc = (zk.DevArray.from_host(v.reshape(-1)) for v in (a, b, c))
        dev.fr_quotient(da, db, dc, curve)
        assert np.array_equal(da.to_host().reshape(-1, 4), hq), "GPU quotient != oracle"
        res["cpu_baseline"] = {"value": round(n / cpu_s / 1e6, 4), "unit": "Mrows/s", "cores": orc.threads(), "kind": "port",
                               "sample": "quotient only (7 transforms + pointwise) of 2^%d rows, oracle/groth16_impl.h, %.2f s; the sums' CPU rate is msm_g1_*'s cpu_baseline" % (lg, cpu_s)}
    S.free()
    return res


def msm_leg(zk, logn, cpu_baseline, curve="bn254"):
    """Second component of BASELINE's metric, "BN254 G1 MSM Mpts/s" (config 4): n = 2^22 uniform scalars
    below r, bases [k_i]G generated on the device, everything resident in HBM when the clock starts;
    the result is checked against the closed form [sum s_i k_i mod r]G by the CPU oracle."""
    n = 1 << logn
    rng = np.random.default_rng(0x4D534D)
    k = rng.integers(1, 2**64, size=n, dtype=np.uint64)
    scal = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    scal[:, 3] &= np.uint64((1 << 60) - 1)                              # < 2^252 < r
    nl = {"bn254": 4, "bls12_381": 6}[curve]
    d_bases = zk.g1_mul_generator(zk.DevArray.from_host(k), curve)
    d_scal = zk.DevArray.from_host(scal.reshape(-1))
    zk.msm_g1_dev(d_bases, d_scal, n, curve)                            # warm the pool
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = zk.msm_g1_dev(d_bases, d_scal, n, curve)
        zk.lib().zk_dev_sync()
        times.append(time.perf_counter() - t0)
    res = {"workload": "BASELINE config 4: %s G1 Pippenger MSM, n=2^%d, c=16 (2n points x 8 windows behind the curve's endomorphism), HBM-resident" % (curve, logn),
           "value": round(n / min(times) / 1e6, 2), "unit": "Mpts/s", "ms": round(min(times) * 1e3, 2)}
    # integer-ALU roofline: 16 n mixed additions into buckets (16 windows x n points, or 8 x 2n behind the
    # endomorphism) + 16 x 2^17 in the bucket reduction (an upper figure with 8 windows), PADD_PRODUCTS Fq products each, against
    # the multiply-add issue ceiling
    products = (16 * n + 16 * (1 << 17)) * PADD_PRODUCTS
    peak = VALU_MAD_PER_S / FQ_MADS[curve]
    res["roofline"] = {"bound": "int-alu", "kernel": "msm_accumulate_kernel (+ sort, bucket reduction)", "achieved": round(products / min(times) / 1e9, 1),
                       "peak": round(peak / 1e9, 1), "unit": "G Fq products/s", "frac": round(products / min(times) / peak, 4),
                       "model": "%d Fq products per sum, %d v_mad_u64_u32 per product at the measured issue rate (profiles/r02/ubench_valu.txt)"
                                % (products, FQ_MADS[curve])}
    if n < (1 << 24):                                                   # window tables index their points with 24 bits
        # the same sum over a window table built once for the (fixed) bases -- what the Groth16 prover uses for its key
        tab = zk.MsmTable(d_bases, n, curve)
        o2 = tab.msm(d_scal)
        assert np.array_equal(o2.to_host(), out.to_host()), "window-table MSM != plain MSM"
        tt = []
        for _ in range(5):
            t0 = time.perf_counter(); tab.msm(d_scal); zk.lib().zk_dev_sync(); tt.append(time.perf_counter() - t0)
        res["window_table"] = {"ms": round(min(tt) * 1e3, 2), "value": round(n / min(tt) / 1e6, 2), "unit": "Mpts/s",
                               "note": "bases expanded once to 2^(16w) P (16x memory), no doublings per sum; precomputation not timed"}
        del tab
    if cpu_baseline:
        import oracle_lib
        orc = oracle_lib.load()
        cv = orc.curve(curve)
        R = cv.r
        w = lambda x: np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64)
        s4 = scal.astype(object)
        sv = s4[:, 0] + (s4[:, 1] << 64) + (s4[:, 2] << 128) + (s4[:, 3] << 192)
        exp, _ = cv.scalar_mul(cv.generator(), w(int((sv * k.astype(object)).sum() % R)))
        assert np.array_equal(out.to_host()[:2 * nl], exp), "GPU MSM != closed form"
        m = 1 << 18                                                     # bounded CPU sample of the same inputs
        hb = np.empty(m * 2 * nl, np.uint64)
        zk._check(zk.lib().zk_dev_download(zk._ptr(hb), d_bases.ptr, m * 16 * nl))
        c_bits = 12
        chunks = max(1, orc.threads() // ((256 + c_bits - 1) // c_bits))   # (window, chunk) tasks over every host thread
        t0 = time.perf_counter()
        got, _ = cv.msm_par(hb, scal[:m].reshape(-1), c_bits, chunks)
        cpu_s = time.perf_counter() - t0
        d_small = zk.msm_g1_dev(d_bases, d_scal, m, curve)
        assert np.array_equal(d_small.to_host()[:2 * nl], got), "GPU MSM != CPU oracle on the sample"
        res["cpu_baseline"] = {"value": round(m / cpu_s / 1e6, 4), "unit": "Mpts/s", "cores": orc.threads(), "kind": "port",
                               "sample": "first 2^18 points of the same input, oracle/ec_impl.h Pippenger c=%d, %d windows x %d chunks over "
                                         "all host threads, %.2f s" % (c_bits, (256 + c_bits - 1) // c_bits, chunks, cpu_s)}
    return res


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def headline(out):
    """BASELINE's whole metric in one small object -- NTT GElem/s + BN254 MSM Mpts/s + prove ms at 2^24 rows -- and the numbers the verdicts
    track beside it.  It is the LAST key of the one stdout line, so that a 2 000-character tail of the line always contains it."""
    agg = out.get("aggregation") or {}
    return {
        "ntt_gelems": out.get("value"), "ntt_pass_hbm_frac": _get(out, "roofline", "pass_hbm_frac"), "ntt_frac": _get(out, "roofline", "frac"),
        "msm_bn254_mpts": _get(out, "msm_g1_bn254", "value"), "msm_bn254_ms": _get(out, "msm_g1_bn254", "ms"), "msm_bn254_frac": _get(out, "msm_g1_bn254", "roofline", "frac"),
        "msm_bls12381_mpts": _get(out, "msm_g1_bls12_381", "value"),
        "prove_2p24_ms": _get(out, "stark_prove", "ms"), "prove_2p24_first_ms": (_get(out, "stark_prove", "ms_runs") or [None])[0],
        "prove_2p24_golden_sha_ok": _get(out, "stark_prove", "golden", "zkin_sha256_matches_golden"), "prove_2p24_frac": _get(out, "stark_prove", "roofline", "frac"),
        "prove_one_shot_s": _get(out, "stark_prove", "one_shot", "s"), "prove_cfg3_ms": _get(out, "stark_prove_cfg3", "ms"),
        "poseidon_gperm": _get(out, "poseidon_merkle_gl", "value"), "poseidon_tree_ms": _get(out, "poseidon_merkle_gl", "ms"), "poseidon_frac": _get(out, "poseidon_merkle_gl", "roofline", "frac"),
        "groth16_bn128_ms": _get(out, "groth16_prove_bn128", "ms"), "groth16_bls12381_ms": _get(out, "groth16_prove_bls12381", "ms"),
        "merkle_bn128_ref_ms": _get(out, "merkle_bn128_ref_shape", "ms"),
        "agg_tasks_per_s": agg.get("tasks_per_s"), "agg_weak_tasks_per_s": _get(agg, "weak", "tasks_per_s"), "agg_end_to_end_s": agg.get("end_to_end_s"),
        "agg_task_latency_s": agg.get("task_latency_s"), "agg_scaling_ceiling": agg.get("scaling_ceiling"),
        "final_stark_ms": _get(agg, "final_wrap", "final_stark_bls12381_ms"),
        "ranks_seen": out.get("ranks_seen"), "built": _get(out, "built", "objects_recompiled"),
    }


def build_info():
    """what __graft_entry__.build() left behind (eigen-zkvm_amd/build_info.json) + whether the library loaded now is that very file"""
    import hashlib
    lib_so = ROOT / "eigen-zkvm_amd" / "libzkgpu.so"
    info = {}
    try:
        info = json.loads((ROOT / "eigen-zkvm_amd" / "build_info.json").read_text())
        info.pop("recompiled", None)
    except Exception:
        info = {"objects_recompiled": None, "note": "no build_info.json: build() has not run in this tree"}
    try:
        info["loaded_lib_sha16"] = hashlib.sha256(lib_so.read_bytes()).hexdigest()[:16]
        info["loaded_is_built"] = info.get("libzkgpu_sha16") == info["loaded_lib_sha16"]
    except Exception:
        pass
    return info


def emit(out):
    """Rank 0's ONE stdout line.  The per-leg detail (12 KB) goes to gpurun_out/bench_detail.json and, as one line, to stderr; the stdout line
    keeps the contract's keys, `roofline`, `cpu_baseline`, a one-number summary per leg and ends with `headline` (tests/test_bench_line.py
    asserts the tail a driver keeps always contains it)."""
    out["built"] = build_info()
    out["headline"] = headline(out)
    detail = json.dumps(out)
    try:
        d = ROOT / "gpurun_out"; d.mkdir(exist_ok=True)
        (d / "bench_detail.json").write_text(detail + "\n")
    except Exception as e:
        print("bench: could not write gpurun_out/bench_detail.json: %s" % e, file=sys.stderr, flush=True)
    print("bench_detail: " + detail, file=sys.stderr, flush=True)
    print(json.dumps(compact_line(out)), flush=True)


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "ranks_seen", "collective_backend", "steps", "warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data")


def compact_line(out):
    line = {k: out[k] for k in CONTRACT_KEYS if k in out}
    line["config"] = {k: v for k, v in (out.get("config") or {}).items()}
    rf = dict(out.get("roofline") or {})
    rf.pop("traffic_source", None); rf.pop("algorithmic_bytes_per_launch", None)
    line["roofline"] = rf
    cb = dict(out.get("cpu_baseline") or {})
    if isinstance(cb.get("sample"), str) and len(cb["sample"]) > 120:
        cb["sample"] = cb["sample"][:117] + "..."
    if cb:
        line["cpu_baseline"] = cb
    legs = {}
    for k, v in out.items():                                             # one number per extra leg; the whole of it is in the detail file
        if isinstance(v, dict) and k not in ("config", "roofline", "cpu_baseline", "headline", "built"):
            legs[k] = "error" if "error" in v else {kk: v[kk] for kk in ("ms", "value", "unit", "tasks_per_s", "end_to_end_s") if kk in v}
    line["legs"] = legs
    line["detail"] = "gpurun_out/bench_detail.json (also one line on stderr, prefixed bench_detail:)"
    line["built"] = out.get("built")
    line["headline"] = out["headline"]                                   # LAST: the tail of the line is the whole metric
    return line


def dry_run(args, rank, world):
    """--dry-run: everything of main() that is NOT GPU work -- the rank environment the launcher made, the process group (gloo), the
    product's RootExchange, the max-over-ranks clock, rank 0's one line -- so that `bench.py --gpus N` can be driven on a box without GPUs
    (tests/test_dist_gloo.py).  The line says so (`dry_run`, value null): it is not a measurement."""
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    ex = _agg().RootExchange(dist, torch.device("cpu"))
    ex.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    (wall,) = ex.max([time.perf_counter() - t0])
    ranks_seen = len({r[0] for r in ex.gather([rank])})
    if rank == 0:
        print(json.dumps({"metric": "Goldilocks NTT GElems/s + BN254 G1 MSM Mpts/s; starky prove ms at 2^24 rows", "value": None, "dry_run": True,
                          "n_gpus": world, "ranks_seen": ranks_seen, "collective_backend": "none (1 rank)" if dist is None else "gloo",
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / max(1, args.steps), 4)}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nbits", type=int, default=NBITS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prove", action="store_true", help="skip the stark_prove leg")
    ap.add_argument("--prove-nbits", type=int, default=24)
    ap.add_argument("--no-one-shot", action="store_true", help="skip the fresh-process stark_prove measurement of the prove leg")
    ap.add_argument("--no-agg", action="store_true", help="skip the aggregation leg (BASELINE config 5)")
    ap.add_argument("--no-bn128", action="store_true", help="skip the BN128 Merkle leg")
    ap.add_argument("--no-poseidon", action="store_true", help="skip the Poseidon-GL Merkle leg")
    ap.add_argument("--no-msm", action="store_true", help="skip the BN254 MSM leg")
    ap.add_argument("--msm-logn", type=int, default=22)
    ap.add_argument("--no-groth16", action="store_true", help="skip the Groth16 leg")
    ap.add_argument("--groth16-log-rows", type=int, default=20)
    ap.add_argument("--dry-run", action="store_true", help="launcher / rank / collective control flow only, no GPU work (CPU tests of --gpus N)")
    args = ap.parse_args()

    # --gpus N without a launcher in front (no RANK / WORLD_SIZE): this process becomes the launcher -- N fresh rank processes of
    # this very command, one per GPU (test/stark_aggregation.sh:70-73 loops over child provers the same way).  Nothing has touched
    # a GPU yet: torch and libzkgpu are imported below, in the children.
    from eigen_zkvm_amd import launcher
    if args.gpus > 1 and not launcher.under_launcher():
        sys.exit(launcher.spawn_ranks([str(pathlib.Path(__file__).resolve())] + sys.argv[1:], args.gpus))

    import numpy as np
    import torch
    import eigen_zkvm_amd, oracle_lib
    zk = eigen_zkvm_amd                                               # the product: ctypes over libzkgpu's C ABI

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench: --gpus %d but the launcher started %d rank(s); the line reports the ranks that exist" % (args.gpus, world), file=sys.stderr, flush=True)
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # ZK_BENCH_SHARED_GPU=1: every rank on GPU 0, collectives over gloo -- the N > 1 control flow of this file on a one-GPU box
    # (tests/test_gpu_aggregation.py); RCCL refuses two ranks on one device.  Never set by the driver: its ranks own a GPU each.
    shared_gpu = os.environ.get("ZK_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    zk.init(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    lib = zk.lib()
    nbits, n = args.nbits, 1 << args.nbits
    x_host = oracle_lib.splitmix64_stream(SEED + rank, n)            # synthetic input (generator only)
    dev = torch.device("cuda", local_rank)
    x = torch.from_numpy(x_host.view(np.int64)).to(dev)              # resident in HBM
    X = torch.empty_like(x); y = torch.empty_like(x); tmp = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    vp = C.c_void_p

    def step():
        rc = lib.zk_gl_ntt_dev(vp(x.data_ptr()), vp(X.data_ptr()), vp(tmp.data_ptr()), 1, nbits, 0, vp(stream))
        rc |= lib.zk_gl_ntt_dev(vp(X.data_ptr()), vp(y.data_ptr()), vp(tmp.data_ptr()), 1, nbits, 1, vp(stream))
        if rc:
            raise RuntimeError(lib.zk_last_error().decode())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if not torch.equal(x, y):                                         # inverse(forward(x)) == x, bit exact
        raise SystemExit("bench: NTT round trip is not bit-exact")

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)                                     # HIP events on the launch stream
    ex = _agg().RootExchange(dist, torch.device("cpu") if shared_gpu else dev)   # the collectives of the path live in the product
    wall, dev_ms = ex.max([wall, dev_ms])
    ranks_seen = len({r[0] for r in ex.gather([rank])})               # an all-gather over the communicator itself (RCCL unless shared_gpu), not an env read

    agg = None
    if not args.no_agg and not args.no_prove:                          # every rank takes part (N = 1: all 8 tasks on this GPU)
        try:
            n_workers = int(os.environ.get("ZK_BENCH_WORKERS", max(1, min(4, (8 + world - 1) // world))))   # no more provers than a rank has tasks
            sys.path.insert(0, str(ROOT / "tools"))
            import aggregation_workload as AW
            agg = aggregation_leg(AW.pool(zk, workers=n_workers), ex, make_wrap=lambda: FinalWrap(zk))
        except Exception as e:                                         # at N = 1 the bench line survives a failing extra leg
            if dist is not None:                                       # (with several ranks the others wait in a collective: fail loudly)
                raise
            agg = {"error": "%s: %s" % (type(e).__name__, e)}
        if dist is not None:
            dist.barrier()

    if rank == 0:
        passes = lib.zk_gl_ntt_passes(nbits)
        launches = 2 * passes * args.steps
        launch_us = dev_ms * 1e3 / launches
        alg_bytes_per_launch = 16.0 * n / passes                       # 16 B/element/transform over its passes
        achieved = alg_bytes_per_launch / (launch_us * 1e-6) / 1e9
        pass_gbs = 16.0 * n / (launch_us * 1e-6) / 1e9
        value = world * 2.0 * n * args.steps / wall / 1e9
        traffic, traffic_src = pmc_traffic(nbits)
        out = {
            "metric": "Goldilocks NTT GElems/s + BN254 G1 MSM Mpts/s; starky prove ms at 2^24 rows",
            "value": round(value, 3), "unit": "GElem/s (Goldilocks NTT, 2^%d, fwd+inv)" % nbits,
            "n_gpus": world, "ranks_seen": ranks_seen, "collective_backend": "none (1 rank)" if dist is None else ("gloo (ranks share GPU 0: test switch)" if shared_gpu else "nccl (RCCL)"),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks, canonical)",
            "data": "synthetic (splitmix64 mod p)",
            "config": {"workload": "BASELINE config 2: 2^%d-point Goldilocks forward+inverse NTT, 1 column, "
                                   "HBM-resident" % nbits, "nbits": nbits, "n_pols": 1,
                       "passes_per_transform": passes, "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "ntt_pass_kernel", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "launch_us": round(launch_us, 2),
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "pass_bytes_gbs": round(pass_gbs, 1), "pass_hbm_frac": round(pass_gbs / HBM_PEAK_GBS, 4)},
        }
        def leg(name, fn, *a):                                         # an extra leg that fails is reported, the line is not lost
            try:
                out[name] = fn(*a)
            except Exception as e:
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                print("bench: leg %s FAILED: %s: %s" % (name, type(e).__name__, e), file=sys.stderr, flush=True)
        if not args.no_msm and world == 1:
            leg("msm_g1_bn254", msm_leg, zk, args.msm_logn, not args.no_cpu_baseline)
            leg("msm_g1_bls12_381", msm_leg, zk, args.msm_logn, not args.no_cpu_baseline, "bls12_381")
        if not args.no_poseidon and world == 1:
            leg("poseidon_merkle_gl", poseidon_leg, zk, 22, 19, not args.no_cpu_baseline)
            leg("poseidon_merkle_gl_ref_shape", poseidon_leg, zk, 24, 10, not args.no_cpu_baseline)   # starky/benches/merklehash.rs:26-27: 2^24 x 10, the one GL shape the reference benches
        if not args.no_bn128 and world == 1:
            leg("merkle_bn128", bn128_merkle_leg, zk, 20, 12, not args.no_cpu_baseline)
            leg("merkle_bn128_ref_shape", bn128_merkle_leg, zk, 24, 10, not args.no_cpu_baseline)   # starky/README.md:55: 2^24 x 10 with the BN128 hash, 11.04 s on the reference's CPU (context, other hardware)
        if not args.no_groth16 and world == 1:
            leg("groth16_prove_bn128", groth16_leg, zk, "BN128", args.groth16_log_rows, not args.no_cpu_baseline)
            leg("groth16_prove_bls12381", groth16_leg, zk, "BLS12381", args.groth16_log_rows, False)
        if not args.no_prove and world == 1:
            leg("stark_prove", prove_leg, zk, args.prove_nbits, True, not args.no_cpu_baseline, True, not args.no_one_shot)
            leg("stark_prove_cfg3", prove_leg, zk, 20, True, False, False)     # BASELINE config 3 itself: 2^20 rows, steps 21/15/11/7/4
        if agg is not None:
            out["aggregation"] = agg
        if not args.no_cpu_baseline and world == 1:                   # a reported baseline of the N = 1 line only
            orc = oracle_lib.load()
            orc.ntt_blocked(x_host[:1 << 16], 16, False)                 # thread pool warm
            t0 = time.perf_counter()
            Xc = orc.ntt_blocked(x_host, nbits, False)
            yc = orc.ntt_blocked(Xc, nbits, True)
            cpu_s = time.perf_counter() - t0
            assert np.array_equal(yc, x_host)
            assert np.array_equal(Xc, X.cpu().numpy().view(np.uint64)), "GPU forward NTT != CPU oracle"
            out["cpu_baseline"] = {"value": round(2.0 * n / cpu_s / 1e9, 5), "unit": "GElem/s", "cores": orc.threads(),
                                   "kind": "port", "sample": "1 step (fwd+inv) of the same 2^%d column, oracle/oracle.c orc_ntt_blocked "
                                   "(in-cache row transforms + transposes over all host threads, the structure of fft_p.rs:174-239), "
                                   "%.2f s" % (nbits, cpu_s)}
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
