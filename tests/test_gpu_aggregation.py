"""The product's aggregation driver (eigen-zkvm_amd/aggregation.py; test/stark_aggregation.sh:70-73, :83-156) with real proofs on the GPU:

  * `aggregate()` on one rank == the same tasks and joins done by hand, one setup and one proof at a time; every kept proof is accepted
    by zk_stark_verify, and the pool's self check (prove.rs:124-132) changes nothing;
  * `tools/zkgpu_prove.py stark_aggregate` as one process and as TWO processes (torchrun's environment, both on this box's one GPU, roots
    exchanged over gloo: RCCL refuses two ranks on one device) end with the same roots -- the whole multi-process path with real provers;
  * `tools/zkgpu_prove.py stark_verify` accepts what `stark_prove` wrote and rejects a tampered copy."""
import importlib
import json
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
D = ROOT / "tests" / "golden" / "starky_data"


def test_aggregate_on_one_rank_equals_the_work_done_by_hand(zk):
    import aggregation_workload as AW
    zk.init(0)
    A = importlib.import_module("eigen_zkvm_amd.aggregation")
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    c12 = importlib.import_module("eigen_zkvm_amd.compressor12")
    n = 3
    pool = AW.pool(zk, workers=2, keep_proofs=True, self_check=True)
    ex = A.RootExchange()
    res = A.aggregate(pool, [pool.task_inputs(u) for u in range(n)], n, ex)
    assert sorted(res["by_task"]) == [0, 1, 2] and (res["join_tree"]["levels"], res["join_tree"]["joins"]) == (2, 2)
    assert len(pool.proofs) == 3 * n + 2
    # by hand: one setup per circuit, proofs one after the other, the joins in tree order
    specs, join, circ = AW.circuits()
    sets = {k: stark.NativeStarkSetup(c, p, s) for k, (c, p, s) in specs.items()}
    D_ = zk.DevArray.from_host
    # a node = root1 of the proof + the digest of its whole zkin text (ProverPool.node_words = 8)
    node = lambda zb: [int(v) for v in json.loads(zb)["root1"]] + A.proof_digest(zb)
    for u in range(n):
        zs = [sets["fib"].gen_bytes(D_(AW.fib_trace(u))), sets["c12"].gen_bytes(D_(circ["c12"].witness(u))), sets["r1"].gen_bytes(D_(circ["r1"].witness(u)))]
        assert [node(z) for z in zs] == res["by_task"][u]
    E = c12.Compressor12Exec(join[1], join[2])
    def join_by_hand(a, b):                                                   # the join's 16 primary inputs = the two child nodes
        d_cm = E.run(D_(join[3]([int(w) for w in a] + [int(w) for w in b])), 1 << AW.STRUCTS["r2"]["nBits"])
        return node(sets["r2"].gen_bytes(d_cm))
    leaves = [res["by_task"][u][2] for u in range(n)]
    assert all(len(l) == 8 for l in leaves)
    assert join_by_hand(join_by_hand(leaves[0], leaves[1]), leaves[2]) == res["join_tree"]["root"]     # the odd one out joins a level up
    # zkit join_zkin on two kept recursive1 proofs: a_* / b_* copies, publics of the first without their last four words, its rootC
    r1s = [json.loads(z) for kind, z in pool.proofs if kind == "r1"][:2]
    j = A.join_zkin(r1s[0], r1s[1])
    assert j["a_root1"] == r1s[0]["root1"] and j["b_root1"] == r1s[1]["root1"] and j["rootC"] == r1s[0]["rootC"]
    pub = r1s[0]["publics"]
    assert j["publics"] == (pub[:-4] if len(pub) >= 4 else pub) and list(j) == sorted(j)      # zkin_join.rs:29-37
    # keep_proofs: the pool kept the join_zkin text of every join whose children it proved itself (here: both joins)
    assert len(pool.join_inputs) == 2
    first = json.loads(pool.join_inputs[0][1])
    as_ints = lambda r: [int(v) for v in (r if isinstance(r, list) else [r])]
    assert as_ints(first["a_root1"]) == res["by_task"][0][2][:len(as_ints(first["a_root1"]))]      # the first join: tasks 0 and 1 (pool.proofs is in completion order)
    assert as_ints(first["b_root1"]) == res["by_task"][1][2][:len(as_ints(first["b_root1"]))] and "b_finalPol" in first
    for kind, z in pool.proofs:                                               # every proof the pool made verifies against its circuit's setup
        assert sets[kind].verify(z) is True
    E.free(); pool.free()
    for s in sets.values():
        s.free()


_RCCL_PROBE = r'''
import importlib, sys
sys.path.insert(0, sys.argv[1])
import eigen_zkvm_amd as zk
zk.init(0)
A = importlib.import_module("eigen_zkvm_amd.aggregation")
ex = A.RcclExchange(zk)
assert "torch" not in sys.modules and (ex.rank, ex.world) == (0, 1)
words = [1, 2, 3, (1 << 63) + 5, (1 << 64) - 1, 0]
assert ex.gather(words) == [words] and ex.gather([]) == [[]]
assert ex.max([3.5, -1.25, 0.0]) == [3.5, -1.25, 0.0]
ex.barrier()
class Stub:                                                               # joins as arithmetic: the tree's control flow on this exchange
    def join(self, a, b): return [(3 * int(a[i]) + 5 * int(b[i]) + i) % (1 << 64) for i in range(4)]
    def sync(self): pass
leaves = [[u + 1, 2, 3, (1 << 62) + u] for u in range(5)]
got = A.join_tree(Stub(), leaves, ex)
assert got == A.join_tree(Stub(), leaves, A.RootExchange()) and got["joins"] == 4
ex.close()
import torch                                                              # with torch in the process the exchange refuses (two ROCm runtime copies)
try:
    A.RcclExchange(zk)
    raise SystemExit("RcclExchange must refuse a process that has torch loaded")
except RuntimeError as e:
    assert "torch" in str(e)
print("rccl exchange ok")
'''


def test_rccl_exchange_without_torch_one_rank_communicator():
    """aggregation.RcclExchange: ncclAllGather of librccl.so through ctypes on the library's device buffers -- the root exchange without
    torch, in a process of its own (this test process has torch loaded by other test modules, and the two do not mix: torch ships its own
    copies of the ROCm runtime).  What one GPU can show: a one-rank communicator (ncclCommInitRank with nranks = 1), the gather / barrier /
    MAX on it, and the join tree running on top of it."""
    r = subprocess.run([sys.executable, "-c", _RCCL_PROBE, str(ROOT)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl exchange ok" in r.stdout, r.stdout + r.stderr


def _cli(args, env=None, **kw):
    return subprocess.run([sys.executable, str(ROOT / "tools" / "zkgpu_prove.py")] + args, capture_output=True, text=True, env=env, timeout=600, **kw)


def test_cli_stark_aggregate_one_and_two_processes_agree(tmp_path):
    one = _cli(["stark_aggregate", "--num_proof", "4", "--workers", "2", "--workspace", str(tmp_path / "one")])
    assert one.returncode == 0, one.stdout + one.stderr
    a = json.load(open(tmp_path / "one" / "aggregation.json"))
    assert a["num_proof"] == 4 and a["ranks"] == 1 and a["ranks_seen"] == 1 and a["each_proof_self_checked"] and (a["join_tree"]["levels"], a["join_tree"]["joins"]) == (2, 3)
    r = _cli(["stark_aggregate", "--num_proof", "4", "--workers", "2", "--workspace", str(tmp_path / "rccl")], env=dict(os.environ, ZK_AGG_BACKEND="rccl"))
    assert r.returncode == 0, r.stdout + r.stderr                             # the torch-free exchange (a one-rank RCCL communicator here)
    c = json.load(open(tmp_path / "rccl" / "aggregation.json"))
    assert c["tasks"] == a["tasks"] and c["join_tree"]["root"] == a["join_tree"]["root"]
    procs = []
    for rank in range(2):                                                     # what torchrun --nproc-per-node 2 sets, by hand; both ranks on GPU 0
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571",
                   ZK_AGG_BACKEND="gloo", ZK_AGG_DEVICE="0")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tools" / "zkgpu_prove.py"), "stark_aggregate", "--num_proof", "4", "--workers", "2",
                                       "--workspace", str(tmp_path / "two")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    b = json.load(open(tmp_path / "two" / "aggregation.json"))
    assert b["ranks"] == 2 and b["tasks"] == a["tasks"] and b["join_tree"]["root"] == a["join_tree"]["root"]
    assert b["join_tree"]["joins"] == 3


def test_cli_stark_verify_accepts_and_rejects(tmp_path):
    ss = {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]}
    (tmp_path / "ss.json").write_text(json.dumps(ss))
    common = ["-s", str(tmp_path / "ss.json"), "-p", str(D / "fib.pil.json"), "--o", str(D / "fib.const")]
    out = _cli(["stark_prove"] + common + ["--m", str(D / "fib.cm"), "-c", str(tmp_path / "v.circom"), "--i", str(tmp_path / "zkin.json")])
    assert out.returncode == 0 and "verified and written" in out.stdout, out.stdout + out.stderr
    ok = _cli(["stark_verify"] + common + ["--i", str(tmp_path / "zkin.json")])
    assert ok.returncode == 0 and "verifies" in ok.stdout, ok.stdout + ok.stderr
    z = json.load(open(tmp_path / "zkin.json"))
    z["evals"][0][0] = str((int(z["evals"][0][0]) + 1) % 0xFFFFFFFF00000001)
    (tmp_path / "bad.json").write_text(json.dumps(z))
    bad = _cli(["stark_verify"] + common + ["--i", str(tmp_path / "bad.json")])
    assert bad.returncode != 0 and "does not verify" in (bad.stdout + bad.stderr)


def test_bench_two_ranks_control_flow_on_one_gpu(tmp_path):
    """bench.py's N > 1 path -- process group, barrier + max-over-ranks timing, the sharded aggregation leg through the product's driver,
    the rank-0 serial tail, ONE JSON line from rank 0 -- as the driver starts it (`torch.distributed.run` environment), with both ranks
    on this box's one GPU and the collectives over gloo (ZK_BENCH_SHARED_GPU=1).  Values mean nothing here (two ranks share a device);
    the shape of the line and the agreement of the ranks do."""
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29573", ZK_BENCH_SHARED_GPU="1")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--nbits", "20"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(ROOT)))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # only rank 0 prints the line (gloo prints a banner of its own)
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert list(b)[-1] == "headline" and b["headline"]["ranks_seen"] == 2 and b["headline"]["ntt_gelems"] == b["value"] and len(lines[0]) < 4000
    assert b["legs"]["aggregation"]["tasks_per_s"] > 0
    # the one stdout line is compact (round 6); the legs' detail is one line on stderr, prefixed "bench_detail: " (and gpurun_out/bench_detail.json)
    (detail,) = [l for l in outs[0][1].splitlines() if l.startswith("bench_detail: ")]
    b = json.loads(detail[len("bench_detail: "):])
    assert b["n_gpus"] == 2 and b["steps"] == 3 and b["warmup"] == 1 and b["scaling"] == "weak" and b["higher_is_better"] is True
    assert b["metric"].startswith("Goldilocks NTT GElems/s") and b["config"]["parallelism"] == "replicas x2" and b["value"] > 0
    assert "roofline" in b and b["roofline"]["bound"] == "hbm"
    a = b["aggregation"]
    assert "error" not in a and a["n_gpus"] == 2 and a["tasks"] == 8 and a["tasks_gathered"] == list(range(8)) and a["distinct_roots"] == 8
    assert (a["join_tree"]["levels"], a["join_tree"]["joins"]) == (3, 7) and len(a["join_tree"]["root"]) == 8
    assert a["final_wrap"].get("final_stark_verified") is True
    assert b["ranks_seen"] == 2 and a["end_to_end"]["includes_final_wrap"] and a["end_to_end"]["root_equals_phase_run"] and a["end_to_end_s"] > 0
    for leg in ("msm_g1_bn254", "stark_prove", "cpu_baseline"):               # N = 1 legs stay out of an N > 1 line
        assert leg not in b


def test_bench_gpus_2_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO rank environment (the driver's N = 1 command with another number): the command itself becomes
    the launcher, two fresh rank processes run, RCCL's stand-in on this one-GPU box (gloo, ZK_BENCH_SHARED_GPU=1) sees both, and the parent
    prints rank 0's one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ZK_BENCH_SHARED_GPU"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--nbits", "20", "--no-agg"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    b = json.loads(lines[0])
    assert list(b)[-1] == "headline" and b["headline"]["ranks_seen"] == 2 and b["headline"]["ntt_gelems"] == b["value"] and len(lines[0]) < 4000
    (detail,) = [l for l in r.stderr.splitlines() if "bench_detail: " in l]      # the launcher relays rank 0's stderr with a "[rank 0] " prefix
    assert json.loads(detail[detail.index("bench_detail: ") + len("bench_detail: "):])["headline"] == b["headline"]
    assert b["n_gpus"] == 2 and b["ranks_seen"] == 2 and b["value"] > 0 and b["config"]["parallelism"] == "replicas x2"


def test_cli_stark_aggregate_gpus_2_starts_its_own_ranks(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ZK_AGG_BACKEND="gloo", ZK_AGG_DEVICE="0")
    r = _cli(["stark_aggregate", "--gpus", "2", "--num_proof", "4", "--workers", "2", "--workspace", str(tmp_path / "self")], env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    a = json.load(open(tmp_path / "self" / "aggregation.json"))
    assert a["ranks"] == 2 and a["ranks_seen"] == 2 and a["join_tree"]["joins"] == 3 and len(a["join_tree"]["root"]) == 8
