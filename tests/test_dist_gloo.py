"""The N>1 path on CPU: the product's aggregation driver (eigen-zkvm_amd/aggregation.py: shard_units, prove_tasks, join_tree,
RootExchange) and bench.py's timing wrapper around it on world_size-2 and -8 gloo processes -- unit sharding, the max-over-ranks
reduction and the 32-byte root all-gather (the only collectives on the path)."""
import os, sys, pathlib
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = pathlib.Path(__file__).resolve().parent.parent


class StubProver:
    """stands in for aggregation.ProverPool in the control-flow test: "proofs" are hashes of the task number"""
    def __init__(self): self.proved = []
    def task_inputs(self, task): return task
    def prove(self, task):
        self.proved.append(task)
        return [[(task * 1000003 + 17 * k + j) % (1 << 64) if j else (1 << 63) + task for j in range(4)] for k in range(3)]
    def join(self, a, b):
        self.joined = getattr(self, "joined", 0) + 1
        return [(3 * int(a[i]) + 5 * int(b[i]) + 7 + i) % (1 << 64) for i in range(4)]
    def sync(self): pass


def _ex(dist):
    import zkgpu_loader, importlib
    zkgpu_loader.load()
    return importlib.import_module("eigen_zkvm_amd.aggregation").RootExchange(dist, torch.device("cpu"))


def _agg_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = StubProver()
    out = bench.aggregation_leg(pr, _ex(dist), n_tasks=7)   # 7: ranks get 4 and 3 tasks
    dist.barrier()
    q.put((rank, pr.proved, out, pr.joined))
    dist.destroy_process_group()


def test_aggregation_leg_control_flow_two_ranks():
    world, port = 2, 29543
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # each rank proves its own tasks once after one warm-up of its first task, nobody else's; then its first task alone (task_latency_s),
    # then seven tasks of its OWN for the weak-scaling number (tasks 7 (rank + 1) + k: no other rank's, none of the fixed set)
    # ... and the fixed set once more in the end-to-end pass (one clock over tasks + joins)
    assert res[0][1] == [0, 0, 2, 4, 6, 0] + list(range(7, 14)) + [0, 2, 4, 6] and res[1][1] == [1, 1, 3, 5, 1] + list(range(14, 21)) + [1, 3, 5]
    assert all(r[2]["weak"]["tasks"] == 14 and r[2]["weak"]["tasks_per_s"] > 0 and r[2]["weak"]["scaling"].startswith("weak") for r in res)
    assert all(r[2]["task_latency_s"] is not None for r in res)
    # the join tree over 7 leaves: 3 + 2 + 1 joins in 3 levels, shared between the ranks, same root everywhere and equal
    # to the tree computed in one process
    st = StubProver()
    leaves = [st.prove(u)[2] for u in range(7)]
    lvl = leaves
    while len(lvl) > 1:
        nxt = [st.join(lvl[2 * j], lvl[2 * j + 1]) for j in range(len(lvl) // 2)]
        lvl = nxt + ([lvl[-1]] if len(lvl) % 2 else [])
    assert res[0][3] + res[1][3] == 2 * 6 and res[0][3] == 2 * 4          # twice: the phase clock and the end-to-end pass
    for r in res:
        jt = r[2]["join_tree"]
        assert (jt["levels"], jt["joins"], jt["chain_depth_of_the_reference"]) == (3, 6, 6) and jt["root"] == lvl[0]
        assert r[2]["end_to_end"]["root_equals_phase_run"] and r[2]["end_to_end_s"] >= r[2]["end_to_end"]["tasks_and_joins_s"] > 0
    for _, _, out, _ in res:                                 # every rank sees every task's roots after the all-gather
        assert out["tasks_gathered"] == list(range(7)) and out["distinct_roots"] == 7 and out["n_gpus"] == 2
        assert out["tasks"] == 7 and out["proofs_per_s"] == round(3 * out["tasks_per_s"], 3) or abs(out["proofs_per_s"] - 3 * out["tasks_per_s"]) < 0.01


def _agg8_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = StubProver()
    out = bench.aggregation_leg(pr, _ex(dist), n_tasks=8)
    dist.barrier()
    q.put((rank, pr.proved, out, getattr(pr, "joined", 0)))
    dist.destroy_process_group()


def test_aggregation_leg_eight_ranks_one_task_each():
    """the driver's 8-GPU shape: every rank holds one task; the joins of the tree thin out over the ranks (4 + 2 + 1), ranks
    without a join at a level pad the gather, and every rank ends with the root a single process computes"""
    world, port = 8, 29547
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agg8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # warm-up, the timed proof, the latency probe, eight tasks of its own for the weak-scaling number, the end-to-end pass: never another rank's task
    assert [r[1] for r in res] == [[k, k, k] + list(range(8 * (k + 1), 8 * (k + 2))) + [k] for k in range(8)]
    assert all(r[2]["weak"]["tasks"] == 64 for r in res)
    assert [r[3] for r in res] == [6, 4, 2, 2, 0, 0, 0, 0]                # join j of a level on rank j (phase clock + end-to-end pass)
    st = StubProver()
    lvl = [st.prove(u)[2] for u in range(8)]
    while len(lvl) > 1:
        lvl = [st.join(lvl[2 * j], lvl[2 * j + 1]) for j in range(len(lvl) // 2)]
    for r in res:
        out = r[2]
        assert out["join_tree"]["root"] == lvl[0] and (out["join_tree"]["levels"], out["join_tree"]["joins"]) == (3, 7)
        assert out["tasks_gathered"] == list(range(8)) and out["distinct_roots"] == 8 and out["n_gpus"] == 8


def test_aggregation_leg_single_rank():
    sys.path.insert(0, str(ROOT))
    import bench
    pr = StubProver()
    out = bench.aggregation_leg(pr, _ex(None))
    assert pr.proved == [0] + list(range(8)) + [0] + list(range(8)) + list(range(8)) and out["tasks_gathered"] == list(range(8)) and out["distinct_roots"] == 8
    assert out["weak"]["tasks"] == 8                                       # one rank: the weak pass proves the fixed set again
    assert (out["join_tree"]["levels"], out["join_tree"]["joins"], pr.joined) == (3, 7, 14)
    assert out["end_to_end"]["includes_final_wrap"] is False and out["end_to_end"]["root_equals_phase_run"]


def test_aggregation_leg_runs_the_wrap_inside_the_end_to_end_clock():
    sys.path.insert(0, str(ROOT))
    import bench
    calls = []
    class Wrap:
        def leg(self, root): calls.append(("leg", list(root))); return {"ms": 1.0}
        def run(self, root): calls.append(("run", list(root))); return {}, None
        def free(self): calls.append(("free",))
    pr = StubProver()
    out = bench.aggregation_leg(pr, _ex(None), make_wrap=Wrap)
    assert [c[0] for c in calls] == ["leg", "run", "free"] and calls[0][1] == calls[1][1] == out["join_tree"]["root"]
    assert out["final_wrap"] == {"ms": 1.0} and out["end_to_end"]["includes_final_wrap"] is True


def test_bench_gpus_2_without_rank_env_starts_two_ranks():
    """`python bench.py --gpus 2` with no launcher in front: the command starts its own two rank processes (eigen-zkvm_amd/launcher.py)
    BEFORE importing torch or the library; the collective (gloo here, RCCL on the GPU box) sees both, rank 0's one line reports them.
    --dry-run: no GPU in this container, so the NTT itself is skipped -- the line says so."""
    import json, subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                                          # ONE line on stdout; library chatter went to stderr
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["ranks_seen"] == 2 and b["dry_run"] is True and b["value"] is None


def test_launcher_reports_a_failing_rank():
    import subprocess
    code = ("import sys, os; sys.path.insert(0, %r); from eigen_zkvm_amd import launcher\n"
            "if launcher.under_launcher():\n    sys.exit(3 if os.environ['RANK'] == '1' else 0)\n"
            "sys.exit(launcher.spawn_ranks([__file__] if False else ['-c', open(sys.argv[1]).read(), sys.argv[1]], 2))\n" % str(ROOT))
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code.replace("\\n", "\n"))
    r = subprocess.run([sys.executable, f.name, f.name], capture_output=True, text=True, timeout=120)
    os.unlink(f.name)
    assert r.returncode == 3 and "rank 1 exited with status 3" in r.stderr, r.stderr


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import zkgpu_loader, importlib
    zkgpu_loader.load()
    A = importlib.import_module("eigen_zkvm_amd.aggregation")
    ex = A.RootExchange.from_env()                       # no GPU here: gloo
    assert (ex.rank, ex.world) == (rank, world)
    mine = A.shard_units(8, rank, world)
    wall, ms = ex.max([1.0 + rank, 10.0 - rank])
    roots = ex.gather([rank + 1, 2, 3, (1 << 62) + rank])
    # the whole driver without bench.py: prove this rank's tasks, exchange, join -- the same result on every rank
    pr = StubProver()
    res = A.aggregate(pr, [pr.task_inputs(u) for u in A.shard_units(5, rank, world)], 5, ex)
    assert sorted(res["by_task"]) == list(range(5)) and res["join_tree"]["joins"] == 4
    roots.append(res["join_tree"]["root"])
    dist.barrier()
    q.put((rank, mine, wall, ms, roots))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_reductions():
    world, port = 2, 29541
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    units = sorted(u for r in res for u in r[1])
    assert units == list(range(8))                       # every unit proved exactly once
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]
    for r in res:
        assert r[2] == 2.0 and r[3] == 10.0              # MAX over ranks
        assert r[4][:2] == [[1, 2, 3, 1 << 62], [2, 2, 3, (1 << 62) + 1]]
    assert res[0][4][2] == res[1][4][2]                  # the join tree's root


def test_single_rank_helpers_without_dist():
    import zkgpu_loader, importlib
    zkgpu_loader.load()
    A = importlib.import_module("eigen_zkvm_amd.aggregation")
    ex = A.RootExchange()
    assert A.shard_units(5, 0, 1) == [0, 1, 2, 3, 4]
    assert ex.max([3.5, 1.0]) == [3.5, 1.0]
    assert ex.gather([1, 2, 3, 4]) == [[1, 2, 3, 4]]
    assert list(A.shard_all_joins(8, 1, 2)) == [(0, 1), (0, 3), (1, 1)] and list(A.shard_all_joins(8, 0, 2)) == [(0, 0), (0, 2), (1, 0), (2, 0)]
    assert A.root1_of(b'{"rootC":["1","2","3","4"],"root1":["5","6","7","8"],"root2":"9"}') == [5, 6, 7, 8]
    assert A.root1_of(b'{"rootC":"1","root1":"5","root2":"9"}') == [5, 0, 0, 0]


def test_join_zkin_follows_zkin_join_rs():
    """starky/src/zkin_join.rs:9-57: a_* / b_* copies, `publics` = the first proof's without its last four words, `rootC` = the first's;
    keys in BTreeMap order, compact serde_json text"""
    import zkgpu_loader, importlib, json
    zkgpu_loader.load()
    A = importlib.import_module("eigen_zkvm_amd.aggregation")
    a = {"root1": ["1", "2", "3", "4"], "rootC": ["9", "9", "9", "9"], "publics": ["5", "6", "7", "c0", "c1", "c2", "c3"], "evals": [["1", "0", "0"]]}
    b = {"root1": ["11", "12", "13", "14"], "rootC": ["8", "8", "8", "8"], "publics": ["15", "c0", "c1", "c2", "c3"], "s0_vals1": [[1]]}
    j = A.join_zkin(a, json.dumps(b))
    assert j == {"a_evals": [["1", "0", "0"]], "a_publics": a["publics"], "a_root1": a["root1"], "a_rootC": a["rootC"],
                 "b_publics": b["publics"], "b_root1": b["root1"], "b_rootC": b["rootC"], "b_s0_vals1": [[1]],
                 "publics": ["5", "6", "7"], "rootC": a["rootC"]}
    assert list(j) == sorted(j)
    assert A.join_zkin({"publics": ["1", "2"]}, {})["publics"] == ["1", "2"]          # fewer than four publics: copied whole (:35-37)
    t = A.join_zkin_text(a, b)
    assert " " not in t and json.loads(t) == j and t.startswith('{"a_evals":')
    d = A.proof_digest(b'{"root1":["1"]}')
    assert len(d) == 4 and all(0 <= w < 0xFFFFFFFF00000001 for w in d) and d != A.proof_digest(b'{"root1":["2"]}')
