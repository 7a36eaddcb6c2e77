"""STARK aggregation over the GPUs of one node (BASELINE config 5): the product's driver behind
`test/stark_aggregation.sh` -- the per-task loop (:70-73, one `recursive_proof_to_snark.sh` per proof: three STARKs in a
row) and the join phase (:83-156, `join_zkin` + `compressor12_exec` + `stark_prove` of recursive2 per join).

  * one process per GPU; task u runs on rank u mod N (`shard_units`): no collective while proving;
  * `ProverPool`: per rank, `workers` host threads, each with its own setups (constants extended and merkelized once,
    constraint kernels compiled once per process) and its own HIP stream -- the proofs of a task stay in order, tasks
    overlap (the reference runs them as parallel processes);
  * the joins form a tree (`join_tree`): level l joins neighbours pairwise, join j of a level on rank j mod N, so
    ceil(log2 n) dependent recursive2 proofs stand on the critical path where the script's chain has n - 1;
  * the ONE exchange (`RootExchange.gather`): an all-gather of the nodes each rank made (a 4-word root + a 4-word digest of
    the whole proof, `ProverPool.node_words`), once after the tasks and once per join level -- `torch.distributed` all_gather, which is `ncclAllGather` of RCCL over xGMI under backend
    "nccl" (and gloo in the CPU tests of this control flow).  Nothing else crosses ranks.

The circuits themselves are the caller's: (constants, `{"starkinfo", "program"}` JSON, StarkStruct JSON) per circuit
name, the join circuit's `.exec` text for `compressor12_exec`.  `tools/zkgpu_prove.py stark_aggregate` drives this
module from the command line; `bench.py` times it; `tests/test_dist_gloo.py` runs its control flow on 2 and 8 gloo ranks.
"""
import json
import os
import threading
import time


def shard_units(n_units, rank, world):
    """Independent proving units owned by `rank`: round-robin, so that every rank gets floor or ceil of
    n_units / world of them and no unit is proved twice (test/stark_aggregation.sh:70-73 is this loop on one host)."""
    return list(range(rank, n_units, world))


class RootExchange:
    """The collectives of the aggregation path, all of them: the all-gather of Merkle roots (4 u64 words each; every
    rank passes the same number of words), a barrier, and a MAX over ranks for timing.  `dist` = an initialised
    `torch.distributed` (backend "nccl" = RCCL on the GPU box) or None on one rank."""

    def __init__(self, dist=None, device=None):
        self.dist, self.device = dist, device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1

    @classmethod
    def from_env(cls, local_rank=None):
        """one process per GPU as torchrun starts them (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); a lone process
        gets the no-op exchange"""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        lr = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
        if os.environ.get("ZK_AGG_BACKEND") == "rccl":                       # RCCL through ctypes, no torch (RcclExchange) -- decided BEFORE torch
            import eigen_zkvm_amd as zk                                     # is imported: torch brings its own librccl / HIP runtime copies, and a
            return RcclExchange.bootstrap(zk, int(os.environ.get("RANK", "0")), world, os.environ.get("MASTER_ADDR", "127.0.0.1"),   # librccl bound to
                                          int(os.environ.get("MASTER_PORT", "29500")) + 1)                                        # those sees no device
        import torch
        if world == 1:
            return cls(None, torch.device("cuda", lr) if torch.cuda.is_available() else torch.device("cpu"))
        import torch.distributed as dist
        # "nccl" (= RCCL over xGMI) whenever the ranks have a GPU each; ZK_AGG_BACKEND=gloo for ranks that share one GPU (tests on a
        # one-GPU box: RCCL refuses two ranks on the same device) -- the 32-byte roots then travel through host memory
        backend = os.environ.get("ZK_AGG_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        on_gpu = backend == "nccl"
        if not dist.is_initialized():
            if on_gpu:
                torch.cuda.set_device(lr)
                dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
            else:
                dist.init_process_group("gloo")
        return cls(dist, torch.device("cuda", lr) if on_gpu else torch.device("cpu"))

    def gather(self, words):
        """-> [world][len(words)]: every rank's words, in rank order (u64 values travel as two's-complement int64)"""
        to_u64 = lambda v: int(v) + (1 << 64) if int(v) < 0 else int(v)
        if self.dist is None:
            return [[int(v) for v in words]]
        import torch
        to_i64 = lambda v: int(v) - (1 << 64) if int(v) >= (1 << 63) else int(v)
        t = torch.tensor([to_i64(v) for v in words], dtype=torch.int64, device=self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[to_u64(v) for v in o.tolist()] for o in out]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, values):
        """MAX-reduce a list of floats over all ranks"""
        if self.dist is None:
            return [float(v) for v in values]
        import torch
        t = torch.tensor(values, device=self.device, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t]


class RcclExchange(RootExchange):
    """The same three operations on RCCL alone, no torch: `ncclAllGather` of librccl.so through ctypes on device buffers of the library
    (C1 of SURVEY 8e taken literally: 32-byte roots over xGMI; the barrier and the MAX are all-gathers of one word).  The communicator's
    128-byte unique id goes from rank 0 to the others over a TCP socket on MASTER_ADDR : MASTER_PORT + 1 (`bootstrap`), the one thing
    that has to travel before RCCL is up.  One process per GPU; `zk.init(device)` must have bound this thread to its GPU.
    Selected by ZK_AGG_BACKEND=rccl (`RootExchange.from_env`); the default stays `torch.distributed` until this path has run on a
    multi-GPU node (a one-rank communicator on one GPU is what the tests of this build could reach: tests/test_gpu_aggregation.py)."""
    NCCL_UINT64 = 5                                                        # rccl.h ncclDataType_t

    def __init__(self, zk, rank=0, world=1, unique_id=None, lib_path=None):
        import ctypes as C
        import sys
        if "torch" in sys.modules and not (lib_path or os.environ.get("ZK_RCCL_LIB")):
            # torch ships copies of librccl / libhsa-runtime64; the system librccl beside them finds an HSA runtime that was never
            # initialised ("no ROCm-capable device is detected", measured).  This exchange is for processes without torch.
            raise RuntimeError("RcclExchange: torch is loaded in this process; use RootExchange (torch.distributed) here, or name torch's own librccl in ZK_RCCL_LIB")
        self.zk, self.rank, self.world, self.dist, self.device = zk, rank, world, None, None
        self._C = C
        L = self._L = C.CDLL(lib_path or os.environ.get("ZK_RCCL_LIB", "librccl.so"))

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        self._UniqueId = UniqueId
        L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]; L.ncclGetUniqueId.restype = C.c_int
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]; L.ncclCommInitRank.restype = C.c_int
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]; L.ncclAllGather.restype = C.c_int
        L.ncclCommDestroy.argtypes = [C.c_void_p]; L.ncclCommDestroy.restype = C.c_int
        L.ncclGetErrorString.argtypes = [C.c_int]; L.ncclGetErrorString.restype = C.c_char_p
        uid = UniqueId()
        if unique_id is None:
            assert world == 1, "RcclExchange: ranks > 0 need rank 0's unique id (RcclExchange.bootstrap)"
            self._check(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        else:
            C.memmove(C.byref(uid), unique_id, 128)
        warm = zk.DevArray(1); zk._check(zk.lib().zk_dev_sync()); warm.free()   # HIP initialises lazily: RCCL finds "no ROCm-capable device" (HSA not up) before the first allocation
        self._comm = C.c_void_p()
        self._check(L.ncclCommInitRank(C.byref(self._comm), world, uid, rank), "ncclCommInitRank")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self._L.ncclGetErrorString(rc).decode()))

    @classmethod
    def bootstrap(cls, zk, rank, world, addr, port):
        """rank 0 makes the unique id and hands it to every other rank over TCP; -> the exchange of this rank"""
        import ctypes as C, socket
        if world == 1:
            return cls(zk)
        if rank == 0:
            L = C.CDLL(os.environ.get("ZK_RCCL_LIB", "librccl.so"))
            buf = C.create_string_buffer(128)
            L.ncclGetUniqueId.argtypes = [C.c_void_p]; L.ncclGetUniqueId.restype = C.c_int
            if L.ncclGetUniqueId(buf) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
            uid = buf.raw
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port)); srv.listen(world)
            for _ in range(world - 1):
                c, _a = srv.accept(); c.sendall(uid); c.close()
            srv.close()
        else:
            uid, deadline = b"", time.time() + 120
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=5)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            while len(uid) < 128:
                chunk = c.recv(128 - len(uid))
                if not chunk:
                    raise RuntimeError("RcclExchange.bootstrap: rank 0 closed the connection early")
                uid += chunk
            c.close()
        return cls(zk, rank, world, uid)

    def gather(self, words):
        import numpy as np
        n = len(words)
        if n == 0:
            return [[] for _ in range(self.world)]
        send = self.zk.DevArray.from_host(np.array([int(v) for v in words], dtype=np.uint64))
        recv = self.zk.DevArray(n * self.world)
        self._check(self._L.ncclAllGather(send.ptr, recv.ptr, n, self.NCCL_UINT64, self._comm, None), "ncclAllGather")   # null stream
        self.zk._check(self.zk.lib().zk_dev_sync())
        out = recv.to_host()
        send.free(); recv.free()
        return [[int(v) for v in out[r * n:(r + 1) * n]] for r in range(self.world)]

    def barrier(self):
        self.gather([0])

    def max(self, values):
        import struct
        bits = [struct.unpack("<Q", struct.pack("<d", float(v)))[0] for v in values]
        rows = self.gather(bits)
        return [max(struct.unpack("<d", struct.pack("<Q", r[i]))[0] for r in rows) for i in range(len(values))]

    def close(self):
        if getattr(self, "_comm", None):
            self._L.ncclCommDestroy(self._comm); self._comm = None


def join_zkin(zkin1, zkin2):
    """`zkit join_zkin` (starky/src/zkin_join.rs:9-57; test/stark_aggregation.sh:87-90, :132-135): the input of one recursive2
    step from two proofs.  Every key k of the first proof becomes `a_k`, of the second `b_k`; `publics` = the first proof's publics
    without their last four words (the root of the constants they end with), `rootC` = the first proof's.  zkin1 / zkin2: dicts (parsed
    zkin) or their JSON text; -> dict with keys in the BTreeMap's (sorted) order, which `join_zkin_text` serialises as serde_json does."""
    a = json.loads(zkin1) if isinstance(zkin1, (bytes, str)) else zkin1
    b = json.loads(zkin2) if isinstance(zkin2, (bytes, str)) else zkin2
    out = {}
    for k in sorted(a):
        v = a[k]
        out["a_" + k] = v
        if k == "publics" and isinstance(v, list):
            out["publics"] = v[:-4] if len(v) >= 4 else v
        if k == "rootC":
            out[k] = v
    for k in sorted(b):
        out["b_" + k] = b[k]
    return {k: out[k] for k in sorted(out)}


def join_zkin_text(zkin1, zkin2):
    """the bytes `join_zkin` writes: serde_json::to_string of the BTreeMap (no spaces, keys sorted)"""
    return json.dumps(join_zkin(zkin1, zkin2), separators=(",", ":"))


def proof_digest(zkin_json):
    """4 words that stand for a whole proof where only 32 bytes may travel (the root gather): sha256 of the zkin text, as four
    little-endian u64 words reduced into the Goldilocks field (they become primary inputs of the join circuit)"""
    import hashlib
    d = hashlib.sha256(zkin_json).digest()
    return [int.from_bytes(d[8 * i: 8 * i + 8], "little") % 0xFFFFFFFF00000001 for i in range(4)]


def root1_of(zkin_json):
    """root1 of a proof's JSON text (bytes) without parsing the openings after it (serializer.rs:146-152: rootC, root1, ...)"""
    i = zkin_json.index(b'"root1":') + 8
    j = zkin_json.index(b']', i) + 1 if zkin_json[i:i + 1] == b'[' else zkin_json.index(b',', i)
    r = json.loads(zkin_json[i:j])
    return [int(v) for v in (r if isinstance(r, list) else [r, 0, 0, 0])]


class ProverPool:
    """The provers of one rank.  circuits: {name: (constants, program_json, stark_struct_json)}; a task is a list of
    (circuit name, HBM-resident trace) proved in that order on one worker's stream; join = (circuit name, .exec text,
    n_witness, witness_fn(primary16) -> host vector) describes the recursive2 step.  keep_proofs: keep every proof's
    zkin text (`proofs`) instead of only its root.  self_check: every proof is verified before it is handed out
    (prove.rs:124-132).

    What a proof hands on (a *node* of the aggregation, `node_words` = 8): root1 of its committed trace + `proof_digest` of its whole
    zkin.  A join's 16 primary inputs are the two child nodes, so a recursive2 proof is bound to both children's full proofs, and the
    8 words are all that has to cross ranks (the root gather).  **The join circuit is a stand-in**: the reference's recursive2 is the
    circom-compiled *verifier* of both children (zkin_join.rs + the witness calculator feed it the full proofs); circom is out of
    scope here, so the joined circuit has the real one's shape and size and takes the children's nodes as inputs, but it does
    not re-verify them -- the tree's root commits to the leaves, it does not attest their validity.  Validity of every proof is what
    `self_check` (the library's own stark_verify on each proof) establishes, per proof."""
    node_words = 8

    def __init__(self, zk, circuits, workers=4, join=None, keep_proofs=False, self_check=False):
        import importlib
        stark = importlib.import_module("eigen_zkvm_amd.stark")
        self.zk, self.workers, self.keep_proofs = zk, max(1, int(workers)), keep_proofs
        self.sets = [{k: stark.NativeStarkSetup(c, p, s, self_check=self_check) for k, (c, p, s) in circuits.items()} for _ in range(self.workers)]
        self.streams = [zk.Stream() for _ in range(self.workers)]
        self.sizes = {k: json.loads(s)["nBits"] for k, (_, _, s) in circuits.items()}
        self.join_kind = self.join_exec = self.join_witness = None
        if join is not None:
            c12 = importlib.import_module("eigen_zkvm_amd.compressor12")
            self.join_kind, exec_text, n_witness, self.join_witness = join
            self.join_exec = c12.Compressor12Exec(exec_text, n_witness)      # read-only handle: shared by the workers
        self.proofs = []
        self.join_inputs = []                      # keep_proofs: (node of the join's proof, zkit join_zkin text of its two children) when both children were proved on this rank
        self._by_node = {}                         # keep_proofs: node -> zkin text of the proof that made it
        self._lock = threading.Lock()
        self.reset_join_times()

    def reset_join_times(self):
        # per worker, summed over its joins: building + uploading the witness vector (host), compressor12 exec (device), the proof
        self.join_exec_s, self.join_exec_dev_s, self.join_prove_s = [0.0] * self.workers, [0.0] * self.workers, [0.0] * self.workers

    def _gen(self, kind, d_cm, worker):
        z = self.sets[worker][kind].gen_bytes(d_cm, self.streams[worker].handle)
        node = root1_of(z) + proof_digest(z)
        if self.keep_proofs:
            with self._lock:
                self.proofs.append((kind, z))
                self._by_node[tuple(node)] = z
        return node

    def prove(self, inputs, worker=0):
        """one task -> the node (root1 + digest of the whole zkin) of each of its proofs, [[8 words]] * len(inputs)"""
        return [self._gen(kind, d_cm, worker) for kind, d_cm in inputs]

    def stage_times(self, inputs, worker=0):
        """one task with the library's stage timers on (ZK_STARK_TIMING): per proof, HIP-event time on the proof's stream against
        the wall time of the call, and its largest stages under the reference's span names"""
        old = os.environ.get("ZK_STARK_TIMING")
        os.environ["ZK_STARK_TIMING"] = "quiet"
        try:
            out, st = {}, self.streams[worker].handle
            for kind, d_cm in inputs:
                t0 = time.perf_counter(); self.sets[worker][kind].gen_bytes(d_cm, st); call_ms = (time.perf_counter() - t0) * 1e3
                t = self.sets[worker][kind].last_timing()
                stages = sorted(((k, v) for k, v in t.items() if k not in ("nBits", "total_gpu_ms", "wall_ms", "call_ms", "zkin_bytes")), key=lambda kv: -kv[1])
                out[kind] = {"call_ms": round(call_ms, 2), "gpu_event_ms": t.get("total_gpu_ms"), "host_after_last_launch_ms": round(call_ms - t.get("total_gpu_ms", 0), 2),
                             "top_stages_ms": {k: round(v, 2) for k, v in stages[:4]}}
            return out
        finally:
            if old is None: os.environ.pop("ZK_STARK_TIMING", None)
            else: os.environ["ZK_STARK_TIMING"] = old

    def _spread(self, jobs, fn):
        """jobs[i] -> fn(jobs[i], worker) on worker i mod workers, the workers side by side; results in job order"""
        out, errs = [None] * len(jobs), []
        def run(w):
            try:
                for i in range(w, len(jobs), self.workers):
                    out[i] = fn(jobs[i], w)
            except BaseException as e:                                    # noqa: BLE001 -- re-raised below
                errs.append(e)
        threads = [threading.Thread(target=run, args=(w,)) for w in range(min(self.workers, len(jobs)))]
        for t in threads: t.start()
        for t in threads: t.join()
        if errs:
            raise errs[0]
        return out

    def prove_all(self, inputs_list):
        return self._spread(inputs_list, self.prove)

    def join(self, node_a, node_b, worker=0):
        """One recursive2 step (test/stark_aggregation.sh:83-128: join_zkin + compressor12_exec + stark_prove) on the stand-in
        circuit (class text): the joined circuit's 16 primary inputs are the two child nodes (root1 + digest of the whole child
        proof each); its trace is born in HBM by compressor12 exec on the device.  -> the join's own node."""
        if self.join_exec is None:
            raise ValueError("ProverPool: no join circuit was given")
        primary = ([int(w) for w in node_a] + [int(w) for w in node_b] + [0] * 16)[:16]
        st = self.streams[worker].handle
        t0 = time.perf_counter()
        d_w = self.zk.DevArray.from_host(self.join_witness(primary))         # what the circom witness calculator would hand over
        t1 = time.perf_counter()
        d_cm = self.join_exec.run(d_w, 1 << self.sizes[self.join_kind], st)   # PlonkAdds + s_map gather (compressor12_exec.rs:58-88)
        t2 = time.perf_counter()
        root = self._gen(self.join_kind, d_cm, worker)
        self.join_exec_s[worker] += t1 - t0; self.join_exec_dev_s[worker] += t2 - t1; self.join_prove_s[worker] += time.perf_counter() - t2
        if self.keep_proofs:                                                 # `zkit join_zkin` of the two children (stark_aggregation.sh:87-90), when this rank holds both
            za, zb = self._by_node.get(tuple(int(w) for w in node_a)), self._by_node.get(tuple(int(w) for w in node_b))
            if za is not None and zb is not None:
                with self._lock:
                    self.join_inputs.append((root, join_zkin_text(za, zb)))
        return root

    def warm_join(self):
        """one join per worker: code objects of the join setups loaded, pool blocks of a join in place"""
        kept, self.keep_proofs = self.keep_proofs, False
        self.join_all([([1, 2, 3, 4, 5, 6, 7, 8], [9, 10, 11, 12, 13, 14, 15, 16])] * self.workers)
        self.keep_proofs = kept
        self.sync()
        self.reset_join_times()

    def join_all(self, pairs):
        """the joins of one tree level: independent of each other"""
        return self._spread(pairs, lambda ab, w: self.join(ab[0], ab[1], w))

    def sync(self):
        self.zk.lib().zk_dev_sync()

    def free(self):
        for s in self.sets:
            for v in s.values():
                v.free()
        self.sets = []
        if self.join_exec is not None:
            self.join_exec.free(); self.join_exec = None
        for st in self.streams:
            st.free()
        self.streams = []


def prove_tasks(pool, inputs, n_tasks, exchange, proofs_per_task=3):
    """The sharded part: this rank's tasks (inputs[i] belongs to task shard_units(n_tasks, rank, world)[i]) through the pool,
    then the all-gather of every task's roots.  -> (roots of this rank's tasks, {task: [root per proof]})"""
    rank, world = exchange.rank, exchange.world
    units = shard_units(n_tasks, rank, world)
    assert len(inputs) == len(units)
    roots = pool.prove_all(inputs) if hasattr(pool, "prove_all") else [pool.prove(i) for i in inputs]
    pool.sync()
    return roots, gather_task_roots(roots, n_tasks, exchange, proofs_per_task, getattr(pool, "node_words", 4))


def gather_task_roots(roots, n_tasks, exchange, proofs_per_task=3, node_words=4):
    """roots: this rank's [[node per proof] per task] -> {task: [node per proof]} on every rank (one all-gather; a task of
    recursive_proof_to_snark.sh is three proofs; a node is `node_words` words: a 4-word root, or root + proof digest)"""
    world = exchange.world
    per_rank = (n_tasks + world - 1) // world
    per_task, nw = proofs_per_task, node_words
    assert all(len(r) == per_task and all(len(x) == nw for x in r) for r in roots)
    flat = [w for task_roots in roots for r in task_roots for w in r]
    flat += [0] * (per_rank * nw * per_task - len(flat))                    # ranks with one task fewer pad their slot
    gathered = exchange.gather(flat)
    by_task = {}
    for rk, words in enumerate(gathered):
        for j, u in enumerate(shard_units(n_tasks, rk, world)):
            by_task[u] = [words[nw * per_task * j + nw * k: nw * per_task * j + nw * k + nw] for k in range(per_task)]
    return by_task


def shard_all_joins(n_leaves, rank, world):
    """(level, join) pairs of the join tree that fall on this rank"""
    n, level = n_leaves, 0
    while n > 1:
        for j in shard_units(n // 2, rank, world):
            yield level, j
        n, level = n // 2 + n % 2, level + 1


def join_tree(pool, leaves, exchange):
    """The join phase as a tree instead of the reference's chain (test/stark_aggregation.sh:83-156 joins proof k + 1 into the
    running aggregate: NUM_PROOF - 1 sequential recursive2 proofs).  Level l joins neighbours pairwise, join j on rank
    j mod world, one all-gather of the new roots per level.  Every rank ends with the same root.
    -> {"levels", "joins", "chain_depth_of_the_reference", "root"}"""
    rank, world = exchange.rank, exchange.world
    nodes, levels, joins = [list(r) for r in leaves], 0, 0
    nw = len(nodes[0]) if nodes else 4
    while len(nodes) > 1:
        n_join = len(nodes) // 2
        mine = shard_units(n_join, rank, world)
        pairs = [(nodes[2 * j], nodes[2 * j + 1]) for j in mine]
        made = pool.join_all(pairs) if hasattr(pool, "join_all") else [pool.join(a, b) for a, b in pairs]
        per_rank = (n_join + world - 1) // world
        flat = [w for r in made for w in r] + [0] * (nw * (per_rank - len(made)))
        gathered = exchange.gather(flat)
        nxt = [None] * n_join
        for rk, words in enumerate(gathered):
            for k, j in enumerate(shard_units(n_join, rk, world)):
                nxt[j] = words[nw * k: nw * k + nw]
        if len(nodes) % 2:
            nxt.append(nodes[-1])                                          # odd one out moves up unjoined
        nodes, levels, joins = nxt, levels + 1, joins + n_join
    pool.sync()
    return {"levels": levels, "joins": joins, "chain_depth_of_the_reference": max(0, len(leaves) - 1), "root": [int(w) for w in nodes[0]]}


def aggregate(pool, inputs, n_tasks, exchange, leaf=-1, proofs_per_task=3):
    """test/stark_aggregation.sh without its final stage: prove this rank's tasks, exchange the roots, join the last
    proof's root of every task (`leaf` = which of a task's proofs feeds the joins) as a tree.
    -> {"by_task": {task: roots}, "join_tree": {...}} -- the same on every rank"""
    _, by_task = prove_tasks(pool, inputs, n_tasks, exchange, proofs_per_task)
    jt = join_tree(pool, [by_task[u][leaf] for u in sorted(by_task)], exchange)
    return {"by_task": by_task, "join_tree": jt}
