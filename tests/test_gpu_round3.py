"""Round-3 additions to the boundary, each against the oracle or a host restatement:
stage timers and setup timing (the reference's `#[time_profiler]` spans, stark_gen.rs:192,624,709,734,785, fri.rs:83,
stark_setup.rs:26) and zk_dev_fill_splitmix."""
import json, os, pathlib, subprocess, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
P = 0xFFFFFFFF00000001


def test_fill_splitmix_matches_the_host_formula(zk):
    zk.init(0)
    n, seed = 1000, 0xABCDEF
    d = zk.DevArray(n)
    assert zk.lib().zk_dev_fill_splitmix(d.ptr, n, seed, None) == 0
    got = d.to_host()
    M = (1 << 64) - 1
    for i in (0, 1, 2, 499, 999):
        z = (seed + i + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M; z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M; z ^= z >> 31
        assert int(got[i]) == (z - P if z >= P else z)
    assert int(got.max()) < P
    assert zk.lib().zk_dev_fill_splitmix(None, 4, 0, None) != 0 and b"null buffer" in zk.lib().zk_last_error()


def test_stage_timers_and_setup_timing(zk, orc):
    """ZK_STARK_TIMING switches the timers on per proof; the proof itself is the proof made without them"""
    import importlib
    import aggregation_workload as AW, poseidong as PG
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    zk.init(0)
    nbits = 12
    ss = PG.stark_struct(nbits)
    setup = stark.NativeStarkSetup(PG.consts(nbits), json.dumps(PG.program(nbits)), json.dumps(ss))
    st = setup.setup_timing()
    assert set(st) >= {"json_parse_ms", "const_lde_merkle_ms", "programs_ms", "hiprtc_compiled", "code_cache_disk_hits", "code_cache_mem_hits", "total_ms"}
    assert st["hiprtc_compiled"] + st["code_cache_disk_hits"] + st["code_cache_mem_hits"] >= 3       # step 3, 42ns, 52ns at least
    # round 6: the step programs compile BESIDE the constants' upload / extension / tree (they start first), so the parts overlap:
    # total = parsing + the programs' span (+ what the pool's pre-sizing still needed after them: 0 for circuits this small), and that
    # span = the constants + whatever the compilers still needed afterwards
    assert abs(st["json_parse_ms"] + st["programs_ms"] + st["pool_prewarm_wait_ms"] - st["total_ms"]) < 1.0
    assert abs(st["const_lde_merkle_ms"] + st["programs_wait_after_constants_ms"] - st["programs_ms"]) < 1.0
    cm = zk.DevArray.from_host(PG.trace(nbits, None, PG.FIRST_ZERO, seed=3))
    old = os.environ.pop("ZK_STARK_TIMING", None)
    try:
        plain = setup.gen(cm)
        assert setup.last_timing() == {}
        os.environ["ZK_STARK_TIMING"] = "quiet"
        timed = setup.gen(cm)
        t = setup.last_timing()
    finally:
        if old is None: os.environ.pop("ZK_STARK_TIMING", None)
        else: os.environ["ZK_STARK_TIMING"] = old
    assert timed == plain
    for k in ("extend", "merkelize", "calculate_exps_parallel", "fri_prove", "evals", "transcript", "openings_readback", "total_gpu_ms", "wall_ms", "call_ms", "zkin_bytes"):
        assert k in t, k
    stages = sum(v for k, v in t.items() if k not in ("nBits", "total_gpu_ms", "wall_ms", "host_json_ms", "host_release_ms", "self_check_ms", "host_copy_ms", "call_ms", "zkin_bytes"))
    assert abs(stages - t["total_gpu_ms"]) < 0.05 * t["total_gpu_ms"] + 0.05 and t["nBits"] == nbits
    second = stark.NativeStarkSetup(PG.consts(nbits), json.dumps(PG.program(nbits)), json.dumps(ss))    # same circuit, same process: no hipRTC
    s2 = second.setup_timing()
    assert s2["hiprtc_compiled"] == 0 and s2["code_cache_mem_hits"] >= 3
    assert second.gen(cm) == plain
    second.free(); setup.free()


def test_threads_prove_on_the_device_of_zk_init(zk):
    """HIP's current device belongs to the host thread and a new thread starts on device 0: after zk_init(d) every thread that
    calls into the library is bound to d (csrc/capi.hip bind_device) -- the prover threads of rank k > 0 must not land on GPU 0.
    One GPU here: the binding is exercised (a fresh thread allocates, transforms, hashes and frees), a device that does not
    exist is refused and leaves the binding alone."""
    import threading
    zk.init(0)
    rng = np.random.default_rng(11)
    x = rng.integers(0, P, size=1 << 12, dtype=np.uint64)
    want = zk.fft(x, 1, 12)
    def root_of(v):
        t = zk.MerkleTreeGL(); t.merkelize(v, 4, 1 << 10); return [int(w) for w in t.root()]
    want_root = root_of(x)
    got = {}

    def work():
        try:
            d = zk.DevArray.from_host(x)
            got["copy"] = d.to_host()
            got["fft"] = zk.fft(x, 1, 12)
            got["root"] = root_of(x)
        except Exception as e:                                   # noqa: BLE001 -- reported by the assertion below
            got["err"] = repr(e)
    t = threading.Thread(target=work); t.start(); t.join()
    assert "err" not in got, got
    assert np.array_equal(got["copy"], x) and np.array_equal(got["fft"], want) and got["root"] == want_root
    n_dev = zk.lib().zk_device_count()
    with pytest.raises(zk.ZkError, match="no such device"):
        zk.init(n_dev + 7)
    assert np.array_equal(zk.fft(x, 1, 12), want)
