"""One-off fuzz of the run-time compiled constraint kernels: tests/test_program.py's random mixed-dimension programs over many more seeds
than the suite carries, and random wide-section shapes.  python tools/fuzz_programs.py FIRST LAST  (needs a GPU; ~1 s per seed)"""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "oracle")]
import numpy as np
import eigen_zkvm_amd as zk, oracle_lib, test_program as TP
orc = oracle_lib.load()
a, b = int(sys.argv[1]), int(sys.argv[2])
bad, t0 = [], time.time()
for seed in range(a, b):
    try:
        TP.test_random_programs_match_reference_interpreter.__wrapped__(zk, orc, seed) if hasattr(TP.test_random_programs_match_reference_interpreter, "__wrapped__") else TP.test_random_programs_match_reference_interpreter(zk, orc, seed)
    except AssertionError as e:
        bad.append(("random", seed)); print("MISMATCH random seed", seed, flush=True)
rng = np.random.default_rng(a)
for k in range((b - a) // 8):
    w_cm1, w_const, nbits = int(rng.integers(5, 200)), int(rng.integers(2, 40)), int(rng.integers(7, 10))
    try:
        TP.test_wide_sections_match_reference_interpreter(zk, orc, w_cm1, w_const, nbits)
    except AssertionError:
        bad.append(("wide", w_cm1, w_const, nbits)); print("MISMATCH wide", w_cm1, w_const, nbits, flush=True)
print("fuzz seeds [%d, %d): %d random programs + %d wide shapes in %.0f s, mismatches: %s" % (a, b, b - a, (b - a) // 8, time.time() - t0, bad), flush=True)
