"""bench.py's ONE stdout line (round-5 verdict, "make the driver see the whole metric"): whatever the legs produce, the line a driver keeps
the last 2 000 characters of must still contain BASELINE's whole metric -- NTT GElem/s, BN254 MSM Mpts/s, prove ms at 2^24 rows -- i.e. the
`headline` object is the last key and short; `roofline` and `cpu_baseline` stay top-level.  CPU only: the line is built from a recorded detail
object (profiles/r05/bench_r05.json) and from a worst case with every leg failed."""
import importlib.util
import json
import pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _line(b, detail):
    detail = dict(detail)
    detail["built"] = b.build_info()
    detail["headline"] = b.headline(detail)
    return json.dumps(b.compact_line(detail)), detail


def test_headline_is_the_tail_of_the_line():
    b = _bench()
    detail = json.loads((ROOT / "profiles" / "r05" / "bench_r05.json").read_text())
    line, full = _line(b, detail)
    assert "\n" not in line
    parsed = json.loads(line)
    assert list(parsed)[-1] == "headline"
    assert len(line) - line.rfind('"headline"') < 1500                       # the verdict's bar
    assert len(line) < 4000                                                  # the whole line is small now; the detail lives in a file
    h = parsed["headline"]
    for k in ("ntt_gelems", "ntt_pass_hbm_frac", "msm_bn254_mpts", "msm_bn254_frac", "msm_bls12381_mpts", "prove_2p24_ms", "prove_2p24_golden_sha_ok",
              "prove_one_shot_s", "agg_tasks_per_s", "agg_end_to_end_s", "agg_task_latency_s", "ranks_seen", "built"):
        assert k in h
    assert h["ntt_gelems"] == detail["value"] and h["msm_bn254_mpts"] == detail["msm_g1_bn254"]["value"] and h["prove_2p24_ms"] == detail["stark_prove"]["ms"]
    assert h["final_stark_ms"] == detail["aggregation"]["final_wrap"]["final_stark_bls12381_ms"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in parsed, k                                                # the contract's keys stay top-level
    assert parsed["roofline"]["frac"] == detail["roofline"]["frac"] and parsed["cpu_baseline"]["kind"] == "port"


def test_failed_legs_do_not_break_the_line():
    b = _bench()
    detail = json.loads((ROOT / "profiles" / "r05" / "bench_r05.json").read_text())
    for k in ("msm_g1_bn254", "stark_prove", "aggregation"):
        detail[k] = {"error": "RuntimeError: " + "x" * 5000}
    line, _ = _line(b, detail)
    parsed = json.loads(line)
    assert parsed["legs"]["stark_prove"] == "error" and parsed["headline"]["prove_2p24_ms"] is None and parsed["headline"]["ntt_gelems"] == detail["value"]
    assert len(line) - line.rfind('"headline"') < 1500
