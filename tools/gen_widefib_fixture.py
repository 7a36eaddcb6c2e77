#!/usr/bin/env python3
"""Writes tests/golden/widefib_w10.program.json: the starkinfo + program (reference serde shape) of
the synthetic wide-Fibonacci PIL (tools/synth_pil.py, W = 10, nBits = 10), produced by the oracle's
restated code generator (oracle/starkinfo.py).  Committed as a fixture so that bench.py never runs
oracle code outside its cpu_baseline leg.  Size-dependent fields are patched by synth_pil.rescale()."""
import json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tools"))
import starkinfo as SI, synth_pil
pil, ss = synth_pil.wide_fib_pil(10, 10), synth_pil.stark_struct(10)
info, prog, _ = SI.generate(pil, ss)
out = ROOT / "tests" / "golden" / "widefib_w10.program.json"
out.write_text(json.dumps(SI.to_json(info, prog), separators=(",", ":")))
print("wrote", out, out.stat().st_size, "bytes")
