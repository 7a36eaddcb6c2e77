#!/usr/bin/env python3
"""Writes tests/golden/{fib,c12shape}.program.json.gz: starkinfo + program (reference serde shape) of the first STARK of a
recursion task (starky/data/fib.pil.json) and of the compressor-shaped PIL (tools/pil/c12_shape.pil), from the oracle's restated
code generator, so that bench.py's aggregation leg never runs oracle code.  The c12shape fixture is generated at 2^10 rows and
patched to 2^15 / 2^18 by synth_pil.rescale()."""
import gzip, json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tools"))
import starkinfo as SI, aggregation_workload as AW
ss10 = AW.STRUCTS["fib"]
for name, pil in (("fib", AW.fib_pil()), ("c12shape", AW.c12_pil(10))):
    info, prog, _ = SI.generate(pil, ss10)
    out = ROOT / "tests" / "golden" / ("%s.program.json.gz" % name)
    out.write_bytes(gzip.compress(json.dumps(SI.to_json(info, prog), separators=(",", ":")).encode(), mtime=0))
    print("wrote", out, out.stat().st_size, "bytes")
