#!/usr/bin/env python3
"""Writes tests/golden/poseidong.program.json.gz: the starkinfo + program (reference serde shape) of the PoseidonG PIL
(tests/golden/poseidong.pil.json, blow-up 2) from the oracle's restated code generator (oracle/starkinfo.py), so that
bench.py never runs oracle code outside its cpu_baseline leg.  Size-dependent fields are patched by synth_pil.rescale()."""
import gzip, json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tools"))
import starkinfo as SI, poseidong as PG
info, prog, _ = SI.generate(PG.pil(10), PG.stark_struct(10))
out = ROOT / "tests" / "golden" / "poseidong.program.json.gz"
out.write_bytes(gzip.compress(json.dumps(SI.to_json(info, prog), separators=(",", ":")).encode(), mtime=0))
print("wrote", out, out.stat().st_size, "bytes")
