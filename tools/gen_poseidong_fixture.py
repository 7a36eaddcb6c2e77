#!/usr/bin/env python3
"""Writes tests/golden/poseidong.pil.json: `starkjs/poseidon/poseidong.pil` compiled by tools/pilc.py at the source's own
size (2^10 rows).  Runs in the authoring container only (reads /root/reference); the output is a data fixture."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import pilc

out = ROOT / "tests" / "golden" / "poseidong.pil.json"
out.write_text(pilc.dumps(pilc.compile_pil("/root/reference/starkjs/poseidon/poseidong.pil")) + "\n")
print("wrote", out)
