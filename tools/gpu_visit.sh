#!/bin/bash
# One short GPU-box visit during a round: gpurun --timeout N -- 'bash tools/gpu_visit.sh <what> [args]'.  Outputs under gpurun_out/.
# (tools/gpu_round.sh is the long end-of-round visit: whole GPU suite + smoke + profiles.)
mkdir -p gpurun_out; export TMPDIR=/tmp
what=$1; shift
case "$what" in
  tests)   # pytest -m gpu on the named files / -k expression
    timeout 3000 python -m pytest "$@" -m gpu -x -q --durations=8 > gpurun_out/visit_tests.log 2>&1; echo "rc=$?" >> gpurun_out/visit_tests.log
    tail -15 gpurun_out/visit_tests.log ;;
  bench)   # bench.py with the given flags; the line into gpurun_out/visit_bench.json
    timeout 1500 python bench.py "$@" > gpurun_out/visit_bench.json 2> gpurun_out/visit_bench.err; echo "rc=$?"
    tail -c 3000 gpurun_out/visit_bench.json; tail -5 gpurun_out/visit_bench.err ;;
  py)      # any script of tools/
    timeout 1500 python "$@" 2>&1 | tee gpurun_out/visit_py.log | tail -60 ;;
  *) echo "gpu_visit.sh: tests|bench|py"; exit 2 ;;
esac
