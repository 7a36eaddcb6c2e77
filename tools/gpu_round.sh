#!/bin/bash
# Every GPU-box visit of a round goes through this one script:  gpurun --timeout N -- 'bash tools/gpu_round.sh <what> [args]'
# Outputs under gpurun_out/ (scratch); copy the summaries you want judged into profiles/rNN/.
#   all                      the end-of-round visit: whole GPU suite, smoke, `profiles`, `small`
#   tests [pytest args]      pytest -m gpu -x on the named files / -k expression (default: tests)
#   bench [bench.py flags]   the bench line -> gpurun_out/bench_stdout.txt (+ bench_detail.json)
#   py script.py [args]      any script of tools/
#   profiles                 everything profiles/rNN/ holds: bench line, rocprofv3 kernel statistics of the NTT bench / a 2^24-row proof / the
#                            MSM / Groth16 / the final STARK, HBM-traffic counters (FETCH_SIZE and WRITE_SIZE in separate passes), SQ counters
#   sq | icache              the SQ instruction-mix counters / the instruction-cache counters alone
#   small                    the three circuits of a recursion task under the profiler: busy / idle split and launch timelines
#   prof NAME cmd...         rocprofv3 --kernel-trace --stats of any command -> gpurun_out/prof_NAME/, top of the kernel statistics printed
#   ab trees|ntt|msm|prove|fr LIB...   parity tests, then alternating timings of library variants (eigen-zkvm_amd/variants/libzkgpu_LIB.so built by
#                            tools/build_variant.sh; "shipped" = the library; LIB@ENV=VAL runs a variant under an environment switch)
#   fuzz [seed]              a fuzz campaign (profiles/rNN/fuzz.txt)
mkdir -p gpurun_out; export TMPDIR=/tmp
what=${1:-all}; shift
stats() { f=$(find "$1" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -${2:-16}; }
clean() { find "$1" -name '*kernel_trace.csv' -delete; find "$1" -name '*.db' -delete; find "$1" -name '*counter_collection.csv' -delete; }
sel() { v=${1%%@*}; e=""; [ "$1" != "$v" ] && e=${1#*@}; if [ $v = shipped ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; fi; }
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
do_tests() { timeout 3300 python -m pytest "${@:-tests}" -m gpu -x -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log; }
do_bench() { timeout 1500 python bench.py "$@" > gpurun_out/bench_stdout.txt 2> gpurun_out/bench_stderr.txt; echo "rc=$?"; tail -c 1500 gpurun_out/bench_stdout.txt; grep -v '^bench_detail' gpurun_out/bench_stderr.txt | tail -5; }
do_sq() {
  local O=gpurun_out/profiles/sq; rm -rf $O; mkdir -p $O
  timeout 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O -o sq -- python3 tools/pmc_ntt.py > $O.log 2>&1
  { echo "# rocprofv3 --pmc SQ_* over tools/pmc_ntt.py (2^24 NTT x8, 2^20->2^21 x20 LDE + Merkle)"; python3 tools/pmc_sq_summarize.py $O; } > gpurun_out/profiles/pmc_sq.txt
  head -20 gpurun_out/profiles/pmc_sq.txt | cut -c1-250; clean $O
}
do_icache() {
  local O=gpurun_out/pmc_ic; rm -rf $O; mkdir -p $O; : > gpurun_out/pmc_icache.txt
  for spec in "msm_bn:tools/msm_bench.py bn254 g1 20" "msm_bls:tools/msm_bench.py bls12_381 g1 20" "g16:tools/groth16_bench.py BLS12381 18" "fs:tools/final_stark_probe.py 2"; do
    n=${spec%%:*}; c=${spec#*:}
    timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/$n -o ic -- python3 $c > $O/$n.log 2>&1
    { echo "== $n"; python3 tools/pmc_sq_summarize.py $O/$n 200; } >> gpurun_out/pmc_icache.txt
  done
  cut -c1-260 gpurun_out/pmc_icache.txt; clean $O
}
do_prof() { local n=$1; shift; local O=gpurun_out/prof_$n; rm -rf $O; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $n -- "$@" > $O.log 2>&1; tail -3 $O.log; stats $O 24; }
do_profiles() {
  local O=gpurun_out/profiles; rm -rf $O; mkdir -p $O
  timeout 1500 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 800 $O/bench.json; cp gpurun_out/bench_detail.json $O/bench_detail.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ntt -o ntt -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-prove --no-msm --no-bn128 --no-groth16 --no-poseidon --no-agg > $O/ntt.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prove -o prove -- python3 tools/prove_bench.py --nbits 24 --reps 3 > $O/prove.log 2>&1; tail -1 $O/prove.log | cut -c1-400
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/msm -o msm -- python3 tools/msm_bench.py bn254 g1 22 > $O/msm.log 2>&1; tail -2 $O/msm.log
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g16_bls -o g16 -- python3 tools/groth16_bench.py BLS12381 18 > $O/g16_bls.log 2>&1; tail -2 $O/g16_bls.log
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g16_bn -o g16 -- python3 tools/groth16_bench.py BN128 18 > $O/g16_bn.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fs -o p -- python3 tools/final_stark_probe.py 8 > $O/fs.log 2>&1; tail -2 $O/fs.log
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$c -o ntt -- python3 tools/pmc_ntt.py > $O/pmc_$c.log 2>&1
  done
  python3 tools/pmc_summarize.py $O/pmc > $O/pmc_hbm_traffic.txt; cp $O/pmc/pmc_hbm_traffic.json $O/; tail -3 $O/pmc_hbm_traffic.txt
  do_sq
  clean $O; du -sh $O
}
do_small() {
  for k in fib c12 r1; do
    rm -rf gpurun_out/sp_$k
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_$k -o p -- python3 tools/small_proof_probe.py $k 20 > gpurun_out/sp_$k.log 2>&1
    f=$(find gpurun_out/sp_$k -name '*kernel_trace.csv' | head -1)
    python3 tools/trace_gaps.py $f > gpurun_out/sp_${k}_gaps.txt 2>&1; head -3 gpurun_out/sp_${k}_gaps.txt
    python3 tools/proof_timeline.py $f > gpurun_out/tl_$k.txt 2>&1
    rm -f $f
  done
  timeout 100 python tools/coop_perm_time.py | tail -1
}
do_ab() {
  fam=$1; shift; out=gpurun_out/ab_$fam.txt; : > $out
  case $fam in
    trees) par="tests/test_gpu_parity.py"; run() { timeout 300 python tools/merkle_bench.py 22 19 22 36 18 12 16 37 24 10 2>&1 | cut -c1-50; } ;;
    ntt)   par="tests/test_gpu_parity.py -k ntt or lde or fft or interpolate or root"; run() { for sh in "24 1" "20 36" "16 12"; do timeout 300 python tools/ntt_time.py $sh; done; timeout 300 python tools/lde_time.py 24 19 36; } ;;
    msm)   par="tests/test_gpu_msm.py"; run() { for c in "bn254 g1 22" "bls12_381 g1 22" "bn254 g1 18"; do timeout 300 python tools/msm_bench.py $c 2>&1 | tail -2; done; timeout 300 python tools/groth16_bench.py BLS12381 18 2>&1 | tail -1; } ;;
    prove) par="tests/test_gpu_stark.py"; run() { timeout 600 python tools/prove_bench.py --nbits 20 24 --reps 3 2>&1 | cut -c1-260; timeout 200 python tools/small_proof_probe.py r1 20 2>&1 | tail -1; } ;;
    fr)    par="tests/test_gpu_bn128.py"; run() { for sh in "bn128 24 10 3" "bn128 22 5 3" "bn128 22 10 3" "bn128 22 13 3" "bn128 20 19 5" "bn128 20 24 5" "bn128 20 36 5" "bn128 20 48 5" "bn128 22 48 3" "bls12381 22 10 3" "bls12381 20 48 5"; do timeout 300 python tools/fr_merkle_time.py $sh 2>&1 | cut -c1-80; done; timeout 300 python tools/final_stark_probe.py 4 2>&1 | tail -1 | cut -c1-200; } ;;
    *) echo "ab: trees|ntt|msm|prove|fr"; exit 2 ;;
  esac
  for a in "$@"; do sel $a; echo "== parity $a" >> $out; env $e timeout 900 python -m pytest $par -m gpu -x -q 2>&1 | tail -2 >> $out; done
  for r in 1 2; do for a in "$@"; do sel $a; echo "== $a (run $r)" >> $out; env $e bash -c "$(declare -f run); run" >> $out 2>&1; done; done
  unset ZKGPU_LIB; cat $out
}
do_fuzz() {
  s=${1:-500}; out=gpurun_out/fuzz.txt; : > $out
  timeout 500 python tools/fuzz_programs.py $((s + 1000)) $((s + 1150)) 2>&1 | tail -2 >> $out
  for k in 1 2 3; do timeout 200 python tools/fuzz_primitives.py $((s + 30 + k)) 3000 2>&1 | tail -1 >> $out; done
  for k in 1 2 3; do timeout 400 python tools/fuzz_proofs.py $((s + 10 + k)) 100 2>&1 | tail -1 >> $out; done
  timeout 600 python tools/fuzz_verify.py $((s + 20)) 150 2>&1 | tail -1 >> $out
  cat $out
}
case "$what" in
  all)      do_tests; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log; do_profiles > gpurun_out/profiles.log 2>&1; tail -8 gpurun_out/profiles.log; do_small ;;
  tests)    do_tests "$@" ;;
  bench)    do_bench "$@" ;;
  py)       timeout 1500 python "$@" 2>&1 | tee gpurun_out/visit_py.log | tail -60 ;;
  profiles) do_profiles ;;
  sq)       mkdir -p gpurun_out/profiles; do_sq ;;
  icache)   do_icache ;;
  small)    do_small ;;
  prof)     do_prof "$@" ;;
  ab)       do_ab "$@" ;;
  fuzz)     do_fuzz "$@" ;;
  *) sed -n 2,18p "$0"; exit 2 ;;
esac
