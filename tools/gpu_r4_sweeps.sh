#!/bin/bash
# round 4: size sweeps of the primitives, looking for cliffs between kernel variants (the large-MSM one was found this way)
mkdir -p gpurun_out/r4sw; O=gpurun_out/r4sw
for nb in 10 12 14 16 17 18 19 20 21 22 23 24 25 26; do timeout 100 python tools/ntt_time.py $nb 1 2>&1 | grep "^nbits" >> $O/ntt_np1.txt; done
for nb in 10 12 14 16 17 18 19 20 21 22; do timeout 100 python tools/ntt_time.py $nb 19 2>&1 | grep "^nbits" >> $O/ntt_np19.txt; done
timeout 600 python tools/merkle_bench.py 10 19 12 19 13 19 14 19 15 19 16 19 17 19 18 19 19 19 20 19 21 19 22 19 23 19 24 19 25 19 2>&1 | grep "^merkelize" > $O/merkle_w19.txt
timeout 600 python tools/merkle_bench.py 12 6 14 6 16 6 18 6 20 6 22 6 24 6 12 36 14 36 16 36 18 36 20 36 22 36 24 36 2>&1 | grep "^merkelize" > $O/merkle_w6_w36.txt
timeout 900 python tools/prove_bench.py --nbits 10 12 14 15 16 17 18 19 20 21 22 23 --reps 3 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); nb = int(d['workload'].split('nBits=')[1].split(',')[0]); ms = min(d['stark_gen_ms'][1:])
    print('prove 2^%d: %.2f ms  %.2f Mrows/s' % (nb, ms, (1 << nb) / ms / 1e3))" > $O/prove_sweep.txt
cat $O/ntt_np1.txt | cut -c1-90; cat $O/ntt_np19.txt | cut -c1-90; cat $O/merkle_w19.txt $O/merkle_w6_w36.txt $O/prove_sweep.txt
