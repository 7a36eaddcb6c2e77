#!/bin/bash
# rocprofv3 kernel trace of the 2^24 NTT alone: per-pass launch durations (tools/ntt_time.py 24 1)
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_ntt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ntt -o ntt -- python3 tools/ntt_time.py 24 1 > gpurun_out/prof_ntt.log 2>&1
tail -2 gpurun_out/prof_ntt.log
f=$(find gpurun_out/prof_ntt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -12
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/prof_ntt/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'ntt_pass' in r['Kernel_Name']]
# group by position within a transform (3 passes)
d=collections.defaultdict(list)
for i,r in enumerate(rows): d[(i%3, r['Kernel_Name'].split('<')[1].split('>')[0])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()): print(k, len(v), 'avg us', round(sum(v)/len(v),1))
PY
