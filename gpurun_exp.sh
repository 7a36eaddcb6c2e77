#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f -o p -- python3 tools/final_stark_probe.py 5 > $O/f.log 2>&1
grep "ms per proof" $O/f.log
python3 - $O/f <<'PY'
import csv,sys,glob,re,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'qsplit_kernel' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
seg=rows[a:b]
t0=int(seg[0]['Start_Timestamp']); t1=int(seg[-1]['End_Timestamp'])
busy=0; cur_e=0
for r in seg:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if s>cur_e: busy+=e-s
    elif e>cur_e: busy+=e-cur_e
    cur_e=max(cur_e,e)
print("span %.2f ms kernels %d busy %.2f ms"%((t1-t0)/1e6,len(seg),busy/1e6))
acc=collections.Counter(); cnt=collections.Counter()
for r in seg:
    k=re.sub(r"\(anonymous namespace\)::|void ","",r['Kernel_Name']).split('(')[0][:60]
    acc[k]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); cnt[k]+=1
for k,v in acc.most_common(8): print("  %-62s %4d %9.1f us"%(k,cnt[k],v/1e3))
for r in seg:
    if 'bn128_' in r['Kernel_Name']:
        print(r['Kernel_Name'].split('(')[0][-28:], r['Grid_Size_X'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete
