#!/bin/bash
# tools/trace_model.sh <model> [extra bench args]: rocprofv3 kernel trace of `bench.py --model <model>`; writes the last
# optimizer step's kernel sequence (start us, duration us, name) to gpurun_out/trace_<model>_seq.txt, the per-kernel
# totals of that step to gpurun_out/trace_<model>_sum.txt and the stats CSV to gpurun_out/trace_<model>_kernel_stats.csv.
M=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tm_$M
rocprofv3 --kernel-trace --stats -d /tmp/tm_$M -o t --output-format csv -- python $R/bench.py --model $M --steps 6 --warmup 2 --cpu-seconds 0 --profile-steps 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0 "$@" > /tmp/tm_$M.log 2>&1
T=$(find /tmp/tm_$M -name "t_kernel_trace.csv" | head -1); S=$(find /tmp/tm_$M -name "t_kernel_stats.csv" | head -1)
cp $S $R/gpurun_out/trace_${M}_kernel_stats.csv
python - <<PY
import csv, collections, re
rows=list(csv.DictReader(open("$T")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
idx=[i for i,n in enumerate(names) if "adam_pair_kernel" in n or "adam_kernel<float" in n or "adam_kernel" in n]
# the last launch of each step's optimizer: step boundaries = gaps between runs of adam launches
ends=[i for k,i in enumerate(idx) if k+1==len(idx) or idx[k+1]!=i+1]
a,b=ends[-2],ends[-1]
t0=int(rows[a]["End_Timestamp"])
tot=collections.OrderedDict(); cnt=collections.Counter()
with open("$R/gpurun_out/trace_${M}_seq.txt","w") as out:
    for r in rows[a+1:b+1]:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        out.write("%9.1f %7.1f %s\n"%((int(r["Start_Timestamp"])-t0)/1e3,d,r["Kernel_Name"][:150]))
        k=re.sub(r"\(.*","",r["Kernel_Name"])[:90]
        tot[k]=tot.get(k,0.0)+d; cnt[k]+=1
span=(int(rows[b]["End_Timestamp"])-t0)/1e3
with open("$R/gpurun_out/trace_${M}_sum.txt","w") as out:
    out.write("step span %.1f us, kernel sum %.1f us, %d launches\n"%(span,sum(tot.values()),b-a))
    for k,v in sorted(tot.items(),key=lambda kv:-kv[1]):
        out.write("%9.1f us %4d x %7.1f  %s\n"%(v,cnt[k],v/cnt[k],k))
PY
