cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/prof6 -o r04 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --profile-steps 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0 > /tmp/prof6.log 2>&1; python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof6/r04_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
idx=[i for i,n in enumerate(names) if "adam_pair_kernel" in n or "adam_kernel<float" in n]
a,b=idx[-2],idx[-1]
out=open("/root/repo/gpurun_out/step_seq.txt","w")
t0=int(rows[a]["End_Timestamp"])
for r in rows[a+1:b+1]:
    out.write("%9.1f %7.1f %s\n"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"][:150]))
out.close()
PY
