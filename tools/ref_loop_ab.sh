R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
GENESIS_AUTOSTEP=0 python $R/tools/ref_loop_time.py 60 2>&1 | grep "reference loop" | sed 's/^/off: /'
GENESIS_AUTOSTEP=1 python $R/tools/ref_loop_time.py 60 2>&1 | grep "reference loop" | sed 's/^/on:  /'
done
