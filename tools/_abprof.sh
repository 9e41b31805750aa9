cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
Q="--cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --profile-steps 0"
for f in 0 1; do
GENESIS_DGRAD_ACT_FUSE=$f rocprofv3 --kernel-trace --stats -d /tmp/ks$f -o ks --output-format csv -- python $R/bench.py --model monet --steps 10 --warmup 3 $Q > /tmp/ks$f.log 2>&1
echo "== fuse=$f"
grep -E "kq_c3h_kernel<3>|bias_act_bwd|plane_sum|chan_sum|wgrad_fast_kernel<0, 3" $(find /tmp/ks$f -name "ks_kernel_stats.csv" | head -1) | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,50), $2}' 
done
