cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
Q="--cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0"
timeout 900 python -m pytest tests -m gpu -x -q -k "gated or genesis or vae or sylvester or running or sync or cross_replica" > gpurun_out/s5/t7.log 2>&1
tail -3 gpurun_out/s5/t7.log
for m in genesis vae; do
  for f in 1 0; do
    GENESIS_GATED_FUSE=$f timeout 300 python bench.py --model $m --steps 30 --warmup 5 $Q 2> gpurun_out/s5/b7_${m}_$f.err > gpurun_out/s5/b7_${m}_$f.json
    python -c "import json;d=json.load(open('gpurun_out/s5/b7_${m}_$f.json'));print('$m fuse=$f',round(d['value']),d['ms_per_step'])"
  done
done
bash tools/trace_model.sh genesis > /dev/null 2>&1
cp gpurun_out/trace_genesis_seq.txt gpurun_out/s5/trace_genesis_seq7.txt
head -1 gpurun_out/trace_genesis_sum.txt
