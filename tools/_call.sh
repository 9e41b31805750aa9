cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
Q="--cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0"
timeout 900 python -m pytest tests -m gpu -x -q -k "monet or genesis or bcast or broadcast or strip or quad or conv1x1 or vae" > gpurun_out/s5/t2.log 2>&1
tail -3 gpurun_out/s5/t2.log
for m in monet genesis; do
  for t in 1 0; do
    GENESIS_BCAST_CHAIN_TAPS=$t timeout 300 python bench.py --model $m --steps 30 --warmup 5 $Q 2> gpurun_out/s5/b2_${m}_$t.err > gpurun_out/s5/b2_${m}_$t.json
    python -c "import json;d=json.load(open('gpurun_out/s5/b2_${m}_$t.json'));print('$m taps=$t',round(d['value']),d['ms_per_step'])"
  done
done
for f in 1024 2048 1024 2048; do
  GENESIS_KQ_FOLD_ABOVE=$f timeout 300 python bench.py --steps 100 --warmup 20 $Q 2> gpurun_out/s5/b2_metric_$f.err > gpurun_out/s5/b2_metric_$f.json
  python -c "import json;d=json.load(open('gpurun_out/s5/b2_metric_$f.json'));print('metric fold_above=$f',round(d['value']),d['ms_per_step'])"
done
bash tools/trace_model.sh monet > /dev/null 2>&1
cp gpurun_out/trace_monet_seq.txt gpurun_out/s5/trace_monet_seq2.txt
grep "kq_c3h_kernel<3" gpurun_out/trace_monet_seq.txt | cut -c1-90
