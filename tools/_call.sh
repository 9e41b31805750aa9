cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
timeout 900 python -m pytest tests -m gpu -x -q -k "s2 or small or encoder or component" > gpurun_out/s5/t5.log 2>&1
tail -3 gpurun_out/s5/t5.log
bash tools/trace_model.sh monet > /dev/null 2>&1
cp gpurun_out/trace_monet_seq.txt gpurun_out/s5/trace_monet_seq5.txt
grep "conv3x3s2" gpurun_out/trace_monet_seq.txt | cut -c1-100
head -1 gpurun_out/trace_monet_sum.txt
