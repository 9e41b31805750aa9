# A/B of environment settings on the bench's timed loop:  tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...   (each twice, interleaved)
ARGS=$1; shift
for rep in 1 2; do for E in "$@"; do
  env $E python bench.py $ARGS --steps 100 --warmup 20 --cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0 --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-44s %9.1f img/s  %.4f ms   steady %.1f' % ('$E', d['value'], d['ms_per_step'], d['steady_state']['value']))"
done; done
