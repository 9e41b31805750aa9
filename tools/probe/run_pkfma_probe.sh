# tools/probe/run_pkfma_probe.sh: the probe alone, then beside a process that runs GENESIS training iterations on the same GPU
cd $GRAFT_REPO_ROOT
P=tools/abl/pkfma_probe
[ -x $P ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -Wno-unused-value -o $P tools/probe/pkfma_probe.hip
echo "=== alone"
$P 0.4
echo "=== beside another process (GENESIS training iterations)"
python - <<'PY' &
import sys, time, torch
sys.path.insert(0, '.')
from tests.test_fullbatch_gpu import Full
gold = Full('genesis_cfg3_b32')
x, nz = gold.x(), gold.noise()
model = gold.build()
t0 = time.time()
while time.time() - t0 < 150:
    out = gold.forward(model, x, nz)
    err, kl = gold.aggregate(out[1])
    (err + kl).backward()
    torch.cuda.synchronize()
PY
LOADPID=$!
sleep 25
$P 1.5
wait $LOADPID
