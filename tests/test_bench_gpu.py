"""bench.py contract: ONE JSON line with the driver's keys, the roofline object (HIP events around the dominant kernel
inside the same process) and the CPU baseline object; the timed value comes from HIP-graph replay."""
import json
import os
import os.path as osp
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_line_contract():
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, osp.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '2', '--cpu-seconds', '2',
                          '--profile-steps', '1', '--host-input-steps', '3', '--extra-leg-steps', '3'], capture_output=True, text=True, env=env,
                         cwd=ROOT, timeout=580)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['unit'] == 'images/sec' and d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and 'workload' in d['config'] and d['config']['launch'] == 'hip-graph'
    assert d['value'] > 0 and abs(d['value'] - 32 * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['value'] > 0 and c['cores'] >= 1
    assert d['pcie_inclusive']['value'] > 0
    # the legs that price what `value` leaves out / what the unchanged train.py loop gets (never `value`)
    assert 0 < d['value_as_written']['value'] and len(d['value_as_written']['mse_rmse']) == 2
    assert 0 < d['value_reference_loop']['value'] < d['value'] * 1.05
    assert abs(d['value_reference_loop']['final_elbo'] - d['final_elbo']) < 0.2 * abs(d['final_elbo'])
    assert 'value_reference_loop' in d['config']['workload']


@pytest.mark.timeout(900)
def test_bench_two_rank_launch_rehearsal():
    """The driver's N > 1 command line (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N) on the
    1-GPU test box: GENESIS_BENCH_REHEARSAL=1 puts both ranks on GPU 0 with gloo carrying the gradient bucket (RCCL refuses
    two ranks on one device).  Everything else is the real path: env:// rendezvous, per-rank shards and seeds, rank-0
    broadcast, the two-graph step around the ONE collective, barriers, max-over-ranks time, one JSON line from rank 0."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, PYTHONPATH=ROOT, GENESIS_BENCH_REHEARSAL='1', GENESIS_BENCH_LONG_STEPS='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                          '127.0.0.1', '--master-port', str(port), osp.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4',
                          '--warmup', '2'], capture_output=True, text=True, env=env, cwd=ROOT, timeout=880)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['parallelism'] == 'dp2'
    assert d['config']['global_batch'] == 64 and d['config']['per_gpu_batch'] == 32
    c = d['config']['collective']
    assert c['ranks_observed'] == 2 and c['all_reduces_per_step'] == 1 and c['bytes_per_all_reduce'] > 10e6
    assert d['config']['launch'] == 'hip-graph(fwd+bwd) | rccl all-reduce | hip-graph(geco+adam)'
    assert 'rehearsal' in d and d['value'] > 0 and abs(d['value'] - 64 * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    assert 'roofline' not in d and 'cpu_baseline' not in d    # rank-0-at-N=1 legs only
    assert c['all_reduce_ms'] > 0 and 0 < c['all_reduce_share_of_step'] < 1


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` BY ITSELF -- the shape of the driver's 1-GPU command with another N (missing 1;
    the reference's multi-GPU mode is one command as well, train.py:133-137,153-155): with no WORLD_SIZE in the environment
    the script execs the one-process-per-GPU launcher, and the command's stdout is still rank 0's ONE JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(PYTHONPATH=ROOT, GENESIS_BENCH_REHEARSAL='1', GENESIS_BENCH_LONG_STEPS='0')
    out = subprocess.run([sys.executable, osp.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2'],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=880)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['collective']['ranks_observed'] == 2 and d['config']['global_batch'] == 64
    assert d['config']['collective']['all_reduce_ms'] > 0 and d['value'] > 0
