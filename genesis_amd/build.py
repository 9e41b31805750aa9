"""Builds genesis_amd/libgenesis_hip.so (the C-ABI library of hand-written gfx950 HIP kernels)
in-tree with hipcc.  `python -m genesis_amd.build [--force]`."""
import os
import os.path as osp
import subprocess
import sys

HERE = osp.dirname(osp.abspath(__file__))
CSRC = osp.join(HERE, 'csrc')
LIB = osp.join(HERE, 'libgenesis_hip.so')
SOURCES = ['gx_api.cpp', 'gx_comm.cpp', 'gx_conv.hip', 'gx_norm.hip', 'gx_attention.hip', 'gx_slots.hip', 'gx_optim.hip', 'gx_misc.hip', 'gx_gated.hip', 'gx_latent.hip', 'gx_dense.hip', 'gx_igemm.hip', 'gx_metrics.hip', 'gx_feed.hip', 'gx_wino.hip', 'gx_kq.hip', 'gx_bcast.hip', 'gx_sbp.hip', 'gx_wgq.hip', 'gx_wstrip.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value',
         '-Wno-pass-failed']


def _deps():
    d = [osp.join(CSRC, s) for s in SOURCES if osp.exists(osp.join(CSRC, s))]
    d += [osp.join(CSRC, 'gx_common.h'), osp.join(osp.dirname(HERE), 'include', 'genesis_hip.h')]
    return d


def needs_build():
    if not osp.exists(LIB):
        return True
    t = osp.getmtime(LIB)
    return any(osp.getmtime(f) > t for f in _deps())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        src = osp.join(CSRC, s)
        if not osp.exists(src):
            continue
        obj = osp.join(CSRC, osp.splitext(s)[0] + '.o')
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + (['-x', 'hip'] if s.endswith('.cpp') else []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % s)
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
