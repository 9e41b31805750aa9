"""The SMALL golden fixtures (tests/golden/<case>.npz: the tiny models of tests/test_model_gpu.py) repeated beside a loading process:
forward + backward and sample(), every output and gradient compared bit for bit with the first repetition.
usage: diag_shared_gpu4.py <tiny|tiny_k6|metric|cfg5> [repetitions]   (NOLOAD=1: control)"""
import os
import subprocess
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)



def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    from tests.test_model_gpu import build, run
    from tests.common import Golden as G
    gold = G(case)
    x, rp, eps = gold.inputs()
    trace = []
    if os.environ.get('TRACE') == '1':          # checksums of the tensor results (and tensor arguments) of EVERY hip_ops call, in call order
        import types
        from genesis_amd import hip_ops as hip

        def cks(t):
            return (float(t.double().sum()), float(t.double().abs().sum())) if t.is_floating_point() else float(t.double().sum())

        def tensors(o):
            if torch.is_tensor(o):
                return [o]
            if isinstance(o, (list, tuple)):
                return [t for e in o for t in tensors(e)]
            return []

        def wrap(name, orig):
            def f(*a, **k):
                r = orig(*a, **k)
                if trace:
                    args = tensors(list(a)) + tensors(list(k.values()))
                    trace[-1].append((name, [cks(t) for t in args if t.is_cuda and t.numel()], [cks(t) for t in tensors(r) if t.is_cuda and t.numel()]))
                return r
            return f
        for name, obj in list(vars(hip).items()):
            if isinstance(obj, types.FunctionType) and not name.startswith('_') and obj.__module__ == hip.__name__ and \
                    name not in ('take_amax', 'amax_link', 'defer_flush', 'lstm_seq_steps', 'lstm_seq_capacity'):
                setattr(hip, name, wrap(name, obj))

    def once():
        if os.environ.get('TRACE') == '1':
            trace.append([])
        model = build(gold).cuda()
        out = run(model, gold, x, rp, eps)
        recon, losses, stats, att, comp = out
        loss = losses['err'].mean() + sum(k.mean() for k in losses['kl_l_k']) if isinstance(losses.get('kl_l_k', None), (list, tuple)) else losses['err'].mean() + losses['kl_l'].mean()
        loss.backward()
        res = {'recon': recon.detach().clone(), 'err': losses['err'].detach().clone()}
        for k, v in stats.items():
            if torch.is_tensor(v):
                res['stats.' + k] = v.detach().clone()
            elif isinstance(v, (list, tuple)) and v and torch.is_tensor(v[0]):
                res['stats.' + k] = torch.stack([t.detach() for t in v]).clone()
        for n, p in model.named_parameters():
            if p.grad is not None:
                res['grad.' + n] = p.grad.detach().clone()
        torch.manual_seed(5)
        model.eval()
        with torch.no_grad():
            s = model.sample(3, gold.cfg['K_steps'])
        res['sample.recon'] = s[0].detach().clone()
        torch.cuda.synchronize()
        return res

    ref = once()
    loadp = None
    if os.environ.get('NOLOAD') != '1':
        loadp = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'load_gpu.py'), '100000'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        import time
        time.sleep(25)
    bad = {}
    nbad = 0
    try:
        for i in range(reps):
            r = once()
            diff = [k for k in ref if not torch.equal(r[k], ref[k])]
            nbad += bool(diff)
            if trace:
                for ci, (a, b) in enumerate(zip(trace[0], trace[-1])):
                    if a[2] != b[2]:        # (every earlier call returned the same bits: this call's operands were the same)
                        k = [j for j, (u, v) in enumerate(zip(a[2], b[2])) if u != v][0]
                        print('repetition %d: first call whose results differ: #%d %s, result tensor %d: first run %s, this run %s' % (
                            i, ci, a[0], k, a[2][k], b[2][k]), flush=True)
                        break
                del trace[1:]
            for k in diff:
                d = (r[k].double() - ref[k].double()).abs()
                b = bad.setdefault(k, [0, 0.0, 0])
                b[0] += 1
                b[1] = max(b[1], float(d.max()))
                b[2] = max(b[2], int((d > 0).sum()))
    finally:
        if loadp is not None:
            loadp.kill()
    print('%s (%s): %d of %d repetitions differ from the first' % (case, 'alone' if loadp is None else 'beside a loading process', nbad, reps), flush=True)
    for k in ref:
        if k in bad:
            print('   %-52s in %3d repetitions, max |d| %.3e, up to %d of %d elements' % (k, bad[k][0], bad[k][1], bad[k][2], ref[k].numel()), flush=True)


if __name__ == '__main__':
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    main()
