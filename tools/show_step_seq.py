"""Pretty-prints gpurun_out/step_seq.txt (tools/trace_step.sh): library kernels between framework kernels."""
import re, sys
L = open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/step_seq.txt').read().splitlines()


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    n = re.sub(r'vectorized_elementwise_kernel<4, ', 'vec<', n)
    n = re.sub(r'elementwise_kernel_manual_unroll<128, 4, gpu_kernel_impl_nocast<', 'ew<', n)
    n = re.sub(r'unrolled_elementwise_kernel<', 'unr<', n)
    n = re.sub(r'std::array<char\*, \d+ul>', '', n)
    n = re.sub(r'BinaryFunctor<float, float, float, binary_internal::', 'Bin<', n)
    n = re.sub(r'BinaryFunctor<float, float, float, at::native::binary_internal::', 'Bin<', n)
    n = re.sub(r'BUnaryFunctor<float, float, float, (at::native::)?binary_internal::', 'BUn<', n)
    n = re.sub(r'AUnaryFunctor<float, float, float, (at::native::)?binary_internal::', 'AUn<', n)
    n = re.sub(r'Cijk_\w+?_MT(\d+x\d+x\d+).*', 'GEMM_MT\\1', n)
    return n[:70]


mine = ('tapconv', 'wgrad', 'gn_relu', 'splitk', 'pack_weights', 'conv1x1', 'maskpool', 'icsbp', 'mixture', 'adam',
        'geco', 'gn_param', 'col_sum', 'step_inc', 'posterior', 'prior_logp', 'row_sum', 'sum_double', 'dense_kernel',
        'lstm_', 'elbo', 'pooled')
res, cur = [], None
tot_lib = tot_mine = 0.0
nlib = 0
for l in L:
    t, d, n = l.split(None, 2)
    if any(m in n for m in mine):
        tot_mine += float(d)
        if cur is None:
            cur = []
            res.append(cur)
        cur.append(short(n).split('(')[0][:20] + ' %.0f' % float(d))
    else:
        cur = None
        tot_lib += float(d); nlib += 1
        res.append('%6.1f %s' % (float(d), short(n)))
for r in res:
    if isinstance(r, list):
        print('   == [%d] ' % len(r) + ', '.join(r[:8]) + (' ...' if len(r) > 8 else ''))
    else:
        print(r)
print('framework kernels %.0f us, library kernels %d = %.0f us' % (tot_mine, nlib, tot_lib))
