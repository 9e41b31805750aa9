// Optimiser-side kernels of the training step: fused flat-buffer Adam and the GECO multiplier update.
//
// Reference: train.py:174-175,262-263 (torch.optim.Adam(lr=1e-4), betas (0.9, 0.999), eps 1e-8, no weight
// decay) and utils/geco.py:35-51.  Both read their step-dependent scalars from device memory so a whole
// training step can be replayed from a HIP graph without host round-trips (the reference's
// `constraint.item()` at geco.py:45 is a device->host sync every iteration).
#include "gx_common.h"

namespace {

// p, g, m, v: flat buffers of n elements.  step: device int64 counter, already incremented for this
// update (t >= 1).  gscale multiplies the gradient first (1/world_size for data-parallel averaging).
template <typename T, bool ZERO_G = false>
__device__ __forceinline__ void adam_body(T* __restrict__ p, T* __restrict__ g, T* __restrict__ m, T* __restrict__ v,
                                          size_t n, const int64_t* __restrict__ step, double lr, double b1, double b2,
                                          double eps, float gscale, unsigned block, unsigned nblocks) {
    // hyper-parameters arrive as doubles and every derived constant is formed in double before it is rounded to T,
    // as torch.optim.Adam does with its Python floats (1 - 0.999 in fp32 is 4.7e-5 off the fp64 value it uses)
    const double t = (double)(*step);
    const double bc1 = 1.0 - pow(b1, t);
    const double bc2 = 1.0 - pow(b2, t);
    const T step_size = (T)(lr / bc1);
    const T bc2_sqrt = (T)sqrt(bc2);
    const T one_m_b1 = (T)(1.0 - b1), one_m_b2 = (T)(1.0 - b2), b2_t = (T)b2, eps_t = (T)eps;
    for (size_t i = (size_t)block * blockDim.x + threadIdx.x; i < n; i += (size_t)nblocks * blockDim.x) {
        const T gi = g[i] * (T)gscale;
        if (ZERO_G) g[i] = (T)0;                                   // the next iteration accumulates into a clean bucket
        const T mi = m[i] + (gi - m[i]) * one_m_b1;                // torch: exp_avg.lerp_(grad, 1 - beta1)
        const T vi = v[i] * b2_t + gi * gi * one_m_b2;             // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        const T denom = sqrt(vi) / bc2_sqrt + eps_t;
        p[i] = p[i] - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
adam_kernel(T* __restrict__ p, const T* __restrict__ g, T* __restrict__ m, T* __restrict__ v, size_t n,
            const int64_t* __restrict__ step, double lr, double b1, double b2, double eps, float gscale) {
    adam_body<T, false>(p, const_cast<T*>(g), m, v, n, step, lr, b1, b2, eps, gscale, blockIdx.x, gridDim.x);
}

// the fp32 and the fp64 parameter groups in one launch (blocks [0, nb32) / [nb32, gridDim.x)), optionally zeroing the
// gradients they consumed: the tail of a training step is this launch and the GECO / step-counter one
struct AdamGroup { void* p; void* g; void* m; void* v; size_t n; };
template <bool ZERO_G>
__global__ void __launch_bounds__(256)
adam_pair_kernel(const AdamGroup a, const AdamGroup b, unsigned nb32, const int64_t* __restrict__ step, double lr,
                 double b1, double b2, double eps, float gscale) {
    if (blockIdx.x < nb32)
        adam_body<float, ZERO_G>((float*)a.p, (float*)a.g, (float*)a.m, (float*)a.v, a.n, step, lr, b1, b2, eps, gscale,
                                 blockIdx.x, nb32);
    else
        adam_body<double, ZERO_G>((double*)b.p, (double*)b.g, (double*)b.m, (double*)b.v, b.n, step, lr, b1, b2, eps,
                                  gscale, blockIdx.x - nb32, gridDim.x - nb32);
}

// torch.optim.RMSprop(lr) defaults (alpha 0.99, eps 1e-8, no momentum, not centred; train.py:171-172) and
// torch.optim.SGD(lr, momentum 0.9) (train.py:175-176): one state buffer `m` (square average / momentum buffer, zero-initialised:
// mu * 0 + g reproduces torch's buf = grad at the first step).  KIND 1 = RMSprop, 2 = SGD with momentum.
template <typename T, int KIND, bool ZERO_G>
__device__ __forceinline__ void simple_opt_body(T* __restrict__ p, T* __restrict__ g, T* __restrict__ m, size_t n, double lr,
                                                double hp, double eps, float gscale, unsigned block, unsigned nblocks) {
    const T lr_t = (T)lr, hp_t = (T)hp, one_m = (T)(1.0 - hp), eps_t = (T)eps;
    for (size_t i = (size_t)block * blockDim.x + threadIdx.x; i < n; i += (size_t)nblocks * blockDim.x) {
        const T gi = g[i] * (T)gscale;
        if (ZERO_G) g[i] = (T)0;
        if (KIND == 1) {
            const T sq = m[i] * hp_t + gi * gi * one_m;            // square_avg.mul_(alpha).addcmul_(g, g, 1 - alpha)
            m[i] = sq;
            p[i] = p[i] - lr_t * (gi / (sqrt(sq) + eps_t));        // p.addcdiv_(g, sqrt(square_avg) + eps, -lr)
        } else {
            const T buf = m[i] * hp_t + gi;                        // buf.mul_(momentum).add_(g)
            m[i] = buf;
            p[i] = p[i] - lr_t * buf;
        }
    }
}
template <int KIND, bool ZERO_G>
__global__ void __launch_bounds__(256)
simple_opt_pair_kernel(const AdamGroup a, const AdamGroup b, unsigned nb32, int64_t* __restrict__ step, double lr, double hp,
                       double eps, float gscale) {
    (void)step;
    if (blockIdx.x < nb32)
        simple_opt_body<float, KIND, ZERO_G>((float*)a.p, (float*)a.g, (float*)a.m, a.n, lr, hp, eps, gscale, blockIdx.x, nb32);
    else
        simple_opt_body<double, KIND, ZERO_G>((double*)b.p, (double*)b.g, (double*)b.m, b.n, lr, hp, eps, gscale,
                                              blockIdx.x - nb32, gridDim.x - nb32);
}

// beta of the fixed-beta objective with linear warm-up (train.py:252-258): beta * iter / (0.2 * train_iter) clamped to
// [0, beta]; iter = the device step counter BEFORE this iteration's increment
__global__ void beta_warmup_kernel(const int64_t* __restrict__ step, float beta, float warm_iters, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float b = beta * (float)(*step) / warm_iters;
    *out = fminf(fmaxf(b, 0.f), beta);
}

// train.py:244-246: mse_b = mean over (c, h, w) of (x - recon)^2, rmse_b = sqrt(mse_b); out = (mean_b mse_b, mean_b rmse_b).
// One workgroup per image, then the last workgroup to finish (a counter) averages: fixed order, no float atomics.
__global__ void __launch_bounds__(256)
mse_rmse_kernel(const float* __restrict__ x, const float* __restrict__ r, int B, int n, float* __restrict__ per_image,
                unsigned* __restrict__ counter, float* __restrict__ out) {
    __shared__ double red[4];
    __shared__ bool last;
    const int b = blockIdx.x;
    const float* xb = x + (size_t)b * n;
    const float* rb = r + (size_t)b * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const float d = xb[i] - rb[i]; s += (double)d * d; }
    s = gx_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        per_image[b] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / n);
        __threadfence();
        last = atomicAdd(counter, 1u) == (unsigned)(B - 1);
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        double m = 0.0, rm = 0.0;
        for (int i = 0; i < B; ++i) { const float v = __builtin_nontemporal_load(per_image + i); m += v; rm += sqrtf(v); }
        out[0] = (float)(m / B); out[1] = (float)(rm / B);
        *counter = 0u;                                     // ready for the next launch (graph replay)
    }
}

// ---- the step's noise in one launch: counter-based (Philox4x32-10), keyed by (seed, the device-side step counter): a
//      replayed HIP graph draws fresh numbers every step without the framework's graph-RNG bookkeeping (two offset fills
//      and one launch per tensor).  Element i of tensor t: counter (i / 4, t, step) -> four 32-bit words.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void __launch_bounds__(256)
philox_noise_kernel(float* __restrict__ u, long long nu, float* __restrict__ z, long long nz, unsigned long long seed,
                    const long long* __restrict__ step) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one Philox call = 4 outputs
    const long long qu = (nu + 3) >> 2, qz = (nz + 3) >> 2;
    if (q >= qu + qz) return;
    const unsigned long long st = step ? (unsigned long long)*step : 0ull;
    const bool normal = q >= qu;
    const long long i4 = normal ? q - qu : q;
    unsigned w[4];
    philox4x32_10((unsigned)i4, (unsigned)(i4 >> 32) ^ (normal ? 0x80000000u : 0u), (unsigned)st, (unsigned)(st >> 32),
                  (unsigned)seed, (unsigned)(seed >> 32), w);
    float v[4];
    if (!normal) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)(w[e] >> 8) * 5.9604644775390625e-8f;              // [0, 1): 24 bits
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {                                                               // Box-Muller, two pairs
            const float u1 = ((float)(w[2 * e] >> 8) + 1.f) * 5.9604644775390625e-8f;                // (0, 1]
            const float th = (float)(w[2 * e + 1] >> 8) * (6.283185307179586f * 5.9604644775390625e-8f);
            const float r = sqrtf(-2.f * logf(u1));
            v[2 * e] = r * cosf(th); v[2 * e + 1] = r * sinf(th);
        }
    }
    float* dst = normal ? z : u;
    const long long n = normal ? nz : nu;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * i4 + e < n) dst[4 * i4 + e] = v[e];
}

__global__ void step_inc_kernel(int64_t* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1;
}

// state = {beta, err_ema, initialised (0/1)}.  err: device scalar (batch-mean reconstruction error of
// THIS step, already averaged over ranks).  utils/geco.py:39-49.
__global__ void geco_update_kernel(float* __restrict__ state, const float* __restrict__ err, float goal,
                                   float step_size, float alpha, float speedup, int use_speedup,
                                   float beta_min, float beta_max, int64_t* __restrict__ step) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (step) *step += 1;          // the optimiser's step counter rides along (gx_geco_update_step)
    const float e = *err;
    float ema = state[1];
    if (state[2] == 0.f) { ema = e; state[2] = 1.f; }
    else ema = (1.0f - alpha) * e + alpha * ema;
    const float constraint = goal - ema;
    float factor;
    if (use_speedup && constraint > 0.f) factor = expf(speedup * step_size * constraint);
    else factor = expf(step_size * constraint);
    float beta = factor * state[0];
    beta = fminf(fmaxf(beta, beta_min), beta_max);
    state[0] = beta;
    state[1] = ema;
}

}  // namespace

extern "C" {

int gx_adam_step(void* p, const void* g, void* m, void* v, size_t n, int is_f64, int64_t* step, double lr,
                 double beta1, double beta2, double eps, float grad_scale, gx_stream_t stream) {
    GX_CHECK_ARG(p && g && m && v && step, "gx_adam_step: null pointer");
    if (n == 0) return GX_OK;
    hipStream_t s = (hipStream_t)stream;
    size_t blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    {
        GxProf pf(KID_ADAM, s, 0.0, (is_f64 ? 8.0 : 4.0) * 7.0 * (double)n);  // read p,g,m,v; write p,m,v
        if (is_f64)
            hipLaunchKernelGGL(adam_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, (double*)p,
                               (const double*)g, (double*)m, (double*)v, n, (const int64_t*)step, lr, beta1, beta2,
                               eps, grad_scale);
        else
            hipLaunchKernelGGL(adam_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (float*)p,
                               (const float*)g, (float*)m, (float*)v, n, (const int64_t*)step, lr, beta1, beta2, eps,
                               grad_scale);
    }
    GX_CHECK_LAUNCH("gx_adam_step");
    return GX_OK;
}

/* nu uniform [0, 1) numbers into u and nz standard-normal numbers into z (either may be empty), a function of (seed, *step,
 * position) only: torch.rand / torch.randn of a training step (modules/attention.py:177-178 rand_pixel,
 * models/genesisv2_config.py:157 rsample) in one graph-replayable launch; step may be NULL (= 0). */
int gx_philox_noise(float* u, long long nu, float* z, long long nz, unsigned long long seed, const int64_t* step,
                    gx_stream_t stream) {
    GX_CHECK_ARG(nu >= 0 && nz >= 0 && (nu == 0 || u) && (nz == 0 || z), "gx_philox_noise: bad arguments");
    const long long calls = ((nu + 3) >> 2) + ((nz + 3) >> 2);
    if (calls == 0) return GX_OK;
    hipLaunchKernelGGL(philox_noise_kernel, dim3((unsigned)((calls + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, nu, z, nz,
                       seed, (const long long*)step);
    GX_CHECK_LAUNCH("gx_philox_noise");
    return GX_OK;
}

int gx_step_increment(int64_t* step, gx_stream_t stream) {
    GX_CHECK_ARG(step, "gx_step_increment: null pointer");
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step);
    GX_CHECK_LAUNCH("gx_step_increment");
    return GX_OK;
}

int gx_geco_update(float* state, const float* err, float goal, float step_size, float alpha, float speedup,
                   int use_speedup, float beta_min, float beta_max, gx_stream_t stream) {
    GX_CHECK_ARG(state && err, "gx_geco_update: null pointer");
    {
        GxProf pf(KID_GECO, (hipStream_t)stream, 0.0, 16.0);
        hipLaunchKernelGGL(geco_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, err, goal, step_size,
                           alpha, speedup, use_speedup, beta_min, beta_max, (int64_t*)nullptr);
    }
    GX_CHECK_LAUNCH("gx_geco_update");
    return GX_OK;
}

/* gx_geco_update + gx_step_increment in one launch */
int gx_geco_update_step(float* state, const float* err, float goal, float step_size, float alpha, float speedup,
                        int use_speedup, float beta_min, float beta_max, int64_t* step, gx_stream_t stream) {
    GX_CHECK_ARG(state && err && step, "gx_geco_update_step: null pointer");
    {
        GxProf pf(KID_GECO, (hipStream_t)stream, 0.0, 24.0);
        hipLaunchKernelGGL(geco_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, err, goal, step_size,
                           alpha, speedup, use_speedup, beta_min, beta_max, step);
    }
    GX_CHECK_LAUNCH("gx_geco_update_step");
    return GX_OK;
}

/* The other optimisers of train.py:170-176 on the same flat buffers: kind 1 = torch.optim.RMSprop(lr) (alpha 0.99, eps 1e-8),
 * kind 2 = torch.optim.SGD(lr, momentum 0.9); m = their one state buffer (square average / momentum buffer). */
int gx_optimiser_step_pair(int kind, float* p32, float* g32, float* m32, size_t n32, double* p64, double* g64, double* m64,
                           size_t n64, int64_t* step, double lr, double hp, double eps, float grad_scale, int zero_grads,
                           gx_stream_t stream) {
    GX_CHECK_ARG(kind == 1 || kind == 2, "gx_optimiser_step_pair: kind must be 1 (RMSprop) or 2 (SGD with momentum)");
    GX_CHECK_ARG(p32 && g32 && m32 && n32 > 0, "gx_optimiser_step_pair: null pointer / empty fp32 group");
    GX_CHECK_ARG(n64 == 0 || (p64 && g64 && m64), "gx_optimiser_step_pair: null fp64 pointer");
    hipStream_t s = (hipStream_t)stream;
    size_t nb32 = (n32 + 1023) / 1024, nb64 = n64 ? (n64 + 1023) / 1024 : 0;
    if (nb32 > 2048) nb32 = 2048;
    if (nb64 > 256) nb64 = 256;
    const AdamGroup a{p32, g32, m32, nullptr, n32}, b{p64, g64, m64, nullptr, n64};
    const dim3 grid((unsigned)(nb32 + nb64));
    {
        GxProf pf(KID_ADAM, s, 0.0, 4.0 * 5.0 * (double)n32 + 8.0 * 5.0 * (double)n64);
#define GX_OPT(K, Z) hipLaunchKernelGGL((simple_opt_pair_kernel<K, Z>), grid, dim3(256), 0, s, a, b, (unsigned)nb32, step, lr, hp, eps, grad_scale)
        if (kind == 1) { if (zero_grads) GX_OPT(1, true); else GX_OPT(1, false); }
        else { if (zero_grads) GX_OPT(2, true); else GX_OPT(2, false); }
#undef GX_OPT
    }
    GX_CHECK_LAUNCH("gx_optimiser_step_pair");
    return GX_OK;
}

int gx_beta_warmup(const int64_t* step, float beta, float warm_iters, float* out, gx_stream_t stream) {
    GX_CHECK_ARG(step && out && warm_iters > 0.f, "gx_beta_warmup: bad arguments");
    hipLaunchKernelGGL(beta_warmup_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, beta, warm_iters, out);
    GX_CHECK_LAUNCH("gx_beta_warmup");
    return GX_OK;
}

size_t gx_mse_rmse_ws_bytes(int B) { return ((size_t)B + 4) * sizeof(float); }
int gx_mse_rmse(const float* x, const float* recon, int B, int n, float* out, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(x && recon && out && ws && B > 0 && n > 0, "gx_mse_rmse: bad arguments");
    GX_CHECK_ARG(ws_bytes >= gx_mse_rmse_ws_bytes(B), "gx_mse_rmse: workspace too small");
    unsigned* counter = (unsigned*)ws;                 // must be zero before the FIRST launch (the kernel re-zeroes it)
    float* per_image = (float*)ws + 4;
    {
        GxProf pf(KID_SMALL_REDUCE, (hipStream_t)stream, 0.0, 8.0 * (double)B * n);
        hipLaunchKernelGGL(mse_rmse_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, recon, B, n, per_image, counter, out);
    }
    GX_CHECK_LAUNCH("gx_mse_rmse");
    return GX_OK;
}

/* gx_adam_step on an fp32 group and an fp64 group (n64 may be 0) in one launch; zero_grads != 0: the gradients are
 * zeroed as they are consumed (replaces the bucket's zero-fill launch at the start of the next iteration) */
int gx_adam_step_pair(float* p32, float* g32, float* m32, float* v32, size_t n32, double* p64, double* g64,
                      double* m64, double* v64, size_t n64, int64_t* step, double lr, double beta1, double beta2,
                      double eps, float grad_scale, int zero_grads, gx_stream_t stream) {
    GX_CHECK_ARG(p32 && g32 && m32 && v32 && step && n32 > 0, "gx_adam_step_pair: null pointer / empty fp32 group");
    GX_CHECK_ARG(n64 == 0 || (p64 && g64 && m64 && v64), "gx_adam_step_pair: null fp64 pointer");
    hipStream_t s = (hipStream_t)stream;
    size_t nb32 = (n32 + 1023) / 1024, nb64 = n64 ? (n64 + 1023) / 1024 : 0;
    if (nb32 > 2048) nb32 = 2048;
    if (nb64 > 256) nb64 = 256;
    const AdamGroup a{p32, g32, m32, v32, n32}, b{p64, g64, m64, v64, n64};
    {
        GxProf pf(KID_ADAM, s, 0.0, 4.0 * 7.0 * (double)n32 + 8.0 * 7.0 * (double)n64);
        if (zero_grads)
            hipLaunchKernelGGL(adam_pair_kernel<true>, dim3((unsigned)(nb32 + nb64)), dim3(256), 0, s, a, b, (unsigned)nb32,
                               (const int64_t*)step, lr, beta1, beta2, eps, grad_scale);
        else
            hipLaunchKernelGGL(adam_pair_kernel<false>, dim3((unsigned)(nb32 + nb64)), dim3(256), 0, s, a, b, (unsigned)nb32,
                               (const int64_t*)step, lr, beta1, beta2, eps, grad_scale);
    }
    GX_CHECK_LAUNCH("gx_adam_step_pair");
    return GX_OK;
}

}  // extern "C"
