// Stick-breaking mask recursion in log space and the K-way Categorical mask KL.
//
// Stick breaking (reference modules/attention.py:31-51 SimpleSBP, :118-124 LatentSBP, models/monet_config.py:141-152):
//     log_m_t = log_s_t + logsigmoid(l_t),   log_s_{t+1} = log_s_t + logsigmoid(-l_t),   log_s_0 given (or 0)
// optionally with the last mask set to the remaining scope (log_m_{T-1} = log_s_{T-1}; genesis_config.py:167-169).
// One thread per pixel walks the T steps: a scan instead of 4 T element-wise launches (logsigmoid x 2, add x 2).
//
// Mask KL (models/monet_config.py:157-170 kl_m_loss): q = max(exp(log_m), 1e-5), p = max(exp(log_m_r), 1e-5), both
// renormalised by torch.distributions.Categorical, KL(q || p) summed over the pixels of an image.
#include "gx_common.h"

namespace {

// torch's log_sigmoid: min(0, x) - log1p(exp(-|x|))
__device__ __forceinline__ float logsigmoid_f(float x) { return fminf(0.f, x) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(256)
sbp_scan_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ log_s0, int T, size_t P, int last_scope,
                    float* __restrict__ log_m, float* __restrict__ log_s) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float s = log_s0 ? log_s0[p] : 0.f;
    for (int t = 0; t < T; ++t) {
        const float l = logits[(size_t)t * P + p];
        const float m = (last_scope && t == T - 1) ? s : s + logsigmoid_f(l);
        s = s + logsigmoid_f(-l);
        log_m[(size_t)t * P + p] = m;
        log_s[(size_t)t * P + p] = s;
    }
}

// g_m / g_s: gradients w.r.t. log_m[t] / log_s[t] (either may be NULL = zero) -> g_logits [T,P], g_s0 [P] (may be NULL)
__global__ void __launch_bounds__(256)
sbp_scan_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ g_m, const float* __restrict__ g_s,
                    int T, size_t P, int last_scope, float* __restrict__ g_logits, float* __restrict__ g_s0) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float carry = 0.f;      // gradient w.r.t. the scope AFTER step t coming from the later steps
    for (int t = T - 1; t >= 0; --t) {
        const size_t i = (size_t)t * P + p;
        const float sg = sigmoid_f(logits[i]);
        const float ga = carry + (g_s ? g_s[i] : 0.f);          // d / d log_s[t]
        const float gm = g_m ? g_m[i] : 0.f;
        // d logsigmoid(l) / dl = 1 - sigmoid(l);  d logsigmoid(-l) / dl = -sigmoid(l)
        g_logits[i] = ((last_scope && t == T - 1) ? 0.f : gm * (1.f - sg)) - ga * sg;
        carry = gm + ga;                                         // both are "scope before the step" + ...
    }
    if (g_s0) g_s0[p] = carry;
}

// one workgroup per image: kl[b] = sum_pixels sum_k qn (log qn - log pn).  1024 threads, every mask value loaded and
// exponentiated ONCE (the first version -- 256 threads, two passes over the K slots with 2 K exps each -- was a chain of
// 16 dependent load rounds per thread: 72.7 us for 3.7 MB at K = 7, B = 32, 64 x 64; now 4 rounds)
// KB = K bucket (4 / 8 / 16 register slots: a pixel loads only the bucket's slots, the ones past K from slot 0); KB = 0: any K, two
// passes over the slots with the exps recomputed (the values and the order of every sum are the bucketed form's, so the result
// does not depend on the bucket) -- MONet.kl_m_loss has no limit on K_steps (monet_config.py:157-170)
constexpr int CKL_KMAX = 16;
template <int KB>
__global__ void __launch_bounds__(1024)
categorical_kl_fwd_kernel(const float* __restrict__ log_m, const float* __restrict__ log_m_r, int K, int B, int HW,
                          float* __restrict__ kl) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    double acc = 0.0;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        float t = 0.f;
        if constexpr (KB > 0) {
            float q[KB], pn[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const size_t i = ((size_t)(k < K ? k : 0) * B + b) * HW + p;
                q[k] = log_m[i]; pn[k] = log_m_r[i];
            }
            float Q = 0.f, Pn = 0.f;
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (k < K) {
                    q[k] = fmaxf(expf(q[k]), 1e-5f); pn[k] = fmaxf(expf(pn[k]), 1e-5f);
                    Q += q[k]; Pn += pn[k];
                }
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (k < K) {
                    const float qn = q[k] / Q, pv = pn[k] / Pn;
                    t += qn * (logf(qn) - logf(pv));
                }
        } else {
            float Q = 0.f, Pn = 0.f;
            for (int k = 0; k < K; ++k) {
                const size_t i = ((size_t)k * B + b) * HW + p;
                Q += fmaxf(expf(log_m[i]), 1e-5f); Pn += fmaxf(expf(log_m_r[i]), 1e-5f);
            }
            for (int k = 0; k < K; ++k) {
                const size_t i = ((size_t)k * B + b) * HW + p;
                const float qn = fmaxf(expf(log_m[i]), 1e-5f) / Q, pv = fmaxf(expf(log_m_r[i]), 1e-5f) / Pn;
                t += qn * (logf(qn) - logf(pv));
            }
        }
        acc += (double)t;
    }
    acc = gx_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        kl[b] = (float)s;
    }
}

// g_kl [B] -> g_log_m [K,B,HW] and (g_log_m_r != NULL) the gradient through the reconstructed masks
__global__ void __launch_bounds__(256)
categorical_kl_bwd_kernel(const float* __restrict__ log_m, const float* __restrict__ log_m_r,
                          const float* __restrict__ g_kl, int K, int B, int HW, float* __restrict__ g_log_m,
                          float* __restrict__ g_log_m_r) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * HW) return;
    const int b = (int)(idx / HW), p = (int)(idx - (size_t)b * HW);
    float Q = 0.f, Pn = 0.f;
    for (int k = 0; k < K; ++k) {
        const size_t i = ((size_t)k * B + b) * HW + p;
        Q += fmaxf(expf(log_m[i]), 1e-5f);
        Pn += fmaxf(expf(log_m_r[i]), 1e-5f);
    }
    float klp = 0.f;
    for (int k = 0; k < K; ++k) {
        const size_t i = ((size_t)k * B + b) * HW + p;
        const float qn = fmaxf(expf(log_m[i]), 1e-5f) / Q;
        const float pn = fmaxf(expf(log_m_r[i]), 1e-5f) / Pn;
        klp += qn * (logf(qn) - logf(pn));
    }
    const float g = g_kl[b];
    for (int k = 0; k < K; ++k) {
        const size_t i = ((size_t)k * B + b) * HW + p;
        const float a = expf(log_m[i]), bb = expf(log_m_r[i]);
        const float qn = fmaxf(a, 1e-5f) / Q;
        const float pn = fmaxf(bb, 1e-5f) / Pn;
        // dKL/du_j = ((log qn_j - log pn_j) - KL) / Q, u = max(exp(log_m), 1e-5): the clamp passes no gradient below it
        g_log_m[i] = a > 1e-5f ? g * ((logf(qn) - logf(pn)) - klp) / Q * a : 0.f;
        // dKL/dv_j = (1 - qn_j / pn_j) / P
        if (g_log_m_r) g_log_m_r[i] = bb > 1e-5f ? g * (1.f - qn / pn) / Pn * bb : 0.f;
    }
}

// backward of log_softmax over the K slots of the decoder's mask-logit channel (models/genesisv2_config.py:216-218,
// monet_config.py:137-139): g_dec[k][c][p] = (c == C-1) ? g[k] - exp(log_m_r[k]) * sum_j g[j] : 0,  dec [K*B, C, HW]
__global__ void __launch_bounds__(256)
logsoftmax_k_bwd_kernel(const float* __restrict__ log_m_r, const float* __restrict__ g, int K, int B, int HW, int C,
                        float* __restrict__ g_dec) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * HW) return;
    const int b = (int)(idx / HW), p = (int)(idx - (size_t)b * HW);
    float gs = 0.f;
    for (int k = 0; k < K; ++k) gs += g[((size_t)k * B + b) * HW + p];
    for (int k = 0; k < K; ++k) {
        const size_t i = ((size_t)k * B + b) * HW + p;
        float* o = g_dec + ((size_t)k * B + b) * C * HW + p;
        for (int c = 0; c < C - 1; ++c) o[(size_t)c * HW] = 0.f;
        o[(size_t)(C - 1) * HW] = g[i] - expf(log_m_r[i]) * gs;
    }
}

// log_m_r[k][b][p] = log_softmax over the K slots of the decoder's mask-logit channel (monet_config.py:137-139):
// (x - max) - log(sum exp(x - max)), torch's evaluation order
__global__ void __launch_bounds__(256)
logsoftmax_k_fwd_kernel(const float* __restrict__ dec, int K, int B, int HW, int C, float* __restrict__ log_m_r) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * HW) return;
    const int b = (int)(idx / HW), p = (int)(idx - (size_t)b * HW);
    const float* x = dec + ((size_t)b * C + (C - 1)) * HW + p;
    const size_t kstride = (size_t)B * C * HW;
    float mx = x[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k * kstride]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(x[k * kstride] - mx);
    const float lse = logf(se);
    for (int k = 0; k < K; ++k) log_m_r[((size_t)k * B + b) * HW + p] = (x[k * kstride] - mx) - lse;
}

}  // namespace

extern "C" {

int gx_logsoftmax_k_fwd(const float* dec, int K, int B, int HW, int C, float* log_m_r, gx_stream_t stream) {
    GX_CHECK_ARG(dec && log_m_r && K > 0 && B > 0 && HW > 0 && C > 0, "gx_logsoftmax_k_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 8.0 * K * (double)B * HW);
        hipLaunchKernelGGL(logsoftmax_k_fwd_kernel, dim3((unsigned)(((size_t)B * HW + 255) / 256)), dim3(256), 0, s, dec, K,
                           B, HW, C, log_m_r);
    }
    GX_CHECK_LAUNCH("gx_logsoftmax_k_fwd");
    return GX_OK;
}

int gx_logsoftmax_k_bwd(const float* log_m_r, const float* g, int K, int B, int HW, int C, float* g_dec,
                        gx_stream_t stream) {
    GX_CHECK_ARG(log_m_r && g && g_dec && K > 0 && B > 0 && HW > 0 && C > 0, "gx_logsoftmax_k_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * (2.0 + C) * K * (double)B * HW);
        hipLaunchKernelGGL(logsoftmax_k_bwd_kernel, dim3((unsigned)(((size_t)B * HW + 255) / 256)), dim3(256), 0, s,
                           log_m_r, g, K, B, HW, C, g_dec);
    }
    GX_CHECK_LAUNCH("gx_logsoftmax_k_bwd");
    return GX_OK;
}


int gx_sbp_scan_fwd(const float* logits, const float* log_s0, int T, size_t P, int last_scope, float* log_m,
                    float* log_s, gx_stream_t stream) {
    GX_CHECK_ARG(logits && log_m && log_s && T > 0 && P > 0, "gx_sbp_scan_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 12.0 * T * (double)P);
        hipLaunchKernelGGL(sbp_scan_fwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, logits, log_s0, T, P,
                           last_scope, log_m, log_s);
    }
    GX_CHECK_LAUNCH("gx_sbp_scan_fwd");
    return GX_OK;
}

int gx_sbp_scan_bwd(const float* logits, const float* g_log_m, const float* g_log_s, int T, size_t P, int last_scope,
                    float* g_logits, float* g_log_s0, gx_stream_t stream) {
    GX_CHECK_ARG(logits && g_logits && T > 0 && P > 0, "gx_sbp_scan_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 16.0 * T * (double)P);
        hipLaunchKernelGGL(sbp_scan_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, logits, g_log_m,
                           g_log_s, T, P, last_scope, g_logits, g_log_s0);
    }
    GX_CHECK_LAUNCH("gx_sbp_scan_bwd");
    return GX_OK;
}

int gx_categorical_kl_fwd(const float* log_m, const float* log_m_r, int K, int B, int HW, float* kl,
                          gx_stream_t stream) {
    GX_CHECK_ARG(log_m && log_m_r && kl && K > 0 && B > 0 && HW > 0, "gx_categorical_kl_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 8.0 * K * (double)B * HW);
        const dim3 blk(HW >= 1024 ? 1024 : 256);
        if (K <= 4) hipLaunchKernelGGL(categorical_kl_fwd_kernel<4>, dim3(B), blk, 0, s, log_m, log_m_r, K, B, HW, kl);
        else if (K <= 8) hipLaunchKernelGGL(categorical_kl_fwd_kernel<8>, dim3(B), blk, 0, s, log_m, log_m_r, K, B, HW, kl);
        else if (K <= CKL_KMAX) hipLaunchKernelGGL(categorical_kl_fwd_kernel<CKL_KMAX>, dim3(B), blk, 0, s, log_m, log_m_r, K, B, HW, kl);
        else hipLaunchKernelGGL(categorical_kl_fwd_kernel<0>, dim3(B), blk, 0, s, log_m, log_m_r, K, B, HW, kl);
    }
    GX_CHECK_LAUNCH("gx_categorical_kl_fwd");
    return GX_OK;
}

int gx_categorical_kl_bwd(const float* log_m, const float* log_m_r, const float* g_kl, int K, int B, int HW,
                          float* g_log_m, float* g_log_m_r, gx_stream_t stream) {
    GX_CHECK_ARG(log_m && log_m_r && g_kl && g_log_m && K > 0 && B > 0 && HW > 0, "gx_categorical_kl_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 16.0 * K * (double)B * HW);
        hipLaunchKernelGGL(categorical_kl_bwd_kernel, dim3((unsigned)(((size_t)B * HW + 255) / 256)), dim3(256), 0, s,
                           log_m, log_m_r, g_kl, K, B, HW, g_log_m, g_log_m_r);
    }
    GX_CHECK_LAUNCH("gx_categorical_kl_bwd");
    return GX_OK;
}

}  // extern "C"
