// GroupNorm(groups, eps, affine) + ReLU, forward and backward, fused with the data movement
// the UNet needs around it (nearest x2 up / x0.5 down sampling and channel-concat placement).
//
// Reference: modules/blocks.py:159-165 (nn.GroupNorm(8, C) + ReLU inside ConvGNReLU),
// modules/unet.py:78,86,89 (F.interpolate nearest 0.5 / 2.0, torch.cat([x_up, skip])),
// models/genesisv2_config.py:91-98 (decoder GroupNorm + ReLU).
//
// HBM-bound.  One workgroup per (image, group) slab of cpg*H*W floats (<= 512 KiB, L2 resident):
// pass 1 accumulates sum / sum-of-squares in fp64 (so mean/var are exact to fp32 rounding and
// independent of the reduction tree), pass 2 re-reads the slab from L2 and writes the normalised,
// rectified tensor straight into up to two destination "views" (a channel slice of a concat
// buffer, optionally 2x up-sampled or 2x down-sampled), so no separate interpolate / cat passes
// exist.  Backward mirrors this: it gathers the incoming gradient from up to two views.
#include "gx_common.h"

namespace {

struct View {
    float* ptr;   // [N, ctot, Hd, Wd]
    int ctot;     // channels of the destination / source buffer
    int c0;       // first channel of this tensor's slice
    int mode;     // 0: same size; 1: buffer is 2x larger (nearest up); 2: buffer is 2x smaller (picks [::2, ::2])
};

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = gx_wave_sum_d(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

__device__ __forceinline__ void store_view(const View& v, int n, int c, int r, int col, int H, int W, float val) {
    if (v.mode == 0) {
        v.ptr[(((size_t)n * v.ctot + v.c0 + c) * H + r) * W + col] = val;
    } else if (v.mode == 1) {
        float* p = v.ptr + (((size_t)n * v.ctot + v.c0 + c) * (2 * H) + 2 * r) * (2 * W) + 2 * col;
        const float2 vv = make_float2(val, val);
        *reinterpret_cast<float2*>(p) = vv;
        *reinterpret_cast<float2*>(p + 2 * W) = vv;
    } else {
        if (((r | col) & 1) == 0)
            v.ptr[(((size_t)n * v.ctot + v.c0 + c) * (H >> 1) + (r >> 1)) * (W >> 1) + (col >> 1)] = val;
    }
}

__device__ __forceinline__ float load_view(const View& v, int n, int c, int r, int col, int H, int W) {
    if (v.mode == 0) {
        return v.ptr[(((size_t)n * v.ctot + v.c0 + c) * H + r) * W + col];
    } else if (v.mode == 1) {
        const float* p = v.ptr + (((size_t)n * v.ctot + v.c0 + c) * (2 * H) + 2 * r) * (2 * W) + 2 * col;
        const float2 a = *reinterpret_cast<const float2*>(p);
        const float2 b = *reinterpret_cast<const float2*>(p + 2 * W);
        return (a.x + a.y) + (b.x + b.y);
    } else {
        if (((r | col) & 1) == 0)
            return v.ptr[(((size_t)n * v.ctot + v.c0 + c) * (H >> 1) + (r >> 1)) * (W >> 1) + (col >> 1)];
        return 0.f;
    }
}

__global__ void __launch_bounds__(512)
gn_relu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                   int C, int H, int W, int groups, float eps, View d0, View d1,
                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double red[16];
    const int n = blockIdx.x / groups, gidx = blockIdx.x % groups;
    const int cpg = C / groups, HW = H * W;
    const int m = cpg * HW;
    const float* slab = y + ((size_t)n * C + (size_t)gidx * cpg) * HW;
    double s = 0.0, ss = 0.0;
    if ((m & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(slab);
        for (int i = threadIdx.x; i < (m >> 2); i += blockDim.x) {
            const float4 v = s4[i];
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    } else {
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const float v = slab[i];
            s += v; ss += (double)v * v;
        }
    }
    s = block_sum_d(s, red);
    ss = block_sum_d(ss, red);
    const double mean = s / m;
    double var = ss / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean;
    const float rstdf = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { mean_out[blockIdx.x] = meanf; rstd_out[blockIdx.x] = rstdf; }
    const int lW = __ffs(W) - 1, lHW = __ffs(HW) - 1;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const int cl = i >> lHW, hw = i & (HW - 1);
        const int c = gidx * cpg + cl;
        const int r = hw >> lW, col = hw & (W - 1);
        float v = (slab[i] - meanf) * rstdf * gamma[c] + beta[c];
        v = v > 0.f ? v : 0.f;
        store_view(d0, n, c, r, col, H, W, v);
        if (d1.ptr) store_view(d1, n, c, r, col, H, W, v);
    }
}

// Backward.  part[n][c][3] = (sum dpre*xhat, sum dpre, sum dy) per (image, channel); a second kernel
// reduces over n in a fixed order.
__global__ void __launch_bounds__(512)
gn_relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                   const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                   int C, int H, int W, int groups, View g0, View g1,
                   float* __restrict__ dy, float* __restrict__ part) {
    __shared__ double red[16];
    __shared__ float ch_part[64][2];  // per channel of the group: sum dpre*xhat, sum dpre  (cpg <= 64)
    const int n = blockIdx.x / groups, gidx = blockIdx.x % groups;
    const int cpg = C / groups, HW = H * W;
    const int m = cpg * HW;
    const float* slab = y + ((size_t)n * C + (size_t)gidx * cpg) * HW;
    float* dslab = dy + ((size_t)n * C + (size_t)gidx * cpg) * HW;
    const float meanf = mean_in[blockIdx.x], rstdf = rstd_in[blockIdx.x];
    const int lW = __ffs(W) - 1;
    // pass 1: per channel sums (channels processed one after another so each needs one block reduction)
    double s1 = 0.0, s2 = 0.0;  // sum dxhat, sum dxhat*xhat over the slab
    for (int cl = 0; cl < cpg; ++cl) {
        const int c = gidx * cpg + cl;
        const float gm = gamma[c], bt = beta[c];
        double a = 0.0, b = 0.0;
        for (int hw = threadIdx.x; hw < HW; hw += blockDim.x) {
            const float xh = (slab[cl * HW + hw] - meanf) * rstdf;
            const float pre = xh * gm + bt;
            if (pre > 0.f) {
                const int r = hw >> lW, col = hw & (W - 1);
                float g = load_view(g0, n, c, r, col, H, W);
                if (g1.ptr) g += load_view(g1, n, c, r, col, H, W);
                a += (double)g * xh;
                b += (double)g;
            }
        }
        a = block_sum_d(a, red);
        b = block_sum_d(b, red);
        if (threadIdx.x == 0) { ch_part[cl][0] = (float)a; ch_part[cl][1] = (float)b; }
        s1 += b * gm;
        s2 += a * gm;
    }
    const float k1 = (float)(s1 / m), k2 = (float)(s2 / m);
    __syncthreads();
    // pass 2: dy = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat*xhat)); per-channel sum of dy (conv bias grad)
    for (int cl = 0; cl < cpg; ++cl) {
        const int c = gidx * cpg + cl;
        const float gm = gamma[c], bt = beta[c];
        double sdy = 0.0;
        for (int hw = threadIdx.x; hw < HW; hw += blockDim.x) {
            const float xh = (slab[cl * HW + hw] - meanf) * rstdf;
            const float pre = xh * gm + bt;
            float dxh = 0.f;
            if (pre > 0.f) {
                const int r = hw >> lW, col = hw & (W - 1);
                float g = load_view(g0, n, c, r, col, H, W);
                if (g1.ptr) g += load_view(g1, n, c, r, col, H, W);
                dxh = g * gm;
            }
            const float d = rstdf * (dxh - k1 - xh * k2);
            dslab[cl * HW + hw] = d;
            sdy += d;
        }
        sdy = block_sum_d(sdy, red);
        if (threadIdx.x == 0) {
            float* p = part + ((size_t)n * C + c) * 3;
            p[0] = ch_part[cl][0]; p[1] = ch_part[cl][1]; p[2] = (float)sdy;
        }
    }
}

__global__ void gn_param_reduce_kernel(const float* __restrict__ part, int N, int C,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ dbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0, d = 0.0;
    for (int n = 0; n < N; ++n) {
        const float* p = part + ((size_t)n * C + c) * 3;
        a += p[0]; b += p[1]; d += p[2];
    }
    dgamma[c] = (float)a;
    dbeta[c] = (float)b;
    if (dbias) dbias[c] = (float)d;
}

int check_view(const char* name, const View& v, int C) {
    GX_CHECK_ARG(v.mode >= 0 && v.mode <= 2, "%s: bad view mode %d", name, v.mode);
    GX_CHECK_ARG(v.c0 >= 0 && v.c0 + C <= v.ctot, "%s: view slice [%d,%d) outside %d channels", name, v.c0,
                 v.c0 + C, v.ctot);
    return GX_OK;
}

}  // namespace

extern "C" {

int gx_gn_relu_fwd(const float* y, const float* gamma, const float* beta, int N, int C, int H, int W, int groups,
                   float eps, float* dst0, int dst0_ctot, int dst0_c0, int dst0_mode, float* dst1, int dst1_ctot,
                   int dst1_c0, int dst1_mode, float* mean, float* rstd, gx_stream_t stream) {
    GX_CHECK_ARG(y && gamma && beta && dst0 && mean && rstd, "gx_gn_relu_fwd: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && groups > 0 && C % groups == 0, "gx_gn_relu_fwd: bad N/C/groups");
    GX_CHECK_ARG(gx_is_pow2(H) && gx_is_pow2(W) && W >= 2 && H >= 2, "gx_gn_relu_fwd: H,W must be powers of two >= 2");
    View d0{dst0, dst0_ctot, dst0_c0, dst0_mode}, d1{dst1, dst1_ctot, dst1_c0, dst1_mode};
    int rc = check_view("gx_gn_relu_fwd", d0, C);
    if (rc) return rc;
    if (dst1) { rc = check_view("gx_gn_relu_fwd", d1, C); if (rc) return rc; }
    const int m = (C / groups) * H * W;
    const int threads = m >= 2048 ? 512 : 256;
    {
        // algorithmic bytes: read y once, write each destination view once
        auto vw = [](int mode) { return mode == 1 ? 4.0 : (mode == 2 ? 0.25 : 1.0); };
        const double el = (double)N * C * H * W;
        GxProf pf(KID_GN_FWD, (hipStream_t)stream, 8.0 * el, 4.0 * el * (1.0 + vw(dst0_mode) + (dst1 ? vw(dst1_mode) : 0.0)));
        hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(N * groups), dim3(threads), 0, (hipStream_t)stream, y, gamma,
                           beta, C, H, W, groups, eps, d0, d1, mean, rstd);
    }
    GX_CHECK_LAUNCH("gx_gn_relu_fwd");
    return GX_OK;
}

size_t gx_gn_relu_bwd_ws_bytes(int N, int C) { return (size_t)N * C * 3 * sizeof(float); }

int gx_gn_relu_bwd(const float* y, const float* gamma, const float* beta, const float* mean, const float* rstd,
                   int N, int C, int H, int W, int groups, const float* g0, int g0_ctot, int g0_c0, int g0_mode,
                   const float* g1, int g1_ctot, int g1_c0, int g1_mode, float* dy, float* dgamma, float* dbeta,
                   float* dbias, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(y && gamma && beta && mean && rstd && g0 && dy && dgamma && dbeta && ws,
                 "gx_gn_relu_bwd: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && groups > 0 && C % groups == 0 && C / groups <= 64,
                 "gx_gn_relu_bwd: bad N/C/groups (channels per group must be <= 64)");
    GX_CHECK_ARG(gx_is_pow2(H) && gx_is_pow2(W) && W >= 2 && H >= 2, "gx_gn_relu_bwd: H,W must be powers of two >= 2");
    GX_CHECK_ARG(ws_bytes >= gx_gn_relu_bwd_ws_bytes(N, C), "gx_gn_relu_bwd: workspace too small");
    View v0{const_cast<float*>(g0), g0_ctot, g0_c0, g0_mode}, v1{const_cast<float*>(g1), g1_ctot, g1_c0, g1_mode};
    int rc = check_view("gx_gn_relu_bwd", v0, C);
    if (rc) return rc;
    if (g1) { rc = check_view("gx_gn_relu_bwd", v1, C); if (rc) return rc; }
    const int m = (C / groups) * H * W;
    const int threads = m >= 2048 ? 512 : 256;
    hipStream_t s = (hipStream_t)stream;
    {
        auto vw = [](int mode) { return mode == 1 ? 4.0 : (mode == 2 ? 0.25 : 1.0); };
        const double el = (double)N * C * H * W;
        GxProf pf(KID_GN_BWD, s, 16.0 * el, 4.0 * el * (2.0 + vw(g0_mode) + (g1 ? vw(g1_mode) : 0.0)));
        hipLaunchKernelGGL(gn_relu_bwd_kernel, dim3(N * groups), dim3(threads), 0, s, y, gamma, beta, mean, rstd, C,
                           H, W, groups, v0, v1, dy, (float*)ws);
    }
    GX_CHECK_LAUNCH("gx_gn_relu_bwd");
    {
        GxProf pf(KID_GN_PARAM_REDUCE, s, 0.0, 12.0 * N * C);
        hipLaunchKernelGGL(gn_param_reduce_kernel, dim3(gx_ceil_div(C, 64)), dim3(64), 0, s, (const float*)ws, N, C,
                           dgamma, dbeta, dbias);
    }
    GX_CHECK_LAUNCH("gx_gn_relu_bwd(reduce)");
    return GX_OK;
}

}  // extern "C"
