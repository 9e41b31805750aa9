// Small supporting kernels of the ComponentVAE path (reference: modules/encoders.py:31-37 MONetCompEncoder,
// modules/decoders.py:25-32 BroadcastDecoder): a generic direct convolution for the tiny strided encoder convs
// (0.5 % of the model's FLOPs -- not worth an MFMA tiling), and the bias/activation backward that turns the
// gradient w.r.t. act(conv + b) into the gradient w.r.t. the conv output plus the bias gradient.
#include "gx_common.h"

namespace {

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : expm1f(v);
    return v;
}
// derivative expressed through the OUTPUT: relu' = [out > 0]; elu' = out > 0 ? 1 : out + 1
__device__ __forceinline__ float act_bwd_from_out(float out, int act) {
    if (act == 1) return out > 0.f ? 1.f : 0.f;
    if (act == 2) return out > 0.f ? 1.f : out + 1.f;
    return 1.f;
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = gx_wave_sum_d(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// dy = g * act'(out); part[n*C + c] = sum_hw dy   (one block per (n, c) plane)
__global__ void __launch_bounds__(256)
bias_act_bwd_kernel(const float* __restrict__ out, const float* __restrict__ g, int HW, int act,
                    float* __restrict__ dy, float* __restrict__ part) {
    __shared__ double red[4];
    const size_t base = (size_t)blockIdx.x * HW;
    double s = 0.0;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float d = g[base + i] * act_bwd_from_out(out[base + i], act);
        dy[base + i] = d;
        s += d;
    }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = (float)s;
}

__global__ void __launch_bounds__(256)
chan_sum_kernel(const float* __restrict__ part, int N, int C, float* __restrict__ out) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) s += part[(size_t)n * C + c];
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) out[c] = (float)s;
}

struct DConv { int N, Cin, Cout, H, W, Ho, Wo, k, stride, pad; };

constexpr int COB = 8;   // output channels per thread
__global__ void __launch_bounds__(256)
dconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                 float* __restrict__ y, DConv d, int act) {
    const int HoWo = d.Ho * d.Wo;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int co0 = blockIdx.y * COB, n = blockIdx.z;
    if (p >= HoWo) return;
    const int oh = p / d.Wo, ow = p - oh * d.Wo;
    float acc[COB];
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[j] = 0.f;
    for (int ci = 0; ci < d.Cin; ++ci) {
        const float* xp = x + ((size_t)n * d.Cin + ci) * d.H * d.W;
        for (int kh = 0; kh < d.k; ++kh) {
            const int ih = oh * d.stride - d.pad + kh;
            if (ih < 0 || ih >= d.H) continue;
            for (int kw = 0; kw < d.k; ++kw) {
                const int iw = ow * d.stride - d.pad + kw;
                if (iw < 0 || iw >= d.W) continue;
                const float xv = xp[ih * d.W + iw];
#pragma unroll
                for (int j = 0; j < COB; ++j)
                    if (co0 + j < d.Cout) acc[j] += xv * w[(((size_t)(co0 + j) * d.Cin + ci) * d.k + kh) * d.k + kw];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < COB; ++j)
        if (co0 + j < d.Cout)
            y[((size_t)n * d.Cout + co0 + j) * HoWo + p] = act_fwd(acc[j] + (bias ? bias[co0 + j] : 0.f), act);
}

// dx[n][ci][ih][iw] = sum_{co,kh,kw} dy[n][co][oh][ow] w[co][ci][kh][kw],  oh*stride - pad + kh == ih
__global__ void __launch_bounds__(256)
dconv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, DConv d) {
    const int HW = d.H * d.W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int ci0 = blockIdx.y * COB, n = blockIdx.z;
    if (p >= HW) return;
    const int ih = p / d.W, iw = p - ih * d.W;
    float acc[COB];
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < d.k; ++kh) {
        const int th = ih + d.pad - kh;
        if (th < 0 || th % d.stride) continue;
        const int oh = th / d.stride;
        if (oh >= d.Ho) continue;
        for (int kw = 0; kw < d.k; ++kw) {
            const int tw = iw + d.pad - kw;
            if (tw < 0 || tw % d.stride) continue;
            const int ow = tw / d.stride;
            if (ow >= d.Wo) continue;
            for (int co = 0; co < d.Cout; ++co) {
                const float g = dy[(((size_t)n * d.Cout + co) * d.Ho + oh) * d.Wo + ow];
#pragma unroll
                for (int j = 0; j < COB; ++j)
                    if (ci0 + j < d.Cin) acc[j] += g * w[(((size_t)co * d.Cin + ci0 + j) * d.k + kh) * d.k + kw];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < COB; ++j)
        if (ci0 + j < d.Cin) dx[((size_t)n * d.Cin + ci0 + j) * HW + p] = acc[j];
}

// dw[co][ci][kh][kw] = sum_{n,oh,ow} dy[n][co][oh][ow] x[n][ci][oh*s-p+kh][ow*s-p+kw]; one block per (co, ci)
constexpr int KKMAX = 25;
__global__ void __launch_bounds__(256)
dconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, DConv d) {
    __shared__ double red[4];
    const int co = blockIdx.x, ci = blockIdx.y;
    const int HoWo = d.Ho * d.Wo;
    const int kk = d.k * d.k;
    float acc[KKMAX];
#pragma unroll
    for (int t = 0; t < KKMAX; ++t) acc[t] = 0.f;
    for (int n = 0; n < d.N; ++n) {
        const float* gp = dy + ((size_t)n * d.Cout + co) * HoWo;
        const float* xp = x + ((size_t)n * d.Cin + ci) * d.H * d.W;
        for (int p = threadIdx.x; p < HoWo; p += blockDim.x) {
            const int oh = p / d.Wo, ow = p - oh * d.Wo;
            const float g = gp[p];
#pragma unroll
            for (int t = 0; t < KKMAX; ++t) {
                if (t < kk) {
                    const int kh = t / d.k, kw = t - kh * d.k;
                    const int ih = oh * d.stride - d.pad + kh, iw = ow * d.stride - d.pad + kw;
                    if (ih >= 0 && ih < d.H && iw >= 0 && iw < d.W) acc[t] += g * xp[ih * d.W + iw];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KKMAX; ++t) {
        if (t < kk) {
            const double s = block_sum_d((double)acc[t], red);
            if (threadIdx.x == 0) dw[((size_t)co * d.Cin + ci) * kk + t] = (float)s;
        }
    }
}

int dconv_geom(const char* name, DConv* d, int N, int Cin, int Cout, int H, int W, int k, int stride, int pad) {
    GX_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad dims", name);
    GX_CHECK_ARG(k >= 1 && k * k <= KKMAX && stride >= 1 && pad >= 0, "%s: kernel <= 5x5, stride >= 1", name);
    d->N = N; d->Cin = Cin; d->Cout = Cout; d->H = H; d->W = W; d->k = k; d->stride = stride; d->pad = pad;
    d->Ho = (H + 2 * pad - k) / stride + 1;
    d->Wo = (W + 2 * pad - k) / stride + 1;
    GX_CHECK_ARG(d->Ho > 0 && d->Wo > 0, "%s: empty output", name);
    return GX_OK;
}

}  // namespace

extern "C" {

size_t gx_bias_act_bwd_ws_bytes(int N, int C) { return (size_t)N * C * sizeof(float); }

int gx_bias_act_bwd(const float* out, const float* g, int N, int C, int H, int W, int act, float* dy, float* dbias,
                    void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(out && g && dy && ws, "gx_bias_act_bwd: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && act >= 0 && act <= 2, "gx_bias_act_bwd: bad dims / act");
    GX_CHECK_ARG(ws_bytes >= gx_bias_act_bwd_ws_bytes(N, C), "gx_bias_act_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_BIAS_ACT_BWD, s, 0.0, 12.0 * N * C * H * W);
        hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(N * C), dim3(256), 0, s, out, g, H * W, act, dy, (float*)ws);
    }
    GX_CHECK_LAUNCH("gx_bias_act_bwd");
    if (dbias) {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * N * C);
        hipLaunchKernelGGL(chan_sum_kernel, dim3(C), dim3(N >= 128 ? 256 : 64), 0, s, (const float*)ws, N, C, dbias);
        GX_CHECK_LAUNCH("gx_bias_act_bwd(reduce)");
    }
    return GX_OK;
}

int gx_conv2d_direct_fwd(const float* x, const float* w, const float* bias, int act, float* y, int N, int Cin,
                         int Cout, int H, int W, int k, int stride, int pad, gx_stream_t stream) {
    GX_CHECK_ARG(x && w && y && act >= 0 && act <= 2, "gx_conv2d_direct_fwd: null pointer / bad act");
    DConv d;
    int rc = dconv_geom("gx_conv2d_direct_fwd", &d, N, Cin, Cout, H, W, k, stride, pad);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * Cin * k * k * d.Ho * d.Wo, 4.0 * N * (Cin * H * W + Cout * d.Ho * d.Wo));
        hipLaunchKernelGGL(dconv_fwd_kernel, dim3(gx_ceil_div(d.Ho * d.Wo, 256), gx_ceil_div(Cout, COB), N), dim3(256), 0,
                           s, x, w, bias, y, d, act);
    }
    GX_CHECK_LAUNCH("gx_conv2d_direct_fwd");
    return GX_OK;
}

int gx_conv2d_direct_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, int k,
                           int stride, int pad, gx_stream_t stream) {
    GX_CHECK_ARG(dy && w && dx, "gx_conv2d_direct_dgrad: null pointer");
    DConv d;
    int rc = dconv_geom("gx_conv2d_direct_dgrad", &d, N, Cin, Cout, H, W, k, stride, pad);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * Cin * k * k * d.Ho * d.Wo, 4.0 * N * (Cin * H * W + Cout * d.Ho * d.Wo));
        hipLaunchKernelGGL(dconv_dgrad_kernel, dim3(gx_ceil_div(H * W, 256), gx_ceil_div(Cin, COB), N), dim3(256), 0, s,
                           dy, w, dx, d);
    }
    GX_CHECK_LAUNCH("gx_conv2d_direct_dgrad");
    return GX_OK;
}

int gx_conv2d_direct_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, int k,
                           int stride, int pad, gx_stream_t stream) {
    GX_CHECK_ARG(x && dy && dw, "gx_conv2d_direct_wgrad: null pointer");
    DConv d;
    int rc = dconv_geom("gx_conv2d_direct_wgrad", &d, N, Cin, Cout, H, W, k, stride, pad);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * Cin * k * k * d.Ho * d.Wo, 4.0 * N * (Cin * H * W + Cout * d.Ho * d.Wo));
        hipLaunchKernelGGL(dconv_wgrad_kernel, dim3(Cout, Cin), dim3(256), 0, s, x, dy, dw, d);
    }
    GX_CHECK_LAUNCH("gx_conv2d_direct_wgrad");
    return GX_OK;
}

}  // extern "C"
