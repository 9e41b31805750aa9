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

// part[n*C + c] = sum_hw x   (one block per plane; 16-byte loads when the plane allows)
__global__ void __launch_bounds__(256)
plane_sum_kernel(const float* __restrict__ x, int HW, float* __restrict__ part) {
    __shared__ double red[4];
    const float* p = x + (size_t)blockIdx.x * HW;
    double s = 0.0;
    if ((HW & 3) == 0) {
        for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
            const float4 v = reinterpret_cast<const float4*>(p)[i];
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += p[i];
    }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = (float)s;
}

// Measurement probe: nothing but v_mfma_f32_32x32x2_f32 on register operands, 8 independent accumulator tiles per
// wave (the tap-conv / Winograd inner loops without any operand traffic).  What this sustains is the chip's practical
// fp32-MFMA ceiling under load (clock / power), the number the conv kernels' TF/s should be read against.
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
// MODE 0: register operands only.  1: the B operand of every MFMA comes from LDS (one ds_read_b32 each).
// 2: A and B from LDS (the tap-conv pattern: two reads per MFMA).  3: as 2, plus a workgroup barrier every 32 MFMAs.
template <int MODE>
__global__ void __launch_bounds__(256, 2) mfma_fp32_probe_kernel(int iters, float* __restrict__ out) {
    __shared__ float opnd[8192];
    probe_f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    // eight pseudo-random operand pairs per lane (full mantissa activity: constant operands would flatter the power draw)
    float a[8], b[8];
    unsigned h = 0x9e3779b9u * (threadIdx.x + 1u) + 0x85ebca6bu * (blockIdx.x + 1u);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        a[i] = (float)(h & 0xffffff) * (1.0f / 16777216.0f) - 0.5f;
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        b[i] = (float)(h & 0xffffff) * (1.0f / 16777216.0f) - 0.5f;
    }
    if (MODE != 0) {
        for (int i = threadIdx.x; i < 8192; i += 256) opnd[i] = a[i & 7] + 1e-3f * i;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = 0; it < (MODE == 4 ? 0 : iters); ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float av = a[(i + r) & 7], bv = b[(i + 3 * r) & 7];
                if (MODE >= 1) bv = opnd[((wave * 32 + r * 8 + i) * 64 + lane + (it & 1) * 2048) & 8191];
                if (MODE >= 2) av = opnd[((wave * 32 + r * 8 + i) * 64 + lane + 4096 + (it & 1) * 2048) & 8191];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
            }
        if (MODE == 3) __syncthreads();
    }
    if (MODE == 4) {   // A and B from LDS with ONE 16-byte read each per four MFMAs (k-contiguous operand layout)
        const f32x4* o4 = reinterpret_cast<const f32x4*>(opnd);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 av = o4[((wave * 8 + i) * 64 + lane + (it & 1) * 512) & 2047];
                const f32x4 bv = o4[((wave * 8 + i) * 64 + lane + 1024 + (it & 1) * 512) & 2047];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[(i + r) & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], bv[r], acc[(i + r) & 7], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    if (s == 123.456f) out[0] = s;      // keeps the loop alive
}

// MODE 5 / 6: the inner loop of gx_kq.hip -- per step four 16-byte LDS operand reads (two A, two B fragments of four
// k-contiguous values) issued one step AHEAD of the 16 MFMAs that consume them (2 x 2 tiles of 32 x 32, two register
// sets, order pinned with sched_group_barrier).  6: 2 x 4 tiles (six reads per 32 MFMAs).  Also measures the shader
// clock: out[1] = s_memtime ticks per 100 MHz wall-clock tick of workgroup 0.
template <int NJ>
__global__ void __launch_bounds__(256, 2) mfma_fp32_probe_kq_kernel(int iters, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float opnd[8192];
    probe_f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NJ; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][n][j] = 0.f;
    unsigned h = 0x9e3779b9u * (threadIdx.x + 1u) + 0x85ebca6bu * (blockIdx.x + 1u);
    for (int i = threadIdx.x; i < 8192; i += 256) {
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        opnd[i] = (float)(h & 0xffffff) * (1.0f / 16777216.0f) - 0.5f;
    }
    __syncthreads();
    const long long c0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* o4 = reinterpret_cast<const f32x4*>(opnd);
    const int a_l = lane, b_l = 1024 + wave * 64 + lane;
    f32x4 fa[2][2], fb[2][NJ];
#define GX_PROBE_READ(step_, set_)                                                           \
    {                                                                                         \
        fa[set_][0] = o4[a_l + (((step_) & 7) << 7)];                                         \
        fa[set_][1] = o4[a_l + (((step_) & 7) << 7) + 64];                                    \
        _Pragma("unroll") for (int n = 0; n < NJ; ++n) fb[set_][n] = o4[(b_l + (((step_) & 3) << 8) + n * 16) & 2047]; \
    }
    GX_PROBE_READ(0, 0)
    for (int it = 0; it < iters; it += (NJ == 2 ? 1 : 2)) {     // 32 MFMAs per wave per unit of `iters`
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int cur = st & 1;
            GX_PROBE_READ(2 * it + st + 1, cur ^ 1)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NJ; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m][j], fb[cur][n][j], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 + NJ, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * NJ, 0);
        }
    }
#undef GX_PROBE_READ
    const long long c1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NJ; ++n) s += acc[m][n][0] + acc[m][n][15];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (float)((double)(c1 - c0) / (double)(w1 - w0));
}

}  // namespace

// out[c] = sum_{n,hw} x[n][c][hw] in two fixed-order launches (part: N*C floats of scratch)
int gx_chan_sums_launch(const float* x, int N, int C, int HW, float* part, float* out, hipStream_t s) {
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * N * C * (double)HW);
        hipLaunchKernelGGL(plane_sum_kernel, dim3(N * C), dim3(256), 0, s, x, HW, part);
    }
    GX_CHECK_LAUNCH("gx_chan_sums(planes)");
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * N * C);
        hipLaunchKernelGGL(chan_sum_kernel, dim3(C), dim3(N >= 128 ? 256 : 64), 0, s, (const float*)part, N, C, out);
    }
    GX_CHECK_LAUNCH("gx_chan_sums(reduce)");
    return GX_OK;
}

// out [K*B, 1 + C, HW]: plane 0 of image (k, b) = mask [K,B,HW] plane (k, b), planes 1 .. C = x [B,C,HW] image b (the slot-major
// ComponentVAE input [log_m_k | x] of modules/component_vae.py:59-66 without x.repeat(K) + torch.cat); HW % 4 == 0
__global__ void __launch_bounds__(256)
mask_image_stack_kernel(const float* __restrict__ mask, const float* __restrict__ x, float* __restrict__ out, int B, int C,
                        int HW4) {
    const int plane = blockIdx.y;                       // (k * B + b) * (1 + C) + c
    const int img = plane / (1 + C), c = plane - img * (1 + C);
    const int b = img % B;
    const f32x4* src = reinterpret_cast<const f32x4*>(c == 0 ? mask + (size_t)img * HW4 * 4 : x + ((size_t)b * C + c - 1) * HW4 * 4);
    f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)plane * HW4 * 4);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW4; i += gridDim.x * 256) dst[i] = src[i];
}

extern "C" {

// launches `wgs` workgroups of 4 waves, each wave issuing 32 * iters MFMA 32x32x2 f32; returns the flop count through
// *flops (the caller times the stream).  Measurement only.
int gx_mfma_fp32_probe(int wgs, int iters, int mode, float* scratch, double* flops, gx_stream_t stream) {
    GX_CHECK_ARG(wgs > 0 && iters > 0 && scratch && flops && mode >= 0 && mode <= 6, "gx_mfma_fp32_probe: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(mfma_fp32_probe_kernel<0>, dim3(wgs), dim3(256), 0, s, iters, scratch);
    else if (mode == 1) hipLaunchKernelGGL(mfma_fp32_probe_kernel<1>, dim3(wgs), dim3(256), 0, s, iters, scratch);
    else if (mode == 2) hipLaunchKernelGGL(mfma_fp32_probe_kernel<2>, dim3(wgs), dim3(256), 0, s, iters, scratch);
    else if (mode == 3) hipLaunchKernelGGL(mfma_fp32_probe_kernel<3>, dim3(wgs), dim3(256), 0, s, iters, scratch);
    else if (mode == 4) hipLaunchKernelGGL(mfma_fp32_probe_kernel<4>, dim3(wgs), dim3(256), 0, s, iters, scratch);
    else if (mode == 5) hipLaunchKernelGGL(mfma_fp32_probe_kq_kernel<2>, dim3(wgs), dim3(256), 0, s, iters & ~1, scratch);
    else hipLaunchKernelGGL(mfma_fp32_probe_kq_kernel<4>, dim3(wgs), dim3(256), 0, s, iters & ~1, scratch);
    GX_CHECK_LAUNCH("gx_mfma_fp32_probe");
    *flops = (double)wgs * 4.0 * iters * 32.0 * (2.0 * 32 * 32 * 2);
    return GX_OK;
}

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

int gx_mask_image_stack(const float* mask, const float* x, float* out, int K, int B, int C, int H, int W, gx_stream_t stream) {
    GX_CHECK_ARG(mask && x && out, "gx_mask_image_stack: null pointer");
    GX_CHECK_ARG(K > 0 && B > 0 && C > 0 && H > 0 && W > 0 && (H * W) % 4 == 0 && (long long)K * B * (1 + C) <= 65535,
                 "gx_mask_image_stack: H W %% 4 == 0, K B (1 + C) <= 65535");
    hipStream_t s = (hipStream_t)stream;
    const int HW4 = H * W / 4;
    {
        GxProf pf(KID_BIAS_ACT_BWD, s, 0.0, 8.0 * K * B * (1 + C) * H * W);
        hipLaunchKernelGGL(mask_image_stack_kernel, dim3(gx_ceil_div(HW4, 256) > 4 ? 4 : gx_ceil_div(HW4, 256), K * B * (1 + C)),
                           dim3(256), 0, s, mask, x, out, B, C, HW4);
    }
    GX_CHECK_LAUNCH("gx_mask_image_stack");
    return GX_OK;
}

}  // extern "C"
