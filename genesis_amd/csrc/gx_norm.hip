// GroupNorm(groups, eps, affine) + ReLU, forward and backward, fused with the data movement
// the UNet needs around it (nearest x2 up / x0.5 down sampling and channel-concat placement).
//
// Reference: modules/blocks.py:159-165 (nn.GroupNorm(8, C) + ReLU inside ConvGNReLU),
// modules/unet.py:78,86,89 (F.interpolate nearest 0.5 / 2.0, torch.cat([x_up, skip])),
// models/genesisv2_config.py:91-98 (decoder GroupNorm + ReLU).
//
// HBM-bound.  One workgroup per (image, group) slab of cpg*H*W floats.  Slabs of up to 32 K floats
// with H*W >= 256 are held in registers and read from HBM once (gn_relu_*_reg_kernel below); other
// shapes use the two-pass kernels (<= 512 KiB slab, L2 resident):
// pass 1 accumulates sum / sum-of-squares in fp64 (so mean/var are exact to fp32 rounding and
// independent of the reduction tree), pass 2 re-reads the slab from L2 and writes the normalised,
// rectified tensor straight into up to two destination "views" (a channel slice of a concat
// buffer, optionally 2x up-sampled or 2x down-sampled), so no separate interpolate / cat passes
// exist.  Backward mirrors this: it gathers the incoming gradient from up to two views.
#include "gx_common.h"

#include <vector>

namespace {

struct View {
    float* ptr;   // [N, ctot, Hd, Wd]
    int ctot;     // channels of the destination / source buffer
    int c0;       // first channel of this tensor's slice
    int mode;     // 0: same size; 1: buffer is 2x larger (nearest up); 2: buffer is 2x smaller (picks [::2, ::2])
                  // 3 (gradient source only): ptr is the gradient of a following 1x1 conv's OUTPUT [N, ctot, H, W]
                  //   and the view's value is its data gradient sum_o proj[o][c] * ptr[n][o][r][col], formed on load
    const float* proj;   // mode 3: the 1x1 conv weight [ctot, C]
    int projC;           // mode 3: C (row length of proj)
    const float* pgate;  // mode 3: optional device scalar multiplying the conv output (SemiConv gate), or null
    int nsplit;          // gradient source, modes 0-2: the buffer is still `nsplit` split-K partial slabs (a data gradient whose
    size_t sstride;      //   reduce launch was skipped), `sstride` floats apart, summed on load in slab order; 0 / 1: a plain tensor
};

// Input of the forward kernels: the conv output, possibly still as `nsplit` split-K partial slabs (summed here in
// slab order, exactly as the stand-alone reduce would) plus the conv bias; the summed tensor is written to `ysum`
// (the backward pass needs it).  nsplit == 1, no bias, no ysum: a plain tensor.
struct InSrc {
    const float* p;
    int nsplit;
    size_t stride;        // floats between partial slabs
    const float* bias;    // per channel, may be null
    float* ysum;          // may be null iff nsplit == 1 and bias == null
};

__device__ __forceinline__ f32x4 load_in4(const InSrc& s, size_t off, int c) {
    f32x4 v = *reinterpret_cast<const f32x4*>(s.p + off);
    for (int z = 1; z < s.nsplit; ++z) v += *reinterpret_cast<const f32x4*>(s.p + (size_t)z * s.stride + off);
    if (s.bias) v += s.bias[c];
    if (s.ysum) *reinterpret_cast<f32x4*>(s.ysum + off) = v;
    return v;
}
__device__ __forceinline__ float load_in1(const InSrc& s, size_t off, int c) {
    float v = s.p[off];
    for (int z = 1; z < s.nsplit; ++z) v += s.p[(size_t)z * s.stride + off];
    if (s.bias) v += s.bias[c];
    if (s.ysum) s.ysum[off] = v;
    return v;
}

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
        const float* p = v.ptr + (((size_t)n * v.ctot + v.c0 + c) * H + r) * W + col;
        float o = *p;
        for (int z = 1; z < v.nsplit; ++z) o += p[(size_t)z * v.sstride];
        return o;
    } else if (v.mode == 1) {
        const float* p = v.ptr + (((size_t)n * v.ctot + v.c0 + c) * (2 * H) + 2 * r) * (2 * W) + 2 * col;
        float2 a = *reinterpret_cast<const float2*>(p);
        float2 b = *reinterpret_cast<const float2*>(p + 2 * W);
        for (int z = 1; z < v.nsplit; ++z) {      // (the slabs first, then the 2 x 2 sum: the order of the stand-alone reduce)
            const float2 a2 = *reinterpret_cast<const float2*>(p + (size_t)z * v.sstride);
            const float2 b2 = *reinterpret_cast<const float2*>(p + (size_t)z * v.sstride + 2 * W);
            a.x += a2.x; a.y += a2.y; b.x += b2.x; b.y += b2.y;
        }
        return (a.x + a.y) + (b.x + b.y);
    } else if (v.mode == 2) {
        if (((r | col) & 1) == 0) {
            const float* p = v.ptr + (((size_t)n * v.ctot + v.c0 + c) * (H >> 1) + (r >> 1)) * (W >> 1) + (col >> 1);
            float o = *p;
            for (int z = 1; z < v.nsplit; ++z) o += p[(size_t)z * v.sstride];
            return o;
        }
        return 0.f;
    } else {
        float s = 0.f;
        for (int o = 0; o < v.ctot; ++o)
            s += v.proj[o * v.projC + c] * v.ptr[(((size_t)n * v.ctot + o) * H + r) * W + col];
        return v.pgate ? *v.pgate * s : s;
    }
}

// Reduces NV per-thread doubles over the block at once: wave shuffles, one LDS hop, result broadcast.
template <int NV>
__device__ __forceinline__ void block_sum_multi(double (&v)[NV], double* red /* [16][NV] + [NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = gx_wave_sum_d(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[w * NV + threadIdx.x];
        red[16 * NV + threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[16 * NV + i];
}

__device__ __forceinline__ void store_view4(const View& v, int n, int c, int r, int col, int H, int W, f32x4 val) {
    // col is a multiple of 4
    if (v.mode == 0) {
        *reinterpret_cast<f32x4*>(v.ptr + (((size_t)n * v.ctot + v.c0 + c) * H + r) * W + col) = val;
    } else if (v.mode == 1) {
        float* p = v.ptr + (((size_t)n * v.ctot + v.c0 + c) * (2 * H) + 2 * r) * (2 * W) + 2 * col;
        f32x4 lo, hi;
        lo[0] = val[0]; lo[1] = val[0]; lo[2] = val[1]; lo[3] = val[1];
        hi[0] = val[2]; hi[1] = val[2]; hi[2] = val[3]; hi[3] = val[3];
        *reinterpret_cast<f32x4*>(p) = lo;
        *reinterpret_cast<f32x4*>(p + 4) = hi;
        *reinterpret_cast<f32x4*>(p + 2 * W) = lo;
        *reinterpret_cast<f32x4*>(p + 2 * W + 4) = hi;
    } else {
        if ((r & 1) == 0) {
            float2 o = make_float2(val[0], val[2]);
            *reinterpret_cast<float2*>(v.ptr + (((size_t)n * v.ctot + v.c0 + c) * (H >> 1) + (r >> 1)) * (W >> 1) +
                                       (col >> 1)) = o;
        }
    }
}

__device__ __forceinline__ f32x4 load_view4(const View& v, int n, int c, int r, int col, int H, int W) {
    f32x4 o;
    if (v.mode == 0) {
        const float* p = v.ptr + (((size_t)n * v.ctot + v.c0 + c) * H + r) * W + col;
        o = *reinterpret_cast<const f32x4*>(p);
        for (int z = 1; z < v.nsplit; ++z) o += *reinterpret_cast<const f32x4*>(p + (size_t)z * v.sstride);
    } else if (v.mode == 1) {
        const float* p = v.ptr + (((size_t)n * v.ctot + v.c0 + c) * (2 * H) + 2 * r) * (2 * W) + 2 * col;
        f32x4 a0 = *reinterpret_cast<const f32x4*>(p), a1 = *reinterpret_cast<const f32x4*>(p + 4);
        f32x4 b0 = *reinterpret_cast<const f32x4*>(p + 2 * W), b1 = *reinterpret_cast<const f32x4*>(p + 2 * W + 4);
        for (int z = 1; z < v.nsplit; ++z) {
            const float* q = p + (size_t)z * v.sstride;
            a0 += *reinterpret_cast<const f32x4*>(q); a1 += *reinterpret_cast<const f32x4*>(q + 4);
            b0 += *reinterpret_cast<const f32x4*>(q + 2 * W); b1 += *reinterpret_cast<const f32x4*>(q + 2 * W + 4);
        }
        o[0] = (a0[0] + a0[1]) + (b0[0] + b0[1]);
        o[1] = (a0[2] + a0[3]) + (b0[2] + b0[3]);
        o[2] = (a1[0] + a1[1]) + (b1[0] + b1[1]);
        o[3] = (a1[2] + a1[3]) + (b1[2] + b1[3]);
    } else if (v.mode == 2) {
        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
        if ((r & 1) == 0) {
            const float* p = v.ptr + (((size_t)n * v.ctot + v.c0 + c) * (H >> 1) + (r >> 1)) * (W >> 1) + (col >> 1);
            float2 t = *reinterpret_cast<const float2*>(p);
            for (int z = 1; z < v.nsplit; ++z) {
                const float2 t2 = *reinterpret_cast<const float2*>(p + (size_t)z * v.sstride);
                t.x += t2.x; t.y += t2.y;
            }
            o[0] = t.x; o[2] = t.y;
        }
    } else {
        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
        for (int q = 0; q < v.ctot; ++q)
            o += v.proj[q * v.projC + c] *
                 *reinterpret_cast<const f32x4*>(v.ptr + (((size_t)n * v.ctot + q) * H + r) * W + col);
        if (v.pgate) o = *v.pgate * o;
    }
    return o;
}

// one partial maximum per workgroup for an armed amax tap (gx_amax_tap: the generic two-pass kernels of the 128 x 128 model's
// large slabs serve it too since round 6); am >= 0; every thread of the workgroup calls
__device__ __forceinline__ void gn_block_amax_out(float am, float* __restrict__ amax_parts) {
    __shared__ float amr_[16];
#pragma unroll
    for (int of = 32; of >= 1; of >>= 1) am = fmaxf(am, __shfl_xor(am, of, 64));
    if ((threadIdx.x & 63) == 0) amr_[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
        for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) r = fmaxf(r, amr_[i]);
        amax_parts[blockIdx.x] = r;
    }
}

// VEC: 16-byte accesses along W (needs W % 4 == 0); scalar otherwise (W == 2).
template <bool VEC>
__global__ void __launch_bounds__(1024)
gn_relu_fwd_kernel(const InSrc src, const float* __restrict__ gamma, const float* __restrict__ beta,
                   int C, int H, int W, int groups, float eps, View d0, View d1,
                   float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ amax_parts) {
    __shared__ double red[16 * 2 + 2];
    const int n = blockIdx.x / groups, gidx = blockIdx.x % groups;
    const int cpg = C / groups, HW = H * W;
    const int m = cpg * HW;
    const size_t slab_off = ((size_t)n * C + (size_t)gidx * cpg) * HW;
    float am = 0.f;
    // pass 2 re-reads what pass 1 produced: the summed tensor if one was written (same thread, same elements)
    const float* slab = (src.ysum ? src.ysum : src.p) + slab_off;
    double acc[2] = {0.0, 0.0};
    if (VEC) {
        for (int i = threadIdx.x; i < (m >> 2); i += blockDim.x) {
            const f32x4 v = load_in4(src, slab_off + 4 * (size_t)i, gidx * cpg + (4 * i) / HW);
            acc[0] += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
            acc[1] += ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
        }
    } else {
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const float v = load_in1(src, slab_off + i, gidx * cpg + i / HW);
            acc[0] += v; acc[1] += (double)v * v;
        }
    }
    block_sum_multi<2>(acc, red);
    const double mean = acc[0] / m;
    double var = acc[1] / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean;
    const float rstdf = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { mean_out[blockIdx.x] = meanf; rstd_out[blockIdx.x] = rstdf; }
    if (!d0.ptr) return;                   // statistics only (the consumer normalises on the fly)
    const int lW = __ffs(W) - 1, lHW = __ffs(HW) - 1;
    if (VEC) {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(slab);
        for (int i = threadIdx.x; i < (m >> 2); i += blockDim.x) {
            const int e = i << 2;
            const int cl = e >> lHW, hw = e & (HW - 1);
            const int c = gidx * cpg + cl;
            const int r = hw >> lW, col = hw & (W - 1);
            const float gm = gamma[c], bt = beta[c];
            const f32x4 v = s4[i];
            f32x4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float t = (v[u] - meanf) * rstdf * gm + bt;
                o[u] = t > 0.f ? t : 0.f;
            }
            store_view4(d0, n, c, r, col, H, W, o);
            if (d1.ptr) store_view4(d1, n, c, r, col, H, W, o);
            am = fmaxf(fmaxf(am, fmaxf(o[0], o[1])), fmaxf(o[2], o[3]));
        }
    } else {
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const int cl = i >> lHW, hw = i & (HW - 1);
            const int c = gidx * cpg + cl;
            const int r = hw >> lW, col = hw & (W - 1);
            float v = (slab[i] - meanf) * rstdf * gamma[c] + beta[c];
            v = v > 0.f ? v : 0.f;
            store_view(d0, n, c, r, col, H, W, v);
            if (d1.ptr) store_view(d1, n, c, r, col, H, W, v);
            am = fmaxf(am, v);
        }
    }
    if (amax_parts) gn_block_amax_out(am, amax_parts);      // (uniform)
}

// Backward.  part[n][c][3] = (sum dpre*xhat, sum dpre, sum dy) per (image, channel); a second kernel
// reduces over n in a fixed order.  Channels of the group are processed 8 at a time so that one
// multi-value block reduction serves 8 channels.
constexpr int GCH = 8;
template <bool VEC>
__global__ void __launch_bounds__(1024)
gn_relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                   const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                   int C, int H, int W, int groups, View g0, View g1,
                   float* __restrict__ dy, float* __restrict__ part, float* __restrict__ amax_parts) {
    __shared__ double red[16 * 2 * GCH + 2 * GCH];
    float am = 0.f;
    const int n = blockIdx.x / groups, gidx = blockIdx.x % groups;
    const int cpg = C / groups, HW = H * W;
    const int m = cpg * HW;
    const float* slab = y + ((size_t)n * C + (size_t)gidx * cpg) * HW;
    float* dslab = dy + ((size_t)n * C + (size_t)gidx * cpg) * HW;
    const float meanf = mean_in[blockIdx.x], rstdf = rstd_in[blockIdx.x];
    const int lW = __ffs(W) - 1;
    constexpr int VW = VEC ? 4 : 1;
    // ---- pass 1: per channel a = sum dpre*xhat, b = sum dpre; slab sums s1 = sum dxhat, s2 = sum dxhat*xhat
    double s1 = 0.0, s2 = 0.0;
    for (int cb = 0; cb < cpg; cb += GCH) {
        double ab[2 * GCH];
#pragma unroll
        for (int i = 0; i < 2 * GCH; ++i) ab[i] = 0.0;
#pragma unroll
        for (int cc = 0; cc < GCH; ++cc) {
            const int cl = cb + cc;
            if (cl < cpg) {
                const int c = gidx * cpg + cl;
                const float gm = gamma[c], bt = beta[c];
                for (int hw = threadIdx.x * VW; hw < HW; hw += blockDim.x * VW) {
                    const int r = hw >> lW, col = hw & (W - 1);
                    float xv[4], gv[4];
                    if (VEC) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(slab + cl * HW + hw);
                        f32x4 g = load_view4(g0, n, c, r, col, H, W);
                        if (g1.ptr) { const f32x4 g2 = load_view4(g1, n, c, r, col, H, W); g += g2; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { xv[u] = t[u]; gv[u] = g[u]; }
                    } else {
                        xv[0] = slab[cl * HW + hw];
                        gv[0] = load_view(g0, n, c, r, col, H, W);
                        if (g1.ptr) gv[0] += load_view(g1, n, c, r, col, H, W);
                    }
#pragma unroll
                    for (int u = 0; u < VW; ++u) {
                        const float xh = (xv[u] - meanf) * rstdf;
                        const float pre = xh * gm + bt;
                        if (pre > 0.f) {
                            ab[2 * cc] += (double)gv[u] * xh;
                            ab[2 * cc + 1] += (double)gv[u];
                        }
                    }
                }
            }
        }
        block_sum_multi<2 * GCH>(ab, red);
#pragma unroll
        for (int cc = 0; cc < GCH; ++cc) {
            const int cl = cb + cc;
            if (cl < cpg) {
                const int c = gidx * cpg + cl;
                const float gm = gamma[c];
                s1 += ab[2 * cc + 1] * gm;
                s2 += ab[2 * cc] * gm;
                if (threadIdx.x == 0) {
                    float* p = part + ((size_t)n * C + c) * 3;
                    p[0] = (float)ab[2 * cc]; p[1] = (float)ab[2 * cc + 1];
                }
            }
        }
    }
    const float k1 = (float)(s1 / m), k2 = (float)(s2 / m);
    // ---- pass 2: dy = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat*xhat)); per-channel sum of dy
    for (int cb = 0; cb < cpg; cb += GCH) {
        double sd[GCH];
#pragma unroll
        for (int i = 0; i < GCH; ++i) sd[i] = 0.0;
#pragma unroll
        for (int cc = 0; cc < GCH; ++cc) {
            const int cl = cb + cc;
            if (cl < cpg) {
                const int c = gidx * cpg + cl;
                const float gm = gamma[c], bt = beta[c];
                for (int hw = threadIdx.x * VW; hw < HW; hw += blockDim.x * VW) {
                    const int r = hw >> lW, col = hw & (W - 1);
                    float xv[4], gv[4], ov[4];
                    if (VEC) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(slab + cl * HW + hw);
                        f32x4 g = load_view4(g0, n, c, r, col, H, W);
                        if (g1.ptr) { const f32x4 g2 = load_view4(g1, n, c, r, col, H, W); g += g2; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { xv[u] = t[u]; gv[u] = g[u]; }
                    } else {
                        xv[0] = slab[cl * HW + hw];
                        gv[0] = load_view(g0, n, c, r, col, H, W);
                        if (g1.ptr) gv[0] += load_view(g1, n, c, r, col, H, W);
                    }
#pragma unroll
                    for (int u = 0; u < VW; ++u) {
                        const float xh = (xv[u] - meanf) * rstdf;
                        const float pre = xh * gm + bt;
                        const float dxh = pre > 0.f ? gv[u] * gm : 0.f;
                        const float d = rstdf * (dxh - k1 - xh * k2);
                        ov[u] = d;
                        sd[cc] += d;
                        am = fmaxf(am, fabsf(d));
                    }
                    if (VEC) {
                        f32x4 o; o[0] = ov[0]; o[1] = ov[1]; o[2] = ov[2]; o[3] = ov[3];
                        *reinterpret_cast<f32x4*>(dslab + cl * HW + hw) = o;
                    } else {
                        dslab[cl * HW + hw] = ov[0];
                    }
                }
            }
        }
        block_sum_multi<GCH>(sd, red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int cc = 0; cc < GCH; ++cc)
                if (cb + cc < cpg) part[((size_t)n * C + gidx * cpg + cb + cc) * 3 + 2] = (float)sd[cc];
        }
    }
    if (amax_parts) gn_block_amax_out(am, amax_parts);      // (uniform)
}

// one block per channel: sums part[n][c][0..2] over n in a fixed tree
__global__ void __launch_bounds__(256)
gn_param_reduce_kernel(const float* __restrict__ part, int N, int C,
                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias) {
    __shared__ double red[16 * 3 + 3];
    const int c = blockIdx.x;
    double v[3] = {0.0, 0.0, 0.0};
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* p = part + ((size_t)n * C + c) * 3;
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2];
    }
    block_sum_multi<3>(v, red);
    if (threadIdx.x == 0) {
        dgamma[c] = (float)v[0];
        dbeta[c] = (float)v[1];
        if (dbias) dbias[c] = (float)v[2];
    }
}

// ---------------------------------------------------------------------------------------------
// Register-resident variants: the (image, group) slab is read from HBM exactly once.
// A channel is cut into P parts of HW/P pixels; a "unit" is one part.  Every wave owns UPW units
// and every lane holds F float4 of each unit (F = HW / (256 P)), so per-channel sums are wave
// reductions (no barrier) and only the group statistics cross waves.  Used when HW >= 256 and the
// slab fits (UPW * F <= 8 float4 per lane and tensor); other shapes take the two-pass kernels.
struct RegPlan { int P, F, UPW, threads; bool ok; };

RegPlan plan_reg(int cpg, int H, int W) {
    RegPlan p{1, 1, 1, 64, false};
    const int HW = H * W;
    if ((W % 4) != 0 || HW < 256 || !gx_is_pow2(cpg)) return p;
    p.P = 1;
    while (HW / (256 * p.P) > 8) p.P *= 2;
    p.F = HW / (256 * p.P);
    const int U = cpg * p.P;
    p.UPW = 1;
    while (U / p.UPW > 16) p.UPW *= 2;
    if (p.UPW * p.F > 8 || p.UPW > 2) return p;
    p.threads = 64 * (U / p.UPW);
    p.ok = true;
    return p;
}

template <int F, int UPW>
__global__ void __launch_bounds__(1024)
gn_relu_fwd_reg_kernel(const InSrc src, const float* __restrict__ gamma, const float* __restrict__ beta,
                       int C, int H, int W, int groups, int P, float eps, View d0, View d1,
                       float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ amax_parts) {
    __shared__ double red[16 * 2 + 2];
    __shared__ float amr[16];      // amax_parts (gx_kq_amax_link): per wave, the largest value stored
    const int n = blockIdx.x / groups, gidx = blockIdx.x % groups;
    const int cpg = C / groups, HW = H * W;
    const int m = cpg * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q4 = (HW >> 2) / P;  // float4 per unit
    const size_t slab_off = ((size_t)n * C + (size_t)gidx * cpg) * HW;
    f32x4 xr[UPW][F];
    double acc[2] = {0.0, 0.0};
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int unit = wave * UPW + u;
        const int cl = unit / P, part = unit - cl * P;
#pragma unroll
        for (int j = 0; j < F; ++j)
            xr[u][j] = load_in4(src, slab_off + (size_t)cl * HW + 4 * (size_t)(part * q4 + j * 64 + lane),
                                gidx * cpg + cl);
    }
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const f32x4 v = xr[u][j];
            acc[0] += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
            acc[1] += ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
        }
    block_sum_multi<2>(acc, red);
    const double mean = acc[0] / m;
    double var = acc[1] / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean;
    const float rstdf = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { mean_out[blockIdx.x] = meanf; rstd_out[blockIdx.x] = rstdf; }
    if (!d0.ptr) return;                   // statistics only
    const int lW = __ffs(W) - 1;
    float am = 0.f;
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int unit = wave * UPW + u;
        const int cl = unit / P, part = unit - cl * P;
        const int c = gidx * cpg + cl;
        const float gm = gamma[c], bt = beta[c];
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const int hw = (part * q4 + j * 64 + lane) << 2;
            const int r = hw >> lW, col = hw & (W - 1);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (xr[u][j][e] - meanf) * rstdf * gm + bt;
                o[e] = t > 0.f ? t : 0.f;
            }
            store_view4(d0, n, c, r, col, H, W, o);
            if (d1.ptr) store_view4(d1, n, c, r, col, H, W, o);
            am = fmaxf(fmaxf(am, fmaxf(o[0], o[1])), fmaxf(o[2], o[3]));      // (o >= 0)
        }
    }
    if (amax_parts) {              // (uniform) one partial maximum of the stored activation per workgroup
#pragma unroll
        for (int of = 32; of >= 1; of >>= 1) am = fmaxf(am, __shfl_xor(am, of, 64));
        if (lane == 0) amr[wave] = am;
        __syncthreads();
        if (threadIdx.x == 0) {
            float r = 0.f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, amr[i]);
            amax_parts[blockIdx.x] = r;
        }
    }
}

// sum_q pw[q] * gs[q][i4] from the staged [ctot][HW] output gradient of the following 1x1 conv
__device__ __forceinline__ f32x4 staged_grad(const float* gsl, const float (&pw)[8], int ctot, int HW, int i4) {
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < ctot) g += pw[q] * reinterpret_cast<const f32x4*>(gsl)[q * (HW >> 2) + i4];
    return g;
}

template <int F, int UPW, bool STAGE>
__global__ void __launch_bounds__(1024)
gn_relu_bwd_reg_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                       const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                       int C, int H, int W, int groups, int P, View g0, View g1,
                       float* __restrict__ dy, float* __restrict__ part_out, float* __restrict__ wpart,
                       float* __restrict__ bpart, float* __restrict__ amax_parts) {
    // XCD-aware slab map: the groups of one image -- which all read the same projected-gradient source -- on one XCD's L2
    const int slab_id = gx_xcd_tile(blockIdx.x, gridDim.x);
    __shared__ double uab[32 * 2];  // per unit: sum dpre*xhat, sum dpre
    __shared__ double usd[32];      // per unit: sum dy
    __shared__ float uam[32];       // per unit: max |dy|  (amax_parts, gx_kq_amax_link)
    __shared__ float uw[32][8];     // STAGE + wpart: per unit, sum_p g_out[q][p] * relu(gn(y))[c][p]
    extern __shared__ __attribute__((aligned(16))) float gsl[];   // stage != 0: g0.ptr[n] ([ctot][HW]) of a mode-3 view
    const int n = slab_id / groups, gidx = slab_id % groups;
    const int cpg = C / groups, HW = H * W;
    const int m = cpg * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q4 = (HW >> 2) / P;
    const int U = cpg * P;
    const size_t slab_off = ((size_t)n * C + (size_t)gidx * cpg) * HW;
    const f32x4* slab4 = reinterpret_cast<const f32x4*>(y + slab_off);
    f32x4* dslab4 = reinterpret_cast<f32x4*>(dy + slab_off);
    const float meanf = mean_in[slab_id], rstdf = rstd_in[slab_id];
    const int lW = __ffs(W) - 1;
    f32x4 xr[UPW][F], gr[UPW][F];
    if (STAGE) {
        // the 1x1 conv's output gradient of this image is shared by all channels of the group: once through LDS
        // instead of cpg times through the caches (the y loads are issued first and fly meanwhile)
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int unit = wave * UPW + u;
            const int cl = unit / P, part = unit - cl * P;
#pragma unroll
            for (int j = 0; j < F; ++j) xr[u][j] = slab4[cl * (HW >> 2) + part * q4 + j * 64 + lane];
        }
        const f32x4* src4 = reinterpret_cast<const f32x4*>(g0.ptr + (size_t)n * g0.ctot * HW);
        for (int i = threadIdx.x; i < (g0.ctot * HW) >> 2; i += blockDim.x)
            reinterpret_cast<f32x4*>(gsl)[i] = src4[i];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int unit = wave * UPW + u;
            const int cl = unit / P, part = unit - cl * P;
            const int c = gidx * cpg + cl;
            float pw[8];
            const float gt = g0.pgate ? *g0.pgate : 1.f;
#pragma unroll
            for (int q = 0; q < 8; ++q)    // wave-uniform (c depends on the wave only): keep them in scalar registers
                pw[q] = q < g0.ctot ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(
                                          int, gt * g0.proj[q * g0.projC + c])))
                                    : 0.f;
#pragma unroll
            for (int j = 0; j < F; ++j) gr[u][j] = staged_grad(gsl, pw, g0.ctot, HW, part * q4 + j * 64 + lane);
            if (wpart) {
                // the following 1x1 conv's weight gradient for this channel: its input relu(gn(y)) exists only here
                const float gm = gamma[c], bt = beta[c];
                float wacc[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) wacc[q] = 0.f;
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    f32x4 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = (xr[u][j][e] - meanf) * rstdf * gm + bt;
                        a[e] = t > 0.f ? t : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q < g0.ctot) {
                            const f32x4 gq = reinterpret_cast<const f32x4*>(gsl)[q * (HW >> 2) + part * q4 + j * 64 + lane];
                            wacc[q] += (gq[0] * a[0] + gq[1] * a[1]) + (gq[2] * a[2] + gq[3] * a[3]);
                        }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < g0.ctot) {
                        float v = wacc[q];
#pragma unroll
                        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
                        if (lane == 0) uw[unit][q] = v;
                    }
            }
        }
        if (bpart && gidx == 0) {   // bias gradient partial of the 1x1 conv: sum_p g_out[q][p], one wave per row q
            for (int q = wave; q < g0.ctot; q += (int)(blockDim.x >> 6)) {
                float v = 0.f;
                for (int i = lane; i < (HW >> 2); i += 64) {
                    const f32x4 t = reinterpret_cast<const f32x4*>(gsl)[q * (HW >> 2) + i];
                    v += (t[0] + t[1]) + (t[2] + t[3]);
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) bpart[(size_t)n * g0.ctot + q] = v;
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int unit = wave * UPW + u;
            const int cl = unit / P, part = unit - cl * P;
            const int c = gidx * cpg + cl;
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const int i4 = part * q4 + j * 64 + lane;
                const int hw = i4 << 2;
                const int r = hw >> lW, col = hw & (W - 1);
                xr[u][j] = slab4[cl * (HW >> 2) + i4];
                f32x4 g = load_view4(g0, n, c, r, col, H, W);
                if (g1.ptr) { const f32x4 g2 = load_view4(g1, n, c, r, col, H, W); g += g2; }
                gr[u][j] = g;
            }
        }
    }
    // pass 1 (registers): xr <- xhat, gr <- dpre = g * [pre > 0]
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int unit = wave * UPW + u;
        const int cl = unit / P;
        const int c = gidx * cpg + cl;
        const float gm = gamma[c], bt = beta[c];
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int j = 0; j < F; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xr[u][j][e] - meanf) * rstdf;
                const float pre = xh * gm + bt;
                const float gv = pre > 0.f ? gr[u][j][e] : 0.f;
                xr[u][j][e] = xh;
                gr[u][j][e] = gv;
                a += (double)gv * xh;
                b += (double)gv;
            }
        a = gx_wave_sum_d(a);
        b = gx_wave_sum_d(b);
        if (lane == 0) { uab[2 * unit] = a; uab[2 * unit + 1] = b; }
    }
    __syncthreads();
    if (STAGE && wpart) {
        for (int t = threadIdx.x; t < cpg * g0.ctot; t += blockDim.x) {
            const int cl = t / g0.ctot, q = t - cl * g0.ctot;
            float v = 0.f;
            for (int p = 0; p < P; ++p) v += uw[cl * P + p][q];
            wpart[((size_t)n * g0.ctot + q) * C + gidx * cpg + cl] = v;
        }
    }
    double s1 = 0.0, s2 = 0.0;
    for (int cl = 0; cl < cpg; ++cl) {
        double a = 0.0, b = 0.0;
        for (int p = 0; p < P; ++p) { a += uab[2 * (cl * P + p)]; b += uab[2 * (cl * P + p) + 1]; }
        const float gm = gamma[gidx * cpg + cl];
        s1 += b * gm;
        s2 += a * gm;
        if ((int)threadIdx.x == cl) {
            float* pp = part_out + ((size_t)n * C + gidx * cpg + cl) * 3;
            pp[0] = (float)a; pp[1] = (float)b;
        }
    }
    const float k1 = (float)(s1 / m), k2 = (float)(s2 / m);
    // pass 2 (registers)
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int unit = wave * UPW + u;
        const int cl = unit / P, part = unit - cl * P;
        const float gm = gamma[gidx * cpg + cl];
        double sd = 0.0;
        float am = 0.f;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = rstdf * (gr[u][j][e] * gm - k1 - xr[u][j][e] * k2);
                o[e] = d;
                sd += d;
            }
            am = fmaxf(fmaxf(am, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            dslab4[cl * (HW >> 2) + part * q4 + j * 64 + lane] = o;
        }
        sd = gx_wave_sum_d(sd);
#pragma unroll
        for (int of = 32; of >= 1; of >>= 1) am = fmaxf(am, __shfl_xor(am, of, 64));
        if (lane == 0) { usd[unit] = sd; uam[unit] = am; }
    }
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
        double s = 0.0;
        for (int p = 0; p < P; ++p) s += usd[threadIdx.x * P + p];
        part_out[((size_t)n * C + gidx * cpg + threadIdx.x) * 3 + 2] = (float)s;
    }
    if (amax_parts && threadIdx.x == 0) {
        float r = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6) * UPW; ++i) r = fmaxf(r, uam[i]);
        amax_parts[blockIdx.x] = r;
    }
    (void)U;
}

// The STAGE case at one unit per wave, built for TWO workgroups per CU: only the activations stay in registers (F
// float4), the projected gradient sum_q pw[q] * g_out[q][p] is formed from the LDS-staged rows in pass 1 AND again in
// pass 2 (4 LDS reads + 4 FMAs per float4, cheaper than the 32 registers that would hold it), and the per-thread sums
// run in fp32 over the thread's F float4 (per vector lane: F terms) before they are combined in double.  At <= 64
// VGPRs the second workgroup's loads fly while the first one computes and stores; the generic kernel (128 VGPRs, one
// workgroup per CU) runs its load, compute and store phases back to back: 176 -> see DESIGN.md section 7 (decoder, 224 images).
// KEEP = false (round 5, finding 35): y is NOT held in registers between the two passes but read again -- at 64 registers the 32
// values of a thread plus the passes' temporaries spill (23 registers: 153 MB of scratch written and read back per launch at
// 224 images), and the second read of a 128 KB slab a few microseconds after the first comes out of L2 / MALL, not HBM.
template <int F, int CT, bool WP, bool KEEP>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
gn_relu_bwd_stage_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                         int C, int H, int W, int groups, int P, View g0, View g1_unused,
                         float* __restrict__ dy, float* __restrict__ part_out, float* __restrict__ wpart,
                         float* __restrict__ bpart, float* __restrict__ amax_parts) {
    // XCD-aware slab map: the groups of one image -- which all read the same projected-gradient source -- on one XCD's L2
    const int slab_id = gx_xcd_tile(blockIdx.x, gridDim.x);
    __shared__ double uab[16 * 2];  // per unit: sum dpre*xhat, sum dpre
    __shared__ double usd[16];      // per unit: sum dy
    __shared__ float uam[16];       // per unit: max |dy|  (amax_parts: this slab's partial maximum for the consumer's fp16 scale)
    __shared__ float uw[16][8];     // wpart: per unit, sum_p g_out[q][p] * relu(gn(y))[c][p]
    extern __shared__ __attribute__((aligned(16))) float gsl[];   // g0.ptr[n] ([ctot][HW])
    (void)g1_unused;
    const int n = slab_id / groups, gidx = slab_id % groups;
    const int cpg = C / groups, HW = H * W;
    const int m = cpg * HW;
    const int lane = threadIdx.x & 63, unit = threadIdx.x >> 6;
    const int q4 = (HW >> 2) / P;
    const int cl = unit / P, part = unit - cl * P;
    const int c = gidx * cpg + cl;
    const size_t slab_off = ((size_t)n * C + (size_t)gidx * cpg) * HW;
    const f32x4* src4 = reinterpret_cast<const f32x4*>(y + slab_off) + cl * (HW >> 2) + part * q4 + lane;
    f32x4* dst4 = reinterpret_cast<f32x4*>(dy + slab_off) + cl * (HW >> 2) + part * q4 + lane;
    const f32x4* gs4 = reinterpret_cast<const f32x4*>(gsl) + part * q4 + lane;
    const float meanf = mean_in[slab_id], rstdf = rstd_in[slab_id];
    f32x4 xr[KEEP ? F : 1];
    if constexpr (KEEP) {
#pragma unroll
        for (int j = 0; j < F; ++j) xr[j] = src4[j * 64];
    }
    {
        const f32x4* g4 = reinterpret_cast<const f32x4*>(g0.ptr + (size_t)n * g0.ctot * HW);
        for (int i = threadIdx.x; i < (g0.ctot * HW) >> 2; i += blockDim.x)
            reinterpret_cast<f32x4*>(gsl)[i] = g4[i];
    }
    __syncthreads();
    float pw[CT];
    {
        const float gt = g0.pgate ? *g0.pgate : 1.f;
#pragma unroll
        for (int q = 0; q < CT; ++q)    // wave-uniform (c depends on the wave only): scalar registers
            pw[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, gt * g0.proj[q * g0.projC + c])));
    }
    const float gm = gamma[c], bt = beta[c];
    constexpr int ctot = CT;      // the launcher picks CT == g0.ctot: no runtime channel tests in the loops
    const int rowq = HW >> 2;
    // pass 1: xr <- xhat; sums of dpre * xhat and dpre; the 1x1 conv's weight gradient for this channel
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int j = 0; j < F; ++j) {
        // one float4 at a time: the LDS offset is made to depend on the previous iteration's sums, or the compiler hoists
        // the LDS reads of all F iterations (4 F float4 registers) and the occupancy is gone
        int jo = j * 64;
        asm volatile("" : "+v"(jo) : "v"(sa), "v"(sb));
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        f32x4& xv = xr[KEEP ? j : 0];
        if constexpr (!KEEP) xv = src4[jo];          // (jo depends on the previous iteration's sums: one load in flight per thread;
                                                     //  a second one in flight measured the same, 153 vs 154 us)
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = (xv[e] - meanf) * rstdf;
#pragma unroll
        for (int q = 0; q < CT; ++q) g += pw[q] * gs4[q * rowq + jo];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pre = xv[e] * gm + bt;
            g[e] = pre > 0.f ? g[e] : 0.f;
        }
        sa += (g[0] * xv[0] + g[1] * xv[1]) + (g[2] * xv[2] + g[3] * xv[3]);
        sb += (g[0] + g[1]) + (g[2] + g[3]);
    }
    if (bpart && gidx == 0) {   // bias gradient partial of the 1x1 conv: sum_p g_out[q][p], one wave per row q
        for (int q = unit; q < ctot; q += (int)(blockDim.x >> 6)) {
            float v = 0.f;
            for (int i = lane; i < rowq; i += 64) {
                const f32x4 t = reinterpret_cast<const f32x4*>(gsl)[q * rowq + i];
                v += (t[0] + t[1]) + (t[2] + t[3]);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) bpart[(size_t)n * ctot + q] = v;
        }
    }
    {
        const double a = gx_wave_sum_d((double)sa);
        const double b = gx_wave_sum_d((double)sb);
        if (lane == 0) { uab[2 * unit] = a; uab[2 * unit + 1] = b; }
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    for (int tc = 0; tc < cpg; ++tc) {
        double a = 0.0, b = 0.0;
        for (int p = 0; p < P; ++p) { a += uab[2 * (tc * P + p)]; b += uab[2 * (tc * P + p) + 1]; }
        const float gmc = gamma[gidx * cpg + tc];
        s1 += b * gmc;
        s2 += a * gmc;
        if ((int)threadIdx.x == tc) {
            float* pp = part_out + ((size_t)n * C + gidx * cpg + tc) * 3;
            pp[0] = (float)a; pp[1] = (float)b;
        }
    }
    const float k1 = (float)(s1 / m), k2 = (float)(s2 / m);
    // pass 2: the projected gradient once more from LDS; the following 1x1 conv's weight gradient for this channel
    // (its input relu(gn(y)) exists only here) rides on the same LDS reads
    float sd = 0.f, am = 0.f;
    // opaque copies: otherwise pass 1's pre-activations and masks (4 F + registers) are kept alive for pass 2
    float gm2 = gm, bt2 = bt;
    asm volatile("" : "+v"(gm2), "+v"(bt2));
    float wacc[CT];
#pragma unroll
    for (int q = 0; q < CT; ++q) wacc[q] = 0.f;
#pragma unroll
    for (int j = 0; j < F; ++j) {
        int jo = j * 64;
        static_assert(CT == 4, "the dependency list below names four accumulators");
        if (WP) asm volatile("" : "+v"(jo) : "v"(sd), "v"(am), "v"(k1), "v"(wacc[0]), "v"(wacc[1]), "v"(wacc[2]), "v"(wacc[3]));
        else asm volatile("" : "+v"(jo) : "v"(sd), "v"(am), "v"(k1));
        f32x4 g = {0.f, 0.f, 0.f, 0.f}, o, a;
        f32x4& xv = xr[KEEP ? j : 0];
        if constexpr (!KEEP) {
            xv = src4[jo];
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = (xv[e] - meanf) * rstdf;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = xv[e] * gm2 + bt2;
            a[e] = t > 0.f ? t : 0.f;
        }
#pragma unroll
        for (int q = 0; q < CT; ++q) {
            const f32x4 gq = gs4[q * rowq + jo];
            g += pw[q] * gq;
            if (WP) wacc[q] += (gq[0] * a[0] + gq[1] * a[1]) + (gq[2] * a[2] + gq[3] * a[3]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pre = xv[e] * gm2 + bt2;
            const float gv = pre > 0.f ? g[e] : 0.f;
            const float d = rstdf * (gv * gm2 - k1 - xv[e] * k2);
            o[e] = d;
        }
        sd += (o[0] + o[1]) + (o[2] + o[3]);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        dst4[jo] = o;
    }
    {
        const double v = gx_wave_sum_d((double)sd);
        if (lane == 0) usd[unit] = v;
#pragma unroll
        for (int of = 32; of >= 1; of >>= 1) am = fmaxf(am, __shfl_xor(am, of, 64));
        if (lane == 0) uam[unit] = am;
    }
    if (WP) {
#pragma unroll
        for (int q = 0; q < CT; ++q) {
            float v = wacc[q];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) uw[unit][q] = v;
        }
    }
    __syncthreads();
    if (WP) {
        for (int t = threadIdx.x; t < cpg * ctot; t += blockDim.x) {
            const int tc = t / ctot, q = t - tc * ctot;
            float v = 0.f;
            for (int p = 0; p < P; ++p) v += uw[tc * P + p][q];
            wpart[((size_t)n * ctot + q) * C + gidx * cpg + tc] = v;
        }
    }
    if ((int)threadIdx.x < cpg) {
        double s = 0.0;
        for (int p = 0; p < P; ++p) s += usd[threadIdx.x * P + p];
        part_out[((size_t)n * C + gidx * cpg + threadIdx.x) * 3 + 2] = (float)s;
    }
    if (amax_parts && threadIdx.x == 0) {
        float r = uam[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) r = fmaxf(r, uam[i]);
        amax_parts[blockIdx.x] = r;
    }
}

// ---- tiny slabs (H*W < 256, e.g. the UNet's 4x4 / 8x8 levels): one float4 per thread, the whole (image, group)
// slab lives in the registers of one small workgroup; per-channel sums go through LDS in a fixed order.  The generic
// two-pass kernel needs ~28 us for these (a single wave walking the channels serially); this one ~5 us.
__global__ void __launch_bounds__(1024)
gn_relu_fwd_small_kernel(const InSrc src, const float* __restrict__ gamma, const float* __restrict__ beta,
                         int C, int H, int W, int groups, float eps, View d0, View d1,
                         float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double red[16 * 2 + 2];
    const int n = blockIdx.x / groups, gidx = blockIdx.x % groups;
    const int cpg = C / groups, HW = H * W;
    const int m = cpg * HW;
    const int e = threadIdx.x * 4;
    const bool act = e < m;
    const size_t slab_off = ((size_t)n * C + (size_t)gidx * cpg) * HW;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (act) v = load_in4(src, slab_off + e, gidx * cpg + e / HW);
    double acc[2];
    acc[0] = ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
    acc[1] = ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
    block_sum_multi<2>(acc, red);
    const double mean = acc[0] / m;
    double var = acc[1] / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean;
    const float rstdf = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { mean_out[blockIdx.x] = meanf; rstd_out[blockIdx.x] = rstdf; }
    if (act && d0.ptr) {
        const int cl = e / HW, hw = e - cl * HW;
        const int c = gidx * cpg + cl;
        const int r = hw / W, col = hw - r * W;
        const float gm = gamma[c], bt = beta[c];
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float t = (v[u] - meanf) * rstdf * gm + bt;
            o[u] = t > 0.f ? t : 0.f;
        }
        store_view4(d0, n, c, r, col, H, W, o);
        if (d1.ptr) store_view4(d1, n, c, r, col, H, W, o);
    }
}

__global__ void __launch_bounds__(1024)
gn_relu_bwd_small_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                         int C, int H, int W, int groups, View g0, View g1,
                         float* __restrict__ dy, float* __restrict__ part_out) {
    __shared__ double ta[1024], tb[1024];     // per-thread partials
    __shared__ double ca[64], cb[64];         // per-channel sums (cpg <= 64)
    const int n = blockIdx.x / groups, gidx = blockIdx.x % groups;
    const int cpg = C / groups, HW = H * W;
    const int m = cpg * HW;
    const int tpc = HW >> 2;                  // threads per channel
    const int e = threadIdx.x * 4;
    const bool act = e < m;
    const size_t slab_off = ((size_t)n * C + (size_t)gidx * cpg) * HW;
    const float meanf = mean_in[blockIdx.x], rstdf = rstd_in[blockIdx.x];
    int cl = 0, c = gidx * cpg, r = 0, col = 0;
    f32x4 xh = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
    float gm = 0.f;
    double a = 0.0, b = 0.0;
    if (act) {
        cl = e / HW;
        const int hw = e - cl * HW;
        c = gidx * cpg + cl;
        r = hw / W; col = hw - r * W;
        const f32x4 xv = reinterpret_cast<const f32x4*>(y + slab_off)[threadIdx.x];
        f32x4 g = load_view4(g0, n, c, r, col, H, W);
        if (g1.ptr) { const f32x4 g2 = load_view4(g1, n, c, r, col, H, W); g += g2; }
        gm = gamma[c];
        const float bt = beta[c];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float h = (xv[u] - meanf) * rstdf;
            const float pre = h * gm + bt;
            const float gg = pre > 0.f ? g[u] : 0.f;
            xh[u] = h; gv[u] = gg;
            a += (double)gg * h;
            b += (double)gg;
        }
    }
    ta[threadIdx.x] = a; tb[threadIdx.x] = b;
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
        double sa = 0.0, sb = 0.0;
        for (int i = 0; i < tpc; ++i) { sa += ta[threadIdx.x * tpc + i]; sb += tb[threadIdx.x * tpc + i]; }
        ca[threadIdx.x] = sa; cb[threadIdx.x] = sb;
        float* pp = part_out + ((size_t)n * C + gidx * cpg + threadIdx.x) * 3;
        pp[0] = (float)sa; pp[1] = (float)sb;
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < cpg; ++i) {
        const float gi = gamma[gidx * cpg + i];
        s1 += cb[i] * gi;
        s2 += ca[i] * gi;
    }
    const float k1 = (float)(s1 / m), k2 = (float)(s2 / m);
    double sd = 0.0;
    if (act) {
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d = rstdf * (gv[u] * gm - k1 - xh[u] * k2);
            o[u] = d;
            sd += d;
        }
        reinterpret_cast<f32x4*>(dy + slab_off)[threadIdx.x] = o;
    }
    __syncthreads();            // ta is re-used for the per-thread dy sums
    ta[threadIdx.x] = sd;
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
        double s = 0.0;
        for (int i = 0; i < tpc; ++i) s += ta[threadIdx.x * tpc + i];
        part_out[((size_t)n * C + gidx * cpg + threadIdx.x) * 3 + 2] = (float)s;
    }
}

// threads for the tiny-slab kernels, or 0 if the shape does not qualify
int small_threads(int cpg, int H, int W) {
    const int HW = H * W, m = cpg * HW;
    if (HW >= 256 || (W % 4) != 0 || cpg > 64 || m > 4096) return 0;
    return gx_round_up(m / 4, 64);
}

// an armed amax link (gx_kq_amax_link) is served when the kernel's first destination is a whole plain tensor (no channel slice, no
// resampling): one partial maximum of the stored activation per workgroup for the conv that reads it next
inline float* gn_take_link_out(const float* tensor, int ctot, int c0, int mode, int C, unsigned nwg, size_t covered, bool second_copy = false) {
    GxAmaxLink& L = gx_amax_link();
    const bool link_ok = !second_copy && L.parts && !L.tensor && tensor && c0 == 0 && mode == 0 && ctot == C && covered == L.numel;
    return gx_amax_producer_out(tensor, link_ok, nwg, covered);      // (+ an armed tap, gx_amax_tap: served whatever the destination views are)
}
template <int F, int UPW>
void launch_fwd_reg(dim3 grid, dim3 block, hipStream_t s, const InSrc src, const float* gamma, const float* beta, int C, int H,
                    int W, int groups, int P, float eps, View d0, View d1, float* mean, float* rstd) {
    float* ap = d0.ptr ? gn_take_link_out(d0.ptr, d0.ctot, d0.c0, d0.mode, C, grid.x, (size_t)(grid.x / groups) * C * H * W, d1.ptr != nullptr) : nullptr;
    hipLaunchKernelGGL((gn_relu_fwd_reg_kernel<F, UPW>), grid, block, 0, s, src, gamma, beta, C, H, W, groups, P, eps, d0, d1,
                       mean, rstd, ap);
}
// the two-workgroups-per-CU STAGE kernel serves one unit per wave, 1024 threads and a 4-channel 1x1 conv (RGB + mask logit)
template <int F, int UPW>
bool launch_bwd_stage(dim3 grid, dim3 block, hipStream_t s, size_t lds, const float* y, const float* gamma,
                      const float* beta, const float* mean, const float* rstd, int C, int H, int W, int groups, int P,
                      View g0, View g1, float* dy, float* part, float* wpart, float* bpart) {
    static const char* env = getenv("GENESIS_GN_STAGE2");
    if (UPW != 1 || block.x != 1024 || g0.ctot != 4 || (env && env[0] == '0')) return false;
    // an armed amax link (gx_kq_amax_link): one partial maximum of dy per workgroup for the conv that reads dy next
    float* amax_parts = gn_take_link_out(dy, C, 0, 0, C, grid.x, (size_t)(grid.x / groups) * C * H * W);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_relu_bwd_stage_kernel<F, 4, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_relu_bwd_stage_kernel<F, 4, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_relu_bwd_stage_kernel<F, 4, true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_relu_bwd_stage_kernel<F, 4, false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    // slabs of F >= 8 float4 per thread re-read y in the second pass instead of spilling it (GENESIS_GN_STAGE_KEEP=1: the
    // register-resident form)
    static const char* keep_env = getenv("GENESIS_GN_STAGE_KEEP");
    const bool keep = F < 8 || (keep_env && keep_env[0] == '1');
#define GX_STAGE_LAUNCH(WP_, KEEP_)                                                                                          \
    hipLaunchKernelGGL((gn_relu_bwd_stage_kernel<F, 4, WP_, KEEP_>), grid, block, lds, s, y, gamma, beta, mean, rstd, C, H, W, \
                       groups, P, g0, g1, dy, part, wpart, bpart, amax_parts)
    if (wpart) { if (keep) GX_STAGE_LAUNCH(true, true); else GX_STAGE_LAUNCH(true, false); }
    else { if (keep) GX_STAGE_LAUNCH(false, true); else GX_STAGE_LAUNCH(false, false); }
#undef GX_STAGE_LAUNCH
    return true;
}

template <int F, int UPW>
void launch_bwd_reg(dim3 grid, dim3 block, hipStream_t s, size_t lds, const float* y, const float* gamma, const float* beta,
                    const float* mean, const float* rstd, int C, int H, int W, int groups, int P, View g0, View g1, float* dy,
                    float* part, float* wpart, float* bpart) {
    if (!lds) {
        float* ap = gn_take_link_out(dy, C, 0, 0, C, grid.x, (size_t)(grid.x / groups) * C * H * W);      // (an armed amax link: dy's partial maxima for its consumer)
        hipLaunchKernelGGL((gn_relu_bwd_reg_kernel<F, UPW, false>), grid, block, 0, s, y, gamma, beta, mean, rstd, C, H, W, groups,
                           P, g0, g1, dy, part, wpart, bpart, ap);
        return;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_relu_bwd_reg_kernel<F, UPW, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    if (launch_bwd_stage<F, UPW>(grid, block, s, lds, y, gamma, beta, mean, rstd, C, H, W, groups, P, g0, g1, dy, part, wpart, bpart))
        return;
    float* ap = gn_take_link_out(dy, C, 0, 0, C, grid.x, (size_t)(grid.x / groups) * C * H * W);
    hipLaunchKernelGGL((gn_relu_bwd_reg_kernel<F, UPW, true>), grid, block, lds, s, y, gamma, beta, mean, rstd, C, H, W, groups, P,
                       g0, g1, dy, part, wpart, bpart, ap);
}

#define GX_GN_REG_DISPATCH(LAUNCH, pl, ...)                                      \
    do {                                                                         \
        if (pl.UPW == 1) {                                                       \
            if (pl.F == 1) LAUNCH<1, 1>(__VA_ARGS__);                            \
            else if (pl.F == 2) LAUNCH<2, 1>(__VA_ARGS__);                       \
            else if (pl.F == 4) LAUNCH<4, 1>(__VA_ARGS__);                       \
            else LAUNCH<8, 1>(__VA_ARGS__);                                      \
        } else {                                                                 \
            if (pl.F == 1) LAUNCH<1, 2>(__VA_ARGS__);                            \
            else if (pl.F == 2) LAUNCH<2, 2>(__VA_ARGS__);                       \
            else LAUNCH<4, 2>(__VA_ARGS__);                                      \
        }                                                                        \
    } while (0)

struct GnRedTable { GxGnRed e[48]; };
__global__ void __launch_bounds__(256)
gn_param_reduce_batch_kernel(const GnRedTable tab) {
    const GxGnRed& r = tab.e[blockIdx.y];
    const int c = blockIdx.x;
    if (c >= r.C) return;
    __shared__ double red[16 * 3 + 3];
    double v[3] = {0.0, 0.0, 0.0};
    for (int n = threadIdx.x; n < r.N; n += blockDim.x) {
        const float* p = r.part + ((size_t)n * r.C + c) * 3;
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2];
    }
    block_sum_multi<3>(v, red);
    if (threadIdx.x == 0) {   // accumulates into the zeroed gradient buffers (see wgrad_reduce_batch_kernel)
        r.dgamma[c] += (float)v[0];
        r.dbeta[c] += (float)v[1];
        if (r.dbias) r.dbias[c] += (float)v[2];
    }
}

int check_view(const char* name, const View& v, int C) {
    GX_CHECK_ARG(v.mode >= 0 && v.mode <= 2, "%s: bad view mode %d", name, v.mode);
    GX_CHECK_ARG(v.c0 >= 0 && v.c0 + C <= v.ctot, "%s: view slice [%d,%d) outside %d channels", name, v.c0,
                 v.c0 + C, v.ctot);
    return GX_OK;
}

// ---- projected-gradient backward for slabs that no single workgroup can hold (128 x 128: 8 channels x 16384 pixels = 512 KB):
// the slab is cut into chunks of 4096 pixels, one workgroup per (image, group, chunk).
//   sums kernel    per chunk: a_c = sum dpre xhat, b_c = sum dpre, X_c = sum xhat, the 1x1 conv's weight / bias gradient partials
//   combine kernel per slab: the chunk partials in chunk order (double), k1 / k2, part_out, wpart, bpart
//   apply kernel   per chunk: dy = rstd (dpre gamma - k1 - xhat k2); y and g_out are read a second time
// 3 x |y| + 2 x |g_out| of traffic instead of the generic two-pass kernel's scalar projection (measured at K = 11, 128 x 128,
// B = 32: 2110 us + 540 us for the separate 1x1 weight-gradient pass).  sum_p dy_c follows from the sums: rstd (gamma_c b_c -
// HW k1 - k2 X_c).
constexpr int kSplitPix = 4096;             // pixels per chunk: 1024 float4, four per thread
constexpr int kSplitRec = 3 * 8 + 8 * 8 + 8;    // floats per chunk record: a[8] b[8] X[8] | w[q 8][c 8] | bq[8]

template <int CT>
__global__ void __launch_bounds__(256)
gn_bwd_proj_split_sums_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                              const float* __restrict__ mean_in, const float* __restrict__ rstd_in, int C, int HW, int groups,
                              View g0, float* __restrict__ rec) {
    // all channels' partial sums stay in registers until the chunk has been read: one reduction through LDS per chunk instead of
    // a shuffle tree + two barriers per channel, and channel c + 1's loads are in flight while channel c is summed (the first
    // version ran at 3.2 TB/s against the 5.3 of the apply kernel: every channel's loads waited behind the previous reduction)
    __shared__ float red[4][8][3 + CT];
    const int chunks = HW / kSplitPix;
    const int chunk = blockIdx.x % chunks, slab = blockIdx.x / chunks;
    const int n = slab / groups, gidx = slab % groups;
    const int cpg = C / groups;             // <= 8 (checked by the caller)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float meanf = mean_in[slab], rstdf = rstd_in[slab];
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y + ((size_t)n * C + (size_t)gidx * cpg) * HW) + chunk * (kSplitPix / 4) + tid;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g0.ptr + (size_t)n * g0.ctot * HW) + chunk * (kSplitPix / 4) + tid;
    f32x4 gq[CT][4];
#pragma unroll
    for (int q = 0; q < CT; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) gq[q][j] = q < g0.ctot ? g4[(size_t)q * (HW >> 2) + 256 * j] : f32x4{0.f, 0.f, 0.f, 0.f};
    const float gt = g0.pgate ? *g0.pgate : 1.f;
    float* out = rec + (size_t)blockIdx.x * kSplitRec;
    f32x4 yv[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) yv[0][j] = y4[256 * j];
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) {
        if (cl < cpg) {
            const int cur = cl & 1;
            if (cl + 1 < cpg) {
#pragma unroll
                for (int j = 0; j < 4; ++j) yv[cur ^ 1][j] = y4[(size_t)(cl + 1) * (HW >> 2) + 256 * j];
            }
            const int c = gidx * cpg + cl;
            const float gm = gamma[c], bt = beta[c];
            float pw[CT];
#pragma unroll
            for (int q = 0; q < CT; ++q) pw[q] = q < g0.ctot ? gt * g0.proj[q * g0.projC + c] : 0.f;
            float v[3 + CT];
#pragma unroll
            for (int i = 0; i < 3 + CT; ++i) v[i] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (yv[cur][j][e] - meanf) * rstdf;
                    const float pre = xh * gm + bt;
                    float g = 0.f;
#pragma unroll
                    for (int q = 0; q < CT; ++q) g = fmaf(pw[q], gq[q][j][e], g);
                    const bool on = pre > 0.f;
                    const float gv = on ? g : 0.f, act = on ? pre : 0.f;
                    v[0] = fmaf(gv, xh, v[0]);
                    v[1] += gv;
                    v[2] += xh;
#pragma unroll
                    for (int q = 0; q < CT; ++q) v[3 + q] = fmaf(gq[q][j][e], act, v[3 + q]);
                }
#pragma unroll
            for (int i = 0; i < 3 + CT; ++i) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v[i] += __shfl_xor(v[i], o, 64);
            }
            if (lane == 0)
#pragma unroll
                for (int i = 0; i < 3 + CT; ++i) red[wave][cl][i] = v[i];
        }
    }
    __syncthreads();
    if (tid < 8 * (3 + CT)) {
        const int cl = tid / (3 + CT), i = tid - cl * (3 + CT);
        if (cl < cpg) {
            const float t = (red[0][cl][i] + red[1][cl][i]) + (red[2][cl][i] + red[3][cl][i]);
            if (i < 3) out[i * 8 + cl] = t;                          // a | b | X
            else out[24 + (i - 3) * 8 + cl] = t;                     // w[q][c]
        }
    }
    // bias-gradient partial of the 1x1 conv: sum_p g_out[q][p] (the same for every group: group 0 writes it)
    if (gidx == 0) {
        float b[CT];
#pragma unroll
        for (int q = 0; q < CT; ++q) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) t += (gq[q][j][0] + gq[q][j][1]) + (gq[q][j][2] + gq[q][j][3]);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
            b[q] = t;
        }
        __syncthreads();
        if (lane == 0)
#pragma unroll
            for (int q = 0; q < CT; ++q) red[wave][0][q] = b[q];
        __syncthreads();
        if (tid < CT) out[24 + 64 + tid] = (red[0][0][tid] + red[1][0][tid]) + (red[2][0][tid] + red[3][0][tid]);
    }
}

__global__ void __launch_bounds__(64)
gn_bwd_proj_split_combine_kernel(const float* __restrict__ rec, const float* __restrict__ gamma, const float* __restrict__ rstd_in,
                                 int C, int HW, int groups, int chunks, int ctot, float* __restrict__ kk,
                                 float* __restrict__ part_out, float* __restrict__ wpart, float* __restrict__ bpart) {
    __shared__ double sv[kSplitRec];
    const int slab = blockIdx.x, n = slab / groups, gidx = slab % groups;
    const int cpg = C / groups;
    for (int i = threadIdx.x; i < kSplitRec; i += 64) {
        double t = 0.0;
        for (int ch = 0; ch < chunks; ++ch) t += (double)rec[((size_t)slab * chunks + ch) * kSplitRec + i];
        sv[i] = t;
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    for (int cl = 0; cl < cpg; ++cl) {
        const double gm = gamma[gidx * cpg + cl];
        s1 += sv[8 + cl] * gm;
        s2 += sv[cl] * gm;
    }
    const double m = (double)cpg * HW;
    const double k1 = s1 / m, k2 = s2 / m;
    if (threadIdx.x == 0) { kk[2 * slab] = (float)k1; kk[2 * slab + 1] = (float)k2; }
    if ((int)threadIdx.x < cpg) {
        const int cl = threadIdx.x, c = gidx * cpg + cl;
        float* pp = part_out + ((size_t)n * C + c) * 3;
        pp[0] = (float)sv[cl];
        pp[1] = (float)sv[8 + cl];
        pp[2] = (float)((double)rstd_in[slab] * ((double)gamma[c] * sv[8 + cl] - (double)HW * k1 - k2 * sv[16 + cl]));
    }
    if (wpart)
        for (int t = threadIdx.x; t < cpg * ctot; t += 64) {
            const int cl = t / ctot, q = t - cl * ctot;
            wpart[((size_t)n * ctot + q) * C + gidx * cpg + cl] = (float)sv[24 + q * 8 + cl];
        }
    if (bpart && gidx == 0 && (int)threadIdx.x < ctot) bpart[(size_t)n * ctot + threadIdx.x] = (float)sv[24 + 64 + threadIdx.x];
}

template <int CT>
__global__ void __launch_bounds__(256)
gn_bwd_proj_split_apply_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean_in, const float* __restrict__ rstd_in, int C, int HW, int groups,
                               View g0, const float* __restrict__ kk, float* __restrict__ dy, float* __restrict__ amax_parts) {
    float am = 0.f;
    const int chunks = HW / kSplitPix;
    const int chunk = blockIdx.x % chunks, slab = blockIdx.x / chunks;
    const int n = slab / groups, gidx = slab % groups;
    const int cpg = C / groups;
    const int tid = threadIdx.x;
    const float meanf = mean_in[slab], rstdf = rstd_in[slab];
    const float k1 = kk[2 * slab], k2 = kk[2 * slab + 1];
    const size_t slab_off = ((size_t)n * C + (size_t)gidx * cpg) * HW;
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y + slab_off) + chunk * (kSplitPix / 4) + tid;
    f32x4* d4 = reinterpret_cast<f32x4*>(dy + slab_off) + chunk * (kSplitPix / 4) + tid;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g0.ptr + (size_t)n * g0.ctot * HW) + chunk * (kSplitPix / 4) + tid;
    f32x4 gq[CT][4];
#pragma unroll
    for (int q = 0; q < CT; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) gq[q][j] = q < g0.ctot ? g4[(size_t)q * (HW >> 2) + 256 * j] : f32x4{0.f, 0.f, 0.f, 0.f};
    const float gt = g0.pgate ? *g0.pgate : 1.f;
    for (int cl = 0; cl < cpg; ++cl) {
        const int c = gidx * cpg + cl;
        const float gm = gamma[c], bt = beta[c];
        float pw[CT];
#pragma unroll
        for (int q = 0; q < CT; ++q) pw[q] = q < g0.ctot ? gt * g0.proj[q * g0.projC + c] : 0.f;
        f32x4 yv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = y4[(size_t)cl * (HW >> 2) + 256 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (yv[j][e] - meanf) * rstdf;
                const float pre = xh * gm + bt;
                float g = 0.f;
#pragma unroll
                for (int q = 0; q < CT; ++q) g = fmaf(pw[q], gq[q][j][e], g);
                const float gv = pre > 0.f ? g : 0.f;
                o[e] = rstdf * (gv * gm - k1 - xh * k2);
                am = fmaxf(am, fabsf(o[e]));
            }
            d4[(size_t)cl * (HW >> 2) + 256 * j] = o;
        }
    }
    if (amax_parts) gn_block_amax_out(am, amax_parts);      // (uniform)
}

// shapes of the split path: a projected-gradient source on slabs that the register kernels cannot hold
inline bool proj_split_ok(int C, int H, int W, int groups, int Cout) {
    if (groups <= 0 || C % groups || C / groups > 8 || Cout < 1 || Cout > 8 || (W % 4)) return false;
    const int HW = H * W;
    return HW % kSplitPix == 0 && HW / kSplitPix >= 2 && !plan_reg(C / groups, H, W).ok;
}

// test diagnostic (gx_gn_relu_active_count): one workgroup per (image, channel) plane
__global__ void __launch_bounds__(256)
gn_relu_active_count_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                            const float* __restrict__ mean, const float* __restrict__ rstd, int C, int HW, int groups,
                            unsigned long long* __restrict__ count) {
    const int n = blockIdx.x / C, c = blockIdx.x - n * C;
    const int g = n * groups + c / (C / groups);
    const float meanf = mean[g], rstdf = rstd[g], gm = gamma[c], bt = beta[c];
    const float* p = y + (size_t)blockIdx.x * HW;
    unsigned cnt = 0;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float t = (p[i] - meanf) * rstdf * gm + bt;
        cnt += t > 0.f ? 1u : 0u;
    }
    __shared__ unsigned red[256];
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(count, (unsigned long long)red[0]);
}


}  // namespace

int gx_defer_flush_gn(const GxGnRed* items, int n, hipStream_t s) {
    constexpr int kPerLaunch = 48;      // = the table's capacity (it travels as a kernel argument)
    // records of one launch accumulate from different workgroups: no two of them may share a destination (a GroupNorm used
    // several times per iteration queues one record per use) -- rounds in queue order, as gx_defer_flush_wgrad
    std::vector<int> pending(n);
    for (int i = 0; i < n; ++i) pending[i] = i;
    while (!pending.empty()) {
        std::vector<int> later;
        GnRedTable tab;
        int cnt = 0, maxc = 1;
        double bytes = 0.0;
        for (int idx : pending) {
            const GxGnRed& it = items[idx];
            bool clash = cnt >= kPerLaunch;
            for (int k = 0; k < cnt && !clash; ++k) clash = tab.e[k].dgamma == it.dgamma;
            if (clash) { later.push_back(idx); continue; }
            tab.e[cnt++] = it;
            maxc = it.C > maxc ? it.C : maxc;
            bytes += 12.0 * it.N * it.C;
        }
        {
            GxProf pf(KID_GN_PARAM_REDUCE, s, 0.0, bytes);
            hipLaunchKernelGGL(gn_param_reduce_batch_kernel, dim3(maxc, cnt), dim3(256), 0, s, tab);
        }
        GX_CHECK_LAUNCH("gx_defer_flush(gn)");
        pending.swap(later);
    }
    return GX_OK;
}

extern "C" {

static int gn_relu_fwd_impl(const InSrc& src, const float* gamma, const float* beta, int N, int C, int H, int W,
                            int groups, float eps, float* dst0, int dst0_ctot, int dst0_c0, int dst0_mode,
                            float* dst1, int dst1_ctot, int dst1_c0, int dst1_mode, float* mean, float* rstd,
                            gx_stream_t stream);

int gx_gn_relu_fwd(const float* y, const float* gamma, const float* beta, int N, int C, int H, int W, int groups,
                   float eps, float* dst0, int dst0_ctot, int dst0_c0, int dst0_mode, float* dst1, int dst1_ctot,
                   int dst1_c0, int dst1_mode, float* mean, float* rstd, gx_stream_t stream) {
    const InSrc src{y, 1, 0, nullptr, nullptr};
    return gn_relu_fwd_impl(src, gamma, beta, N, C, H, W, groups, eps, dst0, dst0_ctot, dst0_c0, dst0_mode, dst1,
                            dst1_ctot, dst1_c0, dst1_mode, mean, rstd, stream);
}

int gx_gn_relu_fwd_parts(const float* parts, int nsplit, size_t split_stride, const float* conv_bias, float* y_sum,
                         const float* gamma, const float* beta, int N, int C, int H, int W, int groups, float eps,
                         float* dst0, int dst0_ctot, int dst0_c0, int dst0_mode, float* dst1, int dst1_ctot,
                         int dst1_c0, int dst1_mode, float* mean, float* rstd, gx_stream_t stream) {
    GX_CHECK_ARG(nsplit >= 1, "gx_gn_relu_fwd_parts: nsplit must be >= 1");
    GX_CHECK_ARG(y_sum || (nsplit == 1 && !conv_bias), "gx_gn_relu_fwd_parts: y_sum is required when summing / biasing");
    const InSrc src{parts, nsplit, split_stride, conv_bias, y_sum};
    return gn_relu_fwd_impl(src, gamma, beta, N, C, H, W, groups, eps, dst0, dst0_ctot, dst0_c0, dst0_mode, dst1,
                            dst1_ctot, dst1_c0, dst1_mode, mean, rstd, stream);
}

static int gn_relu_fwd_impl(const InSrc& src, const float* gamma, const float* beta, int N, int C, int H, int W,
                            int groups, float eps, float* dst0, int dst0_ctot, int dst0_c0, int dst0_mode,
                            float* dst1, int dst1_ctot, int dst1_c0, int dst1_mode, float* mean, float* rstd,
                            gx_stream_t stream) {
    const float* y = src.p;
    GX_CHECK_ARG(y && gamma && beta && mean && rstd, "gx_gn_relu_fwd: null pointer");
    GX_CHECK_ARG(dst0 || !dst1, "gx_gn_relu_fwd: dst1 without dst0");
    GX_CHECK_ARG(N > 0 && C > 0 && groups > 0 && C % groups == 0, "gx_gn_relu_fwd: bad N/C/groups");
    GX_CHECK_ARG(gx_is_pow2(H) && gx_is_pow2(W) && W >= 2 && H >= 2, "gx_gn_relu_fwd: H,W must be powers of two >= 2");
    View d0{dst0, dst0_ctot, dst0_c0, dst0_mode, nullptr, 0, nullptr},
         d1{dst1, dst1_ctot, dst1_c0, dst1_mode, nullptr, 0, nullptr};
    int rc = dst0 ? check_view("gx_gn_relu_fwd", d0, C) : GX_OK;    // dst0 == NULL: statistics only
    if (rc) return rc;
    if (dst1) { rc = check_view("gx_gn_relu_fwd", d1, C); if (rc) return rc; }
    const int m = (C / groups) * H * W;
    const int threads = m >= 16384 ? 1024 : (m >= 2048 ? 512 : 256);
    const bool vec = (W % 4) == 0;
    {
        // algorithmic bytes: read y once, write each destination view once
        auto vw = [](int mode) { return mode == 1 ? 4.0 : (mode == 2 ? 0.25 : 1.0); };
        const double el = (double)N * C * H * W;
        GxProf pf(KID_GN_FWD, (hipStream_t)stream, 8.0 * el,
                  4.0 * el * (src.nsplit + (src.ysum ? 1.0 : 0.0) + (dst0 ? vw(dst0_mode) : 0.0) +
                              (dst1 ? vw(dst1_mode) : 0.0)));
        const RegPlan pl = plan_reg(C / groups, H, W);
        const int st = small_threads(C / groups, H, W);
        if (pl.ok)
            GX_GN_REG_DISPATCH(launch_fwd_reg, pl, dim3(N * groups), dim3(pl.threads), (hipStream_t)stream, src, gamma,
                               beta, C, H, W, groups, pl.P, eps, d0, d1, mean, rstd);
        else if (st)
            hipLaunchKernelGGL(gn_relu_fwd_small_kernel, dim3(N * groups), dim3(st), 0, (hipStream_t)stream, src, gamma,
                               beta, C, H, W, groups, eps, d0, d1, mean, rstd);
        else {
            // (an armed amax tap and the link are served by the generic kernels too -- the 128 x 128 model's decoder: K B x 8 = 2816
            //  partial maxima, which every workgroup of the conv that reads them reduces with 16-byte loads)
            float* ap = dst0 ? gn_take_link_out(dst0, d0.ctot, d0.c0, d0.mode, C, (unsigned)(N * groups), (size_t)N * C * H * W, dst1 != nullptr) : nullptr;
            if (vec)
                hipLaunchKernelGGL(gn_relu_fwd_kernel<true>, dim3(N * groups), dim3(threads), 0, (hipStream_t)stream, src,
                                   gamma, beta, C, H, W, groups, eps, d0, d1, mean, rstd, ap);
            else
                hipLaunchKernelGGL(gn_relu_fwd_kernel<false>, dim3(N * groups), dim3(threads), 0, (hipStream_t)stream, src,
                                   gamma, beta, C, H, W, groups, eps, d0, d1, mean, rstd, ap);
        }
    }
    GX_CHECK_LAUNCH("gx_gn_relu_fwd");
    return GX_OK;
}

size_t gx_gn_relu_bwd_ws_bytes(int N, int C) { return (size_t)N * C * 3 * sizeof(float); }
// ... of gx_gn_relu_bwd_proj: + the chunk records and slab constants of the split path (large slabs)
size_t gx_gn_relu_bwd_proj_ws_bytes(int N, int C, int H, int W, int groups, int Cout) {
    size_t b = gx_gn_relu_bwd_ws_bytes(N, C);
    if (proj_split_ok(C, H, W, groups, Cout))
        b += ((size_t)N * groups * (H * W / kSplitPix) * kSplitRec + (size_t)N * groups * 2) * sizeof(float);
    return b;
}

static int gn_relu_bwd_impl(const float* y, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, int N, int C, int H, int W, int groups, const View& v0, const View& v1,
                            float* dy, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes,
                            gx_stream_t stream, float* wpart = nullptr, float* bpart = nullptr);

int gx_gn_relu_bwd(const float* y, const float* gamma, const float* beta, const float* mean, const float* rstd,
                   int N, int C, int H, int W, int groups, const float* g0, int g0_ctot, int g0_c0, int g0_mode,
                   const float* g1, int g1_ctot, int g1_c0, int g1_mode, float* dy, float* dgamma, float* dbeta,
                   float* dbias, void* ws, size_t ws_bytes, gx_stream_t stream) {
    return gx_gn_relu_bwd_parts(y, gamma, beta, mean, rstd, N, C, H, W, groups, g0, g0_ctot, g0_c0, g0_mode, 1, 0, g1, g1_ctot,
                                g1_c0, g1_mode, 1, 0, dy, dgamma, dbeta, dbias, ws, ws_bytes, stream);
}

int gx_gn_relu_bwd_parts(const float* y, const float* gamma, const float* beta, const float* mean, const float* rstd,
                         int N, int C, int H, int W, int groups, const float* g0, int g0_ctot, int g0_c0, int g0_mode,
                         int g0_nsplit, size_t g0_split_stride, const float* g1, int g1_ctot, int g1_c0, int g1_mode,
                         int g1_nsplit, size_t g1_split_stride, float* dy, float* dgamma, float* dbeta, float* dbias, void* ws,
                         size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(g0, "gx_gn_relu_bwd: null pointer");
    GX_CHECK_ARG(g0_nsplit >= 1 && g1_nsplit >= 1, "gx_gn_relu_bwd_parts: nsplit >= 1");
    View v0{const_cast<float*>(g0), g0_ctot, g0_c0, g0_mode, nullptr, 0, nullptr, g0_nsplit, g0_split_stride},
         v1{const_cast<float*>(g1), g1_ctot, g1_c0, g1_mode, nullptr, 0, nullptr, g1_nsplit, g1_split_stride};
    int rc = check_view("gx_gn_relu_bwd", v0, C);
    if (rc) return rc;
    if (g1) { rc = check_view("gx_gn_relu_bwd", v1, C); if (rc) return rc; }
    return gn_relu_bwd_impl(y, gamma, beta, mean, rstd, N, C, H, W, groups, v0, v1, dy, dgamma, dbeta, dbias, ws,
                            ws_bytes, stream);
}

int gx_gn_relu_bwd_proj(const float* y, const float* gamma, const float* beta, const float* mean, const float* rstd,
                        int N, int C, int H, int W, int groups, const float* g_out, int Cout, const float* w,
                        const float* gate, float* dy, float* dgamma, float* dbeta, float* dbias, float* wpart,
                        float* bpart, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(g_out && w, "gx_gn_relu_bwd_proj: null pointer");
    GX_CHECK_ARG((wpart == nullptr) == (bpart == nullptr), "gx_gn_relu_bwd_proj: wpart and bpart go together");
    GX_CHECK_ARG(Cout >= 1 && Cout <= 8, "gx_gn_relu_bwd_proj: Cout must be 1..8 (got %d)", Cout);
    GX_CHECK_ARG((W % 4) == 0 && ((uintptr_t)g_out % 16) == 0, "gx_gn_relu_bwd_proj: W %% 4 == 0 and 16-byte alignment");
    View v0{const_cast<float*>(g_out), Cout, 0, 3, w, C, gate}, v1{nullptr, 0, 0, 0, nullptr, 0, nullptr};
    return gn_relu_bwd_impl(y, gamma, beta, mean, rstd, N, C, H, W, groups, v0, v1, dy, dgamma, dbeta, dbias, ws,
                            ws_bytes, stream, wpart, bpart);
}

// whether gx_gn_relu_bwd_proj can also produce the 1x1 conv's weight-gradient partials (wpart / bpart)
int gx_gn_relu_bwd_proj_fuses_wgrad(int C, int H, int W, int groups, int Cout) {
    if (C <= 0 || groups <= 0 || C % groups || Cout < 1 || Cout > 8 || (C / groups) * Cout > 1024) return 0;
    if (!gx_is_pow2(H) || !gx_is_pow2(W)) return 0;
    if (proj_split_ok(C, H, W, groups, Cout)) return 1;
    return plan_reg(C / groups, H, W).ok && (size_t)Cout * H * W * 4 <= 128 * 1024;
}

static int gn_relu_bwd_impl(const float* y, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, int N, int C, int H, int W, int groups, const View& v0, const View& v1,
                            float* dy, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes,
                            gx_stream_t stream, float* wpart, float* bpart) {
    const float* g1 = v1.ptr;
    const int g0_mode = v0.mode, g1_mode = v1.mode;
    GX_CHECK_ARG(y && gamma && beta && mean && rstd && dy && dgamma && dbeta && ws,
                 "gx_gn_relu_bwd: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && groups > 0 && C % groups == 0,
                 "gx_gn_relu_bwd: bad N/C/groups");
    GX_CHECK_ARG(gx_is_pow2(H) && gx_is_pow2(W) && W >= 2 && H >= 2, "gx_gn_relu_bwd: H,W must be powers of two >= 2");
    GX_CHECK_ARG(ws_bytes >= gx_gn_relu_bwd_ws_bytes(N, C), "gx_gn_relu_bwd: workspace too small");
    const int hw = H * W;
    const bool vec = (W % 4) == 0;
    const int threads = hw >= 4096 ? 1024 : (hw >= 1024 ? 256 : 64);
    hipStream_t s = (hipStream_t)stream;
    {
        auto vw = [](int mode) { return mode == 1 ? 4.0 : (mode == 2 ? 0.25 : (mode == 3 ? 0.0 : 1.0)); };
        const double el = (double)N * C * H * W;
        GxProf pf(KID_GN_BWD, s, 16.0 * el, 4.0 * el * (2.0 + vw(g0_mode) + (g1 ? vw(g1_mode) : 0.0)));
        const RegPlan pl = plan_reg(C / groups, H, W);
        const int st = small_threads(C / groups, H, W);
        if (v0.mode == 3 && !g1 && proj_split_ok(C, H, W, groups, v0.ctot) &&
            ws_bytes >= gx_gn_relu_bwd_proj_ws_bytes(N, C, H, W, groups, v0.ctot)) {
            // slabs beyond one workgroup's registers (128 x 128): chunked sums -> per-slab constants -> chunked apply
            const int chunks = hw / kSplitPix;
            float* rec = (float*)ws + (size_t)N * C * 3;
            float* kk = rec + (size_t)N * groups * chunks * kSplitRec;
            const dim3 grid(N * groups * chunks);
            if (v0.ctot <= 4)
                hipLaunchKernelGGL(gn_bwd_proj_split_sums_kernel<4>, grid, dim3(256), 0, s, y, gamma, beta, mean, rstd, C, hw,
                                   groups, v0, rec);
            else
                hipLaunchKernelGGL(gn_bwd_proj_split_sums_kernel<8>, grid, dim3(256), 0, s, y, gamma, beta, mean, rstd, C, hw,
                                   groups, v0, rec);
            hipLaunchKernelGGL(gn_bwd_proj_split_combine_kernel, dim3(N * groups), dim3(64), 0, s, (const float*)rec, gamma, rstd,
                               C, hw, groups, chunks, v0.ctot, kk, (float*)ws, wpart, bpart);
            float* ap = gn_take_link_out(dy, C, 0, 0, C, grid.x, (size_t)N * C * H * W);      // (dy: a whole plain tensor)
            if (v0.ctot <= 4)
                hipLaunchKernelGGL(gn_bwd_proj_split_apply_kernel<4>, grid, dim3(256), 0, s, y, gamma, beta, mean, rstd, C, hw,
                                   groups, v0, (const float*)kk, dy, ap);
            else
                hipLaunchKernelGGL(gn_bwd_proj_split_apply_kernel<8>, grid, dim3(256), 0, s, y, gamma, beta, mean, rstd, C, hw,
                                   groups, v0, (const float*)kk, dy, ap);
        }
        else if (pl.ok) {
            // projected-gradient source: stage the image's [Cout][H*W] output gradient in LDS when it fits
            const size_t stage = (v0.mode == 3 && (size_t)v0.ctot * hw * 4 <= 128 * 1024) ? (size_t)v0.ctot * hw * 4 : 0;
            GX_CHECK_ARG(!wpart || stage, "gx_gn_relu_bwd_proj: fused 1x1 weight gradient unsupported for this shape");
            GX_GN_REG_DISPATCH(launch_bwd_reg, pl, dim3(N * groups), dim3(pl.threads), s, stage, y, gamma, beta, mean,
                               rstd, C, H, W, groups, pl.P, v0, v1, dy, (float*)ws, wpart, bpart);
        }
        else if (wpart)
            GX_CHECK_ARG(false, "gx_gn_relu_bwd_proj: fused 1x1 weight gradient unsupported for this shape");
        else if (st)
            hipLaunchKernelGGL(gn_relu_bwd_small_kernel, dim3(N * groups), dim3(st), 0, s, y, gamma, beta, mean, rstd,
                               C, H, W, groups, v0, v1, dy, (float*)ws);
        else {
            float* ap = gx_amax_producer_out(dy, false, (unsigned)(N * groups), (size_t)N * C * H * W);
            if (vec)
                hipLaunchKernelGGL(gn_relu_bwd_kernel<true>, dim3(N * groups), dim3(threads), 0, s, y, gamma, beta, mean,
                                   rstd, C, H, W, groups, v0, v1, dy, (float*)ws, ap);
            else
                hipLaunchKernelGGL(gn_relu_bwd_kernel<false>, dim3(N * groups), dim3(threads), 0, s, y, gamma, beta, mean,
                                   rstd, C, H, W, groups, v0, v1, dy, (float*)ws, ap);
        }
    }
    GX_CHECK_LAUNCH("gx_gn_relu_bwd");
    if (g_gx_defer_on) {
        const GxGnRed rec{(const float*)ws, dgamma, dbeta, dbias, N, C};
        if (gx_defer_push_gn(rec)) return GX_OK;
        return gx_defer_flush_gn(&rec, 1, s);       // queue full: reduce now, ACCUMULATING like the batched flush
    }
    {
        GxProf pf(KID_GN_PARAM_REDUCE, s, 0.0, 12.0 * N * C);
        hipLaunchKernelGGL(gn_param_reduce_kernel, dim3(C), dim3(N >= 128 ? 256 : 64), 0, s, (const float*)ws, N, C,
                           dgamma, dbeta, dbias);
    }
    GX_CHECK_LAUNCH("gx_gn_relu_bwd(reduce)");
    return GX_OK;
}


/* gx_gn_relu_active_count: how many elements of relu(gn(y)) are > 0 -- evaluated with the expression every forward / backward
 * kernel of this file (and the normalise-on-load 1x1 conv) uses for its ReLU decision, t = (y - mean) * rstd * gamma + beta, so the
 * count is the KERNELS' own pattern (a test diagnostic: tests/test_fullbatch_gpu.py compares it with the reference's). */
int gx_gn_relu_active_count(const float* y, const float* gamma, const float* beta, const float* mean, const float* rstd, int N,
                            int C, int H, int W, int groups, unsigned long long* count, gx_stream_t stream) {
    GX_CHECK_ARG(y && gamma && beta && mean && rstd && count && N > 0 && C > 0 && groups > 0 && C % groups == 0 && H > 0 && W > 0,
                 "gx_gn_relu_active_count: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipMemsetAsync(count, 0, sizeof(unsigned long long), s);
    hipLaunchKernelGGL(gn_relu_active_count_kernel, dim3(N * C), dim3(256), 0, s, y, gamma, beta, mean, rstd, C, H * W, groups,
                       count);
    GX_CHECK_LAUNCH("gx_gn_relu_active_count");
    return GX_OK;
}

}  // extern "C"
