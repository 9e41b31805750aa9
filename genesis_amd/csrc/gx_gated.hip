// Gated (de)convolution unit of the sylvester VAE: out = norm_h(h + b_h) * sigmoid(norm_g(g + b_g)), where
// [h | g] = chunk(conv output [N, 2C, H, W], 2, dim=1) and norm is BatchNorm2d (training-mode batch statistics),
// InstanceNorm2d(affine) or nothing.  Reference: third_party/sylvester/layers.py:40-54,87-101.
//
// HBM-bound.  Statistics are per "unit": a channel over (N, H, W) for BatchNorm, an (image, channel) plane for
// InstanceNorm; one workgroup per unit, fp64 accumulation, fixed reduction trees (deterministic).
#include "gx_common.h"

namespace {

enum { NORM_NONE = 0, NORM_BN = 1, NORM_IN = 2 };

template <int NV>
__device__ __forceinline__ void block_sum_multi(double (&v)[NV], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
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

// unit u -> (channel ch in [0, 2C), first image n0, image count): BN: u = ch, all N images; IN: u = n*2C + ch.
__device__ __forceinline__ void unit_decode(int u, int norm, int N, int C2, int* ch, int* n0, int* ncount) {
    if (norm == NORM_BN) { *ch = u; *n0 = 0; *ncount = N; }
    else { *ch = u % C2; *n0 = u / C2; *ncount = 1; }
}

// Per-unit statistics in two steps so that a BatchNorm channel (all N images) is read by many workgroups:
// part[(u * nchunk + z)] = {sum, sum of squares} of (y + bias) over the images of chunk z (fp64), then
// stats[u] = {mean, rstd} from the chunk partials in a fixed order.
__global__ void __launch_bounds__(256)
gated_stats_partial_kernel(const float* __restrict__ y, const float* __restrict__ bias, int N, int C2, int HW,
                           int norm, int nchunk, double* __restrict__ part) {
    __shared__ double red[16 * 2 + 2];
    int ch, n0, nc;
    unit_decode(blockIdx.x, norm, N, C2, &ch, &n0, &nc);
    const int per = (nc + nchunk - 1) / nchunk;
    const int na = n0 + blockIdx.y * per;
    int nb = na + per;
    if (nb > n0 + nc) nb = n0 + nc;
    const float b = bias ? bias[ch] : 0.f;
    double acc[2] = {0.0, 0.0};
    for (int n = na; n < nb; ++n) {
        const float* p = y + ((size_t)n * C2 + ch) * HW;
        if ((HW & 3) == 0) {
            const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
            for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
                const f32x4 t = p4[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double v = (double)(t[e] + b); acc[0] += v; acc[1] += v * v; }
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += blockDim.x) {
                const double v = (double)(p[i] + b);
                acc[0] += v; acc[1] += v * v;
            }
        }
    }
    block_sum_multi<2>(acc, red);
    if (threadIdx.x == 0) {
        part[2 * ((size_t)blockIdx.x * nchunk + blockIdx.y)] = acc[0];
        part[2 * ((size_t)blockIdx.x * nchunk + blockIdx.y) + 1] = acc[1];
    }
}

__global__ void gated_stats_finalize_kernel(const double* __restrict__ part, int units, int nchunk, double m, float eps,
                                            float* __restrict__ stats) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    double s1 = 0.0, s2 = 0.0;
    for (int z = 0; z < nchunk; ++z) { s1 += part[2 * ((size_t)u * nchunk + z)]; s2 += part[2 * ((size_t)u * nchunk + z) + 1]; }
    const double mean = s1 / m;
    double var = s2 / m - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * u] = (float)mean;
    stats[2 * u + 1] = (float)(1.0 / sqrt(var + (double)eps));
    // biased variance kept for the caller's running_var update (BatchNorm uses the unbiased one there)
}

// Cross-replica BatchNorm (the SyncBN-style exchange): the statistics pass ends in raw fp64 {sum, sum of squares}
// per channel -- what the ranks add up -- and {mean, rstd} are formed from the SUMMED pairs and the global count.
__global__ void gated_raw_sums_kernel(const double* __restrict__ part, int units, int nchunk, double* __restrict__ sums) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    double s1 = 0.0, s2 = 0.0;
    for (int z = 0; z < nchunk; ++z) { s1 += part[2 * ((size_t)u * nchunk + z)]; s2 += part[2 * ((size_t)u * nchunk + z) + 1]; }
    sums[2 * u] = s1; sums[2 * u + 1] = s2;
}

__global__ void gated_stats_from_sums_kernel(const double* __restrict__ sums, int units, double m, float eps,
                                             float* __restrict__ stats) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    const double mean = sums[2 * u] / m;
    double var = sums[2 * u + 1] / m - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * u] = (float)mean;
    stats[2 * u + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// nn.BatchNorm2d's running statistics of the two norms of a gated unit from its {mean, rstd} pairs: running = (1 - mom)
// running + mom {mean, unbiased variance}, num_batches_tracked += 1 (one launch instead of ~19 pointwise ones)
__global__ void bn_running_update_kernel(const float* __restrict__ stats, int C, double m, double eps, float mom,
                                         float* __restrict__ rm_h, float* __restrict__ rv_h, float* __restrict__ rm_g,
                                         float* __restrict__ rv_g, long long* __restrict__ nbt_h, long long* __restrict__ nbt_g) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u == 0) { *nbt_h += 1; *nbt_g += 1; }
    if (u >= 2 * C) return;
    const float mean = stats[2 * u];
    const double rstd = (double)stats[2 * u + 1];
    const float var = (float)((1.0 / (rstd * rstd) - eps) * (m / (m > 1.0 ? m - 1.0 : 1.0)));
    float* rm = u < C ? rm_h + u : rm_g + (u - C);
    float* rv = u < C ? rv_h + u : rv_g + (u - C);
    *rm = *rm * (1.f - mom) + mom * mean;
    *rv = *rv * (1.f - mom) + mom * var;
}

__device__ __forceinline__ void unit_stats(const float* stats, int norm, int n, int ch, int C2, float* mean, float* rstd) {
    if (norm == NORM_NONE) { *mean = 0.f; *rstd = 1.f; return; }
    const int u = (norm == NORM_BN) ? ch : n * C2 + ch;
    *mean = stats[2 * u]; *rstd = stats[2 * u + 1];
}

// one partial maximum per workgroup of what it stored, for an armed amax tap / link (gx_amax_tap, gx_kq_amax_link): the 5 x 5 convs
// that read the tensor next run on fp16 pieces and need its scale; am >= 0; every thread of the workgroup calls
__device__ __forceinline__ void gated_block_amax_out(float am, float* __restrict__ amax_parts) {
    __shared__ float amr_[4];
#pragma unroll
    for (int of = 32; of >= 1; of >>= 1) am = fmaxf(am, __shfl_xor(am, of, 64));
    if ((threadIdx.x & 63) == 0) amr_[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) amax_parts[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(amr_[0], amr_[1]), fmaxf(amr_[2], amr_[3]));
}

// the fused forward (round 6): the apply kernel's workgroups form their two units' {mean, rstd} from the statistics pass's chunk
// partials themselves (in the finalize kernel's order and arithmetic: the same bits), the first workgroup of a unit stores them to
// `stats` (the backward pass reads them) and, under BatchNorm, applies nn.BatchNorm2d's running-statistics update -- two launches
// less per unit (gated_stats_finalize_kernel, bn_running_update_kernel: ~5 us each, 20 of them per GENESIS step)
struct GatedFuse {
    const double* part;       // NULL: `stats` holds {mean, rstd} already (the cross-replica path; norm none)
    int nchunk;
    double m;                 // elements per unit
    float eps;
    float* stats_out;
    float* rm_h; float* rv_h; float* rm_g; float* rv_g; long long* nbt_h; long long* nbt_g;     // running statistics (BatchNorm) or NULL
    float mom;
};

// out[n][c][p] = A_h * sigmoid(A_g),  A = ((y + b) - mean) * rstd * gamma + beta
__global__ void __launch_bounds__(256)
gated_apply_kernel(const float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ stats,
                   const float* __restrict__ gh, const float* __restrict__ bh, const float* __restrict__ gg,
                   const float* __restrict__ bg, int N, int C, int HW, int norm, float* __restrict__ out,
                   float* __restrict__ amax_parts, const GatedFuse fu) {
    const int plane = blockIdx.x;           // n * C + c
    const int n = plane / C, c = plane % C;
    const int C2 = 2 * C;
    float mh, rh, mg, rg;
    if (fu.part) {                          // (uniform)
        __shared__ float st4[4];
        if (threadIdx.x < 2) {
            const int ch = threadIdx.x ? C + c : c;
            const size_t u = norm == NORM_BN ? (size_t)ch : (size_t)n * C2 + ch;
            double s1 = 0.0, s2 = 0.0;
            for (int z = 0; z < fu.nchunk; ++z) { s1 += fu.part[2 * (u * fu.nchunk + z)]; s2 += fu.part[2 * (u * fu.nchunk + z) + 1]; }
            const double mean = s1 / fu.m;
            double var = s2 / fu.m - mean * mean;
            if (var < 0.0) var = 0.0;
            const float meanf = (float)mean, rstdf = (float)(1.0 / sqrt(var + (double)fu.eps));
            st4[2 * threadIdx.x] = meanf; st4[2 * threadIdx.x + 1] = rstdf;
            if (blockIdx.y == 0 && (norm != NORM_BN || n == 0)) {
                fu.stats_out[2 * u] = meanf; fu.stats_out[2 * u + 1] = rstdf;
                if (norm == NORM_BN && fu.rm_h) {       // (bn_running_update_kernel's arithmetic, from the stored pair)
                    const double rstd = (double)rstdf, m = fu.m, eps = (double)fu.eps;
                    const float varu = (float)((1.0 / (rstd * rstd) - eps) * (m / (m > 1.0 ? m - 1.0 : 1.0)));
                    float* rm = threadIdx.x ? fu.rm_g + c : fu.rm_h + c;
                    float* rv = threadIdx.x ? fu.rv_g + c : fu.rv_h + c;
                    *rm = *rm * (1.f - fu.mom) + fu.mom * meanf;
                    *rv = *rv * (1.f - fu.mom) + fu.mom * varu;
                    if (c == 0 && threadIdx.x == 0) { *fu.nbt_h += 1; *fu.nbt_g += 1; }
                }
            }
        }
        __syncthreads();
        mh = st4[0]; rh = st4[1]; mg = st4[2]; rg = st4[3];
    } else {
        unit_stats(stats, norm, n, c, C2, &mh, &rh);
        unit_stats(stats, norm, n, C + c, C2, &mg, &rg);
    }
    const float b_h = bias ? bias[c] : 0.f, b_g = bias ? bias[C + c] : 0.f;
    const float g_h = gh ? gh[c] : 1.f, be_h = bh ? bh[c] : 0.f, g_g = gg ? gg[c] : 1.f, be_g = bg ? bg[c] : 0.f;
    const float* ph = y + ((size_t)n * C2 + c) * HW;
    const float* pg = y + ((size_t)n * C2 + C + c) * HW;
    float* po = out + (size_t)plane * HW;
    float am = 0.f;
    // 16-byte accesses where the planes allow them (every layer of the sylvester stacks): the scalar loop moved 2.5 TB/s
    if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const f32x4* ph4 = reinterpret_cast<const f32x4*>(ph);
        const f32x4* pg4 = reinterpret_cast<const f32x4*>(pg);
        f32x4* po4 = reinterpret_cast<f32x4*>(po);
        for (int i = threadIdx.x + blockIdx.y * blockDim.x; i < (HW >> 2); i += blockDim.x * gridDim.y) {
            const f32x4 vh = ph4[i], vg = pg4[i];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ah = ((vh[e] + b_h) - mh) * rh * g_h + be_h;
                const float ag = ((vg[e] + b_g) - mg) * rg * g_g + be_g;
                o[e] = ah * (1.f / (1.f + expf(-ag)));
                am = fmaxf(am, fabsf(o[e]));
            }
            po4[i] = o;
        }
    } else
    for (int i = threadIdx.x + blockIdx.y * blockDim.x; i < HW; i += blockDim.x * gridDim.y) {
        const float ah = ((ph[i] + b_h) - mh) * rh * g_h + be_h;
        const float ag = ((pg[i] + b_g) - mg) * rg * g_g + be_g;
        const float o = ah * (1.f / (1.f + expf(-ag)));
        po[i] = o;
        am = fmaxf(am, fabsf(o));
    }
    if (amax_parts) gated_block_amax_out(am, amax_parts);      // (uniform)
}

// Backward pass 1: per unit sums  S1 = sum dA, S2 = sum dA * xhat  (dA = gradient w.r.t. the affine-norm output).
// sums[u] = {S1, S2}.  dA_h = dout * sig;  dA_g = dout * A_h * sig * (1 - sig).
__global__ void __launch_bounds__(256)
gated_bwd_sums_kernel(const float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ stats,
                      const float* __restrict__ gh, const float* __restrict__ bh, const float* __restrict__ gg,
                      const float* __restrict__ bg, const float* __restrict__ dout, int N, int C, int HW, int norm,
                      int nchunk, double* __restrict__ part) {
    // one workgroup per (h, g) channel PAIR: both units' sums need the same three planes (h, g, dout) and the same
    // sigmoid, so the pair reads them once (the per-unit form read every plane twice: 1.4 TB/s on the 64 x 64 layers)
    __shared__ double red[16 * 4 + 4];
    const int C2 = 2 * C;
    const int nrm = norm == NORM_NONE ? NORM_BN : norm;   // no norm: the "unit" is a channel over all images (S1 = bias gradient)
    const int c = nrm == NORM_BN ? blockIdx.x : blockIdx.x % C;
    const int n0 = nrm == NORM_BN ? 0 : blockIdx.x / C, nc = nrm == NORM_BN ? N : 1;
    const int per = (nc + nchunk - 1) / nchunk;
    const int na = n0 + blockIdx.y * per;
    int nb = na + per;
    if (nb > n0 + nc) nb = n0 + nc;
    const float b_h = bias ? bias[c] : 0.f, b_g = bias ? bias[C + c] : 0.f;
    const float g_h = gh ? gh[c] : 1.f, be_h = bh ? bh[c] : 0.f, g_g = gg ? gg[c] : 1.f, be_g = bg ? bg[c] : 0.f;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};             // h: S1, S2; g: S1, S2
    for (int n = na; n < nb; ++n) {
        float mh, rh, mg, rg;
        unit_stats(stats, norm, n, c, C2, &mh, &rh);
        unit_stats(stats, norm, n, C + c, C2, &mg, &rg);
        const float* ph = y + ((size_t)n * C2 + c) * HW;
        const float* pg = y + ((size_t)n * C2 + C + c) * HW;
        const float* pd = dout + ((size_t)n * C + c) * HW;
        auto term = [&](float vh, float vg, float vd) {
            const float xh = ((vh + b_h) - mh) * rh, xg = ((vg + b_g) - mg) * rg;
            const float ah = xh * g_h + be_h, ag = xg * g_g + be_g;
            const float sg = 1.f / (1.f + expf(-ag));
            const float dAh = vd * sg, dAg = vd * ah * sg * (1.f - sg);
            acc[0] += (double)dAh; acc[1] += (double)dAh * xh;
            acc[2] += (double)dAg; acc[3] += (double)dAg * xg;
        };
        if ((HW & 3) == 0) {
            const f32x4* ph4 = reinterpret_cast<const f32x4*>(ph);
            const f32x4* pg4 = reinterpret_cast<const f32x4*>(pg);
            const f32x4* pd4 = reinterpret_cast<const f32x4*>(pd);
            for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
                const f32x4 vh = ph4[i], vg = pg4[i], vd = pd4[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) term(vh[e], vg[e], vd[e]);
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += blockDim.x) term(ph[i], pg[i], pd[i]);
        }
    }
    block_sum_multi<4>(acc, red);
    if (threadIdx.x == 0) {
        const size_t uh = nrm == NORM_BN ? (size_t)c : (size_t)n0 * C2 + c, ug = uh + C;
        part[2 * (uh * nchunk + blockIdx.y)] = acc[0];
        part[2 * (uh * nchunk + blockIdx.y) + 1] = acc[1];
        part[2 * (ug * nchunk + blockIdx.y)] = acc[2];
        part[2 * (ug * nchunk + blockIdx.y) + 1] = acc[3];
    }
}

__global__ void gated_sums_finalize_kernel(const double* __restrict__ part, int units, int nchunk,
                                           float* __restrict__ sums) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    double s1 = 0.0, s2 = 0.0;
    for (int z = 0; z < nchunk; ++z) { s1 += part[2 * ((size_t)u * nchunk + z)]; s2 += part[2 * ((size_t)u * nchunk + z) + 1]; }
    sums[2 * u] = (float)s1; sums[2 * u + 1] = (float)s2;
}

// the fused backward (round 6, BatchNorm): the apply kernel's workgroups fold the sums pass's chunk partials themselves and the first
// workgroup of a channel stores the affine gradients (gated_sums_finalize_kernel + gated_param_kernel: two launches less per unit)
struct GatedBwdFuse {
    const double* part;       // NULL: `sums` holds the folded {S1, S2}
    int nchunk;
    float* dgh; float* dbh; float* dgg; float* dbg; float* dbias;
};

// Backward pass 2: dy[n][ch][p] = rstd * gamma * (dA - S1/m - xhat * S2/m)   (norm none: dy = dA)
__global__ void __launch_bounds__(256)
gated_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ stats,
                       const float* __restrict__ gh, const float* __restrict__ bh, const float* __restrict__ gg,
                       const float* __restrict__ bg, const float* __restrict__ dout, const float* __restrict__ sums,
                       int N, int C, int HW, int norm, float m_global, float* __restrict__ dy, float* __restrict__ amax_parts,
                       const GatedBwdFuse fu) {
    const int plane = blockIdx.x;           // n * C + c
    const int n = plane / C, c = plane % C;
    const int C2 = 2 * C;
    float mh, rh, mg, rg;
    unit_stats(stats, norm, n, c, C2, &mh, &rh);
    unit_stats(stats, norm, n, C + c, C2, &mg, &rg);
    const float b_h = bias ? bias[c] : 0.f, b_g = bias ? bias[C + c] : 0.f;
    const float g_h = gh ? gh[c] : 1.f, be_h = bh ? bh[c] : 0.f, g_g = gg ? gg[c] : 1.f, be_g = bg ? bg[c] : 0.f;
    float k1h = 0.f, k2h = 0.f, k1g = 0.f, k2g = 0.f;
    if (fu.part) {                          // (uniform; BatchNorm) the sums pass's chunk partials, folded here
        __shared__ float sm4[4];
        if (threadIdx.x < 2) {
            const size_t u = threadIdx.x ? (size_t)C + c : (size_t)c;
            double s1 = 0.0, s2 = 0.0;
            for (int z = 0; z < fu.nchunk; ++z) { s1 += fu.part[2 * (u * fu.nchunk + z)]; s2 += fu.part[2 * (u * fu.nchunk + z) + 1]; }
            const float f1 = (float)s1, f2 = (float)s2;      // (gated_sums_finalize_kernel's values)
            sm4[2 * threadIdx.x] = f1; sm4[2 * threadIdx.x + 1] = f2;
            if (n == 0 && blockIdx.y == 0) {                 // (gated_param_kernel's stores)
                if (threadIdx.x) { if (fu.dgg) fu.dgg[c] = f2; if (fu.dbg) fu.dbg[c] = f1; }
                else { if (fu.dgh) fu.dgh[c] = f2; if (fu.dbh) fu.dbh[c] = f1; }
                if (fu.dbias) fu.dbias[u] = 0.f;
            }
        }
        __syncthreads();
        const float m = m_global > 0.f ? m_global : (float)N * HW;
        k1h = sm4[0] / m; k2h = sm4[1] / m; k1g = sm4[2] / m; k2g = sm4[3] / m;
    } else
    if (norm != NORM_NONE) {
        const float m = (norm == NORM_BN) ? (m_global > 0.f ? m_global : (float)N * HW) : (float)HW;
        const int uh = (norm == NORM_BN) ? c : n * C2 + c, ug = (norm == NORM_BN) ? C + c : n * C2 + C + c;
        k1h = sums[2 * uh] / m; k2h = sums[2 * uh + 1] / m;
        k1g = sums[2 * ug] / m; k2g = sums[2 * ug + 1] / m;
    }
    const float* ph = y + ((size_t)n * C2 + c) * HW;
    const float* pg = y + ((size_t)n * C2 + C + c) * HW;
    const float* pd = dout + (size_t)plane * HW;
    float* dh = dy + ((size_t)n * C2 + c) * HW;
    float* dg = dy + ((size_t)n * C2 + C + c) * HW;
    float am = 0.f;
    if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
        // (16-byte accesses, as in gated_apply_kernel)
        const f32x4* ph4 = reinterpret_cast<const f32x4*>(ph);
        const f32x4* pg4 = reinterpret_cast<const f32x4*>(pg);
        const f32x4* pd4 = reinterpret_cast<const f32x4*>(pd);
        f32x4* dh4 = reinterpret_cast<f32x4*>(dh);
        f32x4* dg4 = reinterpret_cast<f32x4*>(dg);
        for (int i = threadIdx.x + blockIdx.y * blockDim.x; i < (HW >> 2); i += blockDim.x * gridDim.y) {
            const f32x4 vh = ph4[i], vg = pg4[i], vd = pd4[i];
            f32x4 o_h, o_g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = ((vh[e] + b_h) - mh) * rh, xg = ((vg[e] + b_g) - mg) * rg;
                const float ah = xh * g_h + be_h, ag = xg * g_g + be_g;
                const float sg = 1.f / (1.f + expf(-ag));
                const float dAh = vd[e] * sg, dAg = vd[e] * ah * sg * (1.f - sg);
                float oh = dAh, og = dAg;
                if (norm != NORM_NONE) {
                    oh = rh * g_h * (dAh - k1h - xh * k2h);
                    og = rg * g_g * (dAg - k1g - xg * k2g);
                }
                o_h[e] = oh; o_g[e] = og;
                am = fmaxf(am, fmaxf(fabsf(oh), fabsf(og)));
            }
            dh4[i] = o_h; dg4[i] = o_g;
        }
    } else
    for (int i = threadIdx.x + blockIdx.y * blockDim.x; i < HW; i += blockDim.x * gridDim.y) {
        const float xh = ((ph[i] + b_h) - mh) * rh, xg = ((pg[i] + b_g) - mg) * rg;
        const float ah = xh * g_h + be_h, ag = xg * g_g + be_g;
        const float sg = 1.f / (1.f + expf(-ag));
        const float dAh = pd[i] * sg, dAg = pd[i] * ah * sg * (1.f - sg);
        float oh = dAh, og = dAg;
        if (norm != NORM_NONE) {
            oh = rh * g_h * (dAh - k1h - xh * k2h);
            og = rg * g_g * (dAg - k1g - xg * k2g);
        }
        dh[i] = oh; dg[i] = og;
        am = fmaxf(am, fmaxf(fabsf(oh), fabsf(og)));
    }
    if (amax_parts) gated_block_amax_out(am, amax_parts);      // (uniform)
}

// Parameter gradients from the per-unit sums: dgamma = sum S2, dbeta = sum S1 (over images for IN); the conv
// bias gradient is sum dy = 0 under a norm (it cancels in the normalisation) and S1 with no norm.
__global__ void gated_param_kernel(const float* __restrict__ sums, int N, int C, int norm, float* __restrict__ dgh,
                                   float* __restrict__ dbh, float* __restrict__ dgg, float* __restrict__ dbg,
                                   float* __restrict__ dbias) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;   // [0, 2C)
    const int C2 = 2 * C;
    if (ch >= C2) return;
    double s1 = 0.0, s2 = 0.0;
    if (norm == NORM_IN) {
        for (int n = 0; n < N; ++n) { s1 += sums[2 * (n * C2 + ch)]; s2 += sums[2 * (n * C2 + ch) + 1]; }
    } else {
        s1 = sums[2 * ch]; s2 = sums[2 * ch + 1];
    }
    const bool is_g = ch >= C;
    const int c = is_g ? ch - C : ch;
    if (norm != NORM_NONE) {
        if (is_g) { if (dgg) dgg[c] = (float)s2; if (dbg) dbg[c] = (float)s1; }
        else { if (dgh) dgh[c] = (float)s2; if (dbh) dbh[c] = (float)s1; }
    }
    if (dbias) dbias[ch] = (norm == NORM_NONE) ? (float)s1 : 0.f;
}

struct GatedRun { float* rm_h; float* rv_h; float* rm_g; float* rv_g; long long* nbt_h; long long* nbt_g; float mom; };
thread_local GatedRun t_gated_run = {};
bool gated_fuse_on() {      // (read per call: the tests switch it inside one process)
    const char* e = getenv("GENESIS_GATED_FUSE");
    return !(e && e[0] == '0');
}

int nunits(int norm, int N, int C) { return norm == NORM_IN ? N * 2 * C : 2 * C; }
// image chunks per unit: a BatchNorm channel spans all N images -> spread it over ~2048 (statistics) / ~1024 (backward sums, one per channel pair) workgroups
int nchunks(int norm, int N, int C) {
    if (norm == NORM_IN) return 1;
    int z = 2048 / (2 * C);
    if (z < 1) z = 1;
    return z > N ? N : z;
}

}  // namespace

extern "C" {

// {mean, rstd} per unit, followed by the fp64 chunk partials of the statistics pass (scratch)
size_t gx_gated_stats_floats(int norm, int N, int C) {
    return (size_t)2 * nunits(norm, N, C) + (size_t)4 * nunits(norm, N, C) * nchunks(norm, N, C);
}

int gx_gated_norm_fwd(const float* y, const float* bias, int norm, const float* gamma_h, const float* beta_h,
                      const float* gamma_g, const float* beta_g, int N, int C, int H, int W, float eps, float* out,
                      float* stats, gx_stream_t stream) {
    GX_CHECK_ARG(y && out && stats, "gx_gated_norm_fwd: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && norm >= 0 && norm <= 2, "gx_gated_norm_fwd: bad dims / norm");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    GatedFuse fu = {};
    const GatedRun run = t_gated_run;      // (one-shot: gx_gated_bn_running)
    t_gated_run = GatedRun{};
    if (norm != NORM_NONE) {
        GxProf pf(KID_GATED, s, 0.0, 4.0 * N * 2.0 * C * HW);
        const int units = nunits(norm, N, C), nz = nchunks(norm, N, C);
        double* part = reinterpret_cast<double*>(stats + 2 * (size_t)units);
        hipLaunchKernelGGL(gated_stats_partial_kernel, dim3(units, nz), dim3(256), 0, s, y, bias, N, 2 * C, HW, norm,
                           nz, part);
        GX_CHECK_LAUNCH("gx_gated_norm_fwd(stats)");
        const double m = (norm == NORM_BN) ? (double)N * HW : (double)HW;
        if (gated_fuse_on()) {
            fu.part = part; fu.nchunk = nz; fu.m = m; fu.eps = eps; fu.stats_out = stats;
            if (norm == NORM_BN && run.rm_h) {
                fu.rm_h = run.rm_h; fu.rv_h = run.rv_h; fu.rm_g = run.rm_g; fu.rv_g = run.rv_g; fu.nbt_h = run.nbt_h; fu.nbt_g = run.nbt_g;
                fu.mom = run.mom;
            }
        } else {
            hipLaunchKernelGGL(gated_stats_finalize_kernel, dim3(gx_ceil_div(units, 256)), dim3(256), 0, s,
                               (const double*)part, units, nz, m, eps, stats);
            GX_CHECK_LAUNCH("gx_gated_norm_fwd(stats finalize)");
        }
    }
    {
        GxProf pf(KID_GATED, s, 0.0, 4.0 * N * 3.0 * C * HW);
        const dim3 grid(N * C, N * C >= 2048 ? 1 : gx_ceil_div(HW, 1024));      // (many planes: one workgroup each -- and one partial maximum each)
        float* ap = gx_amax_producer_out(out, true, grid.x * grid.y, (size_t)N * C * HW);      // (out: a whole plain tensor)
        hipLaunchKernelGGL(gated_apply_kernel, grid, dim3(256), 0, s, y, bias,
                           (const float*)stats, gamma_h, beta_h, gamma_g, beta_g, N, C, HW, norm, out, ap, fu);
    }
    GX_CHECK_LAUNCH("gx_gated_norm_fwd");
    if (run.rm_h && !(norm == NORM_BN && fu.part)) {
        // an armed running-statistics update the apply kernel did not take (GENESIS_GATED_FUSE=0): the stand-alone launch
        GX_CHECK_ARG(norm == NORM_BN, "gx_gated_bn_running is for BatchNorm units");
        return gx_bn_running_update(stats, C, (double)N * HW, eps, run.mom, run.rm_h, run.rv_h, run.rm_g, run.rv_g, run.nbt_h,
                                    run.nbt_g, stream);
    }
    return GX_OK;
}

/* one-shot, per thread: the NEXT gx_gated_norm_fwd of this thread (norm 1) also applies gx_bn_running_update to these buffers --
 * inside its apply kernel, no launch of its own.  NULL rm_h disarms. */
int gx_gated_bn_running(float* rm_h, float* rv_h, float* rm_g, float* rv_g, long long* nbt_h, long long* nbt_g, float momentum) {
    if (!rm_h) { t_gated_run = GatedRun{}; return GX_OK; }
    GX_CHECK_ARG(rv_h && rm_g && rv_g && nbt_h && nbt_g, "gx_gated_bn_running: null pointer");
    t_gated_run = GatedRun{rm_h, rv_h, rm_g, rv_g, nbt_h, nbt_g, momentum};
    return GX_OK;
}

/* BatchNorm running statistics of a gated unit (layers.py:40-101 with nn.BatchNorm2d norms, momentum 0.1): stats = the
 * {mean, rstd} pairs gx_gated_norm_fwd left for its 2C units, m = N H W; running_mean / running_var [C] of the h and g
 * norms and their num_batches_tracked (int64 scalars) are updated in place. */
int gx_bn_running_update(const float* stats, int C, double m, float eps, float momentum, float* rm_h, float* rv_h, float* rm_g,
                         float* rv_g, long long* nbt_h, long long* nbt_g, gx_stream_t stream) {
    GX_CHECK_ARG(stats && rm_h && rv_h && rm_g && rv_g && nbt_h && nbt_g, "gx_bn_running_update: null pointer");
    GX_CHECK_ARG(C > 0 && m >= 1.0, "gx_bn_running_update: bad dims");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(gx_ceil_div(2 * C, 256)), dim3(256), 0, s, stats, C, m, (double)eps,
                       momentum, rm_h, rv_h, rm_g, rv_g, nbt_h, nbt_g);
    GX_CHECK_LAUNCH("gx_bn_running_update");
    return GX_OK;
}

size_t gx_gated_norm_bwd_ws_bytes(int norm, int N, int C) {
    const int nb = norm == NORM_NONE ? NORM_BN : norm;
    return ((size_t)2 * nunits(norm, N, C) + (size_t)4 * nunits(norm, N, C) * nchunks(nb, N, C)) * sizeof(float);
}

int gx_gated_norm_bwd(const float* y, const float* bias, int norm, const float* gamma_h, const float* beta_h,
                      const float* gamma_g, const float* beta_g, const float* stats, const float* dout, int N, int C,
                      int H, int W, float* dy, float* dgamma_h, float* dbeta_h, float* dgamma_g, float* dbeta_g,
                      float* dbias, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(y && stats && dout && dy && ws, "gx_gated_norm_bwd: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && norm >= 0 && norm <= 2, "gx_gated_norm_bwd: bad dims / norm");
    GX_CHECK_ARG(ws_bytes >= gx_gated_norm_bwd_ws_bytes(norm, N, C), "gx_gated_norm_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    float* sums = (float*)ws;
    GatedBwdFuse bf = {};
    {
        GxProf pf(KID_GATED, s, 0.0, 4.0 * N * 3.0 * C * HW);
        const int units = nunits(norm, N, C), nz = nchunks(norm == NORM_NONE ? NORM_BN : norm, N, C);
        double* part = reinterpret_cast<double*>(sums + 2 * (size_t)units);
        hipLaunchKernelGGL(gated_bwd_sums_kernel, dim3(units / 2, nz), dim3(256), 0, s, y, bias, stats, gamma_h, beta_h,
                           gamma_g, beta_g, dout, N, C, HW, norm, nz, part);
        if (norm == NORM_BN && gated_fuse_on()) {
            bf.part = part; bf.nchunk = nz;
            bf.dgh = dgamma_h; bf.dbh = dbeta_h; bf.dgg = dgamma_g; bf.dbg = dbeta_g; bf.dbias = dbias;
        } else {
            hipLaunchKernelGGL(gated_sums_finalize_kernel, dim3(gx_ceil_div(units, 256)), dim3(256), 0, s,
                               (const double*)part, units, nz, sums);
        }
    }
    GX_CHECK_LAUNCH("gx_gated_norm_bwd(sums)");
    {
        GxProf pf(KID_GATED, s, 0.0, 4.0 * N * 5.0 * C * HW);
        const dim3 grid(N * C, N * C >= 2048 ? 1 : gx_ceil_div(HW, 1024));
        float* ap = gx_amax_producer_out(dy, true, grid.x * grid.y, (size_t)N * 2 * C * HW);
        hipLaunchKernelGGL(gated_bwd_apply_kernel, grid, dim3(256), 0, s, y, bias, stats,
                           gamma_h, beta_h, gamma_g, beta_g, dout, (const float*)sums, N, C, HW, norm, 0.f, dy, ap, bf);
    }
    GX_CHECK_LAUNCH("gx_gated_norm_bwd(apply)");
    if (!bf.part) {
        hipLaunchKernelGGL(gated_param_kernel, dim3(gx_ceil_div(2 * C, 64)), dim3(64), 0, s, (const float*)sums, N, C, norm,
                           dgamma_h, dbeta_h, dgamma_g, dbeta_g, dbias);
        GX_CHECK_LAUNCH("gx_gated_norm_bwd(params)");
    }
    return GX_OK;
}

/* ---- the same unit with the BatchNorm statistics taken over SEVERAL ranks' batches (cross-replica BatchNorm; the reference's
 *      single-device batch statistics at the GLOBAL batch, genesis_config.py:39-40 / layers.py:26-27, when the batch is sharded):
 *      the forward and the backward are cut where the per-channel sums exist, the caller adds them over the ranks in between. */
size_t gx_gated_bn_sums_ws_bytes(int N, int C) { return (size_t)4 * nunits(NORM_BN, N, C) * nchunks(NORM_BN, N, C) * sizeof(float); }

int gx_gated_bn_local_sums(const float* y, const float* bias, int N, int C, int H, int W, double* sums, void* ws,
                           size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(y && sums && ws, "gx_gated_bn_local_sums: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0, "gx_gated_bn_local_sums: bad dims");
    GX_CHECK_ARG(ws_bytes >= gx_gated_bn_sums_ws_bytes(N, C), "gx_gated_bn_local_sums: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int units = 2 * C, nz = nchunks(NORM_BN, N, C), HW = H * W;
    GxProf pf(KID_GATED, s, 0.0, 4.0 * N * 2.0 * C * HW);
    hipLaunchKernelGGL(gated_stats_partial_kernel, dim3(units, nz), dim3(256), 0, s, y, bias, N, 2 * C, HW, (int)NORM_BN, nz,
                       (double*)ws);
    hipLaunchKernelGGL(gated_raw_sums_kernel, dim3(gx_ceil_div(units, 256)), dim3(256), 0, s, (const double*)ws, units, nz, sums);
    GX_CHECK_LAUNCH("gx_gated_bn_local_sums");
    return GX_OK;
}

int gx_gated_bn_apply(const float* y, const float* bias, const double* sums, double m, const float* gamma_h,
                      const float* beta_h, const float* gamma_g, const float* beta_g, int N, int C, int H, int W, float eps,
                      float* out, float* stats, gx_stream_t stream) {
    GX_CHECK_ARG(y && sums && out && stats, "gx_gated_bn_apply: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && m >= 1.0, "gx_gated_bn_apply: bad dims");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    hipLaunchKernelGGL(gated_stats_from_sums_kernel, dim3(gx_ceil_div(2 * C, 256)), dim3(256), 0, s, sums, 2 * C, m, eps, stats);
    GxProf pf(KID_GATED, s, 0.0, 4.0 * N * 3.0 * C * HW);
    hipLaunchKernelGGL(gated_apply_kernel, dim3(N * C, gx_ceil_div(HW, 1024)), dim3(256), 0, s, y, bias, (const float*)stats,
                       gamma_h, beta_h, gamma_g, beta_g, N, C, HW, (int)NORM_BN, out, (float*)nullptr, GatedFuse{});
    GX_CHECK_LAUNCH("gx_gated_bn_apply");
    return GX_OK;
}

/*      backward, first half: the rank's own {S1, S2} per channel (float [2][2C] pairs) and from them the affine / bias
 *      gradients (LOCAL sums: the gradient all-reduce of the step adds the ranks' contributions). */
int gx_gated_bn_bwd_local_sums(const float* y, const float* bias, const float* gamma_h, const float* beta_h,
                               const float* gamma_g, const float* beta_g, const float* stats, const float* dout, int N, int C,
                               int H, int W, float* sums, float* dgamma_h, float* dbeta_h, float* dgamma_g, float* dbeta_g,
                               float* dbias, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(y && stats && dout && sums && ws, "gx_gated_bn_bwd_local_sums: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0, "gx_gated_bn_bwd_local_sums: bad dims");
    GX_CHECK_ARG(ws_bytes >= gx_gated_bn_sums_ws_bytes(N, C), "gx_gated_bn_bwd_local_sums: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W, units = 2 * C, nz = nchunks(NORM_BN, N, C);
    {
        GxProf pf(KID_GATED, s, 0.0, 4.0 * N * 3.0 * C * HW);
        hipLaunchKernelGGL(gated_bwd_sums_kernel, dim3(units / 2, nz), dim3(256), 0, s, y, bias, stats, gamma_h, beta_h,
                           gamma_g, beta_g, dout, N, C, HW, (int)NORM_BN, nz, (double*)ws);
        hipLaunchKernelGGL(gated_sums_finalize_kernel, dim3(gx_ceil_div(units, 256)), dim3(256), 0, s, (const double*)ws, units,
                           nz, sums);
    }
    hipLaunchKernelGGL(gated_param_kernel, dim3(gx_ceil_div(2 * C, 64)), dim3(64), 0, s, (const float*)sums, N, C, (int)NORM_BN,
                       dgamma_h, dbeta_h, dgamma_g, dbeta_g, dbias);
    GX_CHECK_LAUNCH("gx_gated_bn_bwd_local_sums");
    return GX_OK;
}

/*      backward, second half: dy from the sums added over the ranks and the global count m. */
int gx_gated_bn_bwd_apply(const float* y, const float* bias, const float* gamma_h, const float* beta_h, const float* gamma_g,
                          const float* beta_g, const float* stats, const float* dout, const float* sums, double m, int N, int C,
                          int H, int W, float* dy, gx_stream_t stream) {
    GX_CHECK_ARG(y && stats && dout && sums && dy, "gx_gated_bn_bwd_apply: null pointer");
    GX_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && m >= 1.0, "gx_gated_bn_bwd_apply: bad dims");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    GxProf pf(KID_GATED, s, 0.0, 4.0 * N * 5.0 * C * HW);
    hipLaunchKernelGGL(gated_bwd_apply_kernel, dim3(N * C, gx_ceil_div(HW, 1024)), dim3(256), 0, s, y, bias, stats, gamma_h,
                       beta_h, gamma_g, beta_g, dout, sums, N, C, HW, (int)NORM_BN, (float)m, dy, (float*)nullptr, GatedBwdFuse{});
    GX_CHECK_LAUNCH("gx_gated_bn_bwd_apply");
    return GX_OK;
}

}  // extern "C"
