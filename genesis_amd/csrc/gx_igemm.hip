// Generic strided convolution (any k <= 5, stride, pad) as an implicit GEMM on the fp32 matrix cores: the sylvester
// gated 5x5 (de)convolutions of GENESIS / BaselineVAE (third_party/sylvester/layers.py:11-101, VAE.py:85-168) and
// the stride-2 3x3 encoder convs of the ComponentVAE (modules/encoders.py:31-34).  Replaces the first-round direct
// kernels (one pixel x 8 channels per thread, weights re-read from global: ~5 TF/s) behind the same entry points.
//
//   MODE 0  forward   C[co][(img,oh,ow)] = sum_{ci,kh,kw} w[co][ci][kh][kw] * x[img][ci][oh s - p + kh][ow s - p + kw]
//   MODE 1  dgrad     C[ci][(img,ih,iw)] = sum_{co,kh,kw} w[co][ci][kh][kw] * dy[img][co][(ih + p - kh)/s][(iw + p - kw)/s]
//   MODE 2  wgrad     C[co][(ci,kh,kw)]  = sum_{img,oh,ow} dy[img][co][oh][ow] * x[img][ci][oh s - p + kh][ow s - p + kw]
//
// Workgroup tile 64 (M) x 128 (N), K chunks of 16; 4 waves of 32 x 64 (two mfma_f32_32x32x2f32 accumulators); both
// operands are gathered global -> registers one chunk ahead and staged in double-buffered LDS as [k][m] / [k][n] (the
// MFMA fragment reads are unit-stride).  The contraction index of a wave's loads is wave-uniform, so its (ci,kh,kw)
// / (img,oh,ow) decomposition runs on the scalar unit with host-precomputed magic divisors; the tile-column
// decomposition of a thread is done once.  wgrad splits K over gridDim.z into partial slabs + a fixed-order reduce.
#include "gx_common.h"

#include <cstdlib>

namespace {

struct FastDiv {   // n / d for 0 <= n < 2^31 via one mul_hi (Granlund-Montgomery round-up)
    unsigned mul, shift, d;
};
FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    if (d == 1) { f.mul = 0; f.shift = 0; return f; }
    unsigned s = 0;
    while ((1u << s) < d) ++s;                 // s = ceil(log2 d)
    const unsigned long long m = ((1ull << (32 + s)) + d - 1) / d - (1ull << 32);   // 33-bit magic minus 2^32
    f.mul = (unsigned)m;
    f.shift = s;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    if (f.d == 1) return n;
    const unsigned t = __umulhi(n, f.mul);
    return (t + ((n - t) >> 1)) >> (f.shift - 1);
}

struct IG {
    int N, Cin, Cout, H, W, Ho, Wo, k, stride, pad;
    int M, Ncols, K;            // GEMM dims of this mode
    int chunks_per_split;       // K chunks (of 16) per gridDim.z slice
    FastDiv d_kk, d_k;          // / (k*k), / k
    FastDiv d_how, d_wo;        // / (Ho*Wo), / Wo     (mode 2 contraction index)
};

constexpr int BM = 64, BN = 128, BK = 16;

template <int MODE>
__global__ void __launch_bounds__(256)
igemm_kernel(const float* __restrict__ a_src, const float* __restrict__ b_src, const float* __restrict__ bias,
             int act, float* __restrict__ out, IG g) {
    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = gx_xcd_tile(blockIdx.x, gridDim.x) * BN;     // (XCD-aware: neighbouring pixel tiles share halos)
    const int kk2 = g.k * g.k;
    const int HW = g.H * g.W, HoWo = g.Ho * g.Wo;

    // ---- per-thread tile columns (fixed for the whole kernel) ----
    // A: row m = m0 + (tid & 63), k slots kg*4 .. kg*4+3 with kg = wave          (64 x 16 tile, 4 per thread)
    // B: cols n = n0 + (tid & 63) and n + 64, same k slots                        (128 x 16 tile, 8 per thread)
    const int am = m0 + (tid & 63);
    const bool am_ok = am < g.M;
    int bcol_base[2], bc_a[2], bc_b[2];     // per-mode meaning, see below
    bool bn_ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + (tid & 63) + 64 * h;
        bn_ok[h] = n < g.Ncols;
        const int nn = bn_ok[h] ? n : 0;
        if (MODE == 0) {            // n = (img, oh, ow): base = img*Cin*HW, a = oh*s - p, b = ow*s - p
            const int img = nn / HoWo, p = nn - img * HoWo;
            const int oh = p / g.Wo, ow = p - oh * g.Wo;
            bcol_base[h] = img * g.Cin * HW; bc_a[h] = oh * g.stride - g.pad; bc_b[h] = ow * g.stride - g.pad;
        } else if (MODE == 1) {     // n = (img, ih, iw): base = img*Cout*HoWo, a = ih + p, b = iw + p
            const int img = nn / HW, p = nn - img * HW;
            const int ih = p / g.W, iw = p - ih * g.W;
            bcol_base[h] = img * g.Cout * HoWo; bc_a[h] = ih + g.pad; bc_b[h] = iw + g.pad;
        } else {                    // n = (ci, kh, kw): base = ci*HW, a = kh - p, b = kw - p
            const int ci = nn / kk2, r = nn - ci * kk2;
            const int kh = r / g.k, kw = r - kh * g.k;
            bcol_base[h] = ci * HW; bc_a[h] = kh - g.pad; bc_b[h] = kw - g.pad;
        }
    }

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    const int nchunks = (g.K + BK - 1) / BK;
    const int c_begin = blockIdx.z * g.chunks_per_split;
    int c_end = c_begin + g.chunks_per_split;
    if (c_end > nchunks) c_end = nchunks;

    float ra[4], rb[2][4];
// one chunk of both operands into registers; the 4 contraction indices of this wave are wave-uniform
#define IG_LOAD(chunk)                                                                                          \
    {                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const int kk = (chunk) * BK + wave * 4 + j;                                                         \
            const bool kv = kk < g.K;                                                                           \
            const unsigned kq = kv ? (unsigned)kk : 0u;                                                         \
            float av = 0.f, bv0 = 0.f, bv1 = 0.f;                                                               \
            if (MODE == 0) {                                                                                    \
                const int ci = (int)fdiv(kq, g.d_kk), r = (int)kq - ci * kk2;                                   \
                const int kh = (int)fdiv((unsigned)r, g.d_k), kw = r - kh * g.k;                                \
                if (kv && am_ok) av = a_src[(size_t)am * g.K + kq];                                             \
                const int koff = ci * HW + kh * g.W + kw;                                                       \
                const int ih0 = bc_a[0] + kh, iw0 = bc_b[0] + kw, ih1 = bc_a[1] + kh, iw1 = bc_b[1] + kw;       \
                if (kv && bn_ok[0] && ih0 >= 0 && ih0 < g.H && iw0 >= 0 && iw0 < g.W)                           \
                    bv0 = b_src[(size_t)bcol_base[0] + koff + bc_a[0] * g.W + bc_b[0]];                         \
                if (kv && bn_ok[1] && ih1 >= 0 && ih1 < g.H && iw1 >= 0 && iw1 < g.W)                           \
                    bv1 = b_src[(size_t)bcol_base[1] + koff + bc_a[1] * g.W + bc_b[1]];                         \
            } else if (MODE == 1) {                                                                             \
                const int co = (int)fdiv(kq, g.d_kk), r = (int)kq - co * kk2;                                   \
                const int kh = (int)fdiv((unsigned)r, g.d_k), kw = r - kh * g.k;                                \
                if (kv && am_ok) av = a_src[((size_t)co * g.Cin + am) * kk2 + r];                               \
                _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                 \
                    const int th = bc_a[h] - kh, tw = bc_b[h] - kw;                                             \
                    float v = 0.f;                                                                              \
                    if (kv && bn_ok[h] && th >= 0 && tw >= 0) {                                                 \
                        int oh, ow;                                                                             \
                        if (g.stride == 1) { oh = th; ow = tw; }                                                \
                        else if (g.stride == 2) { oh = th >> 1; ow = tw >> 1; }                                 \
                        else { oh = th / g.stride; ow = tw / g.stride; }                                        \
                        if (oh * g.stride == th && ow * g.stride == tw && oh < g.Ho && ow < g.Wo)               \
                            v = b_src[(size_t)bcol_base[h] + (size_t)co * HoWo + oh * g.Wo + ow];               \
                    }                                                                                           \
                    if (h == 0) bv0 = v; else bv1 = v;                                                          \
                }                                                                                               \
            } else {                                                                                            \
                const int img = (int)fdiv(kq, g.d_how), p = (int)kq - img * HoWo;                               \
                const int oh = (int)fdiv((unsigned)p, g.d_wo), ow = p - oh * g.Wo;                              \
                if (kv && am_ok) av = a_src[((size_t)img * g.Cout + am) * HoWo + p];                            \
                const int ihb = oh * g.stride, iwb = ow * g.stride;                                             \
                _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                 \
                    const int ih = ihb + bc_a[h], iw = iwb + bc_b[h];                                           \
                    float v = 0.f;                                                                              \
                    if (kv && bn_ok[h] && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)                           \
                        v = b_src[(size_t)img * g.Cin * HW + bcol_base[h] + ih * g.W + iw];                     \
                    if (h == 0) bv0 = v; else bv1 = v;                                                          \
                }                                                                                               \
            }                                                                                                   \
            ra[j] = av; rb[0][j] = bv0; rb[1][j] = bv1;                                                         \
        }                                                                                                       \
    }
#define IG_COMMIT(buf)                                                                                          \
    {                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            As[buf][wave * 4 + j][tid & 63] = ra[j];                                                            \
            Bs[buf][wave * 4 + j][tid & 63] = rb[0][j];                                                         \
            Bs[buf][wave * 4 + j][(tid & 63) + 64] = rb[1][j];                                                  \
        }                                                                                                       \
    }

    if (c_begin < c_end) IG_LOAD(c_begin)
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        IG_COMMIT(buf)
        __syncthreads();
        if (c + 1 < c_end) IG_LOAD(c + 1)
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const float a = As[buf][2 * ks + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b0 = Bs[buf][2 * ks + (lane >> 5)][wn * 64 + (lane & 31)];
            const float b1 = Bs[buf][2 * ks + (lane >> 5)][wn * 64 + 32 + (lane & 31)];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        }
    }
#undef IG_LOAD
#undef IG_COMMIT

    // ---- epilogue: C/D layout col = lane & 31 (n), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (m) ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= g.Ncols) continue;
        size_t obase;       // address of (m = 0, n); consecutive m are `mstride` apart
        size_t mstride;
        if (MODE == 2) {
            obase = (size_t)blockIdx.z * g.M * g.Ncols + n; mstride = g.Ncols;
        } else {
            const int P = MODE == 0 ? HoWo : HW;
            const int img = n / P, p = n - img * P;
            obase = (size_t)img * g.M * P + p; mstride = P;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (m < g.M) {
                float v = acc[j][reg];
                if (MODE == 0 && gridDim.z > 1) {       // split contraction: raw slab z in the output's layout (igemm_reduce_act_kernel)
                    out[(size_t)blockIdx.z * g.M * g.Ncols + obase + (size_t)m * mstride] = v;
                    continue;
                }
                if (MODE == 0) {
                    if (bias) v += bias[m];
                    if (act == 1) v = v > 0.f ? v : 0.f;
                    else if (act == 2) v = v > 0.f ? v : expm1f(v);
                }
                out[obase + (size_t)m * mstride] = v;
            }
        }
    }
}

// dw[i] = sum_z partial[z][i], fixed order
__global__ void __launch_bounds__(256)
igemm_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int total, int nsplit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float s = partial[i];
    for (int z = 1; z < nsplit; ++z) s += partial[(size_t)z * total + i];
    out[i] = s;
}

// y[n][m][p] = act(sum_z partial[z][n][m][p] + bias[m]), fixed order (the forward conv with a split contraction)
__global__ void __launch_bounds__(256)
igemm_reduce_act_kernel(const float* __restrict__ partial, const float* __restrict__ bias, int act, float* __restrict__ out,
                        int total, int nsplit, int M, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float s = partial[i];
    for (int z = 1; z < nsplit; ++z) s += partial[(size_t)z * total + i];
    if (bias) s += bias[(i / P) % M];
    if (act == 1) s = s > 0.f ? s : 0.f;
    else if (act == 2) s = s > 0.f ? s : expm1f(s);
    out[i] = s;
}
// splits of the forward conv's contraction: only where the output tiles leave most of the chip idle (the ComponentVAE encoder's
// last stride-2 layers, modules/encoders.py:31-34, at K*B = 224 images: 64 -> 64 from 8 x 8 is 28 tiles of 36 K-chunks = 62 us)
int fwd_splits(const IG& g) {
    static const char* env = getenv("GENESIS_DCONV_FWD_SPLIT");
    if (env && env[0] == '0') return 1;
    const int tiles = gx_ceil_div(g.Ncols, BN) * gx_ceil_div(g.M, BM), nchunks = gx_ceil_div(g.K, BK);
    if (tiles >= 128 || nchunks < 8) return 1;
    int ns = gx_ceil_div(256, tiles);
    if (ns > nchunks / 4) ns = nchunks / 4;
    if (ns > 16) ns = 16;
    return ns < 1 ? 1 : ns;
}

int ig_geom(const char* name, IG* g, int mode, int N, int Cin, int Cout, int H, int W, int k, int stride, int pad) {
    GX_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad dims", name);
    GX_CHECK_ARG(k >= 1 && k <= 5 && stride >= 1 && pad >= 0, "%s: kernel <= 5x5, stride >= 1", name);
    g->N = N; g->Cin = Cin; g->Cout = Cout; g->H = H; g->W = W; g->k = k; g->stride = stride; g->pad = pad;
    g->Ho = (H + 2 * pad - k) / stride + 1;
    g->Wo = (W + 2 * pad - k) / stride + 1;
    GX_CHECK_ARG(g->Ho > 0 && g->Wo > 0, "%s: empty output", name);
    const double big = 2.0e9;
    GX_CHECK_ARG((double)N * Cin * H * W < big && (double)N * Cout * g->Ho * g->Wo < big, "%s: tensor too large", name);
    if (mode == 0) { g->M = Cout; g->Ncols = N * g->Ho * g->Wo; g->K = Cin * k * k; }
    else if (mode == 1) { g->M = Cin; g->Ncols = N * H * W; g->K = Cout * k * k; }
    else { g->M = Cout; g->Ncols = Cin * k * k; g->K = N * g->Ho * g->Wo; }
    g->d_kk = make_fastdiv(k * k); g->d_k = make_fastdiv(k);
    g->d_how = make_fastdiv(g->Ho * g->Wo); g->d_wo = make_fastdiv(g->Wo);
    g->chunks_per_split = gx_ceil_div(g->K, BK);
    return GX_OK;
}

int wgrad_splits(const IG& g) {
    const int tiles = gx_ceil_div(g.M, BM) * gx_ceil_div(g.Ncols, BN);
    const int nchunks = gx_ceil_div(g.K, BK);
    int nsplit = gx_ceil_div(768, tiles);
    if (nsplit > nchunks) nsplit = nchunks;
    if (nsplit > 256) nsplit = 256;
    return nsplit < 1 ? 1 : nsplit;
}

}  // namespace

// ---- conv3x3 stride 2 pad 1: data gradient without the stride's structural zeros ---------------------------------
// The ComponentVAE encoder's four stride-2 3x3 convs (modules/encoders.py:31-34; 4 -> 32 -> 32 -> 64 -> 64 channels on the
// K B slot images) are 0.5 % of MONet's / GENESIS' flops, but their data gradients on the generic implicit-GEMM kernel --
// a 64-wide output-channel tile for 4 .. 64 channels, three of four gathered taps structurally zero -- took 0.86 ms per
// step (495 us for the first layer alone, whose dx is needed for ONE channel: the mask; the image channels carry no
// gradient).  Here a thread owns a 2 x 2 block of dx pixels (all four stride parities: 1 + 2 + 2 + 4 = 9 taps, the same
// work for every thread) for CIB input channels and walks the output channels: 4 dy loads feed 9 CIB FMAs, weights
// broadcast from LDS.  Exactly the useful multiply-adds, on the vector ALUs.
template <int CIB, int UB = 1>      // UB: output channels whose dy values are loaded ahead of their use
__global__ void __launch_bounds__(256)
conv3x3s2_dgrad_small_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N, int Cin,
                             int Cout, int H, int W, int cin_n, int dxC) {
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [Cout][CIB][12]: 9 taps, padded to 3 float4
    const int Ho = H >> 1, Wo = W >> 1;
    const int ci0 = blockIdx.x * CIB;          // (x: the channel blocks that read the same dy values are dispatched together)
    for (int e = threadIdx.x; e < Cout * CIB * 9; e += 256) {
        const int t = e % 9, c = (e / 9) % CIB, co = e / (9 * CIB);
        wl[(co * CIB + c) * 12 + t] = ci0 + c < Cin ? w[((size_t)co * Cin + ci0 + c) * 9 + t] : 0.f;
    }
    __syncthreads();
    // threads over (image, 2 x 2 block) TOGETHER: the encoder's last layers have 16 and 64 blocks per image -- one workgroup per
    // (image, channel block) left 240 / 192 of its 256 threads idle (62 / 40 us for the 4 x 4 and 8 x 8 layers at 224 images)
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= N * Ho * Wo) return;
    const int n = q / (Ho * Wo), p = q - n * (Ho * Wo);
    const int i = p / Wo, j = p - i * Wo;
    const bool jr = j + 1 < Wo, ir = i + 1 < Ho;
    const float* d = dy + (size_t)n * Cout * Ho * Wo + (size_t)i * Wo + j;
    float a00[CIB], a01[CIB], a10[CIB], a11[CIB];
#pragma unroll
    for (int c = 0; c < CIB; ++c) { a00[c] = 0.f; a01[c] = 0.f; a10[c] = 0.f; a11[c] = 0.f; }
    // four output channels' dy values are loaded before the first of them is used: the loop is bound by load latency, not by its
    // 36 CIB multiply-adds per channel (the accumulation order per output value stays co = 0, 1, 2, ...)
    // (UB = 8 for the launches of a few hundred workgroups -- the encoder's last layers: nothing else hides the latency there;
    //  the chip-filling layers measured 10 - 15 % slower with it than with UB = 1)
    const size_t cs = (size_t)Ho * Wo;
    // (unconditional loads from clamped addresses, the border's zeros selected afterwards: no branch around a load)
    const int o01 = jr ? 1 : 0, o10 = ir ? Wo : 0, o11 = o01 + o10;
    for (int co0 = 0; co0 < Cout; co0 += UB) {
        float dv[UB][4];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const bool okc = co0 + u < Cout;
            const float* dc = d + (size_t)(okc ? co0 + u : co0) * cs;
            dv[u][0] = dc[0];
            dv[u][1] = dc[o01];
            dv[u][2] = dc[o10];
            dv[u][3] = dc[o11];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            dv[u][1] = jr ? dv[u][1] : 0.f;
            dv[u][2] = ir ? dv[u][2] : 0.f;
            dv[u][3] = (ir && jr) ? dv[u][3] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (co0 + u >= Cout) break;
            const int co = co0 + u;
            const float d00 = dv[u][0], d01 = dv[u][1], d10 = dv[u][2], d11 = dv[u][3];
#pragma unroll
            for (int c = 0; c < CIB; ++c) {
                const f32x4* wp = reinterpret_cast<const f32x4*>(wl + (co * CIB + c) * 12);
                const f32x4 w0 = wp[0], w1 = wp[1], w2 = wp[2];       // taps 0-3 | 4-7 | 8
                // dx(2i + a, 2j + b) = sum over kh = (a + 1) mod 2 .., kw likewise of dy((2i + a + 1 - kh) / 2, ..) w[kh][kw]
                a00[c] += d00 * w1[0];                                              // w[1][1]
                a01[c] += d00 * w1[1] + d01 * w0[3];                                // w[1][2], w[1][0]
                a10[c] += d00 * w1[3] + d10 * w0[1];                                // w[2][1], w[0][1]
                a11[c] += d00 * w2[0] + d01 * w1[2] + d10 * w0[2] + d11 * w0[0];    // w[2][2], w[2][0], w[0][2], w[0][0]
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CIB; ++c) {
        if (ci0 + c >= cin_n) break;
        float* o = dx + (((size_t)n * dxC + ci0 + c) * H + 2 * i) * W + 2 * j;
        *reinterpret_cast<float2*>(o) = make_float2(a00[c], a01[c]);
        *reinterpret_cast<float2*>(o + W) = make_float2(a10[c], a11[c]);
    }
}

// ---- ... its forward where the grid fills the chip (the encoder's first two layers: 4 -> 32 at 64 x 64, 32 -> 32 at 32 x 32, K B = 224
// images): the implicit-GEMM kernel runs a 64-wide output-channel tile half empty and a contraction of 36 / 288 as 16-wide chunks
// (36 / 47 us for 0.5 / 1.1 GFLOP).  Here a thread owns one output pixel and COB output channels and walks the input channels:
// 9 window loads (stride-2 rows: neighbouring lanes overlap, the L1 serves them) feed 9 COB multiply-adds, weights broadcast from
// LDS, bias + activation in the epilogue.  Bound by its stores / the window loads, not by arithmetic.
template <int COB>
__global__ void __launch_bounds__(256)
conv3x3s2_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int act,
                           float* __restrict__ y, int N, int Cin, int Cout, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [Cin][COB][12]: 9 taps, padded to 3 float4
    const int Ho = H >> 1, Wo = W >> 1, HoWo = Ho * Wo;
    const int co0 = blockIdx.x * COB;          // (x: the channel blocks that read the same windows are dispatched together)
    for (int e = threadIdx.x; e < Cin * COB * 9; e += 256) {
        const int t = e % 9, c = (e / 9) % COB, ci = e / (9 * COB);
        wl[(ci * COB + c) * 12 + t] = co0 + c < Cout ? w[((size_t)(co0 + c) * Cin + ci) * 9 + t] : 0.f;
    }
    __syncthreads();
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= N * HoWo) return;
    const int n = q / HoWo, p = q - n * HoWo;
    const int i = p / Wo, j = p - i * Wo;
    // window rows 2i - 1 .. 2i + 1, columns 2j - 1 .. 2j + 1: only the first row / column can be outside (even H, W); clamped
    // addresses, the zeros selected afterwards
    const bool top = i > 0, left = j > 0;
    const float* xp = x + (size_t)n * Cin * H * W + (size_t)(2 * i) * W + 2 * j;
    const int orow = top ? -W : 0, ocol = left ? -1 : 0;
    float acc[COB];
#pragma unroll
    for (int c = 0; c < COB; ++c) acc[c] = 0.f;
#pragma unroll 2
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xc = xp + (size_t)ci * H * W;
        float xw[9];
        xw[0] = xc[orow + ocol]; xw[1] = xc[orow]; xw[2] = xc[orow + 1];
        xw[3] = xc[ocol];        xw[4] = xc[0];    xw[5] = xc[1];
        xw[6] = xc[W + ocol];    xw[7] = xc[W];    xw[8] = xc[W + 1];
        if (!top) { xw[0] = 0.f; xw[1] = 0.f; xw[2] = 0.f; }
        if (!left) { xw[0] = 0.f; xw[3] = 0.f; xw[6] = 0.f; }
#pragma unroll
        for (int c = 0; c < COB; ++c) {
            const f32x4* wp = reinterpret_cast<const f32x4*>(wl + (ci * COB + c) * 12);
            const f32x4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
            float a = acc[c];
            a = fmaf(xw[0], w0[0], a); a = fmaf(xw[1], w0[1], a); a = fmaf(xw[2], w0[2], a);
            a = fmaf(xw[3], w0[3], a); a = fmaf(xw[4], w1[0], a); a = fmaf(xw[5], w1[1], a);
            a = fmaf(xw[6], w1[2], a); a = fmaf(xw[7], w1[3], a); a = fmaf(xw[8], w2[0], a);
            acc[c] = a;
        }
    }
    float* yp = y + ((size_t)n * Cout + co0) * HoWo + p;
#pragma unroll
    for (int c = 0; c < COB; ++c) {
        if (co0 + c >= Cout) break;
        float v = acc[c] + (bias ? bias[co0 + c] : 0.f);
        if (act == 1) v = v > 0.f ? v : 0.f;
        else if (act == 2) v = v > 0.f ? v : expm1f(v);
        yp[(size_t)c * HoWo] = v;
    }
}

// the launch above where it pays: conv3x3 stride 2 pad 1 on even grids, <= 64 input channels, a grid of >= 512 workgroups
static bool fwd_small_ok(int N, int Cin, int Cout, int H, int W, int k, int stride, int pad) {
    static const char* env = getenv("GENESIS_DCONV_FWD_SMALL");
    if (env && env[0] == '0') return false;
    if (k != 3 || stride != 2 || pad != 1 || (H & 1) || (W & 1) || Cin > 64 || (double)N * H * W >= 2.0e9) return false;
    const long wgs = (long)gx_ceil_div(Cout, 8) * gx_ceil_div(N * (H / 2) * (W / 2), 256);
    return wgs >= 512 && gx_ceil_div(N * (H / 2) * (W / 2), 256) <= 65535;
}

// ---- ... and its weight gradient: dw[co][ci][kh][kw] = sum_{n,i,j} dy[n][co][i][j] x[n][ci][2i - 1 + kh][2j - 1 + kw].
// Lanes = output pixels (coalesced dy rows, stride-2 x windows), a thread accumulates 9 taps x COB output channels of ONE
// input channel over its pixels (9 + COB loads feed 9 COB FMAs), the workgroup's 256 partial sets are summed through LDS,
// split-K over gridDim.x with a fixed-order final reduce (igemm_reduce_kernel).  Generic kernel: 44-157 us per layer.
template <int COB>
__global__ void __launch_bounds__(256)
conv3x3s2_wgrad_small_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int N,
                             int Cin, int Cout, int H, int W) {
    constexpr int NV = COB * 9;
    static_assert(NV * 2 <= 256, "one reducer thread per (value, quarter)");
    __shared__ float red[(NV / 2) * 257];
    __shared__ float red2[NV * 4];
    const int Ho = H >> 1, Wo = W >> 1, HoWo = Ho * Wo;
    const int ci = blockIdx.y, co0 = blockIdx.z * COB;
    const int P = N * HoWo;
    const int chunk = (P + gridDim.x - 1) / gridDim.x;
    const int begin = blockIdx.x * chunk, end = begin + chunk < P ? begin + chunk : P;
    float acc[COB][9];
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    for (int p = begin + threadIdx.x; p < end; p += 256) {
        const int n = p / HoWo, r = p - n * HoWo;
        const int i = r / Wo, j = r - i * Wo;
        const float* xp = x + (((size_t)n * Cin + ci) * H + 2 * i) * W + 2 * j;          // window centre row 2i, col 2j
        float xw[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bool ok = (kh > 0 || i > 0) && (kw > 0 || j > 0);                  // row 2i - 1 / col 2j - 1 exist
                xw[kh * 3 + kw] = ok ? xp[(kh - 1) * W + (kw - 1)] : 0.f;
            }
#if defined(GX_DIAG_NO_OVERLAP)     /* diagnosis (tools/diag_shared_gpu2.py, DESIGN.md finding 48): the (2i+1, 2j..2j+1) pair through an explicit load.
                                       1: destination cannot be the address pair, waited for at once; 2: destination IS the address pair, waited
                                       for at once; 3: IS the address pair, waited for after the dy loads; 4: cannot be, waited for after them */
        unsigned long long gx_u = reinterpret_cast<unsigned long long>(xp + W);
        typedef float gx_f2 __attribute__((ext_vector_type(2)));
        gx_f2 gx_pr = {0.f, 0.f};
#if GX_DIAG_NO_OVERLAP == 1
        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(gx_pr) : "v"(gx_u) : "memory");
#elif GX_DIAG_NO_OVERLAP == 2
        asm volatile("global_load_dwordx2 %0, %0, off\n\ts_waitcnt vmcnt(0)" : "+v"(gx_u) : : "memory");
#elif GX_DIAG_NO_OVERLAP == 3
        asm volatile("global_load_dwordx2 %0, %0, off" : "+v"(gx_u) : : "memory");
#else
        asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(gx_pr) : "v"(gx_u) : "memory");
#endif
#endif
        const float* dp = dy + ((size_t)n * Cout + co0) * HoWo + r;
#if defined(GX_DIAG_NO_OVERLAP)
        float gx_d[COB];
#pragma unroll
        for (int c = 0; c < COB; ++c) gx_d[c] = co0 + c < Cout ? dp[(size_t)c * HoWo] : 0.f;
#if GX_DIAG_NO_OVERLAP == 3
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(gx_u) : : "memory");
#elif GX_DIAG_NO_OVERLAP == 4
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(gx_pr) : : "memory");
#endif
#if GX_DIAG_NO_OVERLAP == 2 || GX_DIAG_NO_OVERLAP == 3
        gx_pr[0] = __builtin_bit_cast(float, (unsigned)(gx_u & 0xffffffffu)); gx_pr[1] = __builtin_bit_cast(float, (unsigned)(gx_u >> 32));
#endif
        xw[7] = gx_pr[0]; xw[8] = gx_pr[1];
#endif
#pragma unroll
        for (int c = 0; c < COB; ++c) {
#if defined(GX_DIAG_NO_OVERLAP)
            const float d = gx_d[c];
#else
            const float d = co0 + c < Cout ? dp[(size_t)c * HoWo] : 0.f;
#endif
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[c][t] += d * xw[t];
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {              // 36 values x 256 threads per pass (37 KB of LDS)
#pragma unroll
        for (int c = 0; c < COB / 2; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) red[(c * 9 + t) * 257 + threadIdx.x] = acc[half * (COB / 2) + c][t];
        __syncthreads();
        if (threadIdx.x < NV * 2) {
            const int v = threadIdx.x >> 2, q = threadIdx.x & 3;
            float sum = 0.f;
            for (int l = 0; l < 64; ++l) sum += red[v * 257 + q * 64 + ((l + 16 * q) & 63)];   // quarters on distinct banks
            red2[half * NV * 2 + threadIdx.x] = sum;
        }
        __syncthreads();
    }
    if (threadIdx.x < NV) {
        const int c = threadIdx.x / 9, t = threadIdx.x - c * 9;
        if (co0 + c < Cout) {
            const float* r4 = red2 + threadIdx.x * 4;
            part[(size_t)blockIdx.x * Cout * Cin * 9 + ((size_t)(co0 + c) * Cin + ci) * 9 + t] = (r4[0] + r4[1]) + (r4[2] + r4[3]);
        }
    }
}

extern "C" {

size_t gx_conv3x3s2_wgrad_small_ws_bytes(int N, int Cin, int Cout, int H, int W) {
    (void)N; (void)H; (void)W;
    return (size_t)64 * Cout * Cin * 9 * sizeof(float);
}

/* conv3x3 stride 2 pad 1 (even H, W) weight gradient on the vector ALUs: dw [Cout,Cin,3,3] from x [N,Cin,H,W], dy
 * [N,Cout,H/2,W/2]; ws: gx_conv3x3s2_wgrad_small_ws_bytes. */
int gx_conv3x3s2_wgrad_small(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, void* ws,
                             size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(x && dy && dw && ws, "gx_conv3x3s2_wgrad_small: null pointer");
    GX_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && Cin <= 65535,
                 "gx_conv3x3s2_wgrad_small: even H, W");
    GX_CHECK_ARG(ws_bytes >= gx_conv3x3s2_wgrad_small_ws_bytes(N, Cin, Cout, H, W), "gx_conv3x3s2_wgrad_small: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int P = N * (H / 2) * (W / 2);
    const int blocks = Cin * gx_ceil_div(Cout, 8);
    int S = gx_ceil_div(2048, blocks);                  // ~8 workgroups per CU
    if (S > 64) S = 64;
    if (S > gx_ceil_div(P, 1024)) S = gx_ceil_div(P, 1024);
    if (S < 1) S = 1;
    {
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * (double)Cin * 9 * (H / 2) * (W / 2), 4.0 * N * ((double)Cin * H * W + (double)Cout * (H / 2) * (W / 2)));
        hipLaunchKernelGGL(conv3x3s2_wgrad_small_kernel<8>, dim3(S, Cin, gx_ceil_div(Cout, 8)), dim3(256), 0, s, x, dy, (float*)ws,
                           N, Cin, Cout, H, W);
    }
    GX_CHECK_LAUNCH("gx_conv3x3s2_wgrad_small");
    {
        const int total = Cout * Cin * 9;
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * (S + 1.0) * total);
        hipLaunchKernelGGL(igemm_reduce_kernel, dim3(gx_ceil_div(total, 256)), dim3(256), 0, s, (const float*)ws, dw, total, S);
    }
    GX_CHECK_LAUNCH("gx_conv3x3s2_wgrad_small(reduce)");
    return GX_OK;
}

/* conv3x3 stride 2 pad 1 (even H, W) data gradient on the vector ALUs: dx [N,Cin,H,W] from dy [N,Cout,H/2,W/2], w
 * [Cout,Cin,3,3]; only the first cin_n channels of dx are computed and written (the others are left untouched). */
int gx_conv3x3s2_dgrad_small(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, int cin_n,
                             gx_stream_t stream) {
    return gx_conv3x3s2_dgrad_small_ex(dy, w, dx, N, Cin, Cout, H, W, cin_n, Cin, stream);
}

/* _ex: dx has dx_channels (cin_n <= dx_channels) channels per image -- dx_channels = cin_n: a compact [N,cin_n,H,W] gradient
 * of the leading input channels alone. */
int gx_conv3x3s2_dgrad_small_ex(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, int cin_n,
                                int dx_channels, gx_stream_t stream) {
    GX_CHECK_ARG(dx_channels >= cin_n, "gx_conv3x3s2_dgrad_small_ex: dx_channels (%d) < cin_n (%d)", dx_channels, cin_n);
    const int dxC = dx_channels;
    GX_CHECK_ARG(dy && w && dx, "gx_conv3x3s2_dgrad_small: null pointer");
    GX_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && (double)N * H * W < 4.0 * 256 * 65535,
                 "gx_conv3x3s2_dgrad_small: even H, W; N H W / 1024 <= 65535");
    GX_CHECK_ARG(cin_n >= 1 && cin_n <= Cin && Cout * 4 * 12 * 4 <= 64 * 1024, "gx_conv3x3s2_dgrad_small: 1 <= cin_n <= Cin, Cout <= 341");
    hipStream_t s = (hipStream_t)stream;
    const int npix = (H / 2) * (W / 2);
    {
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * (double)cin_n * 9 * npix, 4.0 * N * ((double)cin_n * H * W + (double)Cout * npix));
        const bool few = (long)gx_ceil_div(cin_n, cin_n <= 2 ? 2 : 4) * gx_ceil_div(N * npix, 256) < 512;
        if (cin_n <= 2) {
            const dim3 grid(gx_ceil_div(cin_n, 2), gx_ceil_div(N * npix, 256));
            if (few) hipLaunchKernelGGL((conv3x3s2_dgrad_small_kernel<2, 8>), grid, dim3(256), (size_t)Cout * 2 * 12 * 4, s, dy, w, dx, N, Cin, Cout, H, W, cin_n, dxC);
            else hipLaunchKernelGGL((conv3x3s2_dgrad_small_kernel<2, 1>), grid, dim3(256), (size_t)Cout * 2 * 12 * 4, s, dy, w, dx, N, Cin, Cout, H, W, cin_n, dxC);
        } else {
            const dim3 grid(gx_ceil_div(cin_n, 4), gx_ceil_div(N * npix, 256));
            if (few) hipLaunchKernelGGL((conv3x3s2_dgrad_small_kernel<4, 8>), grid, dim3(256), (size_t)Cout * 4 * 12 * 4, s, dy, w, dx, N, Cin, Cout, H, W, cin_n, dxC);
            else hipLaunchKernelGGL((conv3x3s2_dgrad_small_kernel<4, 1>), grid, dim3(256), (size_t)Cout * 4 * 12 * 4, s, dy, w, dx, N, Cin, Cout, H, W, cin_n, dxC);
        }
    }
    GX_CHECK_LAUNCH("gx_conv3x3s2_dgrad_small");
    return GX_OK;
}

size_t gx_conv2d_direct_fwd_ws_bytes(int N, int Cin, int Cout, int H, int W, int k, int stride, int pad) {
    IG g;
    if (ig_geom("gx_conv2d_direct_fwd_ws_bytes", &g, 0, N, Cin, Cout, H, W, k, stride, pad)) return 0;
    const int ns = fwd_splits(g);
    return ns > 1 ? (size_t)ns * g.M * g.Ncols * sizeof(float) : 0;
}

int gx_conv2d_direct_fwd_ws(const float* x, const float* w, const float* bias, int act, float* y, int N, int Cin,
                            int Cout, int H, int W, int k, int stride, int pad, void* ws, size_t ws_bytes,
                            gx_stream_t stream) {
    GX_CHECK_ARG(x && w && y && act >= 0 && act <= 2, "gx_conv2d_direct_fwd: null pointer / bad act");
    IG g;
    int rc = ig_geom("gx_conv2d_direct_fwd", &g, 0, N, Cin, Cout, H, W, k, stride, pad);
    if (rc) return rc;
    const int ns = fwd_splits(g);
    if (ns <= 1 || !ws || ws_bytes < (size_t)ns * g.M * g.Ncols * sizeof(float))
        return gx_conv2d_direct_fwd(x, w, bias, act, y, N, Cin, Cout, H, W, k, stride, pad, stream);
    hipStream_t s = (hipStream_t)stream;
    g.chunks_per_split = gx_ceil_div(gx_ceil_div(g.K, BK), ns);
    {
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * Cin * k * k * g.Ho * g.Wo, 4.0 * N * (Cin * H * W + Cout * g.Ho * g.Wo));
        hipLaunchKernelGGL(igemm_kernel<0>, dim3(gx_ceil_div(g.Ncols, BN), gx_ceil_div(g.M, BM), ns), dim3(256), 0, s, w,
                           x, (const float*)nullptr, 0, (float*)ws, g);
    }
    GX_CHECK_LAUNCH("gx_conv2d_direct_fwd(split)");
    {
        const int total = g.M * g.Ncols;
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * (ns + 1.0) * total);
        hipLaunchKernelGGL(igemm_reduce_act_kernel, dim3(gx_ceil_div(total, 256)), dim3(256), 0, s, (const float*)ws, bias, act, y,
                           total, ns, g.M, g.Ho * g.Wo);
    }
    GX_CHECK_LAUNCH("gx_conv2d_direct_fwd(reduce)");
    return GX_OK;
}

int gx_conv2d_direct_fwd(const float* x, const float* w, const float* bias, int act, float* y, int N, int Cin,
                         int Cout, int H, int W, int k, int stride, int pad, gx_stream_t stream) {
    GX_CHECK_ARG(x && w && y && act >= 0 && act <= 2, "gx_conv2d_direct_fwd: null pointer / bad act");
    IG g;
    int rc = ig_geom("gx_conv2d_direct_fwd", &g, 0, N, Cin, Cout, H, W, k, stride, pad);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (fwd_small_ok(N, Cin, Cout, H, W, k, stride, pad)) {      // the ComponentVAE encoder's chip-filling layers: vector ALUs
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * Cin * k * k * g.Ho * g.Wo, 4.0 * N * (Cin * H * W + Cout * g.Ho * g.Wo));
        hipLaunchKernelGGL(conv3x3s2_fwd_small_kernel<8>, dim3(gx_ceil_div(Cout, 8), gx_ceil_div(N * (H / 2) * (W / 2), 256)), dim3(256),
                           (size_t)Cin * 8 * 12 * 4, s, x, w, bias, act, y, N, Cin, Cout, H, W);
        GX_CHECK_LAUNCH("gx_conv2d_direct_fwd(small)");
        return GX_OK;
    }
    {
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * Cin * k * k * g.Ho * g.Wo, 4.0 * N * (Cin * H * W + Cout * g.Ho * g.Wo));
        hipLaunchKernelGGL(igemm_kernel<0>, dim3(gx_ceil_div(g.Ncols, BN), gx_ceil_div(g.M, BM), 1), dim3(256), 0, s, w,
                           x, bias, act, y, g);
    }
    GX_CHECK_LAUNCH("gx_conv2d_direct_fwd");
    return GX_OK;
}

int gx_conv2d_direct_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, int k,
                           int stride, int pad, gx_stream_t stream) {
    GX_CHECK_ARG(dy && w && dx, "gx_conv2d_direct_dgrad: null pointer");
    IG g;
    int rc = ig_geom("gx_conv2d_direct_dgrad", &g, 1, N, Cin, Cout, H, W, k, stride, pad);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * Cin * k * k * g.Ho * g.Wo, 4.0 * N * (Cin * H * W + Cout * g.Ho * g.Wo));
        hipLaunchKernelGGL(igemm_kernel<1>, dim3(gx_ceil_div(g.Ncols, BN), gx_ceil_div(g.M, BM), 1), dim3(256), 0, s, w,
                           dy, (const float*)nullptr, 0, dx, g);
    }
    GX_CHECK_LAUNCH("gx_conv2d_direct_dgrad");
    return GX_OK;
}

size_t gx_conv2d_direct_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W, int k, int stride, int pad) {
    IG g;
    if (ig_geom("gx_conv2d_direct_wgrad_ws_bytes", &g, 2, N, Cin, Cout, H, W, k, stride, pad) != GX_OK) return 0;
    return (size_t)wgrad_splits(g) * g.M * g.Ncols * sizeof(float);
}

int gx_conv2d_direct_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, int k,
                           int stride, int pad, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(x && dy && dw && ws, "gx_conv2d_direct_wgrad: null pointer");
    IG g;
    int rc = ig_geom("gx_conv2d_direct_wgrad", &g, 2, N, Cin, Cout, H, W, k, stride, pad);
    if (rc) return rc;
    const int nsplit = wgrad_splits(g);
    GX_CHECK_ARG(ws_bytes >= (size_t)nsplit * g.M * g.Ncols * sizeof(float), "gx_conv2d_direct_wgrad: workspace too small");
    g.chunks_per_split = gx_ceil_div(gx_ceil_div(g.K, BK), nsplit);
    const int zs = gx_ceil_div(gx_ceil_div(g.K, BK), g.chunks_per_split);
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DCONV, s, 2.0 * N * Cout * Cin * k * k * g.Ho * g.Wo, 4.0 * N * (Cin * H * W + Cout * g.Ho * g.Wo));
        hipLaunchKernelGGL(igemm_kernel<2>, dim3(gx_ceil_div(g.Ncols, BN), gx_ceil_div(g.M, BM), zs), dim3(256), 0, s, dy,
                           x, (const float*)nullptr, 0, (float*)ws, g);
    }
    GX_CHECK_LAUNCH("gx_conv2d_direct_wgrad");
    {
        const int total = g.M * g.Ncols;
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * (zs + 1.0) * total);
        hipLaunchKernelGGL(igemm_reduce_kernel, dim3(gx_ceil_div(total, 256)), dim3(256), 0, s, (const float*)ws, dw, total,
                           zs);
    }
    GX_CHECK_LAUNCH("gx_conv2d_direct_wgrad(reduce)");
    return GX_OK;
}

}  // extern "C"
