// Implicit-GEMM convolutions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, gfx950).
//
// One "tap-conv" kernel covers every dense spatial conv on the GENESIS-V2 path
// (reference: modules/blocks.py:159-165 ConvGNReLU convs, models/genesisv2_config.py:89-99
// ConvTranspose2d k5 s2 p2 op1) and their data gradients:
//
//     out[m][pix] = sum_{k, t}  Wp[t][k][m] * in[k][gather(pix, t)]
//
//   M_C3  : conv3x3 s1 p1 forward, and (with flipped/transposed packing) its dgrad.
//   M_DT0 : deconv5x5 s2 forward, output rows 2r   (kh = 0,2,4), both column parities.
//   M_DT1 : deconv5x5 s2 forward, output rows 2r+1 (kh = 1,3),   both column parities.
//   M_DG  : deconv5x5 s2 dgrad = 5x5 stride-2 conv over dy; dy is de-interleaved into
//           4 parity planes while being staged so every tap reads unit-stride from LDS.
//
// GEMM mapping per workgroup (256 threads = 4 wavefronts of 64):
//   M = 64 output channels, N = 256 base-grid pixels (G images x TH rows x TW cols),
//   K = KC input channels per staged chunk x NT taps.  The input halo tile is staged ONCE
//   per chunk and re-used by all NT taps from LDS (9-25x re-use); each wave owns a
//   64(M) x 64(N) sub-tile = 2x2 MFMA 32x32 accumulators (per output parity class).
// Weight-gradient kernels (wgrad) use the transposed mapping M = Cout, N = Cin, K = pixels
// with deterministic split-K partials + a reduce kernel.
//
// fp32 in / fp32 accumulate: results are an fmaf chain per output (exact fp32), which is
// what the stated fp32 parity tolerance of the path needs (no bf16/xf32 shortcuts).
#include "gx_common.h"

namespace {

enum { M_C3 = 0, M_DT0 = 1, M_DT1 = 2, M_DG = 3 };

template <int MODE> struct TapCfg;
template <> struct TapCfg<M_C3> {
    static constexpr int NT = 9, NCLS = 1, KC = 8, PLANES = 1;
    __host__ __device__ static constexpr int ro(int t) { return t / 3; }
    __host__ __device__ static constexpr int co(int t) { return t % 3; }
    __host__ __device__ static constexpr int cls(int) { return 0; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
};
// ConvTranspose k5 s2 p2 op1: out[2r+a][2c+b] += x[r+dr][c+dc] * W[kh][kw] with kh = a (mod 2),
// dr = (a+2-kh)/2, i.e. halo row offset ro = dr+1 = 2 - kh/2 (same for columns).
template <> struct TapCfg<M_DT0> {
    static constexpr int NT = 15, NCLS = 2, KC = 8, PLANES = 1;
    __host__ __device__ static constexpr int ro(int t) { return 2 - t / 5; }        // kh = 2*(t/5)
    __host__ __device__ static constexpr int co(int t) { return 2 - (t % 5) / 2; }  // kw = t%5
    __host__ __device__ static constexpr int cls(int t) { return (t % 5) & 1; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
};
template <> struct TapCfg<M_DT1> {
    static constexpr int NT = 10, NCLS = 2, KC = 8, PLANES = 1;
    __host__ __device__ static constexpr int ro(int t) { return 2 - t / 5; }        // kh = 2*(t/5)+1
    __host__ __device__ static constexpr int co(int t) { return 2 - (t % 5) / 2; }
    __host__ __device__ static constexpr int cls(int t) { return (t % 5) & 1; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
};
// dgrad of the deconv: dx[r][c] = sum dy[2r-2+kh][2c-2+kw] W[kh][kw]; plane = (kh&1, kw&1),
// in-plane offset (kh/2, kw/2).
template <> struct TapCfg<M_DG> {
    static constexpr int NT = 25, NCLS = 1, KC = 4, PLANES = 4;
    __host__ __device__ static constexpr int ro(int t) { return (t / 5) / 2; }
    __host__ __device__ static constexpr int co(int t) { return (t % 5) / 2; }
    __host__ __device__ static constexpr int cls(int) { return 0; }
    __host__ __device__ static constexpr int plane(int t) { return ((t / 5) & 1) * 2 + ((t % 5) & 1); }
};

struct ConvGeom {
    int N;            // images
    int K;            // reduction channels (actual)
    int M;            // output channels (actual)
    int Kpad, Mpad;   // packed-weight dims
    int Hb, Wb;       // base grid (pixel tile grid)
    int Hi, Wi;       // input tensor spatial dims
    int Ho, Wo;       // output tensor spatial dims
    int lTH, lTW, lG; // log2 of tile rows / cols / images per tile
    int tiles_h, tiles_w;
    int par_a;        // deconv fwd: output row parity handled by this launch
};

// halo-tile positions staged per thread per channel (tile <= MAXPOS*256 floats / channel)
template <int MODE> struct MaxPos { static constexpr int V = (MODE == M_DG) ? 16 : 8; };

template <int MODE>
__global__ void __launch_bounds__(256, 2)
tapconv_kernel(const float* __restrict__ in, const float* __restrict__ wp,
               const float* __restrict__ bias, float* __restrict__ out, ConvGeom g) {
    using TC = TapCfg<MODE>;
    constexpr int NT = TC::NT, NCLS = TC::NCLS, KC = TC::KC, PLANES = TC::PLANES;
    constexpr int MAXPOS = MaxPos<MODE>::V;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int TH = 1 << g.lTH, TW = 1 << g.lTW, G = 1 << g.lG;
    const int HS = TW + 2;                         // halo row stride
    const int PLS = G * (TH + 2) * HS;             // one plane of the halo tile
    const int CHS = PLANES * PLS;                  // per-channel LDS stride
    float* in_tile = lds;                          // [KC][CHS]
    float* w_tile = lds + KC * CHS;                // [NT][KC][64]

    // ---- which tile ----
    int tile = blockIdx.x;
    const int tw_i = tile % g.tiles_w; tile /= g.tiles_w;
    const int th_i = tile % g.tiles_h; tile /= g.tiles_h;
    const int img0 = tile * G;
    const int R0 = th_i * TH, C0 = tw_i * TW;
    const int m0 = blockIdx.y * 64;

    const size_t in_img_stride = (size_t)g.K * g.Hi * g.Wi;
    const float* in_blk = in + (size_t)img0 * in_img_stride;
    const int HiWi = g.Hi * g.Wi;

    // ---- per-thread staging positions (computed once; only the channel term changes) ----
    int goff[MAXPOS];
#pragma unroll
    for (int q = 0; q < MAXPOS; ++q) {
        const int pos = tid + q * 256;
        int off = -1;
        if (pos < CHS) {
            int rem = pos;
            const int plane = rem / PLS; rem -= plane * PLS;
            const int gi = rem / ((TH + 2) * HS); rem -= gi * (TH + 2) * HS;
            const int i = rem / HS;
            const int j = rem - i * HS;
            int row, col;
            if (MODE == M_DG) {
                row = 2 * (R0 + i) - 2 + (plane >> 1);
                col = 2 * (C0 + j) - 2 + (plane & 1);
            } else {
                row = R0 - 1 + i;
                col = C0 - 1 + j;
            }
            if (img0 + gi < g.N && row >= 0 && row < g.Hi && col >= 0 && col < g.Wi)
                off = gi * (int)in_img_stride + row * g.Wi + col;
        }
        goff[q] = off;
    }

    // ---- per-lane fragment offsets ----
    // A (weights): lane -> w_tile[t][2kk + (lane>>5)][mi*32 + (lane&31)]
    const int a_off = (lane >> 5) * 64 + (lane & 31);
    // B (input):   lane -> in_tile[2kk + (lane>>5)][plane][halo(pixel) + tap]
    int b_off[2];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int p = wave * 64 + nj * 32 + (lane & 31);
        const int c = p & (TW - 1);
        const int r = (p >> g.lTW) & (TH - 1);
        const int gi = p >> (g.lTW + g.lTH);
        b_off[nj] = (lane >> 5) * CHS + (gi * (TH + 2) + r) * HS + c;
    }

    f32x16 acc[NCLS][2][2];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][i][j][e] = 0.f;

    for (int ch0 = 0; ch0 < g.Kpad; ch0 += KC) {
        __syncthreads();
        // stage the halo tile of KC channels
#pragma unroll
        for (int ch = 0; ch < KC; ++ch) {
            const bool chv = (ch0 + ch) < g.K;
            const float* src = in_blk + (size_t)(ch0 + ch) * HiWi;
#pragma unroll
            for (int q = 0; q < MAXPOS; ++q) {
                const int pos = tid + q * 256;
                if (pos < CHS) {
                    float v = 0.f;
                    if (chv && goff[q] >= 0) v = src[goff[q]];
                    in_tile[ch * CHS + pos] = v;
                }
            }
        }
        // stage the weight slab: w_tile[t][kc][0..63] <- Wp[t][ch0+kc][m0..m0+63]
        for (int i4 = tid; i4 < NT * KC * 16; i4 += 256) {
            const int q = i4 & 15;
            const int kc = (i4 >> 4) & (KC - 1);
            const int t = i4 / (16 * KC);
            const float4 v = *reinterpret_cast<const float4*>(
                wp + ((size_t)t * g.Kpad + ch0 + kc) * g.Mpad + m0 + q * 4);
            *reinterpret_cast<float4*>(w_tile + (t * KC + kc) * 64 + q * 4) = v;
        }
        __syncthreads();

#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int toff = TC::plane(t) * PLS + TC::ro(t) * HS + TC::co(t);
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                const float a0 = w_tile[(t * KC + 2 * kk) * 64 + a_off];
                const float a1 = w_tile[(t * KC + 2 * kk) * 64 + a_off + 32];
                const float b0 = in_tile[2 * kk * CHS + b_off[0] + toff];
                const float b1 = in_tile[2 * kk * CHS + b_off[1] + toff];
                const int c = TC::cls(t);
                acc[c][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[c][0][0], 0, 0, 0);
                acc[c][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[c][0][1], 0, 0, 0);
                acc[c][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[c][1][0], 0, 0, 0);
                acc[c][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[c][1][1], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: C/D layout col = lane&31 (pixel), row = (reg&3)+8*(reg>>2)+4*(lane>>5) (channel) ----
    const size_t out_img_stride = (size_t)g.M * g.Ho * g.Wo;
    const int HoWo = g.Ho * g.Wo;
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int p = wave * 64 + nj * 32 + (lane & 31);
        const int c = p & (TW - 1);
        const int r = (p >> g.lTW) & (TH - 1);
        const int gi = p >> (g.lTW + g.lTH);
        const int n = img0 + gi;
        if (n >= g.N) continue;
        int orow, ocol;
        if (NCLS == 2) { orow = 2 * (R0 + r) + g.par_a; ocol = 2 * (C0 + c); }
        else { orow = R0 + r; ocol = C0 + c; }
        float* obase = out + (size_t)n * out_img_stride + (size_t)orow * g.Wo + ocol;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m < g.M) {
                    const float bv = bias ? bias[m] : 0.f;
                    if (NCLS == 2) {
                        float2 v;
                        v.x = acc[0][mi][nj][reg] + bv;
                        v.y = acc[NCLS - 1][mi][nj][reg] + bv;
                        *reinterpret_cast<float2*>(obase + (size_t)m * HoWo) = v;
                    } else {
                        obase[(size_t)m * HoWo] = acc[0][mi][nj][reg] + bv;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ weight packing
// Wp[t][k][m] (k padded to Kpad, m padded to Mpad, zero filled).
//   pack 0: conv3x3 fwd    W[co][ci][3][3]  -> m=co, k=ci, t=kh*3+kw
//   pack 1: conv3x3 dgrad  W[co][ci][3][3]  -> m=ci, k=co, t=(kh,kw) reads W[..][2-kh][2-kw]
//   pack 2/3: deconv fwd rows a=0/1, W[ci][co][5][5] -> m=co, k=ci, t=khi*5+kw, kh=2*khi+a
//   pack 4: deconv dgrad   W[ci][co][5][5]  -> m=ci, k=co, t=kh*5+kw
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int pack,
                                    int Co, int Ci, int NT, int Kpad, int Mpad) {
    const int total = NT * Kpad * Mpad;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int m = idx % Mpad;
        const int k = (idx / Mpad) % Kpad;
        const int t = idx / (Mpad * Kpad);
        float v = 0.f;
        if (pack == 0) {
            if (m < Co && k < Ci) v = w[((size_t)m * Ci + k) * 9 + t];
        } else if (pack == 1) {
            if (m < Ci && k < Co) v = w[((size_t)k * Ci + m) * 9 + (8 - t)];
        } else if (pack == 2 || pack == 3) {
            const int kh = 2 * (t / 5) + (pack - 2), kw = t % 5;
            if (m < Co && k < Ci) v = w[((size_t)k * Co + m) * 25 + kh * 5 + kw];
        } else {
            if (m < Ci && k < Co) v = w[((size_t)m * Co + k) * 25 + t];
        }
        wp[idx] = v;
    }
}

// ------------------------------------------------------------------ weight gradients
// D[i = A-channel][j = B-channel] per tap, K = pixels.
//   A source: dy sampled at (SA*row + pa, SA*col + pb)  (SA=1 conv3x3; SA=2 deconv class (pa,pb))
//   B source: x with a 1-pixel halo on the base grid.
enum { W_C3 = 0, W_D00 = 1, W_D01 = 2, W_D10 = 3, W_D11 = 4 };
template <int WM> struct WTap;
template <> struct WTap<W_C3> {
    static constexpr int NT = 9, SA = 1, PA = 0, PB = 0;
    __host__ __device__ static constexpr int ro(int t) { return t / 3; }
    __host__ __device__ static constexpr int co(int t) { return t % 3; }
    __host__ __device__ static constexpr int gt(int t) { return t; }
};
template <int PA_, int PB_> struct WTapD {
    static constexpr int NKH = PA_ ? 2 : 3, NKW = PB_ ? 2 : 3;
    static constexpr int NT = NKH * NKW, SA = 2, PA = PA_, PB = PB_;
    __host__ __device__ static constexpr int kh(int t) { return 2 * (t / NKW) + PA_; }
    __host__ __device__ static constexpr int kw(int t) { return 2 * (t % NKW) + PB_; }
    __host__ __device__ static constexpr int ro(int t) { return 2 - kh(t) / 2; }
    __host__ __device__ static constexpr int co(int t) { return 2 - kw(t) / 2; }
    __host__ __device__ static constexpr int gt(int t) { return kh(t) * 5 + kw(t); }
};
template <> struct WTap<W_D00> : WTapD<0, 0> {};
template <> struct WTap<W_D01> : WTapD<0, 1> {};
template <> struct WTap<W_D10> : WTapD<1, 0> {};
template <> struct WTap<W_D11> : WTapD<1, 1> {};

struct WgradGeom {
    int N;
    int CA, CB;          // channels of A (dy) and B (x)
    int CApad, CBpad;    // padded to 64
    int Hb, Wb;          // base grid = B spatial dims
    int Ha, Wa;          // A spatial dims (= SA * base)
    int lTH, lTW, lG;
    int tiles_h, tiles_w, ntiles;
    int nsplit;
    int Ttot;            // taps in the partial buffer (9 or 25)
};

template <int WM>
__global__ void __launch_bounds__(256, 1)
wgrad_kernel(const float* __restrict__ a_src, const float* __restrict__ b_src,
             float* __restrict__ partial, WgradGeom g) {
    using WT = WTap<WM>;
    constexpr int NT = WT::NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TH = 1 << g.lTH, TW = 1 << g.lTW, G = 1 << g.lG;
    const int PT = TH * TW * G;          // pixels per tile (64 or 128)
    const int lPT = g.lTH + g.lTW + g.lG;
    const int AS = PT + 1;               // padded A row stride (conflict-free column reads)
    const int HS = TW + 2;
    const int CHS = G * (TH + 2) * HS;
    const int BS = CHS | 1;              // odd B channel stride
    float* a_tile = lds;                 // [64][AS]
    float* b_tile = lds + 64 * AS;       // [64][BS]

    const int nbt = g.CBpad / 64;
    const int ca0 = (blockIdx.y / nbt) * 64;
    const int cb0 = (blockIdx.y % nbt) * 64;
    const int wm = wave >> 1, wn = wave & 1;

    const size_t a_img = (size_t)g.CA * g.Ha * g.Wa;
    const size_t b_img = (size_t)g.CB * g.Hb * g.Wb;
    const int HaWa = g.Ha * g.Wa, HbWb = g.Hb * g.Wb;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int a_row = (wm * 32 + (lane & 31)) * AS;
    const int b_row = (wn * 32 + (lane & 31)) * BS;

    for (int tile = blockIdx.x; tile < g.ntiles; tile += g.nsplit) {
        int tt = tile;
        const int tw_i = tt % g.tiles_w; tt /= g.tiles_w;
        const int th_i = tt % g.tiles_h; tt /= g.tiles_h;
        const int img0 = tt * G;
        const int R0 = th_i * TH, C0 = tw_i * TW;
        __syncthreads();
        // stage A: a_tile[ch][p] = dy[n][ca0+ch][SA*(R0+r)+PA][SA*(C0+c)+PB]
        for (int idx = tid; idx < 64 * PT; idx += 256) {
            const int p = idx & (PT - 1);
            const int ch = idx >> lPT;
            const int c = p & (TW - 1);
            const int r = (p >> g.lTW) & (TH - 1);
            const int gi = p >> (g.lTW + g.lTH);
            const int n = img0 + gi;
            float v = 0.f;
            if (n < g.N && ca0 + ch < g.CA)
                v = a_src[(size_t)n * a_img + (size_t)(ca0 + ch) * HaWa +
                          (WT::SA * (R0 + r) + WT::PA) * g.Wa + WT::SA * (C0 + c) + WT::PB];
            a_tile[ch * AS + p] = v;
        }
        // stage B: halo tile of 64 channels
        for (int pos = tid; pos < CHS; pos += 256) {
            int rem = pos;
            const int gi = rem / ((TH + 2) * HS); rem -= gi * (TH + 2) * HS;
            const int i = rem / HS;
            const int j = rem - i * HS;
            const int row = R0 - 1 + i, col = C0 - 1 + j;
            const int n = img0 + gi;
            const bool ok = (n < g.N && row >= 0 && row < g.Hb && col >= 0 && col < g.Wb);
            const float* src = b_src + (size_t)n * b_img + (size_t)cb0 * HbWb + row * g.Wb + col;
            for (int ch = 0; ch < 64; ++ch) {
                float v = 0.f;
                if (ok && cb0 + ch < g.CB) v = src[(size_t)ch * HbWb];
                b_tile[ch * BS + pos] = v;
            }
        }
        __syncthreads();
        // K loop over pixel pairs
#pragma unroll 2
        for (int kk = 0; kk < PT / 2; ++kk) {
            const int p = 2 * kk + (lane >> 5);
            const int c = p & (TW - 1);
            const int r = (p >> g.lTW) & (TH - 1);
            const int gi = p >> (g.lTW + g.lTH);
            const float a = a_tile[a_row + p];
            const float* bp = b_tile + b_row + (gi * (TH + 2) + r) * HS + c;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float b = bp[WT::ro(t) * HS + WT::co(t)];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // partial[split][gt][ca][cb]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* dst = partial + (((size_t)blockIdx.x * g.Ttot + WT::gt(t)) * g.CApad + ca0 + wm * 32) * g.CBpad +
                     cb0 + wn * 32 + (lane & 31);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            dst[(size_t)row * g.CBpad] = acc[t][reg];
        }
    }
}

// dW = sum over splits.  layout 0: W[ca][cb][T] (conv3x3: ca=co, cb=ci); layout 1: W[cb][ca][T] (deconv).
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nsplit,
                                    int Ttot, int CA, int CB, int CApad, int CBpad, int layout) {
    const int total = Ttot * CA * CB;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int cb = idx % CB;
        const int ca = (idx / CB) % CA;
        const int t = idx / (CB * CA);
        const size_t stride = (size_t)Ttot * CApad * CBpad;
        const float* p = partial + ((size_t)t * CApad + ca) * CBpad + cb;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += p[sp * stride];
        if (layout == 0) dw[((size_t)ca * CB + cb) * Ttot + t] = s;
        else dw[((size_t)cb * CA + ca) * Ttot + t] = s;
    }
}

// ------------------------------------------------------------------ host-side geometry
int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// pixel tile of `npix` (256 for tapconv) over a Hb x Wb base grid of N images
void pick_tile(int Hb, int Wb, int npix, int* lTH, int* lTW, int* lG) {
    int TW = Wb < 64 ? Wb : 64;
    if (TW > npix) TW = npix;
    int TH = npix / TW; if (TH > Hb) TH = Hb;
    int G = npix / (TH * TW);
    *lTH = ilog2(TH); *lTW = ilog2(TW); *lG = ilog2(G);
}

template <int MODE>
int launch_tapconv(const float* in, const float* wp, const float* bias, float* out, int N, int K, int M,
                   int Hb, int Wb, int Hi, int Wi, int Ho, int Wo, int par_a, hipStream_t s, const char* name) {
    using TC = TapCfg<MODE>;
    ConvGeom g;
    g.N = N; g.K = K; g.M = M;
    g.Kpad = gx_round_up(K, 8); g.Mpad = gx_round_up(M, 64);
    g.Hb = Hb; g.Wb = Wb; g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo; g.par_a = par_a;
    pick_tile(Hb, Wb, 256, &g.lTH, &g.lTW, &g.lG);
    const int TH = 1 << g.lTH, TW = 1 << g.lTW, G = 1 << g.lG;
    g.tiles_h = Hb / TH; g.tiles_w = Wb / TW;
    const int CHS = TC::PLANES * G * (TH + 2) * (TW + 2);
    if (CHS > MaxPos<MODE>::V * 256) { gx_set_error("%s: halo tile too large (%d)", name, CHS); return GX_EINVAL; }
    const size_t lds = (size_t)(TC::KC * CHS + TC::NT * TC::KC * 64) * sizeof(float);
    if (lds > 160 * 1024) { gx_set_error("%s: LDS %zu > 160KiB", name, lds); return GX_EINVAL; }
    static bool attr_set[4] = {false, false, false, false};
    if (!attr_set[MODE]) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&tapconv_kernel<MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[MODE] = true;
    }
    dim3 grid(g.tiles_h * g.tiles_w * gx_ceil_div(N, G), g.Mpad / 64);
    hipLaunchKernelGGL(tapconv_kernel<MODE>, grid, dim3(256), lds, s, in, wp, bias, out, g);
    GX_CHECK_LAUNCH(name);
    return GX_OK;
}

int launch_pack(const float* w, float* wp, int pack, int Co, int Ci, int NT, int Kpad, int Mpad, hipStream_t s) {
    const int total = NT * Kpad * Mpad;
    const int blocks = gx_ceil_div(total, 256) > 1024 ? 1024 : gx_ceil_div(total, 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, s, w, wp, pack, Co, Ci, NT, Kpad, Mpad);
    GX_CHECK_LAUNCH("pack_weights");
    return GX_OK;
}

struct WgradPlan {
    WgradGeom g;
    size_t lds_bytes;
    size_t ws_floats;
};

int plan_wgrad(int N, int CA, int CB, int Hb, int Wb, int SA, int Ttot, int ncls_launches, WgradPlan* pl) {
    WgradGeom& g = pl->g;
    g.N = N; g.CA = CA; g.CB = CB;
    g.CApad = gx_round_up(CA, 64); g.CBpad = gx_round_up(CB, 64);
    g.Hb = Hb; g.Wb = Wb; g.Ha = SA * Hb; g.Wa = SA * Wb;
    const int npix = (Wb <= 2) ? 64 : 128;
    pick_tile(Hb, Wb, npix, &g.lTH, &g.lTW, &g.lG);
    const int TH = 1 << g.lTH, TW = 1 << g.lTW, G = 1 << g.lG;
    g.tiles_h = Hb / TH; g.tiles_w = Wb / TW;
    g.ntiles = g.tiles_h * g.tiles_w * gx_ceil_div(N, G);
    const int chan_blocks = (g.CApad / 64) * (g.CBpad / 64) * ncls_launches;
    int nsplit = gx_ceil_div(768, chan_blocks);
    if (nsplit > g.ntiles) nsplit = g.ntiles;
    if (nsplit < 1) nsplit = 1;
    g.nsplit = nsplit;
    g.Ttot = Ttot;
    const int CHS = G * (TH + 2) * (TW + 2);
    pl->lds_bytes = (size_t)(64 * (npix + 1) + 64 * (CHS | 1)) * sizeof(float);
    pl->ws_floats = (size_t)nsplit * Ttot * g.CApad * g.CBpad;
    return GX_OK;
}

template <int WM>
int launch_wgrad(const float* a, const float* b, float* partial, const WgradPlan& pl, hipStream_t s, const char* name) {
    if (pl.lds_bytes > 160 * 1024) { gx_set_error("%s: LDS %zu > 160KiB", name, pl.lds_bytes); return GX_EINVAL; }
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<WM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(pl.g.nsplit, (pl.g.CApad / 64) * (pl.g.CBpad / 64));
    hipLaunchKernelGGL(wgrad_kernel<WM>, grid, dim3(256), pl.lds_bytes, s, a, b, partial, pl.g);
    GX_CHECK_LAUNCH(name);
    return GX_OK;
}

int launch_wgrad_reduce(const float* partial, float* dw, const WgradPlan& pl, int layout, hipStream_t s) {
    const int total = pl.g.Ttot * pl.g.CA * pl.g.CB;
    const int blocks = gx_ceil_div(total, 256) > 2048 ? 2048 : gx_ceil_div(total, 256);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, dw, pl.g.nsplit, pl.g.Ttot,
                       pl.g.CA, pl.g.CB, pl.g.CApad, pl.g.CBpad, layout);
    GX_CHECK_LAUNCH("wgrad_reduce");
    return GX_OK;
}

int check_dims(const char* name, int N, int Cin, int Cout, int H, int W) {
    GX_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0, "%s: bad N/C (%d,%d,%d)", name, N, Cin, Cout);
    GX_CHECK_ARG(gx_is_pow2(H) && gx_is_pow2(W) && H >= 2 && W >= 2 && H <= 1024 && W <= 1024,
                 "%s: H,W must be powers of two in [2,1024] (got %dx%d)", name, H, W);
    return GX_OK;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

size_t gx_conv3x3_ws_bytes(int N, int Cin, int Cout, int H, int W) {
    (void)N; (void)H; (void)W;
    // packed weights for fwd (k=Cin, m=Cout) or dgrad (k=Cout, m=Cin): take the max
    size_t f = (size_t)9 * gx_round_up(Cin, 8) * gx_round_up(Cout, 64);
    size_t d = (size_t)9 * gx_round_up(Cout, 8) * gx_round_up(Cin, 64);
    return (f > d ? f : d) * sizeof(float);
}

int gx_conv3x3_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout, int H, int W,
                   void* ws, size_t ws_bytes, gx_stream_t stream) {
    int rc = check_dims("gx_conv3x3_fwd", N, Cin, Cout, H, W);
    if (rc) return rc;
    GX_CHECK_ARG(x && w && y && ws, "gx_conv3x3_fwd: null pointer");
    const int Kpad = gx_round_up(Cin, 8), Mpad = gx_round_up(Cout, 64);
    GX_CHECK_ARG(ws_bytes >= (size_t)9 * Kpad * Mpad * sizeof(float), "gx_conv3x3_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    rc = launch_pack(w, (float*)ws, 0, Cout, Cin, 9, Kpad, Mpad, s);
    if (rc) return rc;
    return launch_tapconv<M_C3>(x, (const float*)ws, nullptr, y, N, Cin, Cout, H, W, H, W, H, W, 0, s,
                                "gx_conv3x3_fwd");
}

int gx_conv3x3_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W,
                     void* ws, size_t ws_bytes, gx_stream_t stream) {
    int rc = check_dims("gx_conv3x3_dgrad", N, Cin, Cout, H, W);
    if (rc) return rc;
    GX_CHECK_ARG(dy && w && dx && ws, "gx_conv3x3_dgrad: null pointer");
    const int Kpad = gx_round_up(Cout, 8), Mpad = gx_round_up(Cin, 64);
    GX_CHECK_ARG(ws_bytes >= (size_t)9 * Kpad * Mpad * sizeof(float), "gx_conv3x3_dgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    rc = launch_pack(w, (float*)ws, 1, Cout, Cin, 9, Kpad, Mpad, s);
    if (rc) return rc;
    return launch_tapconv<M_C3>(dy, (const float*)ws, nullptr, dx, N, Cout, Cin, H, W, H, W, H, W, 0, s,
                                "gx_conv3x3_dgrad");
}

size_t gx_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W) {
    WgradPlan pl;
    plan_wgrad(N, Cout, Cin, H, W, 1, 9, 1, &pl);
    return pl.ws_floats * sizeof(float);
}

int gx_conv3x3_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W,
                     void* ws, size_t ws_bytes, gx_stream_t stream) {
    int rc = check_dims("gx_conv3x3_wgrad", N, Cin, Cout, H, W);
    if (rc) return rc;
    GX_CHECK_ARG(x && dy && dw && ws, "gx_conv3x3_wgrad: null pointer");
    WgradPlan pl;
    plan_wgrad(N, Cout, Cin, H, W, 1, 9, 1, &pl);
    GX_CHECK_ARG(ws_bytes >= pl.ws_floats * sizeof(float), "gx_conv3x3_wgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    rc = launch_wgrad<W_C3>(dy, x, (float*)ws, pl, s, "gx_conv3x3_wgrad");
    if (rc) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, pl, 0, s);
}

size_t gx_deconv5x5s2_ws_bytes(int N, int Cin, int Cout, int Hin, int Win) {
    (void)N; (void)Hin; (void)Win;
    // fwd: two row-parity packs (15 + 10 taps, k=Cin, m=Cout); dgrad: 25 taps (k=Cout, m=Cin)
    size_t f = (size_t)25 * gx_round_up(Cin, 8) * gx_round_up(Cout, 64);
    size_t d = (size_t)25 * gx_round_up(Cout, 8) * gx_round_up(Cin, 64);
    return (f > d ? f : d) * sizeof(float);
}

int gx_deconv5x5s2_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout,
                       int Hin, int Win, void* ws, size_t ws_bytes, gx_stream_t stream) {
    int rc = check_dims("gx_deconv5x5s2_fwd", N, Cin, Cout, Hin, Win);
    if (rc) return rc;
    GX_CHECK_ARG(x && w && y && ws, "gx_deconv5x5s2_fwd: null pointer");
    const int Kpad = gx_round_up(Cin, 8), Mpad = gx_round_up(Cout, 64);
    GX_CHECK_ARG(ws_bytes >= (size_t)25 * Kpad * Mpad * sizeof(float), "gx_deconv5x5s2_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* wp0 = (float*)ws;
    float* wp1 = wp0 + (size_t)15 * Kpad * Mpad;
    rc = launch_pack(w, wp0, 2, Cout, Cin, 15, Kpad, Mpad, s);
    if (rc) return rc;
    rc = launch_pack(w, wp1, 3, Cout, Cin, 10, Kpad, Mpad, s);
    if (rc) return rc;
    rc = launch_tapconv<M_DT0>(x, wp0, bias, y, N, Cin, Cout, Hin, Win, Hin, Win, 2 * Hin, 2 * Win, 0, s,
                               "gx_deconv5x5s2_fwd(a=0)");
    if (rc) return rc;
    return launch_tapconv<M_DT1>(x, wp1, bias, y, N, Cin, Cout, Hin, Win, Hin, Win, 2 * Hin, 2 * Win, 1, s,
                                 "gx_deconv5x5s2_fwd(a=1)");
}

int gx_deconv5x5s2_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cin_out, int Cout,
                         int Hin, int Win, void* ws, size_t ws_bytes, gx_stream_t stream) {
    int rc = check_dims("gx_deconv5x5s2_dgrad", N, Cin, Cout, Hin, Win);
    if (rc) return rc;
    GX_CHECK_ARG(dy && w && dx && ws, "gx_deconv5x5s2_dgrad: null pointer");
    GX_CHECK_ARG(Cin_out > 0 && Cin_out <= Cin, "gx_deconv5x5s2_dgrad: Cin_out out of range");
    // only the first Cin_out input channels get a gradient (the decoder's coordinate channels need none);
    // weights are packed with the full Cin so that W's row stride is right, m is masked at Cin_out.
    const int Kpad = gx_round_up(Cout, 8), Mpad = gx_round_up(Cin, 64);
    GX_CHECK_ARG(ws_bytes >= (size_t)25 * Kpad * Mpad * sizeof(float), "gx_deconv5x5s2_dgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    rc = launch_pack(w, (float*)ws, 4, Cout, Cin, 25, Kpad, Mpad, s);
    if (rc) return rc;
    // launch with M = Cin_out but Mpad of the pack: build geometry by hand
    using TC = TapCfg<M_DG>;
    ConvGeom g;
    g.N = N; g.K = Cout; g.M = Cin_out; g.Kpad = Kpad; g.Mpad = Mpad;
    g.Hb = Hin; g.Wb = Win; g.Hi = 2 * Hin; g.Wi = 2 * Win; g.Ho = Hin; g.Wo = Win; g.par_a = 0;
    pick_tile(Hin, Win, 256, &g.lTH, &g.lTW, &g.lG);
    const int TH = 1 << g.lTH, TW = 1 << g.lTW, G = 1 << g.lG;
    g.tiles_h = Hin / TH; g.tiles_w = Win / TW;
    const int CHS = TC::PLANES * G * (TH + 2) * (TW + 2);
    GX_CHECK_ARG(CHS <= MaxPos<M_DG>::V * 256, "gx_deconv5x5s2_dgrad: halo tile too large");
    const size_t lds = (size_t)(TC::KC * CHS + TC::NT * TC::KC * 64) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&tapconv_kernel<M_DG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(g.tiles_h * g.tiles_w * gx_ceil_div(N, G), gx_ceil_div(Cin_out, 64));
    hipLaunchKernelGGL(tapconv_kernel<M_DG>, grid, dim3(256), lds, s, dy, (const float*)ws, (const float*)nullptr,
                       dx, g);
    GX_CHECK_LAUNCH("gx_deconv5x5s2_dgrad");
    return GX_OK;
}

size_t gx_deconv5x5s2_wgrad_ws_bytes(int N, int Cin, int Cout, int Hin, int Win) {
    WgradPlan pl;
    plan_wgrad(N, Cout, Cin, Hin, Win, 2, 25, 4, &pl);
    return pl.ws_floats * sizeof(float);
}

int gx_deconv5x5s2_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int Hin, int Win,
                         void* ws, size_t ws_bytes, gx_stream_t stream) {
    int rc = check_dims("gx_deconv5x5s2_wgrad", N, Cin, Cout, Hin, Win);
    if (rc) return rc;
    GX_CHECK_ARG(x && dy && dw && ws, "gx_deconv5x5s2_wgrad: null pointer");
    WgradPlan pl;
    plan_wgrad(N, Cout, Cin, Hin, Win, 2, 25, 4, &pl);
    GX_CHECK_ARG(ws_bytes >= pl.ws_floats * sizeof(float), "gx_deconv5x5s2_wgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)ws;
    rc = launch_wgrad<W_D00>(dy, x, part, pl, s, "gx_deconv5x5s2_wgrad(0,0)"); if (rc) return rc;
    rc = launch_wgrad<W_D01>(dy, x, part, pl, s, "gx_deconv5x5s2_wgrad(0,1)"); if (rc) return rc;
    rc = launch_wgrad<W_D10>(dy, x, part, pl, s, "gx_deconv5x5s2_wgrad(1,0)"); if (rc) return rc;
    rc = launch_wgrad<W_D11>(dy, x, part, pl, s, "gx_deconv5x5s2_wgrad(1,1)"); if (rc) return rc;
    return launch_wgrad_reduce(part, dw, pl, 1, s);
}

}  // extern "C"
