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
//   The chunk loop is software-pipelined: chunk c+1 is prefetched global->registers while chunk c's
//   MFMAs run out of a double-buffered LDS tile (one barrier per chunk).  Layers with too few
//   (pixel-tile x channel-tile) workgroups to fill 256 CUs split the channel reduction over
//   gridDim.z and a small deterministic reduce kernel sums the partial outputs.
// Weight-gradient kernels (wgrad) use the transposed mapping M = Cout, N = Cin, K = pixels
// with deterministic split-K partials + a reduce kernel.
//
// fp32 in / fp32 accumulate: results are an fmaf chain per output (exact fp32), which is
// what the stated fp32 parity tolerance of the path needs (no bf16/xf32 shortcuts).
#include "gx_common.h"

#include <cstdlib>
#include <mutex>
#include <vector>

#ifndef GX_WG_ABL
#define GX_WG_ABL 0   /* measurement builds: 1 = no B staging after the first tile, 2 = no A loads after the first, 4 = no barrier */
#endif

namespace {

enum { M_C3 = 0, M_DT0 = 1, M_DT1 = 2, M_DG = 3, M_C5 = 4 };

template <int MODE> struct TapCfg;
template <> struct TapCfg<M_C3> {
    static constexpr int NT = 9, NCLS = 1, KC = 8, PLANES = 1, HALO = 1;
    __host__ __device__ static constexpr int ro(int t) { return t / 3; }
    __host__ __device__ static constexpr int co(int t) { return t % 3; }
    __host__ __device__ static constexpr int cls(int) { return 0; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
};
// ConvTranspose k5 s2 p2 op1: out[2r+a][2c+b] += x[r+dr][c+dc] * W[kh][kw] with kh = a (mod 2),
// dr = (a+2-kh)/2, i.e. halo row offset ro = dr+1 = 2 - kh/2 (same for columns).
template <> struct TapCfg<M_DT0> {
    static constexpr int NT = 15, NCLS = 2, KC = 4, PLANES = 1, HALO = 1;
    __host__ __device__ static constexpr int ro(int t) { return 2 - t / 5; }        // kh = 2*(t/5)
    __host__ __device__ static constexpr int co(int t) { return 2 - (t % 5) / 2; }  // kw = t%5
    __host__ __device__ static constexpr int cls(int t) { return (t % 5) & 1; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
};
template <> struct TapCfg<M_DT1> {
    static constexpr int NT = 10, NCLS = 2, KC = 4, PLANES = 1, HALO = 1;
    __host__ __device__ static constexpr int ro(int t) { return 2 - t / 5; }        // kh = 2*(t/5)+1
    __host__ __device__ static constexpr int co(int t) { return 2 - (t % 5) / 2; }
    __host__ __device__ static constexpr int cls(int t) { return (t % 5) & 1; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
};
// dgrad of the deconv: dx[r][c] = sum dy[2r-2+kh][2c-2+kw] W[kh][kw]; plane = (kh&1, kw&1),
// in-plane offset (kh/2, kw/2).
template <> struct TapCfg<M_DG> {
    static constexpr int NT = 25, NCLS = 1, KC = 2, PLANES = 4, HALO = 1;
    __host__ __device__ static constexpr int ro(int t) { return (t / 5) / 2; }
    __host__ __device__ static constexpr int co(int t) { return (t % 5) / 2; }
    __host__ __device__ static constexpr int cls(int) { return 0; }
    __host__ __device__ static constexpr int plane(int t) { return ((t / 5) & 1) * 2 + ((t % 5) & 1); }
};

// 5x5 stride-1 pad-2 conv (the gated stacks of third_party/sylvester, VAE.py:18-33): 25 taps around a 2-pixel halo
template <> struct TapCfg<M_C5> {
    static constexpr int NT = 25, NCLS = 1, KC = 4, PLANES = 1, HALO = 2;
    __host__ __device__ static constexpr int ro(int t) { return t / 5; }
    __host__ __device__ static constexpr int co(int t) { return t % 5; }
    __host__ __device__ static constexpr int cls(int) { return 0; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
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
    int nsplit;       // split of the channel reduction over gridDim.z (partials written when > 1)
    int chunks_per_split;
    int act;          // epilogue activation after the bias: 0 none, 1 ReLU, 2 ELU (applied only when nsplit == 1)
    const float* zeros;   // zero page for the LDS-DMA staging path (NULL: stage through registers)
    float* stats;         // STATS kernels: per-workgroup (sum, sum of squares) of every 8-channel block of the output,
                          // [N][stats_parts][M/8][2] (GroupNorm statistics without a pass over the output)
    int stats_parts;      // workgroups per image = tiles_h * tiles_w * row parities
};

__device__ __forceinline__ float gx_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : expm1f(v);
    return v;
}

// NPOS = halo-tile positions staged per thread per channel (tile <= NPOS*256 floats per channel).
// MW = waves along M: 1 -> the 4 waves tile 256 pixels (64 each) and all 64 channels; 2 -> a 128-pixel tile, waves 2 x 2
// (32 channels x 64 pixels each): twice the workgroups for grids that cannot fill the chip with 256-pixel tiles.
template <int MODE, int NPOS, bool DMA, int MW = 1, bool STATS = false, bool HH = true>
__device__ __forceinline__ void tapconv_body(const float* __restrict__ in, const float* __restrict__ wp,
                                             const float* __restrict__ bias, float* __restrict__ out,
                                             const ConvGeom& g, float* lds, const int bx, const int by, const int bz,
                                             const int par_a) {
    using TC = TapCfg<MODE>;
    constexpr int NT = TC::NT, NCLS = TC::NCLS, KC = TC::KC, PLANES = TC::PLANES;
    constexpr int NW4 = (NT * KC * 16 + 255) / 256;   // float4 weight loads per thread per chunk

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int TH = 1 << g.lTH, TW = 1 << g.lTW, G = 1 << g.lG;
    constexpr int HL = TC::HALO;
    const int HS = TW + 2 * HL;                    // halo row stride
    const int PLS = G * (TH + 2 * HL) * HS;        // one plane of the halo tile
    const int CHS = PLANES * PLS;                  // per-channel LDS stride
    const int BUF = KC * CHS + NT * KC * 64;       // floats per pipeline stage: [KC][CHS] input + [NT][KC][64] weights

    // ---- which tile ----
    int tile = bx;
    const int tw_i = tile % g.tiles_w; tile /= g.tiles_w;
    const int th_i = tile % g.tiles_h; tile /= g.tiles_h;
    const int img0 = tile * G;
    const int R0 = th_i * TH, C0 = tw_i * TW;
    const int m0 = by * 64;

    const size_t in_img_stride = (size_t)g.K * g.Hi * g.Wi;
    const float* in_blk = in + (size_t)img0 * in_img_stride;
    const int HiWi = g.Hi * g.Wi;

    // ---- per-thread staging positions (computed once; only the channel term changes) ----
    int goff[NPOS];
#pragma unroll
    for (int q = 0; q < NPOS; ++q) {
        const int pos = tid + q * 256;
        int off = -1;
        if (pos < CHS) {
            int rem = pos;
            const int plane = rem / PLS; rem -= plane * PLS;
            const int gi = rem / ((TH + 2 * HL) * HS); rem -= gi * (TH + 2 * HL) * HS;
            const int i = rem / HS;
            const int j = rem - i * HS;
            int row, col;
            if (MODE == M_DG) {
                row = 2 * (R0 + i) - 2 + (plane >> 1);
                col = 2 * (C0 + j) - 2 + (plane & 1);
            } else {
                row = R0 - HL + i;
                col = C0 - HL + j;
            }
            if (img0 + gi < g.N && row >= 0 && row < g.Hi && col >= 0 && col < g.Wi)
                off = gi * (int)in_img_stride + row * g.Wi + col;
        }
        goff[q] = off;
    }
    // register-prefetch path: raw buffer loads over this tile's images (wave-uniform base, per-lane byte offset, the
    // channel in the scalar offset); a halo position outside the image has an out-of-range offset and reads as 0 --
    // no compare / select / 64-bit address per element (these small layers ran 5 VALU + 5.7 SALU per MFMA)
    int voff[NPOS];
#pragma unroll
    for (int q = 0; q < NPOS; ++q) voff[q] = goff[q] >= 0 ? goff[q] * 4 : (int)0x80000000;
    const int imgs_here = g.N - img0 < G ? g.N - img0 : G;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in_blk), 0, (int)((size_t)imgs_here * in_img_stride * 4), 0x00020000);

    // ---- per-lane fragment offsets ----
    // A (weights): lane -> w_tile[t][2kk + (lane>>5)][mi*32 + (lane&31)]
    constexpr int MI = 2 / MW;                     // 32-channel MFMA tiles per wave
    const int wm = MW == 2 ? (wave >> 1) : 0;      // channel half of this wave (MW == 2)
    const int wn = MW == 2 ? (wave & 1) : wave;    // 64-pixel group of this wave
    const int a_off = KC * CHS + (lane >> 5) * 64 + (lane & 31) + wm * 32;
    // output channels 32..63 of this workgroup's tile exist?  (32-channel layers -- MONet's UNet ends and its
    // BroadcastDecoder -- skip the upper MFMA tile instead of computing padding.)  Compile-time: a runtime test in
    // the tap loop cuts it into one basic block per tap, and the scheduler then leaves every LDS read directly in
    // front of the two MFMAs that consume it (round-1 ISA: s_waitcnt lgkmcnt(0) every 2 MFMAs, pipe 0.56 busy).
    constexpr bool hi_half = HH;
    // B (input):   lane -> in_tile[2kk + (lane>>5)][plane][halo(pixel) + tap]
    int b_off[2];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int p = wn * 64 + nj * 32 + (lane & 31);
        const int c = p & (TW - 1);
        const int r = (p >> g.lTW) & (TH - 1);
        const int gi = p >> (g.lTW + g.lTH);
        b_off[nj] = (lane >> 5) * CHS + (gi * (TH + 2 * HL) + r) * HS + c;
    }

    f32x16 acc[NCLS][MI][2];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][i][j][e] = 0.f;

    // ---- software pipeline over channel chunks ----
    float xin[KC][NPOS];
    f32x4 wreg[NW4];
    const int nchunks = g.Kpad / KC;
    const int c_begin = bz * g.chunks_per_split;
    int c_end = c_begin + g.chunks_per_split;
    if (c_end > nchunks) c_end = nchunks;

// (macros, not lambdas: capturing the register arrays by reference sends them to scratch)
#define GX_TAP_PREFETCH(chunk)                                                                              \
    {                                                                                                       \
        const int ch0_ = (chunk) * KC;                                                                      \
        _Pragma("unroll") for (int ch = 0; ch < KC; ++ch) {                                                 \
            const bool chv = (ch0_ + ch) < g.K;       /* wave-uniform: a scalar branch, not a select per load */ \
            const int soff_ = (ch0_ + ch) * HiWi * 4;                                                       \
            _Pragma("unroll") for (int q = 0; q < NPOS; ++q)                                                \
                xin[ch][q] = chv ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, voff[q], soff_, 0)) : 0.f; \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NW4; ++i) {                                                   \
            const int i4 = tid + i * 256;                                                                   \
            if (i4 < NT * KC * 16) {                                                                        \
                const int q4 = i4 & 15;                                                                     \
                const int kc = (i4 >> 4) & (KC - 1);                                                        \
                const int t_ = i4 / (16 * KC);                                                              \
                wreg[i] = *reinterpret_cast<const f32x4*>(wp + ((size_t)t_ * g.Kpad + ch0_ + kc) * g.Mpad +  \
                                                          m0 + q4 * 4);                                     \
            }                                                                                               \
        }                                                                                                   \
    }
#define GX_TAP_COMMIT(buf)                                                                                  \
    {                                                                                                       \
        _Pragma("unroll") for (int ch = 0; ch < KC; ++ch)                                                   \
            _Pragma("unroll") for (int q = 0; q < NPOS; ++q) {                                              \
                const int pos = tid + q * 256;                                                              \
                if (pos < CHS) (buf)[ch * CHS + pos] = xin[ch][q];                                          \
            }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < NW4; ++i) {                                                   \
            const int i4 = tid + i * 256;                                                                   \
            if (i4 < NT * KC * 16) *reinterpret_cast<f32x4*>((buf) + KC * CHS + i4 * 4) = wreg[i];          \
        }                                                                                                   \
    }

    // LDS-DMA staging (global_load_lds): halo positions outside the image / channels beyond K come from a zero page,
    // so a chunk costs no staging VGPRs, no selects and no ds_write; destination = wave-uniform base + lane * size.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define GX_TAP_ISSUE(chunk, buf)                                                                            \
    {                                                                                                       \
        const int ch0_ = (chunk) * KC;                                                                      \
        _Pragma("unroll") for (int ch = 0; ch < KC; ++ch) {                                                 \
            const bool chv = (ch0_ + ch) < g.K;                                                             \
            const float* src = in_blk + (size_t)(ch0_ + ch) * HiWi;                                         \
            _Pragma("unroll") for (int q = 0; q < NPOS; ++q) {                                              \
                if (tid + q * 256 < CHS) {                                                                  \
                    const float* gp = (chv && goff[q] >= 0) ? src + goff[q] : g.zeros;                      \
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,     \
                        (__attribute__((address_space(3))) void*)((buf) + ch * CHS + q * 256 + wave_u * 64), 4, 0, 0); \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NW4; ++i) {                                                   \
            const int i4 = tid + i * 256;                                                                   \
            if (i4 < NT * KC * 16) {                                                                        \
                const int q4 = i4 & 15;                                                                     \
                const int kc = (i4 >> 4) & (KC - 1);                                                        \
                const int t_ = i4 / (16 * KC);                                                              \
                const float* gp = wp + ((size_t)t_ * g.Kpad + ch0_ + kc) * g.Mpad + m0 + q4 * 4;            \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,         \
                    (__attribute__((address_space(3))) void*)((buf) + KC * CHS + (i * 256 + wave_u * 64) * 4), 16, 0, 0); \
            }                                                                                               \
        }                                                                                                   \
    }

    if (DMA) {
        if (c_begin < c_end) GX_TAP_ISSUE(c_begin, lds)
    } else {
        if (c_begin < c_end) GX_TAP_PREFETCH(c_begin)
    }
    for (int c = c_begin; c < c_end; ++c) {
        float* buf = lds + ((c - c_begin) & 1) * BUF;
        if (DMA) {
            __syncthreads();      // this chunk has landed (vmcnt drained before the barrier); the other buffer is free
            if (c + 1 < c_end) GX_TAP_ISSUE(c + 1, lds + ((c + 1 - c_begin) & 1) * BUF)
        } else {
            GX_TAP_COMMIT(buf)
            __syncthreads();
            if (c + 1 < c_end) GX_TAP_PREFETCH(c + 1)
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int toff = TC::plane(t) * PLS + TC::ro(t) * HS + TC::co(t);
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                const float a0 = buf[(t * KC + 2 * kk) * 64 + a_off];
                const float b0 = buf[2 * kk * CHS + b_off[0] + toff];
                const float b1 = buf[2 * kk * CHS + b_off[1] + toff];
                const int cl = TC::cls(t);
                acc[cl][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[cl][0][0], 0, 0, 0);
                acc[cl][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[cl][0][1], 0, 0, 0);
                if (MI == 2 && hi_half) {
                    const float a1 = buf[(t * KC + 2 * kk) * 64 + a_off + 32];
                    acc[cl][MI - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[cl][MI - 1][0], 0, 0, 0);
                    acc[cl][MI - 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[cl][MI - 1][1], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: C/D layout col = lane&31 (pixel), row = (reg&3)+8*(reg>>2)+4*(lane>>5) (channel) ----
    const size_t out_img_stride = (size_t)g.M * g.Ho * g.Wo;
    const int HoWo = g.Ho * g.Wo;
    float* outz = out + (size_t)bz * g.N * out_img_stride;   // partial slab when nsplit > 1
    const bool add_bias = bias != nullptr && g.nsplit == 1;
    const int act = g.nsplit == 1 ? g.act : 0;
    static_assert(!STATS || (MW == 1 && NCLS == 2), "output statistics: full 64-channel workgroups of the deconv only");
    float st_s[STATS ? MI * 4 : 1], st_q[STATS ? MI * 4 : 1];   // per 8-channel block (mi, reg >> 2): sum, sum of squares
    if (STATS) {
#pragma unroll
        for (int i = 0; i < MI * 4; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
    }
    // the lane's 4-channel runs of the bias, loaded once (the channel index does not depend on the pixel block)
    f32x4 bvec[MI][4];
    {
        const bool vec_ok = add_bias && (reinterpret_cast<uintptr_t>(bias) & 15) == 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mb = m0 + (mi + wm) * 32 + 8 * q + 4 * (lane >> 5);
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (vec_ok && mb + 3 < g.M) t = *reinterpret_cast<const f32x4*>(bias + mb);
                else if (add_bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = mb + e < g.M ? bias[mb + e] : 0.f;
                }
                bvec[mi][q] = t;
            }
    }
    // interior tiles whose accumulator rows are all real channels store without per-element guards (gx_kq.hip: the
    // guarded loop is ~1400 VALU + 48 branches per wave, 8 % of the large transposed-conv kernels' time)
    const bool full_tile = img0 + G <= g.N && R0 + TH <= g.Hb && C0 + TW <= g.Wb && m0 + (MI + wm) * 32 <= g.M;
    if (full_tile) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int p = wn * 64 + nj * 32 + (lane & 31);
            const int c = p & (TW - 1);
            const int r = (p >> g.lTW) & (TH - 1);
            const int n = img0 + (p >> (g.lTW + g.lTH));
            int orow, ocol;
            if (NCLS == 2) { orow = 2 * (R0 + r) + par_a; ocol = 2 * (C0 + c); }
            else { orow = R0 + r; ocol = C0 + c; }
            float* obase = outz + (size_t)n * out_img_stride + (size_t)orow * g.Wo + ocol +
                           (size_t)(m0 + wm * 32 + 4 * (lane >> 5)) * HoWo;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float* o = obase + (size_t)(mi * 32 + (reg & 3) + 8 * (reg >> 2)) * HoWo;
                    const float bv = bvec[mi][reg >> 2][reg & 3];
                    if (NCLS == 2) {
                        float2 v;
                        v.x = gx_act(acc[0][mi][nj][reg] + bv, act);
                        v.y = gx_act(acc[NCLS - 1][mi][nj][reg] + bv, act);
                        *reinterpret_cast<float2*>(o) = v;
                    } else {
                        *o = gx_act(acc[0][mi][nj][reg] + bv, act);
                    }
                }
            }
        }
    } else {
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int p = wn * 64 + nj * 32 + (lane & 31);
        const int c = p & (TW - 1);
        const int r = (p >> g.lTW) & (TH - 1);
        const int gi = p >> (g.lTW + g.lTH);
        const int n = img0 + gi;
        if (n >= g.N || R0 + r >= g.Hb || C0 + c >= g.Wb) continue;   // partial tiles of non-power-of-two grids
        int orow, ocol;
        if (NCLS == 2) { orow = 2 * (R0 + r) + par_a; ocol = 2 * (C0 + c); }
        else { orow = R0 + r; ocol = C0 + c; }
        float* obase = outz + (size_t)n * out_img_stride + (size_t)orow * g.Wo + ocol;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + (mi + wm) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m < g.M) {
                    const float bv = bvec[mi][reg >> 2][reg & 3];
                    if (NCLS == 2) {
                        float2 v;
                        v.x = gx_act(acc[0][mi][nj][reg] + bv, act);
                        v.y = gx_act(acc[NCLS - 1][mi][nj][reg] + bv, act);
                        *reinterpret_cast<float2*>(obase + (size_t)m * HoWo) = v;

                    } else {
                        obase[(size_t)m * HoWo] = gx_act(acc[0][mi][nj][reg] + bv, act);
                    }
                }
            }
        }
    }
    }
    if constexpr (STATS) {
        // (a second walk over the accumulators, after the stores: the store loop stays the plain kernel's)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int p = wn * 64 + nj * 32 + (lane & 31);
            const int c = p & (TW - 1);
            const int r = (p >> g.lTW) & (TH - 1);
            const bool ok = img0 + (p >> (g.lTW + g.lTH)) < g.N && R0 + r < g.Hb && C0 + c < g.Wb;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m = m0 + (mi + wm) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    const float bv = bvec[mi][reg >> 2][reg & 3];
                    const float vx = (ok && m < g.M) ? acc[0][mi][nj][reg] + bv : 0.f;
                    const float vy = (ok && m < g.M) ? acc[NCLS - 1][mi][nj][reg] + bv : 0.f;
                    st_s[mi * 4 + (reg >> 2)] += vx + vy;
                    st_q[mi * 4 + (reg >> 2)] += vx * vx + vy * vy;
                }
        }
        // all 256 threads hold the same 8 channel blocks (x their own pixels): transpose through LDS (the tiles are
        // dead by now) so that 16 threads own each of the 16 values, then a 16-lane sum; fixed order throughout
        static_assert(!STATS || MI * 4 == 8, "8 channel blocks per workgroup");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) { lds[i * 256 + tid] = st_s[i]; lds[(8 + i) * 256 + tid] = st_q[i]; }
        __syncthreads();
        const int vi = tid >> 4, sub = tid & 15;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(lds + vi * 256 + sub * 16 + 4 * k);
            v += (r[0] + r[1]) + (r[2] + r[3]);
        }
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        if (sub == 0) {
            const int part = (th_i * g.tiles_w + tw_i) * 2 + par_a;
            const int nblk = g.M >> 3;
            const int blk = (m0 >> 3) + (vi & 7);
            if (blk < nblk && img0 < g.N)
                g.stats[(((size_t)img0 * g.stats_parts + part) * nblk + blk) * 2 + (vi >> 3)] = v;
        }
    }
}

template <int MODE, int NPOS, bool DMA, int MW = 1, bool HH = true>
__global__ void __launch_bounds__(256, 2)
tapconv_kernel(const float* __restrict__ in, const float* __restrict__ wp,
               const float* __restrict__ bias, float* __restrict__ out, ConvGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    tapconv_body<MODE, NPOS, DMA, MW, false, HH>(in, wp, bias, out, g, lds, gx_xcd_tile(blockIdx.x, gridDim.x), blockIdx.y, blockIdx.z, g.par_a);
}

// Both output-row parities of the transposed conv in one launch: blockIdx.y = 2 * channel_tile + parity.  Twice
// the workgroups of a single-parity launch, so mid-sized layers fill the chip without splitting the channel
// reduction (and without the partial-sum traffic and reduce pass that come with it).
template <int NPOS, bool DMA, int MW = 1, bool STATS = false>
__global__ void __launch_bounds__(256, 2)
tapconv_dt_kernel(const float* __restrict__ in, const float* __restrict__ wp0, const float* __restrict__ wp1,
                  const float* __restrict__ bias, float* __restrict__ out, ConvGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (blockIdx.y & 1)
        tapconv_body<M_DT1, NPOS, DMA, MW, STATS>(in, wp1, bias, out, g, lds, gx_xcd_tile(blockIdx.x, gridDim.x), blockIdx.y >> 1, blockIdx.z, 1);
    else
        tapconv_body<M_DT0, NPOS, DMA, MW, STATS>(in, wp0, bias, out, g, lds, gx_xcd_tile(blockIdx.x, gridDim.x), blockIdx.y >> 1, blockIdx.z, 0);
}

// GroupNorm statistics from the per-workgroup block sums the STATS epilogue wrote: stats [N][parts][nblk][2],
// group g of image n = blocks [g*bpg, (g+1)*bpg); fp64 from here on, fixed order.
__global__ void __launch_bounds__(64)
gn_stats_finalize_kernel(const float* __restrict__ stats, int N, int groups, int parts, int nblk, int bpg, double m,
                         float eps, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * groups) return;
    const int n = i / groups, g = i - n * groups;
    double s = 0.0, q = 0.0;
    for (int p = 0; p < parts; ++p)
        for (int b = 0; b < bpg; ++b) {
            const float* e = stats + (((size_t)n * parts + p) * nblk + g * bpg + b) * 2;
            s += (double)e[0];
            q += (double)e[1];
        }
    const double mean = s / m;
    double var = q / m - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_out[i] = (float)mean;
    rstd_out[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// out[i] = sum_z part[z][i] (+ bias[channel]); fixed summation order.
__global__ void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                     float* __restrict__ out, size_t total, int nsplit, int M, int HoWo, int act) {
    const size_t n4 = total >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = reinterpret_cast<const float4*>(part)[i];
        for (int z = 1; z < nsplit; ++z) {
            const float4 v = reinterpret_cast<const float4*>(part + (size_t)z * total)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (bias) {
            const float b = bias[((i * 4) / HoWo) % M];   // HoWo is a multiple of 4
            s.x += b; s.y += b; s.z += b; s.w += b;
        }
        s.x = gx_act(s.x, act); s.y = gx_act(s.y, act); s.z = gx_act(s.z, act); s.w = gx_act(s.w, act);
        reinterpret_cast<float4*>(out)[i] = s;
    }
}

// ------------------------------------------------------------------ weight packing
// Wp[t][k][m] (k padded to Kpad, m padded to Mpad, zero filled).
//   pack 0: conv3x3 fwd    W[co][ci][3][3]  -> m=co, k=ci, t=kh*3+kw
//   pack 1: conv3x3 dgrad  W[co][ci][3][3]  -> m=ci, k=co, t=(kh,kw) reads W[..][2-kh][2-kw]
//   pack 2/3: deconv fwd rows a=0/1, W[ci][co][5][5] -> m=co, k=ci, t=khi*5+kw, kh=2*khi+a
//   pack 4: deconv dgrad   W[ci][co][5][5]  -> m=ci, k=co, t=kh*5+kw
__device__ __forceinline__ float pack_weight_value(const float* __restrict__ w, int pack, int Co, int Ci, int m,
                                                   int k, int t) {
    float v = 0.f;
    if (pack >= 10) pack -= 10;
    if (pack == 0) {
        if (m < Co && k < Ci) v = w[((size_t)m * Ci + k) * 9 + t];
    } else if (pack == 1) {
        if (m < Ci && k < Co) v = w[((size_t)k * Ci + m) * 9 + (8 - t)];
    } else if (pack == 2 || pack == 3) {
        const int kh = 2 * (t / 5) + (pack - 2), kw = t % 5;
        if (m < Co && k < Ci) v = w[((size_t)k * Co + m) * 25 + kh * 5 + kw];
    } else if (pack == 4) {
        if (m < Ci && k < Co) v = w[((size_t)m * Co + k) * 25 + t];
    } else if (pack == 7) {   // 5x5 stride 1, cross-correlation: w [Co = M][Ci = K][5][5]
        if (m < Co && k < Ci) v = w[((size_t)m * Ci + k) * 25 + t];
    } else if (pack == 8) {   // 5x5 stride 1, true convolution with the channel roles swapped: w [Co = K][Ci = M][5][5]
        if (k < Co && m < Ci) v = w[((size_t)k * Ci + m) * 25 + (24 - t)];
    } else {   // 5 / 6: Winograd operands of the conv3x3 forward / data gradient (t = position)
        v = gx_wino_u_value(w, pack - 5, Co, Ci, m, k, t);
    }
    return v;
}
// destination of element (t, k, m): [t][k][m] for the tap-conv kernels, the operand order of gx_wino.hip for packs 5 / 6,
// the k-quad order of gx_kq.hip for packs 10..14 (= packs 0..4 in that layout)
__device__ __forceinline__ size_t pack_dest(int pack, int idx, int m, int k, int t, int Kpad, int NT) {
    if (pack >= 10) return gx_kq_w_slot(m, k, pack == 14 ? gx_kq_dg_tap_slot(t) : t, NT, Kpad);
    return (pack == 5 || pack == 6) ? gx_wino_u_slot(m, k, t, Kpad) : (size_t)idx;
}

// packs 25 / 26 (= 5 / 6: the Winograd operands for gx_wino.hip's bf16-pipe kernel), 20 / 21 (= 0 / 1, gx_kq.hip's Q_C3H) and 22 / 23 / 24 (= 2 / 3 / 4 for the bf16 matrix pipe, gx_kq.hip's QCfgDTH / Q_DGH): every weight as three bf16 pieces, two
// channels per 32-bit word -- the thread of an even k writes the three words of (k, k + 1), the odd one nothing
// packs 42 / 43 / 44: 22 / 23 / 24 as TWO fp16 pieces of w * 2^f16_exp (gx_kq.hip's fp16 x 3 form; f16_exp from the tensor's amax)
__device__ __forceinline__ void pack_h_store(const float* __restrict__ w, float* __restrict__ wp, int pack, int Co, int Ci,
                                             int m, int k, int t, int NT, int Kpad, int f16_exp = 0) {
    if (k & 1) return;
    const bool f16 = pack >= 40;
    if (f16) pack -= 20;
    unsigned wd[3];
    float v[2] = {pack_weight_value(w, pack - 20, Co, Ci, m, k, t), pack_weight_value(w, pack - 20, Co, Ci, m, k + 1, t)};
    unsigned short pc[2][3];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (f16) {
            const float ws = ldexpf(v[e], f16_exp);
            const _Float16 h = (_Float16)ws;
            pc[e][0] = __builtin_bit_cast(unsigned short, h);
            pc[e][1] = __builtin_bit_cast(unsigned short, (_Float16)(ws - (float)h));
            pc[e][2] = 0;
            continue;
        }
        const __bf16 h = (__bf16)v[e];
        const float r1 = v[e] - (float)h;
        const __bf16 mm = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)mm);
        pc[e][0] = __builtin_bit_cast(unsigned short, h);
        pc[e][1] = __builtin_bit_cast(unsigned short, mm);
        pc[e][2] = __builtin_bit_cast(unsigned short, l);
    }
    const int np = f16 ? 2 : 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (q >= np) break;
        wd[q] = (unsigned)pc[0][q] | ((unsigned)pc[1][q] << 16);
        const size_t dst = pack <= 21 ? gx_kq_h32_word(m, k, t, q, NT, Kpad, np)       // 20 / 21: conv3x3, 32-channel tiles
                           : ((pack == 25 || pack == 26) ? gx_wino_h_word(m, k, t, q, Kpad)        // 25 / 26: Winograd operands (= 5 / 6), t = position; 27 / 28: 5 x 5 stride 1 (= 7 / 8)
                                         : gx_kq_h_word(m, k, pack == 24 ? gx_kq_dg_tap_slot(t) : t, q, NT, Kpad, np));
        wp[dst] = __builtin_bit_cast(float, wd[q]);
    }
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int pack,
                                    int Co, int Ci, int NT, int Kpad, int Mpad) {
    const int total = NT * Kpad * Mpad;
    const int f16_exp = pack >= 40 ? gx_f16_scale_exp(((pack == 45 || pack == 46) ? 2.25f : 1.f) * *reinterpret_cast<const float*>(reinterpret_cast<const char*>(wp) + gx_kq_h_amax_off(Kpad, Mpad, NT))) : 0;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int m = idx % Mpad;
        const int k = (idx / Mpad) % Kpad;
        const int t = idx / (Mpad * Kpad);
        if (pack >= 20) { pack_h_store(w, wp, pack, Co, Ci, m, k, t, NT, Kpad, f16_exp); continue; }
        wp[pack_dest(pack, idx, m, k, t, Kpad, NT)] = pack_weight_value(w, pack, Co, Ci, m, k, t);
    }
}

// ------------------------------------------------------------------ weight gradients
// D[i = A-channel][j = B-channel] per tap, K = pixels.
//   A source: dy sampled at (SA*row + pa, SA*col + pb)  (SA=1 conv3x3; SA=2 deconv class (pa,pb))
//   B source: x with a 1-pixel halo on the base grid.
enum { W_C3 = 0, W_D00 = 1, W_D01 = 2, W_D10 = 3, W_D11 = 4, W_C5A = 5, W_C5B = 6, W_C5C = 7 };
template <int WM> struct WTap;
template <> struct WTap<W_C3> {
    static constexpr int NT = 9, SA = 1, PA = 0, PB = 0, HALO = 1;
    __host__ __device__ static constexpr int ro(int t) { return t / 3; }
    __host__ __device__ static constexpr int co(int t) { return t % 3; }
    __host__ __device__ static constexpr int gt(int t) { return t; }
};
template <int PA_, int PB_> struct WTapD {
    static constexpr int NKH = PA_ ? 2 : 3, NKW = PB_ ? 2 : 3;
    static constexpr int NT = NKH * NKW, SA = 2, PA = PA_, PB = PB_, HALO = 1;
    __host__ __device__ static constexpr int kh(int t) { return 2 * (t / NKW) + PA_; }
    __host__ __device__ static constexpr int kw(int t) { return 2 * (t % NKW) + PB_; }
    __host__ __device__ static constexpr int ro(int t) { return 2 - kh(t) / 2; }
    __host__ __device__ static constexpr int co(int t) { return 2 - kw(t) / 2; }
    __host__ __device__ static constexpr int gt(int t) { return kh(t) * 5 + kw(t); }
};
// 5 x 5 stride-1 pad-2 conv (the gated stacks' layers whose rows are shorter than the row-ring tiles' 32 pixels): kernel rows
// 0-1, 2-3 (10 taps each) and 4 (5 taps) as three launches into one slab region -- 15 accumulator tiles (240 registers) next
// to the kernel's per-lane offset tables spill 1 KB per lane; 2-pixel halo
template <int PART> struct WTapC5 {
    static constexpr int NT = PART == 2 ? 5 : 10, SA = 1, PA = 0, PB = 0, HALO = 2;
    __host__ __device__ static constexpr int ro(int t) { return t / 5 + 2 * PART; }
    __host__ __device__ static constexpr int co(int t) { return t % 5; }
    __host__ __device__ static constexpr int gt(int t) { return ro(t) * 5 + co(t); }
};
template <> struct WTap<W_C5A> : WTapC5<0> {};
template <> struct WTap<W_C5B> : WTapC5<1> {};
template <> struct WTap<W_C5C> : WTapC5<2> {};
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
    int nsplit;          // partial slabs (max over classes for the merged deconv launch)
    int Ttot;            // taps in the partial buffer (9 or 25)
    int cls_begin[5];    // merged deconv launch: blockIdx.x range of parity class c is [cls_begin[c], cls_begin[c+1])
    float* bias_part;    // QUAD on the bf16 pipe: [nsplit][256] per-thread sums of the dy values it multiplied (NULL: off) --
                         // the layer's bias gradient rides on the weight gradient's read of dy (gx_conv3x3_wgrad_quad_bias)
};

// A operand (dy) goes global -> registers directly: the k (pixel) slots of the MFMA are assigned so that
// lane-half k owns a contiguous half of the pixel tile, i.e. every lane streams one channel row with
// 16-byte loads (the pairing of pixels to k slots is free as long as A and B agree).  Only the B operand
// (x with its 1-pixel halo, re-used by all taps) is staged in LDS: double-buffered, prefetched
// global -> registers one tile ahead while the current tile's MFMAs run; one barrier per tile.
// sp / nsp: this workgroup's split index and the number of splits of its class (tiles sp, sp+nsp, ...)
template <int WM>
__device__ __forceinline__ void wgrad_body(const float* __restrict__ a_src, const float* __restrict__ b_src,
                                           float* __restrict__ partial, const WgradGeom& g, float* lds,
                                           const int sp, const int nsp) {
    using WT = WTap<WM>;
    constexpr int NT = WT::NT;
    constexpr int NPOS = 1;              // halo tile <= 256 floats per channel (plan_wgrad picks tiles so)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TH = 1 << g.lTH, TW = 1 << g.lTW, G = 1 << g.lG;
    const int PT = TH * TW * G;          // pixels per tile (64 or 128)
    const int HS = TW + 2;
    const int CHS = G * (TH + 2) * HS;
    const int BS = CHS | 1;              // odd B channel stride: lanes = channels read conflict-free
    const int BUF = 64 * BS;

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

    const int ca_l = ca0 + wm * 32 + (lane & 31);     // this lane's A channel
    const bool ca_ok = ca_l < g.CA;
    const int khalf = lane >> 5;                       // k slot of this lane
    const int b_row = (wn * 32 + (lane & 31)) * BS;

    float xin[64][NPOS];
    // 16-byte A loads need 4 consecutive pixels inside one row: tiles >= 4 wide on a grid whose width is a
    // multiple of 4 (all power-of-two layers, the 72-wide broadcast canvas); scalar loads otherwise
    const bool vecA = TW >= 4 && (g.Wb & 3) == 0;

#define GX_WG_TILE_ORIGIN(tile_, img0_, R0_, C0_)                      \
    int img0_, R0_, C0_;                                               \
    {                                                                  \
        int tt_ = (tile_);                                             \
        const int tw_i_ = tt_ % g.tiles_w; tt_ /= g.tiles_w;           \
        const int th_i_ = tt_ % g.tiles_h; tt_ /= g.tiles_h;           \
        img0_ = tt_ * G; R0_ = th_i_ * TH; C0_ = tw_i_ * TW;           \
    }
#define GX_WG_PREFETCH(tile_)                                                                        \
    {                                                                                                \
        GX_WG_TILE_ORIGIN(tile_, pi0, pR0, pC0)                                                      \
        _Pragma("unroll") for (int q = 0; q < NPOS; ++q) {                                           \
            const int pos = tid + q * 256;                                                           \
            int off = -1;                                                                            \
            if (pos < CHS) {                                                                         \
                int rem = pos;                                                                       \
                const int gi = rem / ((TH + 2) * HS); rem -= gi * (TH + 2) * HS;                     \
                const int i = rem / HS;                                                              \
                const int jj = rem - i * HS;                                                         \
                const int row = pR0 - 1 + i, col = pC0 - 1 + jj;                                     \
                if (pi0 + gi < g.N && row >= 0 && row < g.Hb && col >= 0 && col < g.Wb)              \
                    off = gi * (int)b_img + row * g.Wb + col;                                        \
            }                                                                                        \
            const float* src = b_src + (size_t)pi0 * b_img + (size_t)cb0 * HbWb;                     \
            _Pragma("unroll") for (int ch = 0; ch < 64; ++ch) {                                      \
                float v = 0.f;                                                                       \
                if (off >= 0 && cb0 + ch < g.CB) v = src[(size_t)ch * HbWb + off];                   \
                xin[ch][q] = v;                                                                      \
            }                                                                                        \
        }                                                                                            \
    }

    // one batch of A values: 4 groups x 4 consecutive k-steps; halo = LDS offset of each group's first pixel
    // (hx: per-element extra offsets, only used by the scalar path for TW < 4)
    struct ABatch { float v[4][4]; int halo[4]; int hx[4][4]; };
    ABatch acur, anxt;
#define GX_WG_LOAD_A(tile_, bt_, dst_)                                                                        \
    {                                                                                                          \
        GX_WG_TILE_ORIGIN(tile_, ai0, aR0, aC0)                                                                \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                     \
            const int j0 = khalf * (PT >> 1) + 4 * (4 * (bt_) + gq);                                           \
            if (vecA) {                                                                                        \
                const int c = j0 & (TW - 1);                                                                   \
                const int r = (j0 >> g.lTW) & (TH - 1);                                                        \
                const int gi = j0 >> (g.lTW + g.lTH);                                                          \
                const int n = ai0 + gi;                                                                        \
                const bool ok = ca_ok && n < g.N && aR0 + r < g.Hb && aC0 + c < g.Wb;                          \
                const float* ap = ok ? a_src + (size_t)n * a_img + (size_t)ca_l * HaWa +                        \
                                           (size_t)(WT::SA * (aR0 + r) + WT::PA) * g.Wa + WT::SA * (aC0 + c)    \
                                     : a_src;                                                                  \
                if (WT::SA == 1) {                                                                             \
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ap);                                       \
                    dst_.v[gq][0] = ok ? v[0] : 0.f; dst_.v[gq][1] = ok ? v[1] : 0.f;                          \
                    dst_.v[gq][2] = ok ? v[2] : 0.f; dst_.v[gq][3] = ok ? v[3] : 0.f;                          \
                } else {                                                                                       \
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(ap);                                      \
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(ap + 4);                                  \
                    dst_.v[gq][0] = ok ? v0[WT::PB] : 0.f; dst_.v[gq][1] = ok ? v0[WT::PB + 2] : 0.f;          \
                    dst_.v[gq][2] = ok ? v1[WT::PB] : 0.f; dst_.v[gq][3] = ok ? v1[WT::PB + 2] : 0.f;          \
                }                                                                                              \
                dst_.halo[gq] = (gi * (TH + 2) + r) * HS + c;                                                  \
                dst_.hx[gq][0] = 0; dst_.hx[gq][1] = 0; dst_.hx[gq][2] = 0; dst_.hx[gq][3] = 0;                \
            } else {                                                                                           \
                dst_.halo[gq] = 0;                                                                             \
                _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                \
                    const int jx = j0 + u;                                                                     \
                    const int c = jx & (TW - 1);                                                               \
                    const int r = (jx >> g.lTW) & (TH - 1);                                                    \
                    const int gi = jx >> (g.lTW + g.lTH);                                                      \
                    const int n = ai0 + gi;                                                                    \
                    float v = 0.f;                                                                             \
                    if (ca_ok && n < g.N && aR0 + r < g.Hb && aC0 + c < g.Wb)                                  \
                        v = a_src[(size_t)n * a_img + (size_t)ca_l * HaWa +                                    \
                                  (size_t)(WT::SA * (aR0 + r) + WT::PA) * g.Wa + WT::SA * (aC0 + c) + WT::PB]; \
                    dst_.v[gq][u] = v;                                                                         \
                    dst_.hx[gq][u] = (gi * (TH + 2) + r) * HS + c;                                             \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    }

    int tile = sp;
    if (tile < g.ntiles) GX_WG_PREFETCH(tile)
    int it = 0;
    for (; tile < g.ntiles; tile += nsp, ++it) {
        float* buf = lds + (it & 1) * BUF;
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            const int pos = tid + q * 256;
            if (pos < CHS) {
#pragma unroll
                for (int ch = 0; ch < 64; ++ch) buf[ch * BS + pos] = xin[ch][q];
            }
        }
        __syncthreads();
        if (tile + nsp < g.ntiles) GX_WG_PREFETCH(tile + nsp)

        // ---- K loop: this lane's k slot covers pixels [khalf*PT/2, (khalf+1)*PT/2) of the tile.  A values are
        //      fetched global -> registers one batch (4 groups = 16 k-steps = 16*NT MFMAs) ahead of their use.
        const int nbatches = PT >> 5;
        if (it == 0) GX_WG_LOAD_A(tile, 0, acur)
        for (int bt = 0; bt < nbatches; ++bt) {
            if (bt + 1 < nbatches) GX_WG_LOAD_A(tile, bt + 1, anxt)
            else if (tile + nsp < g.ntiles) GX_WG_LOAD_A(tile + nsp, 0, anxt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* bp = buf + b_row + acur.halo[gq] + (vecA ? u : 0) + acur.hx[gq][vecA ? 0 : u];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float b = bp[WT::ro(t) * HS + WT::co(t)];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur.v[gq][u], b, acc[t], 0, 0, 0);
                    }
                }
            }
            acur = anxt;
        }
    }
#undef GX_WG_PREFETCH
#undef GX_WG_LOAD_A
#undef GX_WG_TILE_ORIGIN
    // partial[split][gt][ca][cb]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* dst = partial + (((size_t)sp * g.Ttot + WT::gt(t)) * g.CApad + ca0 + wm * 32) * g.CBpad +
                     cb0 + wn * 32 + (lane & 31);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            dst[(size_t)row * g.CBpad] = acc[t][reg];
        }
    }
}

template <int WM>
__global__ void __launch_bounds__(256, 1)
wgrad_kernel(const float* __restrict__ a_src, const float* __restrict__ b_src,
             float* __restrict__ partial, WgradGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    wgrad_body<WM>(a_src, b_src, partial, g, lds, blockIdx.x, g.nsplit);
}

// fp32 products from six bf16 piece products on v_mfma_f32_32x32x16_bf16 (finding 13; the same helpers as
// gx_wgq.hip's wq_tile_b6): lane half h supplies 8 CONSECUTIVE contraction indices -- for a weight gradient 8 consecutive pixels
typedef __bf16 cw_bf16x8 __attribute__((ext_vector_type(8)));
struct CwB3 { cw_bf16x8 h, m, l; };
template <int N>
__device__ __forceinline__ void cw_split(const float (&v)[N], __bf16 (&h)[N], __bf16 (&m)[N], __bf16 (&l)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        h[i] = (__bf16)v[i];
        const float r1 = v[i] - (float)h[i];
        m[i] = (__bf16)r1;
        l[i] = (__bf16)(r1 - (float)m[i]);
    }
}
__device__ __forceinline__ f32x16 cw_mma6(const CwB3& a, const CwB3& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);      // small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}

// ---- lean variant for the common case (tile width >= 4, grid width a multiple of 4) ------------------------
// PMC on the kernel above (64->64 @64x64, B=32): 3.4 non-MFMA VALU + 1.3 SALU + 0.8 LDS instructions per MFMA and
// one wave per SIMD: the wave cannot issue them all under a 64-cycle MFMA, the matrix pipe sits at 62 %.  Here the
// per-lane A offsets and LDS halo offsets of every (batch, group) are computed ONCE per kernel (the pixel -> k-slot
// assignment does not depend on the tile), the tile width is a template parameter so that every tap / pixel offset
// of the LDS reads is an instruction immediate, and the batch loop is unrolled with two alternating A buffers (no
// register copies): ~1 VALU per MFMA is left.
// QUAD (layers of 32 x 32 channels, e.g. the BroadcastDecoder's canvas convs): the tensors are viewed as [N / 4, 128, H, W] -- four
// images side by side in the channel dimension -- and wave w multiplies the dy channels of sub-image w with the x channels
// of sub-image w (128 staged B rows): all four waves produce a 32 x 32 block that is wanted (a plain 64-channel tile of two
// images has two useful waves, one image one), each into its own quadrant of the 64 x 64 slab; wgrad_quad_reduce_kernel sums
// the quadrants.
// B6 (round 4; tiles at least 8 pixels wide): the same tile on the bf16 matrix pipe -- a group is 8 consecutive pixels of a tile row
// (two of the fp32 path's 4-pixel groups), the lane splits its 8 dy values and the 3 x 10-float halo window of x into bf16 pieces
// in registers, every tap's B operand is a shifted view of the split window: 54 MFMAs of 32 cycles per 16 pixels instead of 72 of 64.
template <int WM, int LTW, bool QUAD = false, bool B6 = false>
__device__ __forceinline__ void
wgrad_fast_body(const float* __restrict__ a_src, const float* __restrict__ b_src, float* __restrict__ partial,
                const float* __restrict__ zeros, const WgradGeom& g, const int bx, const int by) {
    static_assert(!B6 || LTW >= 3, "the bf16-pipe groups are 8 consecutive pixels of a tile row");
    using WT = WTap<WM>;
    constexpr int NT = WT::NT, HL = WT::HALO;
    constexpr int TW = 1 << LTW, HS = TW + 2 * HL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TH = 1 << g.lTH, G = 1 << g.lG;
    const int PT = TH * TW * G;          // pixels per tile (32, 64 or 128)
    const int nb = PT >> 5;              // A batches per tile (1, 2 or 4)
    const int CHS = G * (TH + 2 * HL) * HS;
    const int BS = CHS | 1;
    constexpr int BROWS = QUAD ? 128 : 64;
    const int BUF = BROWS * BS;
    const int sp = bx, nsp = g.nsplit;

    const int nbt = g.CBpad / 64;
    const int ca0 = QUAD ? 0 : (by / nbt) * 64;
    const int cb0 = QUAD ? 0 : (by % nbt) * 64;
    const int wm = wave >> 1, wn = wave & 1;
    const int a_img = g.CA * g.Ha * g.Wa;             // the host dispatch guarantees these fit 31 bits
    const int b_img = g.CB * g.Hb * g.Wb;
    const int HaWa = g.Ha * g.Wa, HbWb = g.Hb * g.Wb;
    const int nvalid_ch = QUAD ? 128 : (g.CB - cb0 < 64 ? g.CB - cb0 : 64);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int ca_l = QUAD ? wave * 32 + (lane & 31) : ca0 + wm * 32 + (lane & 31);
    const bool ca_ok = QUAD || ca_l < g.CA;
    const int khalf = lane >> 5;
    const int b_row = ((QUAD ? wave : wn) * 32 + (lane & 31)) * BS;
    const float* a_lane = a_src + (size_t)(ca_ok ? ca_l : 0) * HaWa + WT::PA * g.Wa;
    const bool exact = (g.tiles_h << g.lTH) == g.Hb && g.tiles_w * TW == g.Wb && (g.N & (G - 1)) == 0;
    const int jlane = khalf * (PT >> 1);               // first pixel of this lane's k slot
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // LDS-DMA bases are wave-uniform: keep them scalar

    // tile origin (image, row, col) of the current and the next tile of this workgroup, advanced by carries
    // (all wave-uniform SALU work; no per-tile integer divisions)
    struct Org { int img, th, tw; };
    const int d_tw = nsp % g.tiles_w, d_th = (nsp / g.tiles_w) % g.tiles_h, d_img = nsp / (g.tiles_w * g.tiles_h);
    Org cur, nxt;
    cur.tw = sp % g.tiles_w; cur.th = (sp / g.tiles_w) % g.tiles_h; cur.img = sp / (g.tiles_w * g.tiles_h);
#define GX_WF_ADVANCE(dst_, src_)                                                    \
    {                                                                                \
        int tw_ = src_.tw + d_tw, th_ = src_.th + d_th, im_ = src_.img + d_img;      \
        if (tw_ >= g.tiles_w) { tw_ -= g.tiles_w; ++th_; }                           \
        if (th_ >= g.tiles_h) { th_ -= g.tiles_h; ++im_; }                           \
        dst_.tw = tw_; dst_.th = th_; dst_.img = im_;                                \
    }
    GX_WF_ADVANCE(nxt, cur)
#define GX_WF_ORIGIN(org_, img0_, R0_, C0_) \
    const int img0_ = org_.img * G, R0_ = org_.th << g.lTH, C0_ = org_.tw * TW;
    // B (x halo tile) goes global -> LDS directly (LDS-DMA): thread `tid` owns halo position `tid` of every channel;
    // positions outside the image (and channels beyond CB) are fetched from a zero page so that no lane needs a
    // register, a select or an LDS store.  Destination: wave-uniform base + lane * 4 bytes.
#define GX_WF_PREFETCH_B(org_, dstbuf_)                                                             \
    {                                                                                                \
        GX_WF_ORIGIN(org_, pi0, pR0, pC0)                                                           \
        if (tid < CHS) {                                                                             \
            int rem = tid;                                                                           \
            const int gi = rem / ((TH + 2 * HL) * HS); rem -= gi * (TH + 2 * HL) * HS;               \
            const int i = rem / HS;                                                                  \
            const int jj = rem - i * HS;                                                             \
            const int row = pR0 - HL + i, col = pC0 - HL + jj;                                       \
            const bool inb = pi0 + gi < g.N && row >= 0 && row < g.Hb && col >= 0 && col < g.Wb;     \
            const float* lp = inb ? b_src + (size_t)(pi0 + gi) * b_img + (size_t)cb0 * HbWb + row * g.Wb + col \
                                  : zeros;                                                           \
            float* ldst = (dstbuf_) + wave_u * 64;                                                     \
            const float* gp = lp;                                                                    \
            int ch = 0;                                                                              \
            _Pragma("unroll 4") for (; ch < nvalid_ch; ++ch) {   /* rolled: one live pointer pair */ \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,  \
                                                 (__attribute__((address_space(3))) void*)ldst, 4, 0, 0); \
                gp += HbWb; ldst += BS;                                                              \
            }                                                                                        \
            for (; ch < BROWS; ++ch) {                            /* channels beyond CB: zeros */    \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)zeros, \
                                                 (__attribute__((address_space(3))) void*)ldst, 4, 0, 0); \
                ldst += BS;                                                                          \
            }                                                                                        \
        }                                                                                            \
    }
    // one batch of A: 4 groups x 4 consecutive pixels of this lane's channel row
#define GX_WF_LOAD_A(org_, bt_, dst_)                                                               \
    if (!(GX_WG_ABL & 2) || it == 0) {                                                               \
        GX_WF_ORIGIN(org_, ai0, aR0, aC0)                                                           \
        const float* abase = a_lane + (size_t)ai0 * a_img + (size_t)(WT::SA * aR0) * g.Wa + WT::SA * aC0; \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                           \
            const int j0 = jlane + 4 * (4 * (bt_) + gq);                                             \
            const int c = j0 & (TW - 1);                                                             \
            const int r = (j0 >> LTW) & (TH - 1);                                                    \
            const int gi = j0 >> (LTW + g.lTH);                                                      \
            bool ok = ca_ok;                                                                         \
            if (!exact) ok = ok && ai0 + gi < g.N && aR0 + r < g.Hb && aC0 + c < g.Wb;               \
            const float* ap = ok ? abase + (gi * a_img + WT::SA * r * g.Wa + WT::SA * c) : a_src;    \
            if (WT::SA == 1) {                                                                       \
                const f32x4 v = *reinterpret_cast<const f32x4*>(ap);                                 \
                dst_[gq][0] = ok ? v[0] : 0.f; dst_[gq][1] = ok ? v[1] : 0.f;                        \
                dst_[gq][2] = ok ? v[2] : 0.f; dst_[gq][3] = ok ? v[3] : 0.f;                        \
            } else {                                                                                 \
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(ap);                                \
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(ap + 4);                            \
                dst_[gq][0] = ok ? v0[WT::PB] : 0.f; dst_[gq][1] = ok ? v0[WT::PB + 2] : 0.f;        \
                dst_[gq][2] = ok ? v1[WT::PB] : 0.f; dst_[gq][3] = ok ? v1[WT::PB + 2] : 0.f;        \
            }                                                                                        \
        }                                                                                            \
    }
    // B values of one group: the taps of 4 consecutive pixels touch (rows used) x (cols used + 3) halo floats; they
    // are read one group AHEAD of the MFMAs that consume them (two alternating register sets), so the LDS latency
    // hides under the previous group's 4 * NT MFMAs instead of stalling the only wave of the SIMD.
    constexpr int RO0 = WT::ro(NT - 1) < WT::ro(0) ? WT::ro(NT - 1) : WT::ro(0);
    constexpr int RO1 = WT::ro(NT - 1) < WT::ro(0) ? WT::ro(0) : WT::ro(NT - 1);
    constexpr int CO0 = WT::co(NT - 1) < WT::co(0) ? WT::co(NT - 1) : WT::co(0);
    constexpr int CO1 = WT::co(NT - 1) < WT::co(0) ? WT::co(0) : WT::co(NT - 1);
    constexpr int NRO = RO1 - RO0 + 1, NCO = CO1 - CO0 + 4;
    float bva[NRO][NCO], bvb[NRO][NCO];
#define GX_WF_LOAD_B(bt_, gq_, dst_)                                                                 \
    {                                                                                                \
        const int j0 = jlane + 4 * (4 * (bt_) + (gq_));                                              \
        const int c = j0 & (TW - 1);                                                                 \
        const int r = (j0 >> LTW) & (TH - 1);                                                        \
        const int gi = j0 >> (LTW + g.lTH);                                                          \
        const float* bp = buf + b_row + (gi * (TH + 2 * HL) + r) * HS + c;                           \
        _Pragma("unroll") for (int rr = 0; rr < NRO; ++rr)                                           \
            _Pragma("unroll") for (int cc = 0; cc < NCO; ++cc)                                       \
                dst_[rr][cc] = bp[(RO0 + rr) * HS + CO0 + cc];                                       \
    }
#define GX_WF_MMA(src_, gq_, bv_)                                                                    \
    {                                                                                                \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                           \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(src_[gq_][u],                          \
                                                              bv_[WT::ro(t) - RO0][WT::co(t) - CO0 + u], \
                                                              acc[t], 0, 0, 0);                      \
        }                                                                                            \
    }
    // order pinned (sched_group_barrier): the LDS reads of the NEXT group first, then this group's 4 * NT MFMAs -- hipcc
    // otherwise sinks each read to just in front of its first use and the only wave of the SIMD waits out the LDS
    // latency (s_waitcnt lgkmcnt(0) one MFMA after the read: round-1 ISA, matrix pipe 0.6 busy)
    constexpr int NDS = NRO * ((NCO + 1) / 2);
#define GX_WF_SCHED()                                                                                \
    __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);                                             \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
    // one batch = 4 groups; the first group's B values were read by the previous batch (or GX_WF_LOAD_B(0, 0, bva)
    // after the tile's barrier); lastb_: last batch of the tile (the next tile's buffer is not readable yet)
#define GX_WF_COMPUTE(bt_, src_, lastb_)                                                             \
    {                                                                                                \
        GX_WF_LOAD_B(bt_, 1, bvb)                                                                    \
        GX_WF_MMA(src_, 0, bva)                                                                      \
        GX_WF_SCHED()                                                                                \
        GX_WF_LOAD_B(bt_, 2, bva)                                                                    \
        GX_WF_MMA(src_, 1, bvb)                                                                      \
        GX_WF_SCHED()                                                                                \
        GX_WF_LOAD_B(bt_, 3, bvb)                                                                    \
        GX_WF_MMA(src_, 2, bva)                                                                      \
        GX_WF_SCHED()                                                                                \
        if (!(lastb_)) GX_WF_LOAD_B((bt_) + 1, 0, bva)                                               \
        GX_WF_MMA(src_, 3, bvb)                                                                      \
        GX_WF_SCHED()                                                                                \
    }

    // bf16 pipe: one pair of groups = 8 consecutive pixels (P_ = 0, 1 within the batch)
    constexpr int NCO8 = CO1 - CO0 + 8;
#define GX_WF_PAIR_B6(bt_, src_, P_)                                                                   \
    {                                                                                                \
        float av_[8];                                                                                \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) { av_[u] = src_[2 * (P_)][u]; av_[4 + u] = src_[2 * (P_) + 1][u]; } \
        if constexpr (QUAD) bsum_ += ((av_[0] + av_[1]) + (av_[2] + av_[3])) + ((av_[4] + av_[5]) + (av_[6] + av_[7])); \
        __bf16 ah_[8], am_[8], al_[8];                                                               \
        cw_split<8>(av_, ah_, am_, al_);                                                             \
        CwB3 a3_;                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) { a3_.h[i] = ah_[i]; a3_.m[i] = am_[i]; a3_.l[i] = al_[i]; } \
        const int j0 = jlane + 4 * (4 * (bt_) + 2 * (P_));                                           \
        const int c = j0 & (TW - 1);                                                                 \
        const int r = (j0 >> LTW) & (TH - 1);                                                        \
        const int gi = j0 >> (LTW + g.lTH);                                                          \
        const float* bp = buf + b_row + (gi * (TH + 2 * HL) + r) * HS + c;                           \
        _Pragma("unroll") for (int rr = 0; rr < NRO; ++rr) {                                         \
            float bw_[NCO8];                                                                         \
            _Pragma("unroll") for (int cc = 0; cc < NCO8; ++cc) bw_[cc] = bp[(RO0 + rr) * HS + CO0 + cc]; \
            __bf16 bh_[NCO8], bm_[NCO8], bl_[NCO8];                                                  \
            cw_split<NCO8>(bw_, bh_, bm_, bl_);                                                      \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                         \
                if (WT::ro(t) - RO0 != rr) continue;                                                 \
                CwB3 b3_;                                                                            \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                      \
                    b3_.h[i] = bh_[WT::co(t) - CO0 + i]; b3_.m[i] = bm_[WT::co(t) - CO0 + i]; b3_.l[i] = bl_[WT::co(t) - CO0 + i]; \
                }                                                                                    \
                acc[t] = cw_mma6(a3_, b3_, acc[t]);                                                  \
            }                                                                                        \
        }                                                                                            \
    }
#define GX_WF_COMPUTE_B6(bt_, src_) { GX_WF_PAIR_B6(bt_, src_, 0) GX_WF_PAIR_B6(bt_, src_, 1) }

    float a0[4][4], a1[4][4];
    float bsum_ = 0.f;        // QUAD && B6: sum of this lane's dy values (one channel of one sub-image, half of each tile's pixels)
    int tile = sp;
    int it = 0;
    if (tile < g.ntiles) {
        GX_WF_PREFETCH_B(cur, lds)
        GX_WF_LOAD_A(cur, 0, a0)
    }
    for (; tile < g.ntiles; tile += nsp, ++it) {
        float* buf = lds + (it & 1) * BUF;
#if !(GX_WG_ABL & 4)
        __syncthreads();     // this tile's B has landed (vmcnt drained before the barrier); the other buffer is free
#endif
        const bool more = tile + nsp < g.ntiles;
#if !(GX_WG_ABL & 1)
        if (more) GX_WF_PREFETCH_B(nxt, lds + ((it + 1) & 1) * BUF)
#endif
        if constexpr (B6) {
            if (nb == 4) {
                GX_WF_LOAD_A(cur, 1, a1)
                GX_WF_COMPUTE_B6(0, a0)
                GX_WF_LOAD_A(cur, 2, a0)
                GX_WF_COMPUTE_B6(1, a1)
                GX_WF_LOAD_A(cur, 3, a1)
                GX_WF_COMPUTE_B6(2, a0)
                if (more) GX_WF_LOAD_A(nxt, 0, a0)
                GX_WF_COMPUTE_B6(3, a1)
            } else if (nb == 2) {
                GX_WF_LOAD_A(cur, 1, a1)
                GX_WF_COMPUTE_B6(0, a0)
                if (more) GX_WF_LOAD_A(nxt, 0, a0)
                GX_WF_COMPUTE_B6(1, a1)
            } else {
                if (more) GX_WF_LOAD_A(nxt, 0, a1)
                GX_WF_COMPUTE_B6(0, a0)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int u = 0; u < 4; ++u) a0[gq][u] = a1[gq][u];
            }
            cur = nxt;
            GX_WF_ADVANCE(nxt, cur)
            continue;
        }
        GX_WF_LOAD_B(0, 0, bva)
        if (nb == 4) {
            GX_WF_LOAD_A(cur, 1, a1)
            GX_WF_COMPUTE(0, a0, false)
            GX_WF_LOAD_A(cur, 2, a0)
            GX_WF_COMPUTE(1, a1, false)
            GX_WF_LOAD_A(cur, 3, a1)
            GX_WF_COMPUTE(2, a0, false)
            if (more) GX_WF_LOAD_A(nxt, 0, a0)
            GX_WF_COMPUTE(3, a1, true)
        } else if (nb == 2) {
            GX_WF_LOAD_A(cur, 1, a1)
            GX_WF_COMPUTE(0, a0, false)
            if (more) GX_WF_LOAD_A(nxt, 0, a0)
            GX_WF_COMPUTE(1, a1, true)
        } else {
            if (more) GX_WF_LOAD_A(nxt, 0, a1)
            GX_WF_COMPUTE(0, a0, true)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int u = 0; u < 4; ++u) a0[gq][u] = a1[gq][u];
        }
        cur = nxt;
        GX_WF_ADVANCE(nxt, cur)
    }
#undef GX_WF_ADVANCE
#undef GX_WF_ORIGIN
#undef GX_WF_PREFETCH_B
#undef GX_WF_LOAD_A
#undef GX_WF_COMPUTE
#undef GX_WF_SCHED
#undef GX_WF_LOAD_B
#undef GX_WF_MMA
#undef GX_WF_PAIR_B6
#undef GX_WF_COMPUTE_B6
    if constexpr (QUAD && B6) {
        if (g.bias_part) g.bias_part[(size_t)sp * 256 + tid] = bsum_;
    }
    // partial[split][gt][ca][cb]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* dst = partial + (((size_t)sp * g.Ttot + WT::gt(t)) * g.CApad + ca0 + wm * 32) * g.CBpad +
                     cb0 + wn * 32 + (lane & 31);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            dst[(size_t)row * g.CBpad] = acc[t][reg];
        }
    }
}

template <int WM, int LTW, bool QUAD = false, bool B6 = false>
__global__ void __launch_bounds__(256, 1)
wgrad_fast_kernel(const float* __restrict__ a_src, const float* __restrict__ b_src,
                  float* __restrict__ partial, const float* __restrict__ zeros, WgradGeom g) {
    wgrad_fast_body<WM, LTW, QUAD, B6>(a_src, b_src, partial, zeros, g, blockIdx.x, blockIdx.y);
}

// Several layers in one grid (blockIdx.z = job): the weight gradients of layers too small for the stream-K launch (4 x 4
// grids) each cost a launch of ~17 us for ~40 MFLOP -- twelve per MONet step (its shared UNet runs six times), two per GENESIS-V2
// step.  With deferral on they are queued like the stream-K jobs and launched together at the flush, in front of the batched
// slab reduce.  (The job record is COPIED out of kernel-argument memory: DESIGN.md finding 26.)
constexpr int kMaxWfJobs = 16;
struct WfJob { const float* a; const float* b; float* partial; WgradGeom g; };
struct WfTable { WfJob job[kMaxWfJobs]; };
template <int WM, int LTW>
__global__ void __launch_bounds__(256, 1)
wgrad_fast_multi_kernel(const WfTable tab, const float* __restrict__ zeros) {
    const WfJob jb = tab.job[blockIdx.z];
    if ((int)blockIdx.x >= jb.g.nsplit || (int)blockIdx.y >= (jb.g.CApad / 64) * (jb.g.CBpad / 64)) return;
    wgrad_fast_body<WM, LTW, false, false>(jb.a, jb.b, jb.partial, zeros, jb.g, blockIdx.x, blockIdx.y);
}

// All four output-parity classes of the transposed conv's weight gradient in ONE launch: blockIdx.x ranges are
// assigned per class in proportion to its taps (9 : 6 : 6 : 4) so that every workgroup does about the same work,
// and the 9-tap class is dispatched first.
__global__ void __launch_bounds__(256, 1)
wgrad_deconv_kernel(const float* __restrict__ a_src, const float* __restrict__ b_src,
                    float* __restrict__ partial, WgradGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int bx = blockIdx.x;
    if (bx < g.cls_begin[1])
        wgrad_body<W_D00>(a_src, b_src, partial, g, lds, bx, g.cls_begin[1]);
    else if (bx < g.cls_begin[2])
        wgrad_body<W_D01>(a_src, b_src, partial, g, lds, bx - g.cls_begin[1], g.cls_begin[2] - g.cls_begin[1]);
    else if (bx < g.cls_begin[3])
        wgrad_body<W_D10>(a_src, b_src, partial, g, lds, bx - g.cls_begin[2], g.cls_begin[3] - g.cls_begin[2]);
    else
        wgrad_body<W_D11>(a_src, b_src, partial, g, lds, bx - g.cls_begin[3], g.cls_begin[4] - g.cls_begin[3]);
}

// QUAD slabs [split][t][64][64] -> dw [C][C][T] (C <= 32): quadrant q = (wm, wn) holds sub-image q's 32 x 32 block.  Block =
// db[c] = sum over the splits' 8 records per channel (4 sub-images x 2 pixel halves: thread = wave * 64 + half * 32 + c), fp64
__global__ void __launch_bounds__(256)
quad_bias_reduce_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ db) {
    __shared__ double red[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < nsplit * 8; r += 256) s += part[(size_t)(r >> 3) * 256 + (r & 7) * 32 + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[c] = (float)red[0];
}

// 64 consecutive (t, ca, cb) outputs x the 4 quadrants x 4 interleaved split groups; fixed summation order.
__global__ void __launch_bounds__(1024)
wgrad_quad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nsplit, int Ttot, int C) {
    __shared__ float red[16][64];
    const int total = Ttot * C * C;
    const int e = threadIdx.x & 63, q = (threadIdx.x >> 6) & 3, grp = threadIdx.x >> 8;     // 4 interleaved split groups
    const int idx = blockIdx.x * 64 + e;
    float s = 0.f;
    int cb = 0, ca = 0, t = 0;
    if (idx < total) {
        cb = idx % C;
        ca = (idx / C) % C;
        t = idx / (C * C);
        const size_t stride = (size_t)Ttot * 64 * 64;
        const float* p = partial + ((size_t)t * 64 + ca + 32 * (q >> 1)) * 64 + cb + 32 * (q & 1);
        float s0 = 0.f, s1 = 0.f;
        int sp = grp;
        for (; sp + 4 < nsplit; sp += 8) { s0 += p[sp * stride]; s1 += p[(sp + 4) * stride]; }
        if (sp < nsplit) s0 += p[sp * stride];
        s = s0 + s1;
    }
    red[grp * 4 + q][e] = s;
    __syncthreads();
    if (threadIdx.x < 64 && idx < total) {
        float qs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) qs[k] = (red[k][e] + red[4 + k][e]) + (red[8 + k][e] + red[12 + k][e]);
        dw[((size_t)ca * C + cb) * Ttot + t] = (qs[0] + qs[1]) + (qs[2] + qs[3]);
    }
}

// dW = sum over splits.  layout 0: W[ca][cb][T] (conv3x3: ca=co, cb=ci); layout 1: W[cb][ca][T] (deconv).
// Block = 64 consecutive (t, ca, cb) elements x 4 interleaved split groups (fixed summation order).
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nsplit_all,
                    int Ttot, int CA, int CB, int CApad, int CBpad, int layout, int ns0, int ns1, int ns2, int ns3,
                    int ca0 = 0, int cb0 = 0, int CAf = 0, int CBf = 0, int accumulate = 0) {
    __shared__ float red[4][64];
    const int total = Ttot * CA * CB;
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;
    float s = 0.f;
    int cb = 0, ca = 0, t = 0;
    if (idx < total) {
        cb = idx % CB;
        ca = (idx / CB) % CA;
        t = idx / (CB * CA);
        // merged deconv launch: the splits of tap t are those of its parity class (kh & 1, kw & 1)
        int nsplit = nsplit_all;
        if (ns0 > 0) {
            const int cls = ((t / 5) & 1) * 2 + ((t % 5) & 1);
            nsplit = cls == 0 ? ns0 : (cls == 1 ? ns1 : (cls == 2 ? ns2 : ns3));
        } else if (ns0 < 0) {
            nsplit = t < 15 ? ns1 : ns2;          // 5 x 5 stride-1 conv: kernel rows 0-2 and 3-4 were two jobs (gx_wgq_c5)
        }
        const size_t stride = (size_t)Ttot * CApad * CBpad;
        const float* p = partial + ((size_t)t * CApad + ca) * CBpad + cb;
        float s0 = 0.f, s1 = 0.f;
        int sp = grp;
        for (; sp + 4 < nsplit; sp += 8) { s0 += p[sp * stride]; s1 += p[(sp + 4) * stride]; }
        if (sp < nsplit) s0 += p[sp * stride];
        s = s0 + s1;
    }
    red[grp][e] = s;
    __syncthreads();
    if (grp == 0 && idx < total) {
        const float r = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (!CAf) { CAf = CA; CBf = CB; }
        float* d = layout == 0 ? dw + ((size_t)(ca0 + ca) * CBf + cb0 + cb) * Ttot + t
                               : dw + ((size_t)(cb0 + cb) * CAf + ca0 + ca) * Ttot + t;
        *d = accumulate ? *d + r : r;
    }
}

// all queued weight-gradient reductions in one launch: blockIdx.y = queue entry (table passed by value).
// A thread owns 4 consecutive cb of one (t, ca) row (the slabs are [split][t][CApad][CBpad], CBpad a multiple of 64:
// 16-byte loads, 1 KB per wave and load) and every 4th split; per element the summation order is the scalar kernel's
// (two alternating partial sums per split group, then the four groups), so both produce the same bits.
constexpr int kRedPerLaunch = 40;      // 40 x 88 B: the table travels as a kernel argument
struct WgradRedTable { GxWgradRed e[kRedPerLaunch]; };
__global__ void __launch_bounds__(256)
wgrad_reduce_batch_kernel(const WgradRedTable tab) {
    const GxWgradRed& r = tab.e[blockIdx.y];
    __shared__ f32x4 red[4][64];
    const int cb4n = r.CBpad >> 2;
    const int total4 = r.Ttot * r.CA * cb4n;
    if ((int)blockIdx.x * 64 >= total4) return;
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int cb = 0, ca = 0, t = 0;
    if (idx < total4) {
        cb = (idx % cb4n) << 2;
        ca = (idx / cb4n) % r.CA;
        t = idx / (cb4n * r.CA);
        int nsplit = r.nsplit;
        if (r.ns0 > 0) {
            const int cls = ((t / 5) & 1) * 2 + ((t % 5) & 1);
            nsplit = cls == 0 ? r.ns0 : (cls == 1 ? r.ns1 : (cls == 2 ? r.ns2 : r.ns3));
        } else if (r.ns0 < 0) {
            nsplit = t < 15 ? r.ns1 : r.ns2;      // 5 x 5 stride-1 conv: kernel rows 0-2 | 3-4
        }
        const size_t stride = (size_t)r.Ttot * r.CApad * r.CBpad;
        const float* p = r.partial + ((size_t)t * r.CApad + ca) * r.CBpad + cb;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        int sp = grp;
#pragma unroll 4
        for (; sp + 4 < nsplit; sp += 8) {
            s0 += *reinterpret_cast<const f32x4*>(p + sp * stride);
            s1 += *reinterpret_cast<const f32x4*>(p + (sp + 4) * stride);
        }
        if (sp < nsplit) s0 += *reinterpret_cast<const f32x4*>(p + sp * stride);
        s = s0 + s1;
    }
    red[grp][e] = s;
    __syncthreads();
    if (grp == 0 && idx < total4) {
        const f32x4 v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        // ACCUMULATES: the destination is a zeroed gradient buffer; a parameter used more than once per iteration
        // (MONet's recurrent UNet) has received its other contributions by the time the queue is flushed
        const int CAf = r.CAf ? r.CAf : r.CA, CBf = r.CAf ? r.CBf : r.CB;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (cb + j < r.CB) {
                if (r.layout == 0) r.dw[((size_t)(r.ca0 + ca) * CBf + r.cb0 + cb + j) * r.Ttot + t] += v[j];
                else r.dw[((size_t)(r.cb0 + cb + j) * CAf + r.ca0 + ca) * r.Ttot + t] += v[j];
            }
    }
}

// ------------------------------------------------------------------ host-side geometry
int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// pixel tile of `npix` (256 for tapconv) over a Hb x Wb base grid of N images: power-of-two TH x TW x G
// maximising the fraction of useful pixels (1.0 for power-of-two grids; 72x72 -> 8x8 tiles of 4 images), then
// the widest rows (coalescing) and tallest tiles among ties; the halo tile must fit `max_chs` floats per channel per plane.
void pick_tile(int Hb, int Wb, int npix, int planes, int max_chs, int* lTH, int* lTW, int* lG, int halo = 1) {
    double best_eff = -1.0;
    int bTW = 1, bTH = 1;
    for (int TW = 1; TW <= npix && TW <= 64; TW <<= 1) {
        if (TW > 1 && (TW >> 1) >= Wb) break;               // no wider than needed
        for (int TH = 1; TH * TW <= npix; TH <<= 1) {
            if (TH > 1 && (TH >> 1) >= Hb) break;
            const int G = npix / (TH * TW);
            if (planes * G * (TH + 2 * halo) * (TW + 2 * halo) > max_chs) continue;
            const double eff = (double)Hb * Wb / ((double)gx_ceil_div(Wb, TW) * TW * gx_ceil_div(Hb, TH) * TH);
            // ties: widest rows, then tallest tile (fewest images per tile = smallest halo)
            if (eff > best_eff + 1e-9 || (eff > best_eff - 1e-9 && (TW > bTW || (TW == bTW && TH > bTH)))) {
                best_eff = eff; bTW = TW; bTH = TH;
            }
        }
    }
    *lTW = ilog2(bTW); *lTH = ilog2(bTH); *lG = ilog2(npix / (bTW * bTH));
}

struct TapPlan {
    ConvGeom g;
    int mw;            // 1: 256-pixel tile, 2: 128-pixel tile (waves 2 x 2)
    int npos;          // template NPOS to use
    size_t lds_bytes;
    size_t out_elems;  // N*M*Ho*Wo
    dim3 grid;
};

template <int MODE>
int plan_tapconv(int N, int K, int M, int Mpad_pack, int Hb, int Wb, int Hi, int Wi, int Ho, int Wo, int par_a,
                 TapPlan* pl, const char* name, int ymult = 1, int npix = 256) {
    using TC = TapCfg<MODE>;
    ConvGeom& g = pl->g;
    g.N = N; g.K = K; g.M = M;
    g.Kpad = gx_round_up(K, 8); g.Mpad = Mpad_pack;
    g.Hb = Hb; g.Wb = Wb; g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo; g.par_a = par_a;
    constexpr int LO_ = (MODE == M_DG) ? 8 : 2;
    pick_tile(Hb, Wb, npix, TC::PLANES, 2 * LO_ * 256, &g.lTH, &g.lTW, &g.lG, TC::HALO);
    pl->mw = 256 / npix;
    const int TH = 1 << g.lTH, TW = 1 << g.lTW, G = 1 << g.lG;
    g.tiles_h = gx_ceil_div(Hb, TH); g.tiles_w = gx_ceil_div(Wb, TW);
    g.act = 0;
    g.zeros = nullptr;
    g.stats = nullptr;
    g.stats_parts = 0;
    const int CHS = TC::PLANES * G * (TH + 2 * TC::HALO) * (TW + 2 * TC::HALO);
    const int need = gx_ceil_div(CHS, 256);
    const int lo = (MODE == M_DG) ? 8 : 2;
    pl->npos = need <= lo ? lo : 2 * lo;
    if (need > 2 * lo) { gx_set_error("%s: halo tile too large (%d floats/channel)", name, CHS); return GX_EINVAL; }
    pl->lds_bytes = (size_t)2 * (TC::KC * CHS + TC::NT * TC::KC * 64) * sizeof(float);
    if (pl->lds_bytes > 160 * 1024) { gx_set_error("%s: LDS %zu > 160KiB", name, pl->lds_bytes); return GX_EINVAL; }
    const int ptiles = g.tiles_h * g.tiles_w * gx_ceil_div(N, G);
    const int mtiles = gx_ceil_div(M, 64);
    const int nchunks = g.Kpad / TC::KC;
    // split the channel reduction when the (pixel-tile x channel-tile) grid cannot fill 256 CUs x 2
    int nsplit = 1;
    const int base = ptiles * mtiles * ymult;   // ymult: parity classes sharing the launch
    if (base < 384) {
        nsplit = gx_ceil_div(512, base);
        if (nsplit > nchunks) nsplit = nchunks;
        if (nsplit > 64) nsplit = 64;
        if (nsplit < 1) nsplit = 1;
    }
    g.chunks_per_split = gx_ceil_div(nchunks, nsplit);
    nsplit = gx_ceil_div(nchunks, g.chunks_per_split);
    g.nsplit = nsplit;
    pl->out_elems = (size_t)N * M * Ho * Wo;
    pl->grid = dim3(ptiles, mtiles, nsplit);
    return GX_OK;
}

template <int MODE, int NPOS, bool DMA, int MW, bool HH>
void launch_tapconv_inst3(const float* in, const float* wp, const float* bias, float* out, const ConvGeom& g,
                          dim3 grid, size_t lds_bytes, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tapconv_kernel<MODE, NPOS, DMA, MW, HH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((tapconv_kernel<MODE, NPOS, DMA, MW, HH>), grid, dim3(256), lds_bytes, s, in, wp, bias, out, g);
}
// layers of <= 32 output channels (conv3x3 and the stride-1 5x5 conv: MONet / GENESIS, e.g. the data gradients of the gated
// 32 -> 2 x 32 stacks) skip the upper MFMA tile
template <int MODE, int NPOS, bool DMA, int MW = 1>
void launch_tapconv_inst2(const float* in, const float* wp, const float* bias, float* out, const ConvGeom& g,
                          dim3 grid, size_t lds_bytes, hipStream_t s) {
    constexpr bool LOW = MODE == M_C3 || MODE == M_C5;
    if (LOW && g.M <= 32) launch_tapconv_inst3<MODE, NPOS, DMA, MW, !LOW>(in, wp, bias, out, g, grid, lds_bytes, s);
    else launch_tapconv_inst3<MODE, NPOS, DMA, MW, true>(in, wp, bias, out, g, grid, lds_bytes, s);
}

const float* zero_page(hipStream_t s);

template <int MODE, int NPOS>
void launch_tapconv_inst(const float* in, const float* wp, const float* bias, float* out, const TapPlan& pl,
                         hipStream_t s) {
    ConvGeom g = pl.g;
    // LDS-DMA staging measured per mode (B=32, K=7): transposed conv fwd 32->64 481 -> 431 us, its dgrad 462 -> 454,
    // conv3x3 1-2 % slower -> on for the 5x5 modes, off for conv3x3 (GENESIS_TAPCONV_DMA=0/1 forces all modes)
    static const char* dma_env = getenv("GENESIS_TAPCONV_DMA");
    const bool dma = dma_env ? dma_env[0] == '1' : MODE != M_C3;
    g.zeros = dma ? zero_page(s) : nullptr;
    if ((MODE == M_C3 || MODE == M_DG) && pl.mw == 2) {
        g.zeros = nullptr;
        launch_tapconv_inst2<MODE, NPOS, false, 2>(in, wp, bias, out, g, pl.grid, pl.lds_bytes, s);
        return;
    }
    if (g.zeros) launch_tapconv_inst2<MODE, NPOS, true>(in, wp, bias, out, g, pl.grid, pl.lds_bytes, s);
    else launch_tapconv_inst2<MODE, NPOS, false>(in, wp, bias, out, g, pl.grid, pl.lds_bytes, s);
}

template <int NPOS, bool DMA, int MW = 1, bool STATS = false>
void launch_dt(dim3 grid, size_t lds_bytes, hipStream_t s, const float* x, const float* wp0, const float* wp1,
               const float* bias, float* dst, const ConvGeom& g) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tapconv_dt_kernel<NPOS, DMA, MW, STATS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((tapconv_dt_kernel<NPOS, DMA, MW, STATS>), grid, dim3(256), lds_bytes, s, x, wp0, wp1, bias,
                       dst, g);
}

// `dst` = final output when nsplit == 1, else the partial slabs [nsplit][N,M,Ho,Wo]
template <int MODE>
int launch_tapconv(const float* in, const float* wp, const float* bias, float* dst, const TapPlan& pl, hipStream_t s,
                   const char* name) {
    using TC = TapCfg<MODE>;
    const ConvGeom& g = pl.g;
    {
        // algorithmic work: every tap counted (zero-padding taps included), useful channels only
        const double flops = 2.0 * g.N * (double)g.M * g.K * TC::NT * g.Hb * g.Wb;
        const double bytes = 4.0 * ((double)g.N * g.K * g.Hi * g.Wi + (double)g.N * g.M * g.Ho * g.Wo / TC::NCLS +
                                    (double)TC::NT * g.K * g.M);
        GxProf pf(MODE == M_C5 ? KID_DCONV : KID_TAPCONV_C3 + MODE, s, flops, bytes);
        constexpr int LO = (MODE == M_DG) ? 8 : 2;
        if (pl.npos == LO) launch_tapconv_inst<MODE, LO>(in, wp, bias, dst, pl, s);
        else launch_tapconv_inst<MODE, 2 * LO>(in, wp, bias, dst, pl, s);
    }
    GX_CHECK_LAUNCH(name);
    return GX_OK;
}

int launch_splitk_reduce(const float* part, const float* bias, float* out, const TapPlan& pl, hipStream_t s) {
    const size_t total = pl.out_elems;
    size_t blocks = (total / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    {
        GxProf pf(KID_SPLITK_REDUCE, s, 0.0, 4.0 * (pl.g.nsplit + 1.0) * (double)total);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, part, bias, out, total,
                           pl.g.nsplit, pl.g.M, pl.g.Ho * pl.g.Wo, pl.g.act);
    }
    GX_CHECK_LAUNCH("splitk_reduce");
    return GX_OK;
}

// ---- packed-weight cache -------------------------------------------------------------------------------
// Weights change once per optimiser step but every conv entry point re-packs its weight tensor (35 launches of
// ~5 us per training step).  A cache (one per training loop) records the (weight pointer, pack mode) pairs seen
// during one recorded iteration; afterwards gx_weight_cache_refresh() re-packs all of them in ONE launch at the
// start of an iteration and the conv entry points are served from the cache until gx_weight_cache_release().
struct PackEntry {
    const float* w; float* wp;
    int pack, Co, Ci, NT, Kpad, Mpad;
    int chunk0;             // first workgroup (chunk of kPackChunk elements) of this entry in the batched launch
};
constexpr int kPackChunk = 2048;
struct PackCache {
    std::vector<PackEntry> entries;
    PackEntry* dev = nullptr;
    int* dev_map = nullptr;     // entry index of every chunk (the batched launch is ONE flat grid: work per workgroup
    int nchunks = 0;            // is the same whatever the entry's size; 128 workgroups per entry took 26 us, set by
    bool alive = false;         // the largest Winograd packing)
};
std::vector<PackCache> g_caches;       // (created / destroyed under g_cache_mutex; a cache is used by the context that made it)
std::mutex g_cache_mutex;
#define g_cache_recording (gx_ctx_flags().cache_recording)   // cache id being recorded by this context, or -1
#define g_cache_active (gx_ctx_flags().cache_active)         // cache id this context's conv calls are served from, or -1

// the fp16 x 3 packs' (kinds >= 40) weight tensors: largest magnitude -> the packing's trailer; one workgroup per cache entry (the
// others return at once); 3 x 3 (kinds 40 / 41) or 5 x 5 weights: Co * Ci * 9 | 25 floats
__global__ void __launch_bounds__(1024)
pack_amax_batch_kernel(const PackEntry* __restrict__ entries) {
    const PackEntry e = entries[blockIdx.x];
    if (e.pack < 40) return;
    const float r = gx_wg1024_amax(e.w, e.Co * e.Ci * ((e.pack <= 41 || e.pack == 45 || e.pack == 46) ? 9 : 25));
    if (threadIdx.x == 0) *reinterpret_cast<float*>(reinterpret_cast<char*>(e.wp) + gx_kq_h_amax_off(e.Kpad, e.Mpad, e.NT)) = r;
}

__global__ void pack_weights_batch_kernel(const PackEntry* __restrict__ entries, const int* __restrict__ map) {
    const PackEntry e = entries[map[blockIdx.x]];
    const int total = e.NT * e.Kpad * e.Mpad;
    const int begin = (blockIdx.x - e.chunk0) * kPackChunk, end = begin + kPackChunk < total ? begin + kPackChunk : total;
    // (45 / 46: the Winograd operands U = G g G^T, |U| <= 2.25 max |g|)
    const int f16_exp = e.pack >= 40 ? gx_f16_scale_exp(((e.pack == 45 || e.pack == 46) ? 2.25f : 1.f) * *reinterpret_cast<const float*>(reinterpret_cast<const char*>(e.wp) + gx_kq_h_amax_off(e.Kpad, e.Mpad, e.NT))) : 0;
    for (int idx = begin + threadIdx.x; idx < end; idx += blockDim.x) {
        const int m = idx % e.Mpad;
        const int k = (idx / e.Mpad) % e.Kpad;
        const int t = idx / (e.Mpad * e.Kpad);
        if (e.pack >= 20) { pack_h_store(e.w, e.wp, e.pack, e.Co, e.Ci, m, k, t, e.NT, e.Kpad, f16_exp); continue; }
        e.wp[pack_dest(e.pack, idx, m, k, t, e.Kpad, e.NT)] = pack_weight_value(e.w, e.pack, e.Co, e.Ci, m, k, t);
    }
}

// Packs w into `wp` (caller workspace) -- or, inside a refreshed cache window, returns the cached packing.
int launch_pack(const float* w, float* wp, int pack, int Co, int Ci, int NT, int Kpad, int Mpad, hipStream_t s,
                const float** wp_used) {
    *wp_used = wp;
    if (g_cache_active >= 0) {
        for (const PackEntry& e : g_caches[g_cache_active].entries)
            if (e.w == w && e.pack == pack && e.Co == Co && e.Ci == Ci) { *wp_used = e.wp; return GX_OK; }
    }
    if (g_cache_recording >= 0) {
        PackCache& c = g_caches[g_cache_recording];
        bool found = false;
        for (const PackEntry& e : c.entries) found = found || (e.w == w && e.pack == pack && e.Co == Co && e.Ci == Ci);
        if (!found) {
            PackEntry e{w, nullptr, pack, Co, Ci, NT, Kpad, Mpad, 0};
            const size_t pbytes = pack >= 20 ? gx_kq_deconv_h_pack_bytes(Kpad, Mpad, NT) : (size_t)NT * Kpad * Mpad * sizeof(float);
            if (hipMalloc((void**)&e.wp, pbytes) != hipSuccess) {
                gx_set_error("weight cache: hipMalloc failed");
                return GX_ELAUNCH;
            }
            c.entries.push_back(e);
        }
    }
    const int total = NT * Kpad * Mpad;
    const int blocks = gx_ceil_div(total, 256) > 1024 ? 1024 : gx_ceil_div(total, 256);
    if (pack >= 40) {
        const int rc = gx_kq_weight_amax_launch(w, Co * Ci * ((pack <= 41 || pack == 45 || pack == 46) ? 9 : 25), reinterpret_cast<float*>(reinterpret_cast<char*>(wp) + gx_kq_h_amax_off(Kpad, Mpad, NT)), s);
        if (rc) return rc;
    }
    {
        GxProf pf(KID_PACK_WEIGHTS, s, 0.0, 8.0 * total);
        hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, s, w, wp, pack, Co, Ci, NT, Kpad, Mpad);
    }
    GX_CHECK_LAUNCH("pack_weights");
    return GX_OK;
}

// conv3x3 plan: 256-pixel tiles; grids that would have to split the channel reduction use 128-pixel tiles instead
// (twice the workgroups, half the partial slabs or none).  Measured on all eight under-filled UNet layers (B=32):
// 5-25 % faster each, 739 -> 690 us forward and 711 -> 660 us dgrad per step.  GENESIS_TAPCONV_MW2=0 disables.
int plan_c3(int N, int K, int M, int Mpad, int H, int W, TapPlan* pl, const char* name) {
    int rc = plan_tapconv<M_C3>(N, K, M, Mpad, H, W, H, W, H, W, 0, pl, name);
    if (rc || pl->g.nsplit == 1) return rc;
    static const char* env = getenv("GENESIS_TAPCONV_MW2");
    if (env && env[0] == '0') return rc;
    TapPlan p2;
    if (plan_tapconv<M_C3>(N, K, M, Mpad, H, W, H, W, H, W, 0, &p2, name, 1, 128) != GX_OK) return rc;
    *pl = p2;
    return rc;
}

struct WgradPlan {
    WgradGeom g;
    size_t lds_bytes;
    size_t ws_floats;
};

int plan_wgrad(int N, int CA, int CB, int Hb, int Wb, int SA, int Ttot, int ncls_launches, WgradPlan* pl, int max_npix = 128,
               int b_rows = 64, int halo = 1) {
    WgradGeom& g = pl->g;
    g.bias_part = nullptr;
    g.N = N; g.CA = CA; g.CB = CB;
    g.CApad = gx_round_up(CA, 64); g.CBpad = gx_round_up(CB, 64);
    g.Hb = Hb; g.Wb = Wb; g.Ha = SA * Hb; g.Wa = SA * Wb;
    // pixel tile: 128 pixels (fewer for tiny grids) as G images x TH rows x TW<=32 cols with a halo tile of
    // <= 256 positions per channel (one staged element per thread per channel); power-of-two tile dims, any grid
    int TW = 1, TH = 1, G = 1;
    {
        double best_eff = -1.0;
        int best_np = 0;
        for (int npix = max_npix; npix >= 32; npix >>= 1) {
            for (int tw = 1; tw <= 32 && tw <= npix; tw <<= 1) {
                if (tw > 1 && (tw >> 1) >= Wb) break;
                for (int th = 1; th * tw <= npix; th <<= 1) {
                    if (th > 1 && (th >> 1) >= Hb) break;
                    const int gg = npix / (th * tw);
                    if (gg * (th + 2 * halo) * (tw + 2 * halo) > 256) continue;
                    const double eff = (double)Hb * Wb / ((double)gx_ceil_div(Wb, tw) * tw * gx_ceil_div(Hb, th) * th);
                    const bool better = eff > best_eff + 1e-9 ||
                                        (eff > best_eff - 1e-9 &&
                                         (npix > best_np || (npix == best_np && (tw > TW || (tw == TW && th > TH)))));
                    if (better) { best_eff = eff; best_np = npix; TW = tw; TH = th; G = gg; }
                }
            }
            if (best_eff > 0.999) break;   // a full-size tile that wastes nothing: keep the largest
        }
    }
    g.lTH = ilog2(TH); g.lTW = ilog2(TW); g.lG = ilog2(G);
    g.tiles_h = gx_ceil_div(Hb, TH); g.tiles_w = gx_ceil_div(Wb, TW);
    g.ntiles = g.tiles_h * g.tiles_w * gx_ceil_div(N, G);
    // one workgroup per CU is resident (LDS-bound, 1 wave/SIMD): size each launch to ~one wave of 256 CUs
    const int chan_blocks = (g.CApad / 64) * (g.CBpad / 64);
    int nsplit = gx_ceil_div(256, chan_blocks);
    if (nsplit > g.ntiles) nsplit = g.ntiles;
    if (nsplit < 1) nsplit = 1;
    g.nsplit = nsplit;
    g.Ttot = Ttot;
    for (int c = 0; c < 5; ++c) g.cls_begin[c] = 0;
    // (the merged kernel carries four unrolled bodies, ~140 KB of code: it thrashes the 64 KB instruction cache once
    //  the classes run long enough to matter, so it is used only where a class launch cannot fill the chip)
    if (ncls_launches == 4 && g.ntiles < 256) {
        // merged launch: hand the ~256/chan_blocks workgroup budget to the classes greedily so that the largest
        // per-workgroup cost taps_c * ceil(ntiles / splits_c) is minimised
        const int taps[4] = {9, 6, 6, 4};
        int sp[4] = {1, 1, 1, 1};
        int budget = gx_ceil_div(256, chan_blocks) - 4;
        auto cost = [&](int c, int k) { return taps[c] * gx_ceil_div(g.ntiles, k); };
        while (budget > 0) {
            int worst = 0;
            for (int c = 1; c < 4; ++c) if (cost(c, sp[c]) > cost(worst, sp[worst])) worst = c;
            if (sp[worst] >= g.ntiles) break;
            // next split count that actually lowers this class's cost
            int k = sp[worst] + 1;
            while (k < g.ntiles && cost(worst, k) == cost(worst, sp[worst])) ++k;
            if (k - sp[worst] > budget) break;
            budget -= k - sp[worst];
            sp[worst] = k;
        }
        int mx = 0;
        for (int c = 0; c < 4; ++c) { g.cls_begin[c + 1] = g.cls_begin[c] + sp[c]; mx = sp[c] > mx ? sp[c] : mx; }
        g.nsplit = mx;
        nsplit = mx;
    }
    const int CHS = G * (TH + 2 * halo) * (TW + 2 * halo);
    pl->lds_bytes = (size_t)2 * b_rows * (CHS | 1) * sizeof(float);   // double-buffered B (x halo) tile
    pl->ws_floats = (size_t)nsplit * Ttot * g.CApad * g.CBpad;
    return GX_OK;
}

// zero page for the LDS-DMA halo loads: 64 channels x (Hb*Wb <= 65536) floats, allocated at first use
constexpr size_t kZeroFloats = (size_t)64 * 65536;
const float* g_zero_page = nullptr;

const float* zero_page(hipStream_t s) {
    if (g_zero_page) return g_zero_page;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    float* p = nullptr;
    if (hipMalloc((void**)&p, kZeroFloats * sizeof(float)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, kZeroFloats * sizeof(float)) != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_zero_page = p;
    return g_zero_page;
}

template <int WM, int LTW>
void launch_wgrad_fast(dim3 grid, size_t lds_bytes, hipStream_t s, const float* a, const float* b, float* partial,
                       const float* zeros, const WgradGeom& g) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fast_kernel<WM, LTW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_fast_kernel<WM, LTW>), grid, dim3(256), lds_bytes, s, a, b, partial, zeros, g);
}

// queued small-layer jobs of each context (conv3x3, tile width 4)
struct WfPending { WfJob job; size_t lds; double flops; };
std::vector<WfPending> g_wf_ctx[kGxMaxCtx];
#define g_wf (g_wf_ctx[gx_cur_ctx()])
bool wf_defer_on() {
    static const char* env = getenv("GENESIS_WGRAD_SMALL_DEFER");
    return !(env && env[0] == '0');
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
    if (WM == W_C3 && pl.g.lTW == 2 && g_gx_defer_on && wf_defer_on() && (pl.g.Wb & 3) == 0 && pl.g.Hb * pl.g.Wb <= 65536 &&
        (double)(1 << pl.g.lG) * pl.g.CA * pl.g.Ha * pl.g.Wa < 2.0e9 && (double)(1 << pl.g.lG) * pl.g.CB * pl.g.Hb * pl.g.Wb < 2.0e9 &&
        !getenv("GENESIS_WGRAD_LEGACY") && gx_defer_wgrad_room() > 0 && zero_page(s)) {
        // a layer too small for the stream-K launch: launched with the other queued ones at the flush (gx_wf_flush).
        // Only while the reduce queue has room for this layer's record (the caller pushes it next): with a full queue the
        // reduce runs at once, so the slabs must have been written by then -- the immediate launch below
        g_wf.push_back(WfPending{WfJob{a, b, partial, pl.g}, pl.lds_bytes,
                                 2.0 * pl.g.N * (double)pl.g.CA * pl.g.CB * WTap<WM>::NT * pl.g.Hb * pl.g.Wb});
        return GX_OK;
    }
    {
        using WT = WTap<WM>;
        const WgradGeom& g = pl.g;
        const double flops = 2.0 * g.N * (double)g.CA * g.CB * WT::NT * g.Hb * g.Wb;
        const double bytes = 4.0 * ((double)g.N * g.CA * g.Hb * g.Wb + (double)g.N * g.CB * g.Hb * g.Wb +
                                    (double)g.nsplit * WT::NT * g.CApad * g.CBpad);
        GxProf pf(KID_WGRAD_C3 + WM, s, flops, bytes);
        // lean kernel: 16-byte A loads (tile width >= 4 on a grid whose width is a multiple of 4), 31-bit offsets
        const bool fast = g.lTW >= 2 && g.lTW <= 5 && (g.Wb & 3) == 0 && g.Hb < 1024 && g.Wb < 1024 &&
                          (double)(1 << g.lG) * g.CA * g.Ha * g.Wa < 2.0e9 &&
                          (double)(1 << g.lG) * g.CB * g.Hb * g.Wb < 2.0e9 && g.Hb * g.Wb <= 65536 &&
                          !getenv("GENESIS_WGRAD_LEGACY");
        const float* zeros = fast ? zero_page(s) : nullptr;
        if (fast && zeros) {
            switch (g.lTW) {
                case 2: launch_wgrad_fast<WM, 2>(grid, pl.lds_bytes, s, a, b, partial, zeros, pl.g); break;
                case 3: launch_wgrad_fast<WM, 3>(grid, pl.lds_bytes, s, a, b, partial, zeros, pl.g); break;
                case 4: launch_wgrad_fast<WM, 4>(grid, pl.lds_bytes, s, a, b, partial, zeros, pl.g); break;
                default: launch_wgrad_fast<WM, 5>(grid, pl.lds_bytes, s, a, b, partial, zeros, pl.g); break;
            }
        } else {
            hipLaunchKernelGGL(wgrad_kernel<WM>, grid, dim3(256), pl.lds_bytes, s, a, b, partial, pl.g);
        }
    }
    GX_CHECK_LAUNCH(name);
    return GX_OK;
}

int wf_flush(hipStream_t s) {
    std::vector<WfPending>& q = g_wf;
    if (q.empty()) return GX_OK;
    const float* zeros = zero_page(s);
    if (!zeros) { q.clear(); gx_set_error("wgrad (queued small layers): no zero page"); return GX_ELAUNCH; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fast_multi_kernel<W_C3, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    for (size_t i = 0; i < q.size(); i += kMaxWfJobs) {
        const int n = (int)(q.size() - i < (size_t)kMaxWfJobs ? q.size() - i : kMaxWfJobs);
        WfTable tab;
        unsigned gx = 1, gy = 1; size_t lds = 0; double flops = 0.0;
        for (int j = 0; j < n; ++j) {
            const WfPending& p = q[i + j];
            tab.job[j] = p.job;
            const unsigned by = (unsigned)((p.job.g.CApad / 64) * (p.job.g.CBpad / 64));
            gx = (unsigned)p.job.g.nsplit > gx ? (unsigned)p.job.g.nsplit : gx;
            gy = by > gy ? by : gy;
            lds = p.lds > lds ? p.lds : lds;
            flops += p.flops;
        }
        for (int j = n; j < kMaxWfJobs; ++j) tab.job[j] = tab.job[0];
        {
            GxProf pf(KID_WGRAD_C3, s, flops, 0.0);
            hipLaunchKernelGGL((wgrad_fast_multi_kernel<W_C3, 2>), dim3(gx, gy, n), dim3(256), lds, s, tab, zeros);
        }
        if (hipGetLastError() != hipSuccess) { q.clear(); gx_set_error("wgrad (queued small layers): launch failed"); return GX_ELAUNCH; }
    }
    q.clear();
    return GX_OK;
}

int launch_wgrad_deconv(const float* a, const float* b, float* partial, const WgradPlan& pl, hipStream_t s,
                        const char* name) {
    if (pl.lds_bytes > 160 * 1024) { gx_set_error("%s: LDS %zu > 160KiB", name, pl.lds_bytes); return GX_EINVAL; }
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_deconv_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const WgradGeom& g = pl.g;
    dim3 grid(g.cls_begin[4], (g.CApad / 64) * (g.CBpad / 64));
    {
        const double flops = 2.0 * g.N * (double)g.CA * g.CB * 25 * g.Hb * g.Wb;
        const double bytes = 4.0 * ((double)g.N * g.CA * g.Ha * g.Wa + (double)g.N * g.CB * g.Hb * g.Wb +
                                    (double)g.cls_begin[4] * 6.25 * g.CApad * g.CBpad);
        GxProf pf(KID_WGRAD_D00, s, flops, bytes);
        hipLaunchKernelGGL(wgrad_deconv_kernel, grid, dim3(256), pl.lds_bytes, s, a, b, partial, pl.g);
    }
    GX_CHECK_LAUNCH(name);
    return GX_OK;
}

// ---- conv3x3 FORWARD for very few input channels (Cin <= 4: the UNets' input layers, modules/unet.py:35-37 on RGB / RGB + scope) ----
// The MFMA kernels pad Cin to an 8-channel chunk and are bound by their output stores at a fraction of the HBM rate (3 -> 64 @ 64 x 64,
// N = 32: 24 us for 33.5 MB; 4 -> 32: 19.6 us for 16.8 MB).  Here a thread owns 4 consecutive pixels of a row and SC_COB output
// channels: its (Cin x 3 x 6)-float input patch sits in registers, the block's weights in LDS (broadcast 16-byte reads), every
// output row segment leaves as one 16-byte store -- exact fp32 FMAs, channel-major / tap-minor summation.
constexpr int SC_COB = 8;
template <int CIN>
__global__ void __launch_bounds__(256)
conv3x3_smallcin_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int act,
                            float* __restrict__ y, int N, int Cout, int H, int W) {
    constexpr int J = CIN * 9, JP = (J + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float wl[SC_COB][JP];
    const int co0 = blockIdx.y * SC_COB;
    for (int i = threadIdx.x; i < SC_COB * JP; i += 256) {
        const int co = i / JP, j = i - co * JP;
        wl[co][j] = (j < J && co0 + co < Cout) ? w[(size_t)(co0 + co) * J + j] : 0.f;
    }
    __syncthreads();
    const int HW = H * W, q4 = HW >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= N * q4) return;
    const int n = q / q4, rem = (q - n * q4) << 2;
    const int r = rem / W, c = rem - r * W;
    float pt[CIN][3][6];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int rr = r + kh - 1;
            const bool rok = rr >= 0 && rr < H;
            const float* xp = x + ((size_t)n * CIN + ci) * HW + (size_t)(rok ? rr : 0) * W + c;
            f32x4 m = {0.f, 0.f, 0.f, 0.f};
            float e0 = 0.f, e1 = 0.f;
            if (rok) {
                m = *reinterpret_cast<const f32x4*>(xp);
                if (c > 0) e0 = xp[-1];
                if (c + 4 < W) e1 = xp[4];
            }
            pt[ci][kh][0] = e0; pt[ci][kh][1] = m[0]; pt[ci][kh][2] = m[1]; pt[ci][kh][3] = m[2]; pt[ci][kh][4] = m[3]; pt[ci][kh][5] = e1;
        }
#pragma unroll
    for (int co = 0; co < SC_COB; ++co) {
        if (co0 + co >= Cout) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float wv[JP];
#pragma unroll
        for (int j4 = 0; j4 < JP / 4; ++j4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&wl[co][4 * j4]);
            wv[4 * j4] = t[0]; wv[4 * j4 + 1] = t[1]; wv[4 * j4 + 2] = t[2]; wv[4 * j4 + 3] = t[3];
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float wj = wv[ci * 9 + kh * 3 + kw];
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[p] = fmaf(wj, pt[ci][kh][kw + p], acc[p]);
                }
        const float bv = bias ? bias[co0 + co] : 0.f;
        f32x4 o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v = acc[p] + bv;
            if (act == 1) v = v > 0.f ? v : 0.f;
            else if (act == 2) v = v > 0.f ? v : expm1f(v);
            o[p] = v;
        }
        *reinterpret_cast<f32x4*>(y + ((size_t)n * Cout + co0 + co) * HW + rem) = o;
    }
}
static bool smallcin_fwd_ok(int Cin, int H, int W) {
    static const char* env = getenv("GENESIS_CONV3_SMALLCIN_FWD");
    // Cin == 4 only by default (MONet's [x | log-scope] input, modules/attention.py:36-40: 19.5 -> 12.2 us per pass).  The RGB
    // layer of GENESIS-V2 gains as much (22.2 -> 14.3 us, + 0.3 % of its step) but another summation order in the network's first
    // layer moved ReLU decisions in the fp64-budget and full-batch-vs-chunks tests (DESIGN.md finding 18) beyond their measured
    // allowances: GENESIS_CONV3_SMALLCIN_FWD=2 enables every Cin <= 4, =0 none
    const bool all = env && env[0] == '2';
    return !(env && env[0] == '0') && (Cin == 4 || (all && Cin >= 1 && Cin <= 4)) && (W % 4) == 0 && (long)H * W >= 1024;
}

// ---- conv3x3 weight gradient for very few input channels (Cin * 9 <= 32: the UNet's RGB input layer) -----------
// The generic kernel pads Cin to a 64-channel block (21x wasted MFMA work: 3 -> 64 @64x64 took 111 us for 0.45
// GFLOP).  Here the (ci, tap) pairs ARE the N dimension: D[co][j = ci*9 + tap] += dy[co][p] * x[ci][p + tap] on
// mfma_f32_16x16x4f32, contraction over pixels; every wave streams 64-pixel row segments (dy with 16-byte loads, the
// shifted x values with bounds-checked scalar loads: x is tiny and cache resident); per-block partials
// pw[blk][co][32] are summed by wgrad_smallcin_reduce_kernel in a fixed order.
__global__ void __launch_bounds__(256)
wgrad_smallcin_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N, int Cin, int Cout, int H,
                      int W, float* __restrict__ pw) {
    __shared__ float red[4][64][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 15, q = lane >> 4;
    const int HW = H * W;
    const int chunks_per_img = HW >> 6;
    const int nchunks = N * chunks_per_img;
    const int co0 = blockIdx.y * 64;
    const int J = Cin * 9;
    f32x4 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[c][t][0] = 0.f; acc[c][t][1] = 0.f; acc[c][t][2] = 0.f; acc[c][t][3] = 0.f; }
    // this lane's two (ci, tap) columns
    int jci[2], jdy[2], jdx[2];
    bool jok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = t * 16 + row;
        jok[t] = j < J;
        jci[t] = jok[t] ? j / 9 : 0;
        const int tap = jok[t] ? j % 9 : 4;
        jdy[t] = tap / 3 - 1; jdx[t] = tap % 3 - 1;
    }
    for (int c = blockIdx.x * 4 + wave; c < nchunks; c += gridDim.x * 4) {
        const int n = c / chunks_per_img;
        const int p0 = (c - n * chunks_per_img) * 64 + q * 16;     // 16 consecutive pixels of one image row
        const int py = p0 / W, px0 = p0 - py * W;
        f32x4 av[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int co = co0 + ct * 16 + row;
            const bool ok = co < Cout;
            const float* ap = dy + ((size_t)n * Cout + (ok ? co : 0)) * HW + p0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[ct][i] = *reinterpret_cast<const f32x4*>(ap + 4 * i);
                if (!ok) { av[ct][i][0] = 0.f; av[ct][i][1] = 0.f; av[ct][i][2] = 0.f; av[ct][i][3] = 0.f; }
            }
        }
        float bv[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int yy = py + jdy[t];
            const bool rok = jok[t] && yy >= 0 && yy < H;
            const float* bp = x + ((size_t)n * Cin + jci[t]) * HW + (size_t)(rok ? yy : 0) * W;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int xx = px0 + e + jdx[t];
                bv[t][e] = (rok && xx >= 0 && xx < W) ? bp[xx] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct][i][u], bv[t][4 * i + u], acc[ct][t],
                                                                          0, 0, 0);
    }
    // C/D layout (16x16): col = lane & 15 (j), row = (lane >> 4) * 4 + reg (co)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][ct * 16 + q * 4 + r][t * 16 + row] = acc[ct][t][r];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) {
        const int co = i >> 5, j = i & 31;
        pw[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + co) * 32 + j] =
            (red[0][co][j] + red[1][co][j]) + (red[2][co][j] + red[3][co][j]);
    }
}

// dw[co][ci][tap] = sum over blocks; one workgroup per output channel: 32 columns x 32 block groups, four loads in
// flight per thread (with 8 groups and one running sum a thread walked 64 dependent-latency loads: 20 us)
__global__ void __launch_bounds__(1024)
wgrad_smallcin_reduce_kernel(const float* __restrict__ pw, int nblk, int Cout, int J, float* __restrict__ dw, int accumulate) {
    __shared__ float red[32][32];
    const int co = blockIdx.x;
    const int j = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int mt = co >> 6, cl = co & 63;
    const float* p = pw + ((size_t)mt * nblk * 64 + cl) * 32 + j;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = grp;
    for (; b + 96 < nblk; b += 128) {
        s0 += p[(size_t)b * 2048];
        s1 += p[(size_t)(b + 32) * 2048];
        s2 += p[(size_t)(b + 64) * 2048];
        s3 += p[(size_t)(b + 96) * 2048];
    }
    for (; b < nblk; b += 32) s0 += p[(size_t)b * 2048];
    red[grp][j] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && j < J) {
        float t = red[0][j];
#pragma unroll
        for (int g2 = 1; g2 < 32; ++g2) t += red[g2][j];
        dw[(size_t)co * J + j] = accumulate ? dw[(size_t)co * J + j] + t : t;      // (deferral on: ADD, like the queued reductions)
    }
    (void)Cout;
}

inline bool smallcin_ok(int Cin, int H, int W) { return Cin * 9 <= 32 && (W % 16) == 0 && ((H * W) % 64) == 0; }
inline int smallcin_blocks(int N, int H, int W) {
    const int nchunks = N * ((H * W) >> 6);
    const int b = gx_ceil_div(nchunks, 4);
    return b > 512 ? 512 : (b < 1 ? 1 : b);     // 512 partial tables: the reduce reads 4 MB instead of 8
}

int launch_wgrad_reduce(const float* partial, float* dw, const WgradPlan& pl, int layout, hipStream_t s) {
    if (g_gx_defer_on) {
        const int* cb = pl.g.cls_begin;
        const bool merged = cb[4] > 0;
        GxWgradRed r{partial, dw, pl.g.nsplit, pl.g.Ttot, pl.g.CA, pl.g.CB, pl.g.CApad, pl.g.CBpad, layout,
                     merged ? cb[1] - cb[0] : 0, merged ? cb[2] - cb[1] : 0, merged ? cb[3] - cb[2] : 0,
                     merged ? cb[4] - cb[3] : 0};
        if (gx_defer_push_wgrad(r)) return GX_OK;
        return gx_defer_flush_wgrad(&r, 1, s);      // queue full: reduce now, ACCUMULATING like the batched flush
    }
    const int total = pl.g.Ttot * pl.g.CA * pl.g.CB;
    const int blocks = gx_ceil_div(total, 64);
    {
        GxProf pf(KID_WGRAD_REDUCE, s, 0.0, 4.0 * ((double)pl.g.nsplit + 1.0) * total);
        const int* cb = pl.g.cls_begin;
        const bool merged = cb[4] > 0;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, dw, pl.g.nsplit, pl.g.Ttot,
                           pl.g.CA, pl.g.CB, pl.g.CApad, pl.g.CBpad, layout, merged ? cb[1] - cb[0] : 0,
                           merged ? cb[2] - cb[1] : 0, merged ? cb[3] - cb[2] : 0, merged ? cb[4] - cb[3] : 0, 0, 0, 0, 0);
    }
    GX_CHECK_LAUNCH("wgrad_reduce");
    return GX_OK;
}

bool c5_fast_plan(int N, int CA, int CB, int H, int W, WgradPlan* pl) {
    const char* env = getenv("GENESIS_C5_FAST");       // 0: such layers stay on the generic implicit-GEMM kernel; 2: every size
    if (env && env[0] == '0') return false;
    // three launches + 25-tap slabs: pays from ~8 GFLOP on (measured on one box: the N = 224 decoder layer of GENESIS 514 ->
    // ~330 us, step + 1.5 %; the N = 32 layer of BaselineVAE three launches of ~20 us against one of 45, step - 2.4 %)
    if (!(env && env[0] == '2') && 50.0 * N * H * W * (double)CA * CB < 8.0e9) return false;
    if (N <= 0 || CA <= 0 || CB <= 0 || H < 4 || W < 4 || (W & 3) || H >= 1024 || W >= 1024 || H * W > 65536) return false;
    if (plan_wgrad(N, CA, CB, H, W, 1, 25, 1, pl, 128, 64, 2) != GX_OK) return false;
    const WgradGeom& g = pl->g;
    return g.lTW >= 2 && g.lTW <= 5 && pl->lds_bytes <= 160 * 1024 && (double)(1 << g.lG) * CA * H * W < 2.0e9 &&
           (double)(1 << g.lG) * CB * H * W < 2.0e9 && !getenv("GENESIS_WGRAD_LEGACY");
}
template <int WM, int LTW>
void launch_c5_fast(dim3 grid, size_t lds, hipStream_t s, const float* a, const float* b, float* partial, const float* zeros,
                           const WgradGeom& g) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fast_kernel<WM, LTW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((wgrad_fast_kernel<WM, LTW>), grid, dim3(256), lds, s, a, b, partial, zeros, g);
}
template <int WM>
void launch_c5_cls(const WgradPlan& pl, hipStream_t s, const float* a, const float* b, float* partial, const float* zeros) {
    const dim3 grid(pl.g.nsplit, (pl.g.CApad / 64) * (pl.g.CBpad / 64));
    const WgradGeom& g = pl.g;
    GxProf pf(KID_DCONV, s, 2.0 * g.N * (double)g.CA * g.CB * WTap<WM>::NT * g.Hb * g.Wb,
              4.0 * ((double)g.N * (g.CA + g.CB) * g.Hb * g.Wb + (double)g.nsplit * WTap<WM>::NT * g.CApad * g.CBpad));
    switch (g.lTW) {
        case 2: launch_c5_fast<WM, 2>(grid, pl.lds_bytes, s, a, b, partial, zeros, g); break;
        case 3: launch_c5_fast<WM, 3>(grid, pl.lds_bytes, s, a, b, partial, zeros, g); break;
        case 4: launch_c5_fast<WM, 4>(grid, pl.lds_bytes, s, a, b, partial, zeros, g); break;
        default: launch_c5_fast<WM, 5>(grid, pl.lds_bytes, s, a, b, partial, zeros, g); break;
    }
}

int check_dims(const char* name, int N, int Cin, int Cout, int H, int W) {
    GX_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0, "%s: bad N/C (%d,%d,%d)", name, N, Cin, Cout);
    GX_CHECK_ARG(H >= 1 && W >= 2 && H <= 4096 && W <= 4096 && (H * W) % 4 == 0,
                 "%s: H in [1,4096], W in [2,4096], H*W a multiple of 4 (got %dx%d)", name, H, W);
    return GX_OK;
}

}  // namespace

// queued small-layer weight-gradient launches (gx_common.h)
int gx_wf_flush(hipStream_t s) { return wf_flush(s); }
int gx_wf_pending(void) { return (int)g_wf.size(); }
void gx_wf_discard(void) { g_wf.clear(); }

// =================================================================== C ABI
int gx_defer_flush_wgrad(const GxWgradRed* items, int n, hipStream_t s) {
    // the records of one launch ACCUMULATE into their destinations from different workgroups: two records of one launch
    // must not share a destination block (a weight used several times per iteration -- MONet's recurrent UNet -- queues
    // one record per use).  Rounds: a launch takes, in queue order, the records whose destination it does not hold yet;
    // a destination therefore receives its contributions in queue order, one launch after the other (deterministic).
    std::vector<int> pending(n);
    for (int i = 0; i < n; ++i) pending[i] = i;
    while (!pending.empty()) {
        std::vector<int> later;
        WgradRedTable tab;
        int m = 0, maxblocks = 1;
        double bytes = 0.0;
        for (int idx : pending) {
            const GxWgradRed& it = items[idx];
            bool clash = m >= kRedPerLaunch;
            for (int k = 0; k < m && !clash; ++k)
                clash = tab.e[k].dw == it.dw && tab.e[k].ca0 == it.ca0 && tab.e[k].cb0 == it.cb0;
            if (clash) { later.push_back(idx); continue; }
            tab.e[m++] = it;
            const int total = it.Ttot * it.CA * it.CB;
            const int total4 = it.Ttot * it.CA * (it.CBpad >> 2);
            maxblocks = gx_ceil_div(total4, 64) > maxblocks ? gx_ceil_div(total4, 64) : maxblocks;
            bytes += 4.0 * (it.nsplit + 1.0) * total;
        }
        {
            GxProf pf(KID_WGRAD_REDUCE, s, 0.0, bytes);
            hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(maxblocks, m), dim3(256), 0, s, tab);
        }
        GX_CHECK_LAUNCH("gx_defer_flush(wgrad)");
        pending.swap(later);
    }
    return GX_OK;
}

int gx_wgrad_reduce_now(const GxWgradRed& r, hipStream_t s, int accumulate) {
    const int total = r.Ttot * r.CA * r.CB;
    {
        GxProf pf(KID_WGRAD_REDUCE, s, 0.0, 4.0 * ((double)r.nsplit + 1.0) * total);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx_ceil_div(total, 64)), dim3(256), 0, s, r.partial, r.dw, r.nsplit,
                           r.Ttot, r.CA, r.CB, r.CApad, r.CBpad, r.layout, r.ns0, r.ns1, r.ns2, r.ns3, r.ca0, r.cb0, r.CAf,
                           r.CBf, accumulate);
    }
    GX_CHECK_LAUNCH("wgrad_reduce");
    return GX_OK;
}

extern "C" {

// workspace = packed weights (+ split-K partial slabs when the plan splits the reduction)
static size_t conv3x3_pack_floats(int Cin, int Cout) {
    // 16 positions: room for the Winograd operands (gx_wino.hip) as well as the 9 taps
    // (24: the bf16-pipe Winograd operands are three 2-byte pieces per value on 16-channel chunks)
    size_t f = (size_t)24 * gx_round_up(Cin, 16) * gx_round_up(Cout, 64);
    size_t d = (size_t)24 * gx_round_up(Cout, 16) * gx_round_up(Cin, 64);
    return (f > d ? f : d) + 4096;      // (+ the slack of the bf16-piece packings, kinds 20 / 21)
}

int gx_weight_cache_create(void) {
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    for (size_t i = 0; i < g_caches.size(); ++i)
        if (!g_caches[i].alive) { g_caches[i] = PackCache(); g_caches[i].alive = true; return (int)i; }
    g_caches.emplace_back();
    g_caches.back().alive = true;
    return (int)g_caches.size() - 1;
}

static int cache_check(const char* name, int id) {
    GX_CHECK_ARG(id >= 0 && id < (int)g_caches.size() && g_caches[id].alive, "%s: bad cache id %d", name, id);
    return GX_OK;
}

int gx_weight_cache_record(int id, int on) {
    int rc = cache_check("gx_weight_cache_record", id);
    if (rc) return rc;
    PackCache& c = g_caches[id];
    if (on) { g_cache_recording = id; return GX_OK; }
    g_cache_recording = -1;
    if (c.dev) { (void)hipFree(c.dev); c.dev = nullptr; }
    if (c.dev_map) { (void)hipFree(c.dev_map); c.dev_map = nullptr; }
    c.nchunks = 0;
    if (!c.entries.empty()) {
        std::vector<int> map;
        for (size_t i = 0; i < c.entries.size(); ++i) {
            PackEntry& e = c.entries[i];
            e.chunk0 = (int)map.size();
            const int n = gx_ceil_div(e.NT * e.Kpad * e.Mpad, kPackChunk);
            map.insert(map.end(), (size_t)n, (int)i);
        }
        c.nchunks = (int)map.size();
        const size_t bytes = c.entries.size() * sizeof(PackEntry);
        if (hipMalloc((void**)&c.dev, bytes) != hipSuccess ||
            hipMemcpy(c.dev, c.entries.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMalloc((void**)&c.dev_map, map.size() * sizeof(int)) != hipSuccess ||
            hipMemcpy(c.dev_map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            gx_set_error("gx_weight_cache_record: table upload failed");
            return GX_ELAUNCH;
        }
    }
    return GX_OK;
}

int gx_weight_cache_size(int id) {
    return (id >= 0 && id < (int)g_caches.size() && g_caches[id].alive) ? (int)g_caches[id].entries.size() : -1;
}

int gx_weight_cache_refresh(int id, gx_stream_t stream) {
    int rc = cache_check("gx_weight_cache_refresh", id);
    if (rc) return rc;
    PackCache& c = g_caches[id];
    GX_CHECK_ARG(g_cache_recording != id, "gx_weight_cache_refresh: cache %d is still recording", id);
    if (c.entries.empty()) { g_cache_active = -1; return GX_OK; }
    hipStream_t s = (hipStream_t)stream;
    bool any_f16 = false;
    for (const PackEntry& e : c.entries) any_f16 = any_f16 || e.pack >= 40;
    if (any_f16) {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 0.0);
        hipLaunchKernelGGL(pack_amax_batch_kernel, dim3((unsigned)c.entries.size()), dim3(1024), 0, s, (const PackEntry*)c.dev);
    }
    {
        double bytes = 0.0;
        for (const PackEntry& e : c.entries) bytes += 8.0 * e.NT * e.Kpad * e.Mpad;
        GxProf pf(KID_PACK_WEIGHTS, s, 0.0, bytes);
        hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)c.nchunks), dim3(256), 0, s,
                           (const PackEntry*)c.dev, (const int*)c.dev_map);
    }
    GX_CHECK_LAUNCH("gx_weight_cache_refresh");
    g_cache_active = id;
    return GX_OK;
}

int gx_weight_cache_release(void) {
    g_cache_active = -1;
    return GX_OK;
}

// the cache's packings are served WITHOUT re-packing: the weights have not changed since the last refresh (the backward pass of
// an iteration whose forward refreshed them; a captured backward graph that runs behind a captured forward graph)
int gx_weight_cache_activate(int id) {
    int rc = cache_check("gx_weight_cache_activate", id);
    if (rc) return rc;
    GX_CHECK_ARG(g_cache_recording != id, "gx_weight_cache_activate: cache %d is still recording", id);
    g_cache_active = g_caches[id].entries.empty() ? -1 : id;
    return GX_OK;
}

int gx_weight_cache_destroy(int id) {
    int rc = cache_check("gx_weight_cache_destroy", id);
    if (rc) return rc;
    PackCache& c = g_caches[id];
    for (PackEntry& e : c.entries) (void)hipFree(e.wp);
    if (c.dev) (void)hipFree(c.dev);
    if (c.dev_map) (void)hipFree(c.dev_map);
    c = PackCache();
    if (g_cache_active == id) g_cache_active = -1;
    if (g_cache_recording == id) g_cache_recording = -1;
    return GX_OK;
}


size_t gx_conv3x3_ws_bytes(int N, int Cin, int Cout, int H, int W) {
    TapPlan pf, pd;
    size_t part = 0;
    if (plan_tapconv<M_C3>(N, Cin, Cout, gx_round_up(Cout, 64), H, W, H, W, H, W, 0, &pf, "ws") == GX_OK && pf.g.nsplit > 1)
        part = pf.g.nsplit * pf.out_elems;
    if (plan_tapconv<M_C3>(N, Cout, Cin, gx_round_up(Cin, 64), H, W, H, W, H, W, 0, &pd, "ws") == GX_OK && pd.g.nsplit > 1)
        part = part > pd.g.nsplit * pd.out_elems ? part : pd.g.nsplit * pd.out_elems;
    return (conv3x3_pack_floats(Cin, Cout) + part + gx_kq_amax_ws_floats()) * sizeof(float);    // (+ the fp16 x 3 form's amax scratch)
}

static int conv3x3_fwd_impl(const float* x, const float* w, const float* bias, int act, float* y, int N, int Cin,
                            int Cout, int H, int W, void* ws, size_t ws_bytes, gx_stream_t stream,
                            const float** parts_out = nullptr, int* nsplit_out = nullptr);

int gx_conv3x3_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout, int H, int W,
                   void* ws, size_t ws_bytes, gx_stream_t stream) {
    return conv3x3_fwd_impl(x, w, nullptr, 0, y, N, Cin, Cout, H, W, ws, ws_bytes, stream);
}

int gx_conv3x3_bias_act_fwd(const float* x, const float* w, const float* bias, int act, float* y, int N, int Cin,
                            int Cout, int H, int W, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(act >= 0 && act <= 2, "gx_conv3x3_bias_act_fwd: act must be 0 (none), 1 (ReLU) or 2 (ELU)");
    return conv3x3_fwd_impl(x, w, bias, act, y, N, Cin, Cout, H, W, ws, ws_bytes, stream);
}

int gx_conv3x3_fwd_parts(const float* x, const float* w, float* y, int N, int Cin, int Cout, int H, int W, void* ws,
                         size_t ws_bytes, const float** parts, int* nsplit, size_t* split_stride, gx_stream_t stream) {
    GX_CHECK_ARG(parts && nsplit && split_stride, "gx_conv3x3_fwd_parts: null out-parameter");
    *split_stride = (size_t)N * Cout * H * W;
    return conv3x3_fwd_impl(x, w, nullptr, 0, y, N, Cin, Cout, H, W, ws, ws_bytes, stream, parts, nsplit);
}

static int conv3x3_fwd_impl(const float* x, const float* w, const float* bias, int act, float* y, int N, int Cin,
                            int Cout, int H, int W, void* ws, size_t ws_bytes, gx_stream_t stream,
                            const float** parts_out, int* nsplit_out) {
    int rc = check_dims("gx_conv3x3_fwd", N, Cin, Cout, H, W);
    if (rc) return rc;
    GX_CHECK_ARG(x && w && y && ws, "gx_conv3x3_fwd: null pointer");
    GX_CHECK_ARG(ws_bytes >= gx_conv3x3_ws_bytes(N, Cin, Cout, H, W), "gx_conv3x3_fwd: workspace too small");
    const int Kpad = gx_round_up(Cin, 8), Mpad = gx_round_up(Cout, 64);
    hipStream_t s = (hipStream_t)stream;
    if (smallcin_fwd_ok(Cin, H, W)) {          // input layers: vector-ALU kernel bound by its stores
        const dim3 grid(gx_ceil_div(N * (H * W / 4), 256), gx_ceil_div(Cout, SC_COB));
        {
            GxProf pf(KID_TAPCONV_C3, s, 2.0 * N * (double)Cout * Cin * 9 * H * W,
                      4.0 * ((double)N * Cin * H * W + (double)N * Cout * H * W));
#define GX_SC_LAUNCH(C_) hipLaunchKernelGGL(conv3x3_smallcin_fwd_kernel<C_>, grid, dim3(256), 0, s, x, w, bias, act, y, N, Cout, H, W)
            if (Cin == 1) GX_SC_LAUNCH(1); else if (Cin == 2) GX_SC_LAUNCH(2); else if (Cin == 3) GX_SC_LAUNCH(3); else GX_SC_LAUNCH(4);
#undef GX_SC_LAUNCH
        }
        GX_CHECK_LAUNCH("gx_conv3x3_fwd(small Cin)");
        if (parts_out) { *parts_out = y; *nsplit_out = 1; }
        return GX_OK;
    }
    TapPlan pl;
    rc = plan_c3(N, Cin, Cout, Mpad, H, W, &pl, "gx_conv3x3_fwd");
    if (rc) return rc;
    float* wp = (float*)ws;
    float* part = wp + conv3x3_pack_floats(Cin, Cout);
    pl.g.act = act;
    const float* wpu;
    static const char* kq_env = getenv("GENESIS_KQ");
    const bool kq_first = kq_env && kq_env[0] == '2';        // benchmarking: the k-quad kernel ahead of Winograd
    const bool wino_ok = !bias && act == 0 && gx_wino_eligible(N, Cin, Cout, H, W);
    // <= 32 output channels: the bf16-pipe kernel with 32-channel workgroups, ahead of Winograd (whose 64-channel tile would be
    // half empty: MONet's UNet 64 -> 32 layer 51 -> 38 us); GENESIS_KQ_C3H_FIRST=0: only where Winograd does not apply
    static const char* c3h_first = getenv("GENESIS_KQ_C3H_FIRST");
    if ((!wino_ok || !(c3h_first && c3h_first[0] == '0')) && gx_kq_c3h_eligible(N, Cin, Cout, H, W)) {
        const int f16 = gx_kq_f16_on() ? 20 : 0;       // pack 40: two fp16 pieces of w * 2^e (gx_kq_precision(2))
        rc = launch_pack(w, wp, 20 + f16, Cout, Cin, 9, gx_round_up(Cin, 16), Mpad, s, &wpu);
        if (rc) return rc;
        float* amax_ws = f16 ? (float*)((char*)ws + gx_conv3x3_ws_bytes(N, Cin, Cout, H, W)) - gx_kq_amax_ws_floats() : nullptr;
        if (f16) { int unused_n; (void)gx_amax_link_take(nullptr, 0, &unused_n); }       // (a pending amax link is not for this call)
        rc = gx_kq_c3h_launch(x, wpu, bias, act, y, N, Cin, Cout, H, W, s, nullptr, 0, amax_ws,
                              f16 ? (const float*)((const char*)wpu + gx_kq_h_amax_off(gx_round_up(Cin, 16), Mpad, 9)) : nullptr);
        if (rc) return rc;
        if (parts_out) { *parts_out = y; *nsplit_out = 1; }
        return GX_OK;
    }
    if ((kq_first || !wino_ok) && gx_kq_c3_eligible(N, Cin, Cout, H, W)) {   // 16-byte operand reads (gx_kq.hip)
        rc = launch_pack(w, wp, 10, Cout, Cin, 9, Kpad, Mpad, s, &wpu);
        if (rc) return rc;
        rc = gx_kq_c3_launch(x, wpu, bias, act, y, N, Cin, Cout, H, W, s);
        if (rc) return rc;
        if (parts_out) { *parts_out = y; *nsplit_out = 1; }
        return GX_OK;
    }
    if (wino_ok) {   // Winograd F(2x2,3x3): 2.25x fewer MFMA passes
        // (25: three bf16 pieces; 45: two fp16 pieces -- the input's maxima were handed in, gx_conv_input_amax)
        rc = gx_wino_h_on() ? launch_pack(w, wp, gx_wino_f16_pending() ? 45 : 25, Cout, Cin, 16, gx_round_up(Cin, 16), Mpad, s, &wpu)
                            : launch_pack(w, wp, 5, Cout, Cin, 16, Kpad, Mpad, s, &wpu);
        if (rc) return rc;
        rc = gx_wino_launch(x, wpu, y, N, Cin, Cout, H, W, s);
        if (rc) return rc;
        if (parts_out) { *parts_out = y; *nsplit_out = 1; }
        return GX_OK;
    }
    rc = launch_pack(w, wp, 0, Cout, Cin, 9, Kpad, Mpad, s, &wpu);
    if (rc) return rc;
    rc = launch_tapconv<M_C3>(x, wpu, bias, pl.g.nsplit > 1 ? part : y, pl, s, "gx_conv3x3_fwd");
    if (rc) return rc;
    if (parts_out) {          // the consumer (gx_gn_relu_fwd_parts) sums the split-K slabs itself
        *parts_out = pl.g.nsplit > 1 ? part : y;
        *nsplit_out = pl.g.nsplit;
        return GX_OK;
    }
    if (pl.g.nsplit > 1) return launch_splitk_reduce(part, bias, y, pl, s);
    return GX_OK;
}

static int conv3x3_dgrad_impl(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, void* ws,
                              size_t ws_bytes, gx_stream_t stream, const float** parts_out, int* nsplit_out);

int gx_conv3x3_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W,
                     void* ws, size_t ws_bytes, gx_stream_t stream) {
    return conv3x3_dgrad_impl(dy, w, dx, N, Cin, Cout, H, W, ws, ws_bytes, stream, nullptr, nullptr);
}

/* ... without the split-K reduce launch: *parts = dx (nsplit 1) or the partial slabs inside ws, *split_stride = N Cin H W floats
 * apart; the consumer (gx_gn_relu_bwd_parts) sums them on load in slab order.  The slabs live in the caller's workspace. */
int gx_conv3x3_dgrad_parts(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, void* ws,
                           size_t ws_bytes, const float** parts, int* nsplit, size_t* split_stride, gx_stream_t stream) {
    GX_CHECK_ARG(parts && nsplit && split_stride, "gx_conv3x3_dgrad_parts: null out-parameter");
    *split_stride = (size_t)N * Cin * H * W;
    return conv3x3_dgrad_impl(dy, w, dx, N, Cin, Cout, H, W, ws, ws_bytes, stream, parts, nsplit);
}

static int conv3x3_dgrad_impl(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, void* ws,
                              size_t ws_bytes, gx_stream_t stream, const float** parts_out, int* nsplit_out) {
    if (parts_out) { *parts_out = dx; *nsplit_out = 1; }
    int rc = check_dims("gx_conv3x3_dgrad", N, Cin, Cout, H, W);
    if (rc) return rc;
    GX_CHECK_ARG(dy && w && dx && ws, "gx_conv3x3_dgrad: null pointer");
    GX_CHECK_ARG(ws_bytes >= gx_conv3x3_ws_bytes(N, Cin, Cout, H, W), "gx_conv3x3_dgrad: workspace too small");
    const int Kpad = gx_round_up(Cout, 8), Mpad = gx_round_up(Cin, 64);
    hipStream_t s = (hipStream_t)stream;
    TapPlan pl;
    rc = plan_c3(N, Cout, Cin, Mpad, H, W, &pl, "gx_conv3x3_dgrad");
    if (rc) return rc;
    float* wp = (float*)ws;
    float* part = wp + conv3x3_pack_floats(Cin, Cout);
    const float* wpu;
    static const char* kq_env = getenv("GENESIS_KQ");
    const bool kq_first = kq_env && kq_env[0] == '2';
    const bool wino_ok = gx_wino_eligible(N, Cout, Cin, H, W);
    static const char* c3h_first = getenv("GENESIS_KQ_C3H_FIRST");
    if ((!wino_ok || !(c3h_first && c3h_first[0] == '0')) && gx_kq_c3h_eligible(N, Cout, Cin, H, W)) {
        const int f16 = gx_kq_f16_on() ? 20 : 0;       // pack 41
        rc = launch_pack(w, wp, 21 + f16, Cout, Cin, 9, gx_round_up(Cout, 16), Mpad, s, &wpu);
        if (rc) return rc;
        float* amax_ws = f16 ? (float*)((char*)ws + gx_conv3x3_ws_bytes(N, Cin, Cout, H, W)) - gx_kq_amax_ws_floats() : nullptr;
        if (f16) { int unused_n; (void)gx_amax_link_take(nullptr, 0, &unused_n); }       // (a pending amax link is not for this call)
        return gx_kq_c3h_launch(dy, wpu, nullptr, 0, dx, N, Cout, Cin, H, W, s, nullptr, 0, amax_ws,
                                f16 ? (const float*)((const char*)wpu + gx_kq_h_amax_off(gx_round_up(Cout, 16), Mpad, 9)) : nullptr);
    }
    if ((kq_first || !wino_ok) && gx_kq_c3_eligible(N, Cout, Cin, H, W)) {
        rc = launch_pack(w, wp, 11, Cout, Cin, 9, Kpad, Mpad, s, &wpu);
        if (rc) return rc;
        return gx_kq_c3_launch(dy, wpu, nullptr, 0, dx, N, Cout, Cin, H, W, s);
    }
    if (wino_ok) {
        rc = gx_wino_h_on() ? launch_pack(w, wp, gx_wino_f16_pending() ? 46 : 26, Cout, Cin, 16, gx_round_up(Cout, 16), Mpad, s, &wpu)
                            : launch_pack(w, wp, 6, Cout, Cin, 16, Kpad, Mpad, s, &wpu);
        if (rc) return rc;
        return gx_wino_launch(dy, wpu, dx, N, Cout, Cin, H, W, s);
    }
    rc = launch_pack(w, wp, 1, Cout, Cin, 9, Kpad, Mpad, s, &wpu);
    if (rc) return rc;
    rc = launch_tapconv<M_C3>(dy, wpu, nullptr, pl.g.nsplit > 1 ? part : dx, pl, s, "gx_conv3x3_dgrad");
    if (rc) return rc;
    if (parts_out && pl.g.nsplit > 1) {       // the consumer sums the split-K slabs itself
        *parts_out = part; *nsplit_out = pl.g.nsplit;
        return GX_OK;
    }
    if (pl.g.nsplit > 1) return launch_splitk_reduce(part, nullptr, dx, pl, s);
    return GX_OK;
}

/* The data gradient of a conv3x3 whose INPUT is a bias + activation layer's output xout: the activation's backward in the
 * epilogue of the bf16-pipe kernel (one more read of xout there instead of a pass that reads da and xout and writes dxa),
 * then the bias gradient as channel sums of dxa. */
int gx_conv3x3_dgrad_act_supported(int N, int Cin, int Cout, int H, int W) {
    static const char* env = getenv("GENESIS_DGRAD_ACT_FUSE");
    if (env && env[0] == '0') return 0;
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    return gx_kq_c3h_eligible(N, Cout, Cin, H, W) ? 1 : 0;
}
size_t gx_conv3x3_dgrad_act_ws_bytes(int N, int Cin, int Cout, int H, int W) {
    return gx_round_up((long)gx_conv3x3_ws_bytes(N, Cin, Cout, H, W), 256) + (size_t)N * Cin * sizeof(float);
}
int gx_conv3x3_dgrad_act(const float* dy, const float* w, const float* xout, int act, float* dxa, float* dbias, int N,
                         int Cin, int Cout, int H, int W, void* ws, size_t ws_bytes, gx_stream_t stream) {
    int rc = check_dims("gx_conv3x3_dgrad_act", N, Cin, Cout, H, W);
    if (rc) return rc;
    GX_CHECK_ARG(dy && w && xout && dxa && ws, "gx_conv3x3_dgrad_act: null pointer");
    GX_CHECK_ARG(act >= 0 && act <= 2, "gx_conv3x3_dgrad_act: bad act");
    GX_CHECK_ARG(ws_bytes >= gx_conv3x3_dgrad_act_ws_bytes(N, Cin, Cout, H, W), "gx_conv3x3_dgrad_act: workspace too small");
    GX_CHECK_ARG(gx_conv3x3_dgrad_act_supported(N, Cin, Cout, H, W), "gx_conv3x3_dgrad_act: shape not supported (gx_conv3x3_dgrad_act_supported)");
    hipStream_t s = (hipStream_t)stream;
    const float* wpu;
    const int f16 = gx_kq_f16_on() ? 20 : 0;           // pack 41
    rc = launch_pack(w, (float*)ws, 21 + f16, Cout, Cin, 9, gx_round_up(Cout, 16), gx_round_up(Cin, 64), s, &wpu);
    if (rc) return rc;
    float* amax_ws = f16 ? (float*)((char*)ws + gx_conv3x3_ws_bytes(N, Cin, Cout, H, W)) - gx_kq_amax_ws_floats() : nullptr;
    if (f16) { int unused_n; (void)gx_amax_link_take(nullptr, 0, &unused_n); }       // (a pending amax link is not for this call)
    rc = gx_kq_c3h_launch(dy, wpu, nullptr, 0, dxa, N, Cout, Cin, H, W, s, xout, act, amax_ws,
                          f16 ? (const float*)((const char*)wpu + gx_kq_h_amax_off(gx_round_up(Cout, 16), gx_round_up(Cin, 64), 9)) : nullptr);
    if (rc || !dbias) return rc;
    float* part = (float*)((char*)ws + gx_round_up((long)gx_conv3x3_ws_bytes(N, Cin, Cout, H, W), 256));
    return gx_chan_sums_launch(dxa, N, Cin, H * W, part, dbias, s);
}

size_t gx_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W) {
    WgradPlan pl;
    plan_wgrad(N, Cout, Cin, H, W, 1, 9, 1, &pl);
    size_t f = pl.ws_floats;
    if (smallcin_ok(Cin, H, W)) {
        const size_t f2 = (size_t)smallcin_blocks(N, H, W) * gx_ceil_div(Cout, 64) * 64 * 32;
        f = f2 > f ? f2 : f;
    }
    return f * sizeof(float);
}

int gx_conv3x3_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W,
                     void* ws, size_t ws_bytes, gx_stream_t stream) {
    int rc = check_dims("gx_conv3x3_wgrad", N, Cin, Cout, H, W);
    if (rc) return rc;
    GX_CHECK_ARG(x && dy && dw && ws, "gx_conv3x3_wgrad: null pointer");
    WgradPlan pl;
    plan_wgrad(N, Cout, Cin, H, W, 1, 9, 1, &pl);
    GX_CHECK_ARG(ws_bytes >= gx_conv3x3_wgrad_ws_bytes(N, Cin, Cout, H, W), "gx_conv3x3_wgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (smallcin_ok(Cin, H, W) && !getenv("GENESIS_WGRAD_LEGACY")) {
        const int nblk = smallcin_blocks(N, H, W), mt = gx_ceil_div(Cout, 64);
        {
            GxProf pf(KID_WGRAD_C3, s, 2.0 * N * (double)Cout * Cin * 9 * H * W,
                      4.0 * ((double)N * Cout * H * W + (double)N * Cin * H * W));
            hipLaunchKernelGGL(wgrad_smallcin_kernel, dim3(nblk, mt), dim3(256), 0, s, x, dy, N, Cin, Cout, H, W,
                               (float*)ws);
        }
        GX_CHECK_LAUNCH("gx_conv3x3_wgrad(small Cin)");
        {
            GxProf pf(KID_WGRAD_REDUCE, s, 0.0, 4.0 * nblk * mt * 64 * 32);
            // while deferral is on every weight-gradient call ADDS to its (zeroed) destination -- a shared parameter's
            // second use must not overwrite the first (include/genesis_hip.h, gx_defer_enable)
            hipLaunchKernelGGL(wgrad_smallcin_reduce_kernel, dim3(Cout), dim3(1024), 0, s, (const float*)ws, nblk, Cout,
                               Cin * 9, dw, g_gx_defer_on ? 1 : 0);
        }
        GX_CHECK_LAUNCH("gx_conv3x3_wgrad(small Cin reduce)");
        return GX_OK;
    }
    if (gx_wgq_c3_eligible(N, Cin, Cout, H, W))      // LDS-DMA staged, grouped with the other layers when deferred
        return gx_wgq_c3(x, dy, dw, N, Cin, Cout, H, W, (float*)ws,
                         (int)(ws_bytes / sizeof(float) / ((size_t)9 * pl.g.CApad * pl.g.CBpad)), s);
    rc = launch_wgrad<W_C3>(dy, x, (float*)ws, pl, s, "gx_conv3x3_wgrad");
    if (rc) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, pl, 0, s);
}

/* conv3x3 weight gradient of a layer with C <= 32 channels on both sides and N % 4 == 0 images (the BroadcastDecoder's 32 -> 32
 * convs on the canvas, modules/decoders.py:21-35): four images per workgroup tile, one per wave (wgrad_fast_kernel QUAD). */
static bool wgrad_quad_plan(int N, int C, int H, int W, WgradPlan* pl) {
    if (N <= 0 || (N & 3) || C != 32 || H < 1 || W < 4 || (W & 3) || H >= 1024 || W >= 1024 || H * W > 65536) return false;
    if (plan_wgrad(N / 4, 64, 64, H, W, 1, 9, 1, pl, 64, 128) != GX_OK) return false;
    const WgradGeom& g = pl->g;
    if (g.lTW < 2 || g.lTW > 5 || pl->lds_bytes > 160 * 1024) return false;
    if ((double)(1 << g.lG) * 128 * H * W >= 2.0e9 || (size_t)128 * H * W > kZeroFloats) return false;
    return !getenv("GENESIS_WGRAD_LEGACY") && !getenv("GENESIS_WGRAD_NOQUAD");
}
int gx_conv3x3_wgrad_quad_supported(int N, int C, int H, int W) {
    WgradPlan pl;
    return wgrad_quad_plan(N, C, H, W, &pl) ? 1 : 0;
}
size_t gx_conv3x3_wgrad_quad_ws_bytes(int N, int C, int H, int W) {
    WgradPlan pl;
    if (!wgrad_quad_plan(N, C, H, W, &pl)) return 0;
    size_t fl = pl.ws_floats;
    if (gx_wstrip_supported(N, C, H, W) && gx_wstrip_ws_floats(N, C, H, W) > fl) fl = gx_wstrip_ws_floats(N, C, H, W);
    return fl * sizeof(float);
}
static int wgrad_quad_impl(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int H, int W, void* ws,
                           size_t ws_bytes, gx_stream_t stream);
int gx_conv3x3_wgrad_quad(const float* x, const float* dy, float* dw, int N, int C, int H, int W, void* ws, size_t ws_bytes,
                          gx_stream_t stream) {
    return wgrad_quad_impl(x, dy, dw, nullptr, N, C, H, W, ws, ws_bytes, stream);
}
/* ... and dbias [C] = sum_{n,hw} dy, the layer's bias gradient, from the same read of dy (bf16-pipe tiles; otherwise a plane-sum
 * pass over dy) */
size_t gx_conv3x3_wgrad_quad_bias_ws_bytes(int N, int C, int H, int W) {
    WgradPlan pl;
    if (!wgrad_quad_plan(N, C, H, W, &pl)) return 0;
    const size_t rec = (size_t)pl.g.nsplit * 256, planes = (size_t)N * C;
    size_t fl = pl.ws_floats + (rec > planes ? rec : planes);
    if (gx_wstrip_supported(N, C, H, W) && gx_wstrip_ws_floats(N, C, H, W) > fl) fl = gx_wstrip_ws_floats(N, C, H, W);
    return fl * sizeof(float);
}
int gx_conv3x3_wgrad_quad_bias(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int H, int W, void* ws,
                               size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(dbias, "gx_conv3x3_wgrad_quad_bias: null pointer");
    GX_CHECK_ARG(ws_bytes >= gx_conv3x3_wgrad_quad_bias_ws_bytes(N, C, H, W) && gx_conv3x3_wgrad_quad_bias_ws_bytes(N, C, H, W) > 0,
                 "gx_conv3x3_wgrad_quad_bias: unsupported shape or workspace too small");
    return wgrad_quad_impl(x, dy, dw, dbias, N, C, H, W, ws, ws_bytes, stream);
}
static int wgrad_quad_impl(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int H, int W, void* ws,
                           size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(x && dy && dw && ws, "gx_conv3x3_wgrad_quad: null pointer");
    WgradPlan pl;
    GX_CHECK_ARG(wgrad_quad_plan(N, C, H, W, &pl), "gx_conv3x3_wgrad_quad: needs 32 channels, N %% 4 == 0, W %% 4 == 0");
    GX_CHECK_ARG(ws_bytes >= pl.ws_floats * sizeof(float), "gx_conv3x3_wgrad_quad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    // grids that are no power of two (the 72 x 72 canvas): column strips straight from global memory (gx_wstrip.hip)
    if (!gx_is_pow2(W) && gx_wstrip_supported(N, C, H, W) && ws_bytes >= gx_wstrip_ws_floats(N, C, H, W) * sizeof(float))
        return gx_wstrip_launch(x, dy, dw, dbias, N, C, H, W, ws, s);
    const float* zeros = zero_page(s);
    if (!zeros) { gx_set_error("gx_conv3x3_wgrad_quad: no zero page (first call inside a stream capture)"); return GX_ELAUNCH; }
    WgradGeom g = pl.g;
    g.CA = g.CB = 128;                     // the [N / 4, 128, H, W] view: image and channel strides of the kernel
    const dim3 grid(g.nsplit, 1);
    bool bias_in_kernel = false;
    {
        GxProf pf(KID_WGRAD_C3, s, 2.0 * N * (double)C * C * 9 * H * W,
                  4.0 * (2.0 * N * C * H * W + (double)g.nsplit * 9 * 64 * 64));
#define GX_QUAD_LAUNCH(LTW_, B6_)                                                                                        \
        {                                                                                                                \
            static bool attr = false;                                                                                    \
            if (!attr) {                                                                                                 \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fast_kernel<W_C3, LTW_, true, B6_>),      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                       \
                attr = true;                                                                                             \
            }                                                                                                            \
            hipLaunchKernelGGL((wgrad_fast_kernel<W_C3, LTW_, true, B6_>), grid, dim3(256), pl.lds_bytes, s, dy, x,      \
                               (float*)ws, zeros, g);                                                                    \
        }
        // tiles >= 8 pixels wide: on the bf16 matrix pipe (six piece products) unless gx_wgq_precision(0) /
        // GENESIS_WGQ_BF16X6=0 / GENESIS_WGRAD_QUAD_B6=0 keep the weight gradients on the fp32 pipe
        static const char* b6env = getenv("GENESIS_WGRAD_QUAD_B6");
        const bool b6 = gx_wgq_bf16_pipe() && !(b6env && b6env[0] == '0');
        bias_in_kernel = dbias && b6 && g.lTW >= 3;
        if (bias_in_kernel) g.bias_part = (float*)ws + pl.ws_floats;
        switch (g.lTW) {
            case 2: GX_QUAD_LAUNCH(2, false) break;
            case 3: if (b6) GX_QUAD_LAUNCH(3, true) else GX_QUAD_LAUNCH(3, false) break;
            case 4: if (b6) GX_QUAD_LAUNCH(4, true) else GX_QUAD_LAUNCH(4, false) break;
            default: if (b6) GX_QUAD_LAUNCH(5, true) else GX_QUAD_LAUNCH(5, false) break;
        }
#undef GX_QUAD_LAUNCH
    }
    GX_CHECK_LAUNCH("gx_conv3x3_wgrad_quad");
    {
        const int total = 9 * C * C;
        GxProf pf(KID_WGRAD_REDUCE, s, 0.0, 4.0 * (4.0 * g.nsplit + 1.0) * total);
        hipLaunchKernelGGL(wgrad_quad_reduce_kernel, dim3(gx_ceil_div(total, 64)), dim3(1024), 0, s, (const float*)ws, dw,
                           g.nsplit, 9, C);
    }
    GX_CHECK_LAUNCH("gx_conv3x3_wgrad_quad(reduce)");
    if (dbias && bias_in_kernel) {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * g.nsplit * 256);
        hipLaunchKernelGGL(quad_bias_reduce_kernel, dim3(C), dim3(256), 0, s, (const float*)g.bias_part, g.nsplit, dbias);
        GX_CHECK_LAUNCH("gx_conv3x3_wgrad_quad_bias(reduce)");
    } else if (dbias) {
        return gx_chan_sums_launch(dy, N, C, H * W, (float*)ws + pl.ws_floats, dbias, s);
    }
    return GX_OK;
}

/* 5 x 5 stride-1 pad-2 convolution on the tap-conv MFMA kernel (mode M_C5): out [N,M,H,W] from in [N,K,H,W].
 *   flip 0: out[m] = sum_k in[k] (cross-correlated with) w[m][k]   -- w [M][K][5][5]: Conv2d forward; the data gradient of a
 *           stride-1 ConvTranspose2d (w = its weight [Cin = M][Cout = K])
 *   flip 1: out[m] = sum_k in[k] (convolved with) w[k][m]          -- w [K][M][5][5]: Conv2d data gradient (in = dy, w [Cout][Cin]);
 *           stride-1 ConvTranspose2d forward (w [Cin = K][Cout = M]) */
size_t gx_conv5x5s1_ws_bytes(int N, int K, int M, int H, int W) {
    TapPlan pl;
    size_t part = 0;
    if (plan_tapconv<M_C5>(N, K, M, gx_round_up(M, 64), H, W, H, W, H, W, 0, &pl, "ws") == GX_OK && pl.g.nsplit > 1)
        part = pl.g.nsplit * pl.out_elems;
    size_t b = ((size_t)25 * gx_round_up(K, 8) * gx_round_up(M, 64) + part) * sizeof(float);
    const size_t bh = gx_kq_deconv_h_pack_bytes(gx_round_up(K, 16), gx_round_up(M, 64), 25);     // the bf16-pipe packing (kinds 27 / 28)
    return (b > bh ? b : bh) + gx_kq_amax_ws_floats() * sizeof(float);       // (+ the fp16 x 3 form's amax scratch, at the end)
}

int gx_conv5x5s1_supported(int N, int K, int M, int H, int W) {
    TapPlan pl;
    return N > 0 && K > 0 && M > 0 && H >= 4 && W >= 4 &&
           plan_tapconv<M_C5>(N, K, M, gx_round_up(M, 64), H, W, H, W, H, W, 0, &pl, "gx_conv5x5s1") == GX_OK;
}

int gx_conv5x5s1(const float* in, const float* w, float* out, int N, int K, int M, int H, int W, int flip, void* ws,
                 size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(in && w && out && ws, "gx_conv5x5s1: null pointer");
    GX_CHECK_ARG(flip == 0 || flip == 1, "gx_conv5x5s1: flip must be 0 or 1");
    GX_CHECK_ARG(ws_bytes >= gx_conv5x5s1_ws_bytes(N, K, M, H, W), "gx_conv5x5s1: workspace too small");
    const int Kpad = gx_round_up(K, 8), Mpad = gx_round_up(M, 64);
    hipStream_t s = (hipStream_t)stream;
    TapPlan pl;
    int rc = plan_tapconv<M_C5>(N, K, M, Mpad, H, W, H, W, H, W, 0, &pl, "gx_conv5x5s1");
    if (rc) return rc;
    float* wp = (float*)ws;
    float* part = wp + (size_t)25 * Kpad * Mpad;
    const float* wpu;
    if (gx_kq_c5h_eligible(N, K, M, H, W)) {       // chip-filling layers: on the bf16 matrix pipe (gx_kq.hip Q_C5H)
        const int f16 = gx_kq_f16_on() ? 20 : 0;       // packs 47 / 48
        rc = flip ? launch_pack(w, wp, 28 + f16, K, M, 25, gx_round_up(K, 16), Mpad, s, &wpu)
                  : launch_pack(w, wp, 27 + f16, M, K, 25, gx_round_up(K, 16), Mpad, s, &wpu);
        if (rc) return rc;
        float* amax_ws = f16 ? (float*)((char*)ws + gx_conv5x5s1_ws_bytes(N, K, M, H, W)) - gx_kq_amax_ws_floats() : nullptr;
        int xn = 0;
        const float* xparts = f16 ? gx_amax_link_take(in, (size_t)N * K * H * W, &xn) : nullptr;      // the input's partial maxima from the kernel that wrote it (the gated unit)
        return gx_kq_c5h_launch(in, wpu, out, N, K, M, H, W, s, amax_ws,
                                f16 ? (const float*)((const char*)wpu + gx_kq_h_amax_off(gx_round_up(K, 16), Mpad, 25)) : nullptr,
                                xparts, xn);
    }
    // pack 7: w [M][K]; pack 8: w [K][M] flipped (launch_pack's (Co, Ci) are the weight tensor's leading dimensions)
    rc = flip ? launch_pack(w, wp, 8, K, M, 25, Kpad, Mpad, s, &wpu) : launch_pack(w, wp, 7, M, K, 25, Kpad, Mpad, s, &wpu);
    if (rc) return rc;
    rc = launch_tapconv<M_C5>(in, wpu, nullptr, pl.g.nsplit > 1 ? part : out, pl, s, "gx_conv5x5s1");
    if (rc) return rc;
    if (pl.g.nsplit > 1) return launch_splitk_reduce(part, nullptr, out, pl, s);
    return GX_OK;
}

/* 5 x 5 stride-1 pad-2 weight gradient on the row-ring tiles of the bf16 pipe (gx_wgq.hip): dw [CA][CB][5][5] =
 * sum_{n,p} a[n][CA][p] * b[n][CB][p + (kh - 2, kw - 2)]; Conv2d: a = dy, b = x (dw [Cout][Cin]); ConvTranspose2d stride 1:
 * a = x, b = dy (dw [Cin][Cout]). */
// ... and, where a layer's rows are shorter than the row-ring tiles' 32 pixels (the gated stacks' 16 x 16 layers), on the lean
// fp32-pipe weight-gradient kernel with a 2-pixel halo: kernel rows 0-1, 2-3 and 4 as three launches into one slab region
int gx_conv5x5_wgrad_supported(int N, int CA, int CB, int H, int W) {
    WgradPlan pl;
    return (gx_wgq_c5_eligible(N, CA, CB, H, W) || c5_fast_plan(N, CA, CB, H, W, &pl)) ? 1 : 0;
}

size_t gx_conv5x5_wgrad_ws_bytes(int N, int CA, int CB, int H, int W) {
    size_t f = (size_t)gx_wgq_max_split(CA, CB) * 25 * gx_round_up(CA, 64) * gx_round_up(CB, 64);
    WgradPlan pl;
    if (!gx_wgq_c5_eligible(N, CA, CB, H, W) && c5_fast_plan(N, CA, CB, H, W, &pl)) f = pl.ws_floats > f ? pl.ws_floats : f;
    return f * sizeof(float);
}

int gx_conv5x5_wgrad(const float* a, const float* b, float* dw, int N, int CA, int CB, int H, int W, void* ws,
                     size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(a && b && dw && ws, "gx_conv5x5_wgrad: null pointer");
    GX_CHECK_ARG(ws_bytes >= gx_conv5x5_wgrad_ws_bytes(N, CA, CB, H, W), "gx_conv5x5_wgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (gx_wgq_c5_eligible(N, CA, CB, H, W)) {
        const size_t slab = (size_t)25 * gx_round_up(CA, 64) * gx_round_up(CB, 64) * sizeof(float);
        return gx_wgq_c5(a, b, dw, N, CA, CB, H, W, (float*)ws, (int)(ws_bytes / slab), s);
    }
    WgradPlan pl;
    GX_CHECK_ARG(c5_fast_plan(N, CA, CB, H, W, &pl), "gx_conv5x5_wgrad: shape not supported (W %% 4 == 0, H, W >= 4)");
    const float* zeros = zero_page(s);
    if (!zeros) { gx_set_error("gx_conv5x5_wgrad: no zero page (first call inside a stream capture)"); return GX_ELAUNCH; }
    launch_c5_cls<W_C5A>(pl, s, a, b, (float*)ws, zeros);
    GX_CHECK_LAUNCH("gx_conv5x5_wgrad(rows 0-1)");
    launch_c5_cls<W_C5B>(pl, s, a, b, (float*)ws, zeros);
    GX_CHECK_LAUNCH("gx_conv5x5_wgrad(rows 2-3)");
    launch_c5_cls<W_C5C>(pl, s, a, b, (float*)ws, zeros);
    GX_CHECK_LAUNCH("gx_conv5x5_wgrad(row 4)");
    return launch_wgrad_reduce((const float*)ws, dw, pl, 0, s);
}

static size_t deconv_pack_floats(int Cin, int Cout) {
    // fwd: two row-parity packs (15 + 10 taps, k=Cin, m=Cout); dgrad: 25 taps (k=Cout, m=Cin)
    size_t f = (size_t)25 * gx_round_up(Cin, 8) * gx_round_up(Cout, 64);
    size_t d = (size_t)25 * gx_round_up(Cout, 8) * gx_round_up(Cin, 64);
    // (bf16-pipe forward, packs 22 / 23: three bf16 pieces per weight = 1.5 x, + the slack of whole-phase copies)
    const size_t h = (gx_kq_deconv_h_pack_bytes(gx_round_up(Cin, 16), Cout, 15) + gx_kq_deconv_h_pack_bytes(gx_round_up(Cin, 16), Cout, 10)) / 4;
    f = f > h ? f : h;
    const size_t dh = gx_kq_deconv_h_pack_bytes(gx_round_up(Cout, 16), Cin, 25) / 4;
    d = d > dh ? d : dh;
    return f > d ? f : d;
}

size_t gx_deconv5x5s2_ws_bytes(int N, int Cin, int Cout, int Hin, int Win) {
    TapPlan pf, pd;
    size_t part = 0;
    if (plan_tapconv<M_DT0>(N, Cin, Cout, gx_round_up(Cout, 64), Hin, Win, Hin, Win, 2 * Hin, 2 * Win, 0, &pf, "ws") == GX_OK &&
        pf.g.nsplit > 1)
        part = pf.g.nsplit * pf.out_elems;
    if (plan_tapconv<M_DG>(N, Cout, Cin, gx_round_up(Cin, 64), Hin, Win, 2 * Hin, 2 * Win, Hin, Win, 0, &pd, "ws") == GX_OK &&
        pd.g.nsplit > 1)
        part = part > pd.g.nsplit * pd.out_elems ? part : pd.g.nsplit * pd.out_elems;
    // (+ the amax scratch of the fp16 x 3 form, at the very end: gx_kq_amax_launch)
    return (deconv_pack_floats(Cin, Cout) + part + gx_kq_amax_ws_floats()) * sizeof(float);
}

static int deconv_fwd_impl(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout,
                           int Hin, int Win, void* ws, size_t ws_bytes, gx_stream_t stream, const float** parts_out,
                           int* nsplit_out, float* stats = nullptr, int* stats_parts_out = nullptr);

int gx_deconv5x5s2_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout,
                       int Hin, int Win, void* ws, size_t ws_bytes, gx_stream_t stream) {
    return deconv_fwd_impl(x, w, bias, y, N, Cin, Cout, Hin, Win, ws, ws_bytes, stream, nullptr, nullptr);
}

int gx_deconv5x5s2_fwd_parts(const float* x, const float* w, float* y, int N, int Cin, int Cout, int Hin, int Win,
                             void* ws, size_t ws_bytes, const float** parts, int* nsplit, size_t* split_stride,
                             gx_stream_t stream) {
    GX_CHECK_ARG(parts && nsplit && split_stride, "gx_deconv5x5s2_fwd_parts: null out-parameter");
    *split_stride = (size_t)N * Cout * 4 * Hin * Win;
    return deconv_fwd_impl(x, w, nullptr, y, N, Cin, Cout, Hin, Win, ws, ws_bytes, stream, parts, nsplit);
}

// upper bound of the epilogue-statistics scratch: one (sum, sumsq) pair per 8-channel block per 128-pixel tile per parity
static size_t deconv_stats_floats(int N, int Cout, int Hin, int Win) {
    return (size_t)N * (size_t)(gx_ceil_div(Hin * Win, 128) * 2 + 2) * (size_t)gx_ceil_div(Cout, 8) * 2;
}

size_t gx_deconv5x5s2_gn_stats_ws_bytes(int N, int Cin, int Cout, int Hin, int Win) {
    return gx_deconv5x5s2_ws_bytes(N, Cin, Cout, Hin, Win) + deconv_stats_floats(N, Cout, Hin, Win) * sizeof(float);
}

int gx_deconv5x5s2_gn_stats_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout,
                                int Hin, int Win, int groups, float eps, float* mean, float* rstd, int* fused,
                                void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(mean && rstd && ws, "gx_deconv5x5s2_gn_stats_fwd: null pointer");
    GX_CHECK_ARG(groups > 0 && Cout % groups == 0, "gx_deconv5x5s2_gn_stats_fwd: Cout %% groups != 0");
    GX_CHECK_ARG(ws_bytes >= gx_deconv5x5s2_gn_stats_ws_bytes(N, Cin, Cout, Hin, Win),
                 "gx_deconv5x5s2_gn_stats_fwd: workspace too small");
    const size_t conv_ws = gx_deconv5x5s2_ws_bytes(N, Cin, Cout, Hin, Win);
    float* stats = (float*)((char*)ws + conv_ws);
    const int cpg = Cout / groups;
    int parts = 0;
    int rc = deconv_fwd_impl(x, w, bias, y, N, Cin, Cout, Hin, Win, ws, conv_ws, stream, nullptr, nullptr,
                             (cpg % 8) == 0 ? stats : nullptr, &parts);
    if (rc) return rc;
    if (fused) *fused = parts > 0;
    if (!parts) return GX_OK;          // the caller runs the stand-alone statistics pass (gx_gn_relu_fwd, dst0 = NULL)
    {
        GxProf pf(KID_SMALL_REDUCE, (hipStream_t)stream, 0.0, 8.0 * N * parts * (Cout / 8));
        hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(gx_ceil_div(N * groups, 64)), dim3(64), 0,
                           (hipStream_t)stream, (const float*)stats, N, groups, parts, Cout / 8, cpg / 8,
                           (double)cpg * 4.0 * Hin * Win, eps, mean, rstd);
    }
    GX_CHECK_LAUNCH("gx_deconv5x5s2_gn_stats_fwd(finalize)");
    return GX_OK;
}

static int deconv_fwd_impl(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout,
                           int Hin, int Win, void* ws, size_t ws_bytes, gx_stream_t stream, const float** parts_out,
                           int* nsplit_out, float* stats, int* stats_parts_out) {
    int rc = check_dims("gx_deconv5x5s2_fwd", N, Cin, Cout, Hin, Win);
    if (rc) return rc;
    GX_CHECK_ARG(x && w && y && ws, "gx_deconv5x5s2_fwd: null pointer");
    GX_CHECK_ARG(ws_bytes >= gx_deconv5x5s2_ws_bytes(N, Cin, Cout, Hin, Win), "gx_deconv5x5s2_fwd: workspace too small");
    const int Kpad = gx_round_up(Cin, 8), Mpad = gx_round_up(Cout, 64);
    hipStream_t s = (hipStream_t)stream;
    float* wp0 = (float*)ws;
    float* wp1 = wp0 + (size_t)15 * Kpad * Mpad;
    float* part = wp0 + deconv_pack_floats(Cin, Cout);
    const float *wpu0, *wpu1;
    if (stats_parts_out) *stats_parts_out = 0;
    if (gx_kq_deconv_h_eligible(N, Cin, Cout, Hin, Win)) {     // ... on the bf16 matrix pipe (fp32 products from bf16 pieces)
        float* wh1 = wp0 + gx_kq_deconv_h_pack_bytes(Cin, Cout, 15) / 4;
        const int f16 = gx_kq_f16_on() ? 20 : 0;       // packs 42 / 43: two fp16 pieces of w * 2^e (gx_kq_precision(2))
        rc = launch_pack(w, wp0, 22 + f16, Cout, Cin, 15, Cin, Mpad, s, &wpu0);
        if (rc) return rc;
        rc = launch_pack(w, wh1, 23 + f16, Cout, Cin, 10, Cin, Mpad, s, &wpu1);
        if (rc) return rc;
        float* amax_ws = f16 ? (float*)((char*)ws + gx_deconv5x5s2_ws_bytes(N, Cin, Cout, Hin, Win)) - gx_kq_amax_ws_floats() : nullptr;
        int xn = 0;
        const float* xparts = f16 ? gx_amax_link_take(x, (size_t)N * Cin * Hin * Win, &xn) : nullptr;      // x's partial maxima from the kernel that wrote it
        rc = gx_kq_deconv_fwd_h_launch(x, wpu0, wpu1, bias, y, N, Cin, Cout, Hin, Win, stats, stats_parts_out, s, amax_ws,
                                       f16 ? (const float*)((const char*)wpu0 + gx_kq_h_amax_off(Cin, Mpad, 15)) : nullptr,
                                       xparts, xn);
        if (rc) return rc;
        if (parts_out) { *parts_out = y; *nsplit_out = 1; }
        return GX_OK;
    }
    if (gx_kq_deconv_eligible(N, Cin, Cout, Hin, Win, 2)) {   // chip-filling layers: 16-byte operand reads (gx_kq.hip)
        rc = launch_pack(w, wp0, 12, Cout, Cin, 15, Kpad, Mpad, s, &wpu0);
        if (rc) return rc;
        rc = launch_pack(w, wp1, 13, Cout, Cin, 10, Kpad, Mpad, s, &wpu1);
        if (rc) return rc;
        rc = gx_kq_deconv_fwd_launch(x, wpu0, wpu1, bias, y, N, Cin, Cout, Hin, Win, stats, stats_parts_out, s);
        if (rc) return rc;
        if (parts_out) { *parts_out = y; *nsplit_out = 1; }
        return GX_OK;
    }
    rc = launch_pack(w, wp0, 2, Cout, Cin, 15, Kpad, Mpad, s, &wpu0);
    if (rc) return rc;
    rc = launch_pack(w, wp1, 3, Cout, Cin, 10, Kpad, Mpad, s, &wpu1);
    if (rc) return rc;
    // both row parities in one launch (they share the pixel tiling and the channel split, so one reduce
    // finishes the layer)
    TapPlan p0;
    rc = plan_tapconv<M_DT0>(N, Cin, Cout, Mpad, Hin, Win, Hin, Win, 2 * Hin, 2 * Win, 0, &p0, "gx_deconv5x5s2_fwd", 2);
    if (rc) return rc;
    static const char* mw_env = getenv("GENESIS_TAPCONV_MW2");
    if (p0.g.nsplit > 1 && !(mw_env && mw_env[0] == '0')) {   // 128-pixel tiles for under-filled grids (see plan_c3)
        TapPlan p2;
        if (plan_tapconv<M_DT0>(N, Cin, Cout, Mpad, Hin, Win, Hin, Win, 2 * Hin, 2 * Win, 0, &p2, "gx_deconv5x5s2_fwd",
                                2, 128) == GX_OK)
            p0 = p2;
    }
    float* dst = p0.g.nsplit > 1 ? part : y;
    if (stats_parts_out) *stats_parts_out = 0;
    {
        const ConvGeom& g = p0.g;
        const double flops = 2.0 * g.N * (double)g.M * g.K * 25 * g.Hb * g.Wb;
        const double bytes = 4.0 * ((double)g.N * g.K * g.Hi * g.Wi + (double)g.N * g.M * g.Ho * g.Wo +
                                    25.0 * g.K * g.M);
        GxProf pf(KID_TAPCONV_DT0, s, flops, bytes);
        dim3 grid(p0.grid.x, p0.grid.y * 2, p0.grid.z);
        ConvGeom gg = p0.g;
        static const char* dma_env = getenv("GENESIS_TAPCONV_DMA");
        gg.zeros = (dma_env ? dma_env[0] == '1' : true) ? zero_page(s) : nullptr;
        if (p0.mw == 2) {
            gg.zeros = nullptr;
            if (p0.npos == 2) launch_dt<2, false, 2>(grid, p0.lds_bytes, s, x, wpu0, wpu1, bias, dst, gg);
            else launch_dt<4, false, 2>(grid, p0.lds_bytes, s, x, wpu0, wpu1, bias, dst, gg);
        } else if (gg.zeros) {
            // output statistics in the epilogue: one image per tile, no channel split, 8-channel blocks
            const bool st_ok = stats && p0.npos == 2 && gg.nsplit == 1 && gg.lG == 0 && (Cout % 8) == 0;
            if (st_ok) {
                gg.stats = stats;
                gg.stats_parts = gg.tiles_h * gg.tiles_w * 2;
                if (stats_parts_out) *stats_parts_out = gg.stats_parts;
                launch_dt<2, true, 1, true>(grid, p0.lds_bytes, s, x, wpu0, wpu1, bias, dst, gg);
            } else if (p0.npos == 2) launch_dt<2, true>(grid, p0.lds_bytes, s, x, wpu0, wpu1, bias, dst, gg);
            else launch_dt<4, true>(grid, p0.lds_bytes, s, x, wpu0, wpu1, bias, dst, gg);
        } else {
            if (p0.npos == 2) launch_dt<2, false>(grid, p0.lds_bytes, s, x, wpu0, wpu1, bias, dst, gg);
            else launch_dt<4, false>(grid, p0.lds_bytes, s, x, wpu0, wpu1, bias, dst, gg);
        }
    }
    GX_CHECK_LAUNCH("gx_deconv5x5s2_fwd");
    if (parts_out) {
        *parts_out = p0.g.nsplit > 1 ? part : y;
        *nsplit_out = p0.g.nsplit;
        return GX_OK;
    }
    if (p0.g.nsplit > 1) return launch_splitk_reduce(part, bias, y, p0, s);
    return GX_OK;
}

int gx_deconv5x5s2_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cin_out, int Cout,
                         int Hin, int Win, void* ws, size_t ws_bytes, gx_stream_t stream) {
    int rc = check_dims("gx_deconv5x5s2_dgrad", N, Cin, Cout, Hin, Win);
    if (rc) return rc;
    GX_CHECK_ARG(dy && w && dx && ws, "gx_deconv5x5s2_dgrad: null pointer");
    GX_CHECK_ARG(Cin_out > 0 && Cin_out <= Cin, "gx_deconv5x5s2_dgrad: Cin_out out of range");
    GX_CHECK_ARG(ws_bytes >= gx_deconv5x5s2_ws_bytes(N, Cin, Cout, Hin, Win), "gx_deconv5x5s2_dgrad: workspace too small");
    // only the first Cin_out input channels get a gradient (the decoder's coordinate channels need none);
    // weights are packed with the full Cin (so W's row stride is right), output channels are masked at Cin_out.
    const int Kpad = gx_round_up(Cout, 8), Mpad = gx_round_up(Cin, 64);
    hipStream_t s = (hipStream_t)stream;
    float* wp = (float*)ws;
    float* part = wp + deconv_pack_floats(Cin, Cout);
    const float* wpu;
    if (gx_kq_deconv_dgrad_h_eligible(N, Cout, Cin_out, Hin, Win)) {     // on the bf16 matrix pipe
        const int f16 = gx_kq_f16_on() ? 20 : 0;       // pack 44 (gx_kq_precision(2))
        rc = launch_pack(w, wp, 24 + f16, Cout, Cin, 25, Cout, Mpad, s, &wpu);
        if (rc) return rc;
        float* amax_ws = f16 ? (float*)((char*)ws + gx_deconv5x5s2_ws_bytes(N, Cin, Cout, Hin, Win)) - gx_kq_amax_ws_floats() : nullptr;
        int xn = 0;
        const float* xparts = f16 ? gx_amax_link_take(dy, (size_t)N * Cout * 4 * Hin * Win, &xn) : nullptr;     // dy's partial maxima from the kernel that wrote it
        return gx_kq_deconv_dgrad_h_launch(dy, wpu, dx, N, Cout, Cin_out, Hin, Win, s, amax_ws,
                                           f16 ? (const float*)((const char*)wpu + gx_kq_h_amax_off(Cout, Mpad, 25)) : nullptr,
                                           xparts, xn);
    }
    if (gx_kq_deconv_eligible(N, Cout, Cin_out, Hin, Win, 1)) {
        rc = launch_pack(w, wp, 14, Cout, Cin, 25, Kpad, Mpad, s, &wpu);
        if (rc) return rc;
        return gx_kq_deconv_dgrad_launch(dy, wpu, dx, N, Cout, Cin_out, Hin, Win, s);
    }
    rc = launch_pack(w, wp, 4, Cout, Cin, 25, Kpad, Mpad, s, &wpu);
    if (rc) return rc;
    TapPlan pl;
    rc = plan_tapconv<M_DG>(N, Cout, Cin_out, Mpad, Hin, Win, 2 * Hin, 2 * Win, Hin, Win, 0, &pl, "gx_deconv5x5s2_dgrad");
    if (rc) return rc;
    static const char* mw_env = getenv("GENESIS_TAPCONV_MW2");
    if (pl.g.nsplit > 1 && !(mw_env && mw_env[0] == '0')) {
        TapPlan p2;
        if (plan_tapconv<M_DG>(N, Cout, Cin_out, Mpad, Hin, Win, 2 * Hin, 2 * Win, Hin, Win, 0, &p2,
                               "gx_deconv5x5s2_dgrad", 1, 128) == GX_OK)
            pl = p2;
    }
    rc = launch_tapconv<M_DG>(dy, wpu, nullptr, pl.g.nsplit > 1 ? part : dx, pl, s, "gx_deconv5x5s2_dgrad");
    if (rc) return rc;
    if (pl.g.nsplit > 1) return launch_splitk_reduce(part, nullptr, dx, pl, s);
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
    if (gx_wgq_deconv_eligible(N, Cin, Cout, Hin, Win))
        return gx_wgq_deconv(x, dy, dw, N, Cin, Cout, Hin, Win, part,
                             (int)(ws_bytes / sizeof(float) / ((size_t)25 * pl.g.CApad * pl.g.CBpad)), s);
    if (pl.g.cls_begin[4] > 0) {
        rc = launch_wgrad_deconv(dy, x, part, pl, s, "gx_deconv5x5s2_wgrad");
        if (rc) return rc;
    } else {
        rc = launch_wgrad<W_D00>(dy, x, part, pl, s, "gx_deconv5x5s2_wgrad(0,0)"); if (rc) return rc;
        rc = launch_wgrad<W_D01>(dy, x, part, pl, s, "gx_deconv5x5s2_wgrad(0,1)"); if (rc) return rc;
        rc = launch_wgrad<W_D10>(dy, x, part, pl, s, "gx_deconv5x5s2_wgrad(1,0)"); if (rc) return rc;
        rc = launch_wgrad<W_D11>(dy, x, part, pl, s, "gx_deconv5x5s2_wgrad(1,1)"); if (rc) return rc;
    }
    return launch_wgrad_reduce(part, dw, pl, 1, s);
}

}  // extern "C"
