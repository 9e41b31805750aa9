// conv3x3 weight gradient of the 32 -> 32 layers on a grid that is not a power of two (the BroadcastDecoder's convs on the
// (S + 2L)^2 canvas, modules/decoders.py:21-35: 72 x 72 at 64 x 64 images) -- "column strips", no LDS staging.
//
// dw[co][ci][kh][kw] = sum_{n,y,x} dy[n][co][y][x] x[n][ci][y + kh - 1][x + kw - 1]: per tap a 32 x 32 matrix with the pixels as
// the contraction index, on v_mfma_f32_32x32x16_bf16 with every fp32 product as six bf16 piece products (the weight gradients'
// arithmetic everywhere else: gx_wgq.hip, gx_conv.hip wgrad_fast_kernel<.., B6>).  Lane half h of the MFMA supplies 8 CONSECUTIVE
// contraction indices = 8 consecutive pixels of one image row, and the two halves need not be neighbours: a wave works on TWO
// strips (image, 8-pixel column group, row segment) at once, lane = (channel, strip).
//
// Round 4's kernel for these layers (four images per tile, wgrad_fast_kernel<.., QUAD, B6>) stages 8 x 8-pixel tiles through LDS
// with 4-byte gathers and re-splits a 3 x 10 window of x for every tile row: 247 us per layer for 36 us of MFMA time (DESIGN.md
// finding 25: one wave per SIMD, ~5 vector instructions per MFMA, 4 us of un-hidden staging per tile).  Here a wave walks DOWN its
// strip: the 3 x 10 window of x is a rolling one -- each step splits ONE new row (10 values) and 8 dy values, every tap's B operand
// is a shifted view of the split window (54 MFMAs per step for ~2.7 vector instructions each) -- the loads are plain 16-byte
// global loads one step ahead (a strip's rows are 32-byte pieces of consecutive cache lines; the neighbouring strips' waves pick
// up the rest of each line from L2), two waves per SIMD, no barrier until the final reduction.
// Partial sums: one 9 x 32 x 32 block per workgroup (the four waves added in wave order through LDS), summed over the workgroups
// in fixed order by two small launches -- bit-reproducible.
#include "gx_common.h"

#include <cstdlib>

// measurement builds (wrong results): 1 = every load reads image row 0 (a footprint that stays in L1 / L2), 2 = no MFMAs,
// 4 = no splits after the prologue
#ifndef GX_WS_ABL
#define GX_WS_ABL 0
#endif

namespace {

typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ws_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int ws_u32x4 __attribute__((ext_vector_type(4)));
struct WsB3 { ws_bf16x8 h, m, l; };

__device__ __forceinline__ f32x16 ws_mma6(const WsB3& a, const WsB3& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);      // small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}

// two fp32 values -> their three bf16 pieces, PACKED (low half = the first value): 3 v_cvt_pk + 4 and / shift + 4 subtractions
__device__ __forceinline__ unsigned ws_pk(float a, float b) {
    const ws_bf16x2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void ws_split2(float v0, float v1, unsigned& ph, unsigned& pm, unsigned& pl) {
    ph = ws_pk(v0, v1);
    const float r0 = v0 - __builtin_bit_cast(float, ph << 16), r1 = v1 - __builtin_bit_cast(float, ph & 0xffff0000u);
    pm = ws_pk(r0, r1);
    pl = ws_pk(r0 - __builtin_bit_cast(float, pm << 16), r1 - __builtin_bit_cast(float, pm & 0xffff0000u));
}

// one split row of the window: 10 values (columns 8 j - 1 .. 8 j + 8) as five packed pairs per piece
struct WsRow { unsigned h[5], m[5], l[5]; };

__device__ __forceinline__ void ws_split_row(const float (&v)[10], WsRow& w) {
#pragma unroll
    for (int i = 0; i < 5; ++i) ws_split2(v[2 * i], v[2 * i + 1], w.h[i], w.m[i], w.l[i]);
}
// the B operand of tap column kw: window elements kw .. kw + 7 (kw = 1: funnel shifts of neighbouring pairs)
template <int KW>
__device__ __forceinline__ ws_bf16x8 ws_view(const unsigned (&p)[5]) {
    ws_u32x4 o;
    if constexpr (KW == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_alignbit(p[i + 1], p[i], 16);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = p[i + KW / 2];
    }
    return __builtin_bit_cast(ws_bf16x8, o);
}

struct StripGeom {
    int N, H, W;
    int G;          // column groups per row: W / 8
    int RS, RH;     // row segments per image, rows per segment (RS * RH >= H)
    int ntask;      // N * RS * G strips
    int npair;      // ceil(ntask / 2): one pair per wave and round
};

constexpr int kStripC = 32;
constexpr int kStripPart = 9 * 16 * 64;          // floats of one partial: [tap][accumulator register][lane]

__global__ void __launch_bounds__(256, 1)
wgrad_strip_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                   float* __restrict__ bias_part, const StripGeom g) {
    __shared__ float red[kStripPart];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int H = g.H, W = g.W;
    const int plane = H * W;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum = 0.f;

    const int nwaves = gridDim.x * 4;
    for (int pair = blockIdx.x * 4 + wave; pair < g.npair; pair += nwaves) {
        int task = 2 * pair + half;
        const bool live = task < g.ntask;
        if (!live) task = g.ntask - 1;            // (an odd strip count: the idle half multiplies zeros)
        const int j = task % g.G;
        const int t2 = task / g.G;
        const int s = t2 % g.RS;
        const int n = t2 / g.RS;
        const int y0 = s * g.RH;
        const float* xp = x + ((size_t)n * kStripC + c) * plane + 8 * j;
        const float* dp = dy + ((size_t)n * kStripC + c) * plane + 8 * j;
        const bool has_l = j > 0, has_r = j + 1 < g.G;

        // raw row of x (10 values: halo | 8 | halo) / of dy (8 values); rows outside the image are zero.  The loads are
        // UNCONDITIONAL (clamped addresses, the values selected to zero afterwards): no branch around a load, so a step is one basic
        // block, its loads stay in flight across steps and the waits count exactly the loads that are older
        const int lo_l = has_l ? -1 : 0, lo_r = has_r ? 8 : 7;
#define GX_WS_LOAD_X(rx_, y_)                                                                         \
        {                                                                                             \
            const int yy_ = (y_);                                                                     \
            const bool ok_ = yy_ >= 0 && yy_ < H;                                                     \
            const float* r_ = xp + (size_t)((ok_ && !(GX_WS_ABL & 1)) ? yy_ : 0) * W;                 \
            const f32x4 v0_ = *reinterpret_cast<const f32x4*>(r_);                                    \
            const f32x4 v1_ = *reinterpret_cast<const f32x4*>(r_ + 4);                                \
            const float vl_ = r_[lo_l], vr_ = r_[lo_r];                                               \
            rx_[0] = (ok_ && has_l) ? vl_ : 0.f;                                                      \
            rx_[9] = (ok_ && has_r) ? vr_ : 0.f;                                                      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) { rx_[1 + i] = ok_ ? v0_[i] : 0.f; rx_[5 + i] = ok_ ? v1_[i] : 0.f; } \
        }
#define GX_WS_LOAD_D(rd_, y_)                                                                         \
        {                                                                                             \
            const int yy_ = (y_);                                                                     \
            const bool ok_ = live && yy_ < H && yy_ < y0 + g.RH;                                      \
            const float* r_ = dp + (size_t)((ok_ && !(GX_WS_ABL & 1)) ? yy_ : 0) * W;                 \
            const f32x4 v0_ = *reinterpret_cast<const f32x4*>(r_);                                    \
            const f32x4 v1_ = *reinterpret_cast<const f32x4*>(r_ + 4);                                \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) { rd_[i] = ok_ ? v0_[i] : 0.f; rd_[4 + i] = ok_ ? v1_[i] : 0.f; } \
        }
#define GX_WS_SPLIT_D(a3_, rd_)                                                                       \
        {                                                                                             \
            ws_u32x4 ah_, am_, al_;                                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
                unsigned h_, m_, l_;                                                                  \
                ws_split2(rd_[2 * i], rd_[2 * i + 1], h_, m_, l_);                                    \
                ah_[i] = h_; am_[i] = m_; al_[i] = l_;                                                \
            }                                                                                         \
            a3_.h = __builtin_bit_cast(ws_bf16x8, ah_); a3_.m = __builtin_bit_cast(ws_bf16x8, am_);   \
            a3_.l = __builtin_bit_cast(ws_bf16x8, al_);                                               \
            bsum += ((rd_[0] + rd_[1]) + (rd_[2] + rd_[3])) + ((rd_[4] + rd_[5]) + (rd_[6] + rd_[7])); \
        }
        // software pipeline, per row r of the strip (y = y0 + r):
        //   loads   x row y + 5, dy row y + 4      -> the raw buffer that is free           (split three steps later: the strips'
        //                                              rows are 32-byte pieces of 64 different cache lines per load instruction)
        //   splits  x row y + 2, dy row y + 1      -> the free window slot / the other A set (vector ALU, independent of ...)
        //   MFMAs   of row y: window rows y - 1, y, y + 1 and A = dy row y                  (... these: the two interleave)
        WsRow w0, w1, w2, w3;
        WsB3 a0, a1;
        float rx0[10], rx1[10], rx2[10], rx3[10], rd0[8], rd1[8], rd2[8], rd3[8];
        GX_WS_LOAD_X(rx0, y0 - 1)
        GX_WS_LOAD_X(rx1, y0)
        GX_WS_LOAD_X(rx2, y0 + 1)
        GX_WS_LOAD_D(rd0, y0)
        ws_split_row(rx0, w0);
        ws_split_row(rx1, w1);
        ws_split_row(rx2, w2);
        GX_WS_SPLIT_D(a0, rd0)
        GX_WS_LOAD_X(rx0, y0 + 2)
        GX_WS_LOAD_D(rd0, y0 + 1)
        GX_WS_LOAD_X(rx1, y0 + 3)
        GX_WS_LOAD_D(rd1, y0 + 2)
        GX_WS_LOAD_X(rx2, y0 + 4)
        GX_WS_LOAD_D(rd2, y0 + 3)
        // the three taps of a kernel row together, piece product by piece product: a tap's six MFMAs form a dependent chain on its
        // accumulator, and with ONE wave per SIMD nothing else fills the matrix pipe while a chain waits -- rotate over the row's
        // three accumulators (emitted tap by tap the kernel ran at a third of its MFMA rate: 0 0 0 0 0 0 1 1 1 ...)
#define GX_WS_PROD(kh_, ap_, bp_, a3_)                                                                \
        {                                                                                             \
            _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) {                                        \
                if (GX_WS_ABL & 2) acc[(kh_) * 3 + kw][0] += (float)a3_.ap_[0] * (float)bv_[kw].bp_[kw];                      \
                else acc[(kh_) * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3_.ap_, bv_[kw].bp_, acc[(kh_) * 3 + kw], 0, 0, 0); \
            }                                                                                         \
        }
#define GX_WS_TAPS(kh_, wrow_, a3_)                                                                   \
        {                                                                                             \
            WsB3 bv_[3];                                                                              \
            bv_[0].h = ws_view<0>(wrow_.h); bv_[0].m = ws_view<0>(wrow_.m); bv_[0].l = ws_view<0>(wrow_.l); \
            bv_[1].h = ws_view<1>(wrow_.h); bv_[1].m = ws_view<1>(wrow_.m); bv_[1].l = ws_view<1>(wrow_.l); \
            bv_[2].h = ws_view<2>(wrow_.h); bv_[2].m = ws_view<2>(wrow_.m); bv_[2].l = ws_view<2>(wrow_.l); \
            GX_WS_PROD(kh_, m, m, a3_)      /* small terms first */                                   \
            GX_WS_PROD(kh_, l, h, a3_)                                                                \
            GX_WS_PROD(kh_, h, l, a3_)                                                                \
            GX_WS_PROD(kh_, m, h, a3_)                                                                \
            GX_WS_PROD(kh_, h, m, a3_)                                                                \
            GX_WS_PROD(kh_, h, h, a3_)                                                                \
        }
#define GX_WS_STEP(wa_, wb_, wc_, wd_, acur_, anxt_, rxc_, rdc_, rxn_, rdn_, r_)                      \
        {                                                                                             \
            if (!(GX_WS_ABL & 4)) { ws_split_row(rxc_, wd_); GX_WS_SPLIT_D(anxt_, rdc_) }             \
            else { wd_.h[0] ^= __builtin_bit_cast(unsigned, rxc_[0] + rdc_[0]); }                     \
            GX_WS_LOAD_X(rxn_, y0 + (r_) + 5)                                                         \
            GX_WS_LOAD_D(rdn_, y0 + (r_) + 4)                                                         \
            GX_WS_TAPS(0, wa_, acur_)                                                                 \
            GX_WS_TAPS(1, wb_, acur_)                                                                 \
            GX_WS_TAPS(2, wc_, acur_)                                                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < 54; ++i_) {                                       \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* one MFMA ... */            \
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      /* ... three vector-ALU instructions */ \
            }                                                                                         \
        }
        // (raw ring of four: step r splits buffer r % 4 and loads buffer (r + 3) % 4)
        for (int r = 0; r < g.RH; r += 4) {
            GX_WS_STEP(w0, w1, w2, w3, a0, a1, rx0, rd0, rx3, rd3, r)
            if (r + 1 < g.RH) GX_WS_STEP(w1, w2, w3, w0, a1, a0, rx1, rd1, rx0, rd0, r + 1)
            if (r + 2 < g.RH) GX_WS_STEP(w2, w3, w0, w1, a0, a1, rx2, rd2, rx1, rd1, r + 2)
            if (r + 3 < g.RH) GX_WS_STEP(w3, w0, w1, w2, a1, a0, rx3, rd3, rx2, rd2, r + 3)
        }
#undef GX_WS_STEP
#undef GX_WS_TAPS
#undef GX_WS_PROD
#undef GX_WS_SPLIT_D
#undef GX_WS_LOAD_X
#undef GX_WS_LOAD_D
    }

    // ---- the workgroup's partial: the four waves' blocks added in wave order
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int idx = (t * 16 + e) * 64 + lane;
                    red[idx] = w == 0 ? acc[t][e] : red[idx] + acc[t][e];
                }
        }
        __syncthreads();
    }
    float* out = part + (size_t)blockIdx.x * kStripPart;
    for (int i = tid; i < kStripPart; i += 256) out[i] = red[i];
    if (bias_part) {
        // the workgroup's 32 channel sums of dy: the two lane halves by a shuffle, the four waves in wave order -- one 32-float record
        // per workgroup (the per-thread records of round 4 left the final launch ONE workgroup walking 2048 records in a chain of
        // dependent-latency loads: 60 us for 256 KB)
        __shared__ float bred[4][32];
        const float b2 = bsum + __shfl_xor(bsum, 32, 64);
        if (lane < 32) bred[wave][lane] = b2;
        __syncthreads();
        if (tid < 32) bias_part[(size_t)blockIdx.x * 32 + tid] = (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]);
    }
}

// first level: slice `blockIdx.y` of the workgroups' partials, element by element
__global__ void __launch_bounds__(256)
wgrad_strip_reduce1_kernel(const float* __restrict__ part, int nb, int per, float* __restrict__ part2) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int b0 = blockIdx.y * per, b1 = b0 + per < nb ? b0 + per : nb;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        s0 += part[(size_t)b * kStripPart + e];
        s1 += part[(size_t)(b + 1) * kStripPart + e];
        s2 += part[(size_t)(b + 2) * kStripPart + e];
        s3 += part[(size_t)(b + 3) * kStripPart + e];
    }
    for (; b < b1; ++b) s0 += part[(size_t)b * kStripPart + e];
    part2[(size_t)blockIdx.y * kStripPart + e] = (s0 + s1) + (s2 + s3);
}

// second level: the slices in order -> dw [32][32][3][3]; block 36: the bias gradient from the per-thread dy sums
__global__ void __launch_bounds__(256)
wgrad_strip_reduce2_kernel(const float* __restrict__ part2, int nslice, float* __restrict__ dw, const float* __restrict__ bias_part,
                           int nb, float* __restrict__ dbias) {
    if (blockIdx.x == kStripPart / 256) {
        // dbias[c] = sum over the workgroups' records: thread (c, k) takes every 8th record, four independent chains
        __shared__ float bs[8][32];
        const int c = threadIdx.x & 31, k = threadIdx.x >> 5;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int rec = k;
        for (; rec + 24 < nb; rec += 32) {
            s0 += bias_part[(size_t)rec * 32 + c];
            s1 += bias_part[(size_t)(rec + 8) * 32 + c];
            s2 += bias_part[(size_t)(rec + 16) * 32 + c];
            s3 += bias_part[(size_t)(rec + 24) * 32 + c];
        }
        for (; rec < nb; rec += 8) s0 += bias_part[(size_t)rec * 32 + c];
        bs[k][c] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (threadIdx.x < 32) {
            float t = 0.f;
            for (int i = 0; i < 8; ++i) t += bs[i][threadIdx.x];
            dbias[threadIdx.x] = t;
        }
        return;
    }
    const int e = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    for (int i = 0; i < nslice; ++i) s += part2[(size_t)i * kStripPart + e];
    const int t = e >> 10, r = (e >> 6) & 15, lane = e & 63;
    const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = lane & 31;      // C / D layout of the 32 x 32 MFMA
    dw[((size_t)co * kStripC + ci) * 9 + t] = s;
}

constexpr int kStripBlocks = 256;       // one workgroup per CU (the 144 accumulators + the rolling window need > 256 registers)
constexpr int kStripSlices = 16;

bool strip_geom(int N, int C, int H, int W, StripGeom* g) {
    if (C != kStripC || N <= 0 || H < 1 || W < 8 || (W & 7) || (double)N * C * H * W >= 2.0e9) return false;
    g->N = N; g->H = H; g->W = W; g->G = W / 8;
    // row segments: enough strips for every wave of the launch to have (about) one pair, never shorter than 8 rows
    const long long per_image = g->G;
    int RS = (int)((2LL * kStripBlocks * 4 + (long long)N * per_image - 1) / ((long long)N * per_image));
    if (RS < 1) RS = 1;
    while (RS > 1 && (H + RS - 1) / RS < 8) --RS;
    g->RH = (H + RS - 1) / RS;
    g->RS = (H + g->RH - 1) / g->RH;
    g->ntask = N * g->RS * g->G;
    g->npair = (g->ntask + 1) / 2;
    return true;
}

}  // namespace

bool gx_wstrip_supported(int N, int C, int H, int W) {
    static const char* env = getenv("GENESIS_WGRAD_STRIP");
    if (env && env[0] == '0') return false;
    StripGeom g;
    return gx_wgq_bf16_pipe() && strip_geom(N, C, H, W, &g);
}

size_t gx_wstrip_ws_floats(int N, int C, int H, int W) {
    (void)N; (void)C; (void)H; (void)W;
    return (size_t)(kStripBlocks + kStripSlices) * kStripPart + (size_t)kStripBlocks * 256;
}

int gx_wstrip_launch(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int H, int W, void* ws,
                     hipStream_t s) {
    StripGeom g;
    GX_CHECK_ARG(strip_geom(N, C, H, W, &g), "gx_wstrip: needs 32 channels and W %% 8 == 0");
    float* part = (float*)ws;
    float* part2 = part + (size_t)kStripBlocks * kStripPart;
    float* bias_part = part2 + (size_t)kStripSlices * kStripPart;
    int nb = kStripBlocks;
    if (nb * 4 > g.npair) nb = gx_ceil_div(g.npair, 4);
    {
        GxProf pf(KID_WGRAD_C3, s, 2.0 * N * (double)C * C * 9 * H * W, 4.0 * (2.0 * N * C * H * W + (double)nb * kStripPart));
        hipLaunchKernelGGL(wgrad_strip_kernel, dim3(nb), dim3(256), 0, s, x, dy, part, dbias ? bias_part : (float*)nullptr, g);
    }
    GX_CHECK_LAUNCH("gx_conv3x3_wgrad_quad(strips)");
    {
        const int per = gx_ceil_div(nb, kStripSlices);
        const int nslice = gx_ceil_div(nb, per);
        GxProf pf(KID_WGRAD_REDUCE, s, 0.0, 4.0 * ((double)nb + 2.0 * nslice + 1.0) * kStripPart);
        hipLaunchKernelGGL(wgrad_strip_reduce1_kernel, dim3(kStripPart / 256, nslice), dim3(256), 0, s, (const float*)part, nb, per,
                           part2);
        hipLaunchKernelGGL(wgrad_strip_reduce2_kernel, dim3(kStripPart / 256 + (dbias ? 1 : 0)), dim3(256), 0, s,
                           (const float*)part2, nslice, dw, (const float*)bias_part, nb, dbias);
    }
    GX_CHECK_LAUNCH("gx_conv3x3_wgrad_quad(strip reduce)");
    return GX_OK;
}
