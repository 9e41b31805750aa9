// Weight gradients of the conv3x3 / transposed-conv 5x5 s2 layers, second generation ("wgq"): both operands staged
// global -> LDS by LDS-DMA (16-byte pieces), many layers per launch.
//
//   dW[A-channel][B-channel][tap] = sum over images and pixels  dy[A-channel][pixel] * x[B-channel][pixel + tap offset]
//   (conv3x3: A = Cout, B = Cin; transposed conv: A = Cout of dy at the output parity of the tap, B = Cin)
//
// Round-2 measurements on the round-1 kernel (wgrad_fast_kernel, tools/kq_time.py + ablation builds):
//   * its A operand (dy) goes global -> registers while its B operand goes global -> LDS by LDS-DMA.  hipcc waits
//     vmcnt(0) at the first use of a register load while a DMA is in flight, so every batch of A values drains the next
//     tile's whole DMA: without the A loads the transposed-conv classes run 117 instead of 94 TFLOP/s, without either
//     staging 129.
//   * a launch costs ~34 us beyond its MFMA time (prologue, slab epilogue, reduce launch): 64 -> 64 @ 64x64 runs
//     83 TFLOP/s at batch 32 and 111 at batch 256.
// Here: (1) A and B tiles both arrive by LDS-DMA in 16-byte pieces (1 KiB per wave instruction, 14-18 instructions per
// thread per tile instead of 64 + register loads), the only vmcnt wait is the one in front of the tile's barrier;
// LDS images are XOR-swizzled per channel so that the 16-byte operand reads of 16 different channels are conflict free
// (the swizzle is applied to the per-lane GLOBAL address, the DMA destination stays lane-linear); (2) both column
// parities of an output-row parity of the transposed conv are accumulated by one workgroup (15 / 10 taps: the 16-byte
// pieces of dy hold both anyway -- half the dy traffic and launches); (3) one launch takes a table of jobs (layers)
// and hands every job a share of the 256 workgroups in proportion to its work: fewer, fatter split-K slabs and one
// prologue / epilogue per GROUP of layers.  Jobs are queued by the deferred-reduction machinery (gx_defer_*) during
// the backward pass and launched by gx_defer_flush.
//
// Workgroup = 4 waves; wave (wm, wn) owns the 32 x 32 block (A channels wm, B channels wn) of all NT taps
// (v_mfma_f32_32x32x2_f32, k = pixels; lane-half h takes the pixels of tile half h, 4 consecutive pixels per group).
// Tile = 64 base pixels (TH x TW, TW = 32, 16 or 8); two LDS stages; one barrier per tile.
#include "gx_common.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

enum { WQ_C3 = 0, WQ_DR0 = 1, WQ_DR1 = 2, WQ_C5A = 3, WQ_C5B = 4 };

template <int CLS> struct WqTap;
template <> struct WqTap<WQ_C3> {
    static constexpr int NT = 9, SA = 1, PA = 0, NPB = 1, NRO = 3, RO0 = 0;
    __host__ __device__ static constexpr int ro(int t) { return t / 3; }
    __host__ __device__ static constexpr int co(int t) { return t % 3; }
    __host__ __device__ static constexpr int pb(int) { return 0; }
    __host__ __device__ static constexpr int gt(int t) { return t; }
};
// transposed conv k5 s2 p2 op1, output rows 2r + PA: taps t = khi * 5 + kw, kh = 2 khi + PA; x halo offsets
// (2 - kh / 2, 2 - kw / 2); the tap's column parity kw & 1 selects which of the two interleaved dy values is its A operand
template <int PA_> struct WqTapDR {
    static constexpr int NKH = PA_ ? 2 : 3;
    static constexpr int NT = NKH * 5, SA = 2, PA = PA_, NPB = 2, NRO = NKH, RO0 = 2 - (NKH - 1);
    __host__ __device__ static constexpr int kh(int t) { return 2 * (t / 5) + PA_; }
    __host__ __device__ static constexpr int kw(int t) { return t % 5; }
    __host__ __device__ static constexpr int ro(int t) { return 2 - kh(t) / 2; }
    __host__ __device__ static constexpr int co(int t) { return 2 - kw(t) / 2; }
    __host__ __device__ static constexpr int pb(int t) { return kw(t) & 1; }
    __host__ __device__ static constexpr int gt(int t) { return kh(t) * 5 + kw(t); }
};
template <> struct WqTap<WQ_DR0> : WqTapDR<0> {};
template <> struct WqTap<WQ_DR1> : WqTapDR<1> {};
// 5 x 5 stride-1 pad-2 conv (the gated stacks of third_party/sylvester, VAE.py:18-33), row-ring tiles only:
// dW[a][b][kh][kw] = sum_p A[a][p] * B[b][p + (kh - 2, kw - 2)].  Kernel rows 0..2 (15 taps) and 3..4 (10 taps) are two
// jobs, as the two output-row parities of the transposed conv are (15 / 10 taps = 240 / 160 accumulator registers).
// ro(t) = kh; the x row of tap row index rr = ro - RO0 is r + RO0 + rr - 1, i.e. RO0 = -1: rows r-2..r; RO0 = 2: r+1, r+2.
template <int HALF> struct WqTapC5 {
    static constexpr int KH0 = HALF ? 3 : 0, NKH = HALF ? 2 : 3;
    static constexpr int NT = NKH * 5, SA = 1, PA = 0, NPB = 1, NRO = NKH, RO0 = HALF ? 2 : -1;
    __host__ __device__ static constexpr int kh(int t) { return KH0 + t / 5; }
    __host__ __device__ static constexpr int ro(int t) { return RO0 + t / 5; }
    __host__ __device__ static constexpr int co(int t) { return t % 5; }
    __host__ __device__ static constexpr int pb(int) { return 0; }
    __host__ __device__ static constexpr int gt(int t) { return kh(t) * 5 + t % 5; }
};
template <> struct WqTap<WQ_C5A> : WqTapC5<0> {};
template <> struct WqTap<WQ_C5B> : WqTapC5<1> {};
// column variants of a class: co = 0 .. NCOV-1 <-> shift s = co + CS0 of the dy operand against the aligned x octet
template <int CLS> struct WqCols { static constexpr int NCOV = 3, CS0 = -1; };
template <> struct WqCols<WQ_C5A> { static constexpr int NCOV = 5, CS0 = -2; };
template <> struct WqCols<WQ_C5B> { static constexpr int NCOV = 5, CS0 = -2; };

struct WqJob {
    const float* a;      // dy [N, CA, SA*Hb, SA*Wb]
    const float* b;      // x  [N, CB, Hb, Wb]
    float* partial;      // split-K slabs [split][Ttot][CApad][CBpad]
    int N, CA, CB, CApad, CBpad, Hb, Wb;
    int tiles_h, tiles_w, ntiles;       // tiles per image column / row, total
    int nsplit, nbt;                    // splits per channel block; B-channel blocks (CBpad / 64)
    int wg_begin;                       // first blockIdx.x of this job
    int Ttot;                           // taps in the slab (9 / 25)
};
constexpr int kMaxJobs = 12;
struct WqTable { int njobs; WqJob job[kMaxJobs]; };

constexpr int kSB = 40;     // pieces per B channel in LDS ((TH + 2) * (TW + 8) / 4, padded to a multiple of 8)

// per-workgroup constants of the tile loop
struct WqW {
    const float* a; const float* b; const float* zeros;
    int CA, CB, ca0, cb0, Hb, Wb, Wa, HaWa, HbWb;
    int wave_u;                                   // wave id as a scalar (DMA destinations are wave-uniform)
    int h, a_base, a_f, b_base, b_f;              // per-lane operand read bases
};

// One tile: LDS-DMA of tile (img, th, tw) into stage `wr` (live = false: past the last tile, every piece comes from the
// zero page -- no branch around the issue, so it shares the MFMA loop's basic block and the scheduler spreads the DMA
// instructions between the MFMAs), operands of the current tile out of stage `rd`.  The two stages are distinct
// __restrict__ parameters of an inlined function: that is what lets hipcc prove that the LDS reads do not touch the
// stage the in-flight DMA writes; with plain pointer arithmetic on one LDS array it puts s_waitcnt vmcnt(0) in front
// of the first ds_read after the DMA issue and the staging never overlaps the MFMAs.
template <int CLS, int LTW, int NI>
__device__ __forceinline__ void wq_issue(float* __restrict__ wr, const WqW& w, const int (&goff)[NI],
                                         const int (&info)[NI], const int img, const int th, const int tw,
                                         const bool live) {
    using WT = WqTap<CLS>;
    constexpr int SA = WT::SA, TW = 1 << LTW, TH = 64 >> LTW;
    constexpr int A_PIECES = 64 * TH * (SA * TW / 4);
    const int R0 = th * TH, C0 = tw * TW;
    const float* a_t = w.a + ((size_t)img * w.CA + w.ca0) * w.HaWa + (size_t)(SA * R0 + WT::PA) * w.Wa + SA * C0;
    const float* b_t = w.b + ((size_t)img * w.CB + w.cb0) * w.HbWb + (ptrdiff_t)(R0 - 1) * w.Wb + (C0 - 4);
    float* dst = wr + w.wave_u * 256;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const bool isA = i * 256 < A_PIECES;          // whole instructions belong to one region
        bool ok = live & !(info[i] & 1);              // bitwise: no short-circuit control flow
        if (!isA) {
            const int row = R0 - 1 + ((info[i] >> 8) & 255), col = C0 - 4 + 4 * (info[i] >> 16);
            ok = ok & ((unsigned)row < (unsigned)w.Hb) & ((unsigned)col < (unsigned)w.Wb);
        }
        const float* gp = ok ? (isA ? a_t : b_t) + goff[i] : w.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    }
}

template <int CLS, int LTW, int NI>
__device__ __forceinline__ void wq_tile(const float* __restrict__ rd, float* __restrict__ wr,
                                        f32x16 (&acc)[WqTap<CLS>::NT], const WqW& w, const int (&goff)[NI],
                                        const int (&info)[NI], const int img, const int th, const int tw,
                                        const bool live) {
    using WT = WqTap<CLS>;
    constexpr int NT = WT::NT, SA = WT::SA, NRO = WT::NRO, RO0 = WT::RO0;
    constexpr int TW = 1 << LTW;
    constexpr int RPA = SA * TW / 4, RPB = (TW + 8) / 4;
    wq_issue<CLS, LTW, NI>(wr, w, goff, info, img, th, tw, live);
    // operand registers of one 4-pixel group: A 4 (conv3x3) / 8 (transposed conv) values, B NRO rows x 6 values
    struct Grp { f32x4 a[WT::NPB]; float e0[NRO], e1[NRO]; f32x4 m[NRO]; };
#define GX_WQ_READ(g_, dst_)                                                                                    \
    {                                                                                                           \
        const int jpix = w.h * 32 + 4 * (g_);                                                                   \
        const int r_ = jpix >> LTW, c_ = jpix & (TW - 1);                                                       \
        const int qa = r_ * RPA + (SA * c_) / 4;                                                                \
        _Pragma("unroll") for (int p = 0; p < WT::NPB; ++p)                                                     \
            dst_.a[p] = *reinterpret_cast<const f32x4*>(rd + w.a_base + (((qa + p) ^ w.a_f) << 2));             \
        _Pragma("unroll") for (int rr = 0; rr < NRO; ++rr) {                                                    \
            const int qb = (r_ + RO0 + rr) * RPB + (c_ >> 2);                                                   \
            dst_.e0[rr] = rd[w.b_base + ((qb ^ w.b_f) << 2) + 3];                                               \
            dst_.m[rr] = *reinterpret_cast<const f32x4*>(rd + w.b_base + (((qb + 1) ^ w.b_f) << 2));            \
            dst_.e1[rr] = rd[w.b_base + (((qb + 2) ^ w.b_f) << 2)];                                             \
        }                                                                                                       \
    }
    // A value of pixel u for column parity pb: conv3x3 a[0][u]; transposed conv: element 2u + pb of the 8 values
#define GX_WQ_AVAL(src_, pb_, u_) (SA == 1 ? src_.a[0][u_] : src_.a[(2 * (u_) + (pb_)) >> 2][(2 * (u_) + (pb_)) & 3])
    // B value at halo column offset x (0..5) of row rr
#define GX_WQ_BVAL(src_, rr_, x_) ((x_) == 0 ? src_.e0[rr_] : ((x_) == 5 ? src_.e1[rr_] : src_.m[rr_][(x_) - 1]))
#define GX_WQ_MMA(src_)                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                           \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                      \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(GX_WQ_AVAL(src_, WT::pb(t), u),                   \
                                                              GX_WQ_BVAL(src_, WT::ro(t) - RO0, WT::co(t) + u), \
                                                              acc[t], 0, 0, 0);                                 \
    }
    constexpr int NDS = WT::NPB + 3 * NRO;     // LDS read instructions per group
    constexpr int VPG = (NI + 7) / 8;          // DMA instructions placed behind each group's MFMAs
    Grp g0, g1;
    GX_WQ_READ(0, g0)
#pragma unroll
    for (int g = 0; g < 8; g += 2) {
        GX_WQ_READ(g + 1, g1)
        GX_WQ_MMA(g0)
        __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, VPG, 0);
        if (g + 2 < 8) GX_WQ_READ(g + 2, g0)
        GX_WQ_MMA(g1)
        if (g + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, VPG, 0);
    }
#undef GX_WQ_READ
#undef GX_WQ_AVAL
#undef GX_WQ_BVAL
#undef GX_WQ_MMA
}

// ---- the same tile on the bf16 matrix pipe: fp32 products from bf16 pieces (finding 13).
// x = x_hi + x_mid + x_lo (three bf16 values hold the 24 mantissa bits); a * b ~ the six piece products of order <= 2,
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (k = 16 pixels: lane half h supplies 8 CONSECUTIVE pixels of its tile
// half -- the contraction index of a weight gradient is the pixel, so the operand octet is 8 consecutive floats of a
// channel row exactly as the DMA laid them down).  Six MFMAs of 32 cycles per 16 pixels and tap instead of eight of 64;
// the splits and the tap-shifted windows are VALU work in the MFMAs' shadow.
typedef __bf16 gx_bf16x8 __attribute__((ext_vector_type(8)));
struct WqB3 { gx_bf16x8 h, m, l; };
template <int N>
__device__ __forceinline__ void wq_split(const float (&v)[N], __bf16 (&h)[N], __bf16 (&m)[N], __bf16 (&l)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        h[i] = (__bf16)v[i];
        const float r1 = v[i] - (float)h[i];
        m[i] = (__bf16)r1;
        l[i] = (__bf16)(r1 - (float)m[i]);
    }
}
template <int N>
__device__ __forceinline__ WqB3 wq_octet(const __bf16 (&h)[N], const __bf16 (&m)[N], const __bf16 (&l)[N], int o) {
    WqB3 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) { r.h[i] = h[o + i]; r.m[i] = m[o + i]; r.l[i] = l[o + i]; }
    return r;
}
__device__ __forceinline__ f32x16 wq_mma6(const WqB3& a, const WqB3& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);      // small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}

template <int CLS, int LTW, int NI>
__device__ __forceinline__ void wq_tile_b6(const float* __restrict__ rd, float* __restrict__ wr,
                                           f32x16 (&acc)[WqTap<CLS>::NT], const WqW& w, const int (&goff)[NI],
                                           const int (&info)[NI], const int img, const int th, const int tw,
                                           const bool live) {
    using WT = WqTap<CLS>;
    constexpr int NT = WT::NT, SA = WT::SA, NRO = WT::NRO, RO0 = WT::RO0;
    constexpr int TW = 1 << LTW;
    constexpr int RPA = SA * TW / 4, RPB = (TW + 8) / 4;
    constexpr int NCO = NT / NRO;              // taps per halo row (3 / 5)
    wq_issue<CLS, LTW, NI>(wr, w, goff, info, img, th, tw, live);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int jpix = w.h * 32 + 8 * g;
        const int r_ = jpix >> LTW, c_ = jpix & (TW - 1);
        // A: the 8 (conv3x3) / 16 (transposed conv: both column parities interleaved) dy values of these 8 base pixels
        const int qa = r_ * RPA + (SA * c_) / 4;
        float av[8 * SA];
#pragma unroll
        for (int p = 0; p < 2 * SA; ++p) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(rd + w.a_base + (((qa + p) ^ w.a_f) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) av[4 * p + e] = t4[e];
        }
        __bf16 ah[8 * SA], am[8 * SA], al[8 * SA];
        wq_split<8 * SA>(av, ah, am, al);
        WqB3 a3[SA];                                  // [column parity]
#pragma unroll
        for (int pb = 0; pb < SA; ++pb)
#pragma unroll
            for (int i = 0; i < 8; ++i) { a3[pb].h[i] = ah[SA * i + pb]; a3[pb].m[i] = am[SA * i + pb]; a3[pb].l[i] = al[SA * i + pb]; }
#pragma unroll
        for (int rr = 0; rr < NRO; ++rr) {
            // B: halo columns c_ - 1 .. c_ + 8 of row r_ + RO0 + rr (window of 10 floats; conv3x3 offsets 0..2, the
            // transposed conv's 0..2 as well: x halo offset 2 - kw / 2)
            const int qb = (r_ + RO0 + rr) * RPB + (c_ >> 2);
            float bv[10];
            bv[0] = rd[w.b_base + ((qb ^ w.b_f) << 2) + 3];
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(rd + w.b_base + (((qb + 1) ^ w.b_f) << 2));
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(rd + w.b_base + (((qb + 2) ^ w.b_f) << 2));
            bv[9] = rd[w.b_base + (((qb + 3) ^ w.b_f) << 2)];
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[1 + e] = m0[e]; bv[5 + e] = m1[e]; }
            __bf16 bh[10], bm[10], bl[10];
            wq_split<10>(bv, bh, bm, bl);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (WT::ro(t) - RO0 != rr) continue;
                const WqB3 b3 = wq_octet<10>(bh, bm, bl, WT::co(t));
                acc[t] = wq_mma6(a3[WT::pb(t)], b3, acc[t]);
            }
        }
    }
}

// One SEGMENT: the tiles t0, t0 + tstep, ... (< tend) of one 64 x 64 channel block (ca0, cb0) of one layer, accumulated
// in registers and written as one slab (`slab` points at [tap 0][row 0][col 0] of this workgroup's block; rows are ldc
// floats apart, taps tstride).  The grouped kernel hands a workgroup one strided segment, the stream-K kernel one or
// more contiguous ones.
// B6: the tiles run on the bf16 matrix pipe (wq_tile_b6: fp32 products from six bf16 piece products) instead of the fp32 one
template <int CLS, int LTW, bool B6>
__device__ __forceinline__ void wq_segment(const float* a, const float* b, const float* zeros, float* lds, int CA, int CB,
                                           int ca0, int cb0, int Hb, int Wb, int tiles_h, int tiles_w, int t0, int tstep,
                                           int tend, float* slab, int ldc, int tstride) {
    using WT = WqTap<CLS>;
    constexpr int NT = WT::NT, SA = WT::SA;
    constexpr int TW = 1 << LTW, TH = 64 >> LTW;
    constexpr int RPA = SA * TW / 4;             // A pieces per tile row
    constexpr int SAP = TH * RPA;                // A pieces per channel (16 or 32)
    constexpr int RPB = (TW + 8) / 4;            // B pieces per halo row
    constexpr int NPB_ = (TH + 2) * RPB;         // B pieces per channel actually used (<= kSB)
    constexpr int A_PIECES = 64 * SAP, B_PIECES = 64 * kSB;
    constexpr int STAGE = (A_PIECES + B_PIECES) * 4;          // floats per stage
    constexpr int NI = (A_PIECES + B_PIECES) / 256;           // DMA instructions per thread per tile
    static_assert((A_PIECES + B_PIECES) % 256 == 0 && A_PIECES % 256 == 0, "whole wave instructions per region");
    static_assert(NPB_ <= kSB, "B halo tile must fit its LDS slot");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    WqW w;
    w.a = a; w.b = b; w.zeros = zeros;
    w.CA = CA; w.CB = CB;
    w.ca0 = ca0; w.cb0 = cb0;
    w.Hb = Hb; w.Wb = Wb; w.Wa = SA * Wb;
    w.HaWa = SA * Hb * w.Wa; w.HbWb = Hb * Wb;
    w.wave_u = __builtin_amdgcn_readfirstlane(wave);

    // ---- DMA pieces of this thread: piece L = i * 256 + tid of the stage image, fixed over tiles.
    //      A region [ch][slot], slot = q ^ (ch & 15); B region [ch][slot], slot = q ^ ((ch >> 1) & 7): the piece that
    //      lands in slot s is global piece q = s ^ f(ch).
    int goff[NI];          // float offset from the tile's A / B origin
    int info[NI];          // bit 0: never valid (padding / channel beyond the tensor); B: halo row << 8 | piece column << 16
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int L = i * 256 + tid;
        if (L < A_PIECES) {
            const int ch = L / SAP, s = L - ch * SAP;
            const int q = s ^ (ch & 15);
            const int row = q / RPA, cp = q - row * RPA;
            goff[i] = ch * w.HaWa + SA * row * w.Wa + 4 * cp;
            info[i] = (w.ca0 + ch < CA) ? 0 : 1;
        } else {
            const int Lb = L - A_PIECES;
            const int ch = Lb / kSB, s = Lb - ch * kSB;
            const int q = s ^ ((ch >> 1) & 7);
            const int ri = q / RPB, cq = q - ri * RPB;
            goff[i] = ch * w.HbWb + ri * w.Wb + 4 * cq;
            info[i] = ((q < NPB_ && w.cb0 + ch < CB) ? 0 : 1) | (ri << 8) | (cq << 16);
        }
    }

    // ---- per-lane operand read bases (float indices inside a stage)
    const int wm = wave >> 1, wn = wave & 1;
    w.h = lane >> 5;                                 // tile half (= k slot) of this lane
    const int cha = wm * 32 + (lane & 31);           // A channel of this lane inside the block
    const int chb = wn * 32 + (lane & 31);
    w.a_base = cha * SAP * 4; w.a_f = cha & 15;
    w.b_base = A_PIECES * 4 + chb * kSB * 4; w.b_f = (chb >> 1) & 7;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // tile -> (image, tile row, tile col), advanced by carries (scalar)
    const int tpi = tiles_h * tiles_w;
    int t_img = t0 / tpi, t_rem = t0 - t_img * tpi;
    int t_th = t_rem / tiles_w, t_tw = t_rem - t_th * tiles_w;
    const int d_img = tstep / tpi, d_rem = tstep - d_img * tpi;
    const int d_th = d_rem / tiles_w, d_tw = d_rem - d_th * tiles_w;

    int tile = t0;
    if (tile < tend) wq_issue<CLS, LTW, NI>(lds, w, goff, info, t_img, t_th, t_tw, true);
    int it = 0;
    for (; tile < tend; tile += tstep, ++it) {
        __syncthreads();          // this tile has landed (vmcnt drained in front of the barrier); the other stage is free
        t_tw += d_tw; t_th += d_th; t_img += d_img;          // next tile of this workgroup
        if (t_tw >= tiles_w) { t_tw -= tiles_w; ++t_th; }
        if (t_th >= tiles_h) { t_th -= tiles_h; ++t_img; }
        if (B6)
            wq_tile_b6<CLS, LTW, NI>(lds + (it & 1) * STAGE, lds + ((it + 1) & 1) * STAGE, acc, w, goff, info, t_img, t_th,
                                     t_tw, tile + tstep < tend);
        else
            wq_tile<CLS, LTW, NI>(lds + (it & 1) * STAGE, lds + ((it + 1) & 1) * STAGE, acc, w, goff, info, t_img, t_th,
                                  t_tw, tile + tstep < tend);
    }

    // ---- slab [tap][ca][cb]  (C/D layout: col = lane & 31 -> cb, row -> ca)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        // (global address space spelled out: out of line, `slab` arrives as a flat pointer)
        auto* dst = (__attribute__((address_space(1))) float*)(slab + (size_t)WT::gt(t) * tstride +
                                                               (size_t)(wm * 32) * ldc + wn * 32 + (lane & 31));
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            dst[(size_t)row * ldc] = acc[t][reg];
        }
    }
}

// ---- "row ring" tiles (round 3): the operands are split into their bf16 planes ONCE, on their way into LDS ---------
// wq_tile_b6 above keeps fp32 tiles in LDS and every lane re-splits what it reads: each dy value is split by the two
// waves that share its channel, each x value by two waves times the taps' row overlap (3 x for a 2-row tile), and every
// tap's column shift is a shifted window of a split 10-float halo row: 5.4 VALU instructions per MFMA on a kernel with one
// wave per SIMD (PMC, round 2), matrix pipe 0.41 busy.  Here:
//   * a tile is ONE base row of one image, W = the full image width (32 or 64): there is no column halo -- what lies
//     beyond either end of the row is the layer's zero padding;
//   * x rows live in a ring of four LDS slots that rolls down the images: tile t = image * H + row keeps its x row in slot
//     t & 3 (H is a multiple of 4), stepping to the next tile brings in ONE new x row (it serves three consecutive
//     tiles; the 5 x 5 classes look two rows up or down) and one new dy row; at an image's first / last rows the
//     rows above / below are padding: those reads are
//     redirected (a wave-uniform select of the read address) to a zero piece kept behind every plane of the ring;
//   * global -> registers -> three bf16 planes (hi / mid / lo, 8 pixels = one 16-byte piece) -> LDS: every value is
//     split exactly once, by the thread that loaded it; ~5.5 VALU per value, ~0.8 per MFMA;
//   * the MFMA contraction slot k <-> pixel assignment is free as long as both operands agree, so the x (B) operand is
//     always an ALIGNED octet (one ds_read_b128 per plane) and a tap's column shift s is applied to the dy (A) operand:
//     A[p - s .. p - s + 8) = the aligned A octet funnel-shifted by one bf16 against its left / right neighbour
//     (4 v_alignbit per plane and shifted variant, shared by every tap row: 24 per 54 MFMAs for conv3x3, 36 per 90 for
//     the 15-tap transposed-conv class); A rows end in a zero piece, so the shift past a row end yields the padding.
// LDS images: A plane [64 ch][W/8 data pieces + 1 zero piece] (odd pitch: the 16 channels of a ds_read_b128 lane group
// fall on 16 distinct 16-byte slots), B plane [4 slots][64 ch][W/8 pieces] + the zero piece, piece q of channel c at
// q ^ f(c) (f = (c>>1)&7 at 8 pieces, (c>>2)&3 at 4).
typedef unsigned int gx_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gx_gptr4;
#ifndef GX_WR_ABL
#define GX_WR_ABL 0          // ablation builds (tools/abl_build.sh): 1 no MFMAs, 2 no split / LDS stores, 4 no global loads
#endif

// The second template parameter of the Wr* templates is a CODE: WC = W + 1000 * HF.
// HF (round 4, "k-split"): a 64 x 64 channel block whose A half (HF & 1: <= 32 A channels left) and / or B half (HF & 2) is
// empty -- the 32-channel layers of the gated stacks and the BroadcastDecoder, first layers with 3 / 4 input channels -- used
// to keep two of the four waves multiplying zero rows.  There the waves of such a pair take the SAME 32 channels and every
// other k-group of the tile instead (wave -> k-groups ksel, ksel + KS, ...); the pair's accumulators are summed through LDS
// once per segment.  The tile's MFMA chain is half (a quarter) as long, its staging work unchanged.
// F16 (round 6, WC += 100000): the operands as TWO fp16 pieces of x * 2^e (e per TENSOR from its largest magnitude, handed over by
// the kernels that wrote the tensors: gx_wgq_operand_amax) instead of three bf16 ones -- hi*hi + hi*lo + lo*hi, three MFMAs per 16
// pixels and tap instead of six (findings 40 and 42), two piece planes in LDS instead of three, a 24-VALU split
// per octet instead of 44; the accumulators are scaled back by 2^-(eA + eB) when the slab is written.
template <int CLS, int WC> struct WrGeo {
    static constexpr bool F16 = WC >= 100000;
    static constexpr int HF = (WC / 1000) % 100, W = WC % 1000;
    static constexpr int NPL = F16 ? 2 : 3;                    // piece planes per operand
    static constexpr int NTERM = F16 ? 3 : 6;                  // piece products per fp32 product
    static constexpr int SPQ = F16 ? 6 : 14;                   // split pieces (slots) per octet
    static constexpr int KS = HF == 3 ? 4 : (HF ? 2 : 1);
    using WT = WqTap<CLS>;
    static constexpr int SA = WT::SA, NPB = WT::NPB, NT = WT::NT, NRO = WT::NRO, RO0 = WT::RO0;
    static constexpr int NCOV = WqCols<CLS>::NCOV, CS0 = WqCols<CLS>::CS0;
    // x rows r + DMIN .. r + DMAX feed tile r.  (Two image rows per tile, W == 16 below: tile rows are row PAIRS -- image rows
    // 2 r + RO0 - 1 .. 2 r + 1 + RO0 + NRO - 2 lie in the pairs floor(. / 2); for the 3 x 3 / transposed-conv classes that is the
    // same range as in single rows, for the 5 x 5 halves it is what makes them fit: rows 0-2 -> pairs -1 .. 0, rows 3-4 -> 0 .. 1)
    static constexpr int fdiv2(int v) { return v >= 0 ? v / 2 : -((1 - v) / 2); }
    static constexpr int DMIN = W == 16 ? fdiv2(RO0 - 1) : RO0 - 1, DMAX = W == 16 ? fdiv2(RO0 + NRO - 1) : RO0 + NRO - 2;
    static_assert(DMAX - DMIN + 2 <= 4 && DMAX - DMIN + 1 <= NRO, "the rows in use plus the one being fetched must fit the four ring slots");
    // W = 16: TWO image rows per tile (R2) -- 16-pixel rows of one plane lie back to back in memory, so a pair of them is a
    // 32-pixel row to everything but (a) the dy addresses of the transposed conv (its two dy rows are two rows apart), (b) the
    // column shifts, which must not carry pixels across the seam: the A rows keep a zero piece between the two halves, and (c)
    // the x operand of k-group g (= image row 2 t + g) and tap row offset d: image row 2 t + g + d = half (g + d) & 1 of the
    // ring row t + ((g + d) >> 1).  H is then the number of row PAIRS.
    static constexpr bool R2 = W == 16;
    // STRIPS (round 4): a base row too wide for the ring (conv3x3 at 128 pixels, the transposed conv at 64 base pixels: the
    // 128 x 128 model's large layers) is cut into NS = 2 column strips; a strip of an image is a "virtual image" to the tile
    // sequence (tile = ((image * NS + strip) * H + row), so the ring still rolls down consecutive rows), its rows are PITCH
    // pixels apart in memory, and the only thing that crosses the cut is the column shift of the dy operand: the pixel next
    // to the strip on either side.  Those two values per (channel, parity) -- real data at the cut, padding at the image
    // edge -- are written into bytes 14..15 / 0..1 of the piece in front of / behind the channel's A row (the piece that is
    // all zero for full-width rows): exactly the halves the funnel shifts take from their neighbours (wr_shift_a1).
    static constexpr int NS = (W == 128 || (SA == 2 && W == 64)) ? 2 : 1;
    static constexpr int LNS = NS == 2 ? 1 : 0;
    static constexpr int WE = R2 ? 32 : W / NS;                // pixels per tile row
    static constexpr int PITCH = WE * NS;                      // pixels per base row in memory
    static constexpr int OPR = WE / 8;                         // octets (16-byte pieces) per tile row
    static constexpr int NG = WE / 16 / KS;                    // MFMA k-groups per tile (16 pixels each) AND WAVE
    static_assert((WE / 16) % KS == 0 && !(HF && R2), "k-split needs whole k-groups per wave (and no row pairs)");
    static constexpr int UPT = OPR / 4;                        // (channel, octet) units per thread: 64 * OPR / 256
    static constexpr int CPS = 256 / OPR;                      // channels covered by one unit index
    static constexpr int APITCH = OPR + (R2 ? 3 : 1);          // pieces per channel of an A plane (odd; R2: o0 o1 Z o2 o3 Z Z)
    static constexpr int AGS1 = R2 ? 48 : 32;                  // bytes from a k-group's octets to the next one's in an A row
    static constexpr int AGS = AGS1 * KS;                      // ... to this wave's next one
    static constexpr int A_PLANE = (64 * APITCH + 1) * 16;     // bytes (one leading zero piece)
    static constexpr int A_BUF = NPB * NPL * A_PLANE;          // one A buffer: column parities x planes
    static constexpr int B_ROW = 64 * OPR * 16;                // one x row of one plane
    static constexpr int B_PLANE = 4 * B_ROW + 16;             // four ring slots + the zero piece
    static constexpr int B_ZERO = 4 * B_ROW;                   // offset of the zero piece inside a plane
    static constexpr int RING0 = 2 * A_BUF;                    // byte offset of the ring
    static constexpr int LDS_BYTES = RING0 + NPL * B_PLANE;
    __host__ __device__ static constexpr int fsw(int ch) { return (ch * OPR / 16) & (OPR - 1); }
    // float offset of the dy row of tile row `ra` of image `ia` (H tile rows per image) / of its x row
    // (ia / ib: VIRTUAL image = image * NS + strip)
    __device__ static size_t a_row(int ia, int CA, int ca0, int H, int ra) {
        if (R2) return (((size_t)ia * CA + ca0) * (SA * H) + (size_t)(SA * ra)) * (SA * WE) + WT::PA * (SA * 16);
        return (((size_t)(ia >> LNS) * CA + ca0) * (SA * H) + (size_t)(SA * ra + WT::PA)) * (SA * PITCH) + (ia & (NS - 1)) * (SA * WE);
    }
    __device__ static size_t b_row(int ib, int CB, int cb0, int H, int rb) {
        return (((size_t)(ib >> LNS) * CB + cb0) * H + (size_t)rb) * PITCH + (ib & (NS - 1)) * WE;
    }
    // R2: unit (k-group g, tap row rr) reads k-group b_g of ring row index b_rr
    __host__ __device__ static constexpr int b_e(int g, int rr) { return g + RO0 - 1 + rr; }          // image-row offset from row 2 t
    __host__ __device__ static constexpr int b_so(int g, int rr) { return b_e(g, rr) >= 0 ? b_e(g, rr) / 2 : -((1 - b_e(g, rr)) / 2); }
    __host__ __device__ static constexpr int b_g(int g, int rr) { return R2 ? b_e(g, rr) - 2 * b_so(g, rr) : g; }
    __host__ __device__ static constexpr int b_rr(int g, int rr) { return R2 ? b_so(g, rr) - DMIN : rr; }
};

__device__ __forceinline__ void wr_split8(const float (&v)[8], gx_u32x4& ph, gx_u32x4& pm, gx_u32x4& pl) {
    __bf16 h[8], m[8], l[8];
    wq_split<8>(v, h, m, l);
    gx_bf16x8 vh, vm, vl;
#pragma unroll
    for (int i = 0; i < 8; ++i) { vh[i] = h[i]; vm[i] = m[i]; vl[i] = l[i]; }
    ph = __builtin_bit_cast(gx_u32x4, vh); pm = __builtin_bit_cast(gx_u32x4, vm); pl = __builtin_bit_cast(gx_u32x4, vl);
}

// two fp16 pieces of v * sc (sc a power of two): hi = the top 11 significant bits (a mask: exactly representable, so the residual
// is exact and the conversion of hi cannot round), lo = the residual rounded to nearest -- 22 significant bits, error of random sign
typedef float gx_f32p __attribute__((ext_vector_type(2)));
typedef _Float16 gx_f16p __attribute__((ext_vector_type(2)));
typedef _Float16 gx_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void wr_split_f16_pair(float v0, float v1, float sc, unsigned& hp, unsigned& lp) {
    const float s0 = v0 * sc, s1 = v1 * sc;
    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s0) & 0xFFFFE000u);
    const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s1) & 0xFFFFE000u);
    hp = __builtin_bit_cast(unsigned, __builtin_convertvector(gx_f32p{h0, h1}, gx_f16p));
    lp = __builtin_bit_cast(unsigned, __builtin_convertvector(gx_f32p{s0 - h0, s1 - h1}, gx_f16p));
}
__device__ __forceinline__ void wr_split8_f16(const float (&v)[8], float sc, gx_u32x4& ph, gx_u32x4& pl) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned h, l; wr_split_f16_pair(v[2 * i], v[2 * i + 1], sc, h, l); ph[i] = h; pl[i] = l; }
}

// per-thread constants of a segment
template <int CLS, int W> struct WrT {
    using G = WrGeo<CLS, W>;
    const float* a; const float* b; const float* zeros;
    int CA, CB, ca0, cb0, H, lh, ntot;     // lh = log2 H; ntot = N * H tiles (= x rows) of the layer
    int goffA[G::UPT], goffB[G::UPT];      // float offsets of this thread's units from the row origin
    bool okA[G::UPT], okB[G::UPT];         // channel inside the tensor
    int stA[G::UPT], stB[G::UPT];          // LDS byte offsets of the units' pieces inside a plane / a ring slot
    int a_rd;                              // MFMA loop: byte offset of (A channel, octet h) inside an A plane
    int b_rd[G::NG];                       //            byte offset of (B channel, octet 2g + h) inside a ring slot
    // strips: this thread's seam item (channel tid >> 2, side (tid >> 1) & 1: 0 left / 1 right, column parity tid & 1)
    int goffS, stS, sideS;                 // float offset from the strip's dy row origin; LDS byte offset inside an A buffer
    bool okS;
    float scA, scB;                        // F16: the operands' power-of-two scales 2^eA, 2^eB
};

// the dy value next to a strip: global address select (the zero page beyond the image edge / for idle items)
template <int CLS, int W>
__device__ __forceinline__ float wr_seam_load(const WrT<CLS, W>& w, int ta) {
    using G = WrGeo<CLS, W>;
    const bool live = (unsigned)ta < (unsigned)w.ntot;
    const int ia = ta >> w.lh, ra = ta & (w.H - 1);
    const int strip = ia & (G::NS - 1);
    const bool inside = w.sideS ? strip < G::NS - 1 : strip > 0;
    const float* sp = (live & w.okS & inside) ? w.a + G::a_row(ia, w.CA, w.ca0, w.H, ra) + w.goffS : w.zeros;
    return *(const __attribute__((address_space(1))) float*)sp;
}
// (fp16 pieces: b[0] = hi, b[1] = lo of v * sc, as wr_split_f16_pair forms them)
__device__ __forceinline__ void wr_seam_split_f16(float v, float sc, unsigned short (&b)[3]) {
    const float s0 = v * sc;
    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s0) & 0xFFFFE000u);
    b[0] = __builtin_bit_cast(unsigned short, (_Float16)h0);
    b[1] = __builtin_bit_cast(unsigned short, (_Float16)(s0 - h0));
    b[2] = 0;
}
__device__ __forceinline__ void wr_seam_split(float v, unsigned short (&b)[3]) {
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    b[0] = __builtin_bit_cast(unsigned short, h); b[1] = __builtin_bit_cast(unsigned short, m); b[2] = __builtin_bit_cast(unsigned short, l);
}
template <int CLS, int W>
__device__ __forceinline__ void wr_seam_store(char* lds, const WrT<CLS, W>& w, int abuf, const unsigned short (&b)[3]) {
    using G = WrGeo<CLS, W>;
#pragma unroll
    for (int pl = 0; pl < G::NPL; ++pl) *reinterpret_cast<unsigned short*>(lds + abuf + w.stS + pl * G::A_PLANE) = b[pl];
}

// global -> registers: the dy row of tile ta and the x row of tile tb (a tile index outside [0, ntot) = nothing to load)
template <int CLS, int W>
__device__ __forceinline__ void wr_fetch(const WrT<CLS, W>& w, int ta, int tb,
                                         f32x4 (&pa)[WrGeo<CLS, W>::UPT][2 * WrGeo<CLS, W>::SA],
                                         f32x4 (&pb)[WrGeo<CLS, W>::UPT][2]) {
    using G = WrGeo<CLS, W>;
    constexpr int SA = G::SA;
    const bool a_live = (unsigned)ta < (unsigned)w.ntot, b_live = (unsigned)tb < (unsigned)w.ntot;
    const int ia = ta >> w.lh, ra = ta & (w.H - 1), ib = tb >> w.lh, rb = tb & (w.H - 1);
    const size_t arow = G::a_row(ia, w.CA, w.ca0, w.H, ra);
    const size_t brow = G::b_row(ib, w.CB, w.cb0, w.H, rb);
    const float* ab = w.a + arow;
    const float* bb = w.b + brow;
#pragma unroll
    for (int j = 0; j < G::UPT; ++j) {
        const float* ap = (a_live & w.okA[j]) ? ab + w.goffA[j] : w.zeros;
        const float* bp = (b_live & w.okB[j]) ? bb + w.goffB[j] : w.zeros;
#pragma unroll
        for (int q = 0; q < 2 * SA; ++q) pa[j][q] = ((gx_gptr4)ap)[q];
#pragma unroll
        for (int q = 0; q < 2; ++q) pb[j][q] = ((gx_gptr4)bp)[q];
    }
}

// registers -> bf16 planes -> LDS: A into buffer `abuf` (byte offset), B into ring slot `bslot` (byte offset of the row)
template <int CLS, int W, bool DOA, bool DOB>
__device__ __forceinline__ void wr_store(char* lds, const WrT<CLS, W>& w, int abuf, int bslot,
                                         const f32x4 (&pa)[WrGeo<CLS, W>::UPT][2 * WrGeo<CLS, W>::SA],
                                         const f32x4 (&pb)[WrGeo<CLS, W>::UPT][2]) {
    using G = WrGeo<CLS, W>;
    constexpr int SA = G::SA;
#pragma unroll
    for (int j = 0; j < G::UPT; ++j) {
        if (DOA) {
#pragma unroll
            for (int par = 0; par < G::NPB; ++par) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = pa[j][(SA * i + par) >> 2][(SA * i + par) & 3];   // de-interleave the column parities
                gx_u32x4 ph, pm, pl;
                char* d = lds + abuf + par * G::NPL * G::A_PLANE + w.stA[j];
                if constexpr (G::F16) {
                    wr_split8_f16(v, w.scA, ph, pm);
                    *reinterpret_cast<gx_u32x4*>(d) = ph;
                    *reinterpret_cast<gx_u32x4*>(d + G::A_PLANE) = pm;
                } else {
                wr_split8(v, ph, pm, pl);
                *reinterpret_cast<gx_u32x4*>(d) = ph;
                *reinterpret_cast<gx_u32x4*>(d + G::A_PLANE) = pm;
                *reinterpret_cast<gx_u32x4*>(d + 2 * G::A_PLANE) = pl;
                }
            }
        }
        if (DOB) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = pb[j][i >> 2][i & 3];
            gx_u32x4 ph, pm, pl;
            char* d = lds + G::RING0 + bslot + w.stB[j];
            if constexpr (G::F16) {
                wr_split8_f16(v, w.scB, ph, pm);
                *reinterpret_cast<gx_u32x4*>(d) = ph;
                *reinterpret_cast<gx_u32x4*>(d + G::B_PLANE) = pm;
            } else {
            wr_split8(v, ph, pm, pl);
            *reinterpret_cast<gx_u32x4*>(d) = ph;
            *reinterpret_cast<gx_u32x4*>(d + G::B_PLANE) = pm;
            *reinterpret_cast<gx_u32x4*>(d + 2 * G::B_PLANE) = pl;
            }
        }
    }
}

// ---- one tile, software-pipelined by hand (one wave per SIMD: nothing else covers a stall) ---------------------------
// A v_mfma_f32_32x32x16_bf16 holds the matrix pipe for 32 cycles; while it runs the wave can issue ~7 other instructions.
// So the tile is written as a chain of SLOTS -- one MFMA, then one small PIECE of other work (<= 8 VALU or <= 3 LDS
// instructions) that does not depend on it -- with a scheduling fence behind each (hipcc neither bunches the MFMAs nor
// sinks the pieces; sched_group_barrier pipelines over a block of this size are not followed).  A tile is NG x NRO UNITS
// (k-group g, x row rr) of NCO x 6 MFMAs on one B operand; the pieces of unit u, in slot order:
//     the B operand of unit u + 1 (3 ds_read_b128);
//     at the first unit of group g: the A octets of group g + 1 (one plane per slot: ds_read_b128 + 2 ds_read_b32);
//     at the last unit of group g: the funnel shifts of group g + 1 into the other A variant set (4 v_alignbit each);
//     from unit U0 on: the split of the rows fetched for the NEXT tile, 10 pieces per octet (bf16 pack | residual |
//     pack | residual | pack | three ds_write_b128).
template <int CLS, int W> struct WrRaw { gx_u32x4 c[WrGeo<CLS, W>::NPB][3]; unsigned pv[WrGeo<CLS, W>::NPB][3], nx[WrGeo<CLS, W>::NPB][3]; };
typedef float gx_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gx_bf16x2 __attribute__((ext_vector_type(2)));

template <int CLS, int W> struct WrSched {
    using G = WrGeo<CLS, W>;
    static constexpr int NRO = G::NRO, NG = G::NG, NPB = G::NPB;
    // ONE k-group per wave (k-split forms of the short rows): the tile's last unit works on operand set 0 -- the set the tail's
    // head pieces would fill for the next tile (with an even group count the last group sits in set 1).  There the tail only
    // reads the next tile's raw A octets; its B operand and the funnel shifts follow at the next tile's start (wr_tile_start).
    static constexpr bool NG1 = NG == 1;
    static_assert(NG == 1 || NG % 2 == 0, "the operand sets alternate by k-group");
    static constexpr int NU = NG * NRO;                       // units per tile
    static constexpr int NCO = G::NT / NRO;                   // taps per unit
    static constexpr int NPL = G::NPL;
    static constexpr int NMF = G::NTERM * NCO;                // MFMAs (slots) per unit
    static constexpr int NOCT = G::UPT * (NPB + 1);           // octets this thread splits per tile
    static constexpr int NQ = G::SPQ * NOCT + (G::NS > 1 ? 3 : 0); // split pieces (strips: + the seam value's split | split | stores)
    static constexpr int NFE = 2 * G::UPT + (G::NS > 1 ? 1 : 0);   // fetch pieces (unit 0): the dy / x loads of unit j (+ the seam load)
    // the last unit ends with: the tile's barrier, then the NEXT tile's first operands (B of its unit 0, A octets and
    // funnel shifts of its group 0) under this tile's last MFMAs
    static constexpr int NTAIL = 2 + NPL * NPB + 2 * NPL * NPB;
    __host__ __device__ static constexpr bool more(int u) { return u + 1 < NU; }
    __host__ __device__ static constexpr bool newg(int u) { return u % NRO == 0 && u / NRO + 1 < NG; }
    __host__ __device__ static constexpr bool lastr(int u) { return u % NRO == NRO - 1 && u / NRO + 1 < NG; }
    __host__ __device__ static constexpr int nfix(int u) {
        return (more(u) ? 1 : 0) + (newg(u) ? NPL * NPB : 0) + (lastr(u) ? 2 * NPL * NPB : 0) + (u == 0 ? NFE : 0);
    }
    __host__ __device__ static constexpr int nfree(int u) { return NMF - nfix(u) - (u == NU - 1 ? NTAIL : 0); }
    // split pieces per free slot: 1 unless the tile's MFMA chain is too short for them (the k-split forms of the 32-pixel rows)
    __host__ __device__ static constexpr int pps() {
        int room = 0;
        for (int u = 1; u < NU; ++u) room += nfree(u);
        return room > 0 ? (NQ + room - 1) / room : 1 << 20;
    }
    static constexpr int PPS = pps();
    __host__ __device__ static constexpr int u0() {           // first unit that carries split pieces: as late as they fit
        int u = NU, room = 0;
        while (room < NQ && u > 1) { --u; room += PPS * nfree(u); }
        return u;
    }
    static constexpr int U0 = u0();
    __host__ __device__ static constexpr int qbase(int u) {   // split pieces placed before unit u
        int q = 0;
        for (int v = U0; v < u; ++v) q += PPS * nfree(v);
        return q;
    }
    static_assert(U0 >= 1 && qbase(NU) >= NQ && PPS <= 8, "the split pieces must fit the tile's free slots");
    static_assert(nfree(0) >= 0 && nfree(NRO - 1) >= 0 && nfree(NU - 1) >= 0, "fixed pieces must fit a unit");
};

// what a step needs besides the per-thread constants (wave-uniform)
template <int CLS, int W> struct WrStep {
    int abuf, anxt, bnew;                                     // A buffer of this tile / of the next; ring slot being filled
    int ta, tb;                                               // tiles whose dy / x row is fetched during this step
    int bs[WrGeo<CLS, W>::NRO], nbs[WrGeo<CLS, W>::NRO];      // ring slot (byte offset) of x row rr: this tile / the next
    bool zr[WrGeo<CLS, W>::NRO], nzr[WrGeo<CLS, W>::NRO];     // ... the row is padding
};

template <int CLS, int W> struct WrTileState {
    using G = WrGeo<CLS, W>;
    WrRaw<CLS, W> raw;
    WqB3 av[2][G::NPB][G::NCOV];    // A variants of the current / the next k-group
    WqB3 bq[2];                     // B operand of the current / the next unit
    float r[8];                     // the octet being split: value, then residuals
    unsigned hp[4], mp[4], lp[4];   // its packed bf16 planes
    float sv;                       // strips: the seam value fetched for the next tile
};

__device__ __forceinline__ unsigned wr_pk(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(gx_f32x2{lo, hi}, gx_bf16x2));
}
__device__ __forceinline__ float wr_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float wr_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// split piece Q of the tile: octet Q / 14 (per unit j: the NPB dy octets, then the x octet), step Q % 14 (<= 4 VALU or one
// ds_write_b128 each)
template <int CLS, int W, int Q>
__device__ __forceinline__ void wr_split_piece(char* lds, const WrT<CLS, W>& w, WrTileState<CLS, W>& st, int anxt, int bnew,
                                               const f32x4 (&pa)[WrGeo<CLS, W>::UPT][2 * WrGeo<CLS, W>::SA],
                                               const f32x4 (&pb)[WrGeo<CLS, W>::UPT][2]) {
    using G = WrGeo<CLS, W>;
    constexpr int SA = G::SA, NPB = G::NPB;
    if constexpr (Q >= G::SPQ * G::UPT * (NPB + 1)) {      // strips: the seam value (r[0..1], hp[0..2] are free by now)
        constexpr int step = Q - G::SPQ * G::UPT * (NPB + 1);
        if constexpr (G::F16) {
            if constexpr (step == 0) {
                st.r[0] = st.sv * w.scA;
                st.r[1] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, st.r[0]) & 0xFFFFE000u);
            } else if constexpr (step == 1) {
                st.hp[0] = __builtin_bit_cast(unsigned short, (_Float16)st.r[1]);
                st.hp[1] = __builtin_bit_cast(unsigned short, (_Float16)(st.r[0] - st.r[1]));
            } else {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    *reinterpret_cast<unsigned short*>(lds + anxt + w.stS + pl * G::A_PLANE) = (unsigned short)st.hp[pl];
            }
        } else
        if constexpr (step == 0) {
            const __bf16 h = (__bf16)st.sv;
            st.r[0] = st.sv - (float)h;
            st.hp[0] = __builtin_bit_cast(unsigned short, h);
        } else if constexpr (step == 1) {
            const __bf16 m = (__bf16)st.r[0];
            st.hp[1] = __builtin_bit_cast(unsigned short, m);
            st.hp[2] = __builtin_bit_cast(unsigned short, (__bf16)(st.r[0] - (float)m));
        } else {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<unsigned short*>(lds + anxt + w.stS + pl * G::A_PLANE) = (unsigned short)st.hp[pl];
        }
        return;
    } else {
    constexpr int oi = Q / G::SPQ, step = Q % G::SPQ, j = oi / (NPB + 1), k = oi % (NPB + 1);
    constexpr bool isA = k < NPB;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = isA ? pa[j][(SA * i + k) >> 2][(SA * i + k) & 3] : pb[j][i >> 2][i & 3];
    if constexpr (G::F16) {
        // fp16 pieces: steps 0..3 one PAIR each (scale | mask | residual | two packed conversions: 6 VALU), 4 / 5 the two stores
        if constexpr (step < 4) {
            wr_split_f16_pair(v[2 * step], v[2 * step + 1], isA ? w.scA : w.scB, st.hp[step], st.mp[step]);
        } else {
            char* d = isA ? lds + anxt + k * G::NPL * G::A_PLANE + w.stA[j] : lds + G::RING0 + bnew + w.stB[j];
            constexpr int ps = isA ? G::A_PLANE : G::B_PLANE;
            const unsigned* src = step == 4 ? st.hp : st.mp;
            *reinterpret_cast<gx_u32x4*>(d + (step - 4) * ps) = gx_u32x4{src[0], src[1], src[2], src[3]};
        }
    } else
    if (step == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st.hp[i] = wr_pk(v[2 * i], v[2 * i + 1]);
    } else if (step >= 1 && step <= 4) {
        constexpr int i = step - 1;
        st.r[2 * i] = v[2 * i] - wr_lo(st.hp[i]); st.r[2 * i + 1] = v[2 * i + 1] - wr_hi(st.hp[i]);
    } else if (step == 5) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st.mp[i] = wr_pk(st.r[2 * i], st.r[2 * i + 1]);
    } else if (step >= 6 && step <= 9) {
        constexpr int i = step - 6;
        st.r[2 * i] -= wr_lo(st.mp[i]); st.r[2 * i + 1] -= wr_hi(st.mp[i]);
    } else if (step == 10) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st.lp[i] = wr_pk(st.r[2 * i], st.r[2 * i + 1]);
    } else {
        char* d = isA ? lds + anxt + k * 3 * G::A_PLANE + w.stA[j] : lds + G::RING0 + bnew + w.stB[j];
        constexpr int ps = isA ? G::A_PLANE : G::B_PLANE;
        const unsigned* src = step == 11 ? st.hp : (step == 12 ? st.mp : st.lp);
        *reinterpret_cast<gx_u32x4*>(d + (step - 11) * ps) = gx_u32x4{src[0], src[1], src[2], src[3]};
    }
    }
}

// one (parity, plane) of the A octets of group g: the aligned octet and its neighbours' adjacent pairs
template <int CLS, int W>
__device__ __forceinline__ void wr_read_a1(const char* lds, const WrT<CLS, W>& w, int abuf, int g, int par, int pl, WrRaw<CLS, W>& r) {
    using G = WrGeo<CLS, W>;
    const char* p = lds + abuf + (par * G::NPL + pl) * G::A_PLANE + w.a_rd + g * G::AGS;
    r.c[par][pl] = *reinterpret_cast<const gx_u32x4*>(p);
    r.pv[par][pl] = *reinterpret_cast<const unsigned*>(p - 4);
    r.nx[par][pl] = *reinterpret_cast<const unsigned*>(p + 16);
}

// variants by column offset co = s - CS0 (conv3x3 / transposed conv: co 1 aligned, co 2 (s = +1) shifted right, co 0
// (s = -1) shifted left; 5 x 5: co 2 aligned, and s = +-2 are whole-register shifts); piece = (parity, plane, which side)
__device__ __forceinline__ gx_bf16x8 wr_bf(const gx_u32x4& v) { return __builtin_bit_cast(gx_bf16x8, v); }
template <int CLS, int W>
__device__ __forceinline__ void wr_shift_a1(const WrRaw<CLS, W>& r, WqB3 (&av)[WrGeo<CLS, W>::NPB][WrGeo<CLS, W>::NCOV], int par,
                                            int pl, int right) {
    using G = WrGeo<CLS, W>;
    constexpr int Z = -G::CS0;              // index of the aligned variant (s = 0)
    const gx_u32x4 c = r.c[par][pl];
    const unsigned pv = r.pv[par][pl], nx = r.nx[par][pl];
    auto put = [&](int co, const gx_u32x4& v) {
        WqB3& d = av[par][co];
        if (pl == 0) d.h = wr_bf(v); else if (pl == 1) d.m = wr_bf(v); else d.l = wr_bf(v);
    };
    if (right) {
        // s = +1: A[p - 1 ..]: the aligned octet funnel-shifted against its left neighbour's last pair
        put(Z + 1, gx_u32x4{__builtin_amdgcn_alignbit(c[0], pv, 16), __builtin_amdgcn_alignbit(c[1], c[0], 16),
                            __builtin_amdgcn_alignbit(c[2], c[1], 16), __builtin_amdgcn_alignbit(c[3], c[2], 16)});
        put(Z, c);
        if (G::NCOV == 5) put(Z + 2, gx_u32x4{pv, c[0], c[1], c[2]});         // s = +2: one pair = one register earlier
    } else {
        put(Z - 1, gx_u32x4{__builtin_amdgcn_alignbit(c[1], c[0], 16), __builtin_amdgcn_alignbit(c[2], c[1], 16),
                            __builtin_amdgcn_alignbit(c[3], c[2], 16), __builtin_amdgcn_alignbit(nx, c[3], 16)});
        if (G::NCOV == 5) put(Z - 2, gx_u32x4{c[1], c[2], c[3], nx});         // s = -2
    }
}

template <int CLS, int W>
__device__ __forceinline__ void wr_read_b(const char* lds, const WrT<CLS, W>& w, int g, int bs, bool zr, WqB3& b3) {
    using G = WrGeo<CLS, W>;
    const char* p = lds + G::RING0 + (zr ? G::B_ZERO : bs + w.b_rd[g]);
    b3.h = wr_bf(*reinterpret_cast<const gx_u32x4*>(p));
    b3.m = wr_bf(*reinterpret_cast<const gx_u32x4*>(p + G::B_PLANE));
    if constexpr (!G::F16) b3.l = wr_bf(*reinterpret_cast<const gx_u32x4*>(p + 2 * G::B_PLANE));
}

// fetch piece i: the dy (i even) / x (i odd) loads of load unit i / 2
template <int CLS, int W, int I>
__device__ __forceinline__ void wr_fetch_piece(const WrT<CLS, W>& w, int ta, int tb,
                                               f32x4 (&pa)[WrGeo<CLS, W>::UPT][2 * WrGeo<CLS, W>::SA],
                                               f32x4 (&pb)[WrGeo<CLS, W>::UPT][2], float& sv) {
    using G = WrGeo<CLS, W>;
    constexpr int SA = G::SA, j = I / 2;
    if constexpr (I == 2 * G::UPT) {
        sv = wr_seam_load<CLS, W>(w, ta);
    } else if (I % 2 == 0) {
        const bool live = (unsigned)ta < (unsigned)w.ntot;
        const int ia = ta >> w.lh, ra = ta & (w.H - 1);
        const size_t arow = G::a_row(ia, w.CA, w.ca0, w.H, ra);
        const float* ap = (live & w.okA[j]) ? w.a + arow + w.goffA[j] : w.zeros;
#pragma unroll
        for (int q = 0; q < 2 * SA; ++q) pa[j][q] = ((gx_gptr4)ap)[q];
    } else {
        const bool live = (unsigned)tb < (unsigned)w.ntot;
        const int ib = tb >> w.lh, rb = tb & (w.H - 1);
        const size_t brow = G::b_row(ib, w.CB, w.cb0, w.H, rb);
        const float* bp = (live & w.okB[j]) ? w.b + brow + w.goffB[j] : w.zeros;
#pragma unroll
        for (int q = 0; q < 2; ++q) pb[j][q] = ((gx_gptr4)bp)[q];
    }
}

// the first operands of a tile: B of unit 0, A octets + funnel shifts of group 0 -- as NTAIL - 1 pieces (I = 0 .. )
template <int CLS, int W, int I>
__device__ __forceinline__ void wr_head_piece(const char* lds, const WrT<CLS, W>& w, int abuf, const int (&bs)[WrGeo<CLS, W>::NRO],
                                              const bool (&zr)[WrGeo<CLS, W>::NRO], WrTileState<CLS, W>& st) {
    constexpr int NPB = WrGeo<CLS, W>::NPB, NPL = WrGeo<CLS, W>::NPL;
    if (I == 0) wr_read_b<CLS, W>(lds, w, WrGeo<CLS, W>::b_g(0, 0), bs[WrGeo<CLS, W>::b_rr(0, 0)], zr[WrGeo<CLS, W>::b_rr(0, 0)], st.bq[0]);
    else if (I < 1 + NPL * NPB) wr_read_a1<CLS, W>(lds, w, abuf, 0, (I - 1) / NPL, (I - 1) % NPL, st.raw);
    else wr_shift_a1<CLS, W>(st.raw, st.av[0], (I - 1 - NPL * NPB) / (2 * NPL), ((I - 1 - NPL * NPB) % (2 * NPL)) / 2, (I - 1 - NPL * NPB) % 2);
}

// the split pieces Q .. Q + N - 1 (those that exist)
template <int CLS, int W, int Q, int N>
__device__ __forceinline__ void wr_split_pieces(char* lds, const WrT<CLS, W>& w, WrTileState<CLS, W>& st, int anxt, int bnew,
                                                const f32x4 (&pa)[WrGeo<CLS, W>::UPT][2 * WrGeo<CLS, W>::SA],
                                                const f32x4 (&pb)[WrGeo<CLS, W>::UPT][2]) {
    if constexpr (Q >= 0 && Q < WrSched<CLS, W>::NQ) wr_split_piece<CLS, W, Q>(lds, w, st, anxt, bnew, pa, pb);
    if constexpr (N > 1) wr_split_pieces<CLS, W, Q + 1, N - 1>(lds, w, st, anxt, bnew, pa, pb);
}

// slot M of unit U: its MFMA, then its piece
template <int CLS, int W, int U, int M>
__device__ __forceinline__ void wr_slot(char* lds, const WrT<CLS, W>& w, const WrStep<CLS, W>& sp,
                                        f32x16 (&acc)[WrGeo<CLS, W>::NT], WrTileState<CLS, W>& st,
                                        f32x4 (&pa)[WrGeo<CLS, W>::UPT][2 * WrGeo<CLS, W>::SA],
                                        f32x4 (&pb)[WrGeo<CLS, W>::UPT][2]) {
    using G = WrGeo<CLS, W>;
    using S = WrSched<CLS, W>;
    using WT = typename G::WT;
    constexpr int NRO = G::NRO, NPB = G::NPB;
    constexpr int g = U / NRO, rr = U % NRO, as = g & 1;
    // ---- the MFMA: term M / NCO (small terms first, as wq_mma6) of tap M % NCO of x row rr -- term-major: consecutive
    //      MFMAs accumulate into different registers (a dependent one waits for its predecessor's write-back)
    {
        int tap = -1, seen = 0;
#pragma unroll
        for (int q = 0; q < WT::NT; ++q)
            if (WT::ro(q) - G::RO0 == rr) { if (seen == M % S::NCO) tap = q; ++seen; }
        const WqB3& a3 = st.av[as][WT::pb(tap)][WT::co(tap)];
        const WqB3& b3 = st.bq[U & 1];
        constexpr int term = M / S::NCO;
        // bf16 pieces: m*m, l*h, h*l, m*h, h*m, h*h; fp16 pieces (plane 1 = lo, kept in .m): lo*hi, hi*lo, hi*hi
        const gx_bf16x8 ao = G::F16 ? (term == 0 ? a3.m : a3.h) : (term == 0 ? a3.m : (term == 1 ? a3.l : (term == 3 ? a3.m : a3.h)));
        const gx_bf16x8 bo = G::F16 ? (term == 1 ? b3.m : b3.h) : (term == 0 ? b3.m : (term == 2 ? b3.l : (term == 4 ? b3.m : b3.h)));
#if !(GX_WR_ABL & 1)
        if constexpr (G::F16)
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gx_f16x8, ao), __builtin_bit_cast(gx_f16x8, bo), acc[tap], 0, 0, 0);
        else
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao, bo, acc[tap], 0, 0, 0);
#else
        acc[tap][M % 16] += __builtin_bit_cast(float, __builtin_bit_cast(gx_u32x4, ao)[0] ^ __builtin_bit_cast(gx_u32x4, bo)[1]);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the piece
    constexpr int NPL = G::NPL;
    constexpr int n_b = S::more(U) ? 1 : 0, n_ra = S::newg(U) ? NPL * NPB : 0, n_sh = S::lastr(U) ? 2 * NPL * NPB : 0;
    constexpr int n_fe = U == 0 ? S::NFE : 0;
    constexpr int tail0 = U == S::NU - 1 ? S::NMF - S::NTAIL : S::NMF;      // first tail slot
    if constexpr (M >= tail0) {
        if constexpr (M == tail0) __syncthreads();         // the rows of the next tile are in LDS; this tile's reads are done
#if !(GX_WR_ABL & 8)
        else if constexpr (!S::NG1 || (M - tail0 - 1 >= 1 && M - tail0 - 1 < 1 + NPL * NPB))
            wr_head_piece<CLS, W, M - tail0 - 1>(lds, w, sp.anxt, sp.nbs, sp.nzr, st);
#endif
    }
#if GX_WR_ABL & 8
    else if constexpr (M < n_b + n_ra + n_sh) {}
#endif
    else if constexpr (M < n_b) {
        constexpr int g1 = (U + 1) / NRO, r1 = (U + 1) % NRO;
        wr_read_b<CLS, W>(lds, w, G::b_g(g1, r1), sp.bs[G::b_rr(g1, r1)], sp.zr[G::b_rr(g1, r1)], st.bq[(U + 1) & 1]);
    } else if constexpr (M < n_b + n_ra) {
        wr_read_a1<CLS, W>(lds, w, sp.abuf, g + 1, (M - n_b) / NPL, (M - n_b) % NPL, st.raw);
    } else if constexpr (M < n_b + n_ra + n_sh) {
        wr_shift_a1<CLS, W>(st.raw, st.av[as ^ 1], (M - n_b - n_ra) / (2 * NPL), ((M - n_b - n_ra) % (2 * NPL)) / 2, (M - n_b - n_ra) % 2);
    } else if constexpr (M < n_b + n_ra + n_sh + n_fe) {
#if !(GX_WR_ABL & 4)
        wr_fetch_piece<CLS, W, M - n_b - n_ra - n_sh>(w, sp.ta, sp.tb, pa, pb, st.sv);
#endif
    } else if constexpr (U >= S::U0) {
        constexpr int q = S::qbase(U) + (M - S::nfix(U)) * S::PPS;
#if !(GX_WR_ABL & 2)
        wr_split_pieces<CLS, W, q, S::PPS>(lds, w, st, sp.anxt, sp.bnew, pa, pb);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M + 1 < S::NMF) wr_slot<CLS, W, U, M + 1>(lds, w, sp, acc, st, pa, pb);
    else if constexpr (U + 1 < S::NU) wr_slot<CLS, W, U + 1, 0>(lds, w, sp, acc, st, pa, pb);
}

template <int CLS, int W, int I>
__device__ __forceinline__ void wr_head(const char* lds, const WrT<CLS, W>& w, int abuf, const int (&bs)[WrGeo<CLS, W>::NRO],
                                        const bool (&zr)[WrGeo<CLS, W>::NRO], WrTileState<CLS, W>& st) {
    constexpr int NPB = WrGeo<CLS, W>::NPB, NPL = WrGeo<CLS, W>::NPL;
    if constexpr (!WrSched<CLS, W>::NG1 || (I >= 1 && I < 1 + NPL * NPB)) wr_head_piece<CLS, W, I>(lds, w, abuf, bs, zr, st);
    if constexpr (I + 2 < WrSched<CLS, W>::NTAIL) wr_head<CLS, W, I + 1>(lds, w, abuf, bs, zr, st);
}
// NG1: what the tail left out -- B of unit 0 and the funnel shifts of group 0, in front of the tile's first MFMA
template <int CLS, int W, int I>
__device__ __forceinline__ void wr_tile_start(const char* lds, const WrT<CLS, W>& w, int abuf, const int (&bs)[WrGeo<CLS, W>::NRO],
                                              const bool (&zr)[WrGeo<CLS, W>::NRO], WrTileState<CLS, W>& st) {
    constexpr int NPB = WrGeo<CLS, W>::NPB, NPL = WrGeo<CLS, W>::NPL;
    if constexpr (I == 0 || I >= 1 + NPL * NPB) wr_head_piece<CLS, W, I>(lds, w, abuf, bs, zr, st);
    if constexpr (I + 2 < WrSched<CLS, W>::NTAIL) wr_tile_start<CLS, W, I + 1>(lds, w, abuf, bs, zr, st);
}

// One SEGMENT of row-ring tiles: the tiles t0 .. t1 - 1 (tile = image * H + base row) of one 64 x 64 channel block,
// accumulated in registers and written as one slab [tap][64][64].
template <int CLS, int W>
__device__ __forceinline__ void wr_segment(const float* a, const float* b, const float* zeros, char* lds, int N, int CA,
                                           int CB, int ca0, int cb0, int H, int t0, int t1, float* slab,
                                           const float* amax2 = nullptr) {
    using G = WrGeo<CLS, W>;
    using WT = typename G::WT;
    constexpr int NT = G::NT, NRO = G::NRO, RO0 = G::RO0, OPR = G::OPR, SA = G::SA, WE = G::WE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (G::R2) H >>= 1;                     // tile rows = pairs of image rows
    WrT<CLS, W> w;
    w.a = a; w.b = b; w.zeros = zeros; w.CA = CA; w.CB = CB; w.ca0 = ca0; w.cb0 = cb0; w.H = H;
    w.lh = 31 - __builtin_clz(H); w.ntot = N * G::NS * H;
    // F16: amax2 = {max |dy|, max |x|} of the two operand TENSORS (uniform loads); x * 2^e with max |x| * 2^e in [2^14, 2^15)
    int f16_e = 0;
    w.scA = w.scB = 1.f;
    if constexpr (G::F16) {
        const int ea = gx_f16_scale_exp(amax2[0]), eb = gx_f16_scale_exp(amax2[1]);
        w.scA = ldexpf(1.f, ea); w.scB = ldexpf(1.f, eb);
        f16_e = -(ea + eb);
    }
#pragma unroll
    for (int j = 0; j < G::UPT; ++j) {
        const int ch = tid / OPR + j * G::CPS, o = tid % OPR;
        // (R2: the second image row's dy starts SA image rows of SA * 16 floats further on, not right behind the first's)
        w.goffA[j] = ch * (SA * H) * (SA * G::PITCH) + SA * 8 * o + (G::R2 && o >= 2 ? 16 * SA * (SA - 1) : 0);
        w.goffB[j] = ch * H * G::PITCH + 8 * o;
        w.okA[j] = ca0 + ch < CA;
        w.okB[j] = cb0 + ch < CB;
        w.stA[j] = (1 + ch * G::APITCH + o + (G::R2 && o >= 2 ? 1 : 0)) * 16;
        w.stB[j] = (ch * OPR + (o ^ G::fsw(ch))) * 16;
    }
    {
        const int ch = tid >> 2, side = (tid >> 1) & 1, par = tid & 1;
        w.sideS = side;
        w.okS = G::NS > 1 && par < G::NPB && ca0 + ch < CA;
        w.goffS = ch * (SA * H) * (SA * G::PITCH) + (side ? SA * WE + par : par - SA);
        // left of channel ch: the last bf16 of the piece in front of its row; right: the first bf16 of the piece behind it;
        // idle items (no second parity): bytes 4..5 of the leading piece, which nothing reads
        w.stS = par < G::NPB ? par * G::NPL * G::A_PLANE + (side ? (1 + ch * G::APITCH + OPR) * 16 : ch * G::APITCH * 16 + 14) : 4;
    }
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
    // k-split: which of the KS interleaved k-group sets this wave takes; the waves of a set share their channels
    const int ksel = G::HF == 1 ? wm : (G::HF == 2 ? wn : (G::HF == 3 ? wave : 0));
    const int cha = ((G::HF & 1) ? 0 : wm * 32) + (lane & 31), chb = ((G::HF & 2) ? 0 : wn * 32) + (lane & 31);
    w.a_rd = (1 + cha * G::APITCH + h) * 16 + ksel * G::AGS1;
#pragma unroll
    for (int g = 0; g < G::NG; ++g) w.b_rd[g] = (chb * OPR + ((2 * (g * G::KS + ksel) + h) ^ G::fsw(chb))) * 16;

    // zero pieces: the leading one and one behind every channel's row of the A planes (both buffers), one per ring plane
    // (R2: the seam piece and the two behind the row as well -- the whole A buffers)
    if (G::R2) {
        for (int i = tid; i < 2 * G::A_BUF / 16; i += 256) *reinterpret_cast<gx_u32x4*>(lds + i * 16) = gx_u32x4{0u, 0u, 0u, 0u};
        __syncthreads();      // (this fill covers the data pieces too: it must land before the prologue's stores of other threads)
    }
    for (int i = tid; i < 2 * G::NPB * G::NPL * 65 + G::NPL; i += 256) {
        const int plane = i / 65, k = i - plane * 65;
        char* d = plane < 2 * G::NPB * G::NPL ? lds + plane * G::A_PLANE + (k == 0 ? 0 : k * G::APITCH) * 16
                                              : lds + G::RING0 + k * G::B_PLANE + G::B_ZERO;
        *reinterpret_cast<gx_u32x4*>(d) = gx_u32x4{0u, 0u, 0u, 0u};
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    f32x4 pa[G::UPT][2 * SA], pb[G::UPT][2];
    // prologue: the x rows of tiles t0 + DMIN .. t0 + DMAX (a neighbour in another image is never read by tile t0, but
    // the ones behind serve tile t0 + 1) and the dy row of t0
    wr_fetch<CLS, W>(w, t0, t0 + G::DMIN, pa, pb);
    if constexpr (G::NS > 1) {
        __syncthreads();      // (the seam bytes lie inside the pieces the fill above zeroes)
        unsigned short sb[3];
        if constexpr (G::F16) wr_seam_split_f16(wr_seam_load<CLS, W>(w, t0), w.scA, sb);
        else wr_seam_split(wr_seam_load<CLS, W>(w, t0), sb);
        wr_seam_store<CLS, W>(lds, w, 0, sb);
    }
    wr_store<CLS, W, true, true>(lds, w, 0, ((t0 + G::DMIN) & 3) * G::B_ROW, pa, pb);
#pragma unroll 1
    for (int k = G::DMIN + 1; k <= G::DMAX; ++k) {
        wr_fetch<CLS, W>(w, -1, t0 + k, pa, pb);
        wr_store<CLS, W, false, true>(lds, w, 0, ((t0 + k) & 3) * G::B_ROW, pa, pb);
    }
    __syncthreads();
    WrStep<CLS, W> sp;
    WrTileState<CLS, W> st;
    auto rows = [&](int t, int (&bs)[NRO], bool (&zr)[NRO]) {
        const int r = t & (H - 1);
#pragma unroll
        for (int rr = 0; rr < NRO; ++rr) {
            const int dr = G::DMIN + rr;                    // x row (row pair) r + dr; entries past DMAX are never read
            bs[rr] = ((t + dr) & 3) * G::B_ROW;
            zr[rr] = (unsigned)(r + dr) >= (unsigned)H;
        }
    };
    rows(t0, sp.nbs, sp.nzr);
    wr_head<CLS, W, 0>(lds, w, 0, sp.nbs, sp.nzr, st);      // the first tile's first operands (later ones: the previous tile's tail)
    sp.anxt = 0;
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        sp.abuf = sp.anxt; sp.anxt = G::A_BUF - sp.abuf;
        sp.bnew = ((t + 1 + G::DMAX) & 3) * G::B_ROW;
        sp.ta = t + 1 < t1 ? t + 1 : -1; sp.tb = t + 1 + G::DMAX;
#pragma unroll
        for (int rr = 0; rr < NRO; ++rr) { sp.bs[rr] = sp.nbs[rr]; sp.zr[rr] = sp.nzr[rr]; }
        rows(t + 1, sp.nbs, sp.nzr);
        if constexpr (WrSched<CLS, W>::NG1) wr_tile_start<CLS, W, 0>(lds, w, sp.abuf, sp.bs, sp.zr, st);
        wr_slot<CLS, W, 0, 0>(lds, w, sp, acc, st, pa, pb);
    }

    if constexpr (G::HF != 0) {
        // k-split: sum the accumulators of the waves that share a 32 x 32 block (fixed order), through LDS: [sender][tap][reg][lane]
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll 1
        for (int round = 0; round < (G::HF == 3 ? 2 : 1); ++round) {
            // HF 1: wave 2 + wn -> wn; HF 2: 2 wm + 1 -> 2 wm; HF 3: round 0: 1 -> 0, 3 -> 2; round 1: 2 -> 0
            bool send, recv; int slot;
            if (G::HF == 1) { send = wm == 1; recv = wm == 0; slot = wn; }
            else if (G::HF == 2) { send = wn == 1; recv = wn == 0; slot = wm; }
            else if (round == 0) { send = (wave & 1) == 1; recv = (wave & 1) == 0; slot = wave >> 1; }
            else { send = wave == 2; recv = wave == 0; slot = 0; }
            __syncthreads();
            if (send) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) red[((slot * NT + t) * 16 + reg) * 64 + lane] = acc[t][reg];
            }
            __syncthreads();
            if (recv) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) acc[t][reg] += red[((slot * NT + t) * 16 + reg) * 64 + lane];
            }
        }
        const bool owner = G::HF == 1 ? wm == 0 : (G::HF == 2 ? wn == 0 : wave == 0);
        if (!owner) return;
    }
    // ---- slab [tap][ca][cb]  (C/D layout: col = lane & 31 -> cb, row -> ca)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        auto* dst = (__attribute__((address_space(1))) float*)(slab + (size_t)WT::gt(t) * 4096 + (size_t)(wm * 32) * 64 +
                                                               wn * 32 + (lane & 31));
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            dst[(size_t)row * 64] = G::F16 ? ldexpf(acc[t][reg], f16_e) : acc[t][reg];
        }
    }
}

template <int CLS, int W>
__device__ __attribute__((noinline)) void wr_segment_call(const float* a, const float* b, const float* zeros, int N, int CA,
                                                          int CB, int ca0, int cb0, int H, int t0, int t1, float* slab,
                                                          const float* amax2 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    wr_segment<CLS, W>(a, b, zeros, reinterpret_cast<char*>(lds_dyn), N, CA, CB, ca0, cb0, H, t0, t1, slab, amax2);
}

// ---- grouped launch: the jobs of one (class, tile width), every workgroup one strided segment
template <int CLS, int LTW, bool B6>
__global__ void __launch_bounds__(256, 1)
wgq_kernel(const WqTable tab, const float* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // which job, channel block and split (wave-uniform scalar work)
    int j = 0;
#pragma unroll 1
    for (int q = 1; q < tab.njobs; ++q) j = (int)blockIdx.x >= tab.job[q].wg_begin ? q : j;
    const WqJob& jb = tab.job[j];
    const int local = (int)blockIdx.x - jb.wg_begin;
    const int nsp = jb.nsplit;
    const int blk = local / nsp, sp = local - blk * nsp;
    const int ca0 = (blk / jb.nbt) * 64, cb0 = (blk % jb.nbt) * 64;
    wq_segment<CLS, LTW, B6>(jb.a, jb.b, zeros, lds, jb.CA, jb.CB, ca0, cb0, jb.Hb, jb.Wb, jb.tiles_h, jb.tiles_w, sp, nsp,
                         jb.ntiles, jb.partial + ((size_t)sp * jb.Ttot * jb.CApad + ca0) * jb.CBpad + cb0, jb.CBpad,
                         jb.CApad * jb.CBpad);
}

// ---- stream-K launch: ONE grid of G workgroups for every queued layer of every class and tile width.  The layers'
// (channel block, tile) work items, weighted by a per-variant cost, form one line of U units; workgroup w takes the
// units [U w / G, U (w + 1) / G), rounded down to tiles: a contiguous run of tiles that crosses into the next block /
// layer at most a few times, one slab per (workgroup, block) it touches.  ~G + #blocks slabs per training step instead
// of (launches x 256): the slab traffic and the reduction shrink 8-fold, and eight launch prologues / epilogues go.
struct WsJob {
    const float* a; const float* b; float* partial;      // partial: this block's region [slab][Ttot][64][64]
    long long ubegin;                                    // first unit of this block on the line
    int CA, CB, ca0, cb0, Hb, Wb, tiles_h, tiles_w, ntiles, Ttot;
    int variant;                                         // class * 3 + (5 - log2 tile width); + 9: on the fp32 matrix pipe;
                                                         // 18..21: row-ring tiles (conv3x3 W 64 / 32, transposed conv rows 0 / 1 at W 32)
                                                         // 22..25: row-ring tiles of the 5 x 5 stride-1 conv (rows 0-2 W 64 / 32, rows 3-4 W 64 / 32)
    int cost;                                            // units per tile
    int w_first;                                         // first workgroup with tiles of this block (slab 0)
    int N;                                               // images (the row-ring variants bound their virtual rows with it)
    const float* amax2;                                  // variants 128 + ..: {max |a|, max |b|} of the operand tensors (fp16 pieces)
};
constexpr int kMaxSJobs = 36;                            // 36 x 96 B: the table travels as a kernel argument
struct WsTable { long long U; int njobs, G; long long* times; WsJob job[kMaxSJobs]; };   // times: GENESIS_WGQ_TIMES (NULL otherwise)

// out-of-line copy of a variant for the stream-K kernel: nine inlined bodies in one function cost hipcc's register
// allocator ~1.5 KB of scratch per lane; as separate functions each keeps the allocation of its own grouped kernel.
// The LDS stage pointer is re-derived from the dynamic LDS symbol inside (a pointer parameter would arrive as a flat
// pointer and turn every ds_read into a flat load).
template <int CLS, int LTW, bool B6>
__device__ __attribute__((noinline)) void wq_segment_call(const float* a, const float* b, const float* zeros, int CA, int CB,
                                                          int ca0, int cb0, int Hb, int Wb, int tiles_h, int tiles_w, int t0,
                                                          int t1, float* slab) {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    wq_segment<CLS, LTW, B6>(a, b, zeros, lds_dyn, CA, CB, ca0, cb0, Hb, Wb, tiles_h, tiles_w, t0, 1, t1, slab, 64, 4096);
}

__host__ __device__ inline void ws_locate(const WsTable& tab, long long B, int* j_out, int* t_out) {
    if (B >= tab.U) { *j_out = tab.njobs; *t_out = 0; return; }
    int j = 0;
    for (int q = 1; q < tab.njobs; ++q) j = B >= tab.job[q].ubegin ? q : j;
    *j_out = j;
    *t_out = (int)((unsigned)(B - tab.job[j].ubegin) / (unsigned)tab.job[j].cost);
}

__global__ void __launch_bounds__(256, 1)
wgq_stream_kernel(const WsTable tab, const float* __restrict__ zeros) {
    const int wg = blockIdx.x;
    const long long t_begin = tab.times ? (long long)__builtin_amdgcn_s_memrealtime() : 0;      // 100 MHz
    int js, ts, je, te;
    ws_locate(tab, tab.U * wg / tab.G, &js, &ts);
    ws_locate(tab, wg + 1 == tab.G ? tab.U : tab.U * (wg + 1) / tab.G, &je, &te);
#pragma unroll 1
    for (int j = js; j <= je && j < tab.njobs; ++j) {
        const WsJob jb = tab.job[j];          // (a copy: 96 bytes of kernel-argument memory into scalar registers)
        const int t0 = j == js ? ts : 0, t1 = j == je ? te : jb.ntiles;
        if (t0 >= t1) continue;
        float* slab = jb.partial + (size_t)(wg - jb.w_first) * jb.Ttot * 4096;
#define GX_WS_CASE(V_, CLS_, LTW_)                                                                                  \
        case V_:                                                                                                    \
            wq_segment_call<CLS_, LTW_, true>(jb.a, jb.b, zeros, jb.CA, jb.CB, jb.ca0, jb.cb0, jb.Hb, jb.Wb,       \
                                              jb.tiles_h, jb.tiles_w, t0, t1, slab);                                \
            break;                                                                                                  \
        case V_ + 9:                                                                                                \
            wq_segment_call<CLS_, LTW_, false>(jb.a, jb.b, zeros, jb.CA, jb.CB, jb.ca0, jb.cb0, jb.Hb, jb.Wb,      \
                                               jb.tiles_h, jb.tiles_w, t0, t1, slab);                               \
            break;
#define GX_WR_CASE(V_, CLS_, W_)                                                                                    \
        case V_:                                                                                                    \
            wr_segment_call<CLS_, W_>(jb.a, jb.b, zeros, jb.N, jb.CA, jb.CB, jb.ca0, jb.cb0, jb.Hb, t0, t1, slab);  \
            break;
        // the same tiles on TWO fp16 pieces per operand (variant + 128: both operands' maxima are known)
#define GX_WF_CASE(V_, CLS_, W_)                                                                                    \
        case 128 + V_:                                                                                              \
            wr_segment_call<CLS_, 100000 + W_>(jb.a, jb.b, zeros, jb.N, jb.CA, jb.CB, jb.ca0, jb.cb0, jb.Hb, t0, t1, slab, jb.amax2); \
            break;
        switch (jb.variant) {
            GX_WS_CASE(0, WQ_C3, 5) GX_WS_CASE(1, WQ_C3, 4) GX_WS_CASE(2, WQ_C3, 3)
            GX_WS_CASE(3, WQ_DR0, 5) GX_WS_CASE(4, WQ_DR0, 4) GX_WS_CASE(5, WQ_DR0, 3)
            GX_WS_CASE(6, WQ_DR1, 5) GX_WS_CASE(7, WQ_DR1, 4) GX_WS_CASE(8, WQ_DR1, 3)
            GX_WR_CASE(18, WQ_C3, 64) GX_WR_CASE(19, WQ_C3, 32) GX_WR_CASE(20, WQ_DR0, 32) GX_WR_CASE(21, WQ_DR1, 32)
            GX_WR_CASE(22, WQ_C5A, 64) GX_WR_CASE(23, WQ_C5A, 32) GX_WR_CASE(24, WQ_C5B, 64) GX_WR_CASE(25, WQ_C5B, 32)
            GX_WR_CASE(26, WQ_C3, 16) GX_WR_CASE(27, WQ_DR0, 16) GX_WR_CASE(28, WQ_DR1, 16)
            GX_WR_CASE(29, WQ_C3, 128) GX_WR_CASE(30, WQ_DR0, 64) GX_WR_CASE(31, WQ_DR1, 64)
            GX_WR_CASE(32, WQ_C5A, 16) GX_WR_CASE(33, WQ_C5B, 16)
            // k-split forms (variant + 32 * HF, width code + 1000 * HF): HF 1 = A half empty, 2 = B half empty, 3 = both
            GX_WR_CASE(32 + 18, WQ_C3, 1064) GX_WR_CASE(32 + 19, WQ_C3, 1032) GX_WR_CASE(32 + 20, WQ_DR0, 1032) GX_WR_CASE(32 + 21, WQ_DR1, 1032)
            GX_WR_CASE(32 + 22, WQ_C5A, 1064) GX_WR_CASE(32 + 23, WQ_C5A, 1032) GX_WR_CASE(32 + 24, WQ_C5B, 1064) GX_WR_CASE(32 + 25, WQ_C5B, 1032)
            GX_WR_CASE(64 + 18, WQ_C3, 2064) GX_WR_CASE(64 + 19, WQ_C3, 2032) GX_WR_CASE(64 + 20, WQ_DR0, 2032) GX_WR_CASE(64 + 21, WQ_DR1, 2032)
            GX_WR_CASE(64 + 22, WQ_C5A, 2064) GX_WR_CASE(64 + 23, WQ_C5A, 2032) GX_WR_CASE(64 + 24, WQ_C5B, 2064) GX_WR_CASE(64 + 25, WQ_C5B, 2032)
            GX_WR_CASE(96 + 18, WQ_C3, 3064)
            GX_WF_CASE(18, WQ_C3, 64) GX_WF_CASE(19, WQ_C3, 32) GX_WF_CASE(20, WQ_DR0, 32) GX_WF_CASE(21, WQ_DR1, 32)
            GX_WF_CASE(26, WQ_C3, 16) GX_WF_CASE(27, WQ_DR0, 16) GX_WF_CASE(28, WQ_DR1, 16)
            GX_WF_CASE(29, WQ_C3, 128) GX_WF_CASE(30, WQ_DR0, 64) GX_WF_CASE(31, WQ_DR1, 64)
            GX_WF_CASE(22, WQ_C5A, 64) GX_WF_CASE(23, WQ_C5A, 32) GX_WF_CASE(24, WQ_C5B, 64) GX_WF_CASE(25, WQ_C5B, 32)
            GX_WF_CASE(32, WQ_C5A, 16) GX_WF_CASE(33, WQ_C5B, 16)
            // ... and their k-split forms (the gated stacks' and MONet's 32-channel blocks)
            GX_WF_CASE(32 + 18, WQ_C3, 1064) GX_WF_CASE(32 + 19, WQ_C3, 1032) GX_WF_CASE(32 + 20, WQ_DR0, 1032)
            GX_WF_CASE(32 + 22, WQ_C5A, 1064) GX_WF_CASE(32 + 23, WQ_C5A, 1032) GX_WF_CASE(32 + 24, WQ_C5B, 1064) GX_WF_CASE(32 + 25, WQ_C5B, 1032)
            GX_WF_CASE(64 + 18, WQ_C3, 2064) GX_WF_CASE(64 + 19, WQ_C3, 2032) GX_WF_CASE(64 + 20, WQ_DR0, 2032)
            GX_WF_CASE(64 + 22, WQ_C5A, 2064) GX_WF_CASE(64 + 23, WQ_C5A, 2032) GX_WF_CASE(64 + 24, WQ_C5B, 2064) GX_WF_CASE(64 + 25, WQ_C5B, 2032)
            GX_WF_CASE(96 + 18, WQ_C3, 3064)
            default: break;
        }
#undef GX_WS_CASE
#undef GX_WR_CASE
#undef GX_WF_CASE
        __syncthreads();          // the next segment's first DMA re-uses stage 0
    }
    if (tab.times && threadIdx.x == 0) {          // measurement: when did this workgroup start and finish
        __builtin_amdgcn_s_waitcnt(0);
        tab.times[2 * wg] = t_begin;
        tab.times[2 * wg + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
struct PendingJob {
    WqJob job;
    int cls, ltw;
    float* dw; int layout;          // where the reduce writes
    double flops;
    int reduce_group;               // jobs of one transposed-conv layer share a reduce record (index of the first)
    // the operands' partial maxima (gx_wgq_operand_amax): a (dy) in up to two arrays, b (x) in up to two; am_out: two floats
    // that receive {max |a|, max |b|} ahead of the stream-K launch.  am_out == NULL: unknown -> bf16 pieces
    const float* am_p[4]; int am_n[4]; float* am_out;
};
std::vector<PendingJob> g_jobs_ctx[kGxMaxCtx];      // queued jobs of each context (gx_common.h)
#define g_jobs (g_jobs_ctx[gx_cur_ctx()])
const float* g_zero16 = nullptr;

const float* zero16(hipStream_t s) {
    if (g_zero16) return g_zero16;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    float* p = nullptr;
    if (hipMalloc((void**)&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_zero16 = p;
    return g_zero16;
}

// ---- the operands' maxima for the fp16-piece tiles (gx_wgq_operand_amax): a one-shot, per-thread hint that the NEXT weight-gradient
// call copies into its job(s)
struct WqAmaxHint { const float* p[4]; int n[4]; float* out; };
thread_local WqAmaxHint t_amax_hint = {{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}, nullptr};
void wgq_take_hint(PendingJob& p) {
    for (int i = 0; i < 4; ++i) { p.am_p[i] = t_amax_hint.p[i]; p.am_n[i] = t_amax_hint.n[i]; }
    p.am_out = (t_amax_hint.out && t_amax_hint.n[0] > 0 && t_amax_hint.n[2] > 0) ? t_amax_hint.out : nullptr;
    t_amax_hint.out = nullptr;
    for (int i = 0; i < 4; ++i) { t_amax_hint.p[i] = nullptr; t_amax_hint.n[i] = 0; }
}
// {max |a|, max |b|} of up to kMaxFin layers from their partial maxima: one workgroup per (layer, operand)
constexpr int kMaxFin = 40;
struct WqFinEntry { const float* p0; const float* p1; float* out; int n0, n1; };
struct WqFinTable { WqFinEntry e[2 * kMaxFin]; };
__global__ void __launch_bounds__(256)
wgq_amax_finalize_kernel(const WqFinTable tab) {
    __shared__ float red[4];
    const WqFinEntry e = tab.e[blockIdx.x];
    float m = 0.f;
    for (int i = threadIdx.x; i < e.n0; i += 256) m = fmaxf(m, fabsf(e.p0[i]));
    for (int i = threadIdx.x; i < e.n1; i += 256) m = fmaxf(m, fabsf(e.p1[i]));
#pragma unroll
    for (int of = 32; of >= 1; of >>= 1) m = fmaxf(m, __shfl_xor(m, of, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *e.out = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// fp16 pieces where both operands' maxima are known: 1 (default); GENESIS_WGQ_F16X3=0 / gx_wgq_precision(1): bf16 pieces everywhere
double g_wgq_flops_f16 = 0.0, g_wgq_flops_all = 0.0;      // of the last stream-K launch (gx_wgq_last_f16_share)
int g_wgq_f16 = -1;
bool wgq_f16() {
    if (g_wgq_f16 < 0) {
        const char* env = getenv("GENESIS_WGQ_F16X3");
        g_wgq_f16 = (env && env[0] == '0') ? 0 : 1;
    }
    return g_wgq_f16 != 0;
}
// units per fp16-piece tile by row-ring variant 18 .. 28 (0: the variant has no fp16 form), re-fitted with GENESIS_WGQ_TIMES on the
// metric step next to the bf16 / LDS-DMA variants of the same launch: 64 - 78 % of the bf16 tile (the 10-tap row parity gains
// least: its split / staging work per tile is that of the 15-tap one).  GENESIS_WGQ_F16_COST="c18,c19,c20,c21,c26,c27,c28" overrides
int g_ws_cost_f16[16] = {3040, 1760, 2650, 2315,
                         5300, 2700, 3560, 1850,      // ... of the 5 x 5 stride-1 conv (the gated stacks): 0.66 x the bf16 tile, estimates
                         1870, 2700, 2380,
                         3200, 2750, 2450,            // ... one strip per tile (the 128 x 128 model's large layers; fitted on the K = 11 step:
                                                      //     workgroups of the 10-tap parity ran 15 % over the others at 2150)
                         2700, 1850};                 // ... 5 x 5, two 16-pixel rows per tile: estimates
bool g_ws_cost_f16_init = false;
int ws_f16cost(int rv) {
    if (!g_ws_cost_f16_init) {
        g_ws_cost_f16_init = true;
        if (const char* env = getenv("GENESIS_WGQ_F16_COST")) {
            int v[7];
            if (sscanf(env, "%d,%d,%d,%d,%d,%d,%d", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6) == 7) {
                static const int idx[7] = {0, 1, 2, 3, 8, 9, 10};
                for (int i = 0; i < 7; ++i) if (v[i] > 0) g_ws_cost_f16[idx[i]] = v[i];
            }
        }
    }
    return g_ws_cost_f16[rv - 18];
}
bool ws_f16_variant(int rv) { return rv >= 18 && rv <= 33; }

// which matrix pipe the weight gradients run on: 1 (default) bf16 pipe, fp32 products from six bf16 piece products
// (wq_tile_b6); 0 the fp32 pipe.  GENESIS_WGQ_BF16X6=0 / gx_wgq_precision(0) select the latter.
int g_wgq_b6 = -1;
bool wgq_b6() {
    if (g_wgq_b6 < 0) {
        const char* env = getenv("GENESIS_WGQ_BF16X6");
        g_wgq_b6 = (env && env[0] == '0') ? 0 : 1;
    }
    return g_wgq_b6 != 0;
}

template <int CLS, int LTW, bool B6>
void wgq_launch_inst(const WqTable& tab, int total_wgs, const float* zeros, hipStream_t s) {
    using WT = WqTap<CLS>;
    constexpr int TW = 1 << LTW, TH = 64 >> LTW;
    constexpr size_t lds = (size_t)2 * (64 * (TH * WT::SA * TW / 4) + 64 * kSB) * 16;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgq_kernel<CLS, LTW, B6>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((wgq_kernel<CLS, LTW, B6>), dim3(total_wgs), dim3(256), lds, s, tab, zeros);
}

template <int CLS, int LTW>
void wgq_launch_pipe(const WqTable& tab, int total_wgs, const float* zeros, hipStream_t s) {
    if (wgq_b6()) wgq_launch_inst<CLS, LTW, true>(tab, total_wgs, zeros, s);
    else wgq_launch_inst<CLS, LTW, false>(tab, total_wgs, zeros, s);
}

void wgq_launch(int cls, int ltw, const WqTable& tab, int total_wgs, const float* zeros, hipStream_t s) {
    if (cls == WQ_C3) {
        if (ltw == 5) wgq_launch_pipe<WQ_C3, 5>(tab, total_wgs, zeros, s);
        else if (ltw == 4) wgq_launch_pipe<WQ_C3, 4>(tab, total_wgs, zeros, s);
        else wgq_launch_pipe<WQ_C3, 3>(tab, total_wgs, zeros, s);
    } else if (cls == WQ_DR0) {
        if (ltw == 5) wgq_launch_pipe<WQ_DR0, 5>(tab, total_wgs, zeros, s);
        else if (ltw == 4) wgq_launch_pipe<WQ_DR0, 4>(tab, total_wgs, zeros, s);
        else wgq_launch_pipe<WQ_DR0, 3>(tab, total_wgs, zeros, s);
    } else {
        if (ltw == 5) wgq_launch_pipe<WQ_DR1, 5>(tab, total_wgs, zeros, s);
        else if (ltw == 4) wgq_launch_pipe<WQ_DR1, 4>(tab, total_wgs, zeros, s);
        else wgq_launch_pipe<WQ_DR1, 3>(tab, total_wgs, zeros, s);
    }
}

int g_wgq_mode = -1;
int wgq_mode() {
    if (g_wgq_mode < 0) {
        const char* env = getenv("GENESIS_WGQ");
        g_wgq_mode = env ? (env[0] == '0' ? 0 : 1) : 1;
    }
    return g_wgq_mode;
}

// fills the geometry part of a job; false if the layer is not eligible
bool wgq_make_job(int sa, const float* a, const float* b, float* partial, int N, int CA, int CB, int Hb, int Wb, int Ttot,
                  int max_split, WqJob* jb, int* ltw) {
    if (!gx_is_pow2(Hb) || !gx_is_pow2(Wb) || Wb < 8) return false;
    const int lt = Wb >= 32 ? 5 : (Wb >= 16 ? 4 : 3);
    const int TW = 1 << lt, TH = 64 >> lt;
    if (Hb < TH) return false;
    if ((double)N * CA * sa * Hb * sa * Wb >= 2.0e9 || (double)N * CB * Hb * Wb >= 2.0e9) return false;   // int offsets per job
    jb->a = a; jb->b = b; jb->partial = partial;
    jb->N = N; jb->CA = CA; jb->CB = CB; jb->CApad = gx_round_up(CA, 64); jb->CBpad = gx_round_up(CB, 64);
    jb->Hb = Hb; jb->Wb = Wb;
    jb->tiles_h = Hb / TH; jb->tiles_w = Wb / TW; jb->ntiles = N * jb->tiles_h * jb->tiles_w;
    jb->nbt = jb->CBpad / 64;
    jb->nsplit = max_split; jb->wg_begin = 0; jb->Ttot = Ttot;
    *ltw = lt;
    return true;
}

// launches the jobs [first, last) of g_jobs that share (cls, ltw) as one grid of ~budget workgroups
int wgq_launch_group(std::vector<PendingJob*>& grp, int budget, hipStream_t s) {
    const float* zeros = zero16(s);
    if (!zeros) { gx_set_error("wgq: zero page unavailable (first call inside a stream capture)"); return GX_ELAUNCH; }
    double total = 0.0;
    for (PendingJob* p : grp) total += p->flops;
    WqTable tab;
    tab.njobs = (int)grp.size();
    int wg = 0;
    double flops = 0.0, bytes = 0.0;
    for (size_t i = 0; i < grp.size(); ++i) {
        PendingJob& p = *grp[i];
        const int nblk = (p.job.CApad / 64) * (p.job.CBpad / 64);
        int ns = (int)(budget * (p.flops / total) / nblk + 0.5);
        if (ns < 1) ns = 1;
        if (ns > p.job.nsplit) ns = p.job.nsplit;          // planned upper bound (workspace size)
        if (ns > p.job.ntiles) ns = p.job.ntiles;
        p.job.nsplit = ns;
        p.job.wg_begin = wg;
        wg += ns * nblk;
        tab.job[i] = p.job;
        flops += p.flops;
        bytes += 4.0 * ((double)p.job.N * p.job.CB * p.job.Hb * p.job.Wb + (double)ns * nblk * 64 * 64 * WqTap<WQ_C3>::NT);
    }
    {
        const int kid = grp[0]->cls == WQ_C3 ? KID_WGRAD_C3 : (grp[0]->cls == WQ_DR0 ? KID_WGRAD_D00 : KID_WGRAD_D10);
        GxProf pf(kid, s, flops, bytes);
        wgq_launch(grp[0]->cls, grp[0]->ltw, tab, wg, zeros, s);
    }
    GX_CHECK_LAUNCH("wgq (grouped weight gradients)");
    return GX_OK;
}


// ---- stream-K host side ------------------------------------------------------------------------------------------
int g_wgq_stream = -1;
bool wgq_stream_on() {
    if (g_wgq_stream < 0) {
        const char* env = getenv("GENESIS_WGQ_STREAM");
        g_wgq_stream = (env && env[0] == '0') ? 0 : 1;
    }
    return g_wgq_stream != 0;
}

// units per tile of a variant (class * 3 + (5 - ltw)): ns per tile and workgroup of single-layer launches (tools/kq_time.py
// wgrad3:256:64 wgrad3:2048:16 wgrad3:8192:8 wgrad:256:32 wgrad:1024:16 wgrad:4096:8: 10.3 us per conv3x3 tile, 26.5 us per
// pair of transposed-conv row-parity tiles, split 15 a + b : 10 a + b), then nudged on the training step itself
// (the stream kernel's duration over five vectors: 921 .. 972 us); GENESIS_WGQ_COST="c0,...,c8" overrides
int g_ws_cost[34] = {10200, 9580, 9600, 15080, 21000, 11800, 11400, 8800, 9700,              // bf16 pipe (measured with
                     10200, 10600, 11200, 16000, 16500, 18200, 11600, 12000, 13300,           // GENESIS_WGQ_TIMES) | fp32 pipe
                     4840, 2450, 4150, 2900,                                                  // row-ring tiles (one base row)
                     8000, 4100, 5400, 2800,                                                  // ... of the 5 x 5 stride-1 conv
                     2450, 4150, 2900,                                                        // ... two 16-pixel rows per tile
                     4900, 4200, 2950,                                                        // ... one strip (half a base row) per tile
                     4100, 2800};                                                             // ... 5 x 5, two 16-pixel rows per tile
// cost of a k-split tile relative to the full one, in percent, by base variant 18..25 (one or the other half) and for both
// halves of variant 18 (measured with GENESIS_WGQ_TIMES on the GENESIS / BaselineVAE / MONet steps: the 32-pixel rows keep one
// k-group per wave -- their tile starts with the operand reads the tail could not take); GENESIS_WGQ_KSPLIT_COST="9 values" overrides
int g_ws_kcost[9] = {54, 68, 68, 78, 54, 58, 55, 65, 41};
bool g_ws_cost_init = false;
void ws_cost_init() {
    if (g_ws_cost_init) return;
    g_ws_cost_init = true;
    const char* env = getenv("GENESIS_WGQ_COST");
    if (const char* er = getenv("GENESIS_WGQ_RING_COST")) {
        int r[4];
        if (sscanf(er, "%d,%d,%d,%d", r, r + 1, r + 2, r + 3) == 4)
            for (int i = 0; i < 4; ++i) if (r[i] > 0) g_ws_cost[18 + i] = r[i];
    }
    if (const char* er = getenv("GENESIS_WGQ_R2_COST")) {       // two-rows-per-tile variants: conv3x3, transposed-conv rows 0 / 1
        int r[3];
        if (sscanf(er, "%d,%d,%d", r, r + 1, r + 2) == 3)
            for (int i = 0; i < 3; ++i) if (r[i] > 0) g_ws_cost[26 + i] = r[i];
    }
    if (const char* er = getenv("GENESIS_WGQ_KSPLIT_COST")) {
        int r[9];
        if (sscanf(er, "%d,%d,%d,%d,%d,%d,%d,%d,%d", r, r + 1, r + 2, r + 3, r + 4, r + 5, r + 6, r + 7, r + 8) == 9)
            for (int i = 0; i < 9; ++i) if (r[i] > 0) g_ws_kcost[i] = r[i];
    }
    if (!env) return;
    int v[9];          // the nine costs of the pipe in use
    if (sscanf(env, "%d,%d,%d,%d,%d,%d,%d,%d,%d", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7, v + 8) == 9)
        for (int i = 0; i < 9; ++i) if (v[i] > 0) g_ws_cost[(wgq_b6() ? 0 : 9) + i] = v[i];
}

// row-ring tiles (wr_segment): on the bf16 pipe, for full-width rows of 32 / 64 pixels; GENESIS_WGQ_RING=0 / gx_wgq_ring(0)
// keep every layer on the 64-pixel LDS-DMA tiles (the A/B reference)
int g_wgq_ring = -1;
bool wgq_ring_on() {
    if (g_wgq_ring < 0) {
        const char* env = getenv("GENESIS_WGQ_RING");
        g_wgq_ring = (env && env[0] == '0') ? 0 : 1;
    }
    return g_wgq_ring != 0;
}
int ws_ring_variant(int cls, int Hb, int Wb) {
    if (!wgq_b6() || !wgq_ring_on() || Hb < 4) return -1;          // (H a multiple of 4: the ring slot of a row is tile & 3)
    static const char* r2env = getenv("GENESIS_WGQ_RING16");       // 0: 16-pixel rows stay on the 64-pixel LDS-DMA tiles
    if (Wb == 16 && !(r2env && r2env[0] == '0') && Hb >= 8 && Hb % 8 == 0 && cls <= WQ_DR1) return 26 + cls;   // two image rows per tile
    if (Wb == 16 && !(r2env && r2env[0] == '0') && Hb >= 8 && Hb % 8 == 0 && (cls == WQ_C5A || cls == WQ_C5B)) return cls == WQ_C5A ? 32 : 33;
    // rows too wide for the ring as two strips (the 128 x 128 model's large layers); GENESIS_WGQ_STRIPS=0: on the 64-pixel tiles
    static const char* stenv = getenv("GENESIS_WGQ_STRIPS");
    const bool strips = !(stenv && stenv[0] == '0');
    if (cls == WQ_C3) return Wb == 64 ? 18 : (Wb == 32 ? 19 : (Wb == 128 && strips ? 29 : -1));
    if (cls == WQ_C5A || cls == WQ_C5B) return Wb == 64 ? (cls == WQ_C5A ? 22 : 24) : (Wb == 32 ? (cls == WQ_C5A ? 23 : 25) : -1);
    if (Wb == 64 && strips) return cls == WQ_DR0 ? 30 : 31;
    if (Wb != 32) return -1;
    return cls == WQ_DR0 ? 20 : 21;
}

// k-split form of a row-ring variant for a channel block with ca_n x cb_n channels (0: none).  GENESIS_WGQ_KSPLIT=0: off
int ws_ksplit(int rv, int ca_n, int cb_n) {
    static const char* env = getenv("GENESIS_WGQ_KSPLIT");
    if ((env && env[0] == '0') || rv < 18 || rv > 25) return 0;
    int hf = (ca_n <= 32 ? 1 : 0) | (cb_n <= 32 ? 2 : 0);
    if (hf == 3 && rv != 18) hf = 1;          // (both halves empty: built for the 64-pixel conv3x3 rows only)
    return hf;
}
struct WsSlot { PendingJob* p; int blk; int nseg; };

// fills tab.job[*].ubegin / w_first and slots[*].nseg for G workgroups; false if a block would need more slabs than its
// region holds or a workgroup boundary leaves a hole in a block's slab numbering
bool ws_plan(WsTable& tab, std::vector<WsSlot>& slots, int G) {
    tab.G = G;
    const int nj = tab.njobs;
    std::vector<int> first(nj, -1), last(nj, -1), count(nj, 0);
    for (int w = 0; w < G; ++w) {
        int js, ts, je, te;
        ws_locate(tab, tab.U * w / G, &js, &ts);
        ws_locate(tab, w + 1 == G ? tab.U : tab.U * (w + 1) / G, &je, &te);
        for (int j = js; j <= je && j < nj; ++j) {
            const int t0 = j == js ? ts : 0, t1 = j == je ? te : tab.job[j].ntiles;
            if (t0 >= t1) continue;
            if (first[j] < 0) first[j] = w;
            last[j] = w;
            ++count[j];
        }
    }
    for (int j = 0; j < nj; ++j) {
        if (first[j] < 0 || count[j] != last[j] - first[j] + 1) return false;
        if (count[j] > slots[j].p->job.nsplit) return false;          // nsplit = slabs the region holds
        tab.job[j].w_first = first[j];
        slots[j].nseg = count[j];
    }
    return true;
}

// launches `jobs` (any classes / tile widths) as stream-K grids and appends their reduce records
int wgq_launch_stream(std::vector<PendingJob*>& jobs, hipStream_t s, std::vector<GxWgradRed>& recs) {
    const float* zeros = zero16(s);
    if (!zeros) { gx_set_error("wgq: zero page unavailable (first call inside a stream capture)"); return GX_ELAUNCH; }
    ws_cost_init();
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgq_stream_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    size_t i = 0;
    while (i < jobs.size()) {
        // a chunk: whole layers (both row parities of a transposed conv stay together) while their blocks fit the table
        WsTable tab;
        std::vector<WsSlot> slots;
        WqFinTable fin;
        int nfin = 0;
        double flops_f16 = 0.0;
        tab.njobs = 0; tab.U = 0;
        double flops = 0.0, bytes = 0.0;
        size_t i_end = i;
        for (; i_end < jobs.size(); ++i_end) {
            PendingJob& p = *jobs[i_end];
            const int nblk = (p.job.CApad / 64) * (p.job.CBpad / 64);
            const bool pair = (p.cls == WQ_DR0 || p.cls == WQ_C5A) && i_end + 1 < jobs.size() &&
                              jobs[i_end + 1]->cls == p.cls + 1 && jobs[i_end + 1]->job.partial == p.job.partial;
            if (tab.njobs + nblk * (pair ? 2 : 1) > kMaxSJobs) break;
            for (int half = 0; half < (pair ? 2 : 1); ++half) {
                PendingJob& q = *jobs[i_end + half];
                for (int blk = 0; blk < nblk; ++blk) {
                    WsJob& jb = tab.job[tab.njobs++];
                    jb.a = q.job.a; jb.b = q.job.b;
                    jb.partial = q.job.partial + (size_t)blk * q.job.nsplit * q.job.Ttot * 4096;
                    jb.ubegin = tab.U;
                    jb.CA = q.job.CA; jb.CB = q.job.CB;
                    jb.ca0 = (blk / q.job.nbt) * 64; jb.cb0 = (blk % q.job.nbt) * 64;
                    jb.Hb = q.job.Hb; jb.Wb = q.job.Wb;
                    jb.tiles_h = q.job.tiles_h; jb.tiles_w = q.job.tiles_w; jb.ntiles = q.job.ntiles;
                    jb.Ttot = q.job.Ttot;
                    jb.variant = q.cls * 3 + (5 - q.ltw) + (wgq_b6() ? 0 : 9);
                    const int rv = ws_ring_variant(q.cls, q.job.Hb, q.job.Wb);
                    // tile = one base row (26..28: two; 29..31: one of its two strips)
                    if (rv >= 0) { jb.variant = rv; jb.ntiles = (rv >= 29 && rv <= 31) ? q.job.N * q.job.Hb * 2 : q.job.N * q.job.Hb / (rv >= 26 ? 2 : 1); }
                    else if (q.cls >= WQ_C5A) { gx_set_error("wgq: the 5x5 classes exist as row-ring tiles only"); return GX_EINVAL; }
                    jb.cost = g_ws_cost[jb.variant];
                    if (rv >= 0) {
                        const int ca_n = q.job.CA - jb.ca0 < 64 ? q.job.CA - jb.ca0 : 64, cb_n = q.job.CB - jb.cb0 < 64 ? q.job.CB - jb.cb0 : 64;
                        const int hf = ws_ksplit(rv, ca_n, cb_n);
                        if (hf) { jb.variant = rv + 32 * hf; jb.cost = jb.cost * g_ws_kcost[hf == 3 ? 8 : rv - 18] / 100; }
                    }
                    jb.w_first = 0; jb.N = q.job.N;
                    jb.amax2 = nullptr;
                    // (the k-split 10-tap row parity at 32 pixels has no fp16 form: its split pieces do not fit the tile's free slots)
                    if (rv >= 0 && ws_f16_variant(rv) && !(jb.variant != rv && (rv == 21 || rv > 25)) && q.am_out && wgq_f16() && nfin < 2 * kMaxFin - 2) {
                        // both operands' maxima are known: two fp16 pieces per value, three piece products (k-split forms: the fp16
                        // tile's cost scaled like the bf16 one's)
                        const int hfv = jb.variant - rv;            // 32 * HF
                        // (... then fitted on the GENESIS step with GENESIS_WGQ_TIMES: the 32-pixel and 10-tap forms gain less, percent)
                        static const int kadj[8] = {100, 100, 108, 100, 91, 109, 104, 119};      // rv 18 .. 25
                        jb.cost = hfv ? (int)((long long)jb.cost * ws_f16cost(rv) / g_ws_cost[rv] * (rv <= 25 ? kadj[rv - 18] : 100) / 100) : ws_f16cost(rv);
                        jb.variant = rv + hfv + 128; jb.amax2 = q.am_out;
                        bool seen = false;
                        for (int k = 0; k < nfin; ++k) seen = seen || fin.e[k].out == q.am_out;
                        if (!seen) {
                            fin.e[nfin++] = WqFinEntry{q.am_p[0], q.am_p[1], q.am_out, q.am_n[0], q.am_n[1]};
                            fin.e[nfin++] = WqFinEntry{q.am_p[2], q.am_p[3], q.am_out + 1, q.am_n[2], q.am_n[3]};
                        }
                    }
                    tab.U += (long long)jb.ntiles * jb.cost;
                    if (jb.variant >= 128) flops_f16 += q.flops / nblk;
                    slots.push_back(WsSlot{&q, blk, 0});
                }
                flops += q.flops;
                bytes += 4.0 * ((double)q.job.N * q.job.CB * q.job.Hb * q.job.Wb +
                                (double)q.job.N * q.job.CA * q.job.Hb * q.job.Wb * ((q.cls == WQ_DR0 || q.cls == WQ_DR1) ? 2 : 1));
            }
            if (pair) ++i_end;
        }
        if (tab.njobs == 0) { gx_set_error("wgq stream: a layer has more channel blocks than the job table"); return GX_EINVAL; }
        // G workgroups: one per CU unless a block would need more slabs than its region holds
        int G = 256;
        int maxcost = 1;
        for (int j = 0; j < tab.njobs; ++j) maxcost = tab.job[j].cost > maxcost ? tab.job[j].cost : maxcost;
        if ((long long)G * maxcost * 2 > tab.U) G = (int)(tab.U / (2LL * maxcost));
        if (G < 1) G = 1;
        while (!ws_plan(tab, slots, G)) {
            if (G == 1) { gx_set_error("wgq stream: no feasible plan"); return GX_EINVAL; }
            G = G > 16 ? G - 16 : G - 1;
        }
        static const bool dbg = getenv("GENESIS_WGQ_DEBUG") != nullptr;
        if (dbg) {
            fprintf(stderr, "[wgq stream] G %d, U %lld, %d block-jobs\n", G, tab.U, tab.njobs);
            for (int j = 0; j < tab.njobs; ++j)
                fprintf(stderr, "  job %2d variant %d  CA %3d CB %3d (block %d,%d)  %dx%d  tiles %6d  cost %5d  share %.3f  slabs %d\n",
                        j, tab.job[j].variant, tab.job[j].CA, tab.job[j].CB, tab.job[j].ca0, tab.job[j].cb0, tab.job[j].Hb,
                        tab.job[j].Wb, tab.job[j].ntiles, tab.job[j].cost,
                        (double)tab.job[j].ntiles * tab.job[j].cost / (double)tab.U, slots[j].nseg);
        }
        // GENESIS_WGQ_TIMES=1 (eager launches only): per-workgroup start / end times of the launch -> how well the cost
        // table balances the 256 workgroups (printed: busy time min / mean / max and the idle share of the launch)
        static const bool want_times = getenv("GENESIS_WGQ_TIMES") != nullptr;
        static long long* d_times = nullptr;
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(s, &cst) != hipSuccess || cst != hipStreamCaptureStatusNone;
        tab.times = nullptr;
        if (want_times && !capturing) {
            if (!d_times) (void)hipMalloc((void**)&d_times, 2 * 256 * sizeof(long long));
            tab.times = d_times;
        }
        g_wgq_flops_f16 = flops_f16; g_wgq_flops_all = flops;
        if (nfin) {
            GxProf pf(KID_SMALL_REDUCE, s, 0.0, 0.0);
            hipLaunchKernelGGL(wgq_amax_finalize_kernel, dim3(nfin), dim3(256), 0, s, fin);
        }
        {
            GxProf pf(KID_WGQ_STREAM, s, flops, bytes);
            hipLaunchKernelGGL(wgq_stream_kernel, dim3(G), dim3(256), 160 * 1024, s, tab, zeros);
        }
        GX_CHECK_LAUNCH("wgq (stream-K weight gradients)");
        if (tab.times) {
            std::vector<long long> h(2 * G);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h.data(), d_times, 2 * G * sizeof(long long), hipMemcpyDeviceToHost);
            long long t0 = h[0], t1 = h[1];
            double busy = 0.0, bmin = 1e30, bmax = 0.0;
            for (int w = 0; w < G; ++w) {
                t0 = h[2 * w] < t0 ? h[2 * w] : t0; t1 = h[2 * w + 1] > t1 ? h[2 * w + 1] : t1;
                const double b = (double)(h[2 * w + 1] - h[2 * w]) * 0.01;
                busy += b; bmin = b < bmin ? b : bmin; bmax = b > bmax ? b : bmax;
            }
            const double span = (double)(t1 - t0) * 0.01;
            fprintf(stderr, "[wgq times] G %d span %.1f us; workgroup busy min %.1f mean %.1f max %.1f us; idle share %.3f\n", G,
                    span, bmin, busy / G, bmax, 1.0 - busy / (G * span));
            // busy time by the job a workgroup STARTS in (same job: same kind of work)
            for (int j = 0; j < tab.njobs; ++j) {
                double sj = 0.0; int nj = 0;
                for (int w = 0; w < G; ++w) {
                    int js, ts;
                    ws_locate(tab, tab.U * w / G, &js, &ts);
                    if (js == j) { sj += (double)(h[2 * w + 1] - h[2 * w]) * 0.01; ++nj; }
                }
                if (nj) fprintf(stderr, "   workgroups starting in job %2d (variant %d, %dx%d): %3d, mean busy %.1f us\n", j,
                                tab.job[j].variant, tab.job[j].Hb, tab.job[j].Wb, nj, sj / nj);
            }
        }
        // reduce records: one per (layer, channel block); the two row parities of a transposed conv share the slabs
        for (int j = 0; j < tab.njobs; ++j) {
            PendingJob& q = *slots[j].p;
            if ((q.cls == WQ_DR1 || q.cls == WQ_C5B) && j > 0) {
                bool paired = false;
                for (int k = 0; k < j; ++k)
                    paired = paired || (slots[k].p->cls == q.cls - 1 && slots[k].p->job.partial == q.job.partial);
                if (paired) continue;
            }
            const WsJob jb = tab.job[j];          // (a copy: 96 bytes of kernel-argument memory into scalar registers)
            const int ca_n = q.job.CA - jb.ca0 < 64 ? q.job.CA - jb.ca0 : 64;
            const int cb_n = q.job.CB - jb.cb0 < 64 ? q.job.CB - jb.cb0 : 64;
            GxWgradRed r{jb.partial, q.dw, slots[j].nseg, jb.Ttot, ca_n, cb_n, 64, 64, q.layout, 0, 0, 0, 0,
                         jb.ca0, jb.cb0, q.job.CA, q.job.CB};
            if (q.cls != WQ_C3) {
                const int c0 = q.cls >= WQ_C5A ? WQ_C5A : WQ_DR0;
                int n0 = q.cls == c0 ? slots[j].nseg : 0, n1 = q.cls == c0 + 1 ? slots[j].nseg : 0;
                for (int k = 0; k < tab.njobs; ++k)
                    if (k != j && tab.job[k].partial == jb.partial) {
                        if (slots[k].p->cls == c0) n0 = slots[k].nseg;
                        if (slots[k].p->cls == c0 + 1) n1 = slots[k].nseg;
                    }
                // taps of a row class the launch did not cover (never the case for gx_wgq_deconv / gx_wgq_c5) would read slab 0
                if (c0 == WQ_DR0) { r.ns0 = r.ns1 = n0 > 0 ? n0 : 1; r.ns2 = r.ns3 = n1 > 0 ? n1 : 1; }
                else { r.ns0 = -1; r.ns1 = n0 > 0 ? n0 : 1; r.ns2 = n1 > 0 ? n1 : 1; r.ns3 = 0; }     // 5 x 5: kernel rows 0-2 | 3-4
                r.nsplit = n0 > n1 ? n0 : n1;
            }
            recs.push_back(r);
        }
        i = i_end;
    }
    return GX_OK;
}

}  // namespace

// ---- internal API (gx_common.h) ----------------------------------------------------------------------------------
// upper bound of the splits a job may get (sizes its slab workspace)
int gx_wgq_max_split(int CA, int CB) {
    const int nblk = (gx_round_up(CA, 64) / 64) * (gx_round_up(CB, 64) / 64);
    return gx_ceil_div(256, nblk);
}

bool gx_wgq_c3_eligible(int N, int Cin, int Cout, int H, int W) {
    WqJob jb; int ltw;
    return wgq_mode() && Cin * 9 > 32 && wgq_make_job(1, nullptr, nullptr, nullptr, N, Cout, Cin, H, W, 9, 1, &jb, &ltw);
}
bool gx_wgq_deconv_eligible(int N, int Cin, int Cout, int Hb, int Wb) {
    WqJob jb; int ltw;
    return wgq_mode() && wgq_make_job(2, nullptr, nullptr, nullptr, N, Cout, Cin, Hb, Wb, 25, 1, &jb, &ltw);
}

static int wgq_run_or_queue(std::vector<PendingJob>& jobs, hipStream_t s) {
    if (g_gx_defer_on && zero16(s)) {          // queued: launched in groups by gx_wgq_flush
        for (PendingJob& p : jobs) g_jobs.push_back(p);
        return GX_OK;
    }
    // immediate.  Deferral on but no queue (zero16 unavailable: a first call inside a stream capture): the reduce ADDS, as
    // the queued one would -- the destination is a zeroed bucket and a shared parameter may be on its second use.
    const int acc = g_gx_defer_on ? 1 : 0;
    if (wgq_stream_on()) {                      // the layer alone on the chip, then its reductions
        std::vector<PendingJob*> ptrs;
        for (PendingJob& p : jobs) ptrs.push_back(&p);
        std::vector<GxWgradRed> recs;
        int rc = wgq_launch_stream(ptrs, s, recs);
        for (size_t i = 0; i < recs.size() && rc == GX_OK; ++i) rc = gx_wgrad_reduce_now(recs[i], s, acc);
        return rc;
    }
    // each (cls) job is its own launch with the whole chip; then the reduce record
    int rc = GX_OK;
    for (PendingJob& p : jobs) {
        std::vector<PendingJob*> one{&p};
        rc = wgq_launch_group(one, 256, s);
        if (rc) return rc;
    }
    PendingJob& f = jobs[0];
    GxWgradRed r{f.job.partial, f.dw, f.job.nsplit, f.job.Ttot, f.job.CA, f.job.CB, f.job.CApad, f.job.CBpad, f.layout, 0, 0, 0, 0,
                 0, 0, 0, 0};
    if (jobs.size() == 2) { r.ns0 = r.ns1 = jobs[0].job.nsplit; r.ns2 = r.ns3 = jobs[1].job.nsplit;
                            r.nsplit = r.ns0 > r.ns2 ? r.ns0 : r.ns2; }
    return gx_wgrad_reduce_now(r, s, acc);     // overwrites dw unless deferral is on (the deferred batch reduce accumulates)
}

// conv3x3: dw [Cout][Cin][3][3] from x [N,Cin,H,W], dy [N,Cout,H,W]; ws holds gx_wgq_max_split slabs
int gx_wgq_c3(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, float* ws,
              int ws_slabs, hipStream_t s) {
    PendingJob p;
    const int cap = ws_slabs < gx_wgq_max_split(Cout, Cin) ? ws_slabs : gx_wgq_max_split(Cout, Cin);
    if (cap < 1 || !wgq_make_job(1, dy, x, ws, N, Cout, Cin, H, W, 9, cap, &p.job, &p.ltw)) {
        gx_set_error("wgq conv3x3: shape not eligible");
        return GX_EINVAL;
    }
    p.cls = WQ_C3; p.dw = dw; p.layout = 0; p.reduce_group = -1;
    wgq_take_hint(p);
    p.flops = 2.0 * N * (double)Cout * Cin * 9 * H * W;
    std::vector<PendingJob> v{p};
    return wgq_run_or_queue(v, s);
}

// transposed conv: dw [Cin][Cout][5][5] from x [N,Cin,Hb,Wb], dy [N,Cout,2Hb,2Wb]
int gx_wgq_deconv(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int Hb, int Wb, float* ws,
                  int ws_slabs, hipStream_t s) {
    PendingJob p0, p1;
    const int ms = ws_slabs < gx_wgq_max_split(Cout, Cin) ? ws_slabs : gx_wgq_max_split(Cout, Cin);
    if (ms < 1 || !wgq_make_job(2, dy, x, ws, N, Cout, Cin, Hb, Wb, 25, ms, &p0.job, &p0.ltw)) {
        gx_set_error("wgq deconv: shape not eligible");
        return GX_EINVAL;
    }
    wgq_take_hint(p0);
    p1 = p0;
    p0.cls = WQ_DR0; p1.cls = WQ_DR1;
    p0.dw = p1.dw = dw; p0.layout = p1.layout = 1;
    p0.flops = 2.0 * N * (double)Cout * Cin * 15 * Hb * Wb;
    p1.flops = 2.0 * N * (double)Cout * Cin * 10 * Hb * Wb;
    p0.reduce_group = p1.reduce_group = -2;      // paired: resolved at flush by (dw, partial)
    std::vector<PendingJob> v{p0, p1};
    return wgq_run_or_queue(v, s);
}

// 5 x 5 stride-1 pad-2: dw [CA][CB][5][5] = sum_p a[CA][p] * b[CB][p + (kh - 2, kw - 2)]  (conv: a = dy, b = x -> [Cout][Cin];
// ConvTranspose2d stride 1: a = x, b = dy -> [Cin][Cout])
bool gx_wgq_c5_eligible(int N, int CA, int CB, int H, int W) {
    WqJob jb; int ltw;
    return wgq_mode() && wgq_stream_on() && ws_ring_variant(WQ_C5A, H, W) >= 0 &&
           wgq_make_job(1, nullptr, nullptr, nullptr, N, CA, CB, H, W, 25, 1, &jb, &ltw);
}
int gx_wgq_c5(const float* a, const float* b, float* dw, int N, int CA, int CB, int H, int W, float* ws, int ws_slabs,
              hipStream_t s) {
    PendingJob p0, p1;
    const int ms = ws_slabs < gx_wgq_max_split(CA, CB) ? ws_slabs : gx_wgq_max_split(CA, CB);
    if (ms < 1 || !gx_wgq_c5_eligible(N, CA, CB, H, W) || !wgq_make_job(1, a, b, ws, N, CA, CB, H, W, 25, ms, &p0.job, &p0.ltw)) {
        gx_set_error("wgq conv5x5: shape not eligible");
        return GX_EINVAL;
    }
    wgq_take_hint(p0);
    p1 = p0;
    p0.cls = WQ_C5A; p1.cls = WQ_C5B;
    p0.dw = p1.dw = dw; p0.layout = p1.layout = 0;
    p0.flops = 2.0 * N * (double)CA * CB * 15 * H * W;
    p1.flops = 2.0 * N * (double)CA * CB * 10 * H * W;
    p0.reduce_group = p1.reduce_group = -2;
    std::vector<PendingJob> v{p0, p1};
    return wgq_run_or_queue(v, s);
}

bool gx_wgq_bf16_pipe(void) { return wgq_b6(); }
int gx_wgq_pending(void) { return (int)g_jobs.size(); }
void gx_wgq_discard(void) { g_jobs.clear(); }

// launches every queued job -- one grid per (class, tile width) -- and queues the slab reductions
int gx_wgq_flush(hipStream_t s) {
    if (g_jobs.empty()) return GX_OK;
    int rc = GX_OK;
    if (wgq_stream_on()) {
        std::vector<PendingJob*> ptrs;
        for (PendingJob& p : g_jobs) ptrs.push_back(&p);
        std::vector<GxWgradRed> recs;
        rc = wgq_launch_stream(ptrs, s, recs);
        for (size_t i = 0; i < recs.size() && rc == GX_OK; ++i)
            if (!gx_defer_push_wgrad(recs[i])) rc = gx_defer_flush_wgrad(&recs[i], 1, s);
        g_jobs.clear();
        return rc;
    }
    for (int cls = 0; cls < 3 && rc == GX_OK; ++cls)
        for (int ltw = 5; ltw >= 3 && rc == GX_OK; --ltw) {
            std::vector<PendingJob*> grp;
            for (PendingJob& p : g_jobs)
                if (p.cls == cls && p.ltw == ltw) grp.push_back(&p);
            // at most kMaxJobs per launch
            for (size_t i = 0; i < grp.size() && rc == GX_OK; i += kMaxJobs) {
                std::vector<PendingJob*> part(grp.begin() + i, grp.begin() + (i + kMaxJobs < grp.size() ? i + kMaxJobs : grp.size()));
                rc = wgq_launch_group(part, 256, s);
            }
        }
    if (rc == GX_OK) {
        // reduce records: one per conv3x3 job, one per transposed-conv layer (its two row-parity jobs share the slabs)
        for (size_t i = 0; i < g_jobs.size() && rc == GX_OK; ++i) {
            PendingJob& p = g_jobs[i];
            if (p.cls == WQ_DR1) continue;
            GxWgradRed r{p.job.partial, p.dw, p.job.nsplit, p.job.Ttot, p.job.CA, p.job.CB, p.job.CApad, p.job.CBpad,
                         p.layout, 0, 0, 0, 0, 0, 0, 0, 0};
            if (p.cls == WQ_DR0) {
                int ns1 = 0;
                for (PendingJob& q : g_jobs)
                    if (q.cls == WQ_DR1 && q.job.partial == p.job.partial) ns1 = q.job.nsplit;
                r.ns0 = r.ns1 = p.job.nsplit; r.ns2 = r.ns3 = ns1;
                r.nsplit = p.job.nsplit > ns1 ? p.job.nsplit : ns1;
            }
            if (!gx_defer_push_wgrad(r)) rc = gx_defer_flush_wgrad(&r, 1, s);
        }
    }
    g_jobs.clear();
    return rc;
}

extern "C" int gx_wgq_precision(int mode) {
    GX_CHECK_ARG(mode >= -1 && mode <= 2, "gx_wgq_precision: mode must be 0 (fp32 matrix pipe), 1 (bf16 pipe, fp32 products from six "
                                          "bf16 piece products), 2 (as 1, three fp16 piece products where both operands' maxima are "
                                          "known) or -1 (the environment's default)");
    if (mode < 0) { g_wgq_b6 = -1; g_wgq_f16 = -1; return GX_OK; }
    g_wgq_b6 = mode ? 1 : 0;
    g_wgq_f16 = mode == 2 ? 1 : 0;
    return GX_OK;
}

extern "C" double gx_wgq_last_f16_share(void) { return g_wgq_flops_all > 0.0 ? g_wgq_flops_f16 / g_wgq_flops_all : 0.0; }

extern "C" int gx_wgq_operand_amax(const float* a0, int na0, const float* a1, int na1, const float* b0, int nb0, const float* b1,
                                   int nb1, float* out2) {
    WqAmaxHint& h = t_amax_hint;
    h.p[0] = a0; h.n[0] = a0 ? na0 : 0; h.p[1] = a1; h.n[1] = a1 ? na1 : 0;
    h.p[2] = b0; h.n[2] = b0 ? nb0 : 0; h.p[3] = b1; h.n[3] = b1 ? nb1 : 0;
    h.out = out2;
    return GX_OK;
}

extern "C" int gx_wgq_ring(int on) {
    GX_CHECK_ARG(on == 0 || on == 1, "gx_wgq_ring: 0 (64-pixel LDS-DMA tiles everywhere) or 1 (row-ring tiles where eligible)");
    g_wgq_ring = on;
    return GX_OK;
}

extern "C" int gx_wgq_policy(int mode) {
    GX_CHECK_ARG(mode >= 0 && mode <= 2, "gx_wgq_policy: mode must be 0 (round-1 kernels), 1 (LDS-DMA kernels, every queued "
                                         "layer in one stream-K launch) or 2 (LDS-DMA kernels, one launch per class and tile width)");
    g_wgq_mode = mode ? 1 : 0;
    if (mode) g_wgq_stream = mode == 1 ? 1 : 0;
    return GX_OK;
}
