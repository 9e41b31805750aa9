// "k-quad" tap convolutions: the chip-filling conv3x3 / transposed-conv 5x5 s2 layers (forward and data gradient) on
// v_mfma_f32_32x32x2_f32 with k-CONTIGUOUS operands, i.e. one 16-byte LDS read feeds four MFMAs.
//
// Round-1 finding: the tap-conv kernels of gx_conv.hip read every MFMA operand with a 4-byte
// ds_read and hipcc leaves those reads directly in front of their consumers (s_waitcnt lgkmcnt(0) every 2-4 MFMAs):
// matrix pipe 0.56-0.63 busy.  Here
//   * a chunk is 8 reduction channels = 4 MFMAs of k = 2: lane-half h of MFMA j takes channel 4h + j (the assignment
//     of channels to k slots is free as long as A and B agree), so a lane's four A values (and its four B values) are
//     4 CONSECUTIVE channels = one ds_read_b128 from a channel-quad-innermost LDS image:
//         input   [quad 2][halo position][4 channels]          (staged global -> registers -> ds_write_b128)
//         weights [tap][quad 2][64 output channels][4 channels] (pre-swizzled by the pack kernel: a straight copy)
//     -> 4 reads per 16 MFMAs (2 x 2 tiles of 32 x 32 per wave) instead of 16;
//   * the operand reads of tap t+1 are issued BEFORE the 16 MFMAs of tap t (two register sets, the order pinned with
//     sched_group_barrier), so no MFMA waits on LDS latency;
//   * no runtime branch inside a phase (one basic block per phase).
// A chunk is processed in PHASES that partition the taps so that the LDS footprint stays small enough for two
// workgroups per CU: transposed conv forward = one phase per kernel row kh (5 taps, the input tile is shared by the
// chunk's phases), its data gradient = one phase per parity plane of dy (9 / 6 / 6 / 4 taps, each with its own plane),
// conv3x3 = one phase of 9 taps.  Input tiles and weight slices are double-buffered separately; one barrier per phase.
//
// Reference ops: modules/blocks.py:159-165 (conv3x3), models/genesisv2_config.py:89-99 (ConvTranspose2d k5 s2 p2 op1).
#include "gx_common.h"

#include <cstdlib>

// measurement builds only (tools/abl_build.sh): 1 = no staging after the first stage (stale LDS, wrong results),
// 2 = also no barriers -- what the MFMA loop alone sustains, 3 = the product kernel; all of them also report the shader
// clock and the main-loop time of workgroup 0 in out[0..1]
#ifndef GX_KQ_ABL
#define GX_KQ_ABL 0
#endif
#ifndef GX_KQ_VAR
#define GX_KQ_VAR 0
#endif
// bf16-pipe path (measurement builds, wrong results): 1 = no output stores, 2 = no input split / LDS store after a tile's first
// chunk, 4 = no MFMAs, 8 = no input loads after the first chunk
#ifndef GX_QH_ABL
#define GX_QH_ABL 0
#endif

namespace {

enum { Q_C3 = 0, Q_DT0 = 1, Q_DT1 = 2, Q_DG = 3, Q_DT0H = 4, Q_DT1H = 5, Q_DGH = 6, Q_C3H = 7, Q_C5H = 8 };     // ..H: on the bf16 matrix pipe (q_body's B16 path)

template <int MODE> struct QCfg;
template <> struct QCfg<Q_C3> {
    static constexpr int NPH = 1, NT = 9, MAXT = 9, NCLS = 1;
    static constexpr bool PLANE_PER_PHASE = false;
    __host__ __device__ static constexpr int ntaps(int) { return 9; }
    __host__ __device__ static constexpr int tbase(int) { return 0; }
    __host__ __device__ static constexpr int ro(int, int i) { return i / 3; }
    __host__ __device__ static constexpr int co(int, int i) { return i % 3; }
    __host__ __device__ static constexpr int cls(int, int) { return 0; }
};
// ConvTranspose k5 s2 p2 op1, output rows 2r + a: kh = 2 p + a, halo row offset 2 - p; kw = i, halo column offset
// 2 - i / 2, output column parity i & 1 (same tap order as gx_conv.hip's packs 2 / 3).
template <> struct QCfg<Q_DT0> {
    static constexpr int NPH = 3, NT = 15, MAXT = 5, NCLS = 2;
    static constexpr bool PLANE_PER_PHASE = false;
    __host__ __device__ static constexpr int ntaps(int) { return 5; }
    __host__ __device__ static constexpr int tbase(int p) { return 5 * p; }
    __host__ __device__ static constexpr int ro(int p, int) { return 2 - p; }
    __host__ __device__ static constexpr int co(int, int i) { return 2 - i / 2; }
    __host__ __device__ static constexpr int cls(int, int i) { return i & 1; }
};
template <> struct QCfg<Q_DT1> {
    static constexpr int NPH = 2, NT = 10, MAXT = 5, NCLS = 2;
    static constexpr bool PLANE_PER_PHASE = false;
    __host__ __device__ static constexpr int ntaps(int) { return 5; }
    __host__ __device__ static constexpr int tbase(int p) { return 5 * p; }
    __host__ __device__ static constexpr int ro(int p, int) { return 2 - p; }
    __host__ __device__ static constexpr int co(int, int i) { return 2 - i / 2; }
    __host__ __device__ static constexpr int cls(int, int i) { return i & 1; }
};
// data gradient of the transposed conv: dx[r][c] = sum dy[2r-2+kh][2c-2+kw] W[kh][kw]; phase p = parity plane
// (kh & 1, kw & 1) of dy, in-plane offset (kh / 2, kw / 2); taps ordered plane-major (gx_kq_tap_slot).
template <> struct QCfg<Q_DG> {
    static constexpr int NPH = 4, NT = 25, MAXT = 9, NCLS = 1;
    static constexpr bool PLANE_PER_PHASE = true;
    __host__ __device__ static constexpr int nkw(int p) { return 3 - (p & 1); }
    __host__ __device__ static constexpr int ntaps(int p) { return (3 - (p >> 1)) * (3 - (p & 1)); }
    __host__ __device__ static constexpr int tbase(int p) { return p == 0 ? 0 : (p == 1 ? 9 : (p == 2 ? 15 : 21)); }
    __host__ __device__ static constexpr int ro(int p, int i) { return i / nkw(p); }
    __host__ __device__ static constexpr int co(int p, int i) { return i % nkw(p); }
    __host__ __device__ static constexpr int cls(int, int) { return 0; }
};

// The transposed conv forward on the bf16 matrix pipe (finding 13): every fp32 product from six bf16
// piece products, v_mfma_f32_32x32x16_bf16 (k = 16 channels: lane half h supplies the 8 consecutive channels of octet h).
// A chunk is 16 channels; the input tile is split into its three bf16 planes ONCE, by the staging
//     input   [piece 3][octet 2][halo position][8 channels (bf16)]      single-buffered (the next chunk waits in registers)
//     weights [tap][piece 3][octet 2][64 output channels][8 channels]   pre-split by the pack kernel (packs 22 / 23)
// so the tap loop has no VALU work: 12 ds_read_b128 per 24 MFMAs of 32 cycles.  Phases of <= 3 taps (a kernel row kh =
// taps {0,1,2} | {3,4}) keep input + double-buffered weights at 72 KB: two workgroups per CU as before.
template <int PA> struct QCfgDTH {
    static constexpr int NKH = PA ? 2 : 3;
    static constexpr int NPH = 2 * NKH, NT = 5 * NKH, MAXT = 3, NCLS = 2;
    static constexpr bool PLANE_PER_PHASE = false;
    __host__ __device__ static constexpr int ntaps(int ph) { return (ph & 1) ? 2 : 3; }
    __host__ __device__ static constexpr int tbase(int ph) { return 5 * (ph >> 1) + 3 * (ph & 1); }
    __host__ __device__ static constexpr int ro(int ph, int) { return 2 - (ph >> 1); }
    __host__ __device__ static constexpr int co(int ph, int j) { return 2 - (3 * (ph & 1) + j) / 2; }
    __host__ __device__ static constexpr int cls(int ph, int j) { return (3 * (ph & 1) + j) & 1; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
    // the phase after `ph` needs a new input tile (here: only the next chunk's first phase)
    __host__ __device__ static constexpr bool newin(int ph) { return ph + 1 == NPH; }
};
// its data gradient there: parity plane p of dy (kh & 1, kw & 1), one phase per kernel row inside the plane
// (3 + 3 + 2 + 2 rows of 3 / 2 / 3 / 2 taps; taps plane-major as in QCfg<Q_DG>); a new input tile per plane
template <> struct QCfg<Q_DGH> {
    static constexpr int NPH = 10, NT = 25, MAXT = 3, NCLS = 1;
    static constexpr bool PLANE_PER_PHASE = true;
    __host__ __device__ static constexpr int plane(int ph) { return ph < 3 ? 0 : (ph < 6 ? 1 : (ph < 8 ? 2 : 3)); }
    __host__ __device__ static constexpr int prow(int ph) { return ph < 3 ? ph : (ph < 6 ? ph - 3 : (ph < 8 ? ph - 6 : ph - 8)); }
    __host__ __device__ static constexpr int nkw(int p) { return 3 - (p & 1); }
    __host__ __device__ static constexpr int pbase(int p) { return p == 0 ? 0 : (p == 1 ? 9 : (p == 2 ? 15 : 21)); }
    __host__ __device__ static constexpr int ntaps(int ph) { return nkw(plane(ph)); }
    __host__ __device__ static constexpr int tbase(int ph) { return pbase(plane(ph)) + prow(ph) * nkw(plane(ph)); }
    __host__ __device__ static constexpr int ro(int ph, int) { return prow(ph); }
    __host__ __device__ static constexpr int co(int, int i) { return i; }
    __host__ __device__ static constexpr int cls(int, int) { return 0; }
    __host__ __device__ static constexpr bool newin(int ph) { return ph + 1 == NPH || plane(ph + 1) != plane(ph); }
};
// conv3x3 (forward / data gradient) of layers with <= 32 output channels there (the BroadcastDecoder's 32 -> 32 convs on the
// (S + 2L)^2 canvas, modules/decoders.py:21-35): one phase per kernel row; a workgroup owns 32 output channels (MI = 1), so the
// weight slices are half as large ([tap][piece 3][octet 2][32 output channels][8 channels], pack kinds 20 / 21) and input
// + weights stay at 72 KB with the 4-slot halo tiles of grids that are not powers of two (8 x 8 pixels x 4 images)
template <> struct QCfg<Q_C3H> {
    static constexpr int NPH = 3, NT = 9, MAXT = 3, NCLS = 1;
    static constexpr bool PLANE_PER_PHASE = false;
    __host__ __device__ static constexpr int ntaps(int) { return 3; }
    __host__ __device__ static constexpr int tbase(int ph) { return 3 * ph; }
    __host__ __device__ static constexpr int ro(int ph, int) { return ph; }
    __host__ __device__ static constexpr int co(int, int j) { return j; }
    __host__ __device__ static constexpr int cls(int, int) { return 0; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
    __host__ __device__ static constexpr bool newin(int ph) { return ph + 1 == NPH; }
};
// 5 x 5 stride-1 pad-2 conv there (the gated stacks of third_party/sylvester, VAE.py:18-33: forward and data gradient are the
// two weight packings 27 / 28 of one kernel): a 2-pixel halo, one phase per (kernel row, taps {0,1,2} | {3,4}) = 10 phases of
// a 16-channel chunk on one input tile; 64-channel workgroups like the transposed convs
template <> struct QCfg<Q_C5H> {
    static constexpr int NPH = 10, NT = 25, MAXT = 3, NCLS = 1;
    static constexpr bool PLANE_PER_PHASE = false;
    __host__ __device__ static constexpr int ntaps(int ph) { return (ph & 1) ? 2 : 3; }
    __host__ __device__ static constexpr int tbase(int ph) { return 5 * (ph >> 1) + 3 * (ph & 1); }
    __host__ __device__ static constexpr int ro(int ph, int) { return ph >> 1; }
    __host__ __device__ static constexpr int co(int ph, int j) { return 3 * (ph & 1) + j; }
    __host__ __device__ static constexpr int cls(int, int) { return 0; }
    __host__ __device__ static constexpr int plane(int) { return 0; }
    __host__ __device__ static constexpr bool newin(int ph) { return ph + 1 == NPH; }
};
template <> struct QCfg<Q_DT0H> : QCfgDTH<0> {};
template <> struct QCfg<Q_DT1H> : QCfgDTH<1> {};
typedef __bf16 q_bf16x8 __attribute__((ext_vector_type(8)));
// fp16 x 3 (template parameter F16 of q_body / q_phase_h; gx_kq_precision(2), DESIGN.md finding 40): every fp32 product from THREE
// fp16 piece products -- x * 2^sx = hi + lo (11 + 11 significant bits), hi*hi + hi*lo + lo*hi; the dropped lo*lo term is 2^-22 of the
// product -- instead of six bf16 ones.  sx, sw: per-tensor power-of-two scales from the tensors' largest magnitudes (gx_f16_scale_exp),
// taken out of the accumulators before the epilogue.  Same LDS layout (the third piece plane stays unused).
typedef _Float16 q_f16x8 __attribute__((ext_vector_type(8)));
constexpr int QH_TAP_BYTES = 3 * 2 * 64 * 16;          // one tap of one 16-channel chunk: 6144 B
static_assert(QH_TAP_BYTES == 6144, "gx_kq_h_amax_off (gx_common.h) spells this number out");
// output-channel rows of a workgroup's weight slice and the bytes of one (tap, chunk) of it
// (NP pieces per value: three bf16 ones, two fp16 ones)
template <int MODE, bool F16 = false> struct QHLay { static constexpr int ROWS = MODE == Q_C3H ? 32 : 64, NP = F16 ? 2 : 3, TAPB = NP * 2 * ROWS * 16; };

// one phase on the bf16 pipe: operands of (tap, mi / nj, piece) straight out of LDS
template <int MODE, int PH, int NCLS, int MI, bool F16 = false>
__device__ __forceinline__ void q_phase_h(f32x16 (&acc)[NCLS][MI][2], const char* ib, const char* wb, const int plane_bytes,
                                          const int a_lane_b, const int b_lane0_b, const int b_lane1_b, const int HS16) {
    using C = QCfg<MODE>;
    constexpr int nt = C::ntaps(PH);
#pragma unroll
    for (int i = 0; i < nt; ++i) {
        const int toff = (C::ro(PH, i) * HS16 + C::co(PH, i) * 16);
        q_bf16x8 a[MI][3], b[2][3];
#pragma unroll
        for (int pc = 0; pc < (F16 ? 2 : 3); ++pc) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi][pc] = *reinterpret_cast<const q_bf16x8*>(wb + i * QHLay<MODE, F16>::TAPB + pc * (QHLay<MODE, F16>::TAPB / QHLay<MODE, F16>::NP) + a_lane_b + mi * 512);
            b[0][pc] = *reinterpret_cast<const q_bf16x8*>(ib + pc * plane_bytes + b_lane0_b + toff);
            b[1][pc] = *reinterpret_cast<const q_bf16x8*>(ib + pc * plane_bytes + b_lane1_b + toff);
        }
        const int cl = C::cls(PH, i);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                f32x16 c = acc[cl][mi][nj];      // pieces: 0 hi, 1 mid, 2 lo; small terms first
#if GX_QH_ABL & 4
                c[0] += (float)a[mi][0][0] * (float)b[nj][0][0] + (float)a[mi][1][1] * (float)b[nj][1][1] + (float)a[mi][2][2] * (float)b[nj][2][2];
#else
                if constexpr (F16) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(q_f16x8, a[mi][1]), __builtin_bit_cast(q_f16x8, b[nj][0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(q_f16x8, a[mi][0]), __builtin_bit_cast(q_f16x8, b[nj][1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(q_f16x8, a[mi][0]), __builtin_bit_cast(q_f16x8, b[nj][0]), c, 0, 0, 0);
                } else {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[nj][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[nj][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[nj][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[nj][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[nj][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[nj][0], c, 0, 0, 0);
                }
#endif
                acc[cl][mi][nj] = c;
            }
    }
}

struct QGeom {
    int x_amax_n;         // partial maxima at x_amax (kAmaxParts from gx_kq_amax_launch, or a producer's count)
    int ilv;              // kq_dth_kernel: row parities interleaved along blockIdx.x (grid.z = 1)
    const float* x_amax; const float* w_amax;      // fp16 x 3 form: the input tensor's / the weight tensor's largest magnitude (device)
    int N, K, M;          // images, reduction channels (a multiple of 8), output channels
    int nchunks;          // K / 8
    int Hb, Wb;           // base (pixel-tile) grid
    int Hi, Wi, Ho, Wo;   // input / output tensor dims
    int lTH, lTW, lG, tiles_h, tiles_w;
    int rt_th, rt_tw;     // Q_C3H on a grid that is not a power of two: the tile is rt_th whole rows of rt_tw = Wb pixels (0: 2^l tiles)
    int act;              // epilogue activation after the bias: 0 none, 1 ReLU, 2 ELU
    int nfull;            // blockIdx.x < nfull: whole tiles; the rest: pairs of half-work workgroups (q_split_tail)
    float* stats;         // STATS: per-workgroup (sum, sum of squares) of every 8-channel block, [N][parts][M/8][2]
    int stats_parts;
    const float* mask;    // Q_C3H as a data gradient: out *= act'(mask) -- mask = the producing layer's activation OUTPUT, laid
    int mask_act;         // out like `out` (the backward of a bias + ReLU / ELU layer without its own pass: gx_conv3x3_dgrad_act)
    float* o_amax;        // Q_C3H as a PRODUCER (an armed gx_amax_tap): one partial maximum of |stored value| per workgroup, or NULL
    int tap_lds_off;      //   ... its four-float LDS slot, in floats from the start of the dynamic LDS (the launch adds 16 bytes)
};

__device__ __forceinline__ float q_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : expm1f(v);
    return v;
}
__device__ __forceinline__ float q_dact(float o, int act) {      // act'(.) from the activation's OUTPUT (gx_misc.hip act_bwd_from_out)
    const float neg = act == 2 ? o + 1.f : (act == 1 ? 0.f : 1.f);          // selects, no branches: 32 of these per lane and tile
    return o > 0.f ? 1.f : neg;
}

// one phase: the taps of phase PH out of the input tile `ib` and the weight slice `wb`; the operands of tap i + 1 are
// read ahead of tap i's MFMAs (two register sets; order pinned by GX_Q_SCHED)
template <int MODE, int PH, int NCLS, int MI>
__device__ __forceinline__ void q_phase(f32x16 (&acc)[NCLS][MI][2], const float* ib, const float* wb, const int a_lane,
                                        const int b_lane0, const int b_lane1, const int HS4) {
    using C = QCfg<MODE>;
    constexpr int nt = C::ntaps(PH);
    f32x4 fa[2][MI], fb[2][2];
#define GX_Q_READ(i_, set_)                                                                            \
    {                                                                                                  \
        const int toff_ = C::ro(PH, i_) * HS4 + C::co(PH, i_) * 4;                                     \
        _Pragma("unroll") for (int mi_ = 0; mi_ < MI; ++mi_)                                           \
            fa[set_][mi_] = *reinterpret_cast<const f32x4*>(wb + (i_) * 512 + a_lane + mi_ * 128);     \
        fb[set_][0] = *reinterpret_cast<const f32x4*>(ib + b_lane0 + toff_);                           \
        fb[set_][1] = *reinterpret_cast<const f32x4*>(ib + b_lane1 + toff_);                           \
    }
    GX_Q_READ(0, 0)
    __builtin_amdgcn_sched_group_barrier(0x100, 2 + MI, 0);
#pragma unroll
    for (int i = 0; i < nt; ++i) {
        const int cur = i & 1;
        if (i + 1 < nt) GX_Q_READ(i + 1, cur ^ 1)
        const int cl = C::cls(PH, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[cl][mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][mi][j], fb[cur][0][j], acc[cl][mi][0], 0, 0, 0);
                acc[cl][mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][mi][j], fb[cur][1][j], acc[cl][mi][1], 0, 0, 0);
            }
        }
#if GX_KQ_VAR == 1      /* one operand read behind every fourth MFMA */
        for (int r = 0; r < 4; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (i + 1 < nt) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#elif GX_KQ_VAR == 2    /* no pinning: hipcc's own placement */
#else
        if (i + 1 < nt) __builtin_amdgcn_sched_group_barrier(0x100, 2 + MI, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * MI, 0);
#endif
    }
#undef GX_Q_READ
}

// the input tensor's largest magnitude from n partial maxima (gx_kq_amax_launch's kAmaxParts = 256, or one per workgroup of the kernel
// that produced the tensor: gx_kq_amax_link): strided loads, wave reduction
__device__ __forceinline__ float q_amax_parts(const float* __restrict__ parts, int n) {
    float m = 0.f;
    int i0 = 0;
    if ((reinterpret_cast<uintptr_t>(parts) & 15) == 0) {       // (thousands of partials -- the generic GroupNorm kernels': 16-byte loads)
        const int n4 = n >> 2;
        for (int i = threadIdx.x & 63; i < n4; i += 64) {
            const f32x4 v = reinterpret_cast<const f32x4*>(parts)[i];
            m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        }
        i0 = n4 << 2;
    }
    for (int i = i0 + (threadIdx.x & 63); i < n; i += 64) m = fmaxf(m, parts[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}

// NQ: (position, quad) slots staged per thread per input tile (2 * CHS <= NQ * 256)
// MI: 32-channel MFMA tiles per wave along M.  2 = the workgroup's whole 64-channel tile; 1 = one half (mh) of it --
// the last tiles of a grid that does not divide the chip are split into two half-work workgroups (q_split_tail).
template <int MODE, int NQ, bool STATS, int MI = 2, bool F16 = false, bool TAP = false>
__device__ __forceinline__ void q_body(const float* __restrict__ in, const float* __restrict__ wp,
                                       const float* __restrict__ bias, float* __restrict__ out, const QGeom& g,
                                       float* lds, const int bx, const int by, const int par_a, const int mh = 0) {
    using C = QCfg<MODE>;
    constexpr bool B16 = MODE >= Q_DT0H;               // bf16 matrix pipe, 16-channel chunks (QCfgDTH)
    constexpr int NPH = C::NPH, NT = C::NT, MAXT = C::MAXT, NCLS = C::NCLS;
    constexpr int NW = (MAXT * 128 + 255) / 256;       // float4 weight loads per thread per phase
    constexpr int WSLOT = NW * 1024;                   // floats per weight buffer (whole float4-per-thread rounds: the
    constexpr int ISLOT = NQ * 1024;                   //   staging stores are unconditional); same for an input buffer

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // row tiles (Q_C3H, grid not a power of two): rt_th whole image rows per workgroup -- every load / store instruction
    // covers 128 contiguous bytes of a channel plane (8 x 8-pixel tiles on the 72 x 72 canvas: four 32-byte segments);
    // pixel slots >= rt_th * rt_tw of the 256 are idle
    const bool RT = MODE == Q_C3H && g.rt_tw > 0;
    const int TH = RT ? g.rt_th : 1 << g.lTH, TW = RT ? g.rt_tw : 1 << g.lTW, G = RT ? 1 : 1 << g.lG;
    constexpr int HL = MODE == Q_C5H ? 2 : 1;          // halo pixels on every side
    const int HS = TW + 2 * HL;
    const int CHS = G * (TH + 2 * HL) * HS;            // halo positions per plane
    float* const ibuf0 = lds;
    float* const wbuf0 = lds + 2 * ISLOT;

    int tile = bx;
    const int tw_i = tile % g.tiles_w; tile /= g.tiles_w;
    const int th_i = tile % g.tiles_h; tile /= g.tiles_h;
    const int img0 = tile * G;
    const int R0 = th_i * TH, C0 = tw_i * TW;
    const int m0 = by * (MODE == Q_C3H ? 32 : 64);
    const int HiWi = g.Hi * g.Wi;

    // ---- staging slots: slot = tid + q * 256 -> (quad, position).  Loads go through buffer descriptors (wave-uniform
    //      base, 32-bit per-lane byte offset, scalar offset for the channel / chunk): a position outside the image gets
    //      an out-of-range offset and the hardware returns 0 -- no select, no branch, no 64-bit address per load
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in), 0, (int)((unsigned)g.N * (unsigned)g.K * (unsigned)HiWi * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(wp), 0,
        B16 ? (int)((unsigned)NT * (unsigned)(g.K / 16) * gridDim.y * (unsigned)QHLay<MODE, F16>::TAPB)
            : (int)((unsigned)NT * (unsigned)g.K * gridDim.y * 256u), 0x00020000);
    int voff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int slot = tid + q * 256;
        const int quad = slot >= CHS ? 1 : 0;
        int rem = slot - quad * CHS;
        bool ok = slot < 2 * CHS;
        const int gi = rem / ((TH + 2 * HL) * HS); rem -= gi * (TH + 2 * HL) * HS;
        const int i = rem / HS;
        const int j = rem - i * HS;
        int row, col;
        if (MODE == Q_DG || MODE == Q_DGH) { row = 2 * (R0 + i) - 2; col = 2 * (C0 + j) - 2; }
        else { row = R0 - HL + i; col = C0 - HL + j; }
        ok = ok && img0 + gi < g.N && row >= 0 && row < g.Hi && col >= 0 && col < g.Wi;
        voff[q] = ok ? (((img0 + gi) * g.K + quad * (B16 ? 8 : 4)) * HiWi + row * g.Wi + col) * 4 : (int)0x80000000;
    }
    const int w_voff = tid * 16;
    const int w_sbase = by * g.nchunks * (NT * 2048);

    f32x16 acc[NCLS][MI][2];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][i][j][e] = 0.f;
    int f16_sx = 0;
    if constexpr (B16 && F16) f16_sx = gx_f16_scale_exp(q_amax_parts(g.x_amax, g.x_amax_n));
    if constexpr (B16) {
        // ---- the bf16-pipe pipeline: chunks of 16 channels, phases of <= 3 taps
        constexpr int TAPB = QHLay<MODE, F16>::TAPB;
        constexpr int NPL = QHLay<MODE, F16>::NP;                          // input piece planes in LDS
        constexpr int NWH = (MAXT * (TAPB / 16) + 255) / 256;              // 16-byte weight pieces per thread per phase
        constexpr int WSLOTB = NWH * 256 * 16;                             // bytes per weight buffer
        // four staging rounds (the 64-pixel-wide tiles of a 64 x 64 base grid: 6 x 66 halo positions): exact planes, so that three
        // of them + the weight buffers still leave room for two workgroups per CU
        constexpr bool TRIM = NQ == 4 && MODE != Q_C3H;
        const int plane_bytes = TRIM ? 2 * CHS * 16 : NQ * 256 * 16;       // bytes per input piece plane
        char* const ibuf = reinterpret_cast<char*>(lds);
        char* const wbufb = ibuf + NPL * plane_bytes;
        const int quad_l = lane >> 5;
        const int a_lane_b = (quad_l * QHLay<MODE, F16>::ROWS + (lane & 31)) * 16 + mh * 512;   // + mi * 512 + piece * TAPB / 3 + tap * TAPB
        int b_lane_b[2];
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int p = wave * 64 + nj * 32 + (lane & 31);
            int c = p & (TW - 1);
            int r = (p >> g.lTW) & (TH - 1);
            int gi = p >> (g.lTW + g.lTH);
            if (RT) { gi = 0; r = p / TW; c = p - r * TW; if (r >= TH) { r = 0; c = 0; } }      // (idle slots read position 0)
            b_lane_b[nj] = (quad_l * CHS + (gi * (TH + 2 * HL) + r) * HS + c) * 16;
        }
        const int HS16 = HS * 16;
        const int nsc = g.K / 16;
        const int w_sbase = by * nsc * (NT * TAPB);
        float xin[NQ][8];
        f32x4 wreg[NWH];
#define GX_QH_LOAD_IN(sc_, plane_)                                                                     \
        {                                                                                              \
            const int po_ = MODE == Q_DGH ? (((plane_) >> 1) * g.Wi + ((plane_) & 1)) * 4 : 0;         \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                             \
                _Pragma("unroll") for (int e = 0; e < 8; ++e)                                          \
                    xin[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, voff[q], ((sc_) * 16 + e) * HiWi * 4 + po_, 0)); \
        }
#define GX_QH_STORE_IN()                                                                               \
        {                                                                                              \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                           \
                q_bf16x8 ph_, pm_, pl_;                                                                \
                if (F16) {                                                                             \
                    q_f16x8 fh_, fl_;                                                                  \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                    \
                        const float xs_ = ldexpf(xin[q][e], f16_sx);                                   \
                        const _Float16 h_ = (_Float16)xs_;                                             \
                        fh_[e] = h_; fl_[e] = (_Float16)(xs_ - (float)h_);                             \
                    }                                                                                  \
                    ph_ = __builtin_bit_cast(q_bf16x8, fh_); pm_ = __builtin_bit_cast(q_bf16x8, fl_);  \
                } else {                                                                               \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                        \
                    const __bf16 h_ = (__bf16)xin[q][e];                                               \
                    const float r1_ = xin[q][e] - (float)h_;                                           \
                    const __bf16 m_ = (__bf16)r1_;                                                     \
                    ph_[e] = h_; pm_[e] = m_; pl_[e] = (__bf16)(r1_ - (float)m_);                      \
                }                                                                                      \
                }                                                                                      \
                char* d_ = ibuf + (tid + q * 256) * 16;                                                \
                if (!TRIM || tid + q * 256 < 2 * CHS) {                                                \
                *reinterpret_cast<q_bf16x8*>(d_) = ph_;                                                \
                *reinterpret_cast<q_bf16x8*>(d_ + plane_bytes) = pm_;                                  \
                if (!F16) *reinterpret_cast<q_bf16x8*>(d_ + 2 * plane_bytes) = pl_;                    \
                }                                                                                      \
            }                                                                                          \
        }
#define GX_QH_LOAD_W(sc_, ph_)                                                                         \
        {                                                                                              \
            const int so_ = w_sbase + ((sc_) * NT + C::tbase(ph_)) * TAPB;                             \
            _Pragma("unroll") for (int i = 0; i < NWH; ++i)                                            \
                wreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (tid + i * 256) * 16, so_, 0)); \
        }
#define GX_QH_STORE_W(dst_)                                                                            \
        {                                                                                              \
            _Pragma("unroll") for (int i = 0; i < NWH; ++i)                                            \
                *reinterpret_cast<f32x4*>((dst_) + (tid + i * 256) * 16) = wreg[i];                    \
        }
        GX_QH_LOAD_IN(0, 0)
        GX_QH_LOAD_W(0, 0)
        GX_QH_STORE_IN()
        GX_QH_STORE_W(wbufb)
        // Q_C3H: the next chunk's input loads are issued at the chunk's FIRST phase (the registers are free: 32 accumulators) --
        // with two chunks per tile a load issued at the last phase has one phase of MFMAs to hide under
        // ... and, since round 6, the transposed-conv forward: its five phases (kernel rows) read ONE input tile per chunk, so the next
        // chunk's loads used to be issued at the last phase with ~0.5 us of MFMAs to hide under (PMC: 48 % of the wave cycles
        // parked); issued at the chunk's first phase they have the whole chunk (24 more live registers, still two workgroups
        // per CU): 11 852 -> 12 004 img/s on the metric step in one A/B call (-DGX_QH_EARLY_DTH=0: the old placement)
#ifndef GX_QH_EARLY_DTH
#define GX_QH_EARLY_DTH 1
#endif
        constexpr bool EARLY_IN = MODE == Q_C3H || (GX_QH_EARLY_DTH && (MODE == Q_DT0H || MODE == Q_DT1H));
        // Q_C3H: ONE weight buffer (a second barrier per phase instead): 36 + 12 KB of LDS = three workgroups per CU -- a tile
        // is short (two chunks), its loads, bf16 split, MFMAs and stores barely overlap inside one workgroup, so the third
        // workgroup is what fills the gaps
        constexpr bool ONE_W = MODE == Q_C3H;
        int s = 0;
        for (int sc = 0; sc < nsc; ++sc) {
            const bool last_chunk = sc + 1 == nsc;
#define GX_QH_STAGE(PH_)                                                                                         \
            {                                                                                                    \
                constexpr int NXT = (PH_) + 1 < NPH ? (PH_) + 1 : 0;                                             \
                constexpr bool last_ph = (PH_) + 1 == NPH;                                                       \
                const bool more = !last_ph || !last_chunk;                                                       \
                const bool in_next = C::newin(PH_) && more;      /* the next phase reads another input tile */   \
                __syncthreads();                                                                                 \
                if (more) GX_QH_LOAD_W(last_ph ? sc + 1 : sc, NXT)                                               \
                if (EARLY_IN) { if ((PH_) == 0 && !last_chunk && !(GX_QH_ABL & 8)) GX_QH_LOAD_IN(sc + 1, 0) }    \
                else if (in_next) GX_QH_LOAD_IN(last_ph ? sc + 1 : sc, C::plane(NXT))                            \
                q_phase_h<MODE, (PH_), NCLS, MI, F16>(acc, ibuf, wbufb + (ONE_W ? 0 : (s & 1)) * WSLOTB, plane_bytes, \
                                                 a_lane_b, b_lane_b[0], b_lane_b[1], HS16);                      \
                if (ONE_W) {                      /* one weight buffer: everyone is done with it (and the tile) */ \
                    if (more) { __syncthreads(); GX_QH_STORE_W(wbufb) }                                          \
                    if (in_next && !(GX_QH_ABL & 2)) GX_QH_STORE_IN()                                            \
                } else {                                                                                         \
                if (more) GX_QH_STORE_W(wbufb + ((s + 1) & 1) * WSLOTB)                                          \
                if (in_next) {                    /* the input tile is single-buffered: everyone is done with it */ \
                    __syncthreads();                                                                             \
                    if (!(GX_QH_ABL & 2)) GX_QH_STORE_IN()                                                       \
                }                                                                                                \
                }                                                                                                \
                ++s;                                                                                             \
            }
            GX_QH_STAGE(0)
            GX_QH_STAGE(1)
            GX_QH_STAGE(2)
            if constexpr (NPH > 3) GX_QH_STAGE(3)
            if constexpr (NPH > 4) { GX_QH_STAGE(4) GX_QH_STAGE(5) }
            if constexpr (NPH > 6) { GX_QH_STAGE(6) GX_QH_STAGE(7) GX_QH_STAGE(8) GX_QH_STAGE(9) }
#undef GX_QH_STAGE
        }
#undef GX_QH_LOAD_IN
#undef GX_QH_STORE_IN
#undef GX_QH_LOAD_W
#undef GX_QH_STORE_W
    } else {
    // ---- per-lane operand addresses (float indices)
    const int quad_l = lane >> 5;
    const int a_lane = (quad_l * 64 + (lane & 31)) * 4 + mh * 128;   // + mi * 128 + tap * 512
    int b_lane[2];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int p = wave * 64 + nj * 32 + (lane & 31);
        const int c = p & (TW - 1);
        const int r = (p >> g.lTW) & (TH - 1);
        const int gi = p >> (g.lTW + g.lTH);
        b_lane[nj] = (quad_l * CHS + (gi * (TH + 2 * HL) + r) * HS + c) * 4;
    }
    const int HS4 = HS * 4;


    f32x4 xin[NQ];
    f32x4 wreg[NW];
#if GX_KQ_ABL
    const long long abl_c0 = __builtin_readcyclecounter();
    const long long abl_w0 = wall_clock64();
#endif

    // global -> registers: the input tile of (chunk, plane) / the weight slice of (chunk, phase)
#define GX_Q_LOAD_IN(chunk_, plane_)                                                                   \
    {                                                                                                  \
        const int so_ = ((chunk_) * 8 * HiWi + (MODE == Q_DG ? ((plane_) >> 1) * g.Wi + ((plane_) & 1) : 0)) * 4; \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                               \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                              \
                xin[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, voff[q], so_ + e * HiWi * 4, 0)); \
        }                                                                                              \
    }
#define GX_Q_STORE_IN(dst_)                                                                            \
    {                                                                                                  \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                                 \
            *reinterpret_cast<f32x4*>((dst_) + (tid + q * 256) * 4) = xin[q];                          \
    }
#define GX_Q_LOAD_W(chunk_, ph_)                                                                       \
    {                                                                                                  \
        const int so_ = w_sbase + ((chunk_) * NT + C::tbase(ph_)) * 2048;                              \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                 \
            wreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff + i * 4096, so_, 0)); \
    }
#define GX_Q_STORE_W(dst_, ph_)                                                                        \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                 \
            *reinterpret_cast<f32x4*>((dst_) + (tid + i * 256) * 4) = wreg[i];                         \
    }
    // ---- pipeline over (chunk, phase): stage s = chunk * NPH + phase; weights alternate per stage, input tiles per
    //      chunk (shared by the chunk's phases) or per stage (PLANE_PER_PHASE)
    GX_Q_LOAD_IN(0, 0)
    GX_Q_LOAD_W(0, 0)
    GX_Q_STORE_IN(ibuf0)
    GX_Q_STORE_W(wbuf0, 0)
    int s = 0;
    for (int chunk = 0; chunk < g.nchunks; ++chunk) {
        const bool last_chunk = chunk + 1 == g.nchunks;
#define GX_Q_STAGE(PH_)                                                                                          \
        {                                                                                                        \
            constexpr int NXT = (PH_) + 1 < NPH ? (PH_) + 1 : 0;                                                 \
            const bool more = (GX_KQ_ABL == 0 || GX_KQ_ABL == 3) && ((PH_) + 1 < NPH || !last_chunk);            \
            const int nchunk = (PH_) + 1 < NPH ? chunk : chunk + 1;                                              \
            constexpr bool new_in = C::PLANE_PER_PHASE || NXT == 0;   /* the next stage needs a new input tile */ \
            const int icur = C::PLANE_PER_PHASE ? (s & 1) : (chunk & 1);                                         \
            const int inxt = C::PLANE_PER_PHASE ? ((s + 1) & 1) : ((chunk + 1) & 1);                             \
            if (GX_KQ_ABL != 2) __syncthreads();                                                                 \
            if (more) {      /* issue the next stage's global loads FIRST (hipcc otherwise sinks them to the    \
                                commit below and the phase ends on their latency) */                             \
                if (new_in) GX_Q_LOAD_IN(nchunk, NXT)                                                            \
                GX_Q_LOAD_W(nchunk, NXT)                                                                         \
                __builtin_amdgcn_sched_group_barrier(0x020, (new_in ? NQ * 4 : 0) + NW, 0);                      \
            }                                                                                                    \
            q_phase<MODE, (PH_), NCLS, MI>(acc, ibuf0 + icur * ISLOT, wbuf0 + (s & 1) * WSLOT, a_lane, b_lane[0],    \
                                       b_lane[1], HS4);                                                  \
            if (more) {                                                                                          \
                if (new_in) GX_Q_STORE_IN(ibuf0 + inxt * ISLOT)                                                  \
                GX_Q_STORE_W(wbuf0 + ((s + 1) & 1) * WSLOT, NXT)                                                 \
                __builtin_amdgcn_sched_group_barrier(0x200, (new_in ? NQ : 0) + NW, 0);                          \
            }                                                                                                    \
            ++s;                                                                                                 \
        }
        GX_Q_STAGE(0)
        if constexpr (NPH > 1) GX_Q_STAGE(1)
        if constexpr (NPH > 2) GX_Q_STAGE(2)
        if constexpr (NPH > 3) GX_Q_STAGE(3)
#undef GX_Q_STAGE
    }
#undef GX_Q_LOAD_IN
#undef GX_Q_STORE_IN
#undef GX_Q_LOAD_W
#undef GX_Q_STORE_W
    }

#if GX_KQ_ABL
    const long long abl_c1 = __builtin_readcyclecounter();
    const long long abl_w1 = wall_clock64();
#endif
    // ---- epilogue: C/D layout col = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
    const size_t out_img_stride = (size_t)g.M * g.Ho * g.Wo;
    const int HoWo = g.Ho * g.Wo;
    const bool add_bias = bias != nullptr;
    const int act = g.act;
    float omax_t = 0.f;           // Q_C3H producer tap: largest |value| this thread stores for this tile
    f32x4 bvec[MI][4];
    {
        const bool vec_ok = add_bias && (reinterpret_cast<uintptr_t>(bias) & 15) == 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mb = m0 + (mi + mh) * 32 + 8 * q + 4 * (lane >> 5);
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (vec_ok && mb + 3 < g.M) t = *reinterpret_cast<const f32x4*>(bias + mb);
                else if (add_bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = mb + e < g.M ? bias[mb + e] : 0.f;
                }
                bvec[mi][q] = t;
            }
    }
    if constexpr (B16 && F16) {
        const int de = -(f16_sx + gx_f16_scale_exp(*g.w_amax));
#pragma unroll
        for (int c = 0; c < NCLS; ++c)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[c][i][j][e] = ldexpf(acc[c][i][j][e], de);
    }
    // interior tiles with all 32 (MI) channels of every accumulator row present (the common case: power-of-two grids,
    // M a multiple of 64) store without per-element guards -- the guarded loop costs ~1400 VALU + 48 branches per wave
    const bool full_tile = !RT && img0 + G <= g.N && R0 + TH <= g.Hb && C0 + TW <= g.Wb && m0 + (MI + mh) * 32 <= g.M;
#if GX_QH_ABL & 1
    if (B16 && acc[0][0][0][0] != 12345.678f) return;
#endif
    if (RT) {
        // row tiles: pixel slot p = row r, column c of the tile's rt_th rows; consecutive lanes = consecutive floats of a plane
        if constexpr (NCLS == 1) {
            size_t ooff[2];
            bool pok[2];
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                const int p = wave * 64 + nj * 32 + (lane & 31);
                const int r = p / TW, c = p - r * TW;
                pok[nj] = r < TH && R0 + r < g.Hb;
                ooff[nj] = (size_t)img0 * out_img_stride + (size_t)(R0 + r) * g.Wo + c + (size_t)(m0 + mh * 32 + 4 * (lane >> 5)) * HoWo;
            }
            // Q_C3H as the data gradient behind a bias + activation layer: act'(that layer's output) for the 16 values of a
            // pixel in one round of loads ahead of its stores -- unconditional (a guarded load is load - wait - branch, 16 times
            // over: the address is clamped instead) and before any store (the compiler cannot prove the mask does not alias
            // out).  (All 32 of both pixels in one round: 120 B more scratch in this 168-register kernel.)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                float mk[MI][16];
                if (MODE == Q_C3H && g.mask) {
                    const float* mbase = g.mask + (pok[nj] ? ooff[nj] : (size_t)0);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int mrow = mi * 32 + (reg & 3) + 8 * (reg >> 2);
                            mk[mi][reg] = mbase[pok[nj] && m0 + mh * 32 + 4 * (lane >> 5) + mrow < g.M ? (size_t)mrow * HoWo : (size_t)0];
                        }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) mk[mi][reg] = q_dact(mk[mi][reg], g.mask_act);
                }
                if (pok[nj]) {
                    float* obase = out + ooff[nj];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int mrow = mi * 32 + (reg & 3) + 8 * (reg >> 2);
                            if (m0 + mh * 32 + 4 * (lane >> 5) + mrow < g.M) {
                                float v = q_act(acc[0][mi][nj][reg] + bvec[mi][reg >> 2][reg & 3], act);
                                if (MODE == Q_C3H && g.mask) v *= mk[mi][reg];
                                obase[(size_t)mrow * HoWo] = v;
                                if constexpr (TAP) omax_t = fmaxf(omax_t, fabsf(v));
                            }
                        }
                }
            }
        }
    } else if (full_tile) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int p = wave * 64 + nj * 32 + (lane & 31);
            const int c = p & (TW - 1);
            const int r = (p >> g.lTW) & (TH - 1);
            const int n = img0 + (p >> (g.lTW + g.lTH));
            int orow, ocol;
            if (NCLS == 2) { orow = 2 * (R0 + r) + par_a; ocol = 2 * (C0 + c); }
            else { orow = R0 + r; ocol = C0 + c; }
            float* obase = out + (size_t)n * out_img_stride + (size_t)orow * g.Wo + ocol +
                           (size_t)(m0 + mh * 32 + 4 * (lane >> 5)) * HoWo;
            float mk[MI][16];
            if (MODE == Q_C3H && g.mask) {
                const float* mbase = g.mask + (obase - out);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        mk[mi][reg] = mbase[(size_t)(mi * 32 + (reg & 3) + 8 * (reg >> 2)) * HoWo];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) mk[mi][reg] = q_dact(mk[mi][reg], g.mask_act);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float* o = obase + (size_t)(mi * 32 + (reg & 3) + 8 * (reg >> 2)) * HoWo;
                    const float bv = bvec[mi][reg >> 2][reg & 3];
                    if (NCLS == 2) {
                        float2 v;
                        v.x = q_act(acc[0][mi][nj][reg] + bv, act);
                        v.y = q_act(acc[NCLS - 1][mi][nj][reg] + bv, act);
                        *reinterpret_cast<float2*>(o) = v;
                    } else {
                        float v = q_act(acc[0][mi][nj][reg] + bv, act);
                        if (MODE == Q_C3H && g.mask) v *= mk[mi][reg];
                        *o = v;
                        if constexpr (TAP) omax_t = fmaxf(omax_t, fabsf(v));
                    }
                }
            }
        }
    } else {
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int p = wave * 64 + nj * 32 + (lane & 31);
        const int c = p & (TW - 1);
        const int r = (p >> g.lTW) & (TH - 1);
        const int gi = p >> (g.lTW + g.lTH);
        const int n = img0 + gi;
        if (n >= g.N || R0 + r >= g.Hb || C0 + c >= g.Wb) continue;
        int orow, ocol;
        if (NCLS == 2) { orow = 2 * (R0 + r) + par_a; ocol = 2 * (C0 + c); }
        else { orow = R0 + r; ocol = C0 + c; }
        float* obase = out + (size_t)n * out_img_stride + (size_t)orow * g.Wo + ocol;
        float mk[MI][16];
        if (MODE == Q_C3H && g.mask) {
            const float* mbase = g.mask + (obase - out);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m = m0 + (mi + mh) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    mk[mi][reg] = mbase[m < g.M ? (size_t)m * HoWo : (size_t)0];
                }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) mk[mi][reg] = q_dact(mk[mi][reg], g.mask_act);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + (mi + mh) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m < g.M) {
                    const float bv = bvec[mi][reg >> 2][reg & 3];
                    if (NCLS == 2) {
                        float2 v;
                        v.x = q_act(acc[0][mi][nj][reg] + bv, act);
                        v.y = q_act(acc[NCLS - 1][mi][nj][reg] + bv, act);
                        *reinterpret_cast<float2*>(obase + (size_t)m * HoWo) = v;
                    } else {
                        float v = q_act(acc[0][mi][nj][reg] + bv, act);
                        if (MODE == Q_C3H && g.mask) v *= mk[mi][reg];
                        obase[(size_t)m * HoWo] = v;
                        if constexpr (TAP) omax_t = fmaxf(omax_t, fabsf(v));
                    }
                }
            }
        }
    }
    }
    if constexpr (TAP) {
        // producer tap (TAP: a variant of its own -- the running maximum costs the untapped kernel, which sits at its register limit,
        // 24 bytes of scratch and 10 us per launch if it is merely a run-time branch): this tile's largest stored magnitude joins the
        // wave's running maximum in its LDS slot (behind the staging planes: no register lives across the main loop for it)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax_t = fmaxf(omax_t, __shfl_xor(omax_t, o, 64));
        if (lane == 0) {
            float* const slot = lds + g.tap_lds_off + (threadIdx.x >> 6);      // (formed here: nothing of it is live in the main loop)
            *slot = fmaxf(*slot, omax_t);
        }
    }
#if GX_KQ_ABL   /* measurement build: shader-clock ticks per 100 MHz tick of workgroup 0's main loop, and its length in us */
    if (bx == 0 && by == 0 && tid == 0 && par_a == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        out[0] = (float)((double)(abl_c1 - abl_c0) / (double)(abl_w1 - abl_w0));
        out[1] = (float)(abl_w1 - abl_w0) * 0.01f;
    }
#endif
    if constexpr (STATS) {
        // GroupNorm statistics of the (pre-activation) output without a pass over it: per 8-channel block sums over this
        // workgroup's pixels, transposed through LDS and summed in a fixed order (same scheme as gx_conv.hip's STATS)
        constexpr int NB = 4 * MI;          // 8-channel blocks of this workgroup
        float st_s[NB], st_q[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int p = wave * 64 + nj * 32 + (lane & 31);
            const int c = p & (TW - 1);
            const int r = (p >> g.lTW) & (TH - 1);
            const bool ok = img0 + (p >> (g.lTW + g.lTH)) < g.N && R0 + r < g.Hb && C0 + c < g.Wb;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m = m0 + (mi + mh) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    const float bv = bvec[mi][reg >> 2][reg & 3];
                    const float vx = (ok && m < g.M) ? acc[0][mi][nj][reg] + bv : 0.f;
                    const float vy = (ok && m < g.M) ? acc[NCLS - 1][mi][nj][reg] + bv : 0.f;
                    st_s[mi * 4 + (reg >> 2)] += vx + vy;
                    st_q[mi * 4 + (reg >> 2)] += vx * vx + vy * vy;
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NB; ++i) { lds[i * 256 + tid] = st_s[i]; lds[(NB + i) * 256 + tid] = st_q[i]; }
        __syncthreads();
        const int vi = tid >> 4, sub = tid & 15;      // 16 threads per value; rows [0, NB) sums, [NB, 2 NB) squares
        float v = 0.f;
        if (vi < 2 * NB) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(lds + vi * 256 + sub * 16 + 4 * k);
                v += (r[0] + r[1]) + (r[2] + r[3]);
            }
        }
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        if (sub == 0 && vi < 2 * NB) {
            const int part = (th_i * g.tiles_w + tw_i) * 2 + par_a;
            const int nblk = g.M >> 3;
            const int blk = (m0 >> 3) + mh * 4 + (vi % NB);
            if (blk < nblk && img0 < g.N)
                g.stats[(((size_t)img0 * g.stats_parts + part) * nblk + blk) * 2 + (vi / NB)] = v;
        }
    }
}

template <int MODE, int NQ>
__global__ void __launch_bounds__(256, 2)
kq_kernel(const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
          float* __restrict__ out, QGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (XCD-aware tile map over the whole-tile workgroups: neighbouring tiles, which share halo rows, on one XCD's L2)
    const int bx = (int)blockIdx.x < g.nfull ? gx_xcd_tile(blockIdx.x, g.nfull) : (int)blockIdx.x;
    if (bx < g.nfull) q_body<MODE, NQ, false, 2>(in, wp, bias, out, g, lds, bx, blockIdx.y, 0);
    else q_body<MODE, NQ, false, 1>(in, wp, bias, out, g, lds, g.nfull + ((bx - g.nfull) >> 1), blockIdx.y, 0, (bx - g.nfull) & 1);
}

// both output-row parities of the transposed conv in one launch; blockIdx.y = channel tile, blockIdx.z = 0: rows 2r
// (15 taps, dispatched first: the longer workgroups lead), 1: rows 2r + 1 (10 taps)
template <int NQ, bool STATS>
__global__ void __launch_bounds__(256, 2)
kq_dt_kernel(const float* __restrict__ in, const float* __restrict__ wp0, const float* __restrict__ wp1,
             const float* __restrict__ bias, float* __restrict__ out, QGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (XCD-aware tile map over the whole-tile workgroups: neighbouring tiles, which share halo rows, on one XCD's L2)
    const int bx = (int)blockIdx.x < g.nfull ? gx_xcd_tile(blockIdx.x, g.nfull) : (int)blockIdx.x;
    if (bx < g.nfull) {
        if (blockIdx.z) q_body<Q_DT1, NQ, STATS, 2>(in, wp1, bias, out, g, lds, bx, blockIdx.y, 1);
        else q_body<Q_DT0, NQ, STATS, 2>(in, wp0, bias, out, g, lds, bx, blockIdx.y, 0);
    } else {
        const int tile = g.nfull + ((bx - g.nfull) >> 1), mh = (bx - g.nfull) & 1;
        if (blockIdx.z) q_body<Q_DT1, NQ, STATS, 1>(in, wp1, bias, out, g, lds, tile, blockIdx.y, 1, mh);
        else q_body<Q_DT0, NQ, STATS, 1>(in, wp0, bias, out, g, lds, tile, blockIdx.y, 0, mh);
    }
}

template <int NQ, bool F16 = false>
__global__ void __launch_bounds__(256, 2)
kq_dgh_kernel(const float* __restrict__ in, const float* __restrict__ wp, float* __restrict__ out, QGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (XCD-aware tile map over the whole-tile workgroups: neighbouring tiles, which share halo rows, on one XCD's L2)
    const int bx = (int)blockIdx.x < g.nfull ? gx_xcd_tile(blockIdx.x, g.nfull) : (int)blockIdx.x;
    if (bx < g.nfull) q_body<Q_DGH, NQ, false, 2, F16>(in, wp, nullptr, out, g, lds, bx, blockIdx.y, 0);
    else q_body<Q_DGH, NQ, false, 1, F16>(in, wp, nullptr, out, g, lds, g.nfull + ((bx - g.nfull) >> 1), blockIdx.y, 0, (bx - g.nfull) & 1);
}

// conv3x3 on the bf16 pipe, 32 output channels per workgroup (blockIdx.y)
template <int NQ, bool F16 = false, bool TAP = false>
__global__ void __launch_bounds__(256, 3)
kq_c3h_kernel(const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
              float* __restrict__ out, QGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // TAP (g.o_amax, an armed gx_amax_tap): four floats behind the staging planes hold the waves' running maxima of |stored value|
    float* const tap4 = lds + g.tap_lds_off;
    if constexpr (TAP) {
        if ((threadIdx.x & 63) == 0) tap4[threadIdx.x >> 6] = 0.f;      // (a slot is only touched by its own wave's lane 0 until the end)
    }
    // persistent workgroups (the grid is ~2 per CU): a tile's output stores drain while the next tile's loads are already in
    // flight -- a workgroup that ends after every tile waits for its stores before its LDS / registers are handed on, and with
    // two 16-channel chunks per tile that tail is a large part of a tile's life
    for (int t = blockIdx.x; t < g.nfull; t += gridDim.x) {
        const int tile = gx_xcd_tile(t, g.nfull);
        q_body<Q_C3H, NQ, false, 1, F16, TAP>(in, wp, bias, out, g, lds, tile, blockIdx.y, 0, 0);
        __syncthreads();          // the next tile's staging overwrites LDS the slowest wave may still be reading
    }
    if constexpr (TAP) {          // one partial maximum per workgroup for the tensor's next reader
        if (threadIdx.x == 0)
            g.o_amax[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(tap4[0], tap4[1]), fmaxf(tap4[2], tap4[3]));
    }
}

// 5 x 5 stride-1 conv on the bf16 pipe: 16 x 16-pixel tiles (20 x 20 halo positions: four staging rounds, exact LDS planes)
template <int NQ, bool F16 = false>
__global__ void __launch_bounds__(256, 2)
kq_c5h_kernel(const float* __restrict__ in, const float* __restrict__ wp, float* __restrict__ out, QGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    q_body<Q_C5H, NQ, false, 2, F16>(in, wp, nullptr, out, g, lds, gx_xcd_tile(blockIdx.x, gridDim.x), blockIdx.y, 0);
}

// the same launch shape on the bf16 matrix pipe (Q_DT0H / Q_DT1H)
template <int NQ, bool STATS, bool F16 = false>
__global__ void __launch_bounds__(256, 2)
kq_dth_kernel(const float* __restrict__ in, const float* __restrict__ wp0, const float* __restrict__ wp1,
              const float* __restrict__ bias, float* __restrict__ out, QGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // g.ilv (finding 41): the two row parities INTERLEAVED along blockIdx.x (grid.z = 1) in groups of eight blocks -- one per XCD, so
    // the 15-tap and the 10-tap workgroup of a tile run on the same L2 right after each other, and the workgroups sharing a CU have
    // different lengths: they drift apart instead of storing their 128 KB tiles in lockstep (the store bursts of a grid of equal
    // workgroups are HBM-write-bound windows between compute-only windows)
    int bxr = (int)blockIdx.x, z = (int)blockIdx.z;
    if (g.ilv) {
        const int gx = (int)gridDim.x >> 1, al = gx & ~7, b = (int)blockIdx.x;
        if (b < 2 * al) { z = (b >> 3) & 1; bxr = ((b >> 4) << 3) | (b & 7); }
        else { const int r = b - 2 * al, rem = gx - al; z = r / rem; bxr = al + r - z * rem; }
    }
    // (XCD-aware tile map over the whole-tile workgroups: neighbouring tiles, which share halo rows, on one XCD's L2)
    const int bx = bxr < g.nfull ? gx_xcd_tile(bxr, g.nfull) : bxr;
    if (bx < g.nfull) {
        if (z) q_body<Q_DT1H, NQ, STATS, 2, F16>(in, wp1, bias, out, g, lds, bx, blockIdx.y, 1);
        else q_body<Q_DT0H, NQ, STATS, 2, F16>(in, wp0, bias, out, g, lds, bx, blockIdx.y, 0);
    } else {
        const int tile = g.nfull + ((bx - g.nfull) >> 1), mh = (bx - g.nfull) & 1;
        if (z) q_body<Q_DT1H, NQ, STATS, 1, F16>(in, wp1, bias, out, g, lds, tile, blockIdx.y, 1, mh);
        else q_body<Q_DT0H, NQ, STATS, 1, F16>(in, wp0, bias, out, g, lds, tile, blockIdx.y, 0, mh);
    }
}

int q_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// 256-pixel tile over the Hb x Wb base grid: widest power-of-two rows, then tallest, images fill the rest
bool q_plan(int N, int K, int M, int Hb, int Wb, int Hi, int Wi, int Ho, int Wo, QGeom* g, int* nq, size_t* lds_bytes,
            int maxt) {
    if (K % 8 != 0 || !gx_is_pow2(Hb) || !gx_is_pow2(Wb) || Hb * Wb > 65536) return false;
    int TW = Wb < 64 ? Wb : 64;
    int TH = 256 / TW < Hb ? 256 / TW : Hb;
    int G = 256 / (TW * TH);
    g->N = N; g->K = K; g->M = M; g->nchunks = K / 8;
    g->Hb = Hb; g->Wb = Wb; g->Hi = Hi; g->Wi = Wi; g->Ho = Ho; g->Wo = Wo;
    g->lTH = q_ilog2(TH); g->lTW = q_ilog2(TW); g->lG = q_ilog2(G);
    g->tiles_h = Hb / TH; g->tiles_w = Wb / TW;
    g->ilv = 0; g->x_amax_n = kAmaxParts; g->x_amax = g->w_amax = nullptr; g->act = 0; g->stats = nullptr; g->stats_parts = 0; g->rt_th = g->rt_tw = 0; g->mask = nullptr; g->mask_act = 0; g->o_amax = nullptr; g->tap_lds_off = 0;
    const int CHS = G * (TH + 2) * (TW + 2);
    if (2 * CHS > 4 * 256) return false;
    *nq = 2 * CHS <= 3 * 256 ? 3 : 4;
    if ((double)N * K * Hi * Wi * 4.0 >= 2.0e9) return false;     // 31-bit byte offsets into the input tensor
    const int nw = (maxt * 128 + 255) / 256;
    *lds_bytes = ((size_t)2 * *nq * 1024 + (size_t)2 * nw * 1024) * sizeof(float);
    static const char* pad_env = getenv("GENESIS_KQ_ONE_WG");     // measurement: one workgroup per CU
    if (pad_env && pad_env[0] == '1' && *lds_bytes < 96 * 1024) *lds_bytes = 96 * 1024;
    return *lds_bytes <= 160 * 1024;
}

// A CU works through its workgroups almost one after the other (the older workgroup wins the matrix pipe), so a grid of
// T equal tiles costs ceil(T / 256) rounds: 896 tiles (the 32 -> 64 decoder layer at K*B = 224) = 3.5 -> 4 rounds.
// When the last round is at most half full, its tiles are split along M into two half-work workgroups each (32 output
// channels): 768 whole + 256 half workgroups = 3 + ~0.6 rounds.  Returns the number of whole-tile workgroups and sets
// the grid width.  Only for full 64-channel tiles (M % 64 == 0 or > 32 in the last tile).
int q_split_tail(int ptiles, int M, unsigned* grid_x) {
    static const char* env = getenv("GENESIS_KQ_TAIL");
    const int r = ptiles % 256;
    const bool split = !(env && env[0] == '0') && ptiles > 256 && r > 0 && r <= 128 && (M % 64 == 0 || M % 64 > 32);
    const int nfull = split ? ptiles - r : ptiles;
    *grid_x = (unsigned)(nfull + 2 * (ptiles - nfull));
    return nfull;
}

int g_kq_mode = -1;   // 0 off, 1 auto (layers whose grid fills the chip), 2 every eligible shape

int kq_mode() {
    if (g_kq_mode < 0) {
        const char* env = getenv("GENESIS_KQ");
        g_kq_mode = env ? (env[0] == '0' ? 0 : (env[0] == '2' ? 2 : 1)) : 1;
    }
    return g_kq_mode;
}

bool q_fills(int N, int Hb, int Wb, int M, int mult) {
    const long wgs = (long)gx_ceil_div(N * Hb * Wb, 256) * gx_ceil_div(M, 64) * mult;
    // one workgroup per CU on most of the chip is enough: 224 tiles (the 16 -> 32 decoder layer's data gradient at
    // K*B = 224) measured 127 us against 142 us on the round-1 kernel; 56 tiles (8 -> 16) 121 against 57
    return kq_mode() == 2 || wgs >= 192;
}

template <typename KernelT>
void q_set_attr(KernelT k, bool* done) {
    if (!*done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        *done = true;
    }
}

}  // namespace

// ---- internal API (gx_common.h) ----------------------------------------------------------------------------------
bool gx_kq_c3_eligible(int N, int K, int M, int H, int W) {
    QGeom g; int nq; size_t lds;
    return kq_mode() != 0 && q_plan(N, K, M, H, W, H, W, H, W, &g, &nq, &lds, 9) && q_fills(N, H, W, M, 1);
}
bool gx_kq_deconv_eligible(int N, int K, int M, int Hb, int Wb, int mult) {
    QGeom g; int nq; size_t lds;
    return kq_mode() != 0 && q_plan(N, K, M, Hb, Wb, Hb, Wb, 2 * Hb, 2 * Wb, &g, &nq, &lds, 9) && q_fills(N, Hb, Wb, M, mult);
}

int gx_kq_c3_launch(const float* in, const float* wp, const float* bias, int act, float* out, int N, int K, int M, int H,
                    int W, hipStream_t s) {
    QGeom g; int nq; size_t lds;
    if (!q_plan(N, K, M, H, W, H, W, H, W, &g, &nq, &lds, 9)) { gx_set_error("kq conv3x3: shape not eligible"); return GX_EINVAL; }
    g.act = act;
    dim3 grid(1, gx_ceil_div(M, 64));
    g.nfull = q_split_tail(g.tiles_h * g.tiles_w * gx_ceil_div(N, 1 << g.lG), M, &grid.x);
    {
        GxProf pf(KID_TAPCONV_C3, s, 2.0 * N * (double)M * K * 9 * H * W,
                  4.0 * ((double)N * K * H * W + (double)N * M * H * W + 9.0 * K * M));
        static bool a3 = false, a4 = false;
        if (nq == 3) { q_set_attr(&kq_kernel<Q_C3, 3>, &a3); hipLaunchKernelGGL((kq_kernel<Q_C3, 3>), grid, dim3(256), lds, s, in, wp, bias, out, g); }
        else { q_set_attr(&kq_kernel<Q_C3, 4>, &a4); hipLaunchKernelGGL((kq_kernel<Q_C3, 4>), grid, dim3(256), lds, s, in, wp, bias, out, g); }
    }
    GX_CHECK_LAUNCH("kq conv3x3");
    return GX_OK;
}

// transposed conv forward: in [N,K,Hb,Wb] -> out [N,M,2Hb,2Wb]; stats != NULL: GroupNorm block sums in the epilogue
// (needs one image per tile: *stats_parts = 0 when the tile spans several images and nothing is written)
int gx_kq_deconv_fwd_launch(const float* in, const float* wp0, const float* wp1, const float* bias, float* out, int N,
                            int K, int M, int Hb, int Wb, float* stats, int* stats_parts, hipStream_t s) {
    QGeom g; int nq; size_t lds;
    if (!q_plan(N, K, M, Hb, Wb, Hb, Wb, 2 * Hb, 2 * Wb, &g, &nq, &lds, 5)) { gx_set_error("kq deconv fwd: shape not eligible"); return GX_EINVAL; }
    const bool st = stats && g.lG == 0 && (M % 8) == 0;
    if (stats_parts) *stats_parts = 0;
    if (st) {
        g.stats = stats;
        g.stats_parts = g.tiles_h * g.tiles_w * 2;
        if (stats_parts) *stats_parts = g.stats_parts;
    }
    dim3 grid(1, gx_ceil_div(M, 64), 2);
    g.nfull = q_split_tail(g.tiles_h * g.tiles_w * gx_ceil_div(N, 1 << g.lG), M, &grid.x);
    {
        GxProf pf(KID_TAPCONV_DT0, s, 2.0 * N * (double)M * K * 25 * Hb * Wb,
                  4.0 * ((double)N * K * Hb * Wb + (double)N * M * 4 * Hb * Wb + 25.0 * K * M));
        static bool a[4] = {false, false, false, false};
        if (nq == 3 && st) { q_set_attr(&kq_dt_kernel<3, true>, &a[0]); hipLaunchKernelGGL((kq_dt_kernel<3, true>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
        else if (nq == 3) { q_set_attr(&kq_dt_kernel<3, false>, &a[1]); hipLaunchKernelGGL((kq_dt_kernel<3, false>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
        else if (st) { q_set_attr(&kq_dt_kernel<4, true>, &a[2]); hipLaunchKernelGGL((kq_dt_kernel<4, true>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
        else { q_set_attr(&kq_dt_kernel<4, false>, &a[3]); hipLaunchKernelGGL((kq_dt_kernel<4, false>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
    }
    GX_CHECK_LAUNCH("kq deconv fwd");
    return GX_OK;
}

// ---- the transposed conv forward on the bf16 matrix pipe (packs 22 / 23: weights pre-split into bf16 pieces)
static int g_kq_h = -1;     // 1 (default): eligible layers run there; GENESIS_KQ_BF16X6=0 / gx_kq_precision(0): fp32 pipe
static void kq_h_init() {
    if (g_kq_h >= 0) return;
    const char* e = getenv("GENESIS_KQ_BF16X6");
    const char* f = getenv("GENESIS_KQ_F16X3");         // 0: six bf16 piece products instead of three fp16 ones (finding 40)
    g_kq_h = (e && e[0] == '0') ? 0 : ((f && f[0] == '0') ? 1 : 2);
}
static bool kq_h_on() { kq_h_init(); return g_kq_h != 0; }
bool gx_kq_f16_on() { kq_h_init(); return g_kq_h == 2; }

// ---- largest magnitude of a tensor, for the fp16 x 3 form's power-of-two scale: ONE launch of kAmaxParts workgroups (1024 threads:
//      16 waves per CU keep the HBM busy) writes kAmaxParts partial maxima; the conv kernels reduce them themselves (one 16-byte load
//      per lane and a wave reduction at the top of q_body: q_amax_parts) -- no second launch, no counter, no atomics: the workspace
//      needs no initial state
__global__ void __launch_bounds__(1024)
amax_partial_kernel(const float* __restrict__ x, size_t n, int vec, float* __restrict__ partials) {
    float m = 0.f;
    if (vec) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        const size_t n4 = n >> 2;
        for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) {
            const f32x4 v = x4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) m = fmaxf(m, fabsf(x[i]));
    } else {
        for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) m = fmaxf(m, fabsf(x[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) r = fmaxf(r, red[i]);
        partials[blockIdx.x] = r;
    }
}
int gx_kq_amax_launch(const float* x, size_t n, float* ws, hipStream_t s) {
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * (double)n);
        hipLaunchKernelGGL(amax_partial_kernel, dim3(kAmaxParts), dim3(1024), 0, s, x, n, (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? 1 : 0, ws);
    }
    GX_CHECK_LAUNCH("kq amax");
    return GX_OK;
}
__global__ void __launch_bounds__(1024)
weight_amax_kernel(const float* __restrict__ w, int n, float* __restrict__ out) {
    const float r = gx_wg1024_amax(w, n);
    if (threadIdx.x == 0) out[0] = r;
}
int gx_kq_weight_amax_launch(const float* w, int n, float* out, hipStream_t s) {
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * n);
        hipLaunchKernelGGL(weight_amax_kernel, dim3(1), dim3(1024), 0, s, w, n, out);
    }
    GX_CHECK_LAUNCH("kq weight amax");
    return GX_OK;
}
// a producer that left THOUSANDS of partial maxima (one per workgroup of a gated unit's or a generic GroupNorm kernel's grid): one
// small launch folds them into one value -- every workgroup of the conv reducing them all costs more (measured: kq_c5h 147 -> 160 us
// per launch at 10 240 partials, kq_dgh 873 -> 911 us at 11 264) than this launch does (~4 us)
static const int kFoldPartsAbove = [] { const char* e = getenv("GENESIS_KQ_FOLD_ABOVE"); return e ? atoi(e) : 1024; }();
__global__ void __launch_bounds__(1024)
amax_fold2_kernel(const float* __restrict__ p0, int n0, const float* __restrict__ p1, int n1, float* __restrict__ out) {
    __shared__ float red[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n0; i += 1024) m = fmaxf(m, p0[i]);
    for (int i = threadIdx.x; i < n1; i += 1024) m = fmaxf(m, p1[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) r = fmaxf(r, red[i]);
        out[0] = r;
    }
}
// the armed gx_conv_input_amax hint (one tensor, or the two halves of a concat buffer) as ONE partial-maxima array for a kernel of this
// file: a single short array as it is, anything else folded into amax_ws[0]
static int kq_hint_parts(const float** parts, int* n, float* amax_ws, hipStream_t s) {
    const float *p0, *p1; int n0, n1;
    *parts = nullptr; *n = 0;
    if (!amax_ws || !gx_conv_input_hint(&p0, &n0, &p1, &n1)) return GX_OK;
    if (!p1 && n0 <= kFoldPartsAbove) { *parts = p0; *n = n0; return GX_OK; }
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * (n0 + n1));
        hipLaunchKernelGGL(amax_fold2_kernel, dim3(1), dim3(1024), 0, s, p0, n0, p1, p1 ? n1 : 0, amax_ws);
    }
    GX_CHECK_LAUNCH("kq amax fold (hint)");
    *parts = amax_ws; *n = 1;
    return GX_OK;
}
static int kq_fold_parts(const float** parts, int* n, float* amax_ws, hipStream_t s) {
    if (!*parts || *n <= kFoldPartsAbove || !amax_ws) return GX_OK;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * *n);
        hipLaunchKernelGGL(weight_amax_kernel, dim3(1), dim3(1024), 0, s, *parts, *n, amax_ws);
    }
    GX_CHECK_LAUNCH("kq amax fold");
    *parts = amax_ws; *n = 1;
    return GX_OK;
}
size_t gx_kq_deconv_h_pack_bytes(int K, int M, int nt) {     // one row parity's packed weights (+ slack: whole-phase copies)
    return (size_t)gx_ceil_div(M, 64) * (K / 16) * nt * QH_TAP_BYTES + 16384;
}
// LDS of the bf16-pipe transposed-conv kernels: three input piece planes (exact at four staging rounds) + two weight buffers;
// 0: the tile does not leave room for two workgroups per CU
// (f16: the launch's allocation in the fp16 x 3 form -- two planes, two-piece weight taps: 48 KB at three staging rounds, i.e. THREE
//  workgroups per CU where the registers allow it (kq_dgh_kernel<3, true>: 156); eligibility is always decided on the three-piece size)
static size_t qh_lds(const QGeom& g, int nq, bool f16 = false) {
    const int np = f16 ? 2 : 3;
    const int NWH = (3 * (np * 2048 / 16) + 255) / 256;
    const int CHS = (1 << g.lG) * ((1 << g.lTH) + 2) * ((1 << g.lTW) + 2);
    const size_t planes = nq == 4 ? (size_t)np * 2 * CHS * 16 : (size_t)np * nq * 256 * 16;
    const size_t lds = planes + (size_t)2 * NWH * 256 * 16;
    static const char* env = getenv("GENESIS_KQ_H_NQ4");           // 0: shapes with four staging rounds stay on the fp32 pipe
    if (nq == 4 && env && env[0] == '0') return 0;
    return lds <= 80 * 1024 ? lds : 0;
}
bool gx_kq_deconv_h_eligible(int N, int K, int M, int Hb, int Wb) {
    if (!kq_h_on() || K % 16 != 0 || !gx_kq_deconv_eligible(N, K, M, Hb, Wb, 2)) return false;
    QGeom g; int nq; size_t lds;
    return q_plan(N, K, M, Hb, Wb, Hb, Wb, 2 * Hb, 2 * Wb, &g, &nq, &lds, 5) && qh_lds(g, nq) > 0;
}
int gx_kq_deconv_fwd_h_launch(const float* in, const float* wp0, const float* wp1, const float* bias, float* out, int N,
                              int K, int M, int Hb, int Wb, float* stats, int* stats_parts, hipStream_t s, float* amax_ws,
                              const float* w_amax, const float* x_parts, int x_nparts) {
    QGeom g; int nq; size_t lds;
    if (!q_plan(N, K, M, Hb, Wb, Hb, Wb, 2 * Hb, 2 * Wb, &g, &nq, &lds, 5) || qh_lds(g, nq) == 0 || K % 16 != 0) {
        gx_set_error("kq deconv fwd (bf16 pipe): shape not eligible"); return GX_EINVAL;
    }
    lds = qh_lds(g, nq, amax_ws != nullptr);
    const bool st = stats && g.lG == 0 && (M % 8) == 0;
    if (stats_parts) *stats_parts = 0;
    if (st) {
        g.stats = stats;
        g.stats_parts = g.tiles_h * g.tiles_w * 2;
        if (stats_parts) *stats_parts = g.stats_parts;
    }
    if (amax_ws && !x_parts) { const int rc = gx_kq_amax_launch(in, (size_t)N * K * Hb * Wb, amax_ws, s); if (rc) return rc; }
    else if (amax_ws) { const int rc = kq_fold_parts(&x_parts, &x_nparts, amax_ws, s); if (rc) return rc; }
    dim3 grid(1, gx_ceil_div(M, 64), 2);
    g.nfull = q_split_tail(g.tiles_h * g.tiles_w * gx_ceil_div(N, 1 << g.lG), M, &grid.x);
    static const char* ilv_env = getenv("GENESIS_KQ_DTH_INTERLEAVE");       // 0: all 15-tap workgroups, then all 10-tap ones (grid.z)
    if (!(ilv_env && ilv_env[0] == '0') && grid.y == 1) { g.ilv = 1; grid.x *= 2; grid.z = 1; }
    {
        GxProf pf(KID_KQ_DTH, s, 2.0 * N * (double)M * K * 25 * Hb * Wb,
                  4.0 * ((double)N * K * Hb * Wb + (double)N * M * 4 * Hb * Wb + 25.0 * K * M));
        static bool a[8] = {false, false, false, false, false, false, false, false};
        if (amax_ws) {          // fp16 x 3: the input's amax (one small launch ahead of this one, outside its profiling record)
            g.x_amax = x_parts ? x_parts : amax_ws;
            if (x_parts) g.x_amax_n = x_nparts;
            g.w_amax = w_amax;
            if (nq == 3 && st) { q_set_attr(&kq_dth_kernel<3, true, true>, &a[4]); hipLaunchKernelGGL((kq_dth_kernel<3, true, true>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
            else if (nq == 3) { q_set_attr(&kq_dth_kernel<3, false, true>, &a[5]); hipLaunchKernelGGL((kq_dth_kernel<3, false, true>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
            else if (st) { q_set_attr(&kq_dth_kernel<4, true, true>, &a[6]); hipLaunchKernelGGL((kq_dth_kernel<4, true, true>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
            else { q_set_attr(&kq_dth_kernel<4, false, true>, &a[7]); hipLaunchKernelGGL((kq_dth_kernel<4, false, true>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
        } else
        if (nq == 3 && st) { q_set_attr(&kq_dth_kernel<3, true>, &a[0]); hipLaunchKernelGGL((kq_dth_kernel<3, true>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
        else if (nq == 3) { q_set_attr(&kq_dth_kernel<3, false>, &a[1]); hipLaunchKernelGGL((kq_dth_kernel<3, false>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
        else if (st) { q_set_attr(&kq_dth_kernel<4, true>, &a[2]); hipLaunchKernelGGL((kq_dth_kernel<4, true>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
        else { q_set_attr(&kq_dth_kernel<4, false>, &a[3]); hipLaunchKernelGGL((kq_dth_kernel<4, false>), grid, dim3(256), lds, s, in, wp0, wp1, bias, out, g); }
    }
    GX_CHECK_LAUNCH("kq deconv fwd (bf16 pipe)");
    return GX_OK;
}

// ---- conv3x3 of <= 32-output-channel layers on the bf16 pipe (pack kinds 20 / 21); grids need not be powers of two: the
//      256-pixel tile takes the largest power-of-two divisors of the width and height, images fill the rest
static bool q_plan_c3h(int N, int K, int M, int H, int W, QGeom* g, int* nq, size_t* lds_bytes) {
    if (K % 16 != 0 || H * W > 65536 || (double)N * K * H * W * 4.0 >= 2.0e9) return false;
    int TW = 1; while (TW < 64 && W % (2 * TW) == 0) TW *= 2;
    int TH = 1; while (TH * TW < 256 && H % (2 * TH) == 0) TH *= 2;
    int G = 256 / (TW * TH);
    g->rt_th = g->rt_tw = 0;
    static const char* rt_env = getenv("GENESIS_KQ_C3H_ROWTILES");
    if (!(rt_env && rt_env[0] == '0') && TW < W && TW < 32 && W <= 128 && 256 / W >= 2) {
        // a width without a large power-of-two factor (the 72 x 72 canvas: 8): tiles of whole rows instead
        g->rt_tw = W; g->rt_th = 256 / W;
        TW = W; TH = g->rt_th; G = 1;
    }
    g->N = N; g->K = K; g->M = M; g->nchunks = K / 8;
    g->Hb = H; g->Wb = W; g->Hi = H; g->Wi = W; g->Ho = H; g->Wo = W;
    g->lTH = q_ilog2(TH); g->lTW = q_ilog2(TW); g->lG = q_ilog2(G);
    g->tiles_h = gx_ceil_div(H, TH); g->tiles_w = W / TW;
    g->ilv = 0; g->x_amax_n = kAmaxParts; g->x_amax = g->w_amax = nullptr; g->act = 0; g->stats = nullptr; g->stats_parts = 0; g->nfull = 0; g->mask = nullptr; g->mask_act = 0; g->o_amax = nullptr; g->tap_lds_off = 0;
    const int CHS = G * (TH + 2) * (TW + 2);
    if (2 * CHS > 4 * 256) return false;
    *nq = 2 * CHS <= 3 * 256 ? 3 : 4;
    constexpr int NWH = (3 * (QHLay<Q_C3H>::TAPB / 16) + 255) / 256;
    *lds_bytes = (size_t)3 * *nq * 256 * 16 + (size_t)NWH * 256 * 16;          // three input piece planes + one weight buffer
    return true;
}
bool gx_kq_c3h_eligible(int N, int K, int M, int H, int W) {
    static const char* env = getenv("GENESIS_KQ_C3H");
    if ((env && env[0] == '0') || !kq_h_on() || kq_mode() == 0 || M > 32) return false;
    QGeom g; int nq; size_t lds;
    if (!q_plan_c3h(N, K, M, H, W, &g, &nq, &lds)) return false;
    return kq_mode() == 2 || (long)g.tiles_h * g.tiles_w * gx_ceil_div(N, 1 << g.lG) * gx_ceil_div(M, 32) >= 512;
}
int gx_kq_c3h_launch(const float* in, const float* wp, const float* bias, int act, float* out, int N, int K, int M, int H,
                     int W, hipStream_t s, const float* mask, int mask_act, float* amax_ws, const float* w_amax) {
    QGeom g; int nq; size_t lds;
    const float* x_parts = nullptr; int x_nparts = 0;
    if (!q_plan_c3h(N, K, M, H, W, &g, &nq, &lds)) { gx_set_error("kq conv3x3 (bf16 pipe): shape not eligible"); return GX_EINVAL; }
    g.act = act; g.mask = mask; g.mask_act = mask_act;
    if (amax_ws) {
        constexpr int NWF = (3 * (QHLay<Q_C3H, true>::TAPB / 16) + 255) / 256;
        lds = (size_t)2 * nq * 256 * 16 + (size_t)NWF * 256 * 16;          // two input piece planes + one two-piece weight buffer
        // the input's partial maxima handed in by the caller (gx_conv_input_amax: the norm kernel that wrote it), else a pass of our own
        int rc = kq_hint_parts(&x_parts, &x_nparts, amax_ws, s); if (rc) return rc;
        if (!x_parts) { rc = gx_kq_amax_launch(in, (size_t)N * K * H * W, amax_ws, s); if (rc) return rc; }
    }
    dim3 grid(g.tiles_h * g.tiles_w * gx_ceil_div(N, 1 << g.lG), gx_ceil_div(M, 32));
    g.nfull = (int)grid.x;                               // tiles; the workgroups loop over them (kq_c3h_kernel)
    static const char* pers_env = getenv("GENESIS_KQ_C3H_PERSIST");
    const int per_cu = pers_env ? atoi(pers_env) : 3;
    if (per_cu > 0 && (int)grid.x > 256 * per_cu / (int)grid.y) grid.x = 256 * per_cu / grid.y;
    // an armed gx_amax_tap: this launch writes every element of `out` -- one partial maximum per workgroup for the tensor's next
    // reader (the next layer of a BroadcastDecoder chain, forward or backward: no amax pass of its own over a 148 MB canvas)
    g.o_amax = amax_ws ? gx_amax_producer_out(out, false, grid.x * grid.y, (size_t)N * M * H * W) : nullptr;      // (the fp16 form only)
    {
        GxProf pf(KID_KQ_C3H, s, 2.0 * N * (double)M * K * 9 * H * W,
                  4.0 * ((double)N * K * H * W + (double)N * M * H * W + 9.0 * K * M));
        static bool a3 = false, a4 = false, f3 = false, f4 = false;
        if (amax_ws) {          // fp16 x 3 (packs 40 / 41)
            g.x_amax = x_parts ? x_parts : amax_ws; g.w_amax = w_amax;
            if (x_parts) g.x_amax_n = x_nparts;
            static bool t3 = false, t4 = false;
            if (g.o_amax) {         // the producer-tap variant
                g.tap_lds_off = (int)(lds / sizeof(float));
                if (nq == 3) { q_set_attr(&kq_c3h_kernel<3, true, true>, &t3); hipLaunchKernelGGL((kq_c3h_kernel<3, true, true>), grid, dim3(256), lds + 16, s, in, wp, bias, out, g); }
                else { q_set_attr(&kq_c3h_kernel<4, true, true>, &t4); hipLaunchKernelGGL((kq_c3h_kernel<4, true, true>), grid, dim3(256), lds + 16, s, in, wp, bias, out, g); }
            } else
            if (nq == 3) { q_set_attr(&kq_c3h_kernel<3, true>, &f3); hipLaunchKernelGGL((kq_c3h_kernel<3, true>), grid, dim3(256), lds, s, in, wp, bias, out, g); }
            else { q_set_attr(&kq_c3h_kernel<4, true>, &f4); hipLaunchKernelGGL((kq_c3h_kernel<4, true>), grid, dim3(256), lds, s, in, wp, bias, out, g); }
        } else
        if (nq == 3) { q_set_attr(&kq_c3h_kernel<3>, &a3); hipLaunchKernelGGL((kq_c3h_kernel<3>), grid, dim3(256), lds, s, in, wp, bias, out, g); }
        else { q_set_attr(&kq_c3h_kernel<4>, &a4); hipLaunchKernelGGL((kq_c3h_kernel<4>), grid, dim3(256), lds, s, in, wp, bias, out, g); }
    }
    GX_CHECK_LAUNCH("kq conv3x3 (bf16 pipe)");
    return GX_OK;
}

// ---- 5 x 5 stride-1 conv on the bf16 pipe (pack kinds 27 / 28)
static bool q_plan_c5h(int N, int K, int M, int H, int W, QGeom* g, size_t* lds_bytes) {
    if (K % 16 != 0 || H % 16 != 0 || W % 16 != 0 || H * W > 65536 || (double)N * K * H * W * 4.0 >= 2.0e9) return false;
    g->N = N; g->K = K; g->M = M; g->nchunks = K / 8;
    g->Hb = H; g->Wb = W; g->Hi = H; g->Wi = W; g->Ho = H; g->Wo = W;
    g->lTH = 4; g->lTW = 4; g->lG = 0;
    g->tiles_h = H / 16; g->tiles_w = W / 16;
    g->rt_th = g->rt_tw = 0; g->ilv = 0; g->x_amax_n = kAmaxParts; g->x_amax = g->w_amax = nullptr; g->act = 0; g->stats = nullptr; g->stats_parts = 0; g->mask = nullptr; g->mask_act = 0; g->o_amax = nullptr; g->tap_lds_off = 0;
    g->nfull = g->tiles_h * g->tiles_w * N;
    constexpr int NWH = (3 * (QH_TAP_BYTES / 16) + 255) / 256;
    *lds_bytes = (size_t)3 * 2 * (20 * 20) * 16 + (size_t)2 * NWH * 256 * 16;       // exact input planes + two weight buffers
    return *lds_bytes <= 80 * 1024;
}
bool gx_kq_c5h_eligible(int N, int K, int M, int H, int W) {
    static const char* env = getenv("GENESIS_KQ_C5H");
    if ((env && env[0] == '0') || !kq_h_on() || kq_mode() == 0) return false;
    QGeom g; size_t lds;
    if (!q_plan_c5h(N, K, M, H, W, &g, &lds)) return false;
    static const int min_wgs = [] { const char* e = getenv("GENESIS_KQ_C5H_MIN_WGS"); return e ? atoi(e) : 192; }();
    return kq_mode() == 2 || (long)g.nfull * gx_ceil_div(M, 64) >= min_wgs;
}
int gx_kq_c5h_launch(const float* in, const float* wp, float* out, int N, int K, int M, int H, int W, hipStream_t s,
                     float* amax_ws, const float* w_amax, const float* x_parts, int x_nparts) {
    QGeom g; size_t lds;
    if (!q_plan_c5h(N, K, M, H, W, &g, &lds)) { gx_set_error("kq conv5x5 (bf16 pipe): shape not eligible"); return GX_EINVAL; }
    if (amax_ws) {
        constexpr int NWF = (3 * (QHLay<Q_C5H, true>::TAPB / 16) + 255) / 256;
        lds = (size_t)2 * 2 * (20 * 20) * 16 + (size_t)2 * NWF * 256 * 16;
        // (x_parts: the input's partial maxima from the kernel that wrote it -- no pass of our own)
        if (!x_parts) { const int rc = gx_kq_amax_launch(in, (size_t)N * K * H * W, amax_ws, s); if (rc) return rc; }
        else { const int rc = kq_fold_parts(&x_parts, &x_nparts, amax_ws, s); if (rc) return rc; }
    }
    dim3 grid(g.nfull, gx_ceil_div(M, 64));
    {
        GxProf pf(KID_KQ_C5H, s, 2.0 * N * (double)M * K * 25 * H * W, 4.0 * ((double)N * K * H * W + (double)N * M * H * W + 25.0 * K * M));
        static bool a4 = false, f4 = false;
        if (amax_ws) {          // fp16 x 3 (packs 47 / 48)
            g.x_amax = x_parts ? x_parts : amax_ws; g.w_amax = w_amax;
            if (x_parts) g.x_amax_n = x_nparts;
            q_set_attr(&kq_c5h_kernel<4, true>, &f4);
            hipLaunchKernelGGL((kq_c5h_kernel<4, true>), grid, dim3(256), lds, s, in, wp, out, g);
        } else {
        q_set_attr(&kq_c5h_kernel<4>, &a4);
        hipLaunchKernelGGL((kq_c5h_kernel<4>), grid, dim3(256), lds, s, in, wp, out, g);
        }
    }
    GX_CHECK_LAUNCH("kq conv5x5 (bf16 pipe)");
    return GX_OK;
}

bool gx_kq_deconv_dgrad_h_eligible(int N, int K, int M, int Hb, int Wb) {
    if (!kq_h_on() || K % 16 != 0 || !gx_kq_deconv_eligible(N, K, M, Hb, Wb, 1)) return false;
    QGeom g; int nq; size_t lds;
    return q_plan(N, K, M, Hb, Wb, 2 * Hb, 2 * Wb, Hb, Wb, &g, &nq, &lds, 9) && qh_lds(g, nq) > 0;
}
int gx_kq_deconv_dgrad_h_launch(const float* dy, const float* wp, float* dx, int N, int K, int M, int Hb, int Wb,
                                hipStream_t s, float* amax_ws, const float* w_amax, const float* x_parts, int x_nparts) {
    QGeom g; int nq; size_t lds;
    if (!q_plan(N, K, M, Hb, Wb, 2 * Hb, 2 * Wb, Hb, Wb, &g, &nq, &lds, 9) || qh_lds(g, nq) == 0 || K % 16 != 0) {
        gx_set_error("kq deconv dgrad (bf16 pipe): shape not eligible"); return GX_EINVAL;
    }
    lds = qh_lds(g, nq, amax_ws != nullptr);
    if (amax_ws && !x_parts) { const int rc = gx_kq_amax_launch(dy, (size_t)N * K * 4 * Hb * Wb, amax_ws, s); if (rc) return rc; }
    else if (amax_ws) { const int rc = kq_fold_parts(&x_parts, &x_nparts, amax_ws, s); if (rc) return rc; }
    dim3 grid(1, gx_ceil_div(M, 64));
    g.nfull = q_split_tail(g.tiles_h * g.tiles_w * gx_ceil_div(N, 1 << g.lG), M, &grid.x);
    {
        GxProf pf(KID_KQ_DGH, s, 2.0 * N * (double)M * K * 25 * Hb * Wb,
                  4.0 * ((double)N * K * 4 * Hb * Wb + (double)N * M * Hb * Wb + 25.0 * K * M));
        static bool a3 = false, a4 = false, f3 = false, f4 = false;
        if (amax_ws) {
            g.x_amax = x_parts ? x_parts : amax_ws;
            if (x_parts) g.x_amax_n = x_nparts;
            g.w_amax = w_amax;
            if (nq == 3) { q_set_attr(&kq_dgh_kernel<3, true>, &f3); hipLaunchKernelGGL((kq_dgh_kernel<3, true>), grid, dim3(256), lds, s, dy, wp, dx, g); }
            else { q_set_attr(&kq_dgh_kernel<4, true>, &f4); hipLaunchKernelGGL((kq_dgh_kernel<4, true>), grid, dim3(256), lds, s, dy, wp, dx, g); }
        } else
        if (nq == 3) { q_set_attr(&kq_dgh_kernel<3>, &a3); hipLaunchKernelGGL((kq_dgh_kernel<3>), grid, dim3(256), lds, s, dy, wp, dx, g); }
        else { q_set_attr(&kq_dgh_kernel<4>, &a4); hipLaunchKernelGGL((kq_dgh_kernel<4>), grid, dim3(256), lds, s, dy, wp, dx, g); }
    }
    GX_CHECK_LAUNCH("kq deconv dgrad (bf16 pipe)");
    return GX_OK;
}

// transposed conv data gradient: dy [N,K,2Hb,2Wb] -> dx [N,M,Hb,Wb]
int gx_kq_deconv_dgrad_launch(const float* dy, const float* wp, float* dx, int N, int K, int M, int Hb, int Wb,
                              hipStream_t s) {
    QGeom g; int nq; size_t lds;
    if (!q_plan(N, K, M, Hb, Wb, 2 * Hb, 2 * Wb, Hb, Wb, &g, &nq, &lds, 9)) { gx_set_error("kq deconv dgrad: shape not eligible"); return GX_EINVAL; }
    dim3 grid(1, gx_ceil_div(M, 64));
    g.nfull = q_split_tail(g.tiles_h * g.tiles_w * gx_ceil_div(N, 1 << g.lG), M, &grid.x);
    {
        GxProf pf(KID_TAPCONV_DG, s, 2.0 * N * (double)M * K * 25 * Hb * Wb,
                  4.0 * ((double)N * K * 4 * Hb * Wb + (double)N * M * Hb * Wb + 25.0 * K * M));
        static bool a3 = false, a4 = false;
        if (nq == 3) { q_set_attr(&kq_kernel<Q_DG, 3>, &a3); hipLaunchKernelGGL((kq_kernel<Q_DG, 3>), grid, dim3(256), lds, s, dy, wp, nullptr, dx, g); }
        else { q_set_attr(&kq_kernel<Q_DG, 4>, &a4); hipLaunchKernelGGL((kq_kernel<Q_DG, 4>), grid, dim3(256), lds, s, dy, wp, nullptr, dx, g); }
    }
    GX_CHECK_LAUNCH("kq deconv dgrad");
    return GX_OK;
}

extern "C" int gx_kq_precision(int mode) {
    GX_CHECK_ARG(mode >= -1 && mode <= 2, "gx_kq_precision: mode must be 0 (fp32 matrix pipe), 1 (bf16 pipe, six piece products), 2 (as 1, the transposed convs from three fp16 piece products) or -1 (the environment's default)");
    g_kq_h = mode;        // (-1: kq_h_init reads GENESIS_KQ_BF16X6 / GENESIS_KQ_F16X3 again)
    return GX_OK;
}

extern "C" int gx_kq_policy(int mode) {
    GX_CHECK_ARG(mode >= 0 && mode <= 2, "gx_kq_policy: mode must be 0 (off), 1 (chip-filling layers) or 2 (every eligible shape)");
    g_kq_mode = mode;
    return GX_OK;
}
