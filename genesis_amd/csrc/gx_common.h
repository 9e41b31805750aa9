// Shared internals of libgenesis_hip.so (not part of the public C ABI; see include/genesis_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/genesis_hip.h"

// ---- error plumbing: C ABI never throws; negative return + thread-local message ----
void gx_set_error(const char* fmt, ...);

#define GX_CHECK_ARG(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            gx_set_error(__VA_ARGS__);           \
            return GX_EINVAL;                    \
        }                                        \
    } while (0)

#define GX_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            gx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return GX_ELAUNCH;                                                  \
        }                                                                       \
    } while (0)

// ---- per-kernel live profiling (HIP events on the launch stream; off unless gx_profile_enable(1)) ----
enum GxKernelId {
    KID_TAPCONV_C3 = 0, KID_TAPCONV_DT0, KID_TAPCONV_DT1, KID_TAPCONV_DG, KID_PACK_WEIGHTS,
    KID_WGRAD_C3, KID_WGRAD_D00, KID_WGRAD_D01, KID_WGRAD_D10, KID_WGRAD_D11, KID_WGRAD_REDUCE,
    KID_GN_FWD, KID_GN_BWD, KID_GN_PARAM_REDUCE, KID_ICSBP_FWD, KID_ICSBP_BWD, KID_MASKPOOL_FWD,
    KID_MASKPOOL_BWD, KID_MIXTURE_FWD, KID_MIXTURE_BWD, KID_CONV1X1_FWD, KID_CONV1X1_DGRAD,
    KID_CONV1X1_WGRAD, KID_SMALL_REDUCE, KID_ADAM, KID_GECO, KID_SPLITK_REDUCE, KID_BIAS_ACT_BWD, KID_DCONV, KID_GATED, KID_LATENT, KID_DENSE, KID_WINO,
    KID_WGQ_STREAM, KID_KQ_DTH, KID_KQ_DGH, KID_KQ_C3H, KID_KQ_C5H, KID_COUNT
};
// ---- contexts: every piece of mutable library state that outlives a call (deferred-reduction queues, queued
// weight-gradient jobs, the packed-weight cache a step is recording / served from, the per-kernel profiling records)
// belongs to a context.  A thread works in its CURRENT context (thread-local; context 0 until gx_ctx_make_current), so
// two training loops -- in one thread one after the other, or in two threads -- never see each other's queues.
// What remains process-wide is one-time initialisation (zero pages, kernel attributes) and the dispatch policies.
constexpr int kGxMaxCtx = 32;
int gx_cur_ctx(void);
struct GxCtxFlags { bool prof_on, defer_on; int cache_recording, cache_active; };
GxCtxFlags& gx_ctx_flags(void);      // of the calling thread's current context
#define g_gx_prof_on (gx_ctx_flags().prof_on)
#define g_gx_defer_on (gx_ctx_flags().defer_on)
void gx_prof_begin(int kid, hipStream_t s, double flops, double bytes);
void gx_prof_end(hipStream_t s);
// RAII: brackets ONE kernel launch with two events when profiling is on; a flag test otherwise.
struct GxProf {
    hipStream_t s; bool on;
    GxProf(int kid, hipStream_t s_, double flops, double bytes) : s(s_), on(g_gx_prof_on) {
        if (on) gx_prof_begin(kid, s, flops, bytes);
    }
    ~GxProf() { if (on) gx_prof_end(s); }
};

// ---- deferred parameter-gradient reductions (gx_defer_*; gx_api.cpp owns the queue) ----
// ca0 / cb0 / CAf / CBf: the record covers the channel block (ca0.., cb0..) of a CAf x CBf weight (CAf == 0: the whole
// weight, CA x CB) -- the stream-K weight-gradient kernel keeps one slab region per 64 x 64 channel block
struct GxWgradRed { const float* partial; float* dw; int nsplit, Ttot, CA, CB, CApad, CBpad, layout, ns0, ns1, ns2, ns3,
                    ca0, cb0, CAf, CBf; };
struct GxGnRed { const float* part; float* dgamma; float* dbeta; float* dbias; int N, C; };
bool gx_defer_push_wgrad(const GxWgradRed& r);   // false: queue full (caller reduces immediately)
int gx_defer_wgrad_room();                      // free slots of the weight-gradient reduce queue of the current context
bool gx_defer_push_gn(const GxGnRed& r);
int gx_defer_flush_wgrad(const GxWgradRed* items, int n, hipStream_t s);   // gx_conv.hip
int gx_defer_flush_gn(const GxGnRed* items, int n, hipStream_t s);         // gx_norm.hip

static inline int gx_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int gx_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int gx_round_up(int a, int b) { return gx_ceil_div(a, b) * b; }

// XCD-aware block -> tile map (cdna_hip_programming.md T1).  The dispatcher is observed to place block b on XCD b % 8, each XCD with
// its own 4 MB L2: with tile = block id, spatially adjacent tiles (which share their halo rows / columns) sit on eight different
// L2s and every halo is fetched from HBM once per tile.  This bijection hands every XCD a CONTIGUOUS run of tiles (the blocks of an
// XCD are dispatched in increasing id, so neighbours in the run are neighbours in time).  Speed only: any placement is correct.
#ifdef __HIPCC__
__device__ __forceinline__ int gx_xcd_tile(int bid, int nwg) {
#ifdef GX_NO_XCD_SWIZZLE
    return bid;
#else
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
#endif
}
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 64-lane wavefront reductions (CDNA4: wave = 64).
__device__ __forceinline__ float gx_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double gx_wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float gx_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// ---- Winograd F(2x2,3x3) weight operand (gx_wino.hip; also produced by the packed-weight cache of gx_conv.hip) ----
// (G g G^T)[xi][nu], p = 4 xi + nu, for output channel m / reduction channel k.  mode 0: g = w[m][k] (forward);
// mode 1: g = w[k][m] rotated by 180 degrees (data gradient).  w is [Co][Ci][3][3].
__device__ __forceinline__ float gx_wino_u_value(const float* __restrict__ w, int mode, int Co, int Ci, int m, int k,
                                                 int p) {
    const int Mact = mode == 0 ? Co : Ci, Kact = mode == 0 ? Ci : Co;
    if (m >= Mact || k >= Kact) return 0.f;
    const int xi = p >> 2, nu = p & 3;
    float t[3];   // row xi of G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float g0, g1, g2;
        if (mode == 0) {
            const float* q = w + ((size_t)m * Ci + k) * 9 + b;
            g0 = q[0]; g1 = q[3]; g2 = q[6];
        } else {
            const float* q = w + ((size_t)k * Ci + m) * 9 + (2 - b);
            g0 = q[6]; g1 = q[3]; g2 = q[0];
        }
        t[b] = xi == 0 ? g0 : (xi == 1 ? 0.5f * (g0 + g1 + g2) : (xi == 2 ? 0.5f * (g0 - g1 + g2) : g2));
    }
    return nu == 0 ? t[0] : (nu == 1 ? 0.5f * (t[0] + t[1] + t[2]) : (nu == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2]));
}
// position of U(p, k, m) in the conv kernel's operand order:
// [m tile][chunk of 8 k][position][lane = 32 ((k >> 2) & 1) + (m & 31)][2 (k & 3) + ((m >> 5) & 1)]
// (lane half h of the chunk's MFMA j multiplies channel 4 h + j: the B operand's four values are one 16-byte LDS read)
__host__ __device__ __forceinline__ size_t gx_wino_u_slot(int m, int k, int p, int Kpad) {
    const size_t base = ((size_t)(m >> 6) * (Kpad >> 3) + (k >> 3)) * 16 + p;
    return base * 512 + (size_t)((((((k >> 2) & 1) << 5) | (m & 31)) << 3) | ((k & 3) << 1) | ((m >> 5) & 1));
}
// The same operands for the bf16 matrix pipe (gx_wino.hip: wino_conv_h_kernel): every U value as three bf16 pieces, chunks of 16
// reduction channels.  32-bit word (two channels k even, k + 1) of piece `piece` of position p:
// [m tile 64][chunk k >> 4][position 16][piece 3][m half (m >> 5) & 1][lane = 32 ((k >> 3) & 1) + (m & 31)][(k & 7) >> 1]
// -- a wave's A operand of (position, piece, m half) is 1 KB, lane-linear: one 16-byte load per lane.
__host__ __device__ __forceinline__ size_t gx_wino_h_word(int m, int k, int p, int piece, int Kpad16) {
    return ((((((size_t)(m >> 6) * (Kpad16 >> 4) + (k >> 4)) * 16 + p) * 3 + piece) * 2 + ((m >> 5) & 1)) * 256) +
           (size_t)((((k >> 3) & 1) * 32 + (m & 31)) * 4 + ((k & 7) >> 1));
}
__host__ __device__ __forceinline__ size_t gx_wino_h_bytes(int Kpad16, int Mpad) {
    return (size_t)(Mpad >> 6) * (Kpad16 >> 4) * 16 * 6144;
}
bool gx_wino_h_on();     // Winograd layers on the bf16 pipe (default; gx_wino_precision(0) / GENESIS_WINO_BF16X6=0: fp32 pipe)
bool gx_conv_input_hint(const float** p0, int* n0, const float** p1, int* n1);      // the armed gx_conv_input_amax hint (gx_wino.hip), not cleared
bool gx_wino_f16_pending(void);   // ... and the NEXT launch of this thread on two fp16 pieces per operand (gx_conv_input_amax armed): pack kinds 45 / 46
// conv with an already packed U (16 * Kpad * Mpad floats; bf16 pipe: gx_wino_h_bytes): out[N,M,H,W] from in[N,K,H,W]
bool gx_wino_eligible(int N, int K, int M, int H, int W);
int gx_wino_launch(const float* in, const float* U, float* out, int N, int K, int M, int H, int W, hipStream_t s);


// ---- k-quad tap convolutions (gx_kq.hip): 16-byte k-contiguous MFMA operand reads for the chip-filling layers ----
// packed-weight layout [m tile 64][chunk of 8 k][tap, phase-major][quad (k >> 2) & 1][m & 63][k & 3]
// tap order: the [t] order of gx_conv.hip's packs, except the transposed conv's data gradient (25 taps, t = kh*5+kw),
// whose taps are grouped by parity plane (kh & 1, kw & 1): planes of 9, 6, 6, 4 taps, (kh / 2, kw / 2) row-major inside
__host__ __device__ __forceinline__ int gx_kq_dg_tap_slot(int t) {
    const int kh = t / 5, kw = t % 5;
    const int pa = kh & 1, pb = kw & 1;
    const int p = pa * 2 + pb;
    const int base = p == 0 ? 0 : (p == 1 ? 9 : (p == 2 ? 15 : 21));
    return base + (kh >> 1) * (3 - pb) + (kw >> 1);
}
__host__ __device__ __forceinline__ size_t gx_kq_w_slot(int m, int k, int tslot, int NT, int Kpad) {
    return ((((size_t)(m >> 6) * (Kpad >> 3) + (k >> 3)) * NT + tslot) * 2 + ((k >> 2) & 1)) * 256 + (m & 63) * 4 + (k & 3);
}
bool gx_kq_c3_eligible(int N, int K, int M, int H, int W);
bool gx_kq_deconv_eligible(int N, int K, int M, int Hb, int Wb, int mult);   // mult: workgroups per (pixel, channel) tile
int gx_kq_c3_launch(const float* in, const float* wp, const float* bias, int act, float* out, int N, int K, int M, int H,
                    int W, hipStream_t s);
int gx_kq_deconv_fwd_launch(const float* in, const float* wp0, const float* wp1, const float* bias, float* out, int N,
                            int K, int M, int Hb, int Wb, float* stats, int* stats_parts, hipStream_t s);
int gx_kq_deconv_dgrad_launch(const float* dy, const float* wp, float* dx, int N, int K, int M, int Hb, int Wb,
                              hipStream_t s);
// ... on the bf16 matrix pipe: weights packed by kinds 22 / 23 = [channel tile][16-channel chunk][tap][piece hi|mid|lo]
// [octet][64 output channels][8 channels] in bf16 (two per 32-bit word)
bool gx_kq_deconv_h_eligible(int N, int K, int M, int Hb, int Wb);
size_t gx_kq_deconv_h_pack_bytes(int K, int M, int nt);
// ... or from THREE fp16 piece products (gx_kq_precision(2), DESIGN.md finding 40): x * 2^e = hi + lo (11 + 11 significant bits),
// hi*hi + hi*lo + lo*hi; e per TENSOR from its largest magnitude so that the pieces sit inside fp16's exponent range.  Pack kinds
// 40 / 41 / 42 / 43 / 44 / 47 / 48 = 20 / 21 / 22 / 23 / 24 / 27 / 28 as two fp16 pieces of w * 2^e (two piece slots per tap); the weight tensor's amax lives in the last
// 64 bytes of the packing's slack (written by the amax launch that precedes the pack, read by the pack and by the conv kernels).
bool gx_kq_f16_on();
__host__ __device__ __forceinline__ size_t gx_kq_h_amax_off(int K, int M, int nt) {      // byte offset of that float
    return (size_t)((M + 63) / 64) * (K / 16) * nt * 6144 + 16384 - 64;
}
__host__ __device__ __forceinline__ int gx_f16_scale_exp(float amax) {     // amax * 2^e in [2^14, 2^15); 0 / inf / nan: e = 0
    if (!(amax > 0.f) || amax > 3.0e38f) return 0;
    int ex;
    (void)frexpf(amax, &ex);
    return 15 - ex;
}
// amax of a tensor (n floats) as kAmaxParts partial maxima in ws (gx_kq_amax_ws_floats() floats, 16-byte aligned): one launch; the
// conv kernels reduce the partials themselves
constexpr int kAmaxParts = 256;
__host__ __device__ __forceinline__ constexpr size_t gx_kq_amax_ws_floats() { return kAmaxParts; }
int gx_kq_amax_launch(const float* x, size_t n, float* ws, hipStream_t s);
// producer -> consumer hand-over of those partial maxima (gx_kq_amax_link, include/genesis_hip.h): armed by the caller with a scratch
// buffer; the next producer that can (the decoder head's GroupNorm backward) writes one partial maximum per workgroup of the tensor it
// stores -- only a launch that covers the WHOLE tensor (numel elements: a chunked producer does not qualify) -- and records the tensor's
// address; the next fp16 x 3 conv whose INPUT is that address (and size) reads them instead of making its own pass.
// One-shot, per thread; any fp16 x 3 conv call clears it.
struct GxAmaxLink { float* parts; int capacity; size_t numel; const float* tensor; int n; int hits; };      // numel: the WHOLE tensor's
GxAmaxLink& gx_amax_link(void);          // (gx_api.cpp)
// TAP (round 6): the same partial maxima for ANY later reader -- the weight-gradient stream-K launch at the end of the backward
// pass needs the scale of BOTH operands of every layer (gx_wgq_operand_amax).  gx_amax_tap(parts, capacity, numel) arms a one-shot,
// per-thread request: the next amax-capable producer launch (whatever its destination views: channel slices of concat buffers
// and resampled second copies hold the same values) writes one partial maximum of the values it stores per workgroup and
// records how many; gx_amax_tap_result() returns that count (0: the launch could not serve) and disarms.
struct GxAmaxTap { float* parts; int capacity; int n; size_t numel; };      // numel: the elements the producer launch must cover
GxAmaxTap& gx_amax_tap_state(void);      // (gx_api.cpp)
// producer side, both mechanisms: where the launch of `nwg` workgroups writes its partial maxima (NULL: nowhere).  link_ok: the
// destination qualifies for the armed link (a whole plain tensor of L.numel elements).
inline float* gx_amax_producer_out(const float* tensor, bool link_ok, unsigned nwg, size_t covered) {
    GxAmaxTap& T = gx_amax_tap_state();
    GxAmaxLink& L = gx_amax_link();
    float* p = nullptr;
    // (a launch that covers only part of the tensor -- the chunked callers -- must not serve: its maxima are not the tensor's)
    if (T.parts && (int)nwg <= T.capacity && covered == T.numel) { p = T.parts; T.n = (int)nwg; T.parts = nullptr; }
    if (link_ok && L.parts && !L.tensor && tensor && (p || (int)nwg <= L.capacity)) {
        if (p) L.parts = p; else p = L.parts;      // (one set of partials serves both readers)
        L.tensor = tensor; L.n = (int)nwg;
    }
    return p;
}
// consumer side: partials of `x` if the armed link holds them (then *n > 0), clearing the link either way
inline const float* gx_amax_link_take(const float* x, size_t numel, int* n) {
    GxAmaxLink& L = gx_amax_link();
    const float* p = nullptr;
    *n = 0;
    if (L.parts && x && L.tensor == x && L.numel == numel && L.n > 0) { p = L.parts; *n = L.n; ++L.hits; }
    L.parts = nullptr; L.capacity = 0; L.numel = 0; L.tensor = nullptr; L.n = 0;
    return p;
}
#ifdef __HIPCC__
// |.|-maximum of n floats by ONE workgroup of 1024 threads (a weight tensor: ~100 k floats; 16-byte loads, eight in flight per thread);
// the result is valid in thread 0
__device__ __forceinline__ float gx_wg1024_amax(const float* __restrict__ w, int n) {
    typedef float gx_amax_f4 __attribute__((ext_vector_type(4)));
    float m = 0.f;
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        const gx_amax_f4* w4 = reinterpret_cast<const gx_amax_f4*>(w);
        const int n4 = n >> 2;
#pragma unroll 8
        for (int i = threadIdx.x; i < n4; i += 1024) {
            const gx_amax_f4 v = w4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float gx_amax_red[16];
    if ((threadIdx.x & 63) == 0) gx_amax_red[threadIdx.x >> 6] = m;
    __syncthreads();
    float r = gx_amax_red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) r = fmaxf(r, gx_amax_red[i]);
    return r;
}
// producer-side tap (gx_amax_producer_out): the workgroup's largest stored magnitude -> parts[idx]; am >= 0, every thread calls
__device__ __forceinline__ void gx_block_amax_store(float am, float* __restrict__ parts, unsigned idx) {
    __shared__ float gx_bam_red[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0) gx_bam_red[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
        for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) r = fmaxf(r, gx_bam_red[i]);
        parts[idx] = r;
    }
}
#endif
int gx_kq_weight_amax_launch(const float* w, int n, float* out, hipStream_t s);          // one workgroup (weights are small)
bool gx_kq_deconv_dgrad_h_eligible(int N, int K, int M, int Hb, int Wb);      // pack 24: all 25 taps, plane-major slots
int gx_kq_deconv_dgrad_h_launch(const float* dy, const float* wp, float* dx, int N, int K, int M, int Hb, int Wb,
                                hipStream_t s, float* amax_ws = nullptr, const float* w_amax = nullptr,
                                const float* x_parts = nullptr, int x_nparts = 0);     // x_parts: the input's partial maxima from its producer
int gx_kq_deconv_fwd_h_launch(const float* in, const float* wp0, const float* wp1, const float* bias, float* out, int N,
                              int K, int M, int Hb, int Wb, float* stats, int* stats_parts, hipStream_t s,
                              float* amax_ws = nullptr, const float* w_amax = nullptr,    // amax_ws != NULL: the fp16 x 3 form (packs 42 / 43)
                              const float* x_parts = nullptr, int x_nparts = 0);
// 32-bit word of element (m, k even, k + 1) of piece `piece` of tap t
// ... kinds 20 / 21 (conv3x3 forward / data gradient of <= 32-output-channel layers): 32 output channels per channel tile
// (NP: pieces per value -- 3 bf16 ones, or 2 fp16 ones in the packs 40 - 48 of the fp16 x 3 form)
__host__ __device__ __forceinline__ size_t gx_kq_h32_word(int m, int k, int t, int piece, int NT, int K, int NP = 3) {
    return ((((size_t)(m >> 5) * (K >> 4) + (k >> 4)) * NT + t) * NP + piece) * 256 + (((k >> 3) & 1) * 32 + (m & 31)) * 4 + ((k & 7) >> 1);
}
bool gx_kq_c3h_eligible(int N, int K, int M, int H, int W);
int gx_kq_c3h_launch(const float* in, const float* wp, const float* bias, int act, float* out, int N, int K, int M, int H,
                     int W, hipStream_t s, const float* mask = nullptr, int mask_act = 0, float* amax_ws = nullptr,
                     const float* w_amax = nullptr);         // amax_ws != NULL: the fp16 x 3 form (packs 40 / 41)
// 5 x 5 stride-1 conv on the bf16 pipe (pack kinds 27 / 28 = the bf16-piece forms of 7 / 8; K a multiple of 16, H and W of 16)
bool gx_kq_c5h_eligible(int N, int K, int M, int H, int W);
int gx_chan_sums_launch(const float* x, int N, int C, int HW, float* part, float* out, hipStream_t s);   // gx_misc.hip
int gx_kq_c5h_launch(const float* in, const float* wp, float* out, int N, int K, int M, int H, int W, hipStream_t s,
                     float* amax_ws = nullptr, const float* w_amax = nullptr, const float* x_parts = nullptr,
                     int x_nparts = 0);      // (packs 47 / 48; x_parts: the input's partial maxima, gx_kq_amax_link)
__host__ __device__ __forceinline__ size_t gx_kq_h_word(int m, int k, int t, int piece, int NT, int K, int NP = 3) {
    return ((((size_t)(m >> 6) * (K >> 4) + (k >> 4)) * NT + t) * NP + piece) * 512 + (((k >> 3) & 1) * 64 + (m & 63)) * 4 + ((k & 7) >> 1);
}

// ---- second-generation weight gradients (gx_wgq.hip): LDS-DMA staging of both operands, grouped launches -------------
bool gx_wgq_c3_eligible(int N, int Cin, int Cout, int H, int W);
bool gx_wgq_deconv_eligible(int N, int Cin, int Cout, int Hb, int Wb);
int gx_wgq_max_split(int CA, int CB);
// ws: room for ws_slabs split-K slabs of Ttot * CApad * CBpad floats.  With deferral on (gx_defer_enable) the job is only
// queued: gx_defer_flush launches all queued jobs in groups and then the batched slab reduce.
int gx_wgq_c3(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, float* ws, int ws_slabs,
              hipStream_t s);
int gx_wgq_deconv(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int Hb, int Wb, float* ws,
                  int ws_slabs, hipStream_t s);
bool gx_wgq_c5_eligible(int N, int CA, int CB, int H, int W);
int gx_wgq_c5(const float* a, const float* b, float* dw, int N, int CA, int CB, int H, int W, float* ws, int ws_slabs,
              hipStream_t s);
// gx_wstrip.hip: the 32 -> 32 conv3x3 weight gradient as column strips (grids that are no power of two, W % 8 == 0)
bool gx_wstrip_supported(int N, int C, int H, int W);
size_t gx_wstrip_ws_floats(int N, int C, int H, int W);
int gx_wstrip_launch(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int H, int W, void* ws, hipStream_t s);
bool gx_wgq_bf16_pipe(void);      // gx_wgq_precision / GENESIS_WGQ_BF16X6: weight gradients on the bf16 matrix pipe (six piece products)
int gx_wgq_pending(void);
void gx_wgq_discard(void);
int gx_wgq_flush(hipStream_t s);
// gx_conv.hip: conv3x3 weight gradients of layers too small for the stream-K launch (4 x 4 grids), queued while deferral is on
int gx_wf_flush(hipStream_t s);
int gx_wf_pending(void);
void gx_wf_discard(void);
int gx_wgrad_reduce_now(const GxWgradRed& r, hipStream_t s, int accumulate = 0);    // gx_conv.hip: dw (+)= sum of the slabs
