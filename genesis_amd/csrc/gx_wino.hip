// 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2, 3x3) on the fp32 matrix cores (modules/blocks.py:159-165,
// modules/unet.py:33-57 -- the UNet's and the heads' conv layers at 32x32 and 64x64; forward and data gradient).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 4x4 input patch d -> 2x2 outputs: 16 multiplies instead of 36,
// i.e. 16 independent GEMMs  M_p[cout][tile] = sum_cin U_p[cout][cin] V_p[cin][tile]  (p = 4 xi + nu), 2.25x fewer
// MFMA passes than the direct tap loop of gx_conv.hip.  fp32 throughout; the transforms only add / subtract (B, A) and
// scale by 1/2, 1/4 (G), so the result differs from the direct sum by ordinary fp32 rounding (measured ~1e-6 relative).
//
// Workgroup = 64 output channels x (8 rows x 16 cols of output = 4 x 8 = 32 Winograd tiles) of one image, 4 waves;
// wave xi owns the four positions (xi, 0..3): 4 positions x 64 channels x 32 tiles = 128 accumulator registers.
// Per chunk of 8 input channels: the raw 10x18 halo patch is staged global -> registers -> LDS (even / odd columns in
// separate planes so that the transform's stride-2 reads are conflict-free); 256 threads = 8 channels x 32 tiles each
// transform their 4x4 patch (B^T d B: 32 adds) into V in LDS; 32 MFMA 32x32x2 per wave.  Raw patch and V are double
// buffered and the loop is software-pipelined with ONE barrier per chunk: while the matrix pipe works through chunk c,
// the same waves transform chunk c + 1 and park the patch of chunk c + 2.  The weight operands are NOT staged in LDS:
// a wave only ever needs its own four positions, so U is packed per (channel tile, chunk, position, lane) and each
// lane loads its 8 values per position with two 16-byte loads straight into the MFMA operand registers (refilled for
// the next chunk as soon as a half has issued).
// Measured (B=32): 64->64 @64x64 63 us (direct tap loop 99 us), 128->64 107 us (183 us); the MFMA instructions alone
// (no staging, no transform, only the LDS operand reads) take 51 / 81 us, i.e. ~105 TF/s on the pipe, where a pure
// register-operand MFMA loop sustains 155 TF/s (gx_mfma_fp32_probe).
// Epilogue: each wave reduces its row of positions over nu (A on the right), the four waves' rows are combined through
// LDS (A^T on the left) and written as float2 pairs.
#include <stdlib.h>

#include "gx_common.h"

namespace {

constexpr int WKC = 8;                 // input channels per chunk
constexpr int WTH = 4, WTW = 8;        // Winograd tiles per workgroup (rows x cols) -> 8 x 16 output pixels
constexpr int WNT = WTH * WTW;         // 32 tiles = one MFMA N block
constexpr int PR = 2 * WTH + 2, PC = 2 * WTW + 2;   // raw patch 10 x 18
constexpr int PLANE = 10;              // floats per (row, column parity) plane row: 9 used; 4 * PLANE = 8 (mod 32)
constexpr int PPITCH = PR * 2 * PLANE; // floats per channel of the raw patch (200)
constexpr int RAW_FLOATS = WKC * PPITCH;            // 1600 per buffer, two buffers
constexpr int V_FLOATS = 16 * WKC * WNT;            // 4096 per buffer, two buffers
constexpr int RAW_PER_THREAD = (WKC * PR * PC + 255) / 256;   // 6

struct WinoGeom {
    int N, K, M;          // images, reduction channels, output channels
    int K1, M1;           // pair variants: reduction channels [0, K1) come from `in`, the rest from `in2` (K1 % 8 == 0);
                          // output channels [0, M1) go to `out`, the rest to `out2` (M1 % 64 == 0).  Single: K1 = Kpad, M1 = Mpad
    int Kpad, Mpad;       // packed weight dims (Kpad % 8 == 0, Mpad % 64 == 0)
    int H, W;             // H % 8 == 0, W % 16 == 0
    int tiles_h, tiles_w; // H / 8, W / 16
    // the fp16-piece form (wino_conv_h_kernel<NB, true>): partial maxima of the input tensor(s), as their producers left them
    // (gx_conv_input_amax): xam0[0 .. xn0) and xam1[0 .. xn1)
    const float* xam0; const float* xam1;
    int xn0, xn1;
};

// U in the conv kernel's operand order (gx_common.h: gx_wino_u_value / gx_wino_u_slot)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int mode, int Co, int Ci, int Kpad,
                                 int Mpad) {
    const int total = 16 * Kpad * Mpad;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int m = idx % Mpad, k = (idx / Mpad) % Kpad, p = idx / (Mpad * Kpad);
        U[gx_wino_u_slot(m, k, p, Kpad)] = gx_wino_u_value(w, mode, Co, Ci, m, k, p);
    }
}

// Two 3x3 layers that read the SAME input (the seg_head and feat_head[0] convs on the encoder features,
// models/genesisv2_config.py:70-73): packed as ONE layer.  Uf: forward, output channels [0, Co1) from w1, the rest from
// w2.  Ud: data gradient, reduction channels [0, Co1) (dy of layer 1) from w1, the rest from w2 -- the sum of the two
// layers' input gradients then happens inside the accumulators.  One launch packs both.
__global__ void wino_pack_pair_kernel(const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ Uf,
                                      float* __restrict__ Ud, int Co1, int Co2, int Ci, int KpadF, int MpadF, int KpadD,
                                      int MpadD) {
    const int totF = 16 * KpadF * MpadF, totD = 16 * KpadD * MpadD;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < totF + totD; idx += gridDim.x * blockDim.x) {
        if (idx < totF) {
            const int m = idx % MpadF, k = (idx / MpadF) % KpadF, p = idx / (MpadF * KpadF);
            Uf[gx_wino_u_slot(m, k, p, KpadF)] = m < Co1 ? gx_wino_u_value(w1, 0, Co1, Ci, m, k, p)
                                                         : gx_wino_u_value(w2, 0, Co2, Ci, m - Co1, k, p);
        } else {
            const int i = idx - totF;
            const int m = i % MpadD, k = (i / MpadD) % KpadD, p = i / (MpadD * KpadD);
            Ud[gx_wino_u_slot(m, k, p, KpadD)] = k < Co1 ? gx_wino_u_value(w1, 1, Co1, Ci, m, k, p)
                                                         : gx_wino_u_value(w2, 1, Co2, Ci, m, k - Co1, p);
        }
    }
}

__global__ void __launch_bounds__(256, 2)
wino_conv_kernel(const float* __restrict__ in, const float* __restrict__ in2, const float* __restrict__ U,
                 float* __restrict__ out, float* __restrict__ out2, const WinoGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* raw = lds;                       // [2][WKC][PR][2][PLANE]
    float* V = lds + 2 * RAW_FLOATS;        // [2][16][WKC][WNT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // XCD-aware (tile, channel tile) map over the whole grid: neighbouring tiles (shared halos) AND the channel tiles of one tile (the
    // same patches) are consecutive blocks of one XCD, i.e. one L2
    const int xcd_l = gx_xcd_tile(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int my = xcd_l % (int)gridDim.y;
    int tile = xcd_l / (int)gridDim.y;
    const int tw_i = tile % g.tiles_w; tile /= g.tiles_w;
    const int th_i = tile % g.tiles_h; tile /= g.tiles_h;
    const int n = tile;
    const int R0 = th_i * (2 * WTH), C0 = tw_i * (2 * WTW);
    const int m0 = my * 64;
    const int HW = g.H * g.W;
    // (pair data gradient: two input tensors, each with its own channel count and per-image block)
    const int Ka = g.K1 < g.K ? g.K1 : g.K, Kb = g.K - Ka;
    const float* in_n = in + (size_t)n * Ka * HW;
    const float* in2_n = in2 + (size_t)n * Kb * HW;
    const int nchunks = g.Kpad / WKC;

    // raw-patch staging slots of this thread: element e = tid + 256 q of [WKC][PR][PC].  The loads are raw buffer loads
    // over this image's [K][H*W] block: voff[q] = byte offset of (channel of the chunk, pixel), or 1 GiB for a halo pixel
    // outside the image / an unused slot; the chunk's first channel rides in the scalar offset.  Out-of-range addresses
    // (padding pixels, channels >= K) read as 0 -- no compare / select per element.
    int voff[RAW_PER_THREAD], loff[RAW_PER_THREAD];       // loff: LDS float offset inside a raw buffer; an unused slot stores
                                                          // its zero into the spare 10th float of a plane row (never read)
#pragma unroll
    for (int q = 0; q < RAW_PER_THREAD; ++q) {
        const int e = tid + 256 * q;
        voff[q] = 0x40000000; loff[q] = PLANE - 1;
        if (e < WKC * PR * PC) {
            const int ch = e / (PR * PC), rem = e - ch * (PR * PC);
            const int pr = rem / PC, pc = rem - pr * PC;
            const int r = R0 - 1 + pr, c = C0 - 1 + pc;
            loff[q] = ch * PPITCH + (pr * 2 + (pc & 1)) * PLANE + (pc >> 1);
            if (r >= 0 && r < g.H && c >= 0 && c < g.W) voff[q] = (ch * HW + r * g.W + c) * 4;
        }
    }
    // descriptor of the chunk that starts at channel k0: based AT that channel and as long as the channels that exist
    // from there, so that the hardware's range check (per-lane offset against the descriptor's length; a scalar offset
    // would not be checked) zeroes exactly the channels >= K of a ragged last chunk and everything of a chunk past the end
    // (empty descriptor: no memory is touched).  Scalar selects, no branch.
    auto chunk_rsrc = [&](int k0, int& soff) {
        const bool second = k0 >= g.K1;
        const int kr = second ? k0 - g.K1 : k0, left = (second ? Kb : Ka) - kr;
        soff = 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((second ? in2_n : in_n) + (size_t)kr * HW), 0,
                                                 left > 0 ? left * HW * 4 : 0, 0x00020000);
    };
    float rawr[RAW_PER_THREAD];
    auto load_raw = [&](int k0) {
        int soff;
        const __amdgpu_buffer_rsrc_t rs = chunk_rsrc(k0, soff);
#pragma unroll
        for (int q = 0; q < RAW_PER_THREAD; ++q)
            rawr[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[q], soff, 0));
    };
    auto store_raw = [&](float* rbuf) {
#pragma unroll
        for (int q = 0; q < RAW_PER_THREAD; ++q)
            rbuf[loff[q]] = rawr[q];
    };
    // this wave's weight operands: [m tile][chunk][position 4 wave + nu][lane][8]
    const float* Uw = U + (((size_t)my * nchunks) * 16 + 4 * wave) * 512 + lane * 8;

    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc[4][2];     // [nu][mi]: one 32x32 accumulator tile each
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[a][b][c] = 0.f;

    // transform role: channel tk, tile (tty, ttx); patch element (i, j) sits at row 2 tty + i, column 2 ttx + j.
    // A wave takes all 8 channels of one tile row: its 64 V values of a position are two runs of 32 consecutive floats
    // (the minimum of two LDS cycles per 64-lane store) and its patch reads fall on banks 8 tk + ttx (two lanes per bank).
    const int tk = lane & 7, ttr = wave * 8 + (lane >> 3);
    const int tty = ttr >> 3, ttx = ttr & 7;
    const int tt = tid & 31;              // (the epilogue's tile index)
    const int tsrc_off = tk * PPITCH + (2 * tty) * 2 * PLANE + ttx;
    // V [position][k quad = channel >> 2][tile][channel & 3]: the four channels a lane half feeds to four successive
    // MFMAs are 16 contiguous bytes, one ds_read_b128 (4-byte operand reads cap the pipe at ~106 TF/s)
    const int tdst_off = (tk >> 2) * (WNT * 4) + ttr * 4 + (tk & 3);
    // V = B^T d B for this thread's (channel, tile): column j of the patch is plane (j & 1), index ttx + (j >> 1)
    auto transform = [&](const float* rbuf, float* vbuf) {
        const float* tsrc = rbuf + tsrc_off;
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = tsrc[(i * 2 + (j & 1)) * PLANE + (j >> 1)];
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j];
            t[1][j] = d[1][j] + d[2][j];
            t[2][j] = d[2][j] - d[1][j];
            t[3][j] = d[1][j] - d[3][j];
        }
        float* tdst = vbuf + tdst_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tdst[(4 * i + 0) * (WKC * WNT)] = t[i][0] - t[i][2];
            tdst[(4 * i + 1) * (WKC * WNT)] = t[i][1] + t[i][2];
            tdst[(4 * i + 2) * (WKC * WNT)] = t[i][2] - t[i][1];
            tdst[(4 * i + 3) * (WKC * WNT)] = t[i][1] - t[i][3];
        }
    };

    const int bn = lane & 31, kh = lane >> 5;

    // ---- software pipeline, ONE barrier per chunk.  In iteration c a wave issues the MFMAs of chunk c (V[c & 1]) and,
    // between them, transforms chunk c + 1 (raw[(c + 1) & 1] -> V[(c + 1) & 1]) and parks the patch of chunk c + 2 in
    // raw[c & 1]; the matrix pipe works through the MFMAs while the wave issues the transform's LDS / VALU work.
    f32x4 ua[4][2];       // [nu][half]: (kk, mi) = (2 half, 0), (2 half, 1), (2 half + 1, 0), (2 half + 1, 1)
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
        ua[nu][0] = *reinterpret_cast<const f32x4*>(Uw + (size_t)nu * 512);
        ua[nu][1] = *reinterpret_cast<const f32x4*>(Uw + (size_t)nu * 512 + 4);
    }
    {
        // the first three chunks' patches in ONE round trip (a chunk that does not exist reads through an empty
        // descriptor: zeros, no branches)
        float r0[RAW_PER_THREAD], r1[RAW_PER_THREAD];
        int so0, so1;
        const __amdgpu_buffer_rsrc_t rs0 = chunk_rsrc(0, so0), rs1 = chunk_rsrc(WKC, so1);
#pragma unroll
        for (int q = 0; q < RAW_PER_THREAD; ++q) {
            r0[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs0, voff[q], so0, 0));
            r1[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, voff[q], so1, 0));
        }
        load_raw(2 * WKC);
#pragma unroll
        for (int q = 0; q < RAW_PER_THREAD; ++q)
        { raw[loff[q]] = r0[q]; raw[RAW_FLOATS + loff[q]] = r1[q]; }
    }
    __syncthreads();
    transform(raw, V);
    __syncthreads();
#ifndef GX_WINO_ABL
#define GX_WINO_ABL 0          // measurement builds (tools/abl_build.sh): 1 no transform, 2 no patch staging, 4 no MFMAs
#endif
    // One chunk.  H1 / H2 / H3: chunk c + 1 / c + 2 / c + 3 exists.  In the steady part of the loop they are compile-time
    // `true`, so the body is ONE basic block (with the runtime tests the body fell into five blocks, the transform into
    // one without a single MFMA: the matrix pipe then depends on the partner workgroup of the CU to stay busy); the last
    // three chunks of a tile take the same body with the runtime tests.
    // The 32 MFMAs go out in eight groups of four; the next chunk's transform is cut into pieces (patch reads | column
    // pass | one output row each) that are pinned between the groups by scheduling fences, so that every piece runs in
    // the shadow of the four MFMAs (256 pipe cycles) issued just before it.
#define GX_WINO_FENCE __builtin_amdgcn_sched_barrier(0);
#define GX_WINO_CHUNK(c, H1, H2, H3)                                                                                  \
    {                                                                                                                 \
        const bool tr = (H1) && !(GX_WINO_ABL & 1);                                                                   \
        const float* tsrc = raw + (((c) + 1) & 1) * RAW_FLOATS + tsrc_off;                                            \
        float* tdst = V + (((c) + 1) & 1) * V_FLOATS + tdst_off;                                                      \
        float td[4][4], tc[4][4];                                                                                     \
        if (!(GX_WINO_ABL & 2) && (H2)) {                                                                             \
            store_raw(raw + ((c) & 1) * RAW_FLOATS);      /* chunk c + 2 (its buffer was consumed in iteration c - 1) */ \
            if (H3) load_raw(((c) + 3) * WKC);                                                                        \
        }                                                                                                             \
        float blast[4];       /* the B values of the last two groups (bq is refilled for the next chunk before them) */ \
        GX_WINO_FENCE                                                                                                 \
        _Pragma("unroll") for (int grp = 0; grp < 8; ++grp) {                                                         \
            const int half = grp >> 2, kk2 = (grp >> 1) & 1, kk = 2 * half + kk2;   /* MFMA kk: channel 4 kh + kk */   \
            _Pragma("unroll") for (int nu = 2 * (grp & 1); nu < 2 * (grp & 1) + 2; ++nu) {                            \
                const float b = grp < 6 ? bq[nu][kk] : blast[nu];                                                     \
                if (GX_WINO_ABL & 4) { acc[nu][0][kk] += b * ua[nu][half][kk2 * 2]; continue; }                       \
                acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[nu][half][kk2 * 2], b, acc[nu][0], 0, 0, 0);     \
                acc[nu][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[nu][half][kk2 * 2 + 1], b, acc[nu][1], 0, 0, 0); \
            }                                                                                                         \
            GX_WINO_FENCE                                                                                             \
            if (tr && grp == 0) {                                                                                     \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) td[i][j] = tsrc[(i * 2 + (j & 1)) * PLANE + (j >> 1)]; \
            }                                                                                                         \
            if (tr && grp == 1) {                                                                                     \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                       \
                    tc[0][j] = td[0][j] - td[2][j];                                                                   \
                    tc[1][j] = td[1][j] + td[2][j];                                                                   \
                    tc[2][j] = td[2][j] - td[1][j];                                                                   \
                    tc[3][j] = td[1][j] - td[3][j];                                                                   \
                }                                                                                                     \
            }                                                                                                         \
            if (tr && grp >= 2 && grp < 6) {                                                                          \
                const int i = grp - 2;                                                                                \
                tdst[(4 * i + 0) * (WKC * WNT)] = tc[i][0] - tc[i][2];                                                \
                tdst[(4 * i + 1) * (WKC * WNT)] = tc[i][1] + tc[i][2];                                                \
                tdst[(4 * i + 2) * (WKC * WNT)] = tc[i][2] - tc[i][1];                                                \
                tdst[(4 * i + 3) * (WKC * WNT)] = tc[i][1] - tc[i][3];                                                \
            }                                                                                                         \
            if ((H1) && !(GX_WINO_ABL & 8) && (grp & 3) == 3)                                                         \
                _Pragma("unroll") for (int nu = 0; nu < 4; ++nu)                                                      \
                    ua[nu][half] = *reinterpret_cast<const f32x4*>(Uw + ((size_t)((c) + 1) * 16 + nu) * 512 + 4 * half); \
            if (grp == 5) {       /* the chunk's only barrier sits HERE: V of chunk c + 1 is complete, and its B values */ \
                _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) blast[nu] = bq[nu][3];   /* are fetched in the shadow */  \
                __syncthreads();                                                          /* of the last eight MFMAs  */  \
                if (H1) {                                                                                             \
                    const float* Vn = V + (((c) + 1) & 1) * V_FLOATS;                                                 \
                    _Pragma("unroll") for (int nu = 0; nu < 4; ++nu)                                                  \
                        bq[nu] = *reinterpret_cast<const f32x4*>(Vn + (((4 * wave + nu) * 2 + kh) * WNT + bn) * 4);   \
                }                                                                                                     \
            }                                                                                                         \
            GX_WINO_FENCE                                                                                             \
        }                                                                                                             \
    }
    f32x4 bq[4];              // this lane's B values of the current chunk: [nu][channel & 3]
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) bq[nu] = *reinterpret_cast<const f32x4*>(V + (((4 * wave + nu) * 2 + kh) * WNT + bn) * 4);
    int c = 0;
    for (; c + 3 < nchunks; ++c) GX_WINO_CHUNK(c, true, true, true)
    for (; c < nchunks; ++c) GX_WINO_CHUNK(c, c + 1 < nchunks, c + 2 < nchunks, c + 3 < nchunks)
#undef GX_WINO_FENCE
#undef GX_WINO_CHUNK

    // ---- output transform.  This wave holds M[xi][nu] (xi = wave): right-multiply by A -> two columns
    //      q0 = M0 + M1 + M2, q1 = M1 - M2 - M3, exchanged through LDS; then Y[0][b] = q(xi=0) + q(1) + q(2),
    //      Y[1][b] = q(1) - q(2) - q(3).  One 32-channel half (mi) at a time: E[xi][b][32 m][32 n] = 32 KB.
    float* E = lds;
    // (pair forward: this channel tile belongs to `out` or to `out2`)
    const bool second_out = m0 >= g.M1;
    const int Mo = second_out ? g.M - g.M1 : (g.M1 < g.M ? g.M1 : g.M), mbase = second_out ? g.M1 : 0;
    float* const outp = second_out ? out2 : out;
    const size_t out_n = (size_t)n * Mo * HW;
    if (GX_WINO_ABL & 16) {
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 16; ++c) sacc += acc[a][b][c];
        outp[out_n + (size_t)(m0 - mbase + (tid >> 2)) * HW + (size_t)R0 * g.W + C0 + (tid & 3)] = sacc;
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        __syncthreads();
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = e + 8 * r4 + 4 * kh;      // C/D layout: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                const int rg = 4 * r4 + e;
                const float q0 = acc[0][mi][rg] + acc[1][mi][rg] + acc[2][mi][rg];
                const float q1 = acc[1][mi][rg] - acc[2][mi][rg] - acc[3][mi][rg];
                E[((wave * 2 + 0) * 32 + row) * 32 + bn] = q0;
                E[((wave * 2 + 1) * 32 + row) * 32 + bn] = q1;
            }
        __syncthreads();
        // 1024 (m, tile) pairs, 4 per thread: thread -> tile = tid & 31, m = (tid >> 5) + 8 j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ml = (tid >> 5) + 8 * j;
            const int m = m0 + mi * 32 + ml;
            if (m >= g.M) continue;
            const int ty = tt >> 3, tx = tt & 7;
            float q[4][2];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                q[xi][0] = E[((xi * 2 + 0) * 32 + ml) * 32 + tt];
                q[xi][1] = E[((xi * 2 + 1) * 32 + ml) * 32 + tt];
            }
            float2 y0, y1;
            y0.x = q[0][0] + q[1][0] + q[2][0];
            y0.y = q[0][1] + q[1][1] + q[2][1];
            y1.x = q[1][0] - q[2][0] - q[3][0];
            y1.y = q[1][1] - q[2][1] - q[3][1];
            float* o = outp + out_n + (size_t)(m - mbase) * HW + (size_t)(R0 + 2 * ty) * g.W + C0 + 2 * tx;
            *reinterpret_cast<float2*>(o) = y0;
            *reinterpret_cast<float2*>(o + g.W) = y1;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same Winograd convolution on the bf16 matrix pipe: every fp32 product U * V as six bf16 piece products accumulated
// in fp32 (finding 13: hi + mid + lo hold all 24 mantissa bits), v_mfma_f32_32x32x16_bf16, chunks of 16
// reduction channels: 48 MFMAs of 32 cycles per wave and chunk where the fp32 kernel above issues 64 of 64 cycles.
//   * V never touches LDS.  Wave xi needs only row xi of B^T d, i.e. TWO rows of the raw patch; lane (tile = lane & 31,
//     octet = lane >> 5) is exactly the B operand's (column, 8 consecutive k) slot, so it reads its own 2 x 4 patch values
//     of its 8 channels from the raw patch in LDS (ds_read2_b32 on the even / odd column planes), forms the row
//     combination once per chunk and, per position nu, the column combination of its 8 channels, splits those 8 values
//     into three bf16 planes in registers and feeds them to the MFMAs.  Every V value is formed by exactly one lane.
//   * The raw patches arrive by LDS-DMA (buffer_load ... lds: no staging registers, halo pixels outside the image and
//     channels past the end read as zero through the descriptor's range check), channel pitch 256 floats so that a
//     thread's patch position is the same for every channel (one offset register), double-buffered, ONE barrier per chunk.
//   * U is pre-split by the pack kernels (gx_wino_h_word) and loaded straight into the A operand registers, one position
//     ahead of the MFMAs that use it.
#ifndef GX_WH_ABL
#define GX_WH_ABL 0           // measurement builds (tools/abl_build.sh; wrong results): 1 no MFMAs, 2 no A loads after the first,
#endif                        // 4 no transform / split (constant B), 8 no patch DMA after the first chunk, 16 no output transform
constexpr int HKC = 16;                        // reduction channels per chunk
typedef __bf16 w_bf16x8 __attribute__((ext_vector_type(8)));
typedef float w_f32x16 __attribute__((ext_vector_type(16)));

// 8 fp32 values -> the three bf16 planes (hi, mid, lo) of an MFMA operand
__device__ __forceinline__ void wino_split8(const float (&v)[8], w_bf16x8& ph, w_bf16x8& pm, w_bf16x8& pl) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)v[e];
        const float r1 = v[e] - (float)h;
        const __bf16 m = (__bf16)r1;
        ph[e] = h; pm[e] = m; pl[e] = (__bf16)(r1 - (float)m);
    }
}

// 8 fp32 values -> TWO fp16 pieces of v * sc (sc a power of two): hi = the top 11 significant bits (a mask: the residual is exact),
// lo = the residual rounded to nearest -- 22 significant bits (findings 40 and 42)
typedef _Float16 w_f16x8 __attribute__((ext_vector_type(8)));
typedef float w_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 w_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned w_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wino_split8_f16(const float (&v)[8], float sc, w_bf16x8& ph, w_bf16x8& pl) {
    w_u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s0 = v[2 * i] * sc, s1 = v[2 * i + 1] * sc;
        const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s0) & 0xFFFFE000u);
        const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s1) & 0xFFFFE000u);
        h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(w_f32x2{h0, h1}, w_f16x2));
        l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(w_f32x2{s0 - h0, s1 - h1}, w_f16x2));
    }
    ph = __builtin_bit_cast(w_bf16x8, h); pl = __builtin_bit_cast(w_bf16x8, l);
}

// NB: 32-tile MFMA column blocks per wave.
//   NB = 1: workgroup = 64 channels x 32 tiles (8 x 16 output pixels), 128 accumulator registers, two workgroups per CU --
//           the layers whose grid would not fill the chip with larger tiles.
//   NB = 2: 64 channels x 64 tiles (16 x 16 output pixels), 256 accumulator registers of the 512 a wave has to itself at one
//           workgroup per CU.  WHY: the A operands (U, 6 bytes per value as three bf16 pieces) are 24 KB per wave and chunk;
//           with 32 tiles per wave the four waves pull 96 KB through the CU's 64 B / clk vector-memory path per 1536 cycles of
//           MFMAs -- the path is saturated exactly when the matrix pipe would be (measured: 48.9 us, 39.3 us without the A
//           loads).  Twice the tiles per A operand halve that traffic per MFMA, and the register budget pays for an A
//           prefetch three positions deep instead of one.
template <int NB, bool F16 = false> struct WHCfg {
    static constexpr int TROWS = 4 * NB;                  // Winograd tile rows per workgroup (8 columns)
    static constexpr int PRH = 2 * TROWS + 2;             // raw patch rows (10 / 18), 18 columns as [parity 2][PLANE 10]
    static constexpr int USED = PRH * 20;                 // floats per channel (200 / 360)
    static constexpr int PITCH = NB == 1 ? 256 : 384;     // channel pitch: whole 64-lane DMA rows
    static constexpr int ROUNDS = (USED + 255) / 256;     // patch positions per thread and channel (1 / 2)
    static constexpr int RAW_FLOATS = HKC * PITCH;        // per buffer; two buffers
    // (fp16 pieces, a set of 16 registers instead of 24: fetching the A operands TWO positions ahead at 32 tiles per wave as well
    //  was built -- 84 bytes of scratch per lane at the 256 registers two workgroups per CU leave a wave; not kept)
    static constexpr int NSETS = NB == 1 ? 2 : 4;         // A register sets (position nu of a chunk uses set nu % NSETS)
    static constexpr int LOOK = NB == 1 ? 1 : 2;          // ... loaded LOOK positions ahead (three sets live at NB = 2: 72 registers)
    static constexpr size_t LDS_BYTES = 2 * RAW_FLOATS * 4 > 32768 ? 2 * RAW_FLOATS * 4 : 32768;    // (the epilogue's exchange buffer: 32 KB)
};

// F16: the operands as TWO fp16 pieces (U * 2^eU packed that way: kinds 45 / 46; V * 2^eV split in registers), three piece products
// per fp32 product instead of six -- half the MFMAs, two thirds of the A-operand traffic, a 24-VALU split per octet instead of 44.
// eU from the weight tensor's largest magnitude (the packing's trailer; |U| <= 2.25 max |g|), eV from the INPUT tensor's, handed
// over by the kernel that wrote it (g.xam*; |V| <= 4 max |d|); the outputs are scaled back by 2^-(eU + eV).
template <int NB, bool F16>
__global__ void __launch_bounds__(256, NB == 1 ? 2 : 1)
wino_conv_h_kernel(const float* __restrict__ in, const float* __restrict__ in2, const unsigned* __restrict__ U,
                   float* __restrict__ out, float* __restrict__ out2, const WinoGeom g) {
    using C = WHCfg<NB, F16>;
    constexpr int NP = F16 ? 2 : 3;
    constexpr int PITCH = C::PITCH, NSETS = C::NSETS, LOOK = C::LOOK;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* raw = lds;                       // [2][HKC][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // XCD-aware (tile, channel tile) map over the whole grid: neighbouring tiles (shared halos) AND the channel tiles of one tile (the
    // same patches) are consecutive blocks of one XCD, i.e. one L2
    const int xcd_l = gx_xcd_tile(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int my = xcd_l % (int)gridDim.y;
    int tile = xcd_l / (int)gridDim.y;
    const int tw_i = tile % g.tiles_w; tile /= g.tiles_w;
    const int th_i = tile % g.tiles_h; tile /= g.tiles_h;
    const int n = tile;
    const int R0 = th_i * (2 * C::TROWS), C0 = tw_i * (2 * WTW);
    const int m0 = my * 64;
    const int HW = g.H * g.W;
    const int Ka = g.K1 < g.K ? g.K1 : g.K, Kb = g.K - Ka;
    const float* in_n = in + (size_t)n * Ka * HW;
    const float* in2_n = in2 + (size_t)n * Kb * HW;
    const int nchunks = g.Kpad / HKC;

    // ---- raw-patch staging: thread t owns patch positions t (+ 256) of EVERY channel: row p / 20, column parity
    // (p % 20) / 10, index p % 10 (9 = the spare slot of a plane row).  voff = byte offset of that pixel inside a channel
    // plane, or 1 GiB (out of every descriptor's range: reads as zero) for halo pixels outside the image and idle slots.
    int voff[C::ROUNDS];
#pragma unroll
    for (int r = 0; r < C::ROUNDS; ++r) {
        const int p = tid + 256 * r;
        voff[r] = 0x40000000;
        if (p < C::USED) {
            const int pr = p / 20, rem = p - pr * 20, par = rem / 10, idx = rem - par * 10;
            const int rr = R0 - 1 + pr, cc = C0 - 1 + 2 * idx + par;
            if (idx < 9 && rr >= 0 && rr < g.H && cc >= 0 && cc < g.W) voff[r] = (rr * g.W + cc) * 4;
        }
    }
    auto stage = [&](int k0, float* rbuf) {       // chunk starting at channel k0 -> rbuf by LDS-DMA
        const bool second = k0 >= g.K1;
        const int kr = second ? k0 - g.K1 : k0, left = (second ? Kb : Ka) - kr;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>((second ? in2_n : in_n) + (size_t)kr * HW), 0, left > 0 ? left * HW * 4 : 0, 0x00020000);
        float* dst = rbuf + __builtin_amdgcn_readfirstlane(wave) * 64;          // (the DMA adds lane * 4 itself)
#pragma unroll
        for (int q = 0; q < HKC; ++q)
#pragma unroll
            for (int r = 0; r < C::ROUNDS; ++r)
                if (r == 0 || wave < (PITCH - 256) / 64)      // (the second round covers positions 256 .. PITCH - 1 only)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + q * PITCH + r * 256),
                                                             4, voff[r] + q * HW * 4, 0, 0, 0);
    };

    // ---- this lane's B-operand slots: tile 32 nb + bn -> (ty, tx), channel octet kh; wave = xi
    const int bn = lane & 31, kh = lane >> 5;
    const int tx = bn & 7;
    // row xi of B^T d = d[ra] + sgn d[rb]:  xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1), rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sgn = wave == 1 ? 1.f : -1.f;
    int ta_off[NB], tb_off[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int ty = (bn >> 3) + 4 * nb;
        ta_off[nb] = (8 * kh) * PITCH + (2 * ty + ra) * 20 + tx;
        tb_off[nb] = (8 * kh) * PITCH + (2 * ty + rb) * 20 + tx;
    }

    // ---- A operands: [m tile][chunk][position][piece][m half][lane][16 B]
    const char* Uw = reinterpret_cast<const char*>(U) + ((size_t)my * nchunks * 16 + 4 * wave) * 6144 + lane * 16;
    auto load_a = [&](w_bf16x8 (&a)[2][3], int c, int nu) {
        const char* p = Uw + ((size_t)c * 16 + nu) * 6144;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                a[mi][pc] = *reinterpret_cast<const w_bf16x8*>(p + (pc * 2 + mi) * 1024);
    };

    w_f32x16 acc[4][2][NB];     // [nu][mi][nb]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[a][b][nb][c] = 0.f;

    w_bf16x8 as[NSETS][2][3];               // A register sets: position 4 c + nu uses set (4 c + nu) % NSETS = nu % NSETS
    // fp16 pieces: the two scales (uniform)
    float scV = 1.f;
    int f16_back = 0;
    if constexpr (F16) {
        float xm = 0.f;
        for (int i = lane; i < g.xn0; i += 64) xm = fmaxf(xm, fabsf(g.xam0[i]));
        for (int i = lane; i < g.xn1; i += 64) xm = fmaxf(xm, fabsf(g.xam1[i]));
#pragma unroll
        for (int of = 32; of >= 1; of >>= 1) xm = fmaxf(xm, __shfl_xor(xm, of, 64));
        const int eV = gx_f16_scale_exp(4.f * xm);
        const float wam = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(U) + gx_kq_h_amax_off(g.Kpad, g.Mpad, 16));
        const int eU = gx_f16_scale_exp(2.25f * wam);
        scV = ldexpf(1.f, eV);
        f16_back = -(eU + eV);
    }
#ifdef GX_WH_STAGGER
    // the two workgroups of a CU start in the same cycle and would run in lockstep (both waiting for their first patch, both
    // on the matrix pipe, both in the epilogue): the second one of each CU starts GX_WH_STAGGER x 64 cycles late
    if (((blockIdx.x + gridDim.x * blockIdx.y) >> 8) & 1)
        for (int i = 0; i < GX_WH_STAGGER; ++i) __builtin_amdgcn_s_sleep(1);
#endif
    stage(0, raw);
#pragma unroll
    for (int i = 0; i < LOOK; ++i) load_a(as[i], 0, i);       // (NSETS <= 4: positions 0 .. LOOK - 1 of chunk 0)

    for (int c = 0; c < nchunks; ++c) {
        __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): this chunk's patch has landed (own DMA) ...
        __syncthreads();                             // ... everyone's has; and everyone is done with the other buffer
        if (c + 1 < nchunks && !(GX_WH_ABL & 8)) stage((c + 1) * HKC, raw + ((c + 1) & 1) * C::RAW_FLOATS);
        const float* rbuf = raw + (c & 1) * C::RAW_FLOATS;
        // row combination t[nb][ch][j], j = patch column: column j sits in plane (j & 1) at index tx + (j >> 1)
        float t[NB][8][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const float* pa = rbuf + ta_off[nb] + ch * PITCH;
                const float* pb = rbuf + tb_off[nb] + ch * PITCH;
#pragma unroll
                for (int j = 0; j < 4; ++j) t[nb][ch][j] = fmaf(sgn, pb[(j & 1) * 10 + (j >> 1)], pa[(j & 1) * 10 + (j >> 1)]);
            }
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            w_bf16x8 (&acur)[2][3] = as[nu % NSETS];
            if (!(GX_WH_ABL & 2)) {
                // position 4 c + nu + LOOK into the set that position 4 c + nu - 1 has just finished with
                constexpr int dummy = 0; (void)dummy;
                const int pn = nu + LOOK;
                if (pn < 4) load_a(as[pn % NSETS], c, pn);
                else if (c + 1 < nchunks) load_a(as[pn % NSETS], c + 1, pn - 4);
            }
            w_bf16x8 b[NB][3];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                // column combination: nu 0: t0 - t2, 1: t1 + t2, 2: t2 - t1, 3: t1 - t3
                float v[8];
#pragma unroll
                for (int ch = 0; ch < 8; ++ch)
                    v[ch] = nu == 0 ? t[nb][ch][0] - t[nb][ch][2]
                                    : (nu == 1 ? t[nb][ch][1] + t[nb][ch][2] : (nu == 2 ? t[nb][ch][2] - t[nb][ch][1] : t[nb][ch][1] - t[nb][ch][3]));
                if (GX_WH_ABL & 4) { b[nb][0] = acur[0][0]; b[nb][1] = acur[0][1]; b[nb][2] = acur[0][2]; }
                else if constexpr (F16) wino_split8_f16(v, scV, b[nb][0], b[nb][1]);
                else wino_split8(v, b[nb][0], b[nb][1], b[nb][2]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    w_f32x16 cc = acc[nu][mi][nb];        // pieces: 0 hi, 1 mid, 2 lo; small terms first
                    if (GX_WH_ABL & 1) { cc[0] += (float)acur[mi][0][0] * (float)b[nb][0][0] + (float)acur[mi][1][1] * (float)b[nb][1][1] + (float)acur[mi][2][2] * (float)b[nb][2][2]; acc[nu][mi][nb] = cc; continue; }
                    if constexpr (F16) {      // pieces: 0 hi, 1 lo; small terms first
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(w_f16x8, acur[mi][1]), __builtin_bit_cast(w_f16x8, b[nb][0]), cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(w_f16x8, acur[mi][0]), __builtin_bit_cast(w_f16x8, b[nb][1]), cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(w_f16x8, acur[mi][0]), __builtin_bit_cast(w_f16x8, b[nb][0]), cc, 0, 0, 0);
                        acc[nu][mi][nb] = cc;
                        continue;
                    }
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[mi][1], b[nb][1], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[mi][2], b[nb][0], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[mi][0], b[nb][2], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[mi][1], b[nb][0], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[mi][0], b[nb][1], cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[mi][0], b[nb][0], cc, 0, 0, 0);
                    acc[nu][mi][nb] = cc;
                }
        }
    }

    // ---- output transform (as in wino_conv_kernel): right-multiply by A inside the wave, A^T across the four waves through
    // LDS, one (32-channel half, 32-tile block) at a time
    float* E = lds;
    const int tt = tid & 31;
    const bool second_out = m0 >= g.M1;
    const int Mo = second_out ? g.M - g.M1 : (g.M1 < g.M ? g.M1 : g.M), mbase = second_out ? g.M1 : 0;
    float* const outp = second_out ? out2 : out;
    const size_t out_n = (size_t)n * Mo * HW;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            __syncthreads();
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = e + 8 * r4 + 4 * kh;      // C/D layout: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                    const int rg = 4 * r4 + e;
                    const float q0 = acc[0][mi][nb][rg] + acc[1][mi][nb][rg] + acc[2][mi][nb][rg];
                    const float q1 = acc[1][mi][nb][rg] - acc[2][mi][nb][rg] - acc[3][mi][nb][rg];
                    E[((wave * 2 + 0) * 32 + row) * 32 + bn] = q0;
                    E[((wave * 2 + 1) * 32 + row) * 32 + bn] = q1;
                }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ml = (tid >> 5) + 8 * j;
                const int m = m0 + mi * 32 + ml;
                if (m >= g.M) continue;
                const int oy = (tt >> 3) + 4 * nb, ox = tt & 7;
                float q[4][2];
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    q[xi][0] = E[((xi * 2 + 0) * 32 + ml) * 32 + tt];
                    q[xi][1] = E[((xi * 2 + 1) * 32 + ml) * 32 + tt];
                }
                float2 y0, y1;
                y0.x = q[0][0] + q[1][0] + q[2][0];
                y0.y = q[0][1] + q[1][1] + q[2][1];
                y1.x = q[1][0] - q[2][0] - q[3][0];
                y1.y = q[1][1] - q[2][1] - q[3][1];
                if constexpr (F16) {
                    y0.x = ldexpf(y0.x, f16_back); y0.y = ldexpf(y0.y, f16_back);
                    y1.x = ldexpf(y1.x, f16_back); y1.y = ldexpf(y1.y, f16_back);
                }
                float* o = outp + out_n + (size_t)(m - mbase) * HW + (size_t)(R0 + 2 * oy) * g.W + C0 + 2 * ox;
                *reinterpret_cast<float2*>(o) = y0;
                *reinterpret_cast<float2*>(o + g.W) = y1;
            }
        }
}

// U for the bf16 pipe: the thread of an even k writes the three words (k, k + 1) of position p
__device__ __forceinline__ void wino_h_store(unsigned* __restrict__ U, float v0, float v1, int m, int k, int p, int Kpad16,
                                             bool f16 = false, int f16_exp = 0) {
    unsigned short pc[2][3];
    const float v[2] = {v0, v1};
    if (f16) {          // two fp16 pieces of v * 2^f16_exp (piece slot 2 stays unused)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float ws = ldexpf(v[e], f16_exp);
            const _Float16 h = (_Float16)ws;
            pc[e][0] = __builtin_bit_cast(unsigned short, h);
            pc[e][1] = __builtin_bit_cast(unsigned short, (_Float16)(ws - (float)h));
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) U[gx_wino_h_word(m, k, p, q, Kpad16)] = (unsigned)pc[0][q] | ((unsigned)pc[1][q] << 16);
        return;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const __bf16 h = (__bf16)v[e];
        const float r1 = v[e] - (float)h;
        const __bf16 mm = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)mm);
        pc[e][0] = __builtin_bit_cast(unsigned short, h);
        pc[e][1] = __builtin_bit_cast(unsigned short, mm);
        pc[e][2] = __builtin_bit_cast(unsigned short, l);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) U[gx_wino_h_word(m, k, p, q, Kpad16)] = (unsigned)pc[0][q] | ((unsigned)pc[1][q] << 16);
}

// amax: non-NULL = the fp16-piece form (two pieces of U * 2^e, e from the weights' largest magnitude *amax)
__global__ void wino_pack_h_kernel(const float* __restrict__ w, unsigned* __restrict__ U, int mode, int Co, int Ci, int Kpad16,
                                   int Mpad, const float* __restrict__ amax) {
    const int total = 16 * (Kpad16 / 2) * Mpad;
    const int f16_exp = amax ? gx_f16_scale_exp(2.25f * *amax) : 0;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int m = idx % Mpad, k = 2 * ((idx / Mpad) % (Kpad16 / 2)), p = idx / (Mpad * (Kpad16 / 2));
        wino_h_store(U, gx_wino_u_value(w, mode, Co, Ci, m, k, p), gx_wino_u_value(w, mode, Co, Ci, m, k + 1, p), m, k, p, Kpad16,
                     amax != nullptr, f16_exp);
    }
}

// the pair variants' operands (wino_pack_pair_kernel) for the bf16 pipe
// f16: two fp16 pieces of U * 2^e, e from max(|w1|, |w2|) -- the two floats wino_pair_amax_kernel left in Uf's trailer
__global__ void __launch_bounds__(1024)
wino_pair_amax_kernel(const float* __restrict__ w1, const float* __restrict__ w2, int n1, int n2, float* __restrict__ tf,
                      float* __restrict__ td) {
    const float r = gx_wg1024_amax(blockIdx.x == 0 ? w1 : w2, blockIdx.x == 0 ? n1 : n2);
    if (threadIdx.x == 0) { tf[1 + blockIdx.x] = r; td[1 + blockIdx.x] = r; }      // (slots 1, 2 of each direction's trailer)
}
__global__ void wino_pack_pair_h_kernel(const float* __restrict__ w1, const float* __restrict__ w2, unsigned* __restrict__ Uf,
                                        unsigned* __restrict__ Ud, int Co1, int Co2, int Ci, int KpadF, int MpadF, int KpadD,
                                        int MpadD, int f16) {
    const int totF = 16 * (KpadF / 2) * MpadF, totD = 16 * (KpadD / 2) * MpadD;
    int f16_exp = 0;
    if (f16) {
        float* tf = reinterpret_cast<float*>(reinterpret_cast<char*>(Uf) + gx_kq_h_amax_off(KpadF, MpadF, 16));
        float* td = reinterpret_cast<float*>(reinterpret_cast<char*>(Ud) + gx_kq_h_amax_off(KpadD, MpadD, 16));
        const float am = fmaxf(tf[1], tf[2]);          // (slots 1, 2: the two tensors' maxima; slot 0: their maximum, what the conv kernel reads)
        f16_exp = gx_f16_scale_exp(2.25f * am);
        if (blockIdx.x == 0 && threadIdx.x == 0) { tf[0] = am; td[0] = am; }
    }
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < totF + totD; idx += gridDim.x * blockDim.x) {
        if (idx < totF) {
            const int m = idx % MpadF, k = 2 * ((idx / MpadF) % (KpadF / 2)), p = idx / (MpadF * (KpadF / 2));
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
                v[e] = m < Co1 ? gx_wino_u_value(w1, 0, Co1, Ci, m, k + e, p) : gx_wino_u_value(w2, 0, Co2, Ci, m - Co1, k + e, p);
            wino_h_store(Uf, v[0], v[1], m, k, p, KpadF, f16 != 0, f16_exp);
        } else {
            const int i = idx - totF;
            const int m = i % MpadD, k = 2 * ((i / MpadD) % (KpadD / 2)), p = i / (MpadD * (KpadD / 2));
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
                v[e] = k + e < Co1 ? gx_wino_u_value(w1, 1, Co1, Ci, m, k + e, p) : gx_wino_u_value(w2, 1, Co2, Ci, m, k + e - Co1, p);
            wino_h_store(Ud, v[0], v[1], m, k, p, KpadD, f16 != 0, f16_exp);
        }
    }
}

}  // namespace

static int g_wino_h = -1;      // 1 (default): the bf16 pipe; GENESIS_WINO_BF16X6=0 / gx_wino_precision(0): the fp32 pipe
bool gx_wino_h_on() {
    if (g_wino_h < 0) {
        const char* env = getenv("GENESIS_WINO_BF16X6");
        g_wino_h = (env && env[0] == '0') ? 0 : 1;
    }
    return g_wino_h == 1;
}

// ---- the fp16-piece form: 2 (default) where the input tensor's partial maxima were handed in (gx_conv_input_amax);
// GENESIS_WINO_F16X3=0 / gx_wino_precision(1): six bf16 piece products everywhere
static double g_wino_flops_f16 = 0.0, g_wino_flops_all = 0.0;      // bf16-pipe launches since the process started (gx_wino_f16_share)
static int g_wino_f16 = -1;
static bool wino_f16_on() {
    if (g_wino_f16 < 0) {
        const char* env = getenv("GENESIS_WINO_F16X3");
        g_wino_f16 = (env && env[0] == '0') ? 0 : 1;
    }
    return g_wino_f16 == 1 && gx_wino_h_on();
}
namespace { struct WinoHint { const float* p0; const float* p1; int n0, n1; }; thread_local WinoHint t_wino_hint = {nullptr, nullptr, 0, 0}; }
// true: the NEXT Winograd launch of this thread runs on fp16 pieces (the caller packs kinds 45 / 46 for it)
// (every workgroup of the Winograd kernel reduces the partial maxima itself: worth it up to ~1500 of them -- GroupNorm(8) at any batch
//  of this workload has N x 8, the 128 x 128 model's pair data gradient 1280; InstanceNorm -- MONet's UNet, one partial per (image,
//  channel) -- has up to 8192: measured 14.7 against 12.2 us per launch on its layers, so those stay on bf16 pieces)
constexpr int kWinoAmaxMax = 1536;
bool gx_wino_f16_pending(void) { return wino_f16_on() && t_wino_hint.p0 && t_wino_hint.n0 > 0 && t_wino_hint.n0 + t_wino_hint.n1 <= kWinoAmaxMax; }
// the armed hint, for the other fp16-piece conv3x3 kernel (gx_kq.hip Q_C3H: <= 32 output channels); not cleared (the caller disarms)
bool gx_conv_input_hint(const float** p0, int* n0, const float** p1, int* n1) {
    if (!t_wino_hint.p0 || t_wino_hint.n0 <= 0) return false;
    *p0 = t_wino_hint.p0; *n0 = t_wino_hint.n0; *p1 = t_wino_hint.p1; *n1 = t_wino_hint.n1;
    return true;
}
extern "C" double gx_wino_f16_share(void) { return g_wino_flops_all > 0.0 ? g_wino_flops_f16 / g_wino_flops_all : 0.0; }
extern "C" int gx_conv_input_amax(const float* p0, int n0, const float* p1, int n1) {
    t_wino_hint.p0 = (p0 && n0 > 0) ? p0 : nullptr; t_wino_hint.n0 = t_wino_hint.p0 ? n0 : 0;
    t_wino_hint.p1 = (t_wino_hint.p0 && p1 && n1 > 0) ? p1 : nullptr; t_wino_hint.n1 = t_wino_hint.p1 ? n1 : 0;
    return GX_OK;
}

static bool wino_shape_ok(int N, int K, int M, int H, int W) {
    return N > 0 && K >= 16 && M >= 16 && H >= 8 && W >= 16 && (H % 8) == 0 && (W % 16) == 0 && H * W <= 32768;
}

// used by gx_conv3x3_fwd / _dgrad: Winograd where the shape allows it AND the grid fills the chip without a channel
// split (under-filled layers stay on the direct kernels, which split the reduction).  GENESIS_WINOGRAD=0 disables.
static int g_wino_mode = -1;   // 0 off, 1 auto (chip-filling layers), 2 every eligible shape; -1: not yet read from the env

bool gx_wino_eligible(int N, int K, int M, int H, int W) {
    if (g_wino_mode < 0) {
        const char* env = getenv("GENESIS_WINOGRAD");
        g_wino_mode = env ? (env[0] == '0' ? 0 : (env[0] == '2' ? 2 : 1)) : 1;
    }
    if (g_wino_mode == 0 || !wino_shape_ok(N, K, M, H, W)) return false;
    // a quarter-filled chip is enough where the direct kernel has next to no reduction to split (<= 32 input channels; measured
    // per layer at 16 x 16, N = 32 with tools/conv3_policy_time.py: 32 -> 64 20.0 -> 17.0 us, the data gradient of 128 -> 32
    // 25.8 -> 17.9 us; 256 -> 64 is 41.3 -> 48.9 and stays direct).  Wider layers would gain 3 - 10 us as well (64 -> 128: 29.5 ->
    // 19.7), but they are GENESIS-V2's, whose B = 32 step is pinned against its own chunks of two images
    // (test_full_batch_equals_sixteen_golden_sized_chunks): a layer that changes kernels with the batch size moves ReLU decisions
    // (DESIGN.md finding 18) for 0.4 % of the step.  GENESIS_WINO_SMALL=0: the chip-filling rule alone
    static const char* small_env = getenv("GENESIS_WINO_SMALL");
    const int wgs = N * (H / 8) * (W / 16) * gx_ceil_div(M, 64);
    return g_wino_mode == 2 || wgs >= 256 || (wgs >= 64 && K <= 32 && N >= 16 && !(small_env && small_env[0] == '0'));
}

// in2 / K1, out2 / M1: the pair variants (WinoGeom); nullptr / 0 for one input tensor and one output tensor
static int wino_launch(const float* in, const float* in2, int K1, const float* U, float* out, float* out2, int M1, int N,
                       int K, int M, int H, int W, hipStream_t s) {
    const bool h = gx_wino_h_on();          // (the operands in U were packed for the same pipe: wino_kpad / the pack kinds)
    const bool f16 = gx_wino_f16_pending(); // (... and, with the input's maxima handed in, as fp16 pieces: the caller asked the same question)
    WinoGeom g;
    g.xam0 = f16 ? t_wino_hint.p0 : nullptr; g.xn0 = f16 ? t_wino_hint.n0 : 0;
    g.xam1 = f16 ? t_wino_hint.p1 : nullptr; g.xn1 = f16 ? t_wino_hint.n1 : 0;
    t_wino_hint = WinoHint{nullptr, nullptr, 0, 0};          // one-shot
    g.N = N; g.H = H; g.W = W; g.K = K; g.M = M;
    g.Kpad = gx_round_up(K, h ? HKC : WKC);
    g.Mpad = gx_round_up(M, 64);
    g.K1 = in2 ? K1 : g.Kpad;
    g.M1 = out2 ? M1 : g.Mpad;
    if (!in2) in2 = in;
    if (!out2) out2 = out;
    // bf16 pipe: 32-tile workgroups, two per CU.  The 64-tile variant (one workgroup per CU, half the A-operand traffic per
    // MFMA, A prefetch two positions deep) was measured SLOWER -- 51.1 vs 48.6 us (64 -> 64 @ 64 x 64, B = 32), 94.6 vs 83.2 us
    // (128 -> 64): one wave per SIMD has nobody to hand the matrix pipe to while it waits -- and is kept behind
    // GENESIS_WINO_NB=2 for measurement only.
    static const char* nb_env = getenv("GENESIS_WINO_NB");
    const int nb = (h && !f16 && (H % 16) == 0 && nb_env && nb_env[0] == '2') ? 2 : 1;
    g.tiles_h = H / (2 * WTH * nb);
    g.tiles_w = W / (2 * WTW);
    static bool attr_set = false;
    const size_t lds = h ? (nb == 2 ? WHCfg<2>::LDS_BYTES : WHCfg<1>::LDS_BYTES)
                         : (size_t)(2 * RAW_FLOATS + 2 * V_FLOATS) * sizeof(float);
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_h_kernel<1, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_h_kernel<1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_h_kernel<2, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    {
        const double flops = 2.0 * N * (double)M * K * 9 * H * W;    // algorithmic (direct-sum) flops
        if (h) { g_wino_flops_all += flops; if (f16 && nb == 1) g_wino_flops_f16 += flops; }
        const double bytes = 4.0 * ((double)N * K * H * W + (double)N * M * H * W + 9.0 * K * M);
        GxProf pf(KID_WINO, s, flops, bytes);
        if (h && nb == 2)
            hipLaunchKernelGGL((wino_conv_h_kernel<2, false>), dim3(N * g.tiles_h * g.tiles_w, g.Mpad / 64), dim3(256), lds, s, in, in2,
                               reinterpret_cast<const unsigned*>(U), out, out2, g);
        else if (h && f16)
            hipLaunchKernelGGL((wino_conv_h_kernel<1, true>), dim3(N * g.tiles_h * g.tiles_w, g.Mpad / 64), dim3(256), lds, s, in, in2,
                               reinterpret_cast<const unsigned*>(U), out, out2, g);
        else if (h)
            hipLaunchKernelGGL((wino_conv_h_kernel<1, false>), dim3(N * g.tiles_h * g.tiles_w, g.Mpad / 64), dim3(256), lds, s, in, in2,
                               reinterpret_cast<const unsigned*>(U), out, out2, g);
        else
        hipLaunchKernelGGL(wino_conv_kernel, dim3(N * g.tiles_h * g.tiles_w, g.Mpad / 64), dim3(256), lds, s, in, in2, U,
                           out, out2, g);
    }
    GX_CHECK_LAUNCH("winograd conv3x3");
    return GX_OK;
}

int gx_wino_launch(const float* in, const float* U, float* out, int N, int K, int M, int H, int W, hipStream_t s) {
    return wino_launch(in, nullptr, 0, U, out, nullptr, 0, N, K, M, H, W, s);
}

extern "C" {

int gx_conv3x3_wino_supported(int N, int Cin, int Cout, int H, int W) { return wino_shape_ok(N, Cin, Cout, H, W); }

// dispatch policy of gx_conv3x3_fwd / _dgrad: 0 never, 1 layers that fill the chip (default), 2 every supported shape
int gx_conv3x3_wino_policy(int mode) {
    GX_CHECK_ARG(mode >= 0 && mode <= 2, "gx_conv3x3_wino_policy: mode must be 0, 1 or 2");
    g_wino_mode = mode;
    return GX_OK;
}

// bytes of one direction's packed operands (either pipe's layout fits: the bf16 one is 1.5 x the fp32 one)
static size_t wino_u_bytes(int K, int M) {
    const size_t f = (size_t)16 * gx_round_up(K, 8) * gx_round_up(M, 64) * sizeof(float);
    const size_t h = gx_wino_h_bytes(gx_round_up(K, HKC), gx_round_up(M, 64)) + 16384;      // (+ the trailer of the fp16-piece packing: the weights' largest magnitude)
    return f > h ? f : h;
}

int gx_wino_precision(int mode) {
    GX_CHECK_ARG(mode >= -1 && mode <= 2, "gx_wino_precision: mode must be 0 (fp32 matrix pipe), 1 (bf16 pipe, six piece products), 2 (as 1, "
                                          "three fp16 piece products where the input's maxima are handed in) or -1 (the environment's default)");
    if (mode < 0) { g_wino_h = -1; g_wino_f16 = -1; return GX_OK; }
    g_wino_h = mode ? 1 : 0;
    g_wino_f16 = mode == 2 ? 1 : 0;
    return GX_OK;
}

size_t gx_conv3x3_wino_ws_bytes(int N, int Cin, int Cout, int H, int W) {
    (void)N; (void)H; (void)W;
    const int c = Cin > Cout ? Cin : Cout;
    return wino_u_bytes(c, c);
}

// mode 0: y[N,Cout,H,W] = conv3x3(x[N,Cin,H,W], w[Cout,Cin,3,3]); mode 1: dx[N,Cin,H,W] from dy[N,Cout,H,W] (same w)
int gx_conv3x3_wino(const float* x, const float* w, float* y, int N, int Cin, int Cout, int H, int W, int mode,
                    void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(x && w && y && ws, "gx_conv3x3_wino: null pointer");
    GX_CHECK_ARG(mode == 0 || mode == 1, "gx_conv3x3_wino: mode must be 0 (forward) or 1 (data gradient)");
    GX_CHECK_ARG(wino_shape_ok(N, Cin, Cout, H, W),
                 "gx_conv3x3_wino: needs H %% 8 == 0, W %% 16 == 0, channels >= 16 (N=%d Cin=%d Cout=%d H=%d W=%d)", N, Cin,
                 Cout, H, W);
    GX_CHECK_ARG(ws_bytes >= gx_conv3x3_wino_ws_bytes(N, Cin, Cout, H, W), "gx_conv3x3_wino: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int K = mode == 0 ? Cin : Cout, M = mode == 0 ? Cout : Cin;
    const bool h = gx_wino_h_on();
    const bool f16 = gx_wino_f16_pending();
    const int Kpad = gx_round_up(K, h ? HKC : WKC), Mpad = gx_round_up(M, 64);
    float* U = (float*)ws;
    float* trailer = reinterpret_cast<float*>(reinterpret_cast<char*>(U) + gx_kq_h_amax_off(Kpad, Mpad, 16));
    if (f16) {
        const int rc = gx_kq_weight_amax_launch(w, Cout * Cin * 9, trailer, s);
        if (rc) return rc;
    }
    {
        const int total = 16 * Kpad * Mpad;
        GxProf pf(KID_PACK_WEIGHTS, s, 0.0, 8.0 * total);
        if (h)
            hipLaunchKernelGGL(wino_pack_h_kernel, dim3(gx_ceil_div(total / 2, 256)), dim3(256), 0, s, w, (unsigned*)U, mode, Cout,
                               Cin, Kpad, Mpad, f16 ? (const float*)trailer : (const float*)nullptr);
        else
        hipLaunchKernelGGL(wino_pack_kernel, dim3(gx_ceil_div(total, 256)), dim3(256), 0, s, w, U, mode, Cout, Cin, Kpad,
                           Mpad);
    }
    GX_CHECK_LAUNCH("gx_conv3x3_wino(pack)");
    return gx_wino_launch(x, U, y, N, K, M, H, W, s);
}

// ---- two conv3x3 layers on one input as one layer (forward: one launch writes both outputs; data gradient: one launch
// sums both layers' input gradients).  Co1 % 64 == 0 and the Winograd shape rules for (Cin, Co1 + Co2).
int gx_conv3x3_pair_supported(int N, int Cin, int Co1, int Co2, int H, int W) {
    return Co1 > 0 && Co2 > 0 && wino_shape_ok(N, Cin, Co1 + Co2, H, W) && Co1 % 64 == 0;
}

size_t gx_conv3x3_pair_ws_bytes(int N, int Cin, int Co1, int Co2, int H, int W) {
    (void)N; (void)H; (void)W;
    const int Co = Co1 + Co2;
    return wino_u_bytes(Cin, Co) + wino_u_bytes(Co, Cin);
}

static void pair_u(void* ws, int Cin, int Co, float** Uf, float** Ud) {
    *Uf = (float*)ws;
    *Ud = *Uf + wino_u_bytes(Cin, Co) / sizeof(float);
}

// both directions' operands of a layer pair in one launch, for the pipe in force
// f16: both directions as fp16 pieces (one scale for the pair, from max(|w1|, |w2|): an amax launch ahead of the pack)
static int pair_pack(const float* w1, const float* w2, float* Uf, float* Ud, int Cin, int Co1, int Co2, hipStream_t s, bool f16) {
    const int Co = Co1 + Co2;
    const bool h = gx_wino_h_on();
    const int kq = h ? HKC : WKC;
    const int KpF = gx_round_up(Cin, kq), MpF = gx_round_up(Co, 64), KpD = gx_round_up(Co, kq), MpD = gx_round_up(Cin, 64);
    const int total = 16 * (KpF * MpF + KpD * MpD);
    if (h && f16) {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * 9 * Cin * Co);
        hipLaunchKernelGGL(wino_pair_amax_kernel, dim3(2), dim3(1024), 0, s, w1, w2, Co1 * Cin * 9, Co2 * Cin * 9,
                           reinterpret_cast<float*>(reinterpret_cast<char*>(Uf) + gx_kq_h_amax_off(KpF, MpF, 16)),
                           reinterpret_cast<float*>(reinterpret_cast<char*>(Ud) + gx_kq_h_amax_off(KpD, MpD, 16)));
    }
    GxProf pf(KID_PACK_WEIGHTS, s, 0.0, 8.0 * total);
    if (h)
        hipLaunchKernelGGL(wino_pack_pair_h_kernel, dim3(gx_ceil_div(total / 2, 256)), dim3(256), 0, s, w1, w2, (unsigned*)Uf,
                           (unsigned*)Ud, Co1, Co2, Cin, KpF, MpF, KpD, MpD, f16 ? 1 : 0);
    else
        hipLaunchKernelGGL(wino_pack_pair_kernel, dim3(gx_ceil_div(total, 256)), dim3(256), 0, s, w1, w2, Uf, Ud, Co1, Co2, Cin,
                           KpF, MpF, KpD, MpD);
    return GX_OK;
}

// y1 [N,Co1,H,W] = conv3x3(x, w1), y2 [N,Co2,H,W] = conv3x3(x, w2); ws keeps the packed weights of BOTH directions
// (gx_conv3x3_pair_dgrad with pack = 0 reuses them within the iteration)
int gx_conv3x3_pair_fwd(const float* x, const float* w1, const float* w2, float* y1, float* y2, int N, int Cin, int Co1,
                        int Co2, int H, int W, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(x && w1 && w2 && y1 && y2 && ws, "gx_conv3x3_pair_fwd: null pointer");
    GX_CHECK_ARG(gx_conv3x3_pair_supported(N, Cin, Co1, Co2, H, W), "gx_conv3x3_pair_fwd: unsupported shape");
    GX_CHECK_ARG(ws_bytes >= gx_conv3x3_pair_ws_bytes(N, Cin, Co1, Co2, H, W), "gx_conv3x3_pair_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int Co = Co1 + Co2;
    float *Uf, *Ud;
    pair_u(ws, Cin, Co, &Uf, &Ud);
    // (an armed gx_conv_input_amax hint: BOTH directions are packed as fp16 pieces -- gx_conv3x3_pair_dgrad with pack = 0 must then
    //  be given its inputs' maxima too, or re-pack)
    pair_pack(w1, w2, Uf, Ud, Cin, Co1, Co2, s, gx_wino_f16_pending());
    GX_CHECK_LAUNCH("gx_conv3x3_pair_fwd(pack)");
    return wino_launch(x, nullptr, 0, Uf, y1, y2, Co1, N, Cin, Co, H, W, s);
}

// dx [N,Cin,H,W] = dgrad(dy1, w1) + dgrad(dy2, w2); pack != 0: (re)pack the weights into ws first
int gx_conv3x3_pair_dgrad(const float* dy1, const float* dy2, const float* w1, const float* w2, float* dx, int N, int Cin,
                          int Co1, int Co2, int H, int W, int pack, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(dy1 && dy2 && w1 && w2 && dx && ws, "gx_conv3x3_pair_dgrad: null pointer");
    GX_CHECK_ARG(gx_conv3x3_pair_supported(N, Cin, Co1, Co2, H, W), "gx_conv3x3_pair_dgrad: unsupported shape");
    GX_CHECK_ARG(ws_bytes >= gx_conv3x3_pair_ws_bytes(N, Cin, Co1, Co2, H, W), "gx_conv3x3_pair_dgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int Co = Co1 + Co2;
    float *Uf, *Ud;
    pair_u(ws, Cin, Co, &Uf, &Ud);
    if (pack) {
        pair_pack(w1, w2, Uf, Ud, Cin, Co1, Co2, s, gx_wino_f16_pending());
        GX_CHECK_LAUNCH("gx_conv3x3_pair_dgrad(pack)");
    }
    return wino_launch(dy1, dy2, Co1, Ud, dx, nullptr, 0, N, Co, Cin, H, W, s);
}

}  // extern "C"

