// Small dense layers (nn.Linear) of the hot path: the UNet bottleneck MLP (modules/unet.py:58-62), z_head
// (models/genesisv2_config.py:76-80), feat_head[1] applied to the pooled slot sums, prior_linear
// (models/genesis_config.py:106).  Rows M are 32..224 and the feature dims 64..2048: 7-17 MFLOP per layer, i.e.
// pure latency.  The BLAS library picks 128x224 / 256x192 macro tiles for them -- a single workgroup walking the
// whole K loop, 15-70 us per call, ~0.8 ms per training step.  Here every 16x16 output tile is its own workgroup,
// the contraction is split over the workgroup's 4-16 waves (fp32 MFMA 16x16x4, exact fp32 products) and the
// per-wave partial tiles are summed through LDS in a fixed order (deterministic); bias, ReLU, the ReLU mask of
// the backward pass and the bias gradient are fused, so a layer is 1 launch forward and 2 backward.
#include "gx_common.h"

namespace {

// C[I,J] = sum_kk A(i,kk) B(kk,j).
//   A_KC: A(i,kk) = a[i*lda + kk] (contraction contiguous)   else a[kk*lda + i]
//   B_KC: B(kk,j) = b[j*ldb + kk]                            else b[kk*ldb + j]
//   MASK: A(i,kk) is multiplied by [mask(i,kk) > 0] (mask has A's layout): the ReLU derivative from the layer output
//   ROWSUM: also emits rs[i] = sum_kk A(i,kk) (bias gradient; rs2: a second copy), by the workgroups of the first
//   column tile.  act: 0 none, 1 ReLU, 2 add to what the destination holds
// Lane (idx = lane & 15, kq = lane >> 4) owns contraction indices k16 + 4 kq + {0..3}; MFMA step jj consumes index
// 4 kq + jj from every kq -- a permutation of the 16 indices that A and B share, so contiguous operands load float4.
// derivative of the activation from the layer OUTPUT m: ReLU [m > 0]; ELU 1 where m > 0, else e^x = m + 1
__device__ __forceinline__ float mask_factor(float m, bool elu) { return m > 0.f ? 1.f : (elu ? m + 1.f : 0.f); }

template <bool A_KC, bool B_KC, bool MASK, bool ROWSUM>
__device__ __forceinline__ void
dense_tile(float (*part)[256], float (*rpart)[16], int bx, int by, const float* __restrict__ a, int lda,
           const float* __restrict__ mask, const float* __restrict__ b, int ldb, const float* __restrict__ bias,
           int act, float* __restrict__ c, int ldc, int I, int J, int Kc, float* __restrict__ rs, int vec_a,
           int vec_b, float* __restrict__ rs2 = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int idx = lane & 15, kq = lane >> 4;
    const int i0 = by * 16, j0 = bx * 16;
    const bool elu_mask = (act & 4) != 0;        // MASK: the mask is an ELU output (derivative y + 1 where y <= 0)
    const bool acc_rs = (act & 8) != 0;          // ROWSUM: add to what rs / rs2 hold (a parameter used again in one iteration)
    act &= 3;
    const int ai = i0 + idx, bj = j0 + idx;
    const bool a_ok = ai < I, b_ok = bj < J;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float rsum = 0.f;
#pragma unroll 2
    for (int k16 = wave * 16; k16 < Kc; k16 += nw * 16) {
        const int kb = k16 + 4 * kq;
        float av[4], bv[4];
        if (A_KC) {
            if (vec_a) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f}, m = {1.f, 1.f, 1.f, 1.f};
                if (a_ok && kb < Kc) {
                    t = *reinterpret_cast<const f32x4*>(a + (size_t)ai * lda + kb);
                    if (MASK) m = *reinterpret_cast<const f32x4*>(mask + (size_t)ai * lda + kb);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) av[jj] = MASK ? t[jj] * mask_factor(m[jj], elu_mask) : t[jj];
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float t = 0.f;
                    if (a_ok && kb + jj < Kc) {
                        t = a[(size_t)ai * lda + kb + jj];
                        if (MASK) t *= mask_factor(mask[(size_t)ai * lda + kb + jj], elu_mask);
                    }
                    av[jj] = t;
                }
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float t = 0.f;
                if (a_ok && kb + jj < Kc) {
                    t = a[(size_t)(kb + jj) * lda + ai];
                    if (MASK) t *= mask_factor(mask[(size_t)(kb + jj) * lda + ai], elu_mask);
                }
                av[jj] = t;
            }
        }
        if (B_KC) {
            if (vec_b) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (b_ok && kb < Kc) t = *reinterpret_cast<const f32x4*>(b + (size_t)bj * ldb + kb);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bv[jj] = t[jj];
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    bv[jj] = (b_ok && kb + jj < Kc) ? b[(size_t)bj * ldb + kb + jj] : 0.f;
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                bv[jj] = (b_ok && kb + jj < Kc) ? b[(size_t)(kb + jj) * ldb + bj] : 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], bv[jj], acc, 0, 0, 0);
            if (ROWSUM) rsum += av[jj];
        }
    }
    // C/D layout (16x16): col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][(kq * 4 + r) * 16 + idx] = acc[r];
    if (ROWSUM) {
        rsum += __shfl_xor(rsum, 16, 64);
        rsum += __shfl_xor(rsum, 32, 64);
        if (kq == 0) rpart[wave][idx] = rsum;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int t = threadIdx.x;
        float s = part[0][t];
        for (int w = 1; w < nw; ++w) s += part[w][t];
        const int ci = i0 + (t >> 4), cj = j0 + (t & 15);
        if (ci < I && cj < J) {
            if (bias) s += bias[cj];
            if (act == 1) s = s > 0.f ? s : 0.f;
            if (act == 2) s += c[(size_t)ci * ldc + cj];      // accumulate into the destination
            if (act == 3) s = s > 0.f ? s : expm1f(s);        // ELU (alpha 1)
            c[(size_t)ci * ldc + cj] = s;
        }
    }
    if (ROWSUM && bx == 0 && threadIdx.x < 16 && i0 + (int)threadIdx.x < I) {
        float s = rpart[0][threadIdx.x];
        for (int w = 1; w < nw; ++w) s += rpart[w][threadIdx.x];
        if (acc_rs) { rs[i0 + threadIdx.x] += s; if (rs2) rs2[i0 + threadIdx.x] += s; }
        else { rs[i0 + threadIdx.x] = s; if (rs2) rs2[i0 + threadIdx.x] = s; }
    }
}

template <bool A_KC, bool B_KC, bool MASK, bool ROWSUM>
__global__ void __launch_bounds__(1024)
dense_kernel(const float* __restrict__ a, int lda, const float* __restrict__ mask, const float* __restrict__ b,
             int ldb, const float* __restrict__ bias, int act, float* __restrict__ c, int ldc, int I, int J, int Kc,
             float* __restrict__ rs, int vec_a, int vec_b, float* __restrict__ rs2) {
    __shared__ float part[16][256];
    __shared__ float rpart[16][16];
    dense_tile<A_KC, B_KC, MASK, ROWSUM>(part, rpart, blockIdx.x, blockIdx.y, a, lda, mask, b, ldb, bias, act, c, ldc,
                                         I, J, Kc, rs, vec_a, vec_b, rs2);
}

// Both halves of a layer's backward in one launch (they are independent, and a kernel boundary costs more than
// either): rows [0, gy_dx) of the grid compute dx[M,K] = dpre[M,N] w[N,K], the rest dw[N,K] = dpre^T x (+ db).
template <bool MASK, bool ROWSUM>
__global__ void __launch_bounds__(1024)
dense_bwd_pair_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ y, const float* __restrict__ w,
                      const float* __restrict__ x, int ldx, float* __restrict__ dx, int lddx, int dx_act,
                      float* __restrict__ dw, float* __restrict__ db, float* __restrict__ db2, int M, int N, int K,
                      int gy_dx, int vec) {
    __shared__ float part[16][256];
    __shared__ float rpart[16][16];
    if ((int)blockIdx.y < gy_dx)
        dense_tile<true, false, MASK, false>(part, rpart, blockIdx.x, blockIdx.y, g, ldg, y, w, K, nullptr, dx_act & 7, dx,
                                             lddx, M, K, N, nullptr, vec, 0);
    else
        dense_tile<false, false, MASK, ROWSUM>(part, rpart, blockIdx.x, blockIdx.y - gy_dx, g, ldg, y, x, ldx, nullptr,
                                               (dx_act & 4) | ((dx_act & 16) ? 10 : 0), dw, K, N, K, M, db, 0, 0, db2);
}

// ------------------------------------------------------------------------------------------------ LSTM cell
// nn.LSTM (single layer, gate order i, f, g, o) unrolled by the caller, one launch per time step: the recurrent
// GEMM h_prev w_hh^T of a 16 (batch) x 16 (hidden units) tile for all four gates, then the cell update in the
// epilogue.  The library path issues ~8 launches per step (GEMM, bias, gate slicing, pointwise update).
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// (one step's work of a workgroup as a device function: the per-step kernel and the whole-sequence kernel below run the SAME
//  instruction sequence, so their results are identical bit for bit; h_prev / c_prev may have been written earlier in the same
//  launch by other workgroups -- plain pointers, no __restrict__)
// COH (the whole-sequence kernels): values other workgroups read / wrote inside this launch move as agent-scope relaxed atomics --
// write-through stores and cache-bypassing loads (global_store / global_load ... sc1), no cache-wide write-back or invalidate
__device__ __forceinline__ f32x4 lstm_load4(const float* p, bool coh) {
    if (!coh) return *reinterpret_cast<const f32x4*>(p);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
__device__ __forceinline__ void lstm_store(float* p, float v, bool coh) {
    if (coh) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

template <bool COH>
__device__ __forceinline__ void
lstm_fwd_body(float (&part)[4][4][256], const float* __restrict__ gx, const float* h_prev, const float* c_prev,
              const float* __restrict__ w_hh, const float* __restrict__ b_hh, int B, int H, float* act, float* c_out,
              float* h_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const bool a_ok = i0 + idx < B;
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[q][0] = 0.f; acc[q][1] = 0.f; acc[q][2] = 0.f; acc[q][3] = 0.f; }
    if (h_prev) {
        for (int k16 = wave * 16; k16 < H; k16 += 64) {
            const int kb = k16 + 4 * kq;
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            if (a_ok) av = lstm_load4(h_prev + (size_t)(i0 + idx) * H + kb, COH);
            f32x4 bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bv[q] = *reinterpret_cast<const f32x4*>(w_hh + (size_t)(q * H + j0 + idx) * H + kb);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], bv[q][jj], acc[q], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][q][(kq * 4 + r) * 16 + idx] = acc[q][r];
    __syncthreads();
    const int t = threadIdx.x;
    const int b = i0 + (t >> 4), u = j0 + (t & 15);
    if (b < B) {
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s = part[0][q][t];
#pragma unroll
            for (int w = 1; w < 4; ++w) s += part[w][q][t];
            pre[q] = s + gx[(size_t)b * 4 * H + q * H + u] + b_hh[q * H + u];
        }
        const float ig = sigmoid_f(pre[0]), fg = sigmoid_f(pre[1]), gg = tanhf(pre[2]), og = sigmoid_f(pre[3]);
        const float cp = c_prev ? c_prev[(size_t)b * H + u] : 0.f;
        const float cn = fg * cp + ig * gg;
        float* ar = act + (size_t)b * 4 * H + u;
        ar[0] = ig; ar[H] = fg; ar[2 * H] = gg; ar[3 * H] = og;
        c_out[(size_t)b * H + u] = cn;
        lstm_store(h_out + (size_t)b * H + u, og * tanhf(cn), COH);
    }
}

__global__ void __launch_bounds__(256)
lstm_step_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ h_prev,
                     const float* __restrict__ c_prev, const float* __restrict__ w_hh,
                     const float* __restrict__ b_hh, int B, int H, float* __restrict__ act,
                     float* __restrict__ c_out, float* __restrict__ h_out) {
    __shared__ float part[4][4][256];
    lstm_fwd_body<false>(part, gx, h_prev, c_prev, w_hh, b_hh, B, H, act, c_out, h_out);
}

// ---- the whole sequence in ONE launch (the AR prior's LSTM: T = K - 1 = 6 steps of B = 32, H = 256 -- six launches of ~6 us each
//      of which ~1 us is work).  Same grid as one step (H / 16 x ceil(B / 16) workgroups, all resident at once: the host refuses
//      grids past kLstmSeqMaxWg); between two steps every workgroup needs every other's h -> a grid-wide barrier on one counter per
//      step boundary in `bar` (kLstmSeqBar zero-initialised unsigned ints owned by the caller).  What crosses a barrier (h forward,
//      dgates backward) is written with agent-scope write-through stores and read with agent-scope loads, addresses no cache of the
//      reading CU / XCD has touched in this launch; the barrier itself is then: wait for this wave's stores, workgroup barrier, one
//      atomic add, spin on agent-scope loads, workgroup barrier.  (First form, measured: release fence + acquire fence around the
//      counter = buffer_wbl2 sc1 / buffer_inv sc1, a write-back and an invalidate of the XCD's whole L2 per step boundary: 38 us
//      forward and 58 us backward for T = 6 -- SLOWER than the 35 + 38 us of the twelve step launches.)  The last workgroup through
//      the final counter zeroes them all: every other workgroup has passed every barrier by then, and the next launch finds the
//      state it needs.
constexpr int kLstmSeqBar = 16;
constexpr int kLstmSeqMaxWg = 128;

__device__ __forceinline__ void lstm_grid_barrier(unsigned* bar, unsigned nwg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__device__ __forceinline__ void lstm_grid_finish(unsigned* bar, unsigned nwg) {
    if (threadIdx.x == 0) {
        const unsigned seen = __hip_atomic_fetch_add(bar + kLstmSeqBar - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == nwg - 1) {
            for (int i = 0; i < kLstmSeqBar; ++i) __hip_atomic_store(bar + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(256)
lstm_seq_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ w_hh, const float* __restrict__ b_hh, int T, int B,
                    int H, float* act, float* c, float* h, unsigned* bar) {
    __shared__ float part[4][4][256];
    const unsigned nwg = gridDim.x * gridDim.y;
    const size_t sh = (size_t)B * H;
    for (int t = 0; t < T; ++t) {
        if (t) lstm_grid_barrier(bar + (t - 1), nwg);
        lstm_fwd_body<true>(part, gx + (size_t)t * 4 * sh, t ? h + (size_t)(t - 1) * sh : nullptr, t ? c + (size_t)(t - 1) * sh : nullptr,
                            w_hh, b_hh, B, H, act + (size_t)t * 4 * sh, c + (size_t)t * sh, h + (size_t)t * sh);
    }
    lstm_grid_finish(bar, nwg);
}

// dh = g_h + dgates_next w_hh; dc = dc_next + dh o (1 - tanh(c)^2); dgates = (dc g i(1-i), dc c_prev f(1-f),
// dc i (1-g^2), dh tanh(c) o(1-o)); dc_prev = dc f.
template <bool COH>
__device__ __forceinline__ void
lstm_bwd_body(float (&part)[16][256], const float* __restrict__ g_h, const float* dgates_next, const float* __restrict__ w_hh,
              const float* __restrict__ act, const float* __restrict__ c, const float* __restrict__ c_prev, const float* dc_next,
              int B, int H, float* dgates, float* dc_prev) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int idx = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const bool a_ok = i0 + idx < B;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (dgates_next) {
        const int Kc = 4 * H;
        for (int k16 = wave * 16; k16 < Kc; k16 += nw * 16) {
            const int kb = k16 + 4 * kq;
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            if (a_ok) av = lstm_load4(dgates_next + (size_t)(i0 + idx) * Kc + kb, COH);
            float bv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) bv[jj] = w_hh[(size_t)(kb + jj) * H + j0 + idx];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], bv[jj], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][(kq * 4 + r) * 16 + idx] = acc[r];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int t = threadIdx.x;
        const int b = i0 + (t >> 4), u = j0 + (t & 15);
        if (b < B) {
            float dh = part[0][t];
            for (int w = 1; w < nw; ++w) dh += part[w][t];
            if (g_h) dh += g_h[(size_t)b * H + u];
            const float* ar = act + (size_t)b * 4 * H + u;
            const float ig = ar[0], fg = ar[H], gg = ar[2 * H], og = ar[3 * H];
            const float tc = tanhf(c[(size_t)b * H + u]);
            const float cp = c_prev ? c_prev[(size_t)b * H + u] : 0.f;
            const float dc = (dc_next ? dc_next[(size_t)b * H + u] : 0.f) + dh * og * (1.f - tc * tc);
            float* dr = dgates + (size_t)b * 4 * H + u;
            lstm_store(dr, dc * gg * ig * (1.f - ig), COH);
            lstm_store(dr + H, dc * cp * fg * (1.f - fg), COH);
            lstm_store(dr + 2 * H, dc * ig * (1.f - gg * gg), COH);
            lstm_store(dr + 3 * H, dh * tc * og * (1.f - og), COH);
            dc_prev[(size_t)b * H + u] = dc * fg;
        }
    }
}

__global__ void __launch_bounds__(1024)
lstm_step_bwd_kernel(const float* __restrict__ g_h, const float* __restrict__ dgates_next,
                     const float* __restrict__ w_hh, const float* __restrict__ act, const float* __restrict__ c,
                     const float* __restrict__ c_prev, const float* __restrict__ dc_next, int B, int H,
                     float* __restrict__ dgates, float* __restrict__ dc_prev) {
    __shared__ float part[16][256];
    lstm_bwd_body<false>(part, g_h, dgates_next, w_hh, act, c, c_prev, dc_next, B, H, dgates, dc_prev);
}

// the backward of the whole sequence in one launch: steps T-1 .. 0, step t reads every workgroup's dgates[t + 1] (a grid barrier per
// step boundary, as above); dc travels through two [B,H] planes of `dc2`, each element written and read by the same thread
__global__ void __launch_bounds__(1024)
lstm_seq_bwd_kernel(const float* __restrict__ g_h, const float* __restrict__ w_hh, const float* __restrict__ act,
                    const float* __restrict__ c, int T, int B, int H, float* dgates, float* dc2, unsigned* bar) {
    __shared__ float part[16][256];
    const unsigned nwg = gridDim.x * gridDim.y;
    const size_t sh = (size_t)B * H;
    for (int t = T - 1; t >= 0; --t) {
        if (t + 1 < T) lstm_grid_barrier(bar + t, nwg);
        lstm_bwd_body<true>(part, g_h + (size_t)t * sh, t + 1 < T ? dgates + (size_t)(t + 1) * 4 * sh : nullptr, w_hh,
                      act + (size_t)t * 4 * sh, c + (size_t)t * sh, t ? c + (size_t)(t - 1) * sh : nullptr,
                      t + 1 < T ? dc2 + (size_t)((t + 1) & 1) * sh : nullptr, B, H, dgates + (size_t)t * 4 * sh,
                      dc2 + (size_t)(t & 1) * sh);
    }
    lstm_grid_finish(bar, nwg);
}


// ---- y[M,N] = x[M,K] w[K,N], K small (a latent size), N large: one thread owns four consecutive columns and MR rows; w is
// read with coalesced 16-byte loads (once per MR rows of x), the x values are wave-uniform (scalar loads)
constexpr int kNN_MR = 8;
__global__ void __launch_bounds__(256)
matmul_nn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int M, int N, int K) {
    const int n0 = (blockIdx.y * 256 + threadIdx.x) * 4;
    const int m0 = blockIdx.x * kNN_MR;
    if (n0 >= N) return;
    float4 acc[kNN_MR];
#pragma unroll
    for (int r = 0; r < kNN_MR; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xr[kNN_MR];
#pragma unroll
    for (int r = 0; r < kNN_MR; ++r) xr[r] = x + (size_t)(m0 + r < M ? m0 + r : M - 1) * K;     // (rows past the end: not stored)
    for (int k = 0; k < K; ++k) {
        const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)k * N + n0);
#pragma unroll
        for (int r = 0; r < kNN_MR; ++r) {
            const float xv = xr[r][k];
            acc[r].x = fmaf(xv, w4.x, acc[r].x); acc[r].y = fmaf(xv, w4.y, acc[r].y);
            acc[r].z = fmaf(xv, w4.z, acc[r].z); acc[r].w = fmaf(xv, w4.w, acc[r].w);
        }
    }
#pragma unroll
    for (int r = 0; r < kNN_MR; ++r)
        if (m0 + r < M) *reinterpret_cast<float4*>(y + (size_t)(m0 + r) * N + n0) = acc[r];
}

// dw[K,N] = sum_m x[m,k] g[m,n]: a thread owns four columns and KC rows of dw; g is read coalesced (K / KC times), x scalar.
// The rows m are dealt to the workgroup's four waves (64 column quads per workgroup) and the four partial sums added in wave
// order through LDS: four times the workgroups and a quarter of the serial loop of the first version, whose 128 workgroups
// each walked all M rows (GENESIS' gated 'fc' layer, M = 224, K = 64, N = 32768: 106 us for 37 MB)
constexpr int kNN_KC = 16;
__global__ void __launch_bounds__(256)
matmul_nn_dw_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dw, int M, int N, int K) {
    __shared__ float4 red[3][kNN_KC][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.y * 64 + lane) * 4;
    const int k0 = blockIdx.x * kNN_KC;
    const bool n_ok = n0 < N;
    float4 acc[kNN_KC];
#pragma unroll
    for (int r = 0; r < kNN_KC; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_ok)
        for (int m = wave; m < M; m += 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(g + (size_t)m * N + n0);
            const float* xm = x + (size_t)m * K;
#pragma unroll
            for (int r = 0; r < kNN_KC; ++r) {
                const float xv = xm[k0 + r < K ? k0 + r : K - 1];
                acc[r].x = fmaf(xv, g4.x, acc[r].x); acc[r].y = fmaf(xv, g4.y, acc[r].y);
                acc[r].z = fmaf(xv, g4.z, acc[r].z); acc[r].w = fmaf(xv, g4.w, acc[r].w);
            }
        }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < kNN_KC; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && n_ok) {
#pragma unroll
        for (int r = 0; r < kNN_KC; ++r) {
            float4 v = acc[r];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 t = red[w][r][lane];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            if (k0 + r < K) *reinterpret_cast<float4*>(dw + (size_t)(k0 + r) * N + n0) = v;
        }
    }
}

// dx[M,K] = sum_n g[m,n] w[k,n]: the long sum over N is cut into `nchunk` contiguous column ranges (blockIdx.x); a block owns a
// 32 x 64 tile of (m, k) (blockIdx.y: row tile, blockIdx.z: k tile), stages 64 columns of g and w at a time in LDS (coalesced
// loads) and every thread accumulates a 2 x 4 register tile; partial[chunk][M][K] is summed by matmul_nn_dx_reduce_kernel in
// chunk order
__global__ void __launch_bounds__(256)
matmul_nn_dx_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ part, int M, int N, int K,
                    int cols_per_chunk) {
    __shared__ float gs[32][68], ws[64][68];
    const int t = threadIdx.x;
    const int m0 = blockIdx.y * 32, k0 = blockIdx.z * 64;
    const int c_begin = blockIdx.x * cols_per_chunk;
    const int c_end = c_begin + cols_per_chunk < N ? c_begin + cols_per_chunk : N;
    const int mi = (t >> 4) * 2, ki = (t & 15) * 4;
    float acc[2][4] = {};
    for (int c0 = c_begin; c0 < c_end; c0 += 64) {
        // stage: 32 x 64 of g, 64 x 64 of w (16 float4 per row)
        for (int e = t; e < 32 * 16; e += 256) {
            const int r = e >> 4, q = (e & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < M && c0 + q < c_end) v = *reinterpret_cast<const float4*>(g + (size_t)(m0 + r) * N + c0 + q);
            *reinterpret_cast<float4*>(&gs[r][q]) = v;
        }
        for (int e = t; e < 64 * 16; e += 256) {
            const int r = e >> 4, q = (e & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + r < K && c0 + q < c_end) v = *reinterpret_cast<const float4*>(w + (size_t)(k0 + r) * N + c0 + q);
            *reinterpret_cast<float4*>(&ws[r][q]) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int c = 0; c < 64; c += 4) {
            float4 a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(&gs[mi + i][c]);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float4*>(&ws[ki + j][c]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = fmaf(a[i].x, b[j].x, fmaf(a[i].y, b[j].y, fmaf(a[i].z, b[j].z, fmaf(a[i].w, b[j].w, acc[i][j]))));
        }
        __syncthreads();
    }
    float* o = part + (size_t)blockIdx.x * M * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + mi + i < M && k0 + ki + j < K) o[(size_t)(m0 + mi + i) * K + k0 + ki + j] = acc[i][j];
}

__global__ void __launch_bounds__(256)
matmul_nn_dx_reduce_kernel(const float* __restrict__ part, float* __restrict__ dx, int MK, int nchunk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= MK) return;
    float s0 = 0.f, s1 = 0.f;
    int c = 0;
    for (; c + 1 < nchunk; c += 2) { s0 += part[(size_t)c * MK + i]; s1 += part[(size_t)(c + 1) * MK + i]; }
    if (c < nchunk) s0 += part[(size_t)c * MK + i];
    dx[i] = s0 + s1;
}

inline int matmul_nn_chunks(int N) {
    const int tiles = gx_ceil_div(N, 64);
    return tiles < 128 ? tiles : 128;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int dense_threads(int Kc) { return Kc >= 1024 ? 1024 : 256; }

}  // namespace

extern "C" {

int gx_linear_fwd_ld(const float* x, int ldx, const float* w, const float* b, int act, float* y, int ldy, int M,
                     int N, int K, gx_stream_t stream) {
    GX_CHECK_ARG(x && w && y, "gx_linear_fwd: null pointer");
    GX_CHECK_ARG(M > 0 && N > 0 && K > 0, "gx_linear_fwd: bad M/N/K (%d,%d,%d)", M, N, K);
    GX_CHECK_ARG(ldx >= K && ldy >= N, "gx_linear_fwd: row strides (%d,%d) shorter than the rows (%d,%d)", ldx, ldy, K, N);
    GX_CHECK_ARG(act >= 0 && act <= 2, "gx_linear_fwd: act must be 0 (none), 1 (ReLU) or 2 (ELU)");
    if (act == 2) act = 3;      // dense_tile's epilogue codes: 2 = accumulate, 3 = ELU
    hipStream_t s = (hipStream_t)stream;
    const int vec_w = (K % 4 == 0) && aligned16(w);
    const int vec_x = vec_w && (ldx % 4 == 0) && aligned16(x);
    {
        GxProf pf(KID_DENSE, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
        hipLaunchKernelGGL((dense_kernel<true, true, false, false>), dim3(gx_ceil_div(N, 16), gx_ceil_div(M, 16)),
                           dim3(dense_threads(K)), 0, s, x, ldx, (const float*)nullptr, w, K, b, act, y, ldy, M, N, K,
                           (float*)nullptr, vec_x, vec_w, (float*)nullptr);
    }
    GX_CHECK_LAUNCH("gx_linear_fwd");
    return GX_OK;
}

int gx_linear_fwd(const float* x, const float* w, const float* b, int act, float* y, int M, int N, int K,
                  gx_stream_t stream) {
    return gx_linear_fwd_ld(x, K, w, b, act, y, N, M, N, K, stream);
}

int gx_linear_bwd_ex(const float* x, int ldx, const float* w, const float* y, const float* g, int ldg, int act,
                     float* dx, int lddx, int dx_accumulate, float* dw, float* db, float* db2, int M, int N, int K,
                     gx_stream_t stream) {
    GX_CHECK_ARG(x && w && g, "gx_linear_bwd: null pointer");
    GX_CHECK_ARG(M > 0 && N > 0 && K > 0, "gx_linear_bwd: bad M/N/K (%d,%d,%d)", M, N, K);
    GX_CHECK_ARG(ldx >= K && ldg >= N && (!dx || lddx >= K), "gx_linear_bwd: a row stride is shorter than its rows");
    GX_CHECK_ARG(act == 0 || ((act == 1 || act == 2) && y), "gx_linear_bwd: act 1 (ReLU) / 2 (ELU) needs the layer output y");
    GX_CHECK_ARG(dw || !db, "gx_linear_bwd: db is produced together with dw");
    GX_CHECK_ARG(db || !db2, "gx_linear_bwd: db2 is a second copy of db");
    hipStream_t s = (hipStream_t)stream;
    const int vec = (N % 4 == 0) && (ldg % 4 == 0) && aligned16(g) && (act == 0 || aligned16(y));
    GX_CHECK_ARG(dx_accumulate >= 0 && dx_accumulate <= 3, "gx_linear_bwd: dx_accumulate is a bit mask (1: dx, 2: dw / db)");
    const int elu = act == 2 ? 4 : 0;          // (bit 2 of the tile's act code: the mask is an ELU output)
    const int dwacc = (dx_accumulate & 2) ? 10 : 0;    // dw += (epilogue code 2), db += (bit 3)
    const int dx_act = ((dx_accumulate & 1) ? 2 : 0) | elu | ((dx_accumulate & 2) ? 16 : 0);
    if (act == 2) act = 1;                     // masked like ReLU from here on
    if (dx && dw) {   // one launch for both
        GxProf pf(KID_DENSE, s, 4.0 * M * N * K, 4.0 * (4.0 * M * N + 2.0 * N * K + 2.0 * M * K));
        const int gy_dx = gx_ceil_div(M, 16);
        const dim3 grid(gx_ceil_div(K, 16), gy_dx + gx_ceil_div(N, 16)), block(dense_threads(M > N ? M : N));
        const float* mask = act == 1 ? y : nullptr;
#define GX_PAIR(MASK, ROWSUM)                                                                                        \
        hipLaunchKernelGGL((dense_bwd_pair_kernel<MASK, ROWSUM>), grid, block, 0, s, g, ldg, mask, w, x, ldx, dx,    \
                           lddx, dx_act, dw, db, db2, M, N, K, gy_dx, vec)
        if (act == 1) { if (db) GX_PAIR(true, true); else GX_PAIR(true, false); }
        else          { if (db) GX_PAIR(false, true); else GX_PAIR(false, false); }
#undef GX_PAIR
        GX_CHECK_LAUNCH("gx_linear_bwd(dx+dw)");
        return GX_OK;
    }
    if (dx) {   // dx[M,K] = dpre[M,N] w[N,K]
        GxProf pf(KID_DENSE, s, 2.0 * M * N * K, 4.0 * (2.0 * M * N + (double)N * K + (double)M * K));
        const dim3 grid(gx_ceil_div(K, 16), gx_ceil_div(M, 16)), block(dense_threads(N));
        if (act == 1)
            hipLaunchKernelGGL((dense_kernel<true, false, true, false>), grid, block, 0, s, g, ldg, y, w, K,
                               (const float*)nullptr, dx_act & 7, dx, lddx, M, K, N, (float*)nullptr, vec, 0, (float*)nullptr);
        else
            hipLaunchKernelGGL((dense_kernel<true, false, false, false>), grid, block, 0, s, g, ldg,
                               (const float*)nullptr, w, K, (const float*)nullptr, dx_act & 7, dx, lddx, M, K, N,
                               (float*)nullptr, vec, 0, (float*)nullptr);
    }
    GX_CHECK_LAUNCH("gx_linear_bwd(dx)");
    if (dw) {   // dw[N,K] = dpre^T[N,M] x[M,K];  db[N] = row sums of dpre^T
        GxProf pf(KID_DENSE, s, 2.0 * M * N * K, 4.0 * (2.0 * M * N + (double)M * K + (double)N * K));
        const dim3 grid(gx_ceil_div(K, 16), gx_ceil_div(N, 16)), block(dense_threads(M));
        if (act == 1) {
            if (db)
                hipLaunchKernelGGL((dense_kernel<false, false, true, true>), grid, block, 0, s, g, ldg, y, x, ldx,
                                   (const float*)nullptr, elu | dwacc, dw, K, N, K, M, db, 0, 0, db2);
            else
                hipLaunchKernelGGL((dense_kernel<false, false, true, false>), grid, block, 0, s, g, ldg, y, x, ldx,
                                   (const float*)nullptr, elu | dwacc, dw, K, N, K, M, (float*)nullptr, 0, 0, (float*)nullptr);
        } else {
            if (db)
                hipLaunchKernelGGL((dense_kernel<false, false, false, true>), grid, block, 0, s, g, ldg,
                                   (const float*)nullptr, x, ldx, (const float*)nullptr, dwacc, dw, K, N, K, M, db, 0, 0,
                                   db2);
            else
                hipLaunchKernelGGL((dense_kernel<false, false, false, false>), grid, block, 0, s, g, ldg,
                                   (const float*)nullptr, x, ldx, (const float*)nullptr, dwacc, dw, K, N, K, M,
                                   (float*)nullptr, 0, 0, (float*)nullptr);
        }
    }
    GX_CHECK_LAUNCH("gx_linear_bwd(dw)");
    return GX_OK;
}

int gx_linear_bwd(const float* x, const float* w, const float* y, const float* g, int act, float* dx, float* dw,
                  float* db, int M, int N, int K, gx_stream_t stream) {
    return gx_linear_bwd_ex(x, K, w, y, g, N, act, dx, K, 0, dw, db, nullptr, M, N, K, stream);
}

int gx_matmul_nn_fwd(const float* x, const float* w, float* y, int M, int N, int K, gx_stream_t stream) {
    GX_CHECK_ARG(x && w && y, "gx_matmul_nn_fwd: null pointer");
    GX_CHECK_ARG(M > 0 && N > 0 && K > 0 && (N % 4) == 0, "gx_matmul_nn_fwd: bad M/N/K (%d,%d,%d; N %% 4 == 0)", M, N, K);
    GX_CHECK_ARG(aligned16(w) && aligned16(y), "gx_matmul_nn_fwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DENSE, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
        hipLaunchKernelGGL(matmul_nn_fwd_kernel, dim3(gx_ceil_div(M, kNN_MR), gx_ceil_div(N, 1024)), dim3(256), 0, s, x, w, y, M,
                           N, K);
    }
    GX_CHECK_LAUNCH("gx_matmul_nn_fwd");
    return GX_OK;
}

size_t gx_matmul_nn_bwd_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)matmul_nn_chunks(N) * M * K * sizeof(float);
}

int gx_matmul_nn_bwd(const float* x, const float* w, const float* g, float* dx, float* dw, int M, int N, int K, void* ws,
                     size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(x && w && g, "gx_matmul_nn_bwd: null pointer");
    GX_CHECK_ARG(M > 0 && N > 0 && K > 0 && (N % 4) == 0, "gx_matmul_nn_bwd: bad M/N/K (%d,%d,%d; N %% 4 == 0)", M, N, K);
    GX_CHECK_ARG(aligned16(w) && aligned16(g) && (!dw || aligned16(dw)), "gx_matmul_nn_bwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    if (dw) {
        GxProf pf(KID_DENSE, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
        hipLaunchKernelGGL(matmul_nn_dw_kernel, dim3(gx_ceil_div(K, kNN_KC), gx_ceil_div(N, 256)), dim3(256), 0, s, x, g, dw, M,
                           N, K);
        GX_CHECK_LAUNCH("gx_matmul_nn_bwd(dw)");
    }
    if (dx) {
        GX_CHECK_ARG(ws && ws_bytes >= gx_matmul_nn_bwd_ws_bytes(M, N, K), "gx_matmul_nn_bwd: workspace too small");
        const int nchunk = matmul_nn_chunks(N);
        const int cols = gx_round_up(gx_ceil_div(N, nchunk), 64);
        const int used = gx_ceil_div(N, cols);
        {
            GxProf pf(KID_DENSE, s, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
            hipLaunchKernelGGL(matmul_nn_dx_kernel, dim3(used, gx_ceil_div(M, 32), gx_ceil_div(K, 64)), dim3(256), 0, s, g, w,
                               (float*)ws, M, N, K, cols);
        }
        GX_CHECK_LAUNCH("gx_matmul_nn_bwd(dx)");
        {
            GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * (used + 1.0) * M * K);
            hipLaunchKernelGGL(matmul_nn_dx_reduce_kernel, dim3(gx_ceil_div(M * K, 256)), dim3(256), 0, s, (const float*)ws, dx,
                               M * K, used);
        }
        GX_CHECK_LAUNCH("gx_matmul_nn_bwd(dx reduce)");
    }
    return GX_OK;
}

int gx_lstm_step_fwd(const float* gx, const float* h_prev, const float* c_prev, const float* w_hh,
                     const float* b_hh, int B, int H, float* act, float* c, float* h, gx_stream_t stream) {
    GX_CHECK_ARG(gx && w_hh && b_hh && act && c && h, "gx_lstm_step_fwd: null pointer");
    GX_CHECK_ARG(B > 0 && H > 0 && H % 16 == 0, "gx_lstm_step_fwd: H must be a multiple of 16 (B=%d, H=%d)", B, H);
    GX_CHECK_ARG((h_prev == nullptr) == (c_prev == nullptr), "gx_lstm_step_fwd: h_prev and c_prev go together");
    GX_CHECK_ARG(aligned16(w_hh) && (!h_prev || aligned16(h_prev)), "gx_lstm_step_fwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DENSE, s, h_prev ? 8.0 * B * H * H : 0.0, 4.0 * (4.0 * H * H + 12.0 * B * H));
        hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(H / 16, gx_ceil_div(B, 16)), dim3(256), 0, s, gx, h_prev,
                           c_prev, w_hh, b_hh, B, H, act, c, h);
    }
    GX_CHECK_LAUNCH("gx_lstm_step_fwd");
    return GX_OK;
}

int gx_lstm_step_bwd(const float* g_h, const float* dgates_next, const float* w_hh, const float* act,
                     const float* c, const float* c_prev, const float* dc_next, int B, int H, float* dgates,
                     float* dc_prev, gx_stream_t stream) {
    GX_CHECK_ARG(w_hh && act && c && dgates && dc_prev, "gx_lstm_step_bwd: null pointer");
    GX_CHECK_ARG(g_h || dgates_next, "gx_lstm_step_bwd: no incoming gradient");
    GX_CHECK_ARG(B > 0 && H > 0 && H % 16 == 0, "gx_lstm_step_bwd: H must be a multiple of 16 (B=%d, H=%d)", B, H);
    GX_CHECK_ARG(!dgates_next || aligned16(dgates_next), "gx_lstm_step_bwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DENSE, s, dgates_next ? 8.0 * B * H * H : 0.0, 4.0 * (4.0 * H * H + 14.0 * B * H));
        hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3(H / 16, gx_ceil_div(B, 16)), dim3(1024), 0, s, g_h,
                           dgates_next, w_hh, act, c, c_prev, dc_next, B, H, dgates, dc_prev);
    }
    GX_CHECK_LAUNCH("gx_lstm_step_bwd");
    return GX_OK;
}

int gx_lstm_seq_max_steps(int B, int H) {
    if (B <= 0 || H <= 0 || H % 16 != 0) return 0;
    return (H / 16) * gx_ceil_div(B, 16) <= kLstmSeqMaxWg ? kLstmSeqBar : 0;
}

size_t gx_lstm_seq_ws_bytes(void) { return kLstmSeqBar * sizeof(unsigned); }

int gx_lstm_seq_fwd(const float* gx, const float* w_hh, const float* b_hh, int T, int B, int H, float* act, float* c,
                    float* h, void* bar, gx_stream_t stream) {
    GX_CHECK_ARG(gx && w_hh && b_hh && act && c && h && bar, "gx_lstm_seq_fwd: null pointer");
    GX_CHECK_ARG(T > 0 && T <= gx_lstm_seq_max_steps(B, H),
                 "gx_lstm_seq_fwd: T=%d steps of B=%d, H=%d do not fit one launch (gx_lstm_seq_max_steps: %d)", T, B, H,
                 gx_lstm_seq_max_steps(B, H));
    GX_CHECK_ARG(aligned16(w_hh) && aligned16(h) && ((size_t)B * H) % 4 == 0, "gx_lstm_seq_fwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DENSE, s, 8.0 * (T - 1) * B * H * H, 4.0 * T * (4.0 * H * H + 12.0 * B * H));
        hipLaunchKernelGGL(lstm_seq_fwd_kernel, dim3(H / 16, gx_ceil_div(B, 16)), dim3(256), 0, s, gx, w_hh, b_hh, T, B, H,
                           act, c, h, (unsigned*)bar);
    }
    GX_CHECK_LAUNCH("gx_lstm_seq_fwd");
    return GX_OK;
}

int gx_lstm_seq_bwd(const float* g_h, const float* w_hh, const float* act, const float* c, int T, int B, int H,
                    float* dgates, float* dc2, void* bar, gx_stream_t stream) {
    GX_CHECK_ARG(g_h && w_hh && act && c && dgates && dc2 && bar, "gx_lstm_seq_bwd: null pointer");
    GX_CHECK_ARG(T > 0 && T <= gx_lstm_seq_max_steps(B, H),
                 "gx_lstm_seq_bwd: T=%d steps of B=%d, H=%d do not fit one launch (gx_lstm_seq_max_steps: %d)", T, B, H,
                 gx_lstm_seq_max_steps(B, H));
    GX_CHECK_ARG(aligned16(dgates), "gx_lstm_seq_bwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_DENSE, s, 8.0 * (T - 1) * B * H * H, 4.0 * T * (4.0 * H * H + 14.0 * B * H));
        hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3(H / 16, gx_ceil_div(B, 16)), dim3(1024), 0, s, g_h, w_hh, act, c, T, B, H,
                           dgates, dc2, (unsigned*)bar);
    }
    GX_CHECK_LAUNCH("gx_lstm_seq_bwd");
    return GX_OK;
}

}  // extern "C"
