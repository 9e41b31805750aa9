// Probe for the next round (DESIGN.md section 7, next targets): fp32 products emulated on the bf16 matrix pipe.
//   a = a_hi + a_mid + a_lo (three bf16 pieces hold all 24 mantissa bits), a*b ~ the six products of order <= 2
//   (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// Measures (1) accuracy of C = A B (64 x 64, K = 1024) against fp64 for: fp32 MFMA, bf16 x1, x3, x6; (2) the rate of a
// register-operand MFMA loop, in "fp32-equivalent" TFLOP/s (useful flops = one fp32 product per six bf16 products).
// build: hipcc --offload-arch=gfx950 -O3 tools/bf16x6_probe.hip -o tools/abl/bf16x6_probe ; run under `timeout` on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// one workgroup of 4 waves: wave (wm, wn) computes the 32 x 32 block of C = A[64][K] * B[K][64] (B given as Bt[64][K])
// mode 0: fp32 MFMA 32x32x2; 1: bf16 (hi only); 3: hi*hi + hi*mid + mid*hi; 6: all six terms
__global__ void __launch_bounds__(256) gemm_probe(const float* A, const float* Bt, float* C, int K, int mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, kh = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* a = A + (size_t)(wm * 32 + r) * K;
    const float* b = Bt + (size_t)(wn * 32 + r) * K;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + kh], b[k + kh], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                __bf16 h, m, l;
                split3(a[k + 8 * kh + j], h, m, l); ah[j] = h; am[j] = m; al[j] = l;
                split3(b[k + 8 * kh + j], h, m, l); bh[j] = h; bm[j] = m; bl[j] = l;
            }
            // small terms first
            if (mode >= 6) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            }
            if (mode >= 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    for (int i = 0; i < 16; ++i) {
        const int row = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh, col = wn * 32 + r;
        C[row * 64 + col] = acc[i];
    }
}

// rate: register operands, 4 independent accumulators per wave, `terms` bf16 MFMAs per fp32-equivalent k16 step
template <int TERMS>
__global__ void __launch_bounds__(256) rate_bf16(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 a[3], b[3];
    for (int q = 0; q < 3; ++q) for (int i = 0; i < 8; ++i) { a[q][i] = (__bf16)(0.001f * (threadIdx.x + q + i)); b[q][i] = (__bf16)(0.002f * (threadIdx.x + 2 * q + i)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < TERMS; ++t)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3], b[(t + j) % 3], acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) rate_f32(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const float a = 0.001f * threadIdx.x, b = 0.002f * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + t, b - t, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int K = 1024;
    std::vector<float> A(64 * K), Bt(64 * K), C(64 * 64);
    srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : Bt) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
    std::vector<double> ref(64 * 64);
    double refmax = 0.0;
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)Bt[j * K + k];
        ref[i * 64 + j] = s; refmax = fmax(refmax, fabs(s));
    }
    float *dA, *dB, *dC, *dO;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, Bt.size() * 4); (void)hipMalloc(&dC, C.size() * 4);
    (void)hipMalloc(&dO, 4096 * 256 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
    for (int mode : {0, 1, 3, 6}) {
        gemm_probe<<<1, 256>>>(dA, dB, dC, K, mode);
        (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0.0, e2 = 0.0, r2 = 0.0;
        for (int i = 0; i < 64 * 64; ++i) { const double e = C[i] - ref[i]; emax = fmax(emax, fabs(e)); e2 += e * e; r2 += ref[i] * ref[i]; }
        printf("accuracy  %-10s max |err| / max |ref| %.3e   relative L2 %.3e\n",
               mode == 0 ? "fp32 MFMA" : (mode == 1 ? "bf16 x1" : (mode == 3 ? "bf16 x3" : "bf16 x6")), emax / refmax, sqrt(e2 / r2));
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, grid = 2048;
    auto timeit = [&](int which) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (which == 0) rate_f32<<<grid, 256>>>(dO, iters);
            else if (which == 1) rate_bf16<1><<<grid, 256>>>(dO, iters);
            else if (which == 3) rate_bf16<3><<<grid, 256>>>(dO, iters);
            else rate_bf16<6><<<grid, 256>>>(dO, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return (double)ms;
    };
    {   // fp32: 8 MFMAs 32x32x2 = one k16 step per accumulator
        const double ms = timeit(0);
        const double flops = (double)grid * 4 /*waves*/ * iters * 4 /*acc*/ * 8 * 4096.0;
        printf("rate      fp32 MFMA   %.1f TFLOP/s\n", flops / ms / 1e9);
    }
    for (int terms : {1, 3, 6}) {
        const double ms = timeit(terms);
        const double useful = (double)grid * 4 * iters * 4 * 32768.0;          // one k16 step of fp32-equivalent products
        printf("rate      bf16 x%d     %.1f fp32-equivalent TFLOP/s  (%.1f TFLOP/s on the bf16 pipe)\n", terms, useful / ms / 1e9,
               useful * terms / ms / 1e9);
    }
    return 0;
}
