// Which packed-fp32 instruction forms return wrong results when OTHER work shares the GPU?  (DESIGN.md finding 48.)
// Every thread evaluates, 512 times per launch, one packed instruction form on operands it (re)loads from global memory (or keeps in
// registers) and the same arithmetic with scalar instructions on the same registers, and counts the lane results that differ bit
// for bit.  Run it alone (all counts 0) and beside another process that keeps the GPU busy (tools/probe/run_pkfma_probe.sh).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o pkfma_probe pkfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

// OP 0: v_pk_fma_f32 (r = a * b + r), 1: v_pk_mul_f32 then scalar adds, 2: v_pk_add_f32 (r = a + b, accumulated by scalar adds)
// S0L / S1L: which half of source 0 / 1 the LOW lane takes; S0H / S1H: the HIGH lane
#define PROBE_KERNEL2(NAME, OP, S0L, S1L, S0H, S1H, S2L, S2H, SG, ASM)                                                                             \
template <bool FROM_MEM>                                                                                                            \
__global__ void __launch_bounds__(256) NAME(const f2* __restrict__ src, int n, int iters, unsigned long long* bad) {                \
    const int tid = blockIdx.x * 256 + threadIdx.x;                                                                                 \
    unsigned long long nb = 0;                                                                                                      \
    f2 acc = {0.f, 0.f};                                                                                                            \
    float s0 = 0.f, s1 = 0.f;                                                                                                       \
    f2 a = src[tid % n], b = src[(tid + 7) % n];                                                                                    \
    for (int it = 0; it < iters; ++it) {                                                                                            \
        if (FROM_MEM) {                                                                                                             \
            const f2* p = src + (size_t)((tid + it * 977) % n);                                                                     \
            const f2* q = src + (size_t)((tid * 3 + it * 131) % n);                                                                 \
            asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off\n\ts_waitcnt vmcnt(0)"                 \
                         : "=&v"(a), "=&v"(b) : "v"(p), "v"(q) : "memory");                                                         \
        } else {                                                                                                                    \
            a[0] = a[0] * 0.999f + 0.001f; b[1] = b[1] * 0.998f - 0.002f;                                                           \
            asm volatile("" : "+v"(a), "+v"(b));                                                                                    \
        }                                                                                                                           \
        unsigned long long a_s = 0;                                                                                                 \
        if (SG == 1) {         /* source 0 in a scalar register pair: the wave's first lane's values */                              \
            const unsigned lo = __builtin_amdgcn_readfirstlane(__float_as_uint(a[0])), hi = __builtin_amdgcn_readfirstlane(__float_as_uint(a[1])); \
            a[0] = __uint_as_float(lo); a[1] = __uint_as_float(hi);                                                                 \
            a_s = ((unsigned long long)hi << 32) | lo;                                                                              \
        }                                                                                                                           \
        if (SG == 2) {         /* source 1 in a scalar register pair */                                                              \
            const unsigned lo = __builtin_amdgcn_readfirstlane(__float_as_uint(b[0])), hi = __builtin_amdgcn_readfirstlane(__float_as_uint(b[1])); \
            b[0] = __uint_as_float(lo); b[1] = __uint_as_float(hi);                                                                 \
            a_s = ((unsigned long long)hi << 32) | lo;                                                                              \
        }                                                                                                                           \
        f2 r = acc;                                                                                                                 \
        float a_l, b_l, a_h, b_h;                                                                                                   \
        a_l = a[S0L]; b_l = b[S1L]; a_h = a[S0H]; b_h = b[S1H];                                                                     \
        const float c_l = S2L ? s1 : s0, c_h = S2H ? s1 : s0;                                                                       \
        asm volatile("" : "+v"(a_l), "+v"(b_l), "+v"(a_h), "+v"(b_h));      /* scalar copies: the reference must not be re-vectorised */ \
        if (OP == 0) {                                                                                                              \
            if (SG == 1) asm volatile(ASM : "+v"(r) : "s"(a_s), "v"(b));                                                            \
            else if (SG == 2) asm volatile(ASM : "+v"(r) : "v"(a), "s"(a_s));                                                       \
            else asm volatile(ASM : "+v"(r) : "v"(a), "v"(b));                                                                      \
            float n0 = __builtin_fmaf(a_l, b_l, c_l); asm volatile("" : "+v"(n0)); float n1 = __builtin_fmaf(a_h, b_h, c_h);        \
            asm volatile("" : "+v"(n1)); s0 = n0; s1 = n1;                                                                          \
        } else {                                                                                                                    \
            f2 m;                                                                                                                   \
            if (SG == 1) asm volatile(ASM : "=v"(m) : "s"(a_s), "v"(b));                                                            \
            else if (SG == 2) asm volatile(ASM : "=v"(m) : "v"(a), "s"(a_s));                                                       \
            else asm volatile(ASM : "=v"(m) : "v"(a), "v"(b));                                                                      \
            float m0 = m[0], m1 = m[1];                                                                                             \
            asm volatile("" : "+v"(m0), "+v"(m1));                                                                                  \
            r[0] = r[0] + m0; asm volatile("" : "+v"(r)); r[1] = r[1] + m1;                                                         \
            const float t0 = OP == 1 ? a_l * b_l : a_l + b_l; float t0v = t0; asm volatile("" : "+v"(t0v));                         \
            const float t1 = OP == 1 ? a_h * b_h : a_h + b_h; float t1v = t1; asm volatile("" : "+v"(t1v));                         \
            s0 = s0 + t0v; asm volatile("" : "+v"(s0)); s1 = s1 + t1v;                                                              \
        }                                                                                                                           \
        asm volatile("" : "+v"(s0), "+v"(s1));                                                                                      \
        const float r0 = r[0], r1 = r[1];                                                                                           \
        if (__float_as_uint(r0) != __float_as_uint(s0) || __float_as_uint(r1) != __float_as_uint(s1)) {                             \
            if (nb == 0 && atomicAdd(bad + 1, 1ull) == 0) {                                                                         \
                float* dbg = reinterpret_cast<float*>(bad + 2);                                                                     \
                dbg[0] = r0; dbg[1] = s0; dbg[2] = r1; dbg[3] = s1; dbg[4] = a[0]; dbg[5] = a[1]; dbg[6] = b[0]; dbg[7] = b[1];     \
                reinterpret_cast<int*>(dbg)[8] = it;                                                                                \
            }                                                                                                                       \
            if (__float_as_uint(r0) != __float_as_uint(s0)) atomicAdd(bad + 7, 1ull);      /* (how many of them in the LOW lane) */   \
            ++nb;                                                                                                                   \
            r[0] = s0; r[1] = s1;                                                                                                   \
        }                                                                                                                           \
        acc = r;                                                                                                                    \
        if ((it & 63) == 63) { acc[0] *= 0.5f; acc[1] *= 0.5f; s0 *= 0.5f; s1 *= 0.5f; }                                            \
    }                                                                                                                               \
    if (nb) atomicAdd(bad, nb);                                                                                                     \
}

#define PROBE_KERNEL(NAME, OP, S0L, S1L, S0H, S1H, ASM) PROBE_KERNEL2(NAME, OP, S0L, S1L, S0H, S1H, 0, 1, 0, ASM)
PROBE_KERNEL(k_fma_dflt, 0, 0, 0, 1, 1, "v_pk_fma_f32 %0, %1, %2, %0")
PROBE_KERNEL(k_fma_bcast, 0, 0, 0, 0, 1, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]")
PROBE_KERNEL(k_fma_01_00, 0, 0, 1, 0, 0, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1]")
PROBE_KERNEL(k_fma_11_00, 0, 1, 1, 0, 0, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,0,1]")
PROBE_KERNEL(k_fma_01_11, 0, 0, 1, 1, 1, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]")
PROBE_KERNEL(k_fma_10_11, 0, 1, 0, 1, 1, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]")
PROBE_KERNEL(k_fma_10_01, 0, 1, 0, 0, 1, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]")
PROBE_KERNEL(k_fma_00_10, 0, 0, 0, 1, 0, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]")
PROBE_KERNEL(k_mul_01_10, 1, 0, 1, 1, 0, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")
PROBE_KERNEL(k_mul_dflt, 1, 0, 0, 1, 1, "v_pk_mul_f32 %0, %1, %2")
PROBE_KERNEL(k_add_01_10, 2, 0, 1, 1, 0, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")
PROBE_KERNEL(k_add_01_11, 2, 0, 1, 1, 1, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]")

// third-source selections and a scalar register pair as source 0 (forms of conv1x1_fwd_vec_kernel)
PROBE_KERNEL2(k_fma_h_101, 0, 0, 0, 1, 0, 0, 1, 0, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]")
PROBE_KERNEL2(k_fma_h_010, 0, 0, 0, 0, 1, 0, 0, 0, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]")
PROBE_KERNEL2(k_fma_h_110, 0, 0, 0, 1, 1, 0, 0, 0, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]")
PROBE_KERNEL2(k_fma_l_001, 0, 0, 0, 1, 1, 1, 1, 0, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1]")
PROBE_KERNEL2(k_fma_l_101, 0, 1, 0, 1, 1, 1, 1, 0, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,1]")
PROBE_KERNEL2(k_fma_l_100h110, 0, 1, 0, 1, 1, 0, 0, 0, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,0]")
PROBE_KERNEL2(k_mul_s_bcast, 1, 0, 0, 0, 1, 0, 1, 1, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")
PROBE_KERNEL2(k_mul_s_dflt, 1, 0, 0, 1, 1, 0, 1, 1, "v_pk_mul_f32 %0, %1, %2")
PROBE_KERNEL2(k_fma_s_h010, 0, 0, 0, 0, 1, 0, 0, 1, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]")
PROBE_KERNEL2(k_fma_s_bcast, 0, 0, 0, 0, 1, 0, 1, 1, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]")

PROBE_KERNEL2(k_fma_s_l101, 0, 1, 0, 1, 1, 1, 1, 1, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,1]")
PROBE_KERNEL2(k_fma_vs_l011, 0, 0, 1, 1, 1, 1, 1, 2, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,1]")
PROBE_KERNEL2(k_mul_vs_l01, 1, 0, 1, 1, 1, 0, 1, 2, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]")
PROBE_KERNEL2(k_mul_sv_l10, 1, 1, 0, 1, 1, 0, 1, 1, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]")
PROBE_KERNEL2(k_add_vs_h10, 2, 0, 0, 1, 0, 0, 1, 2, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]")

template <typename K>
static void run(const char* name, K kern, bool from_mem, const f2* src, int n, unsigned long long* d_bad, double seconds) {
    (void)hipMemset(d_bad, 0, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double spent = 0.0;
    long launches = 0;
    while (spent < seconds) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 8; ++i) { hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, src, n, 512, d_bad); ++launches; }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); spent += ms * 1e-3;
    }
    unsigned long long h[8] = {0};
    (void)hipMemcpy(h, d_bad, 64, hipMemcpyDeviceToHost);
    printf("%-58s %s: %11llu wrong results (%llu in the low lane) in %.3g evaluations\n", name, from_mem ? "operands loaded  " : "operands in regs ", h[0], h[7],
           (double)launches * 2048 * 256 * 512);
    if (h[0]) {
        const float* dbg = reinterpret_cast<const float*>(h + 2);
        printf("      first: packed (%.9g, %.9g) scalar (%.9g, %.9g), a (%.9g, %.9g) b (%.9g, %.9g), iteration %d\n", dbg[0], dbg[2], dbg[1], dbg[3],
               dbg[4], dbg[5], dbg[6], dbg[7], reinterpret_cast<const int*>(dbg)[8]);
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int n = 1 << 20;
    std::vector<f2> h(n);
    srand(5);
    for (int i = 0; i < n; ++i) { h[i][0] = (rand() % 2001 - 1000) * 1e-3f; h[i][1] = (rand() % 2001 - 1000) * 1e-3f; }
    f2* d; unsigned long long* d_bad;
    (void)hipMalloc((void**)&d, n * sizeof(f2)); (void)hipMalloc((void**)&d_bad, 64);
    (void)hipMemcpy(d, h.data(), n * sizeof(f2), hipMemcpyHostToDevice);
#define RUN(K, NAME) run(NAME, K<true>, true, d, n, d_bad, seconds); run(NAME, K<false>, false, d, n, d_bad, seconds);
    const bool only_scalar_forms = argc > 2;
    if (!only_scalar_forms) {
    RUN(k_fma_dflt,  "v_pk_fma_f32  lo = a.lo b.lo, hi = a.hi b.hi (no op_sel)")
    RUN(k_fma_bcast, "v_pk_fma_f32  lo = a.lo b.lo, hi = a.lo b.hi")
    RUN(k_fma_01_00, "v_pk_fma_f32  lo = a.lo b.HI, hi = a.lo b.lo")
    RUN(k_fma_11_00, "v_pk_fma_f32  lo = a.HI b.HI, hi = a.lo b.lo")
    RUN(k_fma_01_11, "v_pk_fma_f32  lo = a.lo b.HI, hi = a.hi b.hi")
    RUN(k_fma_10_11, "v_pk_fma_f32  lo = a.HI b.lo, hi = a.hi b.hi")
    RUN(k_fma_10_01, "v_pk_fma_f32  lo = a.HI b.lo, hi = a.lo b.hi")
    RUN(k_fma_00_10, "v_pk_fma_f32  lo = a.lo b.lo, hi = a.hi b.LO")
    RUN(k_mul_dflt,  "v_pk_mul_f32  lo = a.lo b.lo, hi = a.hi b.hi (no op_sel)")
    RUN(k_mul_01_10, "v_pk_mul_f32  lo = a.lo b.HI, hi = a.hi b.LO")
    RUN(k_add_01_10, "v_pk_add_f32  lo = a.lo + b.HI, hi = a.hi + b.LO")
    RUN(k_add_01_11, "v_pk_add_f32  lo = a.lo + b.HI, hi = a.hi + b.hi")
    RUN(k_fma_h_101, "v_pk_fma_f32  op_sel_hi:[1,0,1]: hi = a.hi b.LO + c.hi")
    RUN(k_fma_h_010, "v_pk_fma_f32  op_sel_hi:[0,1,0]: hi = a.LO b.hi + c.LO")
    RUN(k_fma_h_110, "v_pk_fma_f32  op_sel_hi:[1,1,0]: hi = a.hi b.hi + c.LO")
    RUN(k_fma_l_001, "v_pk_fma_f32  op_sel:[0,0,1]: lo = a.lo b.lo + c.HI")
    RUN(k_fma_l_101, "v_pk_fma_f32  op_sel:[1,0,1]: lo = a.HI b.lo + c.HI")
    RUN(k_fma_l_100h110, "v_pk_fma_f32  op_sel:[1,0,0] op_sel_hi:[1,1,0]")
    RUN(k_mul_s_dflt, "v_pk_mul_f32  source 0 = s[n:n+1] (no op_sel)")
    RUN(k_mul_s_bcast, "v_pk_mul_f32  source 0 = s[n:n+1] op_sel_hi:[0,1]")
    RUN(k_fma_s_bcast, "v_pk_fma_f32  source 0 = s[n:n+1] op_sel_hi:[0,1,1]")
    RUN(k_fma_s_h010, "v_pk_fma_f32  source 0 = s[n:n+1] op_sel_hi:[0,1,0]")
    }
    RUN(k_fma_s_l101, "v_pk_fma_f32  source 0 = s[n:n+1] op_sel:[1,0,1] (lo = s.HI b.lo + c.HI)")
    RUN(k_fma_vs_l011, "v_pk_fma_f32  source 1 = s[n:n+1] op_sel:[0,1,1] (lo = a.lo s.HI + c.HI)")
    RUN(k_mul_vs_l01, "v_pk_mul_f32  source 1 = s[n:n+1] op_sel:[0,1] (lo = a.lo s.HI)")
    RUN(k_mul_sv_l10, "v_pk_mul_f32  source 0 = s[n:n+1] op_sel:[1,0] (lo = s.HI b.lo)")
    RUN(k_add_vs_h10, "v_pk_add_f32  source 1 = s[n:n+1] op_sel_hi:[1,0] (hi = a.hi + s.LO)")
    return 0;
}
