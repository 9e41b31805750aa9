// microbenchmark: cost of a hand-rolled grid barrier + cross-workgroup exchange of a 32 KB state on gfx950 (finding 11).
// build: hipcc --offload-arch=gfx950 -O3 tools/gridbar_probe.hip -o tools/abl/gridbar_probe ; run on the GPU box under `timeout`
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int V> __device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned G, unsigned& gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (!(V & 1)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        unsigned old = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == G - 1) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spins = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                if (!(V & 2)) __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000) { ok = false; break; }
            }
        }
        if (!(V & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    gen++;
    __syncthreads();
    return ok;
}
// each WG owns a slice of the state; every round: read the whole state (32 KB), write own slice = f(sum)
template <int V> __global__ void __launch_bounds__(256) k(float* st0, float* st1, unsigned* bar, int G, int rounds, int nstate, int* err) {
    int bid = blockIdx.x; if (V & 4) { if (bid & 7) return; bid >>= 3; }
    unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* a = st0; float* b = st1;
    const int per = nstate / G;
    for (int r = 0; r < rounds; ++r) {
        float s = 0.f;
        for (int i = threadIdx.x; i < nstate; i += 256) s += __builtin_nontemporal_load(a + i) * 0.f + a[i];
        // wave reduce
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        __shared__ float ws[4];
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
        __syncthreads();
        s = ws[0] + ws[1] + ws[2] + ws[3];
        for (int i = threadIdx.x; i < per; i += 256) b[bid * per + i] = s * (1.0f / nstate) + 1.0f;
        if (!grid_barrier<V>(bar, G, gen)) { if (threadIdx.x == 0) *err = 1; return; }
        float* t = a; a = b; b = t;
    }
}
int main(int argc, char** argv) {
    int nstate = 8192;
    float *s0, *s1; unsigned* bar; int* err;
    hipMalloc(&s0, nstate * 4); hipMalloc(&s1, nstate * 4); hipMalloc(&bar, 8); hipMalloc(&err, 4);
    hipMemset(bar, 0, 8); hipMemset(err, 0, 4);
    std::vector<float> h(nstate, 0.f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int V = 0; V < 8; ++V) for (int G : {32}) for (int rounds : {1, 41}) {
        hipMemcpy(s0, h.data(), nstate * 4, hipMemcpyHostToDevice);
        for (int w = 0; w < 3; ++w) do { int grid = (V & 4) ? G * 8 : G; switch (V) { case 0: k<0><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 1: k<1><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 2: k<2><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 3: k<3><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 4: k<4><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 5: k<5><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 6: k<6><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; default: k<7><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); } } while (0);
        hipDeviceSynchronize();
        hipMemcpy(s0, h.data(), nstate * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        for (int w = 0; w < 20; ++w) do { int grid = (V & 4) ? G * 8 : G; switch (V) { case 0: k<0><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 1: k<1><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 2: k<2><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 3: k<3><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 4: k<4><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 5: k<5><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; case 6: k<6><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); break; default: k<7><<<grid,256>>>(s0,s1,bar,G,rounds,nstate,err); } } while (0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int he; hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost);
        float v; hipMemcpy(&v, (rounds * 20) % 2 ? s1 : s0, 4, hipMemcpyDeviceToHost);
        printf("V %d (1=nofence 2=nosleep 4=oneXCD) G %3d rounds %2d: %.2f us per launch  (err %d, value %.1f expect %.1f)\n", V, G, rounds, ms * 1000 / 20, he, v, (float)(rounds * 20));
    }
    return 0;
}
