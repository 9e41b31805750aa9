// Store-pattern probe: how fast can 256 persistent workgroups write an [N, 32, 72, 72] fp32 tensor in tiles of SEG consecutive floats
// per channel plane (the epilogue pattern of the canvas convs: SEG = 216 = three 72-pixel rows -> 864-byte segments that start and end
// inside 128-byte lines), against SEG = 256 (whole lines) on a [N, 32, 72*72 rounded] tensor, and a plain linear fill.
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/abl/store_probe && tools/abl/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// mode 0: lane = pixel of the segment, 16 stores per lane to 16 channel planes (2 lane halves -> channels +4): the MFMA C/D layout
template <int SEG>
__global__ void __launch_bounds__(256) tile_store(float* out, int ntiles, int tiles_per_img, int plane, int per, int contiguous) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = 0; i < per; ++i) {
        const int tile = contiguous ? blockIdx.x * per + i : blockIdx.x + i * gridDim.x;
        if (tile >= ntiles) return;
        const int img = tile / tiles_per_img, t = tile % tiles_per_img;
        for (int nj = 0; nj < 2; ++nj) {
            const int p = wave * 64 + nj * 32 + (lane & 31);
            if (p >= SEG) continue;
            float* o = out + ((size_t)img * 32 + 4 * (lane >> 5)) * plane + (size_t)t * SEG + p;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) o[(size_t)((reg & 3) + 8 * (reg >> 2)) * plane] = (float)(reg + p);
        }
    }
}
__global__ void linear_fill(float4* out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const int N = 224;
    float* buf;
    const size_t bytes = (size_t)N * 32 * 5632 * 4;      // room for both layouts
    hipMalloc(&buf, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto fn, double mb) {
        for (int i = 0; i < 3; ++i) fn();
        hipEventRecord(a);
        for (int i = 0; i < 20; ++i) fn();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-62s %8.1f us  %6.2f TB/s\n", name, ms * 50.0, mb / (ms / 20.0) * 1e-6);
    };
    const double mb216 = (double)N * 32 * 5184 * 4 / 1e6, mb256 = (double)N * 32 * 5632 * 4 / 1e6;
    run("linear float4 fill", [&] { hipLaunchKernelGGL(linear_fill, dim3(2048), dim3(256), 0, 0, (float4*)buf, (size_t)N * 32 * 5184 / 4); }, mb216);
    for (int contiguous = 0; contiguous < 2; ++contiguous) {
        {
            const int tpi = 24, nt = N * tpi, per = (nt + 255) / 256;
            char nm[128]; snprintf(nm, 128, "216-float segments (864 B), plane 5184, %s tiles", contiguous ? "contiguous runs of" : "grid-stride");
            run(nm, [&] { hipLaunchKernelGGL(tile_store<216>, dim3((nt + per - 1) / per), dim3(256), 0, 0, buf, nt, tpi, 5184, per, contiguous); }, mb216);
        }
        {
            const int tpi = 22, nt = N * tpi, per = (nt + 255) / 256;
            char nm[128]; snprintf(nm, 128, "256-float segments (1 KB), plane 5632, %s tiles", contiguous ? "contiguous runs of" : "grid-stride");
            run(nm, [&] { hipLaunchKernelGGL(tile_store<256>, dim3((nt + per - 1) / per), dim3(256), 0, 0, buf, nt, tpi, 5632, per, contiguous); }, mb256);
        }
        {
            const int tpi = 24, nt = N * tpi, per = (nt + 768 - 1) / 768;
            char nm[128]; snprintf(nm, 128, "216-float segments, 768 workgroups, %s tiles", contiguous ? "contiguous runs of" : "grid-stride");
            run(nm, [&] { hipLaunchKernelGGL(tile_store<216>, dim3((nt + per - 1) / per), dim3(256), 0, 0, buf, nt, tpi, 5184, per, contiguous); }, mb216);
        }
    }
    return 0;
}
