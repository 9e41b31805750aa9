// Input side of the training step: uint8 HWC frames as the datasets store them -> the fp32 NCHW batch
// in [0,1] the model consumes, on the device.  Reference: datasets/multid_config.py:131-135 (transforms.ToTensor(): HWC
// uint8 -> CHW float32 / 255; F.interpolate(size=img_size), default mode 'nearest', when the stored size differs),
// datasets/multi_object_config.py:176-186 (np.moveaxis(img, 3, 1); torch.FloatTensor(img) / 255.; F.interpolate).
// There the conversion runs on the host per sample and the fp32 batch (4x the bytes) crosses PCIe (train.py:218-220);
// here the uint8 frames cross PCIe and one HBM-bound launch converts them.
#include "gx_common.h"

namespace {

// dst[b][c][y][x] = src[b][sy][sx][c] / 255 with the 'nearest' source index of F.interpolate:
// s = min(floor(d * (float)in / out), in - 1)  (ATen nearest_neighbor_compute_source_index, scale = in / out in fp32)
__global__ void __launch_bounds__(256)
u8hwc_to_f32chw_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int B, int Hs, int Ws, int C,
                       int H, int W) {
    const size_t total = (size_t)B * C * H * W;
    const float sh = (float)Hs / (float)H, sw = (float)Ws / (float)W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const int b = (int)(i / ((size_t)W * H * C));
        int sy = (int)floorf((float)y * sh), sx = (int)floorf((float)x * sw);
        sy = sy < Hs - 1 ? sy : Hs - 1;
        sx = sx < Ws - 1 ? sx : Ws - 1;
        const unsigned char v = src[(((size_t)b * Hs + sy) * Ws + sx) * C + c];
        dst[i] = (float)v / 255.0f;          // a true division, as ToTensor's .div(255) is
    }
}

}  // namespace

extern "C" {

int gx_u8hwc_to_f32chw(const unsigned char* src, float* dst, int B, int Hs, int Ws, int C, int H, int W,
                       gx_stream_t stream) {
    GX_CHECK_ARG(src && dst, "gx_u8hwc_to_f32chw: null pointer");
    GX_CHECK_ARG(B > 0 && Hs > 0 && Ws > 0 && C > 0 && H > 0 && W > 0, "gx_u8hwc_to_f32chw: bad dims");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)B * C * H * W;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, (double)B * Hs * Ws * C + 4.0 * total);
        hipLaunchKernelGGL(u8hwc_to_f32chw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, B, Hs, Ws, C, H, W);
    }
    GX_CHECK_LAUNCH("gx_u8hwc_to_f32chw");
    return GX_OK;
}

}  // extern "C"
