// Segmentation metrics (validation / scripts/compute_seg_metrics.py): the contingency table of two integer label
// maps per image.  Both reference metrics are functions of it alone:
//   utils/misc.py:101-114  average_ari       -> sklearn.metrics.adjusted_rand_score(pred, gt) per image, whose
//                                               published algorithm works on the contingency matrix
//   utils/misc.py:173-235  average_segcover  -> iou(A == i, (B == j) & (A >= 0)) = n_ij / (a_i + b_j - n_ij)
// The reference does this on the host with per-image Python loops over numpy / boolean-mask passes (it dominates
// validation once the forward pass is fast); here one workgroup per image histograms its pixels into LDS with
// integer atomics (exact and order-independent) and the tiny [B, KA, KB+1] tables stay on the device.
#include "gx_common.h"

namespace {

// counts[b][i][j], i in [0,KA), j in [0,KB] -- column KB collects segB labels outside [0,KB); pixels whose segA label
// is outside [0,KA) (the reference's "ignore" regions, label < 0) are not counted at all.
__global__ void __launch_bounds__(256)
label_contingency_kernel(const long long* __restrict__ segA, const long long* __restrict__ segB, int HW, int KA,
                         int KB, int* __restrict__ counts) {
    extern __shared__ int hist[];
    const int cells = KA * (KB + 1);
    for (int i = threadIdx.x; i < cells; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const long long* a = segA + (size_t)blockIdx.x * HW;
    const long long* b = segB + (size_t)blockIdx.x * HW;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const long long la = a[p], lb = b[p];
        if (la >= 0 && la < KA) {
            const int j = (lb >= 0 && lb < KB) ? (int)lb : KB;
            atomicAdd(&hist[(int)la * (KB + 1) + j], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += blockDim.x) counts[(size_t)blockIdx.x * cells + i] = hist[i];
}

}  // namespace

extern "C" {

int gx_label_contingency(const long long* segA, const long long* segB, int B, int HW, int KA, int KB, int* counts,
                         gx_stream_t stream) {
    GX_CHECK_ARG(segA && segB && counts, "gx_label_contingency: null pointer");
    GX_CHECK_ARG(B > 0 && HW > 0 && KA > 0 && KB > 0, "gx_label_contingency: bad dims");
    GX_CHECK_ARG((size_t)KA * (KB + 1) * sizeof(int) <= 64 * 1024, "gx_label_contingency: table exceeds 64 KiB of LDS");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 16.0 * B * HW);
        hipLaunchKernelGGL(label_contingency_kernel, dim3(B), dim3(256), (size_t)KA * (KB + 1) * sizeof(int), s, segA,
                           segB, HW, KA, KB, counts);
    }
    GX_CHECK_LAUNCH("gx_label_contingency");
    return GX_OK;
}

}  // extern "C"
