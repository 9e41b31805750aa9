// Spatial broadcast + coordinate-channel concat (reference modules/blocks.py:104-130 BroadcastLayer / PixelCoords),
// and the first 3x3 convolution of the BroadcastDecoder (modules/decoders.py:21-35) applied to it WITHOUT the canvas:
//
//   canvas[n] = [ z[n] broadcast over d x d | g_1 (row coordinate) | g_2 (column coordinate) ]      (L + 2 channels)
//   y1 = act(conv3x3(canvas, w) + b)
//
// z[n] is constant over the image and g_1 / g_2 depend on the row / the column only (meshgrid of one linspace,
// blocks.py:121-126), so inside the canvas
//   conv3x3(canvas, w)[n][co][y][x] = sum_ci z[n][ci] * WS[co][ci]  +  A[co][y]  +  B[co][x]
//   WS[co][ci] = sum_taps w[co][ci],   A[co][y] = sum_ky rowc[y+ky-1] * sum_kx w[co][L][ky][kx],
//                                      B[co][x] = sum_kx colc[x+kx-1] * sum_ky w[co][L+1][ky][kx]
// -- one HBM-bound write of y1 instead of an (L+2)-channel canvas written, read back by an MFMA conv (K*B = 224
// slots, 72 x 72: 83 MB each way) -- and in the backward pass every gradient of this layer is a function of SEVEN
// sums per (n, co) plane of dy1 = g * act'(y1):  D = sum dy, R_k = sum dy * rowc[y+k], C_k = sum dy * colc[x+k]
// (k = -1, 0, 1):  db = sum_n D,  dW[co][ci][t] = sum_n D z[n][ci] (every tap),  dW[co][L][ky][kx] = sum_n R_{ky-1},
// dW[co][L+1][ky][kx] = sum_n C_{kx-1},  dz[n][ci] = sum_co D WS[co][ci]; dy1 itself is never written.
// The decoder runs its L VALID convs as 'same' convs on the canvas and crops the centre: the one-pixel border ring of
// y1 is outside the first valid conv's output and is never read by anything that reaches the crop; the backward sums
// run over the interior [1, d-2]^2 only (= the valid conv's output positions), whatever the ring of g holds.
#include "gx_common.h"

namespace {

__device__ __forceinline__ float b_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : expm1f(v);
    return v;
}
// derivative of the activation in terms of its OUTPUT y
__device__ __forceinline__ float b_dact(float y, int act) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}

// out[n][c] = z[n][c] broadcast (c < D), coords[c - D] otherwise: the decoder input of GENESIS-V2
// (models/genesisv2_config.py:89-90: BroadcastLayer(img_size / 16))
__global__ void __launch_bounds__(256)
broadcast_concat_kernel(const float* __restrict__ z, const float* __restrict__ coords, int N, int D, int dd,
                        float* __restrict__ out) {
    const size_t total = (size_t)N * (D + 2) * dd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % dd);
        const int c = (int)((i / dd) % (D + 2));
        const int n = (int)(i / ((size_t)dd * (D + 2)));
        out[i] = c < D ? z[(size_t)n * D + c] : coords[(size_t)(c - D) * dd + p];
    }
}

// one workgroup per (n, co) plane
__global__ void __launch_bounds__(256)
bcast_conv_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ bias,
                      const float* __restrict__ rowc, const float* __restrict__ colc, int L, int Co, int d, int act,
                      float* __restrict__ out, float* __restrict__ amax_parts) {
    extern __shared__ float sh[];     // A[d] | B[d]
    float* A = sh;
    float* Bc = sh + d;
    __shared__ float s_sh;
    const int n = blockIdx.x / Co, co = blockIdx.x - n * Co;
    const int tid = threadIdx.x;
    const float* wc = w + (size_t)co * (L + 2) * 9;
    if (tid < 64) {     // s = b + sum_ci z[n][ci] * sum_t w[co][ci][t]   (wave 0, fixed order)
        float acc = 0.f;
        for (int ci = tid; ci < L; ci += 64) {
            float ws = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) ws += wc[ci * 9 + t];
            acc += z[(size_t)n * L + ci] * ws;
        }
        acc = gx_wave_sum(acc);
        if (tid == 0) s_sh = acc + (bias ? bias[co] : 0.f);
    }
    // zero-padded coordinate rows / columns (the padded values only reach the unused border ring)
    for (int i = tid; i < 2 * d; i += 256) {
        const bool isrow = i < d;
        const int p = isrow ? i : i - d;
        const float* cw = wc + (L + (isrow ? 0 : 1)) * 9;
        const float* cv = isrow ? rowc : colc;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int q = p + k - 1;
            const float c = (q >= 0 && q < d) ? cv[q] : 0.f;
            const float wk = isrow ? (cw[3 * k] + cw[3 * k + 1] + cw[3 * k + 2]) : (cw[k] + cw[3 + k] + cw[6 + k]);
            acc += c * wk;
        }
        sh[i] = acc;
    }
    __syncthreads();
    const float s = s_sh;
    float* o = out + (size_t)blockIdx.x * d * d;
    const int d4 = d >> 2;     // d is a multiple of 4 (checked by the host)
    float am = 0.f;
    for (int i = tid; i < d * d4; i += 256) {
        const int y = i / d4, x = (i - y * d4) * 4;
        const float a = s + A[y];
        f32x4 v;
        v[0] = b_act(a + Bc[x], act); v[1] = b_act(a + Bc[x + 1], act);
        v[2] = b_act(a + Bc[x + 2], act); v[3] = b_act(a + Bc[x + 3], act);
        *reinterpret_cast<f32x4*>(o + (size_t)y * d + x) = v;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    // an armed gx_amax_tap: the plane's largest magnitude for the canvas conv that reads this tensor next (its fp16 scale)
    if (amax_parts) gx_block_amax_store(am, amax_parts, blockIdx.x);      // (uniform)
}

// seven interior sums of dy = g * act'(y) per (n, co) plane -> sums[plane][8] (D, R-1, R0, R+1, C-1, C0, C+1, 0)
__global__ void __launch_bounds__(256)
bcast_conv_bwd_sums_kernel(const float* __restrict__ y, const float* __restrict__ g, const float* __restrict__ rowc,
                           const float* __restrict__ colc, int d, int act, float* __restrict__ sums) {
    extern __shared__ float sh[];     // rowc[d] | colc[d]
    __shared__ double red[4][7];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * d; i += 256) sh[i] = i < d ? rowc[i] : colc[i - d];
    __syncthreads();
    const float* rc = sh;
    const float* cc = sh + d;
    const float* yp = y + (size_t)blockIdx.x * d * d;
    const float* gp = g + (size_t)blockIdx.x * d * d;
    float acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = 0.f;
    const int d4 = d >> 2;
    for (int i = tid; i < d * d4; i += 256) {
        const int yy = i / d4, x0 = (i - yy * d4) * 4;
        if (yy < 1 || yy > d - 2) continue;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(yp + (size_t)yy * d + x0);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + (size_t)yy * d + x0);
        float rs = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = x0 + e;
            if (x < 1 || x > d - 2) continue;
            const float dy = gv[e] * b_dact(yv[e], act);
            rs += dy;
            acc[4] += dy * cc[x - 1]; acc[5] += dy * cc[x]; acc[6] += dy * cc[x + 1];
        }
        acc[0] += rs;
        acc[1] += rs * rc[yy - 1]; acc[2] += rs * rc[yy]; acc[3] += rs * rc[yy + 1];
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double v = gx_wave_sum_d((double)acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < 8) sums[(size_t)blockIdx.x * 8 + tid] = tid < 7 ? (float)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) : 0.f;
}

// blockIdx.x < Co: dW[co], db[co] (sum over n, fixed order, fp64); blockIdx.x >= Co: dz rows, 4 images per workgroup
__global__ void __launch_bounds__(256)
bcast_conv_bwd_finish_kernel(const float* __restrict__ sums, const float* __restrict__ z, const float* __restrict__ w,
                             int N, int L, int Co, float* __restrict__ dz, float* __restrict__ dw,
                             float* __restrict__ db) {
    __shared__ double red[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x < Co) {
        const int co = blockIdx.x;
        // the seven plane sums over n
        double a7[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) a7[k] = 0.0;
        for (int n = tid; n < N; n += 256) {
            const float* sp = sums + ((size_t)n * Co + co) * 8;
#pragma unroll
            for (int k = 0; k < 7; ++k) a7[k] += (double)sp[k];
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double v = gx_wave_sum_d(a7[k]);
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        float* dwc = dw + (size_t)co * (L + 2) * 9;
        if (tid < 7) {
            const double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            if (tid == 0) { if (db) db[co] = (float)v; }
            else if (tid <= 3) { for (int kx = 0; kx < 3; ++kx) dwc[L * 9 + (tid - 1) * 3 + kx] = (float)v; }          // R_{ky-1}
            else { for (int ky = 0; ky < 3; ++ky) dwc[(L + 1) * 9 + ky * 3 + (tid - 4)] = (float)v; }                   // C_{kx-1}
        }
        // dW[co][ci][every tap] = sum_n D[n][co] z[n][ci]: one wave per ci (round robin)
        for (int ci = wave; ci < L; ci += 4) {
            double a = 0.0;
            for (int n = lane; n < N; n += 64) a += (double)sums[((size_t)n * Co + co) * 8] * (double)z[(size_t)n * L + ci];
            a = gx_wave_sum_d(a);
            if (lane < 9) dwc[ci * 9 + lane] = (float)a;
        }
    } else {
        // dz[n][ci] = sum_co D[n][co] * WS[co][ci]
        const int n = ((int)blockIdx.x - Co) * 4 + wave;
        if (n >= N) return;
        for (int ci = lane; ci < L; ci += 64) {
            double a = 0.0;
            for (int co = 0; co < Co; ++co) {
                const float* wc = w + ((size_t)co * (L + 2) + ci) * 9;
                float ws = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) ws += wc[t];
                a += (double)sums[((size_t)n * Co + co) * 8] * (double)ws;
            }
            dz[(size_t)n * L + ci] = (float)a;
        }
    }
}


// ---- transposed conv (k5, s2, p2, op1) on a spatially BROADCAST input (models/genesisv2_config.py:89-90: the decoder's
// first layer sees z[n] repeated over the d x d grid plus two coordinate channels).  Because the input is constant over
// the pixels, out[n, co, oy, ox] = sum_ci z[n, ci] * Wz[ci][co][oy][ox] + C[co][oy][ox] with
//   Wz[ci][co][oy][ox] = sum over the taps (kh, kw) that reach (oy, ox) from inside the grid of w[ci][co][kh][kw]
//   C [co][oy][ox]     = b[co] + sum_c sum_taps w[D + c][co][kh][kw] * coords[c][iy][ix]       (input independent)
// i.e. a [N, D] x [D, Cout (2d)^2] matrix product (1/(number of taps) of the multiplies, no canvas in memory, and the
// input gradient needs no reduction over pixels).  pack builds Wz^T (nn.Linear weight layout [Cout (2d)^2, D]) and C
// from the layer's parameters every step; unpack folds the gradients of Wz^T and C back onto w and b.
// Tap geometry: oy = 2 iy - 2 + kh, iy in [0, d).
__global__ void __launch_bounds__(256)
bcast_deconv_pack_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ coords,
                         int D, int Cout, int d, float* __restrict__ wz, float* __restrict__ bias) {
    const int P = 4 * d * d, W2 = 2 * d;
    const size_t total = (size_t)Cout * P * D;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int ci = (int)(idx % D);
        const int row = (int)(idx / D);                 // co * P + oy * W2 + ox
        const int co = row / P, pos = row - co * P, oy = pos / W2, ox = pos - oy * W2;
        const float* wc = w + ((size_t)ci * Cout + co) * 25;
        float s = 0.f, c0 = 0.f, c1 = 0.f;
        for (int kh = oy & 1; kh < 5; kh += 2) {
            const int iy = (oy + 2 - kh) >> 1;
            if (iy < 0 || iy >= d) continue;
            for (int kw = ox & 1; kw < 5; kw += 2) {
                const int ix = (ox + 2 - kw) >> 1;
                if (ix < 0 || ix >= d) continue;
                s += wc[kh * 5 + kw];
                if (ci == 0) {
                    c0 += w[((size_t)D * Cout + co) * 25 + kh * 5 + kw] * coords[iy * d + ix];
                    c1 += w[((size_t)(D + 1) * Cout + co) * 25 + kh * 5 + kw] * coords[d * d + iy * d + ix];
                }
            }
        }
        wz[idx] = s;
        if (ci == 0) bias[row] = (b ? b[co] : 0.f) + c0 + c1;
    }
}

__global__ void __launch_bounds__(256)
bcast_deconv_unpack_kernel(const float* __restrict__ dwz, const float* __restrict__ dbias,
                           const float* __restrict__ coords, int D, int Cout, int d, float* __restrict__ dw,
                           float* __restrict__ db) {
    const int P = 4 * d * d, W2 = 2 * d, Dp = D + 2;
    const int total = Dp * Cout * 25;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int ci = idx % Dp;                        // fastest: the reads of dwz are contiguous over ci
        const int r = idx / Dp, co = r / 25, t = r - co * 25, kh = t / 5, kw = t - kh * 5;
        float s = 0.f;
        for (int iy = 0; iy < d; ++iy) {
            const int oy = 2 * iy - 2 + kh;
            if (oy < 0 || oy >= W2) continue;
            for (int ix = 0; ix < d; ++ix) {
                const int ox = 2 * ix - 2 + kw;
                if (ox < 0 || ox >= W2) continue;
                const size_t row = (size_t)co * P + oy * W2 + ox;
                s += ci < D ? dwz[row * D + ci] : dbias[row] * coords[(ci - D) * d * d + iy * d + ix];
            }
        }
        dw[((size_t)ci * Cout + co) * 25 + t] = s;
        if (db && ci == 0 && t == 0) {
            float sb = 0.f;
            for (int q = 0; q < P; ++q) sb += dbias[(size_t)co * P + q];
            db[co] = sb;
        }
    }
}

}  // namespace

extern "C" {

int gx_broadcast_concat(const float* z, const float* coords, float* out, int N, int D, int d, gx_stream_t stream) {
    GX_CHECK_ARG(z && coords && out && N > 0 && D > 0 && d > 0, "gx_broadcast_concat: bad arguments");
    const size_t total = (size_t)N * (D + 2) * d * d;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * total);
        hipLaunchKernelGGL(broadcast_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, s, z, coords, N, D, d * d, out);
    }
    GX_CHECK_LAUNCH("gx_broadcast_concat");
    return GX_OK;
}

int gx_bcast_conv3x3_fwd(const float* z, const float* w, const float* bias, const float* rowc, const float* colc,
                         int act, float* out, int N, int L, int Co, int d, gx_stream_t stream) {
    GX_CHECK_ARG(z && w && rowc && colc && out, "gx_bcast_conv3x3_fwd: null pointer");
    GX_CHECK_ARG(N > 0 && L > 0 && Co > 0 && d >= 4 && (d & 3) == 0 && d <= 4096, "gx_bcast_conv3x3_fwd: bad dims (d a multiple of 4)");
    GX_CHECK_ARG(act >= 0 && act <= 2, "gx_bcast_conv3x3_fwd: act must be 0 (none), 1 (ReLU) or 2 (ELU)");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_CONV1X1_FWD, s, 0.0, 4.0 * N * Co * (double)d * d);
        float* ap = gx_amax_producer_out(out, false, (unsigned)(N * Co), (size_t)N * Co * d * d);
        hipLaunchKernelGGL(bcast_conv_fwd_kernel, dim3(N * Co), dim3(256), 2 * d * sizeof(float), s, z, w, bias, rowc, colc,
                           L, Co, d, act, out, ap);
    }
    GX_CHECK_LAUNCH("gx_bcast_conv3x3_fwd");
    return GX_OK;
}

size_t gx_bcast_conv3x3_bwd_ws_bytes(int N, int Co) { return (size_t)N * Co * 8 * sizeof(float); }

int gx_bcast_conv3x3_bwd(const float* y, const float* g, const float* z, const float* w, const float* rowc,
                         const float* colc, int act, int N, int L, int Co, int d, float* dz, float* dw, float* db,
                         void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(y && g && z && w && rowc && colc && dz && dw && ws, "gx_bcast_conv3x3_bwd: null pointer");
    GX_CHECK_ARG(N > 0 && L > 0 && Co > 0 && d >= 4 && (d & 3) == 0 && d <= 4096, "gx_bcast_conv3x3_bwd: bad dims (d a multiple of 4)");
    GX_CHECK_ARG(act >= 0 && act <= 2, "gx_bcast_conv3x3_bwd: act must be 0, 1 or 2");
    GX_CHECK_ARG(ws_bytes >= gx_bcast_conv3x3_bwd_ws_bytes(N, Co), "gx_bcast_conv3x3_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* sums = (float*)ws;
    {
        GxProf pf(KID_BIAS_ACT_BWD, s, 0.0, 8.0 * N * Co * (double)d * d);
        hipLaunchKernelGGL(bcast_conv_bwd_sums_kernel, dim3(N * Co), dim3(256), 2 * d * sizeof(float), s, y, g, rowc, colc, d,
                           act, sums);
    }
    GX_CHECK_LAUNCH("gx_bcast_conv3x3_bwd(sums)");
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 32.0 * N * Co);
        hipLaunchKernelGGL(bcast_conv_bwd_finish_kernel, dim3(Co + gx_ceil_div(N, 4)), dim3(256), 0, s, (const float*)sums, z,
                           w, N, L, Co, dz, dw, db);
    }
    GX_CHECK_LAUNCH("gx_bcast_conv3x3_bwd(finish)");
    return GX_OK;
}

int gx_bcast_deconv5x5s2_pack(const float* w, const float* b, const float* coords, int D, int Cout, int d, float* wz,
                              float* bias, gx_stream_t stream) {
    GX_CHECK_ARG(w && coords && wz && bias && D > 0 && Cout > 0 && d > 0, "gx_bcast_deconv5x5s2_pack: bad arguments");
    const size_t total = (size_t)Cout * 4 * d * d * D;
    size_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_PACK_WEIGHTS, s, 0.0, 4.0 * total);
        hipLaunchKernelGGL(bcast_deconv_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, b, coords, D, Cout, d, wz,
                           bias);
    }
    GX_CHECK_LAUNCH("gx_bcast_deconv5x5s2_pack");
    return GX_OK;
}

int gx_bcast_deconv5x5s2_unpack(const float* dwz, const float* dbias, const float* coords, int D, int Cout, int d,
                                float* dw, float* db, gx_stream_t stream) {
    GX_CHECK_ARG(dwz && dbias && coords && dw && D > 0 && Cout > 0 && d > 0, "gx_bcast_deconv5x5s2_unpack: bad arguments");
    const int total = (D + 2) * Cout * 25;
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 4.0 * ((double)Cout * 4 * d * d * D + total));
        hipLaunchKernelGGL(bcast_deconv_unpack_kernel, dim3(gx_ceil_div(total, 256)), dim3(256), 0, s, dwz, dbias, coords,
                           D, Cout, d, dw, db);
    }
    GX_CHECK_LAUNCH("gx_bcast_deconv5x5s2_unpack");
    return GX_OK;
}

}  // extern "C"

