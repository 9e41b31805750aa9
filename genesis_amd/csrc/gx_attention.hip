// GENESIS-V2 Instance-Colouring Stick-Breaking Process (IC-SBP), forward + backward.
//
// Reference: modules/attention.py:162-226 (InstanceColouringSBP.forward),
//            modules/blocks.py:63-71 (squared_distance), :18-20 (clamp_preserve_gradients).
//
// Forward (one workgroup per image -- the K-1 steps are sequential through the scope, and each
// step needs an argmax over the whole image, resolved with wavefront shuffles + one LDS hop):
//   scope = exp(log_s); idx = first argmax_hw(rand * scope); seed = colour[:, idx]
//   d = sum_c (colour_c - seed_c)^2 ; alpha = kernel(d, sigma) ; alpha_c = ST-clamp(alpha, .01, .99)
//   log_m_t = log_s + log(alpha_c) ; log_s <- log_s + log(1 - alpha_c) ; last mask = last scope.
// log_s lives in registers across steps; colour (C*HW*4 B per image, L2 resident) is re-read per
// step.  HBM-bound: algorithmic bytes (C + 1 + 2K) * HW * 4 per image.
//
// Backward is per-pixel independent apart from the seed-gradient reduction: the gradient w.r.t.
// the scope entering step t is the suffix sum of the mask gradients, so one sweep t = K-2..0 per
// pixel yields d colour, and block reductions (fixed tree, deterministic) yield the gradient that
// flows through the gathered seed (scatter-add into the seed pixel -- the reference's CopySlices)
// and d log_sigma.
#include "gx_common.h"

namespace {

enum { KERNEL_GAUSSIAN = 0, KERNEL_LAPLACIAN = 1, KERNEL_EPANECHNIKOV = 2 };
constexpr int MAXC = 8;      // colour channels supported (reference hard-codes colour_dim=8, genesisv2_config.py:74)
constexpr int MAXPPT = 16;   // HW <= 1024 * 16 floats of LDS state per image

__device__ __forceinline__ float st_clamp(float a, float lo, float hi) {
    const float c = fminf(fmaxf(a, lo), hi);
    return a + (c - a);  // value of x + (clamp(x) - x).detach()
}

// PPT > 0: the image has exactly PPT pixels per thread (HW = PPT * blockDim.x, PPT <= 4: 64x64 and smaller): the
// thread's colour vectors, its random numbers and its log-scope stay in registers over the K-1 steps (the generic path
// re-reads the colour map from L2 in every step: 54 -> 3x us at K=7, 64x64).  PPT == 0: generic (LDS scope, global colour).
template <int PPT>
__global__ void __launch_bounds__(1024)
icsbp_fwd_kernel(const float* __restrict__ colour, const double* __restrict__ log_sigma,
                 const float* __restrict__ rand_pixel, const int64_t* __restrict__ seed_idx_in,
                 int B, int C, int HW, int K, int kernel_type,
                 float* __restrict__ log_m, float* __restrict__ log_s_out,
                 float* __restrict__ seeds, int64_t* __restrict__ seed_idx_out,
                 float min_mass, int* __restrict__ nsteps_out) {
    // min_mass > 0 (dynamic_K, modules/attention.py:218-219): an image stops at the first step whose mask would hold
    // fewer than min_mass pixels (sum of exp(log_m)); that step's mask becomes the remaining scope, later masks are
    // -1e10 (the padding of models/genesisv2_config.py:126-130) and nsteps_out[b] = number of steps executed.
    __shared__ float red_v[16];
    __shared__ int red_i[16];
    __shared__ float seed_sh[MAXC];
    __shared__ int idx_sh;
    const int b = blockIdx.x;
    const int T = blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
    const float* col = colour + (size_t)b * C * HW;
    const float* rnd = rand_pixel + (size_t)b * HW;
    const float sigma = (float)exp(*log_sigma);  // 0-dim fp64 parameter enters fp32 math as a scalar
    const size_t kstride = (size_t)B * HW;

    // log-scope of this image, carried across the K-1 steps in LDS (each thread only ever touches
    // its own pixels p = tid + q*T, so no barrier is needed around it)
    extern __shared__ __attribute__((aligned(16))) float ls[];
    constexpr int NR = PPT > 0 ? PPT : 1;
    float cv[NR][MAXC], rv[NR], lsr[NR];          // register-resident state of the thread's pixels (PPT > 0)
    if (PPT > 0) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int p = tid + q * T;
            rv[q] = rnd[p];
            lsr[q] = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) cv[q][c] = c < C ? col[(size_t)c * HW + p] : 0.f;
            log_s_out[(size_t)b * HW + p] = 0.f;
        }
    } else {
        for (int p = tid; p < HW; p += T) {
            ls[p] = 0.f;
            log_s_out[(size_t)b * HW + p] = 0.f;
        }
    }

    int n_exec = K - 1;                 // steps executed (dynamic_K may stop earlier)
    for (int t = 0; t < K - 1; ++t) {
        int best_i;
        if (seed_idx_in) {
            best_i = (int)seed_idx_in[(size_t)t * B + b];
        } else {
            float best_v = -INFINITY;
            best_i = 0x7fffffff;
            if (PPT > 0) {
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const float v = rv[q] * expf(lsr[q]);
                    if (v > best_v) { best_v = v; best_i = tid + q * T; }  // ascending p: keeps the first max
                }
            } else {
                for (int p = tid; p < HW; p += T) {
                    const float v = rnd[p] * expf(ls[p]);
                    if (v > best_v) { best_v = v; best_i = p; }  // ascending p: keeps the first max
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(best_v, off, 64);
                const int oi = __shfl_xor(best_i, off, 64);
                if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
            }
            __syncthreads();
            if (lane == 0) { red_v[wave] = best_v; red_i[wave] = best_i; }
            __syncthreads();
            if (tid == 0) {
                float bv = red_v[0]; int bi = red_i[0];
                for (int w = 1; w < nw; ++w)
                    if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
                idx_sh = bi;
            }
            __syncthreads();
            best_i = idx_sh;
        }
        __syncthreads();
        if (tid < C) {
            const float sv = col[(size_t)tid * HW + best_i];
            seed_sh[tid] = sv;
            seeds[((size_t)t * B + b) * C + tid] = sv;
        }
        if (tid == 0) seed_idx_out[(size_t)t * B + b] = best_i;
        __syncthreads();
        if (min_mass > 0.f) {
            float mass = 0.f;
            for (int qq = 0, p = tid; p < HW; p += T, ++qq) {
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    if (c < C) {
                        const float df = (PPT > 0 ? cv[qq < NR ? qq : 0][c] : col[(size_t)c * HW + p]) - seed_sh[c];
                        d += df * df;
                    }
                }
                float alpha;
                if (kernel_type == KERNEL_GAUSSIAN) alpha = expf(-d / sigma);
                else if (kernel_type == KERNEL_LAPLACIAN) alpha = expf(-sqrtf(st_clamp(d, 1e-10f, 1e10f)) / sigma);
                else alpha = fmaxf(1.f - d / sigma, 0.f);
                alpha = st_clamp(alpha, 0.01f, 0.99f);
                mass += expf((PPT > 0 ? lsr[qq < NR ? qq : 0] : ls[p]) + logf(alpha));
            }
            mass = gx_wave_sum(mass);
            __syncthreads();
            if (lane == 0) red_v[wave] = mass;
            __syncthreads();
            float tot = 0.f;
            for (int w = 0; w < nw; ++w) tot += red_v[w];
            __syncthreads();
            if (tot < min_mass) { n_exec = t; break; }          // uniform over the workgroup
        }
        for (int qq = 0, p = tid; p < HW; p += T, ++qq) {
            {
                float d = 0.f;
                if (PPT > 0) {
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) {
                        if (c < C) {
                            const float df = cv[qq < NR ? qq : 0][c] - seed_sh[c];
                            d += df * df;
                        }
                    }
                } else {
                    for (int c = 0; c < C; ++c) {
                        const float df = col[(size_t)c * HW + p] - seed_sh[c];
                        d += df * df;
                    }
                }
                float alpha;
                if (kernel_type == KERNEL_GAUSSIAN) {
                    alpha = expf(-d / sigma);
                } else if (kernel_type == KERNEL_LAPLACIAN) {
                    alpha = expf(-sqrtf(st_clamp(d, 1e-10f, 1e10f)) / sigma);
                } else {
                    alpha = fmaxf(1.f - d / sigma, 0.f);
                }
                alpha = st_clamp(alpha, 0.01f, 0.99f);
                const float log_a = logf(alpha);
                const float log_na = logf(1.f - alpha);
                const float lsp = PPT > 0 ? lsr[qq < NR ? qq : 0] : ls[p];
                log_m[t * kstride + (size_t)b * HW + p] = lsp + log_a;
                if (PPT > 0) lsr[qq < NR ? qq : 0] = lsp + log_na;
                else ls[p] = lsp + log_na;
                log_s_out[(t + 1) * kstride + (size_t)b * HW + p] = lsp + log_na;
            }
        }
    }
    for (int qq = 0, p = tid; p < HW; p += T, ++qq) {
        const float lsp = PPT > 0 ? lsr[qq < NR ? qq : 0] : ls[p];
        log_m[n_exec * kstride + (size_t)b * HW + p] = lsp;                     // last mask = remaining scope
        for (int t = n_exec + 1; t < K; ++t) {
            log_m[t * kstride + (size_t)b * HW + p] = -1e10f;
            log_s_out[t * kstride + (size_t)b * HW + p] = lsp;
        }
    }
    for (int t = n_exec + 1; t < K - 1; ++t) {        // steps never run: no seed
        if (tid < C) seeds[((size_t)t * B + b) * C + tid] = 0.f;
        if (tid == 0) seed_idx_out[(size_t)t * B + b] = 0;
    }
    if (nsteps_out && tid == 0) nsteps_out[b] = n_exec;
}

// Reduces NV per-thread doubles over the block at once (wave shuffles + one LDS hop), result broadcast.
template <int NV>
__device__ __forceinline__ void block_sum_multi(double (&v)[NV], double* red /* [16][NV] + [NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = gx_wave_sum_d(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[w * NV + threadIdx.x];
        red[16 * NV + threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[16 * NV + i];
}

// Backward, stage 1: one thread per pixel (the recursion is per-pixel independent: the gradient w.r.t. the
// scope leaving step t is the suffix sum of the mask gradients).  grid (B, HW/T).  Writes d colour for the
// direct path and, per (image, chunk, step), the block-reduced seed gradient [C] and d sigma:
// part[b][chunk][t][0..C-1] = -sum_p 2 gd diff_c,  part[...][MAXC] = sum_p galpha * d alpha / d sigma.
__global__ void __launch_bounds__(256)
icsbp_bwd_kernel(const float* __restrict__ colour, const double* __restrict__ log_sigma,
                 const float* __restrict__ seeds, const float* __restrict__ g_m, int B, int C, int HW, int K,
                 int kernel_type, float* __restrict__ dcolour, double* __restrict__ part,
                 const int* __restrict__ nsteps) {
    __shared__ double red[16 * (MAXC + 1) + (MAXC + 1)];
    __shared__ float seed_sh[16][MAXC];   // K-1 <= 16
    const int b = blockIdx.x, tid = threadIdx.x;
    const int p = blockIdx.y * blockDim.x + tid;
    const bool ok = p < HW;
    const float sigma = (float)exp(*log_sigma);
    const size_t kstride = (size_t)B * HW;
    for (int i = tid; i < (K - 1) * C; i += blockDim.x) seed_sh[i / C][i % C] = seeds[((size_t)(i / C) * B + b) * C + i % C];
    float col[MAXC], dcol[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        col[c] = (ok && c < C) ? colour[((size_t)b * C + c) * HW + p] : 0.f;
        dcol[c] = 0.f;
    }
    // dynamic_K: this image ran n steps; mask n is its remaining scope, later masks are constants
    const int n = nsteps ? nsteps[b] : K - 1;
    float gs = ok ? g_m[n * kstride + (size_t)b * HW + p] : 0.f;
    __syncthreads();
    for (int t = K - 2; t >= 0; --t) {
        if (t >= n) {                  // (uniform over the workgroup)
            if (tid <= MAXC) part[(((size_t)b * gridDim.y + blockIdx.y) * (K - 1) + t) * (MAXC + 1) + tid] = 0.0;
            continue;
        }
        double vals[MAXC + 1];
        float diff[MAXC];
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            diff[c] = (c < C) ? col[c] - seed_sh[t][c] : 0.f;
            d += diff[c] * diff[c];
        }
        float alpha, dalpha_dd, dalpha_dsig;  // unclamped alpha and its partials
        if (kernel_type == KERNEL_GAUSSIAN) {
            alpha = expf(-d / sigma);
            dalpha_dd = -alpha / sigma;
            dalpha_dsig = alpha * d / (sigma * sigma);
        } else if (kernel_type == KERNEL_LAPLACIAN) {
            const float dist = sqrtf(st_clamp(d, 1e-10f, 1e10f));
            alpha = expf(-dist / sigma);
            dalpha_dd = -alpha / sigma * (0.5f / dist);
            dalpha_dsig = alpha * dist / (sigma * sigma);
        } else {
            const float u = 1.f - d / sigma;
            alpha = fmaxf(u, 0.f);
            const float on = u > 0.f ? 1.f : 0.f;
            dalpha_dd = -on / sigma;
            dalpha_dsig = on * d / (sigma * sigma);
        }
        const float ac = st_clamp(alpha, 0.01f, 0.99f);
        const float gm_t = ok ? g_m[t * kstride + (size_t)b * HW + p] : 0.f;
        // log_m_t = s_t + log(ac); s_{t+1} = s_t + log(1-ac); straight-through: d ac / d alpha = 1
        const float galpha = gm_t / ac - gs / (1.f - ac);
        gs += gm_t;
        const float gd = galpha * dalpha_dd;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const float gc = 2.f * gd * diff[c];
            dcol[c] += gc;
            vals[c] = ok ? -(double)gc : 0.0;
        }
        vals[MAXC] = ok ? (double)(galpha * dalpha_dsig) : 0.0;
        block_sum_multi<MAXC + 1>(vals, red);
        if (tid <= MAXC)
            part[(((size_t)b * gridDim.y + blockIdx.y) * (K - 1) + t) * (MAXC + 1) + tid] = vals[tid < MAXC ? tid : MAXC];
    }
    if (ok) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) dcolour[((size_t)b * C + c) * HW + p] = dcol[c];
    }
}

// Stage 2 (grid B): sums the per-chunk partials in a fixed order, scatter-adds the gradient that flows through
// the gathered seed into the seed pixel (modules/attention.py:190-193), and emits d log_sigma per image.
__global__ void __launch_bounds__(256)
icsbp_bwd_finalize_kernel(const double* __restrict__ part, const int64_t* __restrict__ seed_idx,
                          const double* __restrict__ log_sigma, int B, int C, int HW, int K, int nchunks,
                          float* __restrict__ dcolour, double* __restrict__ dlog_sigma_part) {
    __shared__ double sums[16][MAXC + 1];
    __shared__ double quarter[4][16 * (MAXC + 1)];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = (K - 1) * (MAXC + 1);
    // every 4th chunk per thread quarter (the chunk loop is a chain of dependent-latency loads: 4x shorter, and its
    // loads independent of the running sum), the quarters combined in a fixed order
    {
        const int q = tid >> 6, i0 = tid & 63;
        for (int i = i0; i < n; i += 64) {
            const int t = i / (MAXC + 1), c = i % (MAXC + 1);
            double s0 = 0.0, s1 = 0.0;
            int ch = q;
            for (; ch + 4 < nchunks; ch += 8) {
                s0 += part[(((size_t)b * nchunks + ch) * (K - 1) + t) * (MAXC + 1) + c];
                s1 += part[(((size_t)b * nchunks + ch + 4) * (K - 1) + t) * (MAXC + 1) + c];
            }
            if (ch < nchunks) s0 += part[(((size_t)b * nchunks + ch) * (K - 1) + t) * (MAXC + 1) + c];
            quarter[q][i] = s0 + s1;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x)
        sums[i / (MAXC + 1)][i % (MAXC + 1)] = (quarter[0][i] + quarter[1][i]) + (quarter[2][i] + quarter[3][i]);
    __syncthreads();
    // one thread per colour channel walks the steps in order (two steps may have drawn the same seed pixel): K - 1
    // dependent read-modify-writes per thread instead of (K - 1) C on a single thread
    if (tid < C) {
        for (int t = 0; t < K - 1; ++t) {
            const int idx = (int)seed_idx[(size_t)t * B + b];
            dcolour[((size_t)b * C + tid) * HW + idx] += (float)sums[t][tid];
        }
    }
    if (tid == 64) {
        double ds = 0.0;
        for (int t = 0; t < K - 1; ++t) ds += sums[t][MAXC];
        dlog_sigma_part[b] = ds * exp(*log_sigma);   // d log_sigma = d sigma * sigma
    }
}

__global__ void sum_double_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += part[i];
        *out = s;
    }
}

int threads_for(int HW) {
    int T = HW < 1024 ? HW : 1024;
    if (T < 64) T = 64;
    return T;
}

}  // namespace

extern "C" {

static int icsbp_fwd_impl(const float* colour, const double* log_sigma, const float* rand_pixel,
                          const int64_t* seed_idx_in, int B, int C, int H, int W, int K, int kernel_type, float* log_m,
                          float* log_s, float* seeds, int64_t* seed_idx_out, float min_mass, int* nsteps_out,
                          gx_stream_t stream) {
    GX_CHECK_ARG(colour && log_sigma && rand_pixel && log_m && log_s && seeds && seed_idx_out,
                 "gx_icsbp_fwd: null pointer");
    const int HW = H * W;
    GX_CHECK_ARG(B > 0 && C > 0 && C <= MAXC && K >= 1 && K <= 17, "gx_icsbp_fwd: bad B/C/K (C<=8, K<=17)");
    GX_CHECK_ARG(gx_is_pow2(HW) && HW >= 64 && HW <= 1024 * MAXPPT, "gx_icsbp_fwd: H*W must be a power of two in [64,16384]");
    GX_CHECK_ARG(kernel_type >= 0 && kernel_type <= 2, "gx_icsbp_fwd: no valid kernel");
    {
        GxProf pf(KID_ICSBP_FWD, (hipStream_t)stream, 0.0, 4.0 * B * HW * (C + 1.0 + 2.0 * K));
        const int T = threads_for(HW);
        const int ppt = (HW % T == 0) ? HW / T : 0;
#define GX_ICSBP_LAUNCH(P_)                                                                                          \
        hipLaunchKernelGGL(icsbp_fwd_kernel<P_>, dim3(B), dim3(T), HW * sizeof(float), (hipStream_t)stream, colour,    \
                           log_sigma, rand_pixel, seed_idx_in, B, C, HW, K, kernel_type, log_m, log_s, seeds, seed_idx_out,      \
                           min_mass, nsteps_out)
        if (ppt == 1) GX_ICSBP_LAUNCH(1);
        else if (ppt == 2) GX_ICSBP_LAUNCH(2);
        else if (ppt == 4) GX_ICSBP_LAUNCH(4);
        else GX_ICSBP_LAUNCH(0);
#undef GX_ICSBP_LAUNCH
    }
    GX_CHECK_LAUNCH("gx_icsbp_fwd");
    return GX_OK;
}

int gx_icsbp_fwd(const float* colour, const double* log_sigma, const float* rand_pixel, const int64_t* seed_idx_in,
                 int B, int C, int H, int W, int K, int kernel_type, float* log_m, float* log_s, float* seeds,
                 int64_t* seed_idx_out, gx_stream_t stream) {
    return icsbp_fwd_impl(colour, log_sigma, rand_pixel, seed_idx_in, B, C, H, W, K, kernel_type, log_m, log_s, seeds,
                          seed_idx_out, 0.f, nullptr, stream);
}

/* dynamic_K (modules/attention.py:218-219, models/genesisv2_config.py:118-137): an image stops at the first step whose
 * mask mass sum_p exp(log_m) is below min_mass; nsteps[b] = steps it ran (mask nsteps[b] = remaining scope, later
 * masks -1e10, later seeds 0) */
int gx_icsbp_fwd_dyn(const float* colour, const double* log_sigma, const float* rand_pixel, const int64_t* seed_idx_in,
                     int B, int C, int H, int W, int K, int kernel_type, float min_mass, float* log_m, float* log_s,
                     float* seeds, int64_t* seed_idx_out, int* nsteps, gx_stream_t stream) {
    GX_CHECK_ARG(nsteps && min_mass > 0.f, "gx_icsbp_fwd_dyn: nsteps null or min_mass <= 0");
    return icsbp_fwd_impl(colour, log_sigma, rand_pixel, seed_idx_in, B, C, H, W, K, kernel_type, log_m, log_s, seeds,
                          seed_idx_out, min_mass, nsteps, stream);
}

static int icsbp_bwd_threads(int HW) { return HW < 256 ? HW : 256; }

size_t gx_icsbp_bwd_ws_bytes(int B, int H, int W, int K) {
    const int HW = H * W;
    const int nch = gx_ceil_div(HW, icsbp_bwd_threads(HW));
    return ((size_t)B * nch * (K > 1 ? K - 1 : 1) * (MAXC + 1) + B) * sizeof(double);
}

static int icsbp_bwd_impl(const float* colour, const double* log_sigma, const float* seeds, const int64_t* seed_idx,
                          const float* g_log_m, int B, int C, int H, int W, int K, int kernel_type, float* dcolour,
                          double* dlog_sigma, void* ws, size_t ws_bytes, const int* nsteps, gx_stream_t stream) {
    GX_CHECK_ARG(colour && log_sigma && seeds && seed_idx && g_log_m && dcolour && dlog_sigma && ws,
                 "gx_icsbp_bwd: null pointer");
    const int HW = H * W;
    GX_CHECK_ARG(B > 0 && C > 0 && C <= MAXC && K >= 1 && K <= 17, "gx_icsbp_bwd: bad B/C/K (C<=8, K<=17)");
    GX_CHECK_ARG(gx_is_pow2(HW) && HW >= 64 && HW <= 1024 * MAXPPT, "gx_icsbp_bwd: H*W must be a power of two in [64,16384]");
    GX_CHECK_ARG(kernel_type >= 0 && kernel_type <= 2, "gx_icsbp_bwd: no valid kernel");
    GX_CHECK_ARG(ws_bytes >= gx_icsbp_bwd_ws_bytes(B, H, W, K), "gx_icsbp_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int T = icsbp_bwd_threads(HW);
    const int nch = gx_ceil_div(HW, T);
    double* part = (double*)ws;
    double* per_img = part + (size_t)B * nch * (K > 1 ? K - 1 : 1) * (MAXC + 1);
    {
        GxProf pf(KID_ICSBP_BWD, s, 0.0, 4.0 * B * HW * (2.0 * C + K));
        hipLaunchKernelGGL(icsbp_bwd_kernel, dim3(B, nch), dim3(T), 0, s, colour, log_sigma, seeds, g_log_m, B, C, HW,
                           K, kernel_type, dcolour, part, nsteps);
    }
    GX_CHECK_LAUNCH("gx_icsbp_bwd");
    {
        GxProf pf(KID_SMALL_REDUCE, s, 0.0, 8.0 * B * nch * K * (MAXC + 1));
        hipLaunchKernelGGL(icsbp_bwd_finalize_kernel, dim3(B), dim3(256), 0, s, (const double*)part, seed_idx,
                           log_sigma, B, C, HW, K, nch, dcolour, per_img);
        hipLaunchKernelGGL(sum_double_kernel, dim3(1), dim3(64), 0, s, (const double*)per_img, B, dlog_sigma);
    }
    GX_CHECK_LAUNCH("gx_icsbp_bwd(finalize)");
    return GX_OK;
}

int gx_icsbp_bwd(const float* colour, const double* log_sigma, const float* seeds, const int64_t* seed_idx,
                 const float* g_log_m, int B, int C, int H, int W, int K, int kernel_type, float* dcolour,
                 double* dlog_sigma, void* ws, size_t ws_bytes, gx_stream_t stream) {
    return icsbp_bwd_impl(colour, log_sigma, seeds, seed_idx, g_log_m, B, C, H, W, K, kernel_type, dcolour, dlog_sigma,
                          ws, ws_bytes, nullptr, stream);
}

int gx_icsbp_bwd_dyn(const float* colour, const double* log_sigma, const float* seeds, const int64_t* seed_idx,
                     const float* g_log_m, const int* nsteps, int B, int C, int H, int W, int K, int kernel_type,
                     float* dcolour, double* dlog_sigma, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(nsteps, "gx_icsbp_bwd_dyn: nsteps null");
    return icsbp_bwd_impl(colour, log_sigma, seeds, seed_idx, g_log_m, B, C, H, W, K, kernel_type, dcolour, dlog_sigma,
                          ws, ws_bytes, nsteps, stream);
}

}  // extern "C"
