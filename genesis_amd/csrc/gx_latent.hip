// Slot-latent head of GENESIS / GENESIS-V2: reparameterised posterior sample, its log-density, and the
// log-density under the autoregressive prior -- the Monte-Carlo KL of Genesis.mask_latent_loss.
//
// Reference: models/genesisv2_config.py:154-160 (mu, sigma_ps = z_head(obj).chunk(2); sigma = to_sigma(sigma_ps);
// z = Normal(mu, sigma).rsample()), modules/blocks.py:36-41 (to_sigma = softplus(x + 0.5) + 1e-8,
// to_prior_sigma = sigmoid(x + 4.0) + 1e-4), models/genesis_config.py:288-343 (log_q, log_p per slot; first slot
// N(0,1), later slots N(tanh(lin[:D]), to_prior_sigma(lin[D:])) with lin = prior_linear(prior_lstm(z_{<k}))).
// Normal.log_prob(v) = -(v - loc)^2 / (2 scale^2) - log(scale) - log(sqrt(2 pi)), evaluated in that operation order.
//
// The reference spends ~35 pointwise launches forward and ~70 backward on these [K,B,D] tensors (7 x 32 x 64
// floats): pure launch latency.  Here: one wave per (slot, image) row of D latents, D-sums as wave reductions in
// fp64 (fixed tree), four launches per training step in total.
#include "gx_common.h"

namespace {

constexpr float kHalfLog2Pi = 0.91893853320467274178f;   // log(sqrt(2 pi))

__device__ __forceinline__ float softplus_t(float x) {   // F.softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float softplus_grad_t(float x) {
    if (x > 20.f) return 1.f;
    const float e = expf(x);
    return e / (e + 1.f);
}
__device__ __forceinline__ float sigmoid_t(float x) { return 1.f / (1.f + expf(-x)); }

// zh [B,K,2D] (mu | sigma_ps), eps [K,B,D] -> z, mu, sigma [K,B,D], log_q [K,B]
__global__ void __launch_bounds__(256)
posterior_fwd_kernel(const float* __restrict__ zh, const float* __restrict__ eps, int B, int K, int D,
                     float* __restrict__ z, float* __restrict__ mu_o, float* __restrict__ sigma_o,
                     float* __restrict__ log_q) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // k * B + b
    if (row >= K * B) return;
    const int lane = threadIdx.x & 63;
    const int k = row / B, b = row - k * B;
    const float* zr = zh + ((size_t)b * K + k) * 2 * D;
    double acc = 0.0;
    for (int d = lane; d < D; d += 64) {
        const float mu = zr[d];
        const float sg = softplus_t(zr[D + d] + 0.5f) + 1e-8f;
        const float zz = mu + sg * eps[(size_t)row * D + d];
        const float t = zz - mu;
        const float lp = -(t * t) / (2.f * (sg * sg)) - logf(sg) - kHalfLog2Pi;
        z[(size_t)row * D + d] = zz;
        mu_o[(size_t)row * D + d] = mu;
        sigma_o[(size_t)row * D + d] = sg;
        acc += (double)lp;
    }
    acc = gx_wave_sum_d(acc);
    if (lane == 0) log_q[row] = (float)acc;
}

// gradients: gz, gmu, gsigma [K,B,D], glogq [K,B] (any may be null = zero) -> dzh [B,K,2D]
__global__ void __launch_bounds__(256)
posterior_bwd_kernel(const float* __restrict__ zh, const float* __restrict__ eps, const float* __restrict__ gz,
                     const float* __restrict__ gmu, const float* __restrict__ gsigma,
                     const float* __restrict__ glogq, int B, int K, int D, float* __restrict__ dzh) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= K * B) return;
    const int lane = threadIdx.x & 63;
    const int k = row / B, b = row - k * B;
    const float* zr = zh + ((size_t)b * K + k) * 2 * D;
    float* dr = dzh + ((size_t)b * K + k) * 2 * D;
    const float gl = glogq ? glogq[row] : 0.f;
    for (int d = lane; d < D; d += 64) {
        const size_t i = (size_t)row * D + d;
        const float mu = zr[d];
        const float raw = zr[D + d] + 0.5f;
        const float sg = softplus_t(raw) + 1e-8f;
        const float e = eps[i];
        const float zz = mu + sg * e;
        const float t = zz - mu;
        const float var = sg * sg;
        // log_q as a function of (z, mu, sigma): d/dz = -t/var, d/dmu = +t/var, d/dsigma = t^2/sigma^3 - 1/sigma
        const float dz = (gz ? gz[i] : 0.f) + gl * (-t / var);
        const float dmu = dz + gl * (t / var) + (gmu ? gmu[i] : 0.f);
        const float dsg = dz * e + gl * ((t * t) / (var * sg) - 1.f / sg) + (gsigma ? gsigma[i] : 0.f);
        dr[d] = dmu;
        dr[D + d] = dsg * softplus_grad_t(raw);
    }
}

// z [K,B,D], lin [K-1,B,2D] (null: standard-normal prior for every slot) -> log_p [K,B]
__global__ void __launch_bounds__(256)
prior_logp_fwd_kernel(const float* __restrict__ z, const float* __restrict__ lin, int B, int K, int D,
                      float* __restrict__ log_p) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= K * B) return;
    const int lane = threadIdx.x & 63;
    const int k = row / B;
    const bool ar = lin != nullptr && k > 0;
    const float* lr = ar ? lin + (size_t)(row - B) * 2 * D : nullptr;
    double acc = 0.0;
    for (int d = lane; d < D; d += 64) {
        const float zz = z[(size_t)row * D + d];
        float lp;
        if (ar) {
            const float mu = tanhf(lr[d]);
            const float sg = sigmoid_t(lr[D + d] + 4.f) + 1e-4f;
            const float t = zz - mu;
            lp = -(t * t) / (2.f * (sg * sg)) - logf(sg) - kHalfLog2Pi;
        } else {
            lp = -(zz * zz) / 2.f - kHalfLog2Pi;
        }
        acc += (double)lp;
    }
    acc = gx_wave_sum_d(acc);
    if (lane == 0) log_p[row] = (float)acc;
}

__global__ void __launch_bounds__(256)
prior_logp_bwd_kernel(const float* __restrict__ z, const float* __restrict__ lin, const float* __restrict__ glogp,
                      int B, int K, int D, float* __restrict__ dz, float* __restrict__ dlin) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= K * B) return;
    const int lane = threadIdx.x & 63;
    const int k = row / B;
    const bool ar = lin != nullptr && k > 0;
    const float* lr = ar ? lin + (size_t)(row - B) * 2 * D : nullptr;
    float* dl = ar ? dlin + (size_t)(row - B) * 2 * D : nullptr;
    const float g = glogp[row];
    for (int d = lane; d < D; d += 64) {
        const size_t i = (size_t)row * D + d;
        const float zz = z[i];
        if (ar) {
            const float mu = tanhf(lr[d]);
            const float s = sigmoid_t(lr[D + d] + 4.f);
            const float sg = s + 1e-4f;
            const float t = zz - mu;
            const float var = sg * sg;
            dz[i] = g * (-t / var);
            dl[d] = g * (t / var) * (1.f - mu * mu);
            dl[D + d] = g * ((t * t) / (var * sg) - 1.f / sg) * (s * (1.f - s));
        } else {
            dz[i] = g * (-zz);
        }
    }
}

int check_bkd(const char* name, int B, int K, int D) {
    GX_CHECK_ARG(B > 0 && K > 0 && D > 0, "%s: bad B/K/D (%d,%d,%d)", name, B, K, D);
    return GX_OK;
}

}  // namespace

extern "C" {

int gx_latent_posterior_fwd(const float* zh, const float* eps, int B, int K, int D, float* z, float* mu,
                            float* sigma, float* log_q, gx_stream_t stream) {
    int rc = check_bkd("gx_latent_posterior_fwd", B, K, D);
    if (rc) return rc;
    GX_CHECK_ARG(zh && eps && z && mu && sigma && log_q, "gx_latent_posterior_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (6.0 * K * B * D + K * B));
        hipLaunchKernelGGL(posterior_fwd_kernel, dim3(gx_ceil_div(K * B, 4)), dim3(256), 0, s, zh, eps, B, K, D, z,
                           mu, sigma, log_q);
    }
    GX_CHECK_LAUNCH("gx_latent_posterior_fwd");
    return GX_OK;
}

int gx_latent_posterior_bwd(const float* zh, const float* eps, const float* gz, const float* gmu,
                            const float* gsigma, const float* glogq, int B, int K, int D, float* dzh,
                            gx_stream_t stream) {
    int rc = check_bkd("gx_latent_posterior_bwd", B, K, D);
    if (rc) return rc;
    GX_CHECK_ARG(zh && eps && dzh, "gx_latent_posterior_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (8.0 * K * B * D + K * B));
        hipLaunchKernelGGL(posterior_bwd_kernel, dim3(gx_ceil_div(K * B, 4)), dim3(256), 0, s, zh, eps, gz, gmu,
                           gsigma, glogq, B, K, D, dzh);
    }
    GX_CHECK_LAUNCH("gx_latent_posterior_bwd");
    return GX_OK;
}

int gx_latent_prior_logp_fwd(const float* z, const float* lin, int B, int K, int D, float* log_p,
                             gx_stream_t stream) {
    int rc = check_bkd("gx_latent_prior_logp_fwd", B, K, D);
    if (rc) return rc;
    GX_CHECK_ARG(z && log_p, "gx_latent_prior_logp_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (3.0 * K * B * D + K * B));
        hipLaunchKernelGGL(prior_logp_fwd_kernel, dim3(gx_ceil_div(K * B, 4)), dim3(256), 0, s, z, lin, B, K, D,
                           log_p);
    }
    GX_CHECK_LAUNCH("gx_latent_prior_logp_fwd");
    return GX_OK;
}

int gx_latent_prior_logp_bwd(const float* z, const float* lin, const float* glogp, int B, int K, int D, float* dz,
                             float* dlin, gx_stream_t stream) {
    int rc = check_bkd("gx_latent_prior_logp_bwd", B, K, D);
    if (rc) return rc;
    GX_CHECK_ARG(z && glogp && dz, "gx_latent_prior_logp_bwd: null pointer");
    GX_CHECK_ARG((lin == nullptr) == (dlin == nullptr), "gx_latent_prior_logp_bwd: lin and dlin go together");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (6.0 * K * B * D + K * B));
        hipLaunchKernelGGL(prior_logp_bwd_kernel, dim3(gx_ceil_div(K * B, 4)), dim3(256), 0, s, z, lin, glogp, B, K,
                           D, dz, dlin);
    }
    GX_CHECK_LAUNCH("gx_latent_prior_logp_bwd");
    return GX_OK;
}

}  // extern "C"
