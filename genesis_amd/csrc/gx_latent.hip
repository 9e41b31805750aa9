// Slot-latent head of GENESIS / GENESIS-V2: reparameterised posterior sample, its log-density, and the
// log-density under the autoregressive prior -- the Monte-Carlo KL of Genesis.mask_latent_loss.
//
// Reference: models/genesisv2_config.py:154-160 (mu, sigma_ps = z_head(obj).chunk(2); sigma = to_sigma(sigma_ps);
// z = Normal(mu, sigma).rsample()), modules/blocks.py:22-23,28-36 (to_sigma = softplus(x + 0.5) + 1e-8,
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
                     float* __restrict__ log_q, float* __restrict__ z2, int ldz2) {
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
        if (z2) z2[(size_t)row * ldz2 + d] = zz;      // second copy: columns of the next recurrent step's input rows
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
                     const float* __restrict__ glogq, int B, int K, int D, float* __restrict__ dzh,
                     const float* __restrict__ gz2, int ldgz2) {
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
        const float dz = (gz ? gz[i] : 0.f) + (gz2 ? gz2[(size_t)row * ldgz2 + d] : 0.f) + gl * (-t / var);
        const float dmu = dz + gl * (t / var) + (gmu ? gmu[i] : 0.f);
        const float dsg = dz * e + gl * ((t * t) / (var * sg) - 1.f / sg) + (gsigma ? gsigma[i] : 0.f);
        dr[d] = dmu;
        dr[D + d] = dsg * softplus_grad_t(raw);
    }
}

// z [K,B,D], lin [K-1,B,2D] (null: standard-normal prior for every slot) -> log_p [K,B]
__global__ void __launch_bounds__(256)
prior_logp_fwd_kernel(const float* __restrict__ z, const float* __restrict__ lin,
                      const float* __restrict__ log_q, int B, int K, int D, int first, float* __restrict__ log_p) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= K * B) return;
    const int lane = threadIdx.x & 63;
    const bool ar = lin != nullptr && row >= first;      // first = B: slot 0 under N(0,1); 0: every row has its own (mean, scale)
    const float* lr = ar ? lin + (size_t)(row - first) * 2 * D : nullptr;
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
    if (lane == 0) log_p[row] = log_q ? log_q[row] - (float)acc : (float)acc;   // KL sample log_q - log_p, or log_p
}

// ancestral sample of one slot from the AR prior (models/genesisv2_config.py:235-246): lin [B,2D] = prior_linear(h),
// z = tanh(lin[:D]) + (sigmoid(lin[D:] + 4) + 1e-4) * eps -- the same mean / scale arithmetic as the log-density above
__global__ void __launch_bounds__(256)
prior_sample_kernel(const float* __restrict__ lin, const float* __restrict__ eps, int B, int D, int tanh_mu,
                    float* __restrict__ z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    const float raw = lin[(size_t)b * 2 * D + d];
    const float mu = tanh_mu ? tanhf(raw) : raw;        // Genesis.sample's mask rollout keeps the raw mean (genesis_config.py:358)
    const float sg = sigmoid_t(lin[(size_t)b * 2 * D + D + d] + 4.f) + 1e-4f;
    z[i] = mu + sg * eps[i];
}

__global__ void __launch_bounds__(256)
prior_logp_bwd_kernel(const float* __restrict__ z, const float* __restrict__ lin, const float* __restrict__ glogp,
                      float sign, int B, int K, int D, int first, float* __restrict__ dz, float* __restrict__ dlin) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= K * B) return;
    const int lane = threadIdx.x & 63;
    const bool ar = lin != nullptr && row >= first;
    const float* lr = ar ? lin + (size_t)(row - first) * 2 * D : nullptr;
    float* dl = ar ? dlin + (size_t)(row - first) * 2 * D : nullptr;
    const float g = sign * glogp[row];
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

// ---- loss aggregation (train.py:226-242) + the GECO-weighted objective, one workgroup ----
// out = (loss = err_mean + beta kl_mean, elbo = err_mean + kl_mean, err_mean, kl_mean, beta)
__global__ void __launch_bounds__(256)
elbo_fwd_kernel(const float* __restrict__ err, const float* __restrict__ kl, const float* __restrict__ beta,
                int B, int R, float* __restrict__ out, float* __restrict__ tail, float* __restrict__ loss,
                float* __restrict__ d_err = nullptr, float* __restrict__ d_kl = nullptr) {
    __shared__ double red[2][4];
    if (d_err) {        // the objective's gradients for a unit upstream gradient (the training step's loss.backward())
        const float ge = 1.f / B, gk = *beta / B;
        for (int i = threadIdx.x; i < B; i += blockDim.x) d_err[i] = ge;
        if (d_kl) for (int i = threadIdx.x; i < R * B; i += blockDim.x) d_kl[i] = gk;
    }
    double se = 0.0, sk = 0.0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) se += (double)err[i];
    for (int i = threadIdx.x; i < R * B; i += blockDim.x) sk += (double)kl[i];
    se = gx_wave_sum_d(se); sk = gx_wave_sum_d(sk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = se; red[1][wave] = sk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double e = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / B;
        const double k = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / B;
        const float ef = (float)e, kf = (float)k, bt = *beta;
        out[0] = ef + bt * kf; out[1] = ef + kf; out[2] = ef; out[3] = kf; out[4] = bt;
        if (tail) { tail[0] = ef; tail[1] = kf; }
        if (loss) loss[0] = out[0];
    }
}

__global__ void __launch_bounds__(256)
elbo_bwd_kernel(const float* __restrict__ g, const float* __restrict__ beta, int B, int R,
                float* __restrict__ d_err, float* __restrict__ d_kl) {
    const float ge = g[0] / B, gk = g[0] * (*beta) / B;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) d_err[i] = ge;
    if (i < R * B) d_kl[i] = gk;
}

// ---- pooled slot features -> z_head's LayerNorm input (models/genesisv2_config.py:146-154, :76) ----
// obj = (lin + msum fbias) / (msum + 1e-5)  [feat_head[1] applied to the pooled sums, normalised by the mask
// mass], y = LayerNorm(obj; gamma, beta, eps).  One wave per row; row statistics in fp64.
__global__ void __launch_bounds__(256)
pooled_head_fwd_kernel(const float* __restrict__ lin, const float* __restrict__ msum,
                       const float* __restrict__ fbias, const float* __restrict__ gamma,
                       const float* __restrict__ beta, float eps, int R, int C, float* __restrict__ y,
                       float* __restrict__ stats /* [R][2] mean, rstd */) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const float m = msum[row];
    const float den = m + 1e-5f;
    double s1 = 0.0, s2 = 0.0;
    for (int j = lane; j < C; j += 64) {
        const float o = (lin[(size_t)row * C + j] + m * fbias[j]) / den;
        s1 += (double)o; s2 += (double)o * o;
    }
    s1 = gx_wave_sum_d(s1); s2 = gx_wave_sum_d(s2);
    const double mean = s1 / C;
    double var = s2 / C - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int j = lane; j < C; j += 64) {
        const float o = (lin[(size_t)row * C + j] + m * fbias[j]) / den;
        y[(size_t)row * C + j] = (o - meanf) * rstd * gamma[j] + beta[j];
    }
    if (lane == 0) { stats[2 * row] = meanf; stats[2 * row + 1] = rstd; }
}

// g [R,C] -> dlin [R,C], dmsum [R]; per-row column terms cols[R][3][C] = (g xhat, g, dobj m/den) for the
// parameter gradients (summed over rows by pooled_head_cols_kernel in a fixed order)
__global__ void __launch_bounds__(256)
pooled_head_bwd_kernel(const float* __restrict__ lin, const float* __restrict__ msum,
                       const float* __restrict__ fbias, const float* __restrict__ gamma,
                       const float* __restrict__ stats, const float* __restrict__ g, int R, int C,
                       float* __restrict__ dlin, float* __restrict__ dmsum, float* __restrict__ cols) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const float m = msum[row];
    const float den = m + 1e-5f;
    const float meanf = stats[2 * row], rstd = stats[2 * row + 1];
    double a = 0.0, b = 0.0;   // sum dxhat, sum dxhat * xhat
    for (int j = lane; j < C; j += 64) {
        const float o = (lin[(size_t)row * C + j] + m * fbias[j]) / den;
        const float xh = (o - meanf) * rstd;
        const float dxh = g[(size_t)row * C + j] * gamma[j];
        a += (double)dxh; b += (double)dxh * xh;
    }
    a = gx_wave_sum_d(a); b = gx_wave_sum_d(b);
    const float k1 = (float)(a / C), k2 = (float)(b / C);
    double dm = 0.0;
    for (int j = lane; j < C; j += 64) {
        const float num = lin[(size_t)row * C + j] + m * fbias[j];
        const float o = num / den;
        const float xh = (o - meanf) * rstd;
        const float gv = g[(size_t)row * C + j];
        const float dobj = rstd * (gv * gamma[j] - k1 - xh * k2);
        dlin[(size_t)row * C + j] = dobj / den;
        dm += (double)dobj * ((double)fbias[j] / den - (double)num / ((double)den * den));
        float* cr = cols + (size_t)row * 3 * C;
        cr[j] = gv * xh; cr[C + j] = gv; cr[2 * C + j] = dobj * (m / den);
    }
    dm = gx_wave_sum_d(dm);
    if (lane == 0) dmsum[row] = (float)dm;
}

// 32 columns x 8 row groups per block; row groups are combined through LDS in a fixed order
__global__ void __launch_bounds__(256)
pooled_head_cols_kernel(const float* __restrict__ cols, int R, int C, float* __restrict__ dgamma,
                        float* __restrict__ dbeta, float* __restrict__ dfbias) {
    __shared__ double red[8][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;   // over 3*C
    double s = 0.0;
    if (i < 3 * C)
        for (int r = rg; r < R; r += 8) s += (double)cols[(size_t)r * 3 * C + i];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && i < 3 * C) {
        double t = red[0][cl];
        for (int q = 1; q < 8; ++q) t += red[q][cl];
        const int which = i / C, j = i - which * C;
        (which == 0 ? dgamma : (which == 1 ? dbeta : dfbias))[j] = (float)t;
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
    return gx_latent_posterior_fwd_ex(zh, eps, B, K, D, z, mu, sigma, log_q, nullptr, 0, stream);
}

int gx_latent_posterior_fwd_ex(const float* zh, const float* eps, int B, int K, int D, float* z, float* mu,
                               float* sigma, float* log_q, float* z2, int ldz2, gx_stream_t stream) {
    int rc = check_bkd("gx_latent_posterior_fwd", B, K, D);
    if (rc) return rc;
    GX_CHECK_ARG(zh && eps && z && mu && sigma && log_q, "gx_latent_posterior_fwd: null pointer");
    GX_CHECK_ARG(!z2 || ldz2 >= D, "gx_latent_posterior_fwd_ex: ldz2 (%d) < D (%d)", ldz2, D);
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (6.0 * K * B * D + K * B));
        hipLaunchKernelGGL(posterior_fwd_kernel, dim3(gx_ceil_div(K * B, 4)), dim3(256), 0, s, zh, eps, B, K, D, z,
                           mu, sigma, log_q, z2, ldz2);
    }
    GX_CHECK_LAUNCH("gx_latent_posterior_fwd");
    return GX_OK;
}

int gx_latent_posterior_bwd(const float* zh, const float* eps, const float* gz, const float* gmu,
                            const float* gsigma, const float* glogq, int B, int K, int D, float* dzh,
                            gx_stream_t stream) {
    return gx_latent_posterior_bwd_ex(zh, eps, gz, gmu, gsigma, glogq, nullptr, 0, B, K, D, dzh, stream);
}

int gx_latent_posterior_bwd_ex(const float* zh, const float* eps, const float* gz, const float* gmu,
                               const float* gsigma, const float* glogq, const float* gz2, int ldgz2, int B, int K,
                               int D, float* dzh, gx_stream_t stream) {
    int rc = check_bkd("gx_latent_posterior_bwd", B, K, D);
    if (rc) return rc;
    GX_CHECK_ARG(zh && eps && dzh, "gx_latent_posterior_bwd: null pointer");
    GX_CHECK_ARG(!gz2 || ldgz2 >= D, "gx_latent_posterior_bwd_ex: ldgz2 (%d) < D (%d)", ldgz2, D);
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (8.0 * K * B * D + K * B));
        hipLaunchKernelGGL(posterior_bwd_kernel, dim3(gx_ceil_div(K * B, 4)), dim3(256), 0, s, zh, eps, gz, gmu,
                           gsigma, glogq, B, K, D, dzh, gz2, ldgz2);
    }
    GX_CHECK_LAUNCH("gx_latent_posterior_bwd");
    return GX_OK;
}

int gx_latent_prior_logp_fwd(const float* z, const float* lin, const float* log_q, int B, int K, int D,
                             float* out, gx_stream_t stream) {
    return gx_latent_prior_logp_fwd_ex(z, lin, log_q, B, K, D, 0, out, stream);
}

int gx_latent_prior_logp_fwd_ex(const float* z, const float* lin, const float* log_q, int B, int K, int D, int all_slots,
                                float* out, gx_stream_t stream) {
    float* log_p = out;
    int rc = check_bkd("gx_latent_prior_logp_fwd", B, K, D);
    if (rc) return rc;
    GX_CHECK_ARG(z && log_p, "gx_latent_prior_logp_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (3.0 * K * B * D + K * B));
        hipLaunchKernelGGL(prior_logp_fwd_kernel, dim3(gx_ceil_div(K * B, 4)), dim3(256), 0, s, z, lin, log_q, B, K,
                           D, all_slots ? 0 : B, log_p);
    }
    GX_CHECK_LAUNCH("gx_latent_prior_logp_fwd");
    return GX_OK;
}

int gx_latent_prior_sample(const float* lin, const float* eps, int B, int D, float* z, gx_stream_t stream) {
    return gx_latent_prior_sample_ex(lin, eps, B, D, 1, z, stream);
}

int gx_latent_prior_sample_ex(const float* lin, const float* eps, int B, int D, int tanh_mu, float* z,
                              gx_stream_t stream) {
    GX_CHECK_ARG(lin && eps && z && B > 0 && D > 0, "gx_latent_prior_sample: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 16.0 * B * D);
        hipLaunchKernelGGL(prior_sample_kernel, dim3(gx_ceil_div(B * D, 256)), dim3(256), 0, s, lin, eps, B, D, tanh_mu,
                           z);
    }
    GX_CHECK_LAUNCH("gx_latent_prior_sample");
    return GX_OK;
}

int gx_latent_prior_logp_bwd(const float* z, const float* lin, const float* g_out, int kl_mode, int B, int K,
                             int D, float* dz, float* dlin, gx_stream_t stream) {
    return gx_latent_prior_logp_bwd_ex(z, lin, g_out, kl_mode, B, K, D, 0, dz, dlin, stream);
}

int gx_latent_prior_logp_bwd_ex(const float* z, const float* lin, const float* g_out, int kl_mode, int B, int K,
                                int D, int all_slots, float* dz, float* dlin, gx_stream_t stream) {
    const float* glogp = g_out;
    int rc = check_bkd("gx_latent_prior_logp_bwd", B, K, D);
    if (rc) return rc;
    GX_CHECK_ARG(z && glogp && dz, "gx_latent_prior_logp_bwd: null pointer");
    GX_CHECK_ARG((lin == nullptr) == (dlin == nullptr), "gx_latent_prior_logp_bwd: lin and dlin go together");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (6.0 * K * B * D + K * B));
        hipLaunchKernelGGL(prior_logp_bwd_kernel, dim3(gx_ceil_div(K * B, 4)), dim3(256), 0, s, z, lin, glogp,
                           kl_mode ? -1.f : 1.f, B, K, D, all_slots ? 0 : B, dz, dlin);
    }
    GX_CHECK_LAUNCH("gx_latent_prior_logp_bwd");
    return GX_OK;
}

int gx_elbo_fwd(const float* err, const float* kl, const float* beta, int B, int R, float* out, float* tail,
                float* loss, gx_stream_t stream) {
    GX_CHECK_ARG(err && beta && out, "gx_elbo_fwd: null pointer");
    GX_CHECK_ARG(B > 0 && R >= 0 && (R == 0 || kl), "gx_elbo_fwd: bad B/R (%d,%d) or missing kl", B, R);
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (B + (double)R * B));
        hipLaunchKernelGGL(elbo_fwd_kernel, dim3(1), dim3(256), 0, s, err, kl, beta, B, R, out, tail, loss);
    }
    GX_CHECK_LAUNCH("gx_elbo_fwd");
    return GX_OK;
}

/* gx_elbo_fwd that also writes d loss / d err [B] = 1 / B and d loss / d kl [R,B] = beta / B (the gradients gx_elbo_bwd gives for
 * an upstream gradient of one): a training step starts its backward pass from them and needs no second launch. */
int gx_elbo_fwd_grads(const float* err, const float* kl, const float* beta, int B, int R, float* out, float* tail,
                      float* loss, float* d_err, float* d_kl, gx_stream_t stream) {
    GX_CHECK_ARG(err && beta && out && d_err, "gx_elbo_fwd_grads: null pointer");
    GX_CHECK_ARG(B > 0 && R >= 0 && (R == 0 || (kl && d_kl)), "gx_elbo_fwd_grads: bad B/R (%d,%d) or missing kl / d_kl", B, R);
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 8.0 * (B + (double)R * B));
        hipLaunchKernelGGL(elbo_fwd_kernel, dim3(1), dim3(256), 0, s, err, kl, beta, B, R, out, tail, loss, d_err, d_kl);
    }
    GX_CHECK_LAUNCH("gx_elbo_fwd_grads");
    return GX_OK;
}

int gx_elbo_bwd(const float* g_loss, const float* beta, int B, int R, float* d_err, float* d_kl,
                gx_stream_t stream) {
    GX_CHECK_ARG(g_loss && beta && d_err, "gx_elbo_bwd: null pointer");
    GX_CHECK_ARG(B > 0 && R >= 0 && (R == 0 || d_kl), "gx_elbo_bwd: bad B/R (%d,%d) or missing d_kl", B, R);
    hipStream_t s = (hipStream_t)stream;
    const int n = R > 0 ? R * B : B;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * (B + (double)R * B));
        hipLaunchKernelGGL(elbo_bwd_kernel, dim3(gx_ceil_div(n, 256)), dim3(256), 0, s, g_loss, beta, B, R, d_err,
                           d_kl);
    }
    GX_CHECK_LAUNCH("gx_elbo_bwd");
    return GX_OK;
}

int gx_pooled_head_fwd(const float* lin, const float* msum, const float* fbias, const float* gamma,
                       const float* beta, float eps, int R, int C, float* y, float* stats, gx_stream_t stream) {
    GX_CHECK_ARG(lin && msum && fbias && gamma && beta && y && stats, "gx_pooled_head_fwd: null pointer");
    GX_CHECK_ARG(R > 0 && C > 0, "gx_pooled_head_fwd: bad R/C (%d,%d)", R, C);
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 8.0 * R * C);
        hipLaunchKernelGGL(pooled_head_fwd_kernel, dim3(gx_ceil_div(R, 4)), dim3(256), 0, s, lin, msum, fbias, gamma,
                           beta, eps, R, C, y, stats);
    }
    GX_CHECK_LAUNCH("gx_pooled_head_fwd");
    return GX_OK;
}

size_t gx_pooled_head_bwd_ws_bytes(int R, int C) { return (size_t)R * 3 * C * sizeof(float); }

int gx_pooled_head_bwd(const float* lin, const float* msum, const float* fbias, const float* gamma,
                       const float* stats, const float* g, int R, int C, float* dlin, float* dmsum, float* dfbias,
                       float* dgamma, float* dbeta, void* ws, size_t ws_bytes, gx_stream_t stream) {
    GX_CHECK_ARG(lin && msum && fbias && gamma && stats && g && dlin && dmsum && dfbias && dgamma && dbeta && ws,
                 "gx_pooled_head_bwd: null pointer");
    GX_CHECK_ARG(R > 0 && C > 0, "gx_pooled_head_bwd: bad R/C (%d,%d)", R, C);
    GX_CHECK_ARG(ws_bytes >= gx_pooled_head_bwd_ws_bytes(R, C), "gx_pooled_head_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * 6.0 * R * C);
        hipLaunchKernelGGL(pooled_head_bwd_kernel, dim3(gx_ceil_div(R, 4)), dim3(256), 0, s, lin, msum, fbias, gamma,
                           stats, g, R, C, dlin, dmsum, (float*)ws);
    }
    GX_CHECK_LAUNCH("gx_pooled_head_bwd");
    {
        GxProf pf(KID_LATENT, s, 0.0, 4.0 * 3.0 * R * C);
        hipLaunchKernelGGL(pooled_head_cols_kernel, dim3(gx_ceil_div(3 * C, 32)), dim3(256), 0, s,
                           (const float*)ws, R, C, dgamma, dbeta, dfbias);
    }
    GX_CHECK_LAUNCH("gx_pooled_head_bwd(cols)");
    return GX_OK;
}

}  // extern "C"
