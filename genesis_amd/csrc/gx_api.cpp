// Error plumbing + version for libgenesis_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "gx_common.h"

static thread_local char g_err[512] = "";

void gx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#include <atomic>
#include <mutex>

namespace {
thread_local int t_ctx = 0;
std::mutex g_ctx_mutex;
std::atomic<bool> g_ctx_alive[kGxMaxCtx] = {{true}};      // context 0 always exists (atomic: make_current reads it unlocked)
GxCtxFlags g_ctx_flags[kGxMaxCtx] = {};
struct CtxInit { CtxInit() { for (int i = 0; i < kGxMaxCtx; ++i) g_ctx_flags[i] = GxCtxFlags{false, false, -1, -1}; } } g_ctx_init;
}  // namespace
int gx_cur_ctx(void) { return t_ctx; }
namespace { thread_local GxAmaxLink t_amax_link = {nullptr, 0, 0, nullptr, 0, 0}; }
GxAmaxLink& gx_amax_link(void) { return t_amax_link; }
extern "C" int gx_kq_amax_link(float* parts, int capacity, size_t numel) {
    GxAmaxLink& L = t_amax_link;
    L.parts = (parts && capacity > 0 && numel > 0) ? parts : nullptr;
    L.capacity = L.parts ? capacity : 0;
    L.numel = L.parts ? numel : 0;
    L.tensor = nullptr; L.n = 0;
    return GX_OK;
}
extern "C" int gx_kq_amax_link_hits(void) { return t_amax_link.hits; }
namespace { thread_local GxAmaxTap t_amax_tap = {nullptr, 0, 0, 0}; }
GxAmaxTap& gx_amax_tap_state(void) { return t_amax_tap; }
extern "C" int gx_amax_tap(float* parts, int capacity, size_t numel) {
    t_amax_tap.parts = (parts && capacity > 0 && numel > 0) ? parts : nullptr;
    t_amax_tap.capacity = t_amax_tap.parts ? capacity : 0;
    t_amax_tap.numel = t_amax_tap.parts ? numel : 0;
    t_amax_tap.n = 0;
    return GX_OK;
}
extern "C" int gx_amax_parts(const float* x, size_t n, float* parts, gx_stream_t stream) {
    GX_CHECK_ARG(x && parts && n > 0, "gx_amax_parts: null pointer / empty tensor");
    return gx_kq_amax_launch(x, n, parts, (hipStream_t)stream);
}
extern "C" int gx_amax_tap_result(void) {
    const int n = t_amax_tap.parts ? 0 : t_amax_tap.n;      // (still armed: no producer served it)
    t_amax_tap.parts = nullptr; t_amax_tap.capacity = 0; t_amax_tap.n = 0; t_amax_tap.numel = 0;
    return n;
}
GxCtxFlags& gx_ctx_flags(void) { return g_ctx_flags[t_ctx]; }

namespace {
struct ProfEntry { int kid; hipEvent_t a, b; double flops, bytes; };
std::vector<ProfEntry> g_entries_ctx[kGxMaxCtx];
#define g_entries (g_entries_ctx[t_ctx])
const char* const kKernelNames[KID_COUNT] = {
    "tapconv_kernel<0>", "tapconv_kernel<1>", "tapconv_kernel<2>", "tapconv_kernel<3>", "pack_weights_kernel",
    "wgrad_kernel<0>", "wgrad_kernel<1>", "wgrad_kernel<2>", "wgrad_kernel<3>", "wgrad_kernel<4>",
    "wgrad_reduce_kernel", "gn_relu_fwd_kernel", "gn_relu_bwd_kernel", "gn_param_reduce_kernel",
    "icsbp_fwd_kernel", "icsbp_bwd_kernel", "maskpool_fwd_kernel", "maskpool_bwd_kernel",
    "mixture_kernel<false>", "mixture_kernel<true>", "conv1x1_fwd_kernel", "conv1x1_dgrad_kernel",
    "conv1x1_wgrad_kernel", "small_reduce_kernels", "adam_kernel", "geco_update_kernel", "splitk_reduce_kernel", "bias_act_bwd_kernel", "dconv_kernels", "gated_norm_kernels", "latent_kernels", "dense_kernel", "wino_conv_kernel",
    "wgq_stream_kernel", "kq_dth_kernel", "kq_dgh_kernel", "kq_c3h_kernel", "kq_c5h_kernel"};
}  // namespace

void gx_prof_begin(int kid, hipStream_t s, double flops, double bytes) {
    ProfEntry e;
    e.kid = kid; e.flops = flops; e.bytes = bytes;
    (void)hipEventCreate(&e.a);
    (void)hipEventCreate(&e.b);
    (void)hipEventRecord(e.a, s);
    g_entries.push_back(e);
}

void gx_prof_end(hipStream_t s) {
    if (!g_entries.empty()) (void)hipEventRecord(g_entries.back().b, s);
}

extern "C" {
int gx_profile_enable(int on) {
    g_gx_prof_on = on != 0;
    return GX_OK;
}

int gx_profile_num_kernels(void) { return KID_COUNT; }

const char* gx_profile_kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? kKernelNames[kid] : ""; }

/* Waits for the recorded events, ACCUMULATES per kernel id into the caller's arrays (each of length
 * gx_profile_num_kernels()) and clears the record list. */
int gx_profile_collect(double* total_ms, double* launches, double* flops, double* bytes) {
    GX_CHECK_ARG(total_ms && launches && flops && bytes, "gx_profile_collect: null pointer");
    for (ProfEntry& e : g_entries) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            total_ms[e.kid] += ms;
            launches[e.kid] += 1.0;
            flops[e.kid] += e.flops;
            bytes[e.kid] += e.bytes;
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    g_entries.clear();
    return GX_OK;
}

const char* gx_last_error(void) { return g_err; }
int gx_version(void) { return 1; }
}

// ---- deferred parameter-gradient reductions -------------------------------------------------------------
// Weight gradients and GroupNorm affine gradients only feed the optimiser, yet each of them ends in its own small
// reduce launch (32 per training step).  With deferral on, those entry points queue the reduce instead; one
// launch per kind (gx_defer_flush) finishes all of them after the backward pass.  The caller keeps the queued
// workspaces alive until the flush.
namespace {
constexpr int kMaxDefer = 128;     // (MONet's shared UNet queues 6 passes x 10 layers)
struct DeferQueues { GxWgradRed wq[kMaxDefer]; GxGnRed gq[kMaxDefer]; int nw = 0, ng = 0; };
DeferQueues g_defer[kGxMaxCtx];
#define g_wq (g_defer[t_ctx].wq)
#define g_gq (g_defer[t_ctx].gq)
#define g_nw (g_defer[t_ctx].nw)
#define g_ng (g_defer[t_ctx].ng)
}  // namespace
bool gx_defer_push_wgrad(const GxWgradRed& r) {
    if (g_nw >= kMaxDefer) return false;
    g_wq[g_nw++] = r;
    return true;
}
int gx_defer_wgrad_room() { return kMaxDefer - g_nw; }
bool gx_defer_push_gn(const GxGnRed& r) {
    if (g_ng >= kMaxDefer) return false;
    g_gq[g_ng++] = r;
    return true;
}

extern "C" {

/* contexts (see gx_common.h): create returns an id > 0, or a negative error code */
int gx_ctx_create(void) {
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    for (int i = 1; i < kGxMaxCtx; ++i)
        if (!g_ctx_alive[i]) {
            g_ctx_alive[i] = true;
            g_ctx_flags[i] = GxCtxFlags{false, false, -1, -1};
            g_defer[i].nw = g_defer[i].ng = 0;
            return i;
        }
    gx_set_error("gx_ctx_create: all %d contexts are in use", kGxMaxCtx);
    return GX_EINVAL;
}

int gx_ctx_make_current(int id) {
    GX_CHECK_ARG(id >= 0 && id < kGxMaxCtx && g_ctx_alive[id], "gx_ctx_make_current: bad context %d", id);
    t_ctx = id;
    return GX_OK;
}

int gx_ctx_current(void) { return t_ctx; }

int gx_ctx_destroy(int id) {
    GX_CHECK_ARG(id > 0 && id < kGxMaxCtx && g_ctx_alive[id], "gx_ctx_destroy: bad context %d", id);
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    const int prev = t_ctx;
    t_ctx = id;
    g_nw = 0; g_ng = 0; gx_wgq_discard(); gx_wf_discard();
    for (ProfEntry& e : g_entries) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    g_entries.clear();
    g_ctx_alive[id] = false;
    t_ctx = prev == id ? 0 : prev;
    return GX_OK;
}

int gx_defer_enable(int on) {
    g_gx_defer_on = on > 0;
    if (on < 0) { g_nw = 0; g_ng = 0; gx_wgq_discard(); gx_wf_discard(); }   // discard whatever is queued (error recovery)
    return GX_OK;
}

int gx_defer_pending(void) { return g_nw + g_ng + gx_wgq_pending() + gx_wf_pending(); }

int gx_defer_flush(gx_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    // queued weight-gradient jobs first: grouped launches, which queue their slab reductions below
    int rc = gx_wgq_flush(s);
    if (rc == GX_OK) rc = gx_wf_flush(s);          // queued small-layer launches, also ahead of their slab reductions
    if (rc == GX_OK && g_nw) rc = gx_defer_flush_wgrad(g_wq, g_nw, s);
    g_nw = 0;
    if (rc == GX_OK && g_ng) rc = gx_defer_flush_gn(g_gq, g_ng, s);
    g_ng = 0;
    return rc;
}

}  // extern "C"
