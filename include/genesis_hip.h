/* libgenesis_hip.so -- C ABI of the MI355X-native GENESIS-V2 hot path.
 *
 * The reference (applied-ai-lab/genesis) is pure Python/PyTorch: it has NO native/FFI
 * boundary.  The drop-in boundary is therefore the reference's Python interface
 * (models/genesisv2_config.py:45-46 `load(cfg)`, :110-203 `forward(x)`), mirrored by
 * genesis_amd/genesisv2_config.py; this library sits UNDER that module and each entry
 * point replaces the ATen op group the reference executes at the cited file:line.
 *
 * Conventions:
 *   - arguments are raw device pointers + explicit sizes; tensors are contiguous NCHW fp32;
 *   - the caller (PyTorch caching allocator) owns every buffer, including workspaces, whose
 *     required size is returned by the matching *_ws_bytes() query;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t); no allocation, no sync;
 *   - return 0 on success, negative GX_E* on error (gx_last_error() gives the message,
 *     thread-local); nothing throws across the boundary.
 */
#ifndef GENESIS_HIP_H
#define GENESIS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gx_stream_t; /* hipStream_t */

#define GX_OK 0
#define GX_EINVAL (-1)  /* bad argument / unsupported shape */
#define GX_ELAUNCH (-2) /* kernel launch failed */

const char* gx_last_error(void);
int gx_version(void);

/* ---- conv3x3 stride 1 pad 1, no bias: modules/blocks.py:159-165 (ConvGNReLU[0]);
 *      UNet blocks modules/unet.py:53-56, seg_head/feat_head models/genesisv2_config.py:78-80.
 *      w is the nn.Conv2d weight [Cout,Cin,3,3]. fwd/dgrad repack w into `ws`. */
size_t gx_conv3x3_ws_bytes(int N, int Cin, int Cout, int H, int W);
int gx_conv3x3_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout, int H, int W,
                   void* ws, size_t ws_bytes, gx_stream_t stream);
int gx_conv3x3_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W,
                     void* ws, size_t ws_bytes, gx_stream_t stream);
int gx_conv3x3_dgrad_parts(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, void* ws,
                           size_t ws_bytes, const float** parts, int* nsplit, size_t* split_stride, gx_stream_t stream);
size_t gx_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W);
int gx_conv3x3_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W,
                     void* ws, size_t ws_bytes, gx_stream_t stream);

/* ---- the same 3x3 convolution by Winograd F(2x2,3x3) (2.25x fewer multiplies; fp32, ordinary rounding differences
 *      against the direct sum).  mode 0: forward y = conv(x, w); mode 1: data gradient dx from dy (same w [Cout,Cin,3,3]).
 *      Eligible shapes: gx_conv3x3_wino_supported (H % 8 == 0, W % 16 == 0, >= 16 channels). */
int gx_conv3x3_wino_supported(int N, int Cin, int Cout, int H, int W);
/*      which layers gx_conv3x3_fwd / _dgrad send to it: 0 none, 1 those whose grid fills the chip (default; also
 *      GENESIS_WINOGRAD=0/1/2 in the environment), 2 every supported shape. */
int gx_conv3x3_wino_policy(int mode);
/*      which matrix pipe the Winograd layers' products run on: 1 (default; GENESIS_WINO_BF16X6=0/1 in the environment) = the
 *      bf16 pipe, every fp32 product U * V from six bf16 piece products accumulated in fp32 (hi + mid + lo pieces hold all 24
 *      mantissa bits: fp32 accuracy, tests/test_kernels_gpu.py::test_conv3x3_winograd); 0 = v_mfma_f32_32x32x2_f32.  Packed
 *      operands are laid out for the pipe in force when they were packed: switch between iterations, not inside one. */
int gx_wino_precision(int mode);
/*      k-quad tap-conv kernels (gx_kq.hip: 16-byte k-contiguous MFMA operand reads) behind gx_conv3x3_fwd / _dgrad and
 *      gx_deconv5x5s2_fwd / _dgrad: 0 never, 1 layers whose grid fills the chip (default; GENESIS_KQ=0/1/2 in the
 *      environment), 2 every eligible shape (power-of-two grids, reduction channels a multiple of 8). */
int gx_kq_policy(int mode);
/*      weight gradients of layers of width >= 8 (gx_wgq.hip): 1 (default; GENESIS_WGQ=0/1, GENESIS_WGQ_STREAM=0/1) =
 *      LDS-DMA staged kernels, every queued layer in ONE stream-K launch at gx_defer_flush; 2 = the same kernels, one
 *      launch per (tap class, tile width); 0 = the round-1 kernels everywhere. */
int gx_wgq_policy(int mode);
/*      Which matrix pipe the LDS-DMA weight-gradient kernels multiply on.  1 (default): the bf16 pipe -- every fp32
 *      operand is split into three bf16 pieces (hi + mid + lo = all 24 mantissa bits) and a product is the six piece
 *      products of order <= 2, accumulated in fp32 (v_mfma_f32_32x32x16_bf16): the error against fp64 is that of the
 *      fp32 pipe (tools/bf16x6_probe.hip: 4.8e-7 vs 5.8e-7 relative L2; tests/test_kernels_gpu.py compares both pipes
 *      with fp64), at 6 x 32 instead of 8 x 64 matrix-pipe cycles per 16 contraction steps.  0: the fp32 pipe
 *      (v_mfma_f32_32x32x2_f32).  Environment: GENESIS_WGQ_BF16X6=0. */
int gx_wgq_precision(int mode);
/*      Tile shape of the bf16-pipe weight gradients for layers whose base rows are 32 or 64 pixels wide (conv3x3 at
 *      32 / 64, transposed conv from 32 x 32): 1 (default) row-ring tiles -- a tile is one full-width base row, the x rows
 *      roll through a four-slot LDS ring, every operand value is split into its bf16 planes once on its way into LDS and
 *      the taps' column shifts are funnel shifts of the dy operand -- 0 the 64-pixel LDS-DMA tiles whose lanes split
 *      what they read (the A/B reference).  Environment: GENESIS_WGQ_RING=0.  Replaces the same reference ops as
 *      gx_conv3x3_wgrad / gx_deconv5x5s2_wgrad (modules/blocks.py:159-165, models/genesisv2_config.py:89-99). */
/*      Mode 2 of gx_wgq_precision (the default since round 6; GENESIS_WGQ_F16X3=0: mode 1): the row-ring tiles form every fp32
 *      product from THREE fp16 piece products (x * 2^e = hi + lo, one power-of-two scale per operand TENSOR) instead of six
 *      bf16 ones -- for the layers whose operands' largest magnitudes are known without a pass over them.
 *      gx_wgq_operand_amax(...) is the one-shot, per-thread hint the NEXT gx_conv3x3_wgrad / gx_deconv5x5s2_wgrad call takes:
 *      partial maxima of dy in a0[0..na0) (+ a1[0..na1)), of x in b0[0..nb0) (+ b1[0..nb1): a concat buffer written by two
 *      producers), as left by gx_amax_tap / gx_amax_parts, and out2 -- two floats that receive {max |dy|, max |x|} in a small
 *      launch ahead of the stream-K launch.  All of them stay alive until the queued launch has run.  Layers without the hint
 *      (or with partial maxima of only one operand) run on bf16 pieces.  Range: a value below max|tensor| * 2^-18 keeps fewer
 *      than 22 bits (its low piece is an fp16 subnormal), below max|tensor| * 2^-40 it is flushed -- absolute accuracy
 *      2^-40 max|tensor|; tests/test_kernels_gpu.py *fp16x3* bound the error per output channel against fp64. */
int gx_wgq_operand_amax(const float* a0, int na0, const float* a1, int na1, const float* b0, int nb0, const float* b1,
                        int nb1, float* out2);
/*      Measurement: the share of the LAST stream-K launch's algorithmic flops that ran on three fp16 piece products (the rest:
 *      six bf16 ones) -- what bench.py prices the launch's matrix-pipe ceiling with. */
double gx_wgq_last_f16_share(void);
int gx_wgq_ring(int on);
/*      The same choice for the chip-filling transposed-conv forward / data-gradient layers (gx_kq.hip): 1
 *      bf16 pipe -- the staging splits the input tile into its three bf16 planes, the pack kernel the weights; needs a
 *      multiple of 16 reduction channels and a tile of <= 384 halo positions, other layers stay on the fp32 pipe --
 *      0 fp32 pipe.  Environment: GENESIS_KQ_BF16X6=0.  2: as 1, the transposed-conv forward / data gradient from THREE fp16
 *      piece products instead (x * 2^sx = hi + lo, 22 significant bits; hi*hi + hi*lo + lo*hi), one power-of-two scale per
 *      tensor from its largest magnitude -- the input's by one small launch ahead of the conv (partial maxima at the end of the
 *      conv's workspace), the weights' at pack time; error against fp64 at or below the bf16 form's on every operand set of the
 *      tests, two thirds of its matrix-pipe time: the DEFAULT since round 5 (GENESIS_KQ_F16X3=0: mode 1).  -1: back to the
 *      environment's default.
 *      RANGE of mode 2 (the same holds for gx_wgq_precision(2) and gx_wino_precision(2)): the scale is ONE power of two per tensor,
 *      max |x| * 2^e in [2^14, 2^15).  A value keeps its 22 significant bits down to max |x| * 2^-18; below that its low piece is
 *      an fp16 subnormal and bits drop off one by one; below max |x| * 2^-28 the high piece is subnormal too and below 2^-40 the
 *      value is flushed -- i.e. the representation error is max(2^-23 |x|, 2^-40 max |x|): absolute, not relative, accuracy for the
 *      smallest values of a tensor with one extreme outlier (an element 1e9 times the rest: tests/test_kernels_gpu.py
 *      ::test_fp16x3_transposed_conv_scales_follow_the_tensors[one_huge] pins exactly that bound).  Mode 1 has fp32's range. */
int gx_kq_precision(int mode);
/*      Mode 2's per-tensor maximum without a second pass over the tensor: gx_kq_amax_link(parts, capacity, numel) arms a one-shot,
 *      per-thread hand-over -- the next producer that supports it (the register-resident GroupNorm + ReLU kernels behind gx_gn_relu_fwd_parts /
 *      gx_gn_relu_bwd_parts and the decoder head's backward behind gx_gn_relu_bwd_proj: models/genesisv2_config.py:90-99) writes one partial maximum of the gradient it stores
 *      per workgroup into `parts` (<= capacity floats) -- only a launch that covers all `numel` elements of the tensor: a chunked
 *      producer leaves the link alone -- and remembers that tensor's address; the next mode-2 conv call whose input IS
 *      that address and size (gx_deconv5x5s2_fwd* / gx_deconv5x5s2_dgrad / gx_conv5x5s1) reduces those partials instead of launching its own
 *      pass (more than 1024 of them: one small launch folds them first).  Producers: the GroupNorm + ReLU kernels (every form since
 *      round 6, the chunked decoder-head backward's apply kernel included) and the gated units' apply kernels (gx_gated_norm_fwd: out;
 *      gx_gated_norm_bwd: dy).  Any mode-2 conv call clears
 *      the link; results are bit-identical with and without it.  gx_kq_amax_link_hits(): hand-overs taken so far (this thread). */
int gx_kq_amax_link(float* parts, int capacity, size_t numel);
int gx_kq_amax_link_hits(void);
/*      The same partial maxima for a LATER reader (the weight gradients' stream-K launch at the end of the backward pass,
 *      gx_wgq_operand_amax below): gx_amax_tap(parts, capacity, numel) arms a one-shot, per-thread request -- the next producer launch
 *      that supports it (the GroupNorm + ReLU kernels, forward and backward; the gated units' apply kernels; the BroadcastDecoder
 *      chain's producers -- gx_bcast_conv3x3_fwd, the <= 32-output-channel conv3x3 of gx_conv3x3_bias_act_fwd / gx_conv3x3_dgrad_act /
 *      gx_conv3x3_dgrad in mode 2 of gx_kq_precision, gx_conv1x1_bwd_act's data gradient) writes one partial maximum of the
 *      values it stores per workgroup into parts[0 .. n) -- provided it stores exactly `numel` values: a chunked producer does not
 *      serve -- whatever its destination views are (the channel slice of a concat
 *      buffer and a resampled second copy hold the same values); gx_amax_tap_result() returns n (0: that launch could not
 *      serve, the request is void) and disarms.  The caller keeps `parts` alive until its reader has RUN.
 *      gx_amax_parts(x, n, parts, stream): the partial maxima of any tensor by a pass of its own -- 256 floats. */
int gx_amax_tap(float* parts, int capacity, size_t numel);
/*      The consumer side for the Winograd conv3x3 layers (gx_wino.hip; modules/blocks.py:159-165 forward and data gradient):
 *      gx_conv_input_amax(p0, n0, p1, n1) is the one-shot, per-thread hint that the INPUT tensor of the next gx_conv3x3_fwd* /
 *      gx_conv3x3_dgrad* / gx_conv3x3_pair_* / gx_conv3x3_wino call has its partial maxima in p0[0 .. n0) (+ p1[0 .. n1): a
 *      concat buffer written by two producers, or the pair data gradient's second tensor).  With it -- and gx_wino_precision(2),
 *      the default; GENESIS_WINO_F16X3=0: mode 1 -- a layer that takes the Winograd kernel forms every fp32 product from THREE
 *      fp16 piece products (U * 2^eU packed as two pieces, eU from max |w|; V * 2^eV split in registers, eV from 4 max |x|)
 *      instead of six bf16 ones -- up to 1536 partial maxima (every workgroup of that kernel reduces them itself; more: six bf16
 *      pieces).  A layer with <= 32 output channels (gx_kq.hip, mode 2 of gx_kq_precision) reads its input's scale from the same
 *      hint instead of making its own amax pass, any count.  Layers on other kernels ignore it.  NULL / 0 clears.  Same range note
 *      as gx_wgq_operand_amax. */
int gx_conv_input_amax(const float* p0, int n0, const float* p1, int n1);
/*      Measurement: the share of the bf16-pipe Winograd launches' algorithmic flops (since the process started) on fp16 pieces. */
double gx_wino_f16_share(void);
int gx_amax_tap_result(void);
int gx_amax_parts(const float* x, size_t n, float* parts, gx_stream_t stream);
size_t gx_conv3x3_wino_ws_bytes(int N, int Cin, int Cout, int H, int W);
int gx_conv3x3_wino(const float* x, const float* w, float* y, int N, int Cin, int Cout, int H, int W, int mode,
                    void* ws, size_t ws_bytes, gx_stream_t stream);
/*      Two conv3x3 layers on ONE input as one layer (seg_head and feat_head[0] on the encoder features,
 *      models/genesisv2_config.py:70-73, :110-149): forward = one launch that writes y1 [N,Co1,H,W] = conv3x3(x, w1)
 *      and y2 [N,Co2,H,W] = conv3x3(x, w2); data gradient = one launch dx = dgrad(dy1, w1) + dgrad(dy2, w2) (the sum of
 *      the two consumers' input gradients forms in the accumulators: no second dgrad, no accumulation pass).
 *      Winograd kernel only: gx_conv3x3_pair_supported (Co1 % 64 == 0, gx_conv3x3_wino's shape rules).
 *      ws keeps the packed weights of both directions; dgrad with pack = 0 reuses what the forward of the same
 *      iteration packed. */
int gx_conv3x3_pair_supported(int N, int Cin, int Co1, int Co2, int H, int W);
size_t gx_conv3x3_pair_ws_bytes(int N, int Cin, int Co1, int Co2, int H, int W);
int gx_conv3x3_pair_fwd(const float* x, const float* w1, const float* w2, float* y1, float* y2, int N, int Cin, int Co1,
                        int Co2, int H, int W, void* ws, size_t ws_bytes, gx_stream_t stream);
int gx_conv3x3_pair_dgrad(const float* dy1, const float* dy2, const float* w1, const float* w2, float* dx, int N, int Cin,
                          int Co1, int Co2, int H, int W, int pack, void* ws, size_t ws_bytes, gx_stream_t stream);

/* ---- ConvTranspose2d(k=5, s=2, p=2, output_padding=1) + bias:
 *      models/genesisv2_config.py:90-98 (decoder_module.{1,4,7,10}).
 *      w is the nn.ConvTranspose2d weight [Cin,Cout,5,5]; x [N,Cin,Hin,Win]; y [N,Cout,2Hin,2Win].
 *      dgrad writes only the first Cin_out channels of dx ([N,Cin_out,Hin,Win]). */
size_t gx_deconv5x5s2_ws_bytes(int N, int Cin, int Cout, int Hin, int Win);
int gx_deconv5x5s2_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout,
                       int Hin, int Win, void* ws, size_t ws_bytes, gx_stream_t stream);
/*      Forward + the GroupNorm statistics of its output (mean, rstd [N*groups], as gx_gn_relu_fwd with dst0 == NULL
 *      would compute them) without a pass over y: the conv epilogue sums every 8-channel block per workgroup, a tiny
 *      kernel finishes in fp64.  *fused = 0 (shape not eligible: channel split, several images per tile, groups not
 *      multiples of 8 channels): y is written, mean / rstd are NOT -- run gx_gn_relu_fwd(dst0 = NULL). */
size_t gx_deconv5x5s2_gn_stats_ws_bytes(int N, int Cin, int Cout, int Hin, int Win);
int gx_deconv5x5s2_gn_stats_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int Cout,
                                int Hin, int Win, int groups, float eps, float* mean, float* rstd, int* fused,
                                void* ws, size_t ws_bytes, gx_stream_t stream);
int gx_deconv5x5s2_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cin_out, int Cout,
                         int Hin, int Win, void* ws, size_t ws_bytes, gx_stream_t stream);
size_t gx_deconv5x5s2_wgrad_ws_bytes(int N, int Cin, int Cout, int Hin, int Win);
int gx_deconv5x5s2_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int Hin, int Win,
                         void* ws, size_t ws_bytes, gx_stream_t stream);

/* ---- GroupNorm(groups, eps, affine) + ReLU: modules/blocks.py:159-165, models/genesisv2_config.py:91-98;
 *      fused with the UNet's nearest resampling + torch.cat placement (modules/unet.py:78,86,89).
 *      A "view" (ptr, ctot, c0, mode) names channels [c0, c0+C) of a buffer [N, ctot, Hd, Wd]:
 *      mode 0 same size, 1 buffer is 2x up-sampled (nearest), 2 buffer is 2x down-sampled ([::2, ::2]).
 *      fwd writes relu(gn(y)) to dst0 (and dst1 if non-NULL), and mean/rstd [N*groups].
 *      bwd reads d(out) from g0 (+ g1 if non-NULL), writes dy [N,C,H,W], dgamma/dbeta [C] and, if
 *      dbias != NULL, sum_{n,hw} dy (the gradient of a per-channel bias added before the norm).
 *      fwd with dst0 == NULL computes the statistics only: the consumer (gx_conv1x1_gn_fwd) normalises on load.
 *      gx_gn_relu_bwd_proj is the matching backward: d(out) is not read from memory but formed on load as the data
 *      gradient of the following 1x1 conv, gate * sum_o w[o][c] g_out[n][o][h][w] (g_out [N,Cout,H,W], w [Cout,C],
 *      Cout <= 8, gate an optional device scalar):
 *      the [N,C,H,W] activation and its gradient never exist in memory (genesisv2_config.py:97-98, last decoder stage). */
int gx_gn_relu_fwd(const float* y, const float* gamma, const float* beta, int N, int C, int H, int W, int groups,
                   float eps, float* dst0, int dst0_ctot, int dst0_c0, int dst0_mode, float* dst1, int dst1_ctot,
                   int dst1_c0, int dst1_mode, float* mean, float* rstd, gx_stream_t stream);
size_t gx_gn_relu_bwd_ws_bytes(int N, int C);
int gx_gn_relu_bwd(const float* y, const float* gamma, const float* beta, const float* mean, const float* rstd,
                   int N, int C, int H, int W, int groups, const float* g0, int g0_ctot, int g0_c0, int g0_mode,
                   const float* g1, int g1_ctot, int g1_c0, int g1_mode, float* dy, float* dgamma, float* dbeta,
                   float* dbias, void* ws, size_t ws_bytes, gx_stream_t stream);
/*      ..._parts: a gradient source (modes 0-2) may still be the `nsplit` split-K partial slabs of the data gradient that
 *      produced it (gx_conv3x3_dgrad_parts), `split_stride` floats apart: summed on load in slab order -- the stand-alone
 *      reduce launch of the <= 16 x 16 layers is skipped. */
int gx_gn_relu_bwd_parts(const float* y, const float* gamma, const float* beta, const float* mean, const float* rstd,
                         int N, int C, int H, int W, int groups, const float* g0, int g0_ctot, int g0_c0, int g0_mode,
                         int g0_nsplit, size_t g0_split_stride, const float* g1, int g1_ctot, int g1_c0, int g1_mode,
                         int g1_nsplit, size_t g1_split_stride, float* dy, float* dgamma, float* dbeta, float* dbias, void* ws,
                         size_t ws_bytes, gx_stream_t stream);
int gx_gn_relu_bwd_proj(const float* y, const float* gamma, const float* beta, const float* mean, const float* rstd,
                        int N, int C, int H, int W, int groups, const float* g_out, int Cout, const float* w,
                        const float* gate, float* dy, float* dgamma, float* dbeta, float* dbias, float* wpart,
                        float* bpart, void* ws, size_t ws_bytes, gx_stream_t stream);
/*      wpart [N,Cout,C] / bpart [N,Cout] (both or neither; allowed when gx_gn_relu_bwd_proj_fuses_wgrad(...) != 0): the
 *      kernel also emits the per-image partials of the 1x1 conv's weight / bias gradient (its input exists only inside
 *      this kernel); gx_conv1x1_gn_wgrad_finish sums them over the images -- no separate weight-gradient pass over y. */
int gx_gn_relu_bwd_proj_fuses_wgrad(int C, int H, int W, int groups, int Cout);
/*      Test diagnostic: the number of elements of relu(gn(y)) that are > 0, evaluated with the expression the kernels above use
 *      for their ReLU decision ((y - mean) * rstd * gamma + beta > 0) -- the count the reference's nn.ReLU modules
 *      (modules/blocks.py:164, models/genesisv2_config.py:92-98) would report for (out > 0).sum().  *count: device uint64. */
int gx_gn_relu_active_count(const float* y, const float* gamma, const float* beta, const float* mean, const float* rstd, int N,
                            int C, int H, int W, int groups, unsigned long long* count, gx_stream_t stream);
/*      workspace of gx_gn_relu_bwd_proj (>= gx_gn_relu_bwd_ws_bytes): slabs too large for one workgroup (128 x 128 images) are
 *      processed in 4096-pixel chunks -- chunk sums, per-slab constants, chunked apply -- whose records live here */
size_t gx_gn_relu_bwd_proj_ws_bytes(int N, int C, int H, int W, int groups, int Cout);

/* ---- Instance-Colouring stick-breaking attention: modules/attention.py:162-226.
 *      colour [B,C<=8,H,W]; log_sigma: device pointer to the fp64 0-dim parameter; rand_pixel [B,1,H,W];
 *      seed_idx_in: NULL (argmax of rand*scope, first max) or [K-1,B] int64 seeds to force.
 *      kernel_type 0 gaussian, 1 laplacian, 2 epanechnikov.
 *      Outputs: log_m [K,B,1,H,W], log_s [K,B,1,H,W], seeds [K-1,B,C], seed_idx_out [K-1,B] int64.
 *      bwd: g_log_m [K,B,1,H,W] -> dcolour [B,C,H,W] (fully written), dlog_sigma (fp64 scalar). */
int gx_icsbp_fwd(const float* colour, const double* log_sigma, const float* rand_pixel, const int64_t* seed_idx_in,
                 int B, int C, int H, int W, int K, int kernel_type, float* log_m, float* log_s, float* seeds,
                 int64_t* seed_idx_out, gx_stream_t stream);
size_t gx_icsbp_bwd_ws_bytes(int B, int H, int W, int K);
int gx_icsbp_bwd(const float* colour, const double* log_sigma, const float* seeds, const int64_t* seed_idx,
                 const float* g_log_m, int B, int C, int H, int W, int K, int kernel_type, float* dcolour,
                 double* dlog_sigma, void* ws, size_t ws_bytes, gx_stream_t stream);
/*      dynamic_K (modules/attention.py:218-219, models/genesisv2_config.py:118-137): an image stops at the first step
 *      whose mask mass sum_p exp(log_m) falls below min_mass (the reference: 20); nsteps[b] = steps it ran; its mask
 *      nsteps[b] is the remaining scope, later masks are -1e10 (the reference's padding), later seeds 0.  The backward
 *      takes nsteps and treats the masks accordingly. */
int gx_icsbp_fwd_dyn(const float* colour, const double* log_sigma, const float* rand_pixel, const int64_t* seed_idx_in,
                     int B, int C, int H, int W, int K, int kernel_type, float min_mass, float* log_m, float* log_s,
                     float* seeds, int64_t* seed_idx_out, int* nsteps, gx_stream_t stream);
int gx_icsbp_bwd_dyn(const float* colour, const double* log_sigma, const float* seeds, const int64_t* seed_idx,
                     const float* g_log_m, const int* nsteps, int B, int C, int H, int W, int K, int kernel_type,
                     float* dcolour, double* dlog_sigma, void* ws, size_t ws_bytes, gx_stream_t stream);

/* ---- masked slot pooling: models/genesisv2_config.py:146-152.
 *      S[b,k,c] = sum_hw exp(log_m[k,b]) * f[b,c];  msum[b,k] = sum_hw exp(log_m[k,b]).
 *      (feat_head's 1x1 conv commutes with the pooling and is applied to S by the caller.) */
int gx_maskpool_fwd(const float* f, const float* log_m, int B, int C, int H, int W, int K, float* S, float* msum,
                    gx_stream_t stream);
int gx_maskpool_bwd(const float* f, const float* log_m, const float* gS, const float* gmsum, int B, int C, int H,
                    int W, int K, float* df, float* dlog_m, gx_stream_t stream);

/* ---- mixture likelihood + reconstruction: models/genesisv2_config.py:212-223,
 *      models/monet_config.py:137-139, models/genesis_config.py:273-286.
 *      dec [K*B,4,H,W] (slot-major rows k*B+b; ch 0-2 RGB pre-activation, ch 3 mask logit), x [B,3,H,W].
 *      Outputs recon [B,3,H,W], x_r [K,B,3,H,W], log_m_r [K,B,1,H,W], err [B].
 *      bwd: g_err [B] -> ddec [K*B,4,H,W]. */
size_t gx_mixture_ws_bytes(int B, int H, int W);
int gx_mixture_fwd(const float* x, const float* dec, int B, int H, int W, int K, float pixel_std, int pixel_bound,
                   float* recon, float* x_r, float* log_m_r, float* err, void* ws, size_t ws_bytes,
                   gx_stream_t stream);
int gx_mixture_bwd(const float* x, const float* dec, const float* g_err, int B, int H, int W, int K,
                   float pixel_std, int pixel_bound, float* ddec, gx_stream_t stream);

/* ---- 1x1 convolution with Cout <= 8: y = gate * (W x + b) + addend.
 *      modules/blocks.py:175-178 (SemiConv: gate = ScalarGate parameter on device, addend = uv [Cout,H,W]),
 *      models/genesisv2_config.py:99 (decoder_module.13; gate = addend = NULL).
 *      bwd writes dx, dw [Cout,Cin], db [Cout] (if non-NULL) and dgate (iff gate given). */
int gx_conv1x1_fwd(const float* x, const float* w, const float* bias, const float* gate, const float* addend,
                   int N, int Cin, int Cout, int H, int W, float* y, gx_stream_t stream);
size_t gx_conv1x1_bwd_ws_bytes(int N, int Cin, int Cout, int H, int W);
int gx_conv1x1_bwd(const float* x, const float* dy, const float* w, const float* bias, const float* gate, int N,
                   int Cin, int Cout, int H, int W, float* dx, float* dw, float* db, float* dgate, void* ws,
                   size_t ws_bytes, gx_stream_t stream);
/*      ..._ex with accumulate = 1: dw / db are ADDED to what the destinations hold (the MONet UNet's final_conv is used K-1
 *      times per iteration); plain conv only (gate == NULL). */
int gx_conv1x1_bwd_ex(const float* x, const float* dy, const float* w, const float* bias, const float* gate, int N,
                      int Cin, int Cout, int H, int W, float* dx, float* dw, float* db, float* dgate, int accumulate,
                      void* ws, size_t ws_bytes, gx_stream_t stream);
/*      ..._act: the conv's input x is the output of a bias + activation layer (modules/decoders.py:25-32: ..., ReLU,
 *      Conv2d(1x1)): dxa = dx * act'(x) (act 1 ReLU, 2 ELU) in the data-gradient kernel and dbx [Cin] (NULL to skip) = sum_{n,hw} dxa, that
 *      layer's bias gradient -- gx_conv1x1_bwd followed by gx_bias_act_bwd without the latter's pass. */
size_t gx_conv1x1_bwd_act_ws_bytes(int N, int Cin, int Cout, int H, int W);
int gx_conv1x1_bwd_act(const float* x, const float* dy, const float* w, const float* bias, int N, int Cin, int Cout, int H,
                       int W, int act, float* dxa, float* dw, float* db, float* dbx, void* ws, size_t ws_bytes,
                       gx_stream_t stream);
/*      The same 1x1 conv reading the PRE-norm tensor y of the GroupNorm+ReLU layer in front of it (statistics from
 *      gx_gn_relu_fwd with dst0 == NULL): relu(gn(y)) is formed on load, forward and in the weight gradient; the data
 *      gradient is folded into gx_gn_relu_bwd_proj.  Cin <= 64 (wgrad), H*W % 256 == 0. */
int gx_conv1x1_gn_fwd(const float* y_pre, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, int groups, const float* w, const float* bias, const float* gate,
                      const float* addend, int N, int Cin, int Cout, int H, int W, float* out, gx_stream_t stream);
size_t gx_conv1x1_gn_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W);
int gx_conv1x1_gn_wgrad(const float* y_pre, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, int groups, const float* g_out, const float* w, const float* bias,
                        const float* gate, int N, int Cin, int Cout, int H, int W, float* dw, float* db,
                        float* dgate, void* ws, size_t ws_bytes, gx_stream_t stream);
size_t gx_conv1x1_gn_wgrad_finish_ws_bytes(int Cin, int Cout);
int gx_conv1x1_gn_wgrad_finish(const float* wpart, const float* bpart, int N, int Cin, int Cout, const float* w,
                               const float* bias, const float* gate, float* dw, float* db, float* dgate, void* ws,
                               size_t ws_bytes, gx_stream_t stream);

/* ---- optimiser side of the training step.
 *      gx_adam_step: torch.optim.Adam update (train.py:174-175,263) on flat buffers p/g/m/v of n elements
 *      (fp32, or fp64 when is_f64 -- att_process.log_sigma is an fp64 parameter); `step` is a device int64
 *      holding t (>= 1, bump it with gx_step_increment first); the gradient is multiplied by grad_scale.
 *      gx_geco_update: utils/geco.py:39-49 on device; state = {beta, err_ema, initialised}; err = device
 *      scalar with the batch-mean reconstruction error; no host sync (the reference calls .item()). */
int gx_adam_step(void* p, const void* g, void* m, void* v, size_t n, int is_f64, int64_t* step, double lr,
                 double beta1, double beta2, double eps, float grad_scale, gx_stream_t stream);
int gx_step_increment(int64_t* step, gx_stream_t stream);
/*      the step's noise (torch.rand for the IC-SBP's rand_pixel, modules/attention.py:177-178; torch.randn for the latents'
 *      rsample, models/genesisv2_config.py:157) in one launch: Philox4x32-10 keyed by (seed, *step, position), nu uniform [0, 1)
 *      numbers into u and nz standard normals into z; a replayed HIP graph draws fresh numbers every step. */
int gx_philox_noise(float* u, long long nu, float* z, long long nz, unsigned long long seed, const int64_t* step,
                    gx_stream_t stream);
int gx_geco_update(float* state, const float* err, float goal, float step_size, float alpha, float speedup,
                   int use_speedup, float beta_min, float beta_max, gx_stream_t stream);
/*      The tail of a training step in two launches: gx_geco_update_step = gx_geco_update + gx_step_increment;
 *      gx_adam_step_pair = gx_adam_step on the fp32 group and on the fp64 group (n64 may be 0) and, with
 *      zero_grads != 0, zeroes the gradients as it consumes them (the next iteration's backward accumulates into a
 *      clean bucket without a separate fill launch). */
int gx_geco_update_step(float* state, const float* err, float goal, float step_size, float alpha, float speedup,
                        int use_speedup, float beta_min, float beta_max, int64_t* step, gx_stream_t stream);
int gx_adam_step_pair(float* p32, float* g32, float* m32, float* v32, size_t n32, double* p64, double* g64,
                      double* m64, double* v64, size_t n64, int64_t* step, double lr, double beta1, double beta2,
                      double eps, float grad_scale, int zero_grads, gx_stream_t stream);

/*      The other branches of train.py's objective / optimiser set-up.
 *      gx_optimiser_step_pair: kind 1 = torch.optim.RMSprop(params, lr) (alpha = hp = 0.99, eps 1e-8, no momentum;
 *      train.py:171-172), kind 2 = torch.optim.SGD(params, lr, momentum = hp = 0.9) (:175-176); one state buffer m per group
 *      (square average / momentum buffer, zero before the first step); fp32 + fp64 groups in one launch like
 *      gx_adam_step_pair.
 *      gx_beta_warmup: *out = clamp(beta * (*step) / warm_iters, 0, beta) -- the fixed-beta objective's linear warm-up over
 *      warm_iters = 0.2 * train_iter iterations (train.py:252-258); *step = the iteration index (the optimiser's step counter
 *      before this iteration's increment).
 *      gx_mse_rmse: out[0] = mean_b mean_{chw} (x - recon)^2, out[1] = mean_b sqrt(mean_{chw} (x - recon)^2)
 *      (train.py:244-246; x, recon [B, n]); ws: gx_mse_rmse_ws_bytes(B), its first 16 bytes zero before the first launch. */
int gx_optimiser_step_pair(int kind, float* p32, float* g32, float* m32, size_t n32, double* p64, double* g64, double* m64,
                           size_t n64, int64_t* step, double lr, double hp, double eps, float grad_scale, int zero_grads,
                           gx_stream_t stream);
int gx_beta_warmup(const int64_t* step, float beta, float warm_iters, float* out, gx_stream_t stream);
size_t gx_mse_rmse_ws_bytes(int B);
int gx_mse_rmse(const float* x, const float* recon, int B, int n, float* out, void* ws, size_t ws_bytes, gx_stream_t stream);

/* ---- live per-kernel profiling (bench.py's roofline leg): when enabled every kernel launch is bracketed
 *      by two HIP events on its launch stream and tagged with its ALGORITHMIC flops / bytes
 *      (DESIGN.md "kernels and rooflines"); gx_profile_collect accumulates per kernel symbol. */
int gx_profile_enable(int on);
int gx_profile_num_kernels(void);
const char* gx_profile_kernel_name(int kid);
int gx_profile_collect(double* total_ms, double* launches, double* flops, double* bytes);

/* ---- ComponentVAE / MONet path (modules/component_vae.py:45-93, modules/encoders.py:31-37,
 *      modules/decoders.py:25-32, models/monet_config.py:74-128).
 *      gx_conv3x3_bias_act_fwd: conv3x3 s1 p1 + bias + activation (act 0 none, 1 ReLU, 2 ELU) on any HxW grid.
 *      The BroadcastDecoder's VALID 3x3 convs run as this 'same' conv on the (img+2L)^2 broadcast canvas: the
 *      centre crop after L layers equals the valid-conv chain exactly (border pollution advances one pixel per
 *      layer and is cropped; its gradients are identically zero).
 *      gx_bias_act_bwd: dy = g * act'(out) (derivative from the OUTPUT) and dbias[c] = sum dy (NULL to skip).
 *      gx_conv2d_direct_*: generic k<=5 / stride / pad convolution (implicit GEMM on the fp32 matrix cores, operands
 *      gathered; split-K + fixed-order reduce for the weight gradient): the stride-2 encoder convs here and the
 *      sylvester 5x5 gated (de)convolutions below.
 *      gx_mixture_w_*: mixture likelihood with EXTERNAL mixing log-weights log_w [K,B,1,H,W] (MONet mixes with
 *      the attention masks, not with log_softmax(logits)) and a separate std for the first slot; dec has dec_ch
 *      channels per slot: 4 (RGB + logit, MONet) or 3 (RGB, GENESIS); bwd also returns dlog_w; the logit channel
 *      of ddec (if any) is zero. */
int gx_conv3x3_bias_act_fwd(const float* x, const float* w, const float* bias, int act, float* y, int N, int Cin,
                            int Cout, int H, int W, void* ws, size_t ws_bytes, gx_stream_t stream);
size_t gx_bias_act_bwd_ws_bytes(int N, int C);
int gx_bias_act_bwd(const float* out, const float* g, int N, int C, int H, int W, int act, float* dy, float* dbias,
                    void* ws, size_t ws_bytes, gx_stream_t stream);
/*      gx_conv3x3_dgrad_act: the data gradient of a conv3x3 whose input was such a layer's output xout [N,Cin,H,W]
 *      (modules/decoders.py:25-32: Conv2d, ReLU, Conv2d, ...): dxa = dgrad(dy, w) * act'(xout) and dbias [Cin] (NULL to
 *      skip) = sum_{n,hw} dxa -- gx_conv3x3_dgrad followed by gx_bias_act_bwd with the activation's backward in the conv
 *      kernel's epilogue.  Shapes of the bf16-pipe <= 32-channel kernel only: gx_conv3x3_dgrad_act_supported.
 *      Environment: GENESIS_DGRAD_ACT_FUSE=0 (supported() answers 0: the two separate calls). */
int gx_conv3x3_dgrad_act_supported(int N, int Cin, int Cout, int H, int W);
size_t gx_conv3x3_dgrad_act_ws_bytes(int N, int Cin, int Cout, int H, int W);
int gx_conv3x3_dgrad_act(const float* dy, const float* w, const float* xout, int act, float* dxa, float* dbias, int N,
                         int Cin, int Cout, int H, int W, void* ws, size_t ws_bytes, gx_stream_t stream);
/* ---- 5 x 5 stride-1 pad-2 weight gradient of the gated stacks (third_party/sylvester/VAE.py:18-33, layers.py:40-101) on
 *      the bf16-pipe row-ring tiles: dw [CA][CB][5][5] = sum_{n,p} a[n][CA][p] * b[n][CB][p + (kh - 2, kw - 2)].
 *      Conv2d: a = dy, b = x -> dw [Cout][Cin][5][5]; ConvTranspose2d stride 1: a = x, b = dy -> dw [Cin][Cout][5][5].
 *      Queued into the step's stream-K launch when deferral is on (gx_defer_*), else launched at once. */
/*      conv3x3 weight gradient of a layer with 32 channels on both sides (the BroadcastDecoder's canvas convs,
 *      modules/decoders.py:21-35) and N % 4 == 0 images: four images per workgroup, one per wave, so that every wave's
 *      32 x 32 block is a wanted one; x, dy [N,32,H,W] -> dw [32,32,3,3] (overwritten).  H x W any grid with W % 4 == 0. */
int gx_conv3x3_wgrad_quad_supported(int N, int C, int H, int W);
size_t gx_conv3x3_wgrad_quad_ws_bytes(int N, int C, int H, int W);
int gx_conv3x3_wgrad_quad(const float* x, const float* dy, float* dw, int N, int C, int H, int W, void* ws, size_t ws_bytes,
                          gx_stream_t stream);
/*      ..._bias: also dbias [C] = sum_{n,hw} dy -- the layer's bias gradient from the weight gradient's own read of dy (per-thread
 *      sums inside the bf16-pipe kernel, one small reduce; a plane-sum pass over dy where that kernel is not the one that runs). */
size_t gx_conv3x3_wgrad_quad_bias_ws_bytes(int N, int C, int H, int W);
int gx_conv3x3_wgrad_quad_bias(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int H, int W, void* ws,
                               size_t ws_bytes, gx_stream_t stream);
/*      the layers themselves on the tap-conv MFMA kernel: out [N,M,H,W] from in [N,K,H,W];
 *      flip 0: cross-correlation with w [M][K][5][5] (Conv2d forward; data gradient of a stride-1 ConvTranspose2d),
 *      flip 1: convolution with w [K][M][5][5] (Conv2d data gradient, in = dy; stride-1 ConvTranspose2d forward). */
int gx_conv5x5s1_supported(int N, int K, int M, int H, int W);
size_t gx_conv5x5s1_ws_bytes(int N, int K, int M, int H, int W);
int gx_conv5x5s1(const float* in, const float* w, float* out, int N, int K, int M, int H, int W, int flip, void* ws,
                 size_t ws_bytes, gx_stream_t stream);
int gx_conv5x5_wgrad_supported(int N, int CA, int CB, int H, int W);
size_t gx_conv5x5_wgrad_ws_bytes(int N, int CA, int CB, int H, int W);
int gx_conv5x5_wgrad(const float* a, const float* b, float* dw, int N, int CA, int CB, int H, int W, void* ws,
                     size_t ws_bytes, gx_stream_t stream);

/*      conv3x3 stride 2 pad 1 data gradient (the ComponentVAE encoder, modules/encoders.py:31-34) without the stride's
 *      structural zeros, on the vector ALUs: only the first cin_n channels of dx [N,Cin,H,W] are computed and written
 *      (the encoder's first layer needs the mask channel's gradient alone). */
int gx_conv3x3s2_dgrad_small(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, int cin_n,
                             gx_stream_t stream);
/*      _ex: dx holds dx_channels >= cin_n channels per image (dx_channels = cin_n: the compact gradient of the mask channel). */
int gx_conv3x3s2_dgrad_small_ex(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, int cin_n,
                                int dx_channels, gx_stream_t stream);
/*      the slot-major ComponentVAE input [log_m_k | x] (modules/component_vae.py:59-66: torch.cat((log_m_k, x), 1) per slot;
 *      models/monet_config.py / genesis_config.py batch the K slots): out [K*B, 1 + C, H, W] from mask [K,B,1,H,W] and
 *      x [B,C,H,W] in one pass -- no x.repeat(K), no torch.cat.  H W % 4 == 0. */
int gx_mask_image_stack(const float* mask, const float* x, float* out, int K, int B, int C, int H, int W, gx_stream_t stream);
/*      ... and its weight gradient dw [Cout,Cin,3,3] (lanes = output pixels, 9 taps x 8 output channels per thread, fixed-
 *      order split reduction; ws: gx_conv3x3s2_wgrad_small_ws_bytes). */
size_t gx_conv3x3s2_wgrad_small_ws_bytes(int N, int Cin, int Cout, int H, int W);
int gx_conv3x3s2_wgrad_small(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, void* ws,
                             size_t ws_bytes, gx_stream_t stream);
int gx_conv2d_direct_fwd(const float* x, const float* w, const float* bias, int act, float* y, int N, int Cin,
                         int Cout, int H, int W, int k, int stride, int pad, gx_stream_t stream);
/*      ... with a workspace: layers whose output tiles leave most of the chip idle (the ComponentVAE encoder's last stride-2
 *      convs, modules/encoders.py:31-34) split the contraction over workgroups and finish in a fixed-order reduce + bias +
 *      activation launch; ws_bytes 0 (or a NULL ws) = the single-launch form above */
size_t gx_conv2d_direct_fwd_ws_bytes(int N, int Cin, int Cout, int H, int W, int k, int stride, int pad);
int gx_conv2d_direct_fwd_ws(const float* x, const float* w, const float* bias, int act, float* y, int N, int Cin,
                            int Cout, int H, int W, int k, int stride, int pad, void* ws, size_t ws_bytes,
                            gx_stream_t stream);
int gx_conv2d_direct_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, int H, int W, int k,
                           int stride, int pad, gx_stream_t stream);
size_t gx_conv2d_direct_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W, int k, int stride, int pad);
int gx_conv2d_direct_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, int k,
                           int stride, int pad, void* ws, size_t ws_bytes, gx_stream_t stream);
int gx_mixture_w_fwd(const float* x, const float* dec, const float* log_w, int dec_ch, int B, int H, int W, int K,
                     float pixel_std1, float pixel_std2, int pixel_bound, float* recon, float* x_r, float* err,
                     void* ws, size_t ws_bytes, gx_stream_t stream);
int gx_mixture_w_bwd(const float* x, const float* dec, const float* log_w, const float* g_err, int dec_ch, int B,
                     int H, int W, int K, float pixel_std1, float pixel_std2, int pixel_bound, float* ddec,
                     float* dlog_w, gx_stream_t stream);

/* ---- gated (de)convolution unit of the sylvester VAE (third_party/sylvester/layers.py:40-54,87-101):
 *      out = norm_h(h + b_h) * sigmoid(norm_g(g + b_g)), [h | g] = the two channel halves of y [N,2C,H,W];
 *      norm 0 none, 1 BatchNorm2d with training-mode batch statistics, 2 InstanceNorm2d(affine); `bias` [2C] is the
 *      (de)conv bias, folded in here.  stats: gx_gated_stats_floats() floats ({mean, rstd} per unit), kept for bwd
 *      (and for the caller's running-statistics update).  bwd returns dy [N,2C,H,W] and the affine / bias grads. */
size_t gx_gated_stats_floats(int norm, int N, int C);
int gx_gated_norm_fwd(const float* y, const float* bias, int norm, const float* gamma_h, const float* beta_h,
                      const float* gamma_g, const float* beta_g, int N, int C, int H, int W, float eps, float* out,
                      float* stats, gx_stream_t stream);
/*      nn.BatchNorm2d running statistics of a gated unit's two norms (momentum update with the unbiased variance,
 *      num_batches_tracked += 1) from the {mean, rstd} pairs gx_gated_norm_fwd left in `stats`; m = N H W. */
int gx_bn_running_update(const float* stats, int C, double m, float eps, float momentum, float* rm_h, float* rv_h, float* rm_g,
                         float* rv_g, long long* nbt_h, long long* nbt_g, gx_stream_t stream);
/*      ... or the same update without a launch of its own: gx_gated_bn_running arms a one-shot, per-thread request that the NEXT
 *      gx_gated_norm_fwd of this thread (norm 1) applies inside its apply kernel, whose workgroups also fold the statistics pass's
 *      partial sums themselves (two launches less per unit; GENESIS_GATED_FUSE=0: the stand-alone launches; rm_h NULL disarms).
 *      gx_gated_norm_bwd (norm 1) folds its sums and writes the affine gradients the same way. */
int gx_gated_bn_running(float* rm_h, float* rv_h, float* rm_g, float* rv_g, long long* nbt_h, long long* nbt_g, float momentum);
size_t gx_gated_norm_bwd_ws_bytes(int norm, int N, int C);
int gx_gated_norm_bwd(const float* y, const float* bias, int norm, const float* gamma_h, const float* beta_h,
                      const float* gamma_g, const float* beta_g, const float* stats, const float* dout, int N, int C,
                      int H, int W, float* dy, float* dgamma_h, float* dbeta_h, float* dgamma_g, float* dbeta_g,
                      float* dbias, void* ws, size_t ws_bytes, gx_stream_t stream);

/* ---- the gated unit's BatchNorm over the batches of SEVERAL ranks (cross-replica BatchNorm: the reference's single-device
 *      statistics at the global batch, models/genesis_config.py:39-40 -> third_party/sylvester/layers.py:26-27, when the batch is
 *      sharded over GPUs).  Forward: gx_gated_bn_local_sums leaves fp64 {sum, sum of squares} of (y + bias) per
 *      channel in sums [2C][2]; the caller adds them over the ranks; gx_gated_bn_apply forms {mean, rstd} from the summed pairs
 *      and the GLOBAL count m = sum over ranks of N H W (written to `stats` [2C][2], kept for bwd and gx_bn_running_update) and
 *      applies the unit.  Backward: gx_gated_bn_bwd_local_sums leaves the rank's {S1, S2} per channel (float [2C][2]) and
 *      writes the affine / bias gradients from them (the step's gradient all-reduce adds the ranks); the caller adds the sums
 *      over the ranks; gx_gated_bn_bwd_apply writes dy from the summed pairs and m.  ws: gx_gated_bn_sums_ws_bytes(). */
size_t gx_gated_bn_sums_ws_bytes(int N, int C);
int gx_gated_bn_local_sums(const float* y, const float* bias, int N, int C, int H, int W, double* sums, void* ws,
                           size_t ws_bytes, gx_stream_t stream);
int gx_gated_bn_apply(const float* y, const float* bias, const double* sums, double m, const float* gamma_h,
                      const float* beta_h, const float* gamma_g, const float* beta_g, int N, int C, int H, int W, float eps,
                      float* out, float* stats, gx_stream_t stream);
int gx_gated_bn_bwd_local_sums(const float* y, const float* bias, const float* gamma_h, const float* beta_h,
                               const float* gamma_g, const float* beta_g, const float* stats, const float* dout, int N, int C,
                               int H, int W, float* sums, float* dgamma_h, float* dbeta_h, float* dgamma_g, float* dbeta_g,
                               float* dbias, void* ws, size_t ws_bytes, gx_stream_t stream);
int gx_gated_bn_bwd_apply(const float* y, const float* bias, const float* gamma_h, const float* beta_h, const float* gamma_g,
                          const float* beta_g, const float* stats, const float* dout, const float* sums, double m, int N, int C,
                          int H, int W, float* dy, gx_stream_t stream);

/* ---- stick-breaking mask recursion in log space (modules/attention.py:31-51 SimpleSBP, :118-124 LatentSBP):
 *      log_m[t] = s_t + logsigmoid(l_t), s_{t+1} = s_t + logsigmoid(-l_t), s_0 = log_s0 (NULL: 0); logits / log_m /
 *      log_s [T,P] (log_s[t] = the scope AFTER step t); last_scope: log_m[T-1] = s_{T-1}, the remaining scope
 *      (genesis_config.py:167-169).  bwd: g_log_m / g_log_s [T,P] (either may be NULL) -> g_logits [T,P],
 *      g_log_s0 [P] (may be NULL). */
int gx_sbp_scan_fwd(const float* logits, const float* log_s0, int T, size_t P, int last_scope, float* log_m,
                    float* log_s, gx_stream_t stream);
int gx_sbp_scan_bwd(const float* logits, const float* g_log_m, const float* g_log_s, int T, size_t P, int last_scope,
                    float* g_logits, float* g_log_s0, gx_stream_t stream);
/* ---- MONet.kl_m_loss (models/monet_config.py:157-170): per image, sum over pixels of KL(Cat(q) || Cat(p)) with
 *      q = max(exp(log_m), 1e-5), p = max(exp(log_m_r), 1e-5) renormalised over K; log_m / log_m_r [K,B,HW] slot-major,
 *      kl [B].  bwd: g_kl [B] -> g_log_m and, when g_log_m_r != NULL (detach_mr_in_klm = False,
 *      genesisv2_config.py:172-176), the gradient through the reconstructed masks. */
int gx_categorical_kl_fwd(const float* log_m, const float* log_m_r, int K, int B, int HW, float* kl,
                          gx_stream_t stream);
int gx_categorical_kl_bwd(const float* log_m, const float* log_m_r, const float* g_kl, int K, int B, int HW,
                          float* g_log_m, float* g_log_m_r, gx_stream_t stream);
/*      the reconstructed masks log_m_r [K,B,HW] = log_softmax over the K slots of the decoder output's last channel
 *      (MONet.get_mask_recon_stack, monet_config.py:137-139; dec [K*B, C, HW] slot-major), and their gradient back into
 *      the decoder output: g [K,B,HW] -> g_dec [K*B, C, HW] (channels < C-1 zero) */
int gx_logsoftmax_k_fwd(const float* dec, int K, int B, int HW, int C, float* log_m_r, gx_stream_t stream);
int gx_logsoftmax_k_bwd(const float* log_m_r, const float* g, int K, int B, int HW, int C, float* g_dec,
                        gx_stream_t stream);

/* ---- spatial broadcast + coordinate channels (modules/blocks.py:104-130 BroadcastLayer / PixelCoords).
 *      gx_broadcast_concat: out [N, D+2, d, d] = [ z[n] broadcast | coords[0] | coords[1] ], z [N,D], coords [2,d,d]
 *      (the GENESIS-V2 decoder input, models/genesisv2_config.py:89-90).
 *      gx_bcast_conv3x3_*: the BroadcastDecoder's first layer act(conv3x3(broadcast canvas, w) + b)
 *      (modules/decoders.py:25-28) WITHOUT the canvas: w [Co, L+2, 3, 3]; rowc / colc [d] = the row / column
 *      coordinate (coords[0][:, 0] / coords[1][0, :]: channel L is constant along x, channel L+1 along y);
 *      out / y / g [N, Co, d, d] on the d x d canvas, d a multiple of 4.  Inside the canvas the result equals the
 *      conv on the materialised canvas; the one-pixel border ring (outside the VALID conv's output) holds the
 *      interior formula and must not be consumed.  bwd: g = dL/dout; gradients are taken over the interior
 *      [1, d-2]^2 (the valid conv's outputs): dz [N,L], dw [Co,L+2,3,3], db [Co] (may be NULL); act 0 none, 1 ReLU,
 *      2 ELU. */
int gx_broadcast_concat(const float* z, const float* coords, float* out, int N, int D, int d, gx_stream_t stream);
/*      The GENESIS-V2 decoder's first layer, ConvTranspose2d(D+2, Cout, 5, 2, 2, 1) on that broadcast input
 *      (models/genesisv2_config.py:89-90), WITHOUT the canvas: the input is constant over the pixels, so
 *      out[n,co,oy,ox] = sum_ci z[n,ci] Wz[ci][co][oy][ox] + C[co][oy][ox] -- one [N,D] x [D, Cout (2d)^2] product
 *      (gx_linear_fwd / gx_linear_bwd on wz).  pack: w [D+2,Cout,5,5], b [Cout] (may be NULL), coords [2,d,d] ->
 *      wz [Cout (2d)^2, D] (the tap sums, nn.Linear weight layout) and bias [Cout (2d)^2] (b + the coordinate
 *      channels' contribution).  unpack: the gradients of wz and bias -> dw [D+2,Cout,5,5], db [Cout] (may be NULL). */
int gx_bcast_deconv5x5s2_pack(const float* w, const float* b, const float* coords, int D, int Cout, int d, float* wz,
                              float* bias, gx_stream_t stream);
int gx_bcast_deconv5x5s2_unpack(const float* dwz, const float* dbias, const float* coords, int D, int Cout, int d,
                                float* dw, float* db, gx_stream_t stream);
int gx_bcast_conv3x3_fwd(const float* z, const float* w, const float* bias, const float* rowc, const float* colc,
                         int act, float* out, int N, int L, int Co, int d, gx_stream_t stream);
size_t gx_bcast_conv3x3_bwd_ws_bytes(int N, int Co);
int gx_bcast_conv3x3_bwd(const float* y, const float* g, const float* z, const float* w, const float* rowc,
                         const float* colc, int act, int N, int L, int Co, int d, float* dz, float* dw, float* db,
                         void* ws, size_t ws_bytes, gx_stream_t stream);

/* ---- slot-latent head: reparameterised posterior sample + Monte-Carlo KL terms
 *      (models/genesisv2_config.py:154-160: mu, sigma_ps = z_head(obj).chunk(2); sigma = to_sigma(sigma_ps);
 *      z = Normal(mu, sigma).rsample(); modules/blocks.py:22-23 to_sigma = softplus(x + 0.5) + 1e-8, :28-36
 *      to_prior_sigma = sigmoid(x + 4) + 1e-4; models/genesis_config.py:288-343 mask_latent_loss:
 *      log_q = q_z_k.log_prob(z_k).sum(1), log_p under N(0,1) for the first slot and under
 *      N(tanh(lin[:D]), to_prior_sigma(lin[D:])) for later slots, lin = prior_linear(prior_lstm(z_{<k}))).
 *      zh [B,K,2D] = z_head output (mu | sigma_ps), eps [K,B,D] standard normal; z, mu, sigma [K,B,D] slot-major,
 *      log_q / log_p [K,B].  lin [K-1,B,2D], or NULL for the standard-normal prior on every slot.
 *      gx_latent_prior_logp_fwd writes out = log_p, or the KL sample log_q - log_p when log_q is given
 *      (kl_mode = 1 in bwd: g_out is then dL/d(log_q - log_p)).
 *      bwd: incoming gradients gz / gmu / gsigma [K,B,D], glogq [K,B] may each be NULL (= zero). */
int gx_latent_posterior_fwd(const float* zh, const float* eps, int B, int K, int D, float* z, float* mu,
                            float* sigma, float* log_q, gx_stream_t stream);
int gx_latent_posterior_bwd(const float* zh, const float* eps, const float* gz, const float* gmu,
                            const float* gsigma, const float* glogq, int B, int K, int D, float* dzh,
                            gx_stream_t stream);
/* _ex: the recurrent posterior of GENESIS' LatentSBP (modules/attention.py:103-118: the sampled z_{k-1} is concatenated to
 * the encoder features as the next LSTM input) without torch.cat -- fwd also writes z to z2 [K*B rows of ldz2 floats] (the
 * z columns of the next step's input rows; NULL: none); bwd adds a second incoming gz2 [K*B rows of ldgz2 floats] (the
 * input-projection gradient's z columns) to gz. */
int gx_latent_posterior_fwd_ex(const float* zh, const float* eps, int B, int K, int D, float* z, float* mu,
                               float* sigma, float* log_q, float* z2, int ldz2, gx_stream_t stream);
int gx_latent_posterior_bwd_ex(const float* zh, const float* eps, const float* gz, const float* gmu,
                               const float* gsigma, const float* glogq, const float* gz2, int ldgz2, int B, int K,
                               int D, float* dzh, gx_stream_t stream);
int gx_latent_prior_logp_fwd(const float* z, const float* lin, const float* log_q, int B, int K, int D,
                             float* out, gx_stream_t stream);
int gx_latent_prior_logp_bwd(const float* z, const float* lin, const float* g_out, int kl_mode, int B, int K,
                             int D, float* dz, float* dlin, gx_stream_t stream);
/*      ..._ex with all_slots = 1: every slot has a conditional prior, lin [K,B,2D] (Genesis' component prior,
 *      models/genesis_config.py:229-247: p(z_c | z_m) = N(tanh(mlp[:L]), to_prior_sigma(mlp[L:])) for all K slots) */
int gx_latent_prior_logp_fwd_ex(const float* z, const float* lin, const float* log_q, int B, int K, int D, int all_slots,
                                float* out, gx_stream_t stream);
int gx_latent_prior_logp_bwd_ex(const float* z, const float* lin, const float* g_out, int kl_mode, int B, int K,
                                int D, int all_slots, float* dz, float* dlin, gx_stream_t stream);
/*      one ancestral step of GenesisV2.sample (models/genesisv2_config.py:235-246): lin [B,2D] = prior_linear(lstm out),
 *      eps [B,D] standard normal -> z [B,D] = tanh(lin[:D]) + to_prior_sigma(lin[D:]) * eps */
int gx_latent_prior_sample(const float* lin, const float* eps, int B, int D, float* z, gx_stream_t stream);
/*      the same with the mean's tanh selectable: Genesis.sample's mask rollout (models/genesis_config.py:352-362) uses
 *      the RAW first half of prior_linear's output as the mean (tanh_mu = 0); its component prior
 *      (models/genesis_config.py:391-397: tanh(prior_mlp[:L]), to_prior_sigma(prior_mlp[L:])) is tanh_mu = 1 */
int gx_latent_prior_sample_ex(const float* lin, const float* eps, int B, int D, int tanh_mu, float* z,
                              gx_stream_t stream);

/* ---- loss aggregation of the training loop (train.py:226-242) with GECO's beta (utils/geco.py:47-49):
 *      err_mean = mean_b err[b]; kl_mean = sum_r mean_b kl[r][b] (kl [R,B], R = 0 / NULL: no KL term);
 *      out[5] = (loss = err_mean + beta kl_mean, err_mean + kl_mean, err_mean, kl_mean, beta); beta is read from
 *      device memory (GECO state); tail (NULL to skip) receives (err_mean, kl_mean) -- the gradient bucket's
 *      piggy-backed scalars; loss (NULL to skip) receives out[0] again (a separate one-element objective tensor).  bwd: d_err[b] = g/B, d_kl[r][b] = g beta / B for the scalar g = dL/d loss. */
int gx_elbo_fwd(const float* err, const float* kl, const float* beta, int B, int R, float* out, float* tail,
                float* loss, gx_stream_t stream);
/*      gx_elbo_fwd + the gradients gx_elbo_bwd returns for an upstream gradient of one, in one launch */
int gx_elbo_fwd_grads(const float* err, const float* kl, const float* beta, int B, int R, float* out, float* tail,
                      float* loss, float* d_err, float* d_kl, gx_stream_t stream);
int gx_elbo_bwd(const float* g_loss, const float* beta, int B, int R, float* d_err, float* d_kl,
                gx_stream_t stream);

/* ---- pooled slot features -> z_head[0] (models/genesisv2_config.py:146-154: obj_feat = feat_head(enc) * mask
 *      summed over pixels / (mask.sum + 1e-5); :76 nn.LayerNorm(2 feat_dim)).  lin [R,C] = pooled sums through
 *      feat_head[1]'s weight (gx_linear_fwd), msum [R] mask mass, fbias [C] feat_head[1].bias:
 *      obj = (lin + msum fbias) / (msum + 1e-5); y = LayerNorm(obj; gamma, beta, eps); stats [R,2] = (mean, rstd). */
int gx_pooled_head_fwd(const float* lin, const float* msum, const float* fbias, const float* gamma,
                       const float* beta, float eps, int R, int C, float* y, float* stats, gx_stream_t stream);
size_t gx_pooled_head_bwd_ws_bytes(int R, int C);
int gx_pooled_head_bwd(const float* lin, const float* msum, const float* fbias, const float* gamma,
                       const float* stats, const float* g, int R, int C, float* dlin, float* dmsum, float* dfbias,
                       float* dgamma, float* dbeta, void* ws, size_t ws_bytes, gx_stream_t stream);

/* ---- small dense layers (nn.Linear: modules/unet.py:58-62 bottleneck MLP, models/genesisv2_config.py:76-80
 *      z_head, :73 feat_head[1] on the pooled slot sums, models/genesis_config.py:106 prior_linear).
 *      y[M,N] = act(x[M,K] w[N,K]^T + b[N]), act 0 none / 1 ReLU / 2 ELU, b may be NULL.
 *      bwd: g = dL/dy; dpre = g * [y > 0] for act 1, g * (y > 0 ? 1 : y + 1) for act 2 (y = the layer OUTPUT, may be NULL for act 0);
 *      dx[M,K] = dpre w, dw[N,K] = dpre^T x, db[N] = column sums of dpre; each of dx / dw / db may be NULL to skip
 *      (db needs dw).  Row-major contiguous; fixed reduction order. */
int gx_linear_fwd(const float* x, const float* w, const float* b, int act, float* y, int M, int N, int K,
                  gx_stream_t stream);
int gx_linear_bwd(const float* x, const float* w, const float* y, const float* g, int act, float* dx, float* dw,
                  float* db, int M, int N, int K, gx_stream_t stream);
/*      Strided variants for operands that live inside larger buffers (the UNet bottleneck MLP writes its last
 *      layer straight into the first up-block's concat buffer, modules/unet.py:83-84, and reads that layer's
 *      gradient out of the concat buffer's gradient; the AR prior's LSTM input z[:-1] is the leading part of z):
 *      ldx / ldy / ldg / lddx = row strides in floats (>= the row length); y and g of the backward share ldg.
 *      dx_accumulate: bit 0: dx += dpre w (the second consumer of a tensor adds its gradient in place); bit 1: dw / db / db2
 *      += (a parameter used again in the same iteration: MONet's recurrent UNet);
 *      db2 (may be NULL): a second copy of db (nn.LSTM's b_ih and b_hh have the same gradient). */
int gx_linear_fwd_ld(const float* x, int ldx, const float* w, const float* b, int act, float* y, int ldy, int M,
                     int N, int K, gx_stream_t stream);
int gx_linear_bwd_ex(const float* x, int ldx, const float* w, const float* y, const float* g, int ldg, int act,
                     float* dx, int lddx, int dx_accumulate, float* dw, float* db, float* db2, int M, int N, int K,
                     gx_stream_t stream);

/* ---- plain matrix product with the weight in [K, N] layout: y[M,N] = x[M,K] w[K,N] -- the gated ConvTranspose2d 'fc' layer
 *      of the sylvester stacks applied to a 1 x 1 input (third_party/sylvester/VAE.py:27-33, layers.py:62-101: the weight
 *      [z, 2c, k, k] flattened to [z, 2c k k] IS w; no transposed copy).  K is small (the latent size), N large: an
 *      HBM-bound product on the vector ALUs (w is read once per 8 rows of x, coalesced; x rides in scalar registers).
 *      bwd: dw[K,N] = x^T g (dw may be NULL);  dx[M,K] = g w^T (dx may be NULL) summed over N in fixed-order partial
 *      slabs held in ws (gx_matmul_nn_bwd_ws_bytes).  Row-major contiguous, N % 4 == 0, 16-byte aligned. */
int gx_matmul_nn_fwd(const float* x, const float* w, float* y, int M, int N, int K, gx_stream_t stream);
size_t gx_matmul_nn_bwd_ws_bytes(int M, int N, int K);
int gx_matmul_nn_bwd(const float* x, const float* w, const float* g, float* dx, float* dw, int M, int N, int K,
                     void* ws, size_t ws_bytes, gx_stream_t stream);

/* ---- LSTM cell step (nn.LSTM, one layer, gate order i, f, g, o: models/genesis_config.py:105 prior_lstm, :297-307;
 *      modules/attention.py LatentSBP core).  The caller computes gx = x w_ih^T + b_ih for all steps with
 *      gx_linear_fwd and unrolls time:
 *      pre = gx[B,4H] + h_prev[B,H] w_hh[4H,H]^T + b_hh; act[B,4H] = (sigm i | sigm f | tanh g | sigm o);
 *      c = f c_prev + i g; h = o tanh(c).  h_prev = c_prev = NULL: zero initial state.
 *      bwd of one step: dh = g_h[B,H] (NULL = 0) + dgates_next[B,4H] w_hh (NULL at the last step);
 *      dc = dc_next (NULL = 0) + dh o (1 - tanh(c)^2); dgates[B,4H] = pre-activation gradients; dc_prev = dc f.
 *      Weight gradients follow from gx_linear_bwd on the stacked dgates.  H % 16 == 0. */
int gx_lstm_step_fwd(const float* gx, const float* h_prev, const float* c_prev, const float* w_hh,
                     const float* b_hh, int B, int H, float* act, float* c, float* h, gx_stream_t stream);
int gx_lstm_step_bwd(const float* g_h, const float* dgates_next, const float* w_hh, const float* act,
                     const float* c, const float* c_prev, const float* dc_next, int B, int H, float* dgates,
                     float* dc_prev, gx_stream_t stream);
/*      the whole sequence (zero initial state) in ONE launch each way -- the same per-step arithmetic in the same order, i.e. the
 *      results of the unrolled gx_lstm_step_* calls bit for bit; the steps are separated by grid-wide barriers inside the
 *      kernel, so the grid (H / 16 x ceil(B / 16) workgroups) has to be resident at once: gx_lstm_seq_max_steps(B, H) is
 *      the longest sequence one launch takes for these sizes (0: use the per-step calls).
 *      fwd: gx [T,B,4H] (input projection of all steps) -> act [T,B,4H], c [T,B,H], h [T,B,H].
 *      bwd: g_h [T,B,H] (dL/dh of every step), act, c of the forward -> dgates [T,B,4H]; dc2: two [B,H] planes of scratch.
 *      bar: gx_lstm_seq_ws_bytes() bytes of device memory, ZERO before the first launch that uses it; a launch leaves it
 *      zero again.  One `bar` must not be shared by launches that can run concurrently.
 *      (models/genesis_config.py:297-307 prior_lstm over z_{<k}) */
int gx_lstm_seq_max_steps(int B, int H);
size_t gx_lstm_seq_ws_bytes(void);
int gx_lstm_seq_fwd(const float* gx, const float* w_hh, const float* b_hh, int T, int B, int H, float* act, float* c,
                    float* h, void* bar, gx_stream_t stream);
int gx_lstm_seq_bwd(const float* g_h, const float* w_hh, const float* act, const float* c, int T, int B, int H,
                    float* dgates, float* dc2, void* bar, gx_stream_t stream);

/* ---- packed-weight cache.  The conv / deconv entry points above re-pack their weight tensor into the MFMA
 *      operand layout on every call; inside a training loop the weights change once per optimiser step, so:
 *      id = gx_weight_cache_create(); gx_weight_cache_record(id, 1); <one iteration>; gx_weight_cache_record(id, 0)
 *      registers every (weight pointer, layout) the iteration used (device buffers are allocated here -- do not
 *      record inside a stream capture).  From then on gx_weight_cache_refresh(id, stream) re-packs all of them in
 *      ONE launch and the entry points are served from the cache (matching weight pointer and shape) until
 *      gx_weight_cache_release().  The caller guarantees the weights do not change inside that window. */
int gx_weight_cache_create(void);
int gx_weight_cache_record(int id, int on);
int gx_weight_cache_size(int id);
int gx_weight_cache_refresh(int id, gx_stream_t stream);
int gx_weight_cache_release(void);
/*      gx_weight_cache_activate(id): serve the calls that follow from cache `id` as gx_weight_cache_refresh does, WITHOUT re-packing
 *      -- the weights have not changed since the last refresh (a backward pass issued separately from its forward pass). */
int gx_weight_cache_activate(int id);
int gx_weight_cache_destroy(int id);

/* ---- contexts.  Library state that outlives a call -- the deferred-reduction queues, queued weight-gradient jobs, the
 *      packed-weight cache a step records / is served from, per-kernel profiling records -- belongs to a context; a
 *      thread works in its current context (thread-local, context 0 by default).  One context per training loop makes
 *      the library re-entrant: loops in different threads, or interleaved in one thread, never share a queue.
 *      gx_ctx_create returns an id > 0 (negative: error). */
int gx_ctx_create(void);
int gx_ctx_make_current(int id);
int gx_ctx_current(void);
int gx_ctx_destroy(int id);

/* ---- deferred parameter-gradient reductions.  gx_conv3x3_wgrad, gx_deconv5x5s2_wgrad and gx_gn_relu_bwd each end
 *      in a small reduce launch whose result (dw / dgamma, dbeta) only the optimiser reads.  While
 *      gx_defer_enable(1) is in effect those calls queue that reduce instead (up to 48 of each kind; beyond that they
 *      reduce immediately) and gx_defer_flush(stream) finishes all queued ones in one launch per kind, ADDING each
 *      result to its destination (which the caller has zeroed: a gradient bucket).  Contract: the
 *      workspaces and output buffers handed to the queued calls stay valid and untouched until the flush.
 *      gx_defer_enable(-1) switches deferral off and discards the queue without running it (error recovery). */
int gx_defer_enable(int on);
int gx_defer_pending(void);
int gx_defer_flush(gx_stream_t stream);

/* ---- conv -> GroupNorm without the split-K reduce pass.  Layers that cannot fill the chip split the channel
 *      reduction into `nsplit` partial output slabs which a small reduce kernel normally sums.  The *_parts variants
 *      stop before that reduce and report where the slabs are (*parts: inside ws when *nsplit > 1, else = y, already
 *      complete; *split_stride floats apart; no bias applied); gx_gn_relu_fwd_parts sums them while it reads its
 *      input (same order as the reduce kernel: identical bits), adds the conv bias (may be NULL), writes the summed
 *      pre-norm tensor to y_sum (the backward pass needs it; may alias parts when nsplit == 1) and continues as
 *      gx_gn_relu_fwd.  ws must stay untouched until gx_gn_relu_fwd_parts has run. */
int gx_conv3x3_fwd_parts(const float* x, const float* w, float* y, int N, int Cin, int Cout, int H, int W, void* ws,
                         size_t ws_bytes, const float** parts, int* nsplit, size_t* split_stride, gx_stream_t stream);
int gx_deconv5x5s2_fwd_parts(const float* x, const float* w, float* y, int N, int Cin, int Cout, int Hin, int Win,
                             void* ws, size_t ws_bytes, const float** parts, int* nsplit, size_t* split_stride,
                             gx_stream_t stream);
int gx_gn_relu_fwd_parts(const float* parts, int nsplit, size_t split_stride, const float* conv_bias, float* y_sum,
                         const float* gamma, const float* beta, int N, int C, int H, int W, int groups, float eps,
                         float* dst0, int dst0_ctot, int dst0_c0, int dst0_mode, float* dst1, int dst1_ctot,
                         int dst1_c0, int dst1_mode, float* mean, float* rstd, gx_stream_t stream);

/* ---- segmentation metrics on the device (utils/misc.py:101-114 average_ari, :173-235 average_segcover):
 *      counts[b][i][j] = #{p : segA[b][p] == i and segB[b][p] == j}, int32 [B, KA, KB+1]; column KB collects segB labels
 *      outside [0,KB); pixels with segA outside [0,KA) (negative = ignore regions) are skipped.  Labels are int64
 *      [B, HW] as the reference's instance maps / argmax outputs are.  Bit-exact integer work. */
int gx_label_contingency(const long long* segA, const long long* segB, int B, int HW, int KA, int KB, int* counts,
                         gx_stream_t stream);

/* ---- input feeder (datasets/multid_config.py:131-135 ToTensor + F.interpolate(size), multi_object_config.py:176-186):
 *      uint8 frames [B, Hs, Ws, C] (HWC, as stored) -> fp32 [B, C, H, W] = value / 255, nearest-neighbour resampled to
 *      H x W when the stored size differs (F.interpolate's default mode).  Bit-exact against the torch ops. */
int gx_u8hwc_to_f32chw(const unsigned char* src, float* dst, int B, int Hs, int Ws, int C, int H, int W,
                       gx_stream_t stream);

/* ---- the step's one collective without PyTorch (the reference's only multi-GPU mode is nn.DataParallel,
 *      train.py:153-155: replicas gathered on GPU 0 every iteration).  One process per GPU; each rank's flat fp32 gradient
 *      bucket (parameters' gradients + the err / kl tail, genesis_amd/dp.py) is summed IN PLACE over the ranks by one
 *      ncclAllReduce over RCCL / xGMI, enqueued on `stream` (capturable into the step's HIP graph).  RCCL is resolved at run
 *      time (a copy the process already holds, else librccl.so.1; GENESIS_RCCL_LIB overrides), so the library loads without it.
 *      Rank 0 calls gx_allreduce_unique_id and hands the gx_allreduce_unique_id_bytes() (= 128) bytes to every rank out of band
 *      (file, socket, MPI, a torch store); every rank then calls gx_allreduce_init on its device (collective: it returns when
 *      all `world` ranks have joined). */
size_t gx_allreduce_unique_id_bytes(void);
int gx_allreduce_unique_id(void* id, size_t id_bytes);
int gx_allreduce_init(const void* id, size_t id_bytes, int rank, int world, void** comm);
int gx_allreduce_run(void* comm, float* buf, size_t count, gx_stream_t stream);
int gx_allreduce_destroy(void* comm);

/* ---- measurement probe: `wgs` workgroups x 4 waves x 32 * iters v_mfma_f32_32x32x2_f32.  mode 0: register operands
 *      only (the fp32-MFMA ceiling); 1: B operand from LDS; 2: A and B from LDS (two ds_read_b32 per MFMA, the tap-conv
 *      pattern); 3: as 2 plus a workgroup barrier every 32 MFMAs; 4: A and B from LDS with one 16-byte read per four MFMAs;
 *      5: the gx_kq.hip inner loop (four 16-byte reads per 16 MFMAs, issued a step ahead); 6: the same on 2 x 4 tiles; 5 / 6
 *      store the shader clock they ran at (s_memtime ticks per 100 MHz tick) in scratch[1].  *flops = executed flops; time the stream around it
 *      (tools/mfma_peak.py). */
int gx_mfma_fp32_probe(int wgs, int iters, int mode, float* scratch, double* flops, gx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GENESIS_HIP_H */
