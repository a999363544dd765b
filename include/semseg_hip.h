/* libsemseg_hip.so -- C ABI of the MI355X (gfx950) hot path that replaces the torch
 * operators called on SegmentationModule.forward/backward of
 * CSAILVision/semantic-segmentation-pytorch (reference: /root/reference/mit_semseg).
 *
 * The reference is pure Python: it has no FFI for this path.  Each entry point below
 * therefore names the torch call site (reference file:line) whose arithmetic it
 * replaces; INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; fp32 unless said
 *  - activations are NHWC: element (n,h,w,c) at ((n*H+h)*W+w)*ld + c, ld >= C is the
 *    pixel stride in floats (ld > C addresses a channel slice of a wider concat buffer)
 *  - conv weights are KRSC (= a torch [K,C,R,S] tensor in channels_last memory format)
 *  - `stream` is a hipStream_t; nothing allocates, synchronises or reads back to the
 *    host, so every call can be captured into a hipGraph
 *  - return value: 0 on success, a hipError_t (>0) from the launch, or SEMSEG_E* (<0)
 */
#ifndef SEMSEG_HIP_H
#define SEMSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEMSEG_EINVAL   (-1)   /* bad argument (shape/alignment not supported) */
#define SEMSEG_EWORKSPACE (-2) /* workspace too small */
#define SEMSEG_ECOMM   (-3)   /* RCCL not loadable in this process, or an RCCL call failed */

int semseg_abi_version(void);

/* ---------------- convolution (nn.Conv2d; resnet.py:18-21,61-66,130; models.py:163,406,448,
 *                  456,461,519-540; hrnet.py:26-29,188-205,316-338) ---------------------- */

/* bytes of scratch the three conv entry points may need for this geometry (split-K / split-M
 * partial sums).  Pass a buffer at least this large as `workspace`. */
size_t semseg_conv2d_workspace_bytes(int N, int H, int W, int C, int K, int R, int S,
                                     int stride, int pad, int dil);

/* y[n,oh,ow,k] = bias[k] + sum_{r,s,c} x[n, oh*stride-pad+r*dil, ow*stride-pad+s*dil, c] * w[k,r,s,c]
 * x: [N,H,W,C] (ld x_ld), w: [K,R,S,C], bias: [K] or NULL, y: [N,OH,OW,K] (ld y_ld).
 * Any C/K; C % 4 == 0 with 16-byte aligned x (x_ld % 4 == 0) takes the float4 path, anything else
 * (the 3-channel stem, resnet.py:100) a scalar-load path of the same kernel. */
int semseg_conv2d_fwd(const float* x, int x_ld, const float* w, const float* bias, float* y, int y_ld,
                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                      void* workspace, size_t workspace_bytes, void* stream);

/* dx = conv2d_fwd^T(dy).  wt: the weights transposed to [C,R,S,K] by semseg_weight_krsc_to_crsk.
 * dy: [N,OH,OW,K] (ld dy_ld), dx: [N,H,W,C] (ld dx_ld). */
int semseg_conv2d_dgrad(const float* dy, int dy_ld, const float* wt, float* dx, int dx_ld,
                        int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                        void* workspace, size_t workspace_bytes, void* stream);

/* dw[k,r,s,c] = sum_{n,oh,ow} dy[n,oh,ow,k] * x[n, oh*stride-pad+r*dil, ow*stride-pad+s*dil, c]
 * db[k] (optional) = sum dy[.,k]. */
int semseg_conv2d_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, float* dw, float* db,
                        int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                        void* workspace, size_t workspace_bytes, void* stream);

/* [K,R,S,C] -> [C,R,S,K] (T = R*S taps) */
int semseg_weight_krsc_to_crsk(const float* w, float* wt, int K, int T, int C, void* stream);

/* ---------------- convolution, fp32-accurate on the bf16 MFMA ("s3": 3-way bf16 split, 6 products) -------------
 * Same call sites as above.  Every fp32 operand is first split into three bf16 planes
 *     v = v0 + v1 + v2,  v0 = bf16(v), v1 = bf16(v - v0), v2 = bf16(v - v0 - v1)
 * by semseg_split3; the conv entry points consume ONLY split operands and accumulate the six products of weight
 * >= 2^-16 in fp32 (error ~2^-23 per product: fp32 class, see csrc/conv_s3.hip).
 * Split layout: `xs` = bf16 [3][rows][pitch] holding Cp = C rounded up to 32 channels per row (zero filled);
 * pitch = Cp, or Cp + 128 when Cp*2 bytes is a multiple of 2 KB (L2-channel skew); size = semseg_split3_bytes; 16-byte aligned.
 *   activations : rows = N*H*W pixels (NHWC)
 *   weights fwd : rows = K*R*S of the KRSC tensor;  dgrad: rows = C*R*S of the CRSK tensor (semseg_weight_krsc_to_crsk) */
size_t semseg_split3_bytes(int rows, int C);
int semseg_split3(const float* x, int x_ld, void* xs, int rows, int C, void* stream);
/* scratch (bytes) for the three *_s3 conv entry points */
size_t semseg_conv2d_s3_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
/* y = conv(x, w) + bias.  xs: split of x [N*H*W][C]; ws: split of w [K*R*S][C]. */
int semseg_conv2d_fwd_s3(const void* xs, const void* ws, const float* bias, float* y, int y_ld,
                         int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                         void* workspace, size_t workspace_bytes, void* stream);
/* dx = conv^T(dy).  dys: split of dy [N*OH*OW][K]; wts: split of the CRSK weights [C*R*S][K]. */
int semseg_conv2d_dgrad_s3(const void* dys, const void* wts, float* dx, int dx_ld,
                           int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                           void* workspace, size_t workspace_bytes, void* stream);
/* dw[k,r,s,c] = sum_m dy[m,k] * x[pix(m,r,s),c].  xs: split of x [N*H*W][C]; dys: split of dy [N*OH*OW][K]. */
int semseg_conv2d_wgrad_s3(const void* xs, const void* dys, float* dw,
                           int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                           void* workspace, size_t workspace_bytes, void* stream);
/* Pin the launch plan of one conv geometry (host-side tuner, mit_semseg/tuner.py).  pass: 0 fwd, 1 dgrad, 2 wgrad;
 * tile: fwd/dgrad 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 256x128 LDS-DMA; wgrad 0 = 128x128, 1 = 64x64; split >= 1 (split-K / split-M factor,
 * clamped to the k-tile count).  tile < 0 removes the pin (the built-in wave-quantisation heuristic then applies).
 * semseg_conv2d_s3_workspace_bytes reflects the pinned plans. */
int semseg_conv2d_s3_set_plan(int pass, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                              int tile, int split);
/* ---------------- convolution, fp32-accurate on the fp16 MFMA ("h2": 2-way fp16 split, 3 products) --------------
 * Same call sites and argument meaning as the *_s3 family; half the MFMA work.  semseg_split_h2 first finds the
 * tensor's max |x| (one extra read pass), picks the exponent e with 2^e * max in [2^14, 2^15), and stores
 *     X = 2^e x = X0 + X1 (+ rho),  X0 = fp16(X), X1 = fp16(X - X0)      |rho| <= max(2^-22 |X|, 2^-40 max|X|)
 * The conv entry points accumulate X0 W0 + X0 W1 + X1 W0 in fp32 and descale by 2^-(ex+ew) in their epilogue
 * (per-product error ~2^-21: below the fp32 accumulation noise of any reduction >= 27 terms; csrc/conv_split.hip).
 * Split layout: fp16 [2][rows][pitch] (pitch as for s3) + 256 B of zeros + a 4352-byte header (int32 e, partial maxima);
 * size = semseg_split_h2_bytes; 16-byte aligned.
 * semseg_conv2d_h2_set_plan pass 3 pins the tile (0, 6..10 or 14; split 1) of the batched Winograd forward GEMM, keyed
 * (N = tiles, H = W = 1, C, K, 3, 3, 1, 1, 1).
 * Tile ids for semseg_conv2d_h2_set_plan -- fwd/dgrad (pass 0/1): 0..3 as s3, 4 = 256x128 LDS-DMA 3-slot ring, 5 = 256x256
 * LDS-DMA, 6 = 128x128 LDS-DMA (two blocks per CU), 7/8/9 = 3/5/6 with software-pipelined fragment reads (one barrier
 * per k-tile), 10 = the ring 4 with the same pipeline; wgrad (pass 2): 0 = 128x128, 1 = 64x64 register staged, 2 = 128x128,
 * 3 = 256x128 ring, 4 = 256x256 LDS-DMA, 5/6 = 2/4 software pipelined.  Every variant computes the same sums (only the
 * fp32 summation order differs); tests/test_gpu_ops.py::test_h2_conv_every_tile_pinned runs each one. */
size_t semseg_split_h2_bytes(int rows, int C);
int semseg_split_h2(const float* x, int x_ld, void* xs, int rows, int C, void* stream);
/* semseg_split_h2 when an upper bound of max|x| is known as `nbounds` (<= 8) DEVICE scalars (max of them is used): one launch,
 * no absmax pass over x.  semseg_bound_sum: out[0] = the sum of such scalars (rounded up) -- the bound of a sum of tensors
 * (hrnet.py:231-248) from the bounds of its terms. */
int semseg_split_h2_bounds(const float* x, int x_ld, void* xs, int rows, int C, const float* const* bounds_host, int nbounds,
                           void* stream);
int semseg_bound_sum(const float* const* bounds_host, int nbounds, float* out, void* stream);
/* out[0] = max |x| over the fp32 window [rows][0..C) with row stride x_ld (a NaN anywhere gives NaN); workspace >= 4 KiB.
 * The bound the Winograd input transform needs when no producer kernel carried one (evaluation-mode forward). */
int semseg_absmax(const float* x, int x_ld, int rows, int C, float* out, void* workspace, size_t workspace_bytes,
                  void* stream);
/* Weight gradients of a whole backward pass reduced in ONE launch (round 4).  semseg_conv2d_wgrad_slabs_h2 = the kernel of
 * semseg_conv2d_wgrad_h2 on the same launch plan with its per-pixel-chunk partial sums left in `slabs` ([*splits_out][K*R*S*C] fp32,
 * KRSC order; semseg_conv2d_wgrad_slabs_bytes bytes, 16-byte aligned) and no reduce launch; semseg_reduce_slabs_multi sums the slabs
 * of any number of such tensors -- out = slabs[0] + slabs[1] + ... in slab order, the order of the per-tensor reduce, so the result is
 * bit-identical -- once per training step (autograd of nn.Conv2d.weight at every call site of the path; train.py:44 loss.backward()). */
typedef struct {
    const float* slabs; float* out; int64_t numel; int splits;
} semseg_slab_tensor;
size_t semseg_conv2d_wgrad_slabs_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
int semseg_conv2d_wgrad_slabs_h2(const void* xs, const void* dys, float* slabs, size_t slabs_bytes, int* splits_out,
                                 int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, void* stream);
int semseg_reduce_slabs_multi(const semseg_slab_tensor* tensors_host, int n, void* stream);
/* MANY weight gradients as slabs in ONE launch (round 4): the weight gradients are read by the optimizer only, so the host may hold
 * the small ones of a backward pass back and run their blocks side by side, 24 problems per launch and the longest blocks first, instead of one launch of 2 - 60 blocks each (HRNetV2: 241 such
 * launches).  Problem i leaves problems[i].splits partial sums in problems[i].slabs exactly as semseg_conv2d_wgrad_slabs_h2 would
 * (same launch plan, same blocks, same summation order: same bits).  Only geometries whose launch plan is the register-staged 64 x 64
 * tile -- semseg_conv2d_wgrad_tile_h2(geometry) == 1 -- can be batched; any other -> SEMSEG_EINVAL before anything is launched. */
typedef struct {
    const void* xs; const void* dys; float* slabs; size_t slabs_bytes;
    int N, H, W, C, K, R, S, stride, pad, dil;
    int splits;     /* out */
} semseg_wgrad_problem;
int semseg_conv2d_wgrad_tile_h2(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
int semseg_conv2d_wgrad_multi_h2(semseg_wgrad_problem* problems_host, int n, void* stream);
/* member mode of the weight-gradient plans (round 6): while on, a pinned plan of a SMALL weight tensor (at most 4 tiles of 64 x 64:
 * K, C <= 128; SEMSEG_WGRAD_MEMBER_TILES) on any other tile is read as the register-staged 64 x 64 tile with the same number of
 * blocks -- the form semseg_conv2d_wgrad_multi_h2 batches.  The host turns it on around the calls of its deferral only
 * (semseg_conv2d_wgrad_tile_h2 / _slabs_bytes / _slabs_h2 / _multi_h2), never while a plan is being timed.  Returns the previous mode. */
int semseg_conv2d_wgrad_member_plan(int enable);
size_t semseg_conv2d_h2_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
int semseg_conv2d_fwd_h2(const void* xs, const void* ws, const float* bias, float* y, int y_ld,
                         int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                         void* workspace, size_t workspace_bytes, void* stream);
/* semseg_conv2d_fwd_h2 (no bias) that ALSO gathers the BatchNorm statistics of its result in the GEMM epilogue -- the conv -> BN
 * pairs of resnet.py:72-92 / models.py:160-167 without the statistics sweep over the conv output: the waves of a block that share a
 * column range combine the fp64 column sums / sums of squares and fp32 column min / max of their sub-tiles through LDS, and every
 * BLOCK ROW TILE writes one partial row into stats_ws (semseg_conv2d_fwd_stats_bytes(K) bytes = room for 512 rows), zeroes
 * *bound_word, and *parts_out (host) = the number of partial rows (= row tiles of the launch plan), to be handed to
 * semseg_bn_fwd_finish_fused.  *parts_out == 0: the launch plan of this geometry splits the reduction (or has more than 512
 * row tiles) and nothing was gathered -- run semseg_bn_fwd_stats_fused on y instead. */
size_t semseg_conv2d_fwd_stats_bytes(int K);
int semseg_conv2d_fwd_stats_h2(const void* xs, const void* ws, float* y, int y_ld,
                               int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                               void* workspace, size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes,
                               float* bound_word, int* parts_out, void* stream);
int semseg_conv2d_dgrad_h2(const void* dys, const void* wts, float* dx, int dx_ld,
                           int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                           void* workspace, size_t workspace_bytes, void* stream);
int semseg_conv2d_wgrad_h2(const void* xs, const void* dys, float* dw,
                           int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                           void* workspace, size_t workspace_bytes, void* stream);
int semseg_conv2d_h2_set_plan(int pass, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                              int tile, int split);
/* db[k] = sum_m dy[m,k]  (bias gradient of the classifier convs, models.py:461,463,540) */
int semseg_bias_grad(const float* dy, int dy_ld, float* db, int M, int K, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---------------- batch norm (lib/nn/modules/batchnorm.py:56-61 = F.batch_norm) ---------- */

/* scratch (bytes) for bn_stats / bn_bwd_reduce partial sums */
size_t semseg_bn_workspace_bytes(int P, int C);

/* stats[0..C) = sum_p z[p,c], stats[C..2C) = sum_p z[p,c]^2, stats[2C] = P   (all fp64).
 * z: [P,C] dense.  A data-parallel caller all-reduces `stats` (2C+1 doubles) before finalize
 * (replaces batchnorm.py:63-76,98-117). */
int semseg_bn_stats(const float* z, int P, int C, double* stats, void* workspace, size_t workspace_bytes,
                    void* stream);

/* mean = S/n; var = SS/n - mean^2 (biased); invstd = 1/sqrt(var+eps);
 * scale = gamma*invstd; shift = beta - mean*scale;
 * running_mean = (1-m)*running_mean + m*mean; running_var = (1-m)*running_var + m*var*n/(n-1).
 * num_batches_tracked[0] += 1 (int64, torch _BatchNorm buffer).  running_* / num_batches_tracked may be NULL.
 * mean/invstd/scale/shift: [C]. */
int semseg_bn_finalize(const double* stats, int C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked,
                       float momentum, float eps,
                       float* mean, float* invstd, float* scale, float* shift, void* stream);

/* eval mode: scale = gamma/sqrt(running_var+eps); shift = beta - running_mean*scale;
 * also fills mean = running_mean, invstd = 1/sqrt(running_var+eps). */
int semseg_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, int C,
                          float* mean, float* invstd, float* scale, float* shift, void* stream);

/* y = act(z*scale + shift (+ residual)), act = relu if relu!=0 (resnet.py:76-90 fused).
 * z,residual: [P,C] dense (residual may be NULL, res_ld its pixel stride); y: [P,C] with ld y_ld. */
int semseg_bn_apply(const float* z, const float* scale, const float* shift, const float* residual, int res_ld,
                    int relu, float* y, int y_ld, int P, int C, void* stream);

/* g = dy * (relu ? y>0 : 1);  sums[0..C) = sum_p g, sums[C..2C) = sum_p g*xhat, xhat=(z-mean)*invstd
 * (fp64; all-reduced by a data-parallel caller).  dgamma = sum g*xhat, dbeta = sum g (local, fp32). */
int semseg_bn_bwd_reduce(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                         const float* mean, const float* invstd, int relu, int P, int C,
                         double* sums, float* dgamma, float* dbeta,
                         void* workspace, size_t workspace_bytes, void* stream);

/* training: dz = gamma*invstd*(g - sums[c]/n - xhat*sums[C+c]/n), n = stats_count[0] (device fp64:
 *           the all-reduced element count, i.e. stats[2C] of the forward)
 * eval    : dz = gamma*invstd*g
 * dres (optional, dense [P,C]) = g  -- gradient of the fused residual input. */
int semseg_bn_bwd_apply(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                        const float* mean, const float* invstd, const float* gamma,
                        const double* sums, const double* stats_count, int training, int relu,
                        float* dz, float* dres, int P, int C, void* stream);

/* ---------------- BN kernels of the fused conv -> BN -> (ReLU) -> conv chain on the h2 path --------------------
 * The h2 split of a BN result does not need a pass over the result to find its exponent: per-channel min/max of z
 * (gathered by the statistics pass) bound |y| rigorously, and the backward sums bound |dz| (csrc/bn.hip).
 * Same reference call sites as the plain BN entry points above (batchnorm.py:56-61 and its autograd backward). */
size_t semseg_bn_mm_workspace_bytes(int P, int C);
/* semseg_bn_stats + zmm[0..C) = min_p z[p,c], zmm[C..2C) = max_p z[p,c] (fp32; local to this rank) */
int semseg_bn_stats_mm(const float* z, int P, int C, double* stats, float* zmm, void* workspace,
                       size_t workspace_bytes, void* stream);
/* the sweep half of semseg_bn_stats_mm alone (partial sums into `workspace`, semseg_bn_mm_workspace_bytes; no other output):
 * the cost a split-K launch plan adds to a conv -> BN pair, timed by the launch-plan tuner */
int semseg_bn_stats_mm_partial(const float* z, int P, int C, void* workspace, size_t workspace_bytes, void* stream);
/* semseg_bn_finalize + absmax_out[0] = an upper bound of max |act(BN(z) + residual)| given |residual| <= res_absmax[0]
 * (device scalars; res_absmax NULL = no residual; absmax_out may be NULL) and, if y_planes != NULL, the exponent word
 * of that h2 split buffer (P rows x C channels) derived from the bound. */
int semseg_bn_finalize_mm(const double* stats, const float* zmm, int C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          float momentum, float eps, int relu, const float* res_absmax,
                          float* mean, float* invstd, float* scale, float* shift,
                          float* absmax_out, void* y_planes, int P, void* stream);
/* semseg_bn_apply (y dense, ld C) that ALSO writes the h2 split planes of y into y_planes (semseg_split_h2_bytes(P, C)).
 * C % 8 == 0.  Exponent: blockbound == NULL -> the header word set by semseg_bn_finalize_mm; else the maximum of the
 * ceil(C/16) per-block bounds left by semseg_bn_fwd_stats_fused (the kernel then also publishes the header word and, if
 * absmax_out != NULL, the bound itself).  y == NULL (round 4): only the planes are written -- a BN output whose one consumer is the
 * next convolution's planes (bn1 / bn2 of a bottleneck, resnet.py:72-82) needs no fp32 copy. */
int semseg_bn_apply_h2(const float* z, const float* scale, const float* shift, const float* residual, int res_ld,
                       int relu, float* y, void* y_planes, int P, int C, const void* blockbound, float* absmax_out,
                       void* stream);
/* semseg_bn_apply_h2 (relu != 0) + the ReLU decisions as a bitmask: gate[p * C/8 + c/8] bit (c % 8) = (y[p][c] > 0), P * C/8 bytes.
 * semseg_bn_bwd_reduce_fused(_peer) and semseg_bn_bwd_apply_h2 take it in place of y when called with y = gate, y_ld = 0 (and no
 * gate_scale): the backward of a BN with residual (resnet.py:84-92) then reads 1 bit instead of 4 bytes per element for the gate. */
int semseg_bn_apply_h2_gate(const float* z, const float* scale, const float* shift, const float* residual, int res_ld,
                            int relu, float* y, void* y_planes, int P, int C, const void* blockbound, float* absmax_out,
                            void* stream, unsigned char* gate);
/* semseg_bn_bwd_reduce + gmax[c] = max_p |g[p,c]| */
int semseg_bn_bwd_reduce_mm(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                            const float* mean, const float* invstd, int relu, int P, int C,
                            double* sums, float* gmax, float* dgamma, float* dbeta,
                            void* workspace, size_t workspace_bytes, void* stream);
/* exponent word of dz_planes from |dz| <= |gamma| invstd (gmax + |sums[c]|/n + max|xhat| |sums[C+c]|/n)
 * (after the data-parallel all-reduce of `sums`) */
int semseg_bn_bwd_bound(const double* sums, const double* stats_count, const float* gmax, const float* zmm,
                        const float* mean, const float* invstd, const float* gamma, int C, int training,
                        void* dz_planes, int P, void* stream);
/* semseg_bn_bwd_apply writing dz ONLY as h2 split planes (P rows x C channels; C % 8 == 0); dres stays fp32.
 * gate_scale/gate_shift != NULL (BN without residual): the ReLU gate is recomputed as fmaf(z, scale, shift) > 0 -- the
 * forward's own decision -- and y is not read.  blockbound: as semseg_bn_apply_h2 (from semseg_bn_bwd_reduce_fused). */
int semseg_bn_bwd_apply_h2(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                           const float* mean, const float* invstd, const float* gamma,
                           const double* sums, const double* stats_count, int training, int relu,
                           void* dz_planes, float* dres, int P, int C,
                           const float* gate_scale, const float* gate_shift, const void* blockbound, void* stream);
/* Single-rank fused forms (no all-reduce between the partial sums and their use): 3 launches per BN pass instead of 4.
 * semseg_bn_fwd_stats_fused = semseg_bn_stats_mm + semseg_bn_finalize_mm, leaving ceil(C/16) per-block bounds (uint32 bit
 * patterns) in `blockbound` for semseg_bn_apply_h2.  semseg_bn_bwd_reduce_fused = semseg_bn_bwd_reduce_mm + the per-channel
 * part of semseg_bn_bwd_bound, leaving the bounds for semseg_bn_bwd_apply_h2. */
int semseg_bn_fwd_stats_fused(const float* z, int P, int C, double* stats, float* zmm, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                              float momentum, float eps, int relu, const float* res_absmax,
                              float* mean, float* invstd, float* scale, float* shift, void* blockbound,
                              void* workspace, size_t workspace_bytes, void* stream);
int semseg_bn_bwd_reduce_fused(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                               const float* mean, const float* invstd, const float* gate_scale, const float* gate_shift,
                               int relu, int P, int C, const double* stats_count, const float* zmm, const float* gamma,
                               int training, double* sums, float* dgamma, float* dbeta, void* blockbound,
                               void* workspace, size_t workspace_bytes, void* stream);
/* semseg_bn_fwd_stats_fused + the bound of |y| itself in absmax_out[0] (max over the per-block bounds): for a BN whose output is
 * not written as planes (no ReLU: shortcut / fuse-layer BNs, resnet.py:84-88, hrnet.py:188-205) -- semseg_bn_apply follows. */
int semseg_bn_fwd_stats_fused_bound(const float* z, int P, int C, double* stats, float* zmm, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                    float momentum, float eps, int relu, const float* res_absmax,
                                    float* mean, float* invstd, float* scale, float* shift, void* blockbound,
                                    void* workspace, size_t workspace_bytes, void* stream, float* absmax_out);
/* SyncBN forms of the two (batchnorm.py:63-117: the statistics are those of the batch over ALL ranks): `peer` is a context of
 * the one-node peer exchange below (semseg_peer_create, every peer attached).  The finish kernel pushes the per-channel
 * [sum, sum^2] (+ this rank's pixel count P) / [sum g, sum g xhat] to the peers' inboxes and continues with the sums over the
 * ranks: stats / sums hold the GLOBAL values afterwards (stats[2C] = total pixel count); dgamma / dbeta stay this rank's own
 * sums (parameter gradients are reduced with the gradient buckets); min / max and bounds stay rank-local.  Every rank of the
 * context must issue the same sequence of exchanges. */
int semseg_bn_fwd_stats_fused_peer(const float* z, int P, int C, double* stats, float* zmm, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                   float momentum, float eps, int relu, const float* res_absmax,
                                   float* mean, float* invstd, float* scale, float* shift, void* blockbound,
                                   void* workspace, size_t workspace_bytes, void* stream, void* peer, float* absmax_out /* nullable */);
/* the finish half of semseg_bn_fwd_stats_fused[_bound|_peer] alone, on the partial sums the convolution's epilogue gathered
 * (semseg_conv2d_fwd_stats_h2: `parts` rows in `partials`, which also zeroed *absmax_out's word): mean / invstd / scale /
 * shift, running statistics, per-block bounds -- no pass over the conv output.  peer / absmax_out: NULL or as above. */
int semseg_bn_fwd_finish_fused(const void* partials, size_t partials_bytes, int parts, int P, int C, double* stats, float* zmm,
                               const float* gamma, const float* beta, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, float momentum, float eps, int relu, const float* res_absmax,
                               float* mean, float* invstd, float* scale, float* shift, void* blockbound, void* stream,
                               void* peer /* nullable */, float* absmax_out /* nullable */);
int semseg_bn_bwd_reduce_fused_peer(const float* dy, int dy_ld, const float* y, int y_ld, const float* z,
                                    const float* mean, const float* invstd, const float* gate_scale, const float* gate_shift,
                                    int relu, int P, int C, const double* stats_count, const float* zmm, const float* gamma,
                                    int training, double* sums, float* dgamma, float* dbeta, void* blockbound,
                                    void* workspace, size_t workspace_bytes, void* stream, void* peer);

/* semseg_bn_bwd_reduce_fused(_peer) / semseg_bn_bwd_apply_h2 with the incoming gradient given as TWO addends, dy + dy2: the two
 * gradients that meet where the autograd graph forks (a block output feeds the next block's first conv AND its shortcut,
 * resnet.py:72-92; an encoder map feeds two heads, models.py:253-268).  torch adds them with a kernel of its own before the BN
 * backward reads the sum twice; here the sum is formed where it is consumed -- the same single fp32 add per element, so the results
 * are those of the materialised sum bit for bit.  dy2: [P][dy2_ld], dy2_ld >= C; peer: NULL on a single rank. */
int semseg_bn_bwd_reduce_fused_sum2(const float* dy, int dy_ld, const float* dy2, int dy2_ld, const float* y, int y_ld,
                                    const float* z, const float* mean, const float* invstd, const float* gate_scale,
                                    const float* gate_shift, int relu, int P, int C, const double* stats_count, const float* zmm,
                                    const float* gamma, int training, double* sums, float* dgamma, float* dbeta, void* blockbound,
                                    void* workspace, size_t workspace_bytes, void* stream, void* peer /* nullable */);
int semseg_bn_bwd_apply_h2_sum2(const float* dy, int dy_ld, const float* dy2, int dy2_ld, const float* y, int y_ld, const float* z,
                                const float* mean, const float* invstd, const float* gamma,
                                const double* sums, const double* stats_count, int training, int relu,
                                void* dz_planes, float* dres, int P, int C,
                                const float* gate_scale, const float* gate_shift, const void* blockbound, void* stream);

/* ---------------- elementwise helpers ------------------------------------------------------ */
/* out = act(a + b) (hrnet.py:231-248 fuse sums); a,b,out [P,C] with their own ld */
int semseg_add_act(const float* a, int a_ld, const float* b, int b_ld, int relu, float* out, int out_ld,
                   int P, int C, void* stream);
/* dx = dy * (y > 0) */
int semseg_relu_bwd(const float* dy, int dy_ld, const float* y, int y_ld, float* dx, int dx_ld,
                    int P, int C, void* stream);
/* nn.ReLU6's upper clamp (mobilenet.py:26,34; the lower clamp is the fused ReLU of the BN kernels): y = min(x, cap);
   backward dx = dy * (y < cap) -- the open interval of torch's hardtanh backward */
int semseg_clamp_max(const float* x, int x_ld, float cap, float* y, int y_ld, int P, int C, void* stream);
int semseg_clamp_max_bwd(const float* dy, int dy_ld, const float* y, int y_ld, float cap, float* dx, int dx_ld,
                         int P, int C, void* stream);
/* nn.Dropout2d's per-(n,c) keep mask (models.py:460,464): mask[i] = Bernoulli(1-p) / (1-p), i < n, from a counter-based hash;
   state = uint64[2] {seed, launch counter} in device memory, the counter is advanced by the kernel (fresh mask per hipGraph
   replay) */
int semseg_dropout_mask(float* mask, int n, float p, void* state, void* stream);
/* inference head fused (models.py:480-484,578-582; eval.py:66-71): out[N,OH,OW,C] (+)= weight * softmax_C(bilinear(logits
   [N,IH,IW,C], align_corners=False) at (OH, OW)); accumulate != 0 adds to `out` (the multi-scale average, weight = 1/#scales) */
int semseg_upsample_softmax(const float* logits, int x_ld, float* out, int out_ld, int accumulate, float weight,
                            int N, int IH, int IW, int OH, int OW, int C, void* stream);
/* strided 2-D copy (torch.cat slices, models.py:424,476,553,575; hrnet.py:434) */
int semseg_copy2d(const float* src, int src_ld, float* dst, int dst_ld, int P, int C, int accumulate,
                  void* stream);
/* y[n,p,c] = x[n,p,c] * mask[n,c]  (nn.Dropout2d models.py:460,464 with an explicit mask; fwd == bwd) */
int semseg_scale_nc(const float* x, const float* mask, float* y, int N, int HW, int C, void* stream);
/* NCHW <-> NHWC layout changes at the API edge */
int semseg_nchw_to_nhwc(const float* x, float* y, int N, int C, int HW, void* stream);
int semseg_nhwc_to_nchw(const float* x, float* y, int N, int C, int HW, void* stream);

/* ---------------- pooling / resize ---------------------------------------------------------- */
/* nn.MaxPool2d(3,2,1) resnet.py:109.  idx: uint8 [N,OH,OW,C] window position (r*3+s) of the first max */
int semseg_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
int semseg_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C,
                            void* stream);
/* nn.AdaptiveAvgPool2d(s) models.py:447,511: bins [floor(i*H/s), ceil((i+1)*H/s)) */
int semseg_adaptive_avgpool_fwd(const float* x, int x_ld, float* y, int N, int H, int W, int C, int OH, int OW,
                                void* stream);
int semseg_adaptive_avgpool_bwd(const float* dy, float* dx, int dx_ld, int accumulate, int N, int H, int W, int C,
                                int OH, int OW, void* stream);
/* F.interpolate(bilinear, align_corners=False) models.py:420-423,472-475,549-552,561-574; hrnet.py:241-245,427-432
 * x: [N,IH,IW,C] ld x_ld; y: [N,OH,OW,C] ld y_ld; accumulate: y += (FPN top-down add, HRNet fuse);
 * relu: y = max(y, 0) after the (accumulated) result (hrnet.py:248) */
int semseg_bilinear_fwd(const float* x, int x_ld, float* y, int y_ld, int accumulate, int relu,
                        int N, int IH, int IW, int OH, int OW, int C, void* stream);
/* dx[N,IH,IW,C] (+)= bilinear^T(dy[N,OH,OW,C]) -- gather form, deterministic */
int semseg_bilinear_bwd(const float* dy, int dy_ld, float* dx, int dx_ld, int accumulate,
                        int N, int IH, int IW, int OH, int OW, int C, void* stream);

/* Pyramid pooling: nn.AdaptiveAvgPool2d of the SAME map to several square grids at once (models.py:447-450,511-514;
 * scales 1, 2, 3, 6): the map is read once (forward) / the gradient written once (backward) instead of once per scale plus
 * three accumulation passes.  sizes_host / ys_host / dys_host: host arrays of nscale (<= 4, sum of sizes <= 16) entries;
 * ys[i] / dys[i]: device [N][s_i][s_i][C] dense. */
size_t semseg_adaptive_avgpool_multi_workspace_bytes(int N, int H, int C, const int* sizes_host, int nscale);
int semseg_adaptive_avgpool_multi_fwd(const float* x, int x_ld, int N, int H, int W, int C, int nscale,
                                      const int* sizes_host, void* const* ys_host, void* workspace, size_t workspace_bytes,
                                      void* stream);
int semseg_adaptive_avgpool_multi_bwd(void* const* dys_host, const int* sizes_host, int nscale, float* dx, int dx_ld,
                                      int N, int H, int W, int C, void* stream);

/* ---------------- head: softmax / NLL / pixel accuracy -------------------------------------- */
/* F.log_softmax(dim=1) models.py:383,492-493,584 on [P,C] rows (C <= 1024) */
int semseg_log_softmax_fwd(const float* z, float* logp, int P, int C, void* stream);
int semseg_log_softmax_bwd(const float* dlogp, const float* logp, float* dz, int P, int C, void* stream);
/* F.softmax(dim=1) of the inference branch models.py:482-483 */
int semseg_softmax_fwd(const float* z, float* prob, int P, int C, void* stream);
/* nn.NLLLoss(ignore_index) train.py:154 + pixel_acc models.py:12-18 in one pass over logp [P,C]:
 * out[0] = -sum_valid logp[p,label]/n_valid, out[1] = acc = hits/(n_valid+1e-10), out[2] = n_valid (fp32).
 * label: int64 [P]. */
int semseg_nll_acc_fwd(const float* logp, const int64_t* label, int ignore_index, int P, int C,
                       float* out, void* workspace, size_t workspace_bytes, void* stream);
/* dlogp[p,c] = -(gloss[0]/n_valid) if c==label[p] (valid) else 0 */
int semseg_nll_bwd(const float* gloss, const float* nll_out, const int64_t* label, int ignore_index,
                   float* dlogp, int P, int C, void* stream);

/* ---------------- weight preparation for the h2 convolutions (multi-tensor) ------------------
 * For every conv weight w [K][T][C] (KRSC, T = R*S): the h2 split of w (rows K*T, channels C: operand of fwd) and of
 * its CRSK transpose (rows C*T, channels K: operand of dgrad), sharing one exponent -- bit-identical to
 * semseg_split_h2(w) and semseg_split_h2(semseg_weight_krsc_to_crsk(w)), in 2 launches per 64 tensors instead of 5 per
 * tensor.  The weights change only in the optimiser step (train.py:117-126): call once after semseg_sgd_step.
 * krsc / crsk: buffers of semseg_split_h2_bytes(K*T, C) / semseg_split_h2_bytes(C*T, K) bytes. */
typedef struct {
    const float* w; void* krsc; void* crsk; int K, T, C;
    void* wino;   /* optional, T == 9 only: Winograd-transformed weights U = G g G^T as h2 planes, rows 16*K (f, k), channels C:
                     semseg_split_h2_bytes(16*K, C) bytes; NULL = not wanted */
    void* wino_t; /* optional, T == 9 only: the transformed weights of the DATA GRADIENT (taps flipped, C and K swapped), rows 16*C
                     (f, c), channels K: semseg_split_h2_bytes(16*C, K) bytes; NULL = not wanted */
} semseg_wprep_tensor;
int semseg_weights_prepare_h2(const semseg_wprep_tensor* tensors_host, int n, void* stream);
/* the partial-maximum slots of a conv weight's KRSC plane buffer (device address inside `krsc_planes`), for semseg_sgd_step_fused;
 * and the preparation that trusts them (have_absmax != 0: EVERY tensor of the call had its slots filled by the fused SGD kernel
 * since its weights last changed; 0: as semseg_weights_prepare_h2) */
void* semseg_weights_absmax_slots(void* krsc_planes, int K, int T, int C);
int semseg_weights_prepare_h2_after_sgd(const semseg_wprep_tensor* tensors_host, int n, int have_absmax, void* stream);

/* ---------------- evaluation metrics (eval.py:74-84, utils.py:128-156) -----------------------
 * pred[p] = argmax_c scores[p, c] (first maximum, as torch.max) on [P, C] rows with pixel stride ld; with `label`
 * (int64 [P], < 0 = unlabeled) the exact integer tallies of accuracy() and intersectionAndUnion() are ADDED to
 * counts (int64 [2 + 3C]: acc_sum, valid_sum, area_intersection[C], area_pred[C], area_lab[C]; area_union =
 * area_pred + area_lab - area_intersection).  pred_out may be NULL; label/counts may be NULL (argmax only). */
int semseg_argmax_metrics(const float* scores, int ld, const int64_t* label, int P, int C, int64_t* pred_out,
                          int64_t* counts, void* stream);

/* the same tallies from an existing prediction map (pred, label: int64 [P]); counts as above, accumulated */
int semseg_label_metrics(const int64_t* pred, const int64_t* label, int P, int C, int64_t* counts, void* stream);

/* ---------------- Winograd F(2x2, 3x3) forward convolution on h2 planes (csrc/winograd.hip) -------------
 * For 3x3, stride 1, pad == dil convolutions (resnet.py:61-66 as dilated by models.py:209-251; models.py:163,456-457):
 *   semseg_winograd_input_h2 : x (fp32 NHWC) -> V = B^T d B as h2 planes, rows 16*tiles (f, tile), channels C;
 *                              bounds_host: nbounds (<= 8) DEVICE scalars, max|x| <= max of them (exponent = f(4*max))
 *   semseg_winograd_gemm_h2  : M[f] = V[f] U[f]^T for the 16 frequencies in one batched launch; M fp32 [16*tiles][K]
 *   semseg_winograd_output   : z = A^T M A (fp32 NHWC, ld z_ld)
 * tiles = semseg_winograd_tiles(N, H, W, dil); U comes from semseg_weights_prepare_h2 (field `wino`).  2.25x fewer MFMA
 * products than semseg_conv2d_fwd_h2, same error class. */
int semseg_winograd_tiles(int N, int H, int W, int dil);
int semseg_winograd_input_h2(const float* x, int x_ld, const float* const* bounds_host, int nbounds, void* v_planes,
                             int N, int H, int W, int C, int dil, void* stream);
/* the input transform with the source given as h2 planes ([N*H*W] rows x C channels; exponent of V = exponent of the source - 2):
 * the DATA GRADIENT of the same layers in the Winograd domain is  input_planes(dz) -> gemm(V, U' = field `wino_t` of
 * semseg_weights_prepare_h2, tiles, C = filters, K = input channels) -> output(dx): the 3x3 stride-1 convolution of dz with the
 * flipped, transposed weights */
int semseg_winograd_input_planes_h2(const void* src_planes, void* v_planes, int N, int H, int W, int C, int dil, void* stream);
int semseg_winograd_gemm_h2(const void* v_planes, const void* u_planes, float* M, int tiles, int C, int K, void* stream);
int semseg_winograd_output(const float* M, float* z, int z_ld, int N, int H, int W, int K, int dil, void* stream);
/* gemm + output in ONE launch (round 4): every block owns [128 tiles] x [128 output channels] for all 16 frequencies -- an f-loop
 * over the same LDS-DMA ring, one accumulator set for M[f], four for the 2 x 2 outputs -- and writes z = A^T M A as pixels; the fp32
 * intermediate M (16 * tiles * K floats written by _gemm_h2 and read back by _output) never exists.  C = channels of V / U (the
 * reduction), K = output channels; used for the data gradients of the wide 3x3 convs (models.py:455-465, 538-540; there C = the
 * layer's filters, K = its input channels, u_planes = field `wino_t`).  form: 0 / 1 / 2 = 8 waves (32 x 64 per wave) on a 3- / 4- / 5-slot ring, 3 / 4 = 4 waves (64 x 64 per wave) on 4 / 5 slots, 5 / 6 = 64-deep
 * k-tiles (C % 64 == 0 after padding to 32) with full 128-byte lines per DMA piece on a ring of five half tiles, 8 / 4 waves. */
int semseg_winograd_gemm_output_h2(const void* v_planes, const void* u_planes, float* z, int z_ld,
                                   int N, int H, int W, int C, int K, int dil, int form, void* stream);
/* Weight gradient of the same layers in the Winograd domain (the autograd of nn.Conv2d.weight at those call sites):
 *   semseg_winograd_dm_h2         : h2 planes of dz [N*H*W][K] -> dM = A dz A^T as h2 planes, rows 16*tiles (f, tile)
 *   semseg_winograd_wgrad_gemm_h2 : dU[f] = dM[f]^T V[f] (V = the forward's input transform), fp32 [16][K][C], one batched
 *                                   launch; workspace of semseg_winograd_wgrad_workspace_bytes for its split partials
 *   semseg_winograd_dg            : dw = G^T dU G, fp32 KRSC [K][3][3][C] */
int semseg_winograd_dm_h2(const void* dz_planes, void* dm_planes, int N, int H, int W, int K, int dil, void* stream);
size_t semseg_winograd_wgrad_workspace_bytes(int tiles, int C, int K);
int semseg_winograd_wgrad_gemm_h2(const void* v_planes, const void* dm_planes, float* dU, int tiles, int C, int K,
                                  void* workspace, size_t workspace_bytes, void* stream);
int semseg_winograd_dg(const float* dU, float* dw, int K, int C, void* stream);

/* ---------------- depthwise 3x3 convolution (csrc/depthwise.hip) -----------------------------------------------------
 * nn.Conv2d(C, C, 3, stride, padding, dilation, groups=C, bias=False) of MobileNetV2 (models/mobilenet.py:48,60, geometry
 * rewritten by models.py:297-311) and its autograd: fp32 NHWC, C % 4 == 0, weights TAP-MAJOR w_taps[9][C]
 * (w_taps[r*3+s][c] = weight[c][0][r][s]).  dgrad is a gather (deterministic); wgrad reduces per-chunk partial sums in a
 * fixed order (workspace of semseg_depthwise3x3_workspace_bytes).  Opt-in behind layers.GroupedConv2d
 * (SEMSEG_DEPTHWISE_DIRECT=1) until it has run on a GPU; the default is the block-diagonal dense path. */
size_t semseg_depthwise3x3_workspace_bytes(int N, int H, int W, int C, int stride, int pad, int dil);
int semseg_depthwise3x3_fwd(const float* x, int x_ld, const float* w_taps, float* y, int y_ld, int N, int H, int W, int C,
                            int stride, int pad, int dil, void* stream);
int semseg_depthwise3x3_dgrad(const float* dy, int dy_ld, const float* w_taps, float* dx, int dx_ld, int N, int H, int W, int C,
                              int stride, int pad, int dil, void* stream);
int semseg_depthwise3x3_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, float* dw_taps, int N, int H, int W, int C,
                              int stride, int pad, int dil, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------- grouped 3x3 convolution (csrc/grouped.hip) ------------------------------------------------------------
 * nn.Conv2d(C, K, 3, stride, padding, dilation, groups=g, bias=False) of the ResNeXt bottleneck (models/resnext.py:30-31,
 * g = 32) and its autograd: fp32 NHWC, (C/g) % 4 == 0 and (K/g) % 4 == 0, weights w_taps[K][9][C/g]
 * (w_taps[k][r*3+s][ci] = weight[k][ci][r][s]).  Same structure and determinism as the depthwise family; opt-in behind
 * layers.GroupedConv2d (SEMSEG_GROUPED_DIRECT=1) until it has run on a GPU. */
size_t semseg_grouped3x3_workspace_bytes(int N, int H, int W, int C, int K, int groups, int stride, int pad, int dil);
int semseg_grouped3x3_fwd(const float* x, int x_ld, const float* w_taps, float* y, int y_ld, int N, int H, int W, int C, int K,
                          int groups, int stride, int pad, int dil, void* stream);
int semseg_grouped3x3_dgrad(const float* dy, int dy_ld, const float* w_taps, float* dx, int dx_ld, int N, int H, int W, int C,
                            int K, int groups, int stride, int pad, int dil, void* stream);
int semseg_grouped3x3_wgrad(const float* x, int x_ld, const float* dy, int dy_ld, float* dw_taps, int N, int H, int W, int C,
                            int K, int groups, int stride, int pad, int dil, void* workspace, size_t workspace_bytes,
                            void* stream);

/* ---------------- input pipeline: training-batch assembly on the device (csrc/input_pipeline.hip) -------------------
 * Contract of mit_semseg/dataset.py:110-199 (TrainDataset.__getitem__) from DECODED uint8 arrays (the JPEG/PNG decode stays
 * on the host): `imresize(img, (w, h), 'bilinear')` = Pillow BILINEAR (dataset.py:9-19,167; antialiased triangle filter,
 * 22-bit fixed-point taps, horizontal then vertical pass), `imresize(segm, ..., 'nearest')` twice (dataset.py:168-179),
 * random flip (dataset.py:162-164), img_transform (dataset.py:53-58: /255, Normalize) and segm_transform (dataset.py:60-63:
 * long - 1), placement into the zero-initialised batch tensors (dataset.py:143-151,188-189).  Bit-exact against Pillow.
 *   host-side table builders (no GPU needed):
 *     semseg_input_resample_ksize / _coeffs : Pillow precompute_coeffs + normalize_coeffs_8bpc; bounds [out][2] =
 *                                             (first source index, tap count), kk [out][ksize] int32
 *     semseg_input_nearest_table            : Pillow ImagingScaleAffine source index per output index (-1 = outside)
 *   device kernels (uint8 HWC RGB in, pointers are device pointers, tables int32 on the device):
 *     semseg_input_resample_h_u8            : src [H][W][3] (mirrored when flip) -> tmp [H][ow][3]
 *     semseg_input_resample_v_normalize     : tmp -> fp32 NHWC slice of the batch (pixel pitch 3, row pitch BW*3):
 *                                             ((u8 / 255) - mean[c]) / std[c]; mean_std_host = {mean[3], std[3]} (host)
 *     semseg_input_label_gather             : dst[y][x] = (ytab[y] >= 0 && xtab[x] >= 0 ? src[ytab[y]][xtab[x]] : 0) - 1,
 *                                             int64, row pitch LW (both resizes, the canvas and the flip folded into the
 *                                             tables by the caller) */
int semseg_input_resample_ksize(int in_size, int out_size);
int semseg_input_resample_coeffs(int in_size, int out_size, int32_t* bounds_host, int32_t* kk_host);
int semseg_input_nearest_table(int in_size, int out_size, int32_t* tab_host);
int semseg_input_resample_h_u8(const uint8_t* src, int H, int W, int flip, const int32_t* bounds, const int32_t* kk, int ksize,
                               uint8_t* tmp, int ow, void* stream);
int semseg_input_resample_v_normalize(const uint8_t* tmp, int H, int ow, const int32_t* bounds, const int32_t* kk, int ksize,
                                      int oh, const float* mean_std_host, float* dst, int BW, void* stream);
int semseg_input_label_gather(const uint8_t* src, int W, const int32_t* ytab, const int32_t* xtab, int lh, int lw, int64_t* dst,
                              int LW, void* stream);

/* ---------------- optimiser (torch.optim.SGD, train.py:117-126) ----------------------------- */
typedef struct {
    float* param; const float* grad; float* momentum_buf; int64_t numel; float weight_decay; int first_step;
} semseg_sgd_tensor;
/* tensors_host: host array of n descriptors, copied into the kernel launch by value in chunks;
 * lr: device pointer to the current learning rate (so a captured graph can be replayed while
 * the poly schedule of train.py:130-139 changes it).  g = grad*grad_scale + wd*p; buf = m*buf + g
 * (buf = g on first_step); p -= lr*buf. */
int semseg_sgd_step(const semseg_sgd_tensor* tensors_host, int n, const float* lr, float momentum,
                    float grad_scale, void* stream);
/* The same update (train.py:115-127) as ONE pass over the weights where the step made three (round 6): a tensor may name
 *  - slabs / splits: its gradient still lies as `splits` partial sums of numel floats each (semseg_conv2d_wgrad_slabs_h2 /
 *    _wgrad_multi_h2): the kernel adds them in slab order 0, 1, 2, ... -- the bits semseg_reduce_slabs_multi would produce --
 *    writes the sum to `grad` and uses it; slabs == NULL: `grad` is read as before;
 *  - absmax_slots: 512 uint32 slots (semseg_weights_absmax_slots) that receive the block maxima of |w| of the UPDATED weights
 *    (bit patterns; unused slots zero): semseg_weights_prepare_h2_after_sgd(have_absmax = 1) then derives the exponent of
 *    the weight planes from them without a pass of its own over the weights.  NULL: not wanted. */
typedef struct {
    float* param; float* grad; float* momentum_buf; int64_t numel; float weight_decay; int first_step;
    const float* slabs; int splits; void* absmax_slots;
} semseg_sgd_tensor2;
int semseg_sgd_step_fused(const semseg_sgd_tensor2* tensors_host, int n, const float* lr, float momentum,
                          float grad_scale, void* stream);

/* ---------------- collectives over RCCL / xGMI ------------------------------------
 * Replace the reference's single-process thread rendezvous: lib/nn/modules/comm.py:46-131 (SyncMaster / SlavePipe queues),
 * lib/nn/modules/batchnorm.py:98-117 (_data_parallel_master: reduce sum / ssum to device 0, broadcast mean / inv_std) and the
 * gradient reduction nn.DataParallel performs (train.py:184-190).  One process per GPU; every rank calls the same sequence.
 * RCCL is bound at run time (the librccl the process already holds); without it these return SEMSEG_ECOMM and everything
 * else in the library keeps working.  All-reduces are enqueued on `stream` (no host sync; hipGraph-capturable). */
int semseg_comm_available(void);                       /* 1 if an RCCL library could be bound */
int semseg_comm_version(void);                         /* ncclGetVersion code, 0 if unavailable */
int semseg_comm_unique_id(void* id128);                /* rank 0: 128-byte id to hand to the other ranks (host-side rendezvous) */
int semseg_comm_init(int rank, int world, const void* id128, void** comm_out);     /* collective over all ranks */
int semseg_comm_count(void* comm, int* count_out);    /* ncclCommCount: ranks RCCL reports for this communicator */
int semseg_comm_allreduce_sum_f32(void* comm, float* buf, size_t count, void* stream);    /* in place: gradient buckets */
int semseg_comm_allreduce_sum_f64(void* comm, double* buf, size_t count, void* stream);   /* in place: BN [sum, sum^2, n] */
/* `n` payloads in one RCCL group (statistics of independent BN layers: PPM branches, HRNet branches) */
int semseg_comm_allreduce_sum_f64_multi(void* comm, double* const* bufs, const size_t* counts, int n, void* stream);
int semseg_comm_destroy(void* comm);

/* ---------------- one-node peer exchange for the SyncBN payloads (csrc/peer.hip) ----------------
 * The same replacement target as above (comm.py:46-131, batchnorm.py:98-117), for the latency-bound part: an all-reduce(sum)
 * of <= max_doubles doubles as ONE small kernel -- every rank stores its payload (data + tag in each 8-byte word) into every
 * peer's uncached inbox over xGMI and sums, in rank order, what arrives in its own.  No RCCL, no host step, capturable into a
 * hipGraph like any other kernel; the result is bit-identical on all ranks.
 *   create (every rank, current device) -> handle (64-byte hipIpcMemHandle_t of the inbox; the host side carries it to the
 *   other ranks) -> attach (one per peer; attach_local for contexts of the same process) -> allreduce ... -> destroy (after a
 *   host-side barrier).  world <= semseg_peer_max_world() ranks of ONE node.  A rank that waits longer than timeout_s poisons
 *   its result with NaN and raises semseg_peer_status() (SEMSEG_ECOMM) instead of hanging. */
int semseg_peer_max_world(void);
int semseg_peer_create(int rank, int world, int max_doubles, double timeout_s, void** peer_out);
int semseg_peer_set_timeout(void* peer, double timeout_s);   /* for exchanges launched from now on (self-test: short, run: long) */
int semseg_peer_handle(void* peer, void* handle64);
int semseg_peer_attach(void* peer, int src_rank, const void* handle64);
int semseg_peer_attach_local(void* peer, int src_rank, void* other_peer);
int semseg_peer_allreduce_sum_f64(void* peer, double* buf, size_t count, void* stream);   /* in place, on `stream` */
int semseg_peer_status(void* peer);                    /* 0, or SEMSEG_ECOMM once an exchange timed out (no device sync) */
int semseg_peer_destroy(void* peer);
/* widest BatchNorm (channels) whose in-kernel exchange (the _fused_peer entry points above: ceil(C / 16) blocks that wait for the
 * peers' payloads) fits the CURRENT device at once; checked by every rank when the exchange is built (comm.peer_init), so that no
 * rank can drop out of an exchange inside a step.  0: the occupancy query failed. */
int semseg_bn_peer_channel_capacity(void);

/* ---------------- box probes (bench.py `box` block and the backward timeline of its scaling model; no reference call site:
 *                  the reference measures nothing about the device it runs on, train.py:50-66 prints wall-clock averages only) ----
 * timestamp: one constant-rate timestamp (s_memrealtime) into the 8-byte aligned device word `slot` -- an ordinary kernel, so a
 *   captured training step can carry markers (when each gradient bucket of parallel.GradientBuckets is complete) and a replay
 *   yields the timeline; differences of two slots are scaled by the host with the replay's HIP-event time.
 * mfma_f16: `blocks` x 4 waves x `iters` x 8 back-to-back v_mfma_f32_32x32x16_f16 (2*32*32*16 flop each) on register operands;
 *   cycles[block] (uint64, may be NULL) = shader cycles (s_memtime) of wave 0's loop: flop / time = what the matrix pipes of
 *   this box sustain, cycles / time = the clock it holds while they are busy.
 * copy: 16 bytes per lane streaming copy of `bytes` (multiple of 16) -- achievable HBM bandwidth = 2 * bytes / time.
 * empty: a one-wave kernel that does nothing -- N of them in a captured chain measure the per-node cost of a hipGraph replay. */
int semseg_probe_timestamp(void* slot, void* stream);
int semseg_probe_mfma_f16(void* sink, int blocks, int iters, void* cycles, void* stream);
int semseg_probe_copy(const void* src, void* dst, size_t bytes, void* stream);
int semseg_probe_empty(void* stream);
/* operand ingest of a CU: `blocks` x 4 waves gather `steps` x 32 KiB each from a rows x pitch byte matrix in pieces of
 * (1024 / seg_bytes) rows x seg_bytes -- by LDS-DMA (dma != 0) or by plain 16-byte loads into registers; tools/probes/gather_ingest.py */
int semseg_probe_gather(const void* src, unsigned rows, unsigned pitch, int seg_bytes, int dma, int blocks, int steps, void* sink,
                        void* stream);

/* ---------------- side-by-side launches of independent sub-networks (csrc/batch.h, csrc/batch.hip) -----------------------------
 * HRNetV2's parallel branches (reference hrnet.py:225-227: `for i in range(self.num_branches): x[i] = self.branches[i](x[i])`)
 * run the same sequence of conv / BN launches on four geometries; the reference leaves them to four sequential cuDNN / ATen call
 * chains.  Between begin and end the launches of the converted kernel families are RECORDED per branch (every entry point still
 * does its host work -- plans, out-parameters -- at call time) and end() issues the records that sit at the same position of every
 * branch as ONE launch whose blocks find their problem in a table in the kernel arguments: same blocks, same arithmetic, same
 * order inside a branch -- bit-identical to the sequential launches.  One scope per process at a time.
 *   begin(branches, stream): open a scope of 1 ... 16 branches whose launches ALL leave on `stream` (whatever stream argument the
 *   calls inside carry -- the host runs every branch under a stream context of its own so that the caching allocator keeps the
 *   branches' temporaries apart); branch(i): the calls that follow belong to branch i; next_op(): the host has finished one C-ABI
 *   call of the current branch (launches pair up only under the same ordinal); flush(): issue what has been recorded, keep the
 *   scope; end(): flush + close, returns the first launch error of the scope; abort(): drop everything (a failed forward / backward);
 *   active(): 1 inside a scope; stats(out[4]): scopes opened, launches recorded, launches the zip issued for them, direct launches
 *   inside a scope (each forced a flush: an unconverted kernel on a branch's path). */
int semseg_batch_begin(int branches, void* stream);
int semseg_batch_branch(int index);
int semseg_batch_next_op(void);
int semseg_batch_next_unit(void);   /* the next unit of the branch (one conv -> BN node, forward or backward): ordinals restart at a multiple of 256 */
int semseg_batch_flush(void);
int semseg_batch_end(void);
int semseg_batch_abort(void);
int semseg_batch_active(void);
int semseg_batch_stats(long long* out4);
/* launch plan of the forward / data-gradient GEMMs recorded inside a scope (csrc/conv_split.hip batch_plan): ONE tile form without
 * split-K for every problem of at most max_tiles 64 x 64 tiles, so that the branches' GEMMs are the same kernel instantiation and
 * pair up; tile -1: the per-geometry plans (what the sequential launches run: bit-identical results); max_tiles <= 0: unchanged.
 * Defaults: SEMSEG_BATCH_TILE (-1: measured best on HRNetV2, csrc/conv_split.hip), SEMSEG_BATCH_MAX_TILES (1024).  Returns the previous tile. */
int semseg_batch_plan(int tile, int max_tiles);

#ifdef __cplusplus
}
#endif
#endif
