/*
 * nefii_amd.h - C ABI of the MI355X (gfx950) hot-path library  libnefii_hip.so
 *
 * The reference (FuxiComputerVision/Nefii) has no FFI/plugin boundary: its hot path is eager
 * PyTorch.  This header is the boundary a maintainer would bind instead
 * (ctypes stub in INTEGRATION.md).  Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is called h_* (host);
 *   - all tensors are contiguous float32 unless stated; masks are uint8 (0/1);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - nothing allocates, frees or synchronises: the caller owns every buffer including the
 *     workspaces, whose sizes come from the *_workspace_bytes() helpers (hipGraph-capturable);
 *   - return value: 0 = ok, negative = NEFII_E_* (bad argument), positive = hipError_t.
 */
#ifndef NEFII_AMD_H
#define NEFII_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEFII_ABI_VERSION 15
#define NEFII_MAX_LAYERS 12
#define NEFII_TILE_ROWS 32          /* points per workgroup tile */
#define NEFII_MAX_WIDTH 512         /* widest hidden layer / feature vector */
#define NEFII_MAX_ENC 96            /* widest encoded-input block (padded) */

enum { NEFII_ACT_RELU = 0, NEFII_ACT_ELU = 1, NEFII_ACT_SOFTPLUS100 = 2 };
enum { NEFII_HEAD_NONE = 0, NEFII_HEAD_TANH01 = 1, NEFII_HEAD_POW2 = 2, NEFII_HEAD_SIGMOID = 3,
       NEFII_HEAD_RELU = 4, NEFII_HEAD_ABS = 5, NEFII_HEAD_RELU_INIT = 6 };
enum { NEFII_E_ARG = -1, NEFII_E_SHAPE = -2, NEFII_E_UNSUPPORTED = -3 };

/* One linear layer of a fused MLP.  Inputs of a layer are the concatenation
 *   [ hidden/feature block (k_x wide, lives in the LDS "X" buffer) | encoded-input block (k_e wide, LDS "E" buffer) ]
 * which covers the reference's three concat patterns:
 *   ImplicitNetwork skip layer   cat[x, PE(p)]/sqrt(2)         implicit_differentiable_renderer.py:97-98
 *   RenderingNetwork layer 0     cat[PE(p), PE(v), n, feat]    implicit_differentiable_renderer.py:205
 *   EnvmapMaterialNetwork lyr 0  cat[PE(p), feat]              sg_envmap_material.py:362-365
 * All widths are padded to multiples of 32 (pad weights are zero).  w_fwd / w_bwd are in the MFMA
 * fragment order produced by nefii_pack_linear(). */
typedef struct nefii_layer {
    int32_t k_x, k_e;        /* padded input widths (multiples of 32; either may be 0) */
    int32_t n_out;           /* true output width */
    int32_t n_pad;           /* padded output width (multiple of 32, <= 512) */
    const float *w_fwd;      /* packed [ (k_x+k_e)/8 ][ n_pad/32 ][64][4] */
    const float *w_bwd;      /* packed transpose for input gradients, or NULL */
    const float *bias;       /* [n_pad] */
    const void *w_f16x3;     /* optional: fp16 hi/lo split of w_fwd * 64 in 32x32x16 fragment order
                                [ (k_x+k_e)/16 ][ n_pad/32 ][ hi | lo ][64 lanes][8 halves]  (nefii_pack_linear_f16x3) */
    const void *w_bwd_f16x3; /* optional: the transposed fragments of the same split for input-gradient GEMMs
                                [ n_pad/16 ][ (k_x+k_e)/32 ][ hi | lo ][64 lanes][8 halves]  (nefii_pack_linear_f16x3_bwd) */
} nefii_layer;

typedef struct nefii_mlp {
    int32_t n_layers;
    int32_t act;             /* NEFII_ACT_* applied after every layer but the last */
    int32_t head;            /* NEFII_HEAD_* applied to the last layer's output */
    int32_t enc_freqs[3];    /* positional-encoding octaves of raw inputs a,b,c; -1 = input absent
                                (embedder.py:38-50: x, sin(2^k x), cos(2^k x), k < L) */
    int32_t feat_width;      /* per-point feature vector loaded into X before layer 0 (0 = none) */
    int32_t reserved;        /* layout of w_stream / matrix instruction of the pipelined evaluator: 0 = 32x32x16 fragments,
                                1 = 16x16x32 fragments (8 % faster per tile on MI355X; what nefii_amd.ops packs) */
    const void *w_stream;    /* optional (SDF nets, split precision): the hidden layers' w_f16x3 fragments re-packed as one
                                stream per wave for the pipelined tile evaluator (nefii_pack_sdf_stream), or NULL */
    nefii_layer layer[NEFII_MAX_LAYERS];
} nefii_mlp;

int nefii_abi_version(void);

/* Padded width of a hidden/output block of v columns: multiples of 32 up to 32, multiples of 64 beyond (every
 * 16-deep k-step count of the split-precision kernels is then a multiple of 4).  Encoded-input blocks pad to 32. */
int nefii_padded_width(int v);

/* Re-order one layer's effective weight W [n_out][k_in] (PyTorch nn.Linear layout, after weight-norm)
 * into MFMA fragment order, multiplying by `scale` (1/sqrt(2) for skip layers).
 * Input columns [x_src0, x_src0+x_len) feed the X block, [e_src0, e_src0+e_len) the E block
 * (X block and n_out padded with zeros to nefii_padded_width, E block to a multiple of 32).  Writes w_fwd ( (kx+ke)*n_pad floats ),
 * optionally w_bwd ( n_pad*(kx+ke) floats ) and bias_pad ( n_pad floats, from bias or zeros ). */
int nefii_pack_linear(const float *W, const float *bias, int n_out, int k_in,
                      int x_src0, int x_len, int e_src0, int e_len, float scale,
                      float *w_fwd, float *w_bwd, float *bias_pad, void *stream);

/* fp16 hi/lo split of the same weights for the 3-MFMA split-precision path (x*w ~ xh*wh + xh*wl + xl*wh, fp32
 * accumulate; the dropped xl*wl term is ~2^-22 relative).  Weights are pre-multiplied by 64 (exact) so that
 * the lo halves of typical |w| ~ 0.05 stay normal fp16 numbers; the kernel scales accumulators by 1/64. */
int nefii_pack_linear_f16x3(const float *W, int n_out, int k_in, int x_src0, int x_len, int e_src0, int e_len,
                            float scale, void *w_f16x3, void *stream);

/* The packings above (nefii_pack_linear, and - for the layers whose w_f16x3 / w_bwd_f16x3 pointers are set -
 * nefii_pack_linear_f16x3 / _f16x3_bwd) for EVERY layer of h_mlp in ONE launch: the radiance / material weights train
 * (idr_train.py:771-775 steps both optimizers every iteration), so they are re-packed every step.  h_layers: host array
 * [n_layers] of the sources; the destinations are the device pointers in h_mlp->layer[l] (bias always; w_fwd / w_bwd
 * unless skip_f32, which the fp16-MFMA kernels never read).  (ABI v7) */
typedef struct nefii_pack_source {
    const float *W;          /* [n_out][k_in], PyTorch nn.Linear layout, after weight-norm */
    const float *bias;       /* [n_out] or NULL */
    int32_t n_out, k_in, x_src0, x_len, e_src0, e_len;     /* as nefii_pack_linear */
    float scale;
    int32_t skip_f32;        /* 1: leave w_fwd / w_bwd alone */
} nefii_pack_source;
int nefii_pack_mlp(const nefii_mlp *h_mlp, const nefii_pack_source *h_layers, void *stream);

/* Fused MLP forward over n points (replaces ImplicitNetwork.forward :85-108, RenderingNetwork.forward
 * :196-241 and EnvmapMaterialNetwork's diffuse_albedo_layers sg_envmap_material.py:369).
 *   in_a/in_b/in_c : raw [n,3] inputs to be positional-encoded (NULL when enc_freqs[i] < 0)
 *   feat           : [n, feat_width] or NULL
 *   out            : [n, out_stride] receives the last layer's n_out columns (head applied)
 *   hidden_out     : optional [n, hid_stride]: activation entering the last layer (use_last_as_f feature)
 *   stash          : optional training stash [n_layers][n][stash_stride]: slot l < L-1 holds the
 *                    post-activation output of layer l (= input of layer l+1), slot L-1 the last layer's
 *                    pre-head output; consumed by nefii_mlp_backward and the weight-gradient GEMMs.
 *                    stash_stride >= widest n_pad. */
int nefii_mlp_forward(const nefii_mlp *h_mlp, const float *in_a, const float *in_b, const float *in_c,
                      const float *feat, int64_t n, float *out, int out_stride,
                      float *hidden_out, int hid_stride, float *stash, int stash_stride, void *stream);

/* Backward of nefii_mlp_forward wrt the hidden activations (the MLP inputs are not differentiated: geometry
 * is frozen in Step-2).  d_out [n, out_stride] is dL/d(head output).  Writes dz [n_layers][n][dz_stride]:
 * the gradient wrt every layer's pre-activation; parameter gradients follow with nefii_mlp_wgrad per layer. */
int nefii_mlp_backward(const nefii_mlp *h_mlp, const float *d_out, int out_stride, const float *stash,
                       int stash_stride, int64_t n, float *dz, int dz_stride, void *stream);

/* Parameter gradients of one layer from the buffers of nefii_mlp_forward (stash) / nefii_mlp_backward (dz):
 * dW [n_out][k_in] = scale * dz[:, :n_out]^T x[:, :k_in] (PyTorch nn.Linear layout), db [n_out] = column sums of dz
 * (db may be NULL).  x = nefii_encode_inputs output for layer 0, stash slot l-1 for layer l.  Overwrites dW / db. */
int nefii_mlp_wgrad(const float *dz, int dz_stride, const float *x, int x_stride, int64_t n, int n_out, int k_in,
                    float scale, float *dW, float *db, void *stream);

/* fp16-MFMA variants of the three entry points above for the radiance and material MLPs (the north star's "fused MLP
 * forward/backward with MFMA fp16 tiles"; every layer must carry w_f16x3 / w_bwd_f16x3); stash, dz, dW and db stay
 * fp32; not for the SDF network.
 *   forward: single_pass = 0 - split precision (fp16 hi/lo operand pairs, 3 MFMAs per k-step, fp32-class accuracy: the
 *     default, its outputs are held to the north-star tolerance); 1 - operands rounded to fp16 once (measured 1.3e-3
 *     relative L2 on rendered RGB on config 3 - above the 1e-3 bar, kept for measurement only);
 *   backward / wgrad: one fp16 pass, fp32 accumulation.  Gradients sit far below fp16's normal range, so the GEMMs
 *     carry dz x *scale, a power of two that nefii_mlp_grad_scale derives on the device from max |d_out|; `scale` is a
 *     device pointer to one float.  Parameter gradients come out within ~1e-3 (2e-2 through weight-norm's projection). */
int nefii_mlp_forward_f16(const nefii_mlp *h_mlp, const float *in_a, const float *in_b, const float *in_c,
                          const float *feat, int64_t n, float *out, int out_stride,
                          float *hidden_out, int hid_stride, float *stash, int stash_stride, int single_pass,
                          void *stream);
int nefii_mlp_grad_scale(const float *d_out, int64_t count, float *scale, void *stream);
int nefii_mlp_backward_f16(const nefii_mlp *h_mlp, const float *d_out, int out_stride, const float *stash,
                           int stash_stride, int64_t n, float *dz, int dz_stride, const float *scale, void *stream);
int nefii_mlp_wgrad_f16(const float *dz, int dz_stride, const float *x, int x_stride, int64_t n, int n_out, int k_in,
                        float scale, const float *gscale, float *dW, float *db, void *stream);

/* ABI 10 - the same three calls with the training state in HALVES, for nets nefii_mlp_h16_supported() accepts (streamed
 * forward and backward kernels: 512-wide hidden layers, no skip layer, a head of <= 8 outputs - the reference's material,
 * radiance and indirect-radiance networks, sg_envmap_material.py:131-176, implicit_differentiable_renderer.py:126-193).
 * What autograd keeps between RenderingNetwork.forward and its backward in the reference (fp32 activations per layer) is
 * here: stash16 = [n_layers - 1][n][stash_stride] halves holding 16 * h_l (the forward's own fp16 operand image),
 * z_last = [n][8] floats (pre-activations of the head), dz16 = [n_layers][n][dz_stride] halves holding scale[0] * dz_l.
 * nefii_mlp_wgrad_f16h consumes one layer's dz16 with x = its input: fp32 rows (x_half = 0: nefii_encode_inputs' matrix,
 * layer 0) or the stash16 slice of the layer below (x_half = 1).  Weight gradients are bit-identical to the fp32-stash
 * calls (the GEMM rounded its operands to these very halves); the backward's act'(h) is taken from the fp16 h.
 * x0_16 (optional) = [n][nefii_mlp_x0_width()] halves: 16 * layer 0's input as the kernel lays it out - its padded feature
 * columns, then the encodings of a, b, c, zero-padded - so that layer 0's weight gradient needs no nefii_encode_inputs
 * matrix: nefii_mlp_wgrad_f16h(dz16 of layer 0, x0_16, x_half = 1, k_in = that width) yields dW in THAT column order
 * (columns [0, x_len) = the Linear's feature columns, [k_x, k_x + e_len) = its encoding columns).
 * Strides count elements of the array's own type. */
int nefii_mlp_h16_supported(const nefii_mlp *h_mlp);
int nefii_mlp_x0_width(const nefii_mlp *h_mlp);
int nefii_mlp_forward_f16h(const nefii_mlp *h_mlp, const float *in_a, const float *in_b, const float *in_c,
                           const float *feat, int64_t n, float *out, int out_stride, float *hidden_out, int hid_stride,
                           void *stash16, int stash_stride, float *z_last, void *x0_16, void *stream);
int nefii_mlp_backward_f16h(const nefii_mlp *h_mlp, const float *d_out, int out_stride, const void *stash16,
                            int stash_stride, const float *z_last, int64_t n, void *dz16, int dz_stride,
                            const float *scale, void *stream);
int nefii_mlp_wgrad_f16h(const void *dz16, int dz_stride, const void *x, int x_stride, int x_half, int64_t n, int n_out,
                         int k_in, float scale, const float *gscale, float *dW, float *db, void *stream);

/* ABI 12 - nefii_mlp_wgrad_f16h for EVERY layer of a net in (at most) one zero-fill and one launch per kernel form: what
 * autograd does layer by layer behind RenderingNetwork.forward / diffuse_albedo_layers (one mm + one sum per nn.Linear).
 * Items are independent; results equal the per-layer calls (the same kernels' bodies on a flat grid). */
#define NEFII_MAX_WGRAD_ITEMS 12
typedef struct nefii_wgrad_item {
    const void *dz16;        /* [n][dz_stride] halves: scale[0] * dz of this layer (nefii_mlp_backward_f16h) */
    const void *x;           /* the layer's input rows: halves holding 16 h (x_half = 1) or floats (x_half = 0) */
    float *dW, *db;          /* [n_out][k_in]; [n_out] or NULL */
    int32_t dz_stride, x_stride, x_half, n_out, k_in;
    float scale;
} nefii_wgrad_item;
int nefii_mlp_wgrad_f16h_batch(const nefii_wgrad_item *h_items, int n_items, int64_t n, const float *gscale, void *stream);

/* The reference's layer-0 concatenation [PE(a) | PE(b) | PE(c) | feat] as a dense [n, width] matrix. */
int nefii_encode_inputs(const nefii_mlp *h_mlp, const float *in_a, const float *in_b, const float *in_c,
                        const float *feat, int64_t n, float *out, int width, void *stream);

/* SDF value, optional last-hidden feature, and d sdf / d x in one pass
 * (replaces implicit_network(points) + ImplicitNetwork.gradient, implicit_differentiable_renderer.py:110-123,
 * 533-540; the reference runs three SDF passes on the same points).  `ws`: nefii_sdf_value_grad_workspace_bytes(h_mlp, n)
 * bytes - n*(n_layers-1)*512 floats for the generic 32-row kernels; one 128 KiB slot per hidden layer and workgroup
 * (at most 256 workgroups) for 512-wide softplus nets with a one-column last layer, a fragment stream (w_stream) and
 * transposed fragments on every layer, which run forward AND backward on the stream (64-row tiles, the tracer's
 * split-precision evaluator followed by the same k-loop over the transposed layers). */
int nefii_sdf_value_grad(const nefii_mlp *h_mlp, const float *x, int64_t n, float *sdf_out, int out_stride,
                         float *feat_out, int feat_stride, float *grad_out, float *ws, void *stream);
size_t nefii_sdf_value_grad_workspace_bytes(const nefii_mlp *h_mlp, int64_t n);

/* RayTracing.forward (ray_tracing.py:29-101) for rays with per-ray origins: bounding-sphere intersection
 * (rend_util.py:200-221), both-ends sphere tracing with back-off (:104-193), 100-sample bracket search +
 * bisection for rays that did not converge (:195-280) and, when `training`, the min-SDF search for rays
 * that miss (:309-337).  Bisection stops per ray (documented difference, <=1e-6 in t). */
typedef struct nefii_tracer_params {
    float object_bounding_sphere, sdf_threshold, line_search_step;
    int32_t line_step_iters, sphere_tracing_iters, n_steps, n_rootfind_steps;
    int32_t training;
    int32_t bisect_levels;   /* bisection steps resolved per round by evaluating the next levels of the bisection tree
                                speculatively (2^levels - 1 queries per ray per round, bit-identical result):
                                1..5, 0 = default 3.  5 suits small latency-bound batches, 3 large ones. */
    int32_t precision;       /* SDF evaluation inside the tracer: 0 = f32-input MFMA (exact fp32),
                                1 = 3x fp16 split MFMA, 32-query tiles; 2 = the same arithmetic on 64-query tiles
                                (8 waves, weight fragments shared by two row tiles); 1 and 2 need w_f16x3 */
    float coarse_tau;        /* > 0 (precision 2, pipelined shapes, 16x16x32 stream layout): the n_steps samples of the
                                bracket search (:203-219) and of the min-SDF search (:316-331) are first evaluated in ONE
                                fp16 pass (a third of the matrix work); coarse_tau bounds |coarse - split| of a sample.
                                Only samples whose coarse value cannot decide - within coarse_tau of zero where the
                                first sign change is looked for, within 2 coarse_tau of the minimum where the argmin
                                is - are re-evaluated in split precision, so every decision (and the outputs) is the
                                split evaluator's PROVIDED the bound holds for every sample taken: it is the CALLER'S
                                claim about this net, measured (nefii_sdf_eval_coarse against nefii_sdf_eval, with a
                                safety factor), not proven - a sample whose true difference exceeded it could decide
                                differently.  0 = off. */
    int32_t coarse_cap;      /* most samples of one ray re-evaluated individually; a ray with more takes all n_steps
                                in split precision instead.  <= 0: 64.  At most 100. */
    int32_t minsdf_group;    /* > 0: minsdf_steps holds one row of n_steps uniform draws per minsdf_group consecutive rays
                                (several batches - each with the draw the reference makes per call, ray_tracing.py:316 -
                                traced as ONE call); 0: one row for all rays */
    int32_t small_round;     /* rounds of at most this many split-precision queries run on 32-query tiles (more CUs, shorter
                                round; 1.5x the chip time per query); 0: 8192.  Traces that overlap other work pass a
                                smaller value. */
    int32_t trace_tier;      /* ABI 12 - tiered sphere tracing (needs coarse_tau > 0; ignored without it).  1: a sphere-tracing
                                evaluation (ray_tracing.py:128-134,164-168,182-183) whose front is still far from the surface
                                - the step that led to the query exceeds tier_gate * coarse_tau; the first evaluation on the
                                bounding sphere always - runs on the single-pass evaluator, and its value v16 is TAKEN AS THE
                                VALUE when |v16| > max(tier_kappa * coarse_tau, coarse_tau + 2 sdf_threshold): both decisions
                                the recurrence makes with it (v <= sdf_threshold :140-148, v < 0 :170-171,186-187) are then
                                the split evaluator's, but the front advances by v16 instead of v.  Inside that band the
                                query is repeated in split precision first.  UNLIKE the coarse pass of the two dense searches
                                this changes values, not only schedules: fronts differ by up to coarse_tau per step on the
                                way and end where the exact value is <= sdf_threshold as before (depths within ~sdf_threshold
                                / cos of the split-precision trace; measured: DESIGN.md, "tiered sphere tracing").  0: off. */
    float tier_kappa;        /* <= 0: 2 */
    float tier_gate;         /* <= 0: 4 */
    float minsdf_lipschitz;  /* ABI 13 - staged min-SDF search (needs coarse_tau > 0, 16 <= n_steps <= 128; ignored otherwise).
                                > 0: the caller's bound L on |sdf(p) - sdf(q)| / |p - q| along a ray inside the bounding sphere
                                (rays have unit directions).  The search (ray_tracing.py:309-337: argmin over n_steps depths)
                                then evaluates ceil(n_steps / 4) of the depths - evenly spread over their SORTED order, both
                                ends included - in the single-pass evaluator first; a depth s between evaluated neighbours
                                a < s < b whose lower bound  max(v_a - L (t_s - t_a), v_b - L (t_b - t_s)) - coarse_tau  exceeds
                                (lowest value seen) + coarse_tau cannot be the argmin and is NEVER evaluated; the others go to
                                the single-pass evaluator one by one, and the refinement in split precision proceeds as
                                before over the depths that were evaluated.  The argmin (first index of the exact minimum) is
                                the full search's PROVIDED L holds: like coarse_tau a measured claim about this net (largest
                                |grad sdf| seen on a sample of the ball, with a safety factor), not a proof.  Audited online:
                                every depth of the second stage - plus, per search, ONE of the skipped depths picked by a
                                hash and evaluated after all - is checked against the lower bound it was given,
                                counters[r][12].  The BRACKET search (:195-257) is staged the same way for eval-mode traces and
                                for rays outside the object mask: a quarter row of its n_steps samples spread over the row
                                first; a sample whose lower bound proves it positive - and, unless the ray lies inside the
                                mask and surely has a negative sample in front of which only signs matter, above the lowest
                                value seen + coarse_tau - is never evaluated; the first negative sample, the bracket and the
                                argmin fallback are the full row's.  0: off (all n_steps depths / samples are evaluated). */
    int32_t unread_misses;   /* ABI 14 - 1: the caller reads nothing of the rays that end WITHOUT a hit (out_points / out_dists of
                                such rays are then unspecified, out_hit is exact).  Honoured for eval-mode traces (training == 0):
                                the bracket search's argmin fallback (:221-231: a ray without a negative sample takes the depth
                                of its lowest sample) is not computed - no refinement of the samples near the minimum, and the
                                staged search skips every sample its bound proves positive.  Hit masks, hit points and hit
                                depths are those of unread_misses == 0 bit for bit.  What the Monte-Carlo renderer's secondary
                                rays need (path_tracing_render.py: visibility and the radiance at secondary HITS). */
    int32_t split_fp8;       /* ABI 15 - 1: the split-precision evaluations of the trace (every query the coarse pass and the tier do
                                not take) run on the "16f" evaluator: the main product x_h w_h on fp16 MFMAs as before, the two
                                correction products x_h w_l and x_l w_h on the block-scaled fp8 MFMA
                                (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands, constant scales).  A THIRD arithmetic: max |sdf
                                error| against fp64 ~1e-5 (1.6e-6 near the surface) where the fp16 split gives 5e-7 - values, not
                                only schedules, change at that level (depths of converged rays within ~1e-5; DESIGN.md section
                                4g).  Needs precision 2 and a 512-wide net with the fifth stream copy
                                (nefii_sdf_fp8corr_supported); ignored otherwise.  0: off (default). */
} nefii_tracer_params;
#define NEFII_TRACE_COUNTERS 14  /* int32 counters per round, see nefii_trace_rays */

/* The pipelined evaluator behind nefii_trace_rays (precision 2) and nefii_sdf_eval reads the hidden layers' fragments as
 * ONE stream per wave, 4 KiB per 16-deep unit of the layer sequence: [8 waves][units][4 fragments][64 lanes][8 halves];
 * the fragments are 32x32x16 ones (hi/lo of the wave's two column tiles) or, with nefii_mlp.reserved == 1, 16x16x32 ones
 * (hi/lo of two of the wave's four 16-feature tiles, alternating between the halves of a 32-deep k-step).
 * With reserved == 1 a third copy follows for the single-pass (coarse) evaluator: hi fragments only, one 32-deep k-step
 * of the wave's feature tiles per unit, K zero-padded to multiples of 128.
 * When every layer also carries transposed fragments (w_bwd_f16x3) and the last layer has one column, a fourth copy
 * follows for nefii_sdf_value_grad: per wave the forward units again, then for l = n_layers-2 .. 1 the 16-deep units of
 * the transposed layer (the wave's 64 hidden inputs as output features) - one cursor runs forward and backward.
 * nefii_sdf_stream_bytes: size of that buffer, 0 if the net's shape does not qualify (every hidden layer 512 wide,
 * k_x in {0,512}, k_e in {0,64}, 512-deep last layer) - such nets run on the generic kernel and leave w_stream NULL.
 * nefii_pack_sdf_stream: device-side copy from the layers' w_f16x3 (call after nefii_pack_linear_f16x3). */
size_t nefii_sdf_stream_bytes(const nefii_mlp *h_sdf);
int nefii_pack_sdf_stream(const nefii_mlp *h_sdf, void *w_stream, void *stream);

/* The same idea for the radiance / material MLPs (RenderingNetwork.forward, implicit_differentiable_renderer.py:196-241;
 * EnvmapMaterialNetwork's MLPs, sg_envmap_material.py:357-425) on the split-precision forward nefii_mlp_forward_f16: nets
 * whose hidden layers are all 512 wide (layer 0: up to 512 feature columns + up to 128 encoding columns; last layer: at
 * most 8 outputs) keep their hidden layers' hi/lo fragments as one stream per wave, [8 waves][units][4 fragments][64
 * lanes][8 halves], a unit = 16-deep half step of a layer whose K is rounded up to a multiple of 64 with zero weights.
 * With w_stream set the forward runs 48- / 64-row tiles on the tracer's pipelined evaluator structure.
 * nefii_mlp_stream_bytes: size of the buffer, 0 when the shape does not qualify; nefii_pack_mlp_stream: device-side
 * gather from the layers' w_f16x3 - call after every nefii_pack_linear_f16x3 (the weights train). */
size_t nefii_mlp_stream_bytes(const nefii_mlp *h_mlp);
int nefii_pack_mlp_stream(const nefii_mlp *h_mlp, void *w_stream, void *stream);

/* Transposed counterpart for the split-precision input-gradient GEMMs (nefii_sdf_value_grad picks its split-precision
 * kernel when every layer carries both w_f16x3 and w_bwd_f16x3). */
int nefii_pack_linear_f16x3_bwd(const float *W, int n_out, int k_in, int x_src0, int x_len, int e_src0, int e_len,
                                float scale, void *w_bwd_f16x3, void *stream);

/* sdf_out[i] = implicit_network(x[i])[:, 0] (implicit_differentiable_renderer.py:85-108) with the tracer's
 * split-precision tile evaluator (the arithmetic of precision 2): the bulk SDF query the reference issues from
 * ray_tracing.py:128-131,211-216,322-327 and pixel_pair_generator.py:52, as a stand-alone call.  x [n][3], needs
 * w_f16x3 in every layer (and uses w_stream when set); bias arrays hold n_pad floats, 16-byte aligned. */
int nefii_sdf_eval(const nefii_mlp *h_sdf, const float *x, int64_t n, float *sdf_out, void *stream);
/* The same points through the tracer's single-pass (coarse) evaluator; NEFII_E_UNSUPPORTED when the net has no
 * single-pass stream (nefii_sdf_coarse_supported).  max |nefii_sdf_eval_coarse - nefii_sdf_eval| over points of the
 * bounding sphere, with a safety factor, is what a caller passes as nefii_tracer_params.coarse_tau. */
int nefii_sdf_eval_coarse(const nefii_mlp *h_sdf, const float *x, int64_t n, float *sdf_out, void *stream);
int nefii_sdf_coarse_supported(const nefii_mlp *h_sdf);
/* ABI 15 - the same points through the "16f" evaluator (nefii_tracer_params.split_fp8: correction products on block-scaled fp8);
 * NEFII_E_UNSUPPORTED unless nefii_sdf_fp8corr_supported (512-wide hidden layers, reserved == 1, w_stream packed by this
 * library version: the fifth copy of nefii_pack_sdf_stream - per wave and 128-deep chunk of each layer's 128-padded K eight 4-KiB
 * units: four 32-deep k-steps of hi fragments, then per 16-feature tile [e4m3(w_l 2^10) 32 B | e4m3(w_h 2^-1) 32 B] per lane). */
int nefii_sdf_eval_fp8corr(const nefii_mlp *h_sdf, const float *x, int64_t n, float *sdf_out, void *stream);
int nefii_sdf_fp8corr_supported(const nefii_mlp *h_sdf);

size_t nefii_trace_workspace_bytes(int64_t n_rays, const nefii_tracer_params *h_params);
int nefii_trace_max_rounds(const nefii_tracer_params *h_params);
/* lin_steps: the n_steps values of torch.linspace(0,1,n_steps); minsdf_steps: the n_steps uniforms of
 * minimal_sdf_points (only read when training).  counters (optional, int32 [max_rounds][NEFII_TRACE_COUNTERS])
 * receives per round: [r][0] single queries, [r][1] rays with n_steps dense queries in split precision, [r][2] rays in
 * bisection (2^levels - 1 speculative queries each), [r][3] bisection evaluations actually consumed, [r][4] coarse-pass
 * samples re-evaluated in split precision, [r][5] QUARTER rows (ceil(n_steps / 4) samples of one ray) in the single-pass (coarse) evaluator (ABI 11; whole rows before),
 * [r][6] rays entering a dense search (the reference evaluates n_steps samples for each), [r][7] = (2^levels - 1)*[2],
 * the speculative bisection evaluations executed, [r][8] (the bits of a float >= 0) the largest |coarse - split| among the
 * coarse-pass samples this round re-evaluated in split precision: the online audit of coarse_tau - every refined sample
 * is evaluated both ways anyway; a value above coarse_tau means the caller's bound does not hold for this net;
 * [r][9] (ABI 12, trace_tier) sphere-tracing queries in the single-pass evaluator, [r][10] single queries of [r][0] that
 * repeat such a query in split precision (they take part in the audit of [r][8]);
 * [r][11] (ABI 13, minsdf_lipschitz) depths of staged min-SDF searches evaluated one by one in the single-pass evaluator (the
 * first stage's depths are a quarter row of [r][5]); [r][12] (the bits of a float >= 0) the largest amount by which such a
 * depth's value fell below the lower bound that minsdf_lipschitz gave it: above 0 the bound does not hold for this net;
 * [r][13] (ABI 15) samples the staged searches had SKIPPED and evaluated after all as probes of that audit: one per search picked
 * by a hash of (ray, position), plus up to 6 whose bound cleared the limit by less than 2 coarse_tau (they are among [r][11]).
 * Algorithmic evaluations (what the reference's recurrences need) = [0] + [9] - [10] + n_steps*[6] + [3]; executed in split
 * precision = [0] + n_steps*[1] + [7] + [4]; executed in the coarse evaluator = ceil(n_steps / 4)*[5] + [9] + [11]. */
int nefii_trace_rays(const nefii_mlp *h_sdf, const nefii_tracer_params *h_params,
                     const float *origins, const float *dirs, const uint8_t *object_mask, int64_t n_rays,
                     const float *lin_steps, const float *minsdf_steps,
                     float *out_points, uint8_t *out_hit, float *out_dists,
                     void *workspace, size_t workspace_bytes, int32_t *counters, void *stream);

/* The same, restricted to rounds [round_begin, round_end) (round_end <= 0: up to nefii_trace_max_rounds).  Rounds
 * after the last one that emitted a query are empty launches; a caller that synchronises anyway can run a prefix,
 * read counters[round_end-1][0..2] and continue with round_begin = round_end only if any of them is non-zero
 * (ray state and counters persist in `workspace` between the calls; outputs are complete once none is pending). */
int nefii_trace_rays_rounds(const nefii_mlp *h_sdf, const nefii_tracer_params *h_params,
                            const float *origins, const float *dirs, const uint8_t *object_mask, int64_t n_rays,
                            const float *lin_steps, const float *minsdf_steps,
                            float *out_points, uint8_t *out_hit, float *out_dists,
                            void *workspace, size_t workspace_bytes, int32_t *counters,
                            int round_begin, int round_end, void *stream);
/* The same for n_groups contiguous chunks of the ray batch (rays [group_begin[g], group_begin[g+1]) ), each with its own
 * workspace and hipStream_t, enqueued round-major so that all chunks advance together.  Rays are independent: results
 * are those of one nefii_trace_rays_rounds call over the whole batch.  What it buys: the latency-bound rounds of a small
 * batch (fewer 64-query tiles than CUs - one tile time per round, however few the queries) of different chunks overlap
 * on the chip.  counters: [n_groups][max_rounds][4].  The caller orders the streams against its own (events). */
int nefii_trace_rays_groups(const nefii_mlp *h_sdf, const nefii_tracer_params *h_params,
                            const float *origins, const float *dirs, const uint8_t *object_mask,
                            int n_groups, const int64_t *group_begin,
                            const float *lin_steps, const float *minsdf_steps,
                            float *out_points, uint8_t *out_hit, float *out_dists,
                            void *const *workspaces, const size_t *workspace_bytes, int32_t *counters,
                            int round_begin, int round_end, void *const *streams);

/* Measurement hooks (bench.py): when enabled, nefii_trace_rays brackets every SDF-evaluation launch with HIP
 * events on the launch stream; nefii_trace_profile_read returns the summed launch durations (ms), the number
 * of launches and the span of the last trace call, and clears the record.  Off by default. */
int nefii_trace_profile_enable(int on);
int nefii_trace_profile_read(double *h_eval_ms, int *h_n_eval, double *h_span_ms);
/* per-launch durations (ms) in launch order into h_ms[cap]; returns the count (does not clear the record) */
int nefii_trace_profile_launches(float *h_ms, int cap);

/* rend_util.get_camera_params + lift (rend_util.py:90-142): uv [B,S,2], pose [B,4,4], intrinsics [B,4,4]
 * -> unit ray dirs [B,S,3] and per-ray origins [B,S,3] (camera centre broadcast). */
int nefii_camera_rays(const float *uv, const float *pose, const float *intrinsics, int batch, int64_t samples,
                      float *out_dirs, float *out_origins, void *stream);

/* render_with_sg (sg_render.py:164-295) for one base material (K=1) with global roughness [1] and
 * specular [3]: outputs rgb / specular / diffuse [n,3]. */
int nefii_sg_render_forward(const float *lgtSGs, int n_lobes, const float *specular, const float *roughness,
                            const float *albedo, const float *normal, const float *view, int64_t n,
                            float *rgb, float *spec_rgb, float *diff_rgb, void *stream);
/* Gradients of sum(d_rgb*rgb + d_spec*spec_rgb + d_diff*diff_rgb) wrt albedo [n,3], roughness [1], specular [3]
 * and lgtSGs [n_lobes,7] (the last three accumulated atomically into zero-initialised buffers). */
int nefii_sg_render_backward(const float *lgtSGs, int n_lobes, const float *specular, const float *roughness,
                             const float *albedo, const float *normal, const float *view, int64_t n,
                             const float *d_rgb, const float *d_spec, const float *d_diff,
                             float *g_albedo, float *g_roughness, float *g_specular, float *g_lgtSGs, void *stream);

/* IDRNetwork.get_background_rgb (implicit_differentiable_renderer.py:646-663): sum of light SGs along dirs. */
int nefii_env_radiance_forward(const float *lgtSGs, int n_lobes, const float *dirs, int64_t n, float eps,
                               float *rgb, void *stream);
int nefii_env_radiance_backward(const float *lgtSGs, int n_lobes, const float *dirs, int64_t n, float eps,
                                const float *d_rgb, float *g_lgtSGs, void *stream);

/* The three importance-sampled directions per surface point and their 3x3 pdf table for multiple importance
 * sampling (cos_sampling :128, brdf_sampling :61, mix_sg_sampling :168 and the pdf_fn_* of
 * path_tracing_render.py; table as :1312-1325).  uniforms [n,7] = (cos r1 r2 | ggx r1 r2 | mix r0 r1 r2) in the
 * reference's draw order; roughness [n].  Outputs: wi [3,n,3], own_pdf [3,n] (clamped at 1e-6),
 * pdf_table [3,n,3] (row = direction, column = pdf of strategy j for that direction). */
int nefii_mis_sample(const float *lgtSGs, int n_lobes, const float *roughness, const float *normal, const float *view,
                     const float *uniforms, int64_t n, float *wi, float *own_pdf, float *pdf_table, void *stream);

/* Per-point MC shading sum of pt_render_diff_shadow_indirect_mlp (diff_geo=False), path_tracing_render.py:1406-1476:
 * light [3,n,3] = sum of light SGs along wi (nefii_env_radiance_forward with eps 1e-6), visibility [3,n],
 * indirect [3,n,3] radiance at secondary hits; specular [3] global, roughness [n], albedo [n,3]. */
int nefii_mc_shade_forward(const float *specular, const float *roughness, const float *albedo, const float *normal,
                           const float *view, const float *wi, const float *own_pdf, const float *pdf_table,
                           const float *light, const float *visibility, const float *indirect, int64_t n,
                           float *rgb, float *spec_rgb, float *diff_rgb, void *stream);
/* Gradients wrt light [3,n,3], indirect [3,n,3], albedo [n,3], roughness [n] (overwritten) and, when non-NULL,
 * the global specular [3] (accumulated atomically into a zero-initialised buffer). */
int nefii_mc_shade_backward(const float *specular, const float *roughness, const float *albedo, const float *normal,
                            const float *view, const float *wi, const float *own_pdf, const float *pdf_table,
                            const float *light, const float *visibility, const float *indirect, int64_t n,
                            const float *d_rgb, const float *d_spec, const float *d_diff, float *g_light,
                            float *g_indirect, float *g_albedo, float *g_roughness, float *g_specular, void *stream);

/* IDRLoss.forward (code/model/loss.py:278-320) for the terms the shipped confs weight: masked L1/L2/SmoothL1 of idr_rgb
 * and sg_rgb against the ground truth over rays with network_object_mask & object_mask (:163-184), the mask BCE over
 * the others (:186-196), the normal-smoothness variance over 2r x 2r patches (:198-207) and the background colour term
 * over rays missing both masks; eikonal is zero under frozen geometry, SSIM / view-diff / roughness-smooth have weight 0
 * in every conf.  losses[0..5] = loss, idr_rgb_loss, sg_rgb_loss, mask_loss, normalsmooth_loss, background_rgb_loss;
 * d_idr_rgb / d_sg_rgb [n,3] (either may be NULL) receive d loss / d input in the same launch. */
typedef struct nefii_loss_params {
    float idr_rgb_weight, sg_rgb_weight, mask_weight, alpha, normalsmooth_weight, background_rgb_weight;
    int32_t loss_type;       /* 0 L1, 1 L2, 2 SmoothL1(beta 1) */
    int32_t env_loss_type;   /* 0 L1, 1 L2 */
    int32_t r_patch;         /* patches of (2 r)^2 consecutive rays for the normal-smoothness term; < 1: off */
    int32_t reserved;
} nefii_loss_params;
int nefii_idr_loss(const nefii_loss_params *h_params, const float *idr_rgb, const float *sg_rgb, const float *rgb_gt,
                   const uint8_t *network_object_mask, const uint8_t *object_mask, const float *sdf_output,
                   const float *normals, int64_t n, float *losses, float *d_idr_rgb, float *d_sg_rgb, void *stream);

/* Assembly of the per-ray output buffers of IDRNetwork.forward (implicit_differentiable_renderer.py:441-501): the reference
 * allocates eight [n_rays, C] buffers of ones / zeros and writes the shaded hit rays into them by boolean mask, one pair of
 * launches per buffer.  Here: all buffers in two launches.  Block b: dst [rows, cols] is filled with `fill`, then row
 * where[i] takes row i of src (src_row_stride floats apart: cols, or 0 for one row broadcast to every hit), i < n_src.
 * Several i may name the same row (the graph step pads its hit list with a scratch row): any of them wins.
 * nefii_gather_rows is the adjoint: dst [n_src, cols] <- src[where[i]] (src [rows, cols]: the gradient of an assembled
 * buffer; a block with a NULL src or dst is skipped). */
#define NEFII_MAX_ROW_BLOCKS 12
typedef struct nefii_row_block {
    const float *src;
    float *dst;
    int32_t cols;
    int32_t src_row_stride;
    float fill;
    int32_t reserved;
} nefii_row_block;
int nefii_assemble_rows(const nefii_row_block *h_blocks, int n_blocks, const int64_t *where, int64_t n_src, int64_t rows,
                        void *stream);
int nefii_gather_rows(const nefii_row_block *h_blocks, int n_blocks, const int64_t *where, int64_t n_src, int64_t rows,
                      void *stream);

/* ABI 12 - the inputs of get_rbg_value for the compacted hit rays in one launch (implicit_differentiable_renderer.py:358-364:
 * points[mask], -ray_dirs[mask]; :537-545: gradient / (norm + 1e-6), view / (norm + 1e-6)): row i of the outputs is taken from
 * row where[i] of points / ray_dirs / grad [rows, 3] (the SDF gradient at the traced points) and, when feat_cols > 0, of
 * feat_src [rows, feat_cols] -> pts_out, view_out (= -dir, normalised), nrm_out [n, 3], feat_out [n, feat_cols]. */
int nefii_prepare_hits(const float *points, const float *ray_dirs, const float *grad, const float *feat_src, int feat_cols,
                       const int64_t *where, int64_t n, int64_t rows, float *pts_out, float *view_out, float *nrm_out,
                       float *feat_out, void *stream);

/* ABI 12 - EnvmapMaterialNetwork.forward's head for GLOBAL roughness / specular parameters (physg.conf;
 * sg_envmap_material.py:381-414) in one launch each way: rough_out [1] = (1 - 0.089) sigmoid(rough_param[0]) + 0.089,
 * spec_out [3] = 0.16 sigmoid(spec_param)^2 (n_spec = 1: white specular, the value repeated; 3: per channel); fake_rough /
 * fake_spec (the warm-up flags, idr_train.py:705-713) put 0.5 in place of the sigmoid.  Backward: d_rough [1], d_spec [3]
 * (either may be NULL) -> g_rough_param [1], g_spec_param [n_spec] (overwritten). */
int nefii_material_head_global(const float *rough_param, const float *spec_param, int n_spec, int fake_rough, int fake_spec,
                               float *rough_out, float *spec_out, void *stream);
int nefii_material_head_global_backward(const float *rough_param, const float *spec_param, int n_spec, int fake_rough,
                                        int fake_spec, const float *d_rough, const float *d_spec, float *g_rough_param,
                                        float *g_spec_param, void *stream);

/* MEASUREMENT, not part of the reference's path: what the matrix cores of THIS device sustain on dense fp16 MFMAs with
 * random operands and nothing else in the instruction stream (v_mfma_f32_16x16x32_f16, four accumulator chains per wave,
 * one wave per SIMD, every CU): runs `groups` groups of 8 MFMAs per wave on 256 workgroups, synchronises, and returns the
 * elapsed milliseconds and the FLOPs executed.  MI355X is power-limited under such a loop (2.0 PFLOP/s on zero operands,
 * ~1.5 on random ones: profiles/r04/slot_probe.txt), so bench.py quotes roofline.sustained_peak beside the 2.5 PFLOP/s
 * spec peak the roofline fraction is priced against. */
int nefii_mfma_sustained_probe(int groups, float *h_ms, double *h_flops, void *stream);
/* The same loop on `chains` = 4 or 8 independent accumulator chains (a group is then 2 x chains MFMAs).  With 4 chains the
 * loop takes 148 cycles per 8 MFMAs against 128 for a free-running pipe (each accumulator is wanted again after 64 cycles of
 * issue); 8 chains leave every dependency 128 cycles - bench.py quotes both, so that the sustained figure is not an artefact
 * of the probe's own issue rate.  (ABI 12) */
int nefii_mfma_sustained_probe_chains(int groups, int chains, float *h_ms, double *h_flops, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NEFII_AMD_H */
