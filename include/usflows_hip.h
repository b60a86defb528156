/*
 * usflows_hip.h -- C ABI of libusflows_hip.so: the MI355X (gfx950) kernels behind the USFlows
 * coupling-flow hot path (Flow.log_prob / Flow.sample / Flow.backward / Flow._forward).
 *
 * Boundary contract:
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise; sizes are int64_t;
 *   - nothing is allocated inside; the caller owns every buffer for the duration of the call;
 *   - work is enqueued on the given hipStream_t (pass torch's current stream) and the call
 *     returns immediately; no host synchronisation, safe to capture into a hipGraph;
 *   - return value: 0 = ok, negative = argument error, positive = hipError_t of the launch;
 *     usf_last_error() gives a human-readable message for the calling thread. No exceptions
 *     cross the ABI.
 *
 * The reference (aai-institute/USFlows) has no native layer: each entry point below replaces a
 * sequence of ATen ops issued by the cited reference lines (paths relative to /root/reference).
 */
#ifndef USFLOWS_HIP_H
#define USFLOWS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* usf_stream_t; /* hipStream_t */

#define USF_ABI_VERSION 35

/* activation ids (conditioner nonlinearity, networks.py:717,737) */
#define USF_ACT_NONE 0
#define USF_ACT_LEAKY_RELU 1 /* slope 0 == ReLU */
#define USF_ACT_GATE 2       /* usf_linear_f32 only: `addend` is NOT added but read as a gate h [M,N]:
                                C = (A W^T + bias) * (h > 0 ? 1 : slope) -- the (Leaky)ReLU backward from the saved layer
                                output (usf_act_grad_f32) folded into the data-gradient GEMM's epilogue */

/* base distribution ids */
#define USF_BASE_LAPLACE 0 /* torch Laplace.log_prob summed over D (Independent, distributions.py:709-728) */
#define USF_BASE_NORMAL 1  /* torch Normal.log_prob summed over D */
#define USF_BASE_LPNORM1 2 /* r = ||z-loc||_1  (RadialDistribution.log_prob, distributions.py:501-505) */
#define USF_BASE_LPNORM2 3 /* r = ||z-loc||_2 */
#define USF_BASE_LPNORMINF 4 /* r = ||z-loc||_inf */
#define USF_BASE_ROWSUM 5  /* usf_base_logprob_f32 only: logp[m] = sum_d z[m,d] + logdet (z = the partial sums a fused
                              epilogue left, usf_gemm_planes_desc.base_part; loc / scale unused, D <= 8) */

/*
 * Fused dense layer:   C = epilogue( prologue(A) @ W^T )        (row-major, fp32, exact-f32 MFMA)
 *
 *   prologue(A)[m,k] = (A[m,k] / pre_div[k]) - pre_sub[k]       (each optional)
 *   acc[m,n]         = sum_k prologue(A)[m,k] * W[n,k]
 *   v                = act(acc + bias[n] + addend[m,n])         (bias, addend optional)
 *   v                = residual[m,n] + res_sign * v             (residual optional)
 *   C[m,n]           = v * post_mul[n]                          (post_mul optional)
 *
 * replaces, depending on the flags:
 *   BlockAffineTransform.backward  y=(y-b)@Minv^T        transforms.py:936-962 (pre_sub=b, W=Minv)
 *   ... preceded by ScaleTransform.backward x/scale      transforms.py:116-125 (pre_div=scale)
 *   BlockAffineTransform.forward   y=x@M^T+b             transforms.py:913-934 (bias=b, W=M)
 *   ... followed by ScaleTransform.forward x*scale       transforms.py:105-114 (post_mul=scale)
 *   one nn.Linear (+LeakyReLU) of the conditioner MLP    networks.py:739-751
 *   the masked residual of MaskedCoupling                transforms.py:285-290, 301-306
 *     (mask-aware: A/W/C are the pass-through / transformed column segments)
 * Requirements: K % 4 == 0, lda/ldw/ldc/ldr % 4 == 0, all base pointers 16-byte aligned.
 */
typedef struct usf_linear_desc {
  const float* A;        int64_t lda;   /* [M,K] activations */
  const float* W;        int64_t ldw;   /* [N,K] weight, K contiguous (torch nn.Linear layout) */
  const float* bias;                    /* [N] or NULL */
  const float* pre_div;                 /* [K] or NULL */
  const float* pre_sub;                 /* [K] or NULL */
  const float* residual; int64_t ldr;   /* [M,N] or NULL */
  const float* addend;   int64_t ldadd; /* [M,N] or NULL: added BEFORE the activation (context branch
                                           h + layers[1](context), networks.py:741-743) */
  const float* post_mul;                /* [N] or NULL */
  float*       C;        int64_t ldc;   /* [M,N] */
  int64_t M, N, K;
  float res_sign;                       /* +1 (coupling forward) / -1 (coupling backward) */
  float slope;                          /* LeakyReLU negative slope */
  int32_t act;                          /* USF_ACT_* */
  int32_t reserved;
  /* Optional split-precision copy of W for the bf16x3 path (fp32-equivalent accuracy on the bf16 matrix
   * cores): three bf16 planes W1+W2+W3 == W (round-to-nearest residual split), plane p at
   * W_split + p*split_plane_stride, each [N, ldw_split] bf16 with ldw_split >= ceil32(K), zero-padded.
   * Used when non-NULL and the op has no residual / addend and K % 8 == 0; W must still be given. */
  const void* W_split; int64_t ldw_split; int64_t split_plane_stride;
  /* Optional side output (ABI 32): the three row-major bf16 planes of the INPUT A (A == p1 + p2 + p3 exactly, the
   * split the bf16x3 kernel makes of its operand anyway), plane p at A_planes_out + p * planes_out_stride elements,
   * each [ceil32(M), ldp_out] bf16 with ldp_out >= ceil32(K), ldp_out % 8 == 0; rows [M, ceil32(M)) are NOT written
   * (the caller's buffer holds zeros there: usf_wgrad_planes_f32 sums over them), columns [K, ceil32(K)) receive finite
   * padding.  This is the operand layout of usf_wgrad_planes_f32: the
   * data-gradient / forward GEMM of a layer hands the weight gradient of the same layer its operand already split
   * (flows.py:196-203: loss.backward() through every F.linear of the flow).  Not with pre_div / pre_sub. */
  void* A_planes_out; int64_t ldp_out; int64_t planes_out_stride;
} usf_linear_desc;

int usf_linear_f32(const usf_linear_desc* d, usf_stream_t stream);

/*
 * Which kernel family / instantiation usf_linear_f32 would launch for this descriptor (nothing is launched):
 *   1000                          small-batch kernel (M <= 768)
 *   2000 + 100 TM + 10 TN + WM    exact-f32 MFMA tile
 *   3000 + 100 TN + 10 WM + NB    bf16x3 tile (TN x 32 columns, WM waves x 32 rows, NB weight buffers)
 * 0 for a descriptor with empty extents.  The parity tests use it to prove that every instantiation the BASELINE
 * configurations select is compared with the reference arithmetic (tests/test_configs_gpu.py).
 */
int usf_linear_variant(const usf_linear_desc* d);

/*
 * Fused additive coupling layer (MaskedCoupling.forward/backward, transforms.py:277-306) with a
 * dense (leaky-)ReLU conditioner (ConditionalDenseNN networks.py:739-751 / pyro DenseNN):
 *
 *   out[:, pass]  = z[:, pass]                                   (only if out != z)
 *   out[:, trans] = z[:, trans] + sign * MLP(z[:, pass] [, context])
 *
 * Mask-aware: W_in holds only the columns of layers[0].weight that multiply pass-through
 * features (the others meet x*mask == 0), W_out/b_out only the rows that produce transformed
 * features (the others are multiplied by (1-mask) == 0).  Hidden activations never leave the
 * CU (registers).  The fused kernel takes n_hidden in [1, 3] and hidden widths <= usf_coupling_max_width().
 *
 * Padding contract (lets the kernel load weights without clamps or selects): with
 * Hp = usf_coupling_padded_width(max hidden width), Kp = ceil32(n_pass), Np = ceil32(n_trans):
 *   W_in  readable as [Hp, Kp] (ldw_in  >= Kp), W_hid[i] as [Hp, Hp] (ldw_hid >= Hp),
 *   W_out readable as [Np, Hp] (ldw_out >= Hp), b_in / b_hid[i] / W_ctx / b_ctx as [Hp], b_out as [Np],
 * all ZERO outside their true extents.  z/out: in place (out == z, ldo == ldz).
 */
#define USF_MAX_HIDDEN 4
typedef struct usf_coupling_desc {
  const float* z;   int64_t ldz;        /* [M, >= off_pass+n_pass, off_trans+n_trans] */
  float*       out; int64_t ldo;        /* may alias z (in place on the transformed half) */
  int64_t M;
  int64_t off_pass, n_pass;             /* conditioning (mask==1) column segment */
  int64_t off_trans, n_trans;           /* transformed (mask==0) column segment */
  int32_t n_hidden;
  int32_t hidden[USF_MAX_HIDDEN];
  const float* W_in;   int64_t ldw_in;  /* [hidden[0], n_pass] */
  const float* b_in;                    /* [hidden[0]] */
  const float* W_hid[USF_MAX_HIDDEN];   /* W_hid[i]: [hidden[i+1], hidden[i]], i < n_hidden-1 */
  const float* b_hid[USF_MAX_HIDDEN];
  int64_t      ldw_hid[USF_MAX_HIDDEN];
  const float* W_out;  int64_t ldw_out; /* [n_trans, hidden[last]] */
  const float* b_out;                   /* [n_trans] */
  const float* context;                 /* [M] (context_dim == 1, flows.py:188-191,564) or NULL */
  const float* W_ctx;                   /* [hidden[0]] (layers[1].weight[:,0]) or NULL */
  const float* b_ctx;                   /* [hidden[0]] or NULL */
  const float* post_sub;                /* [n_trans]+[n_pass] reserved, must be NULL */
  float sign;                           /* +1 forward, -1 backward */
  float slope;
  int32_t act;
  int32_t reserved;
  /* Optional split-precision copies of the (padded) weights for the bf16x3 path: three bf16 planes each,
   * plane q at base + q*plane (elements), rows/ld as in the padding contract above with ld >= the padded K.
   * split_hid / split_out additionally have their K (hidden-unit) axis permuted inside every block of 32:
   * position 8g+j holds unit 4g+j (j<4) or 16+4g+(j-4) (j>=4), the order in which the kernel's accumulators
   * present the previous layer.  Used when all needed planes are given, hidden width in (128, 256], M >= 1024. */
  const void* split_in;  int64_t split_in_ld,  split_in_plane;
  const void* split_hid[USF_MAX_HIDDEN]; int64_t split_hid_ld, split_hid_plane;
  const void* split_out; int64_t split_out_ld, split_out_plane;
  /* Optional (ABI 32): hidden_out[l] [M, ld_hidden_out] receives the activations of hidden layer l (after the
   * nonlinearity; the padded width, zeros in the padding) -- what the backward pass of the training step otherwise computes a
   * second time (two GEMMs per coupling layer; flows.py:196-203).  Only the bf16x3 kernel stores them (its eligibility rule
   * above): a descriptor that sets hidden_out and is served by another kernel is rejected.  ld_hidden_out % 4 == 0,
   * 16-byte aligned bases. */
  float* hidden_out[USF_MAX_HIDDEN]; int64_t ld_hidden_out;
  /* act == USF_ACT_GATE (ABI 32; bf16x3 kernel only, no context): hidden layer l's pre-activation is not passed through
   * the nonlinearity but multiplied by (gate[l][m, j] > 0 ? 1 : slope), gate[l] [M, ld_gate] -- the conditioner's BACKWARD
   * pass on the same kernel: with the transposed weights (W_in = W_out^T, W_hid reversed and transposed, W_out = W_in^T),
   * zero biases, the column segments swapped (pass <-> trans) and gate[l] = the forward's saved activation of hidden layer
   * n_hidden - 1 - l, the launch computes  g[:, pass] += sign * d_h1 W_in  from g[:, trans]  and hidden_out receives the
   * gradients at the hidden activations (d_h of the last hidden layer first): what autograd derives for the MLP of
   * MaskedCoupling (transforms.py:277-306, networks.py:739-751) in three GEMM launches + gates. */
  const float* gate[USF_MAX_HIDDEN]; int64_t ld_gate;
} usf_coupling_desc;

int usf_coupling_additive_f32(const usf_coupling_desc* d, usf_stream_t stream);
/* which kernel usf_coupling_additive_f32 launches for this descriptor (nothing is launched, no pointer is dereferenced):
 * 3 the tiny-layer kernel (M <= 256, segments and hidden widths <= 64, the layer's images in 64 KB of LDS: usf_coupling_tiny.hip),
 * 2 the bf16x3 kernel, 1 the exact-f32 MFMA kernel, 0 for a NULL descriptor */
int usf_coupling_variant(const usf_coupling_desc* d);
int usf_coupling_max_width(void);       /* widest hidden layer the fused kernel accepts */
int usf_coupling_padded_width(int h);   /* Hp of the padding contract for hidden width h (-1 if unsupported) */

/*
 * Tail of Flow.log_prob (flows.py:245): per-sample reduction over the feature axis.
 *   LAPLACE/NORMAL: logp[m] = sum_d base_d(z[m,d]) + logdet_const (+ (float)*logdet_dev when logdet_dev != NULL: the
 *                   flow's parameter-only log-det as a DEVICE fp64 scalar, so that a caller whose parameters just changed
 *                   -- every optimiser step -- does not read it back to the host first)
 *   LPNORM*:        logp[m] = ||z[m,:] - loc||_p   (the caller finishes RadialDistribution.log_prob
 *                             on the [M] vector: distributions.py:506-511)
 * loc/scale: [D] (scale unused for LPNORM*). If sum_out != NULL, sum_out[0] += sum_m logp[m] and
 * sum_out[1] += M (fp64 accumulators, for the data-parallel mean: one RCCL all-reduce of 2 scalars).
 * The per-feature tables (loc; for LAPLACE / NORMAL also scale and the density's constant) live in LDS: D <= 8192 for
 * LAPLACE / NORMAL, D <= 24576 for LPNORM*.
 */
int usf_base_logprob_f32(const float* z, int64_t ldz, int64_t M, int64_t D, int32_t base,
                         const float* loc, const float* scale, float logdet_const, const double* logdet_dev,
                         float* logp, double* sum_out, usf_stream_t stream);
/* per-feature tables of a Laplace / Normal base for usf_gemm_planes_desc.base_tab: tab[d] = loc[d],
 * tab[stride + d] = 1 / scale[d], tab[2 stride + d] = the density's constant (-log(2 b) resp. -log(sigma) - log sqrt(2 pi)),
 * zeros for D <= d < stride (stride: a multiple of 4, >= D). */
int usf_base_tables_f32(int32_t base, const float* loc, const float* scale, int64_t D, float* tab, int64_t stride,
                        usf_stream_t stream);

/*
 * Head of Flow.sample (flows.py:258): z ~ base, counter-based Philox4x32-10 RNG.
 *   LAPLACE: u~U(eps-1,1); z = loc - scale*sign(u)*log1p(-|u|)   (torch Laplace.rsample)
 *   NORMAL : Box-Muller;   z = loc + scale*n
 * Element (m,d) always consumes counter (m*D+d)/4 of stream (seed, offset): results do not
 * depend on the launch geometry, and ranks draw disjoint substreams via `row_offset`.
 */
int usf_base_sample_f32(float* z, int64_t ldz, int64_t M, int64_t D, int32_t base,
                        const float* loc, const float* scale, uint64_t seed, uint64_t offset,
                        int64_t row_offset, usf_stream_t stream);

/*
 * RadialDistribution.sample (distributions.py:474-499): z[m,:] = loc + r[m] * u_m with u_m uniform on
 * the unit Lp sphere as UniformUnitLpBall.sample draws it (distributions.py:283-319): base = USF_BASE_LPNORM1
 * (Dirichlet(1..1) x random signs), LPNORM2 (normalised normals), LPNORMINF (Uniform(-1,1) coordinates, one uniformly
 * chosen coordinate set to +1.0).  r [M]: radii drawn by the caller from the norm distribution.  Philox substreams as
 * usf_base_sample_f32 (row m of rank-offset row_offset always consumes the same counters).
 */
int usf_radial_sample_f32(float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc, const float* r,
                          uint64_t seed, uint64_t offset, int64_t row_offset, usf_stream_t stream);

/*
 * RadialDistribution.log_prob in ONE launch (distributions.py:501-549 -- the base of every live image configuration:
 * experiments/mnist/mnist.yaml:79-92, fashion/fashionclasses_veriflow.yaml:79-93, cifar/cifar.yaml), replacing
 * usf_base_logprob_f32(LPNORM*) + the torch finishing formula on the [M] radius vector:
 *     r[m]    = ||z[m,:] - loc||_p                       p_id = USF_BASE_LPNORM1 / LPNORM2 / LPNORMINF; D = prod(event shape)
 *     logp[m] = log sum_k pi_k f_k(r[m]) - (logdv_const + (D - 1) log r[m]) + logdet_const (+ *logdet_dev)
 * with the norm distribution a K-component mixture (1 <= K <= 64; K == 1: no mixture, logits may be NULL) of
 *     USF_NORM_LOGNORMAL: f_k = torch LogNormal(par_a[k], softplus(par_b[k]))        (distributions.py:181-197, 822-834)
 *     USF_NORM_GAMMA:     f_k = torch Gamma(softplus(par_a[k]), softplus(par_b[k]))  (distributions.py:162-179, 674-707)
 * pi = softmax(logits) (MixtureSameFamily).  par_* are the modules' STORED parameters (positive ones through softplus, as
 * DistributionModule._get_distribution_params applies it); OR-ing USF_NORM_RAW_PARAMS into `norm` takes them as they are
 * (a plain torch distribution).  logdv_const = the r-independent part of log_delta_volume(p, r) (distributions.py:513-549),
 * computed by the caller in fp64.  The O(K) finishing math runs in fp64 inside the kernel.  r_out (optional, [M]) receives the
 * radii for the backward pass; sum_out as usf_base_logprob_f32.
 *
 * usf_radial_logprob_grad_f32: from g_lp [M] (gradient at logp) and the saved r:
 *     g[m,d]   = g_lp[m] * dlogp/dr * dr/dz[m,d]   (0 for D <= d < ldg; ATen's norm backward: p = 1 sign(t), p = 2 t / r,
 *                                                   p = inf sign(t) / (number of d with |t| == r) where |t| == r: tied
 *                                                   maxima share the gradient evenly, as x.norm(p=inf) does)
 *     r == 0 (z == loc exactly): log r = -inf and (D - 1) / r = inf enter the row as they do in the reference
 *     (distributions.py:506-549): logp is +-inf or NaN depending on the norm distribution and the row of g is NaN (p = 2: t / r
 *     is taken as 0) -- nothing is clamped.
 *     d_loc[d] = -sum_m g[m,d];   d_a / d_b / d_logits [K] = gradients of the STORED parameters (chain rule through softplus)
 * each output pointer but g is optional.  z == NULL (both entry points): the radii are GIVEN -- the forward reads r_out as its
 * input, the backward writes g [M] = the gradient at r (d_loc must be NULL): the finishing formula alone, for callers whose own
 * tail kernel reduces the radius (the flat training path).  Partial sums are added in a fixed order (bit-reproducible).  workspace: at least
 * usf_radial_logprob_grad_workspace(M, D) bytes, 8-byte aligned.
 */
#define USF_NORM_LOGNORMAL 0
#define USF_NORM_GAMMA 1
#define USF_NORM_RAW_PARAMS 0x100
int usf_radial_logprob_f32(const float* z, int64_t ldz, int64_t M, int64_t D, int32_t p_id, const float* loc, int32_t norm,
                           int32_t K, const float* par_a, const float* par_b, const float* logits, double logdv_const,
                           float logdet_const, const double* logdet_dev, float* logp, float* r_out, double* sum_out,
                           usf_stream_t stream);
int64_t usf_radial_logprob_grad_workspace(int64_t M, int64_t D);
int usf_radial_logprob_grad_f32(const float* z, int64_t ldz, const float* r, const float* g_lp, int64_t M, int64_t D, int32_t p_id,
                                const float* loc, int32_t norm, int32_t K, const float* par_a, const float* par_b,
                                const float* logits, float* g, int64_t ldg, float* d_loc, float* d_a, float* d_b, float* d_logits,
                                void* workspace, int64_t workspace_bytes, usf_stream_t stream);

/*
 * The random-word -> variate maps of the two head kernels above, applied to caller-supplied 32-bit words:
 *   u[i]           = ((bits[i] >> 9) + 0.5) * 2^-23      in (0,1), never 0 or 1 (every step exact in fp32)
 *   laplace[i]     = -sign(2u-1) * log1p(-|2u-1|)        (torch Laplace.rsample, standard scale; always finite)
 *   exponential[i] = -log(u)                             (the Dirichlet / Box-Muller radius ingredient)
 * Each output is optional (NULL).  Used by the tests to pin the extreme words 0 and 0xFFFFFFFF; a caller that brings
 * its own generator can use it as the inverse-CDF stage of Flow.sample (flows.py:258).
 */
int usf_variates_from_bits_f32(const uint32_t* bits, int64_t n, float* u, float* laplace, float* exponential,
                               usf_stream_t stream);

/* ScaleTransform.forward / backward as a standalone layer (transforms.py:105-125): y = x*s or x/s */
int usf_scale_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t D,
                  const float* s, int32_t divide, usf_stream_t stream);

/*
 * Affine (scale-and-shift) coupling -- the EXTENSION named by BASELINE.json's north_star.  The reference has additive
 * coupling only (transforms.py:254-347, log_abs_det_jacobian == 0.0 at :316-326; the vestige AdditiveAffineNN,
 * networks.py:14-37, fixes log_scale to 0): there is nothing to be identical to -- parity unpinned, opt-in
 * (usflows_amd.transforms.AffineMaskedCoupling), and such a flow is NOT uniformly scaling (no UDL property).
 * Masked scale / shift apply on the n transformed columns + the per-sample log|det J| reduction (wave shuffle):
 *   inverse == 0:  z[m,j] = z[m,j] * exp(s[m,j]) + t[m,j],      logdet[m] += sum_j s[m,j]
 *   inverse != 0:  z[m,j] = (z[m,j] - t[m,j]) * exp(-s[m,j]),   logdet[m] -= sum_j s[m,j]
 * with s = bound * tanh(raw / bound) (bound > 0) or raw (bound <= 0); z / t / s row-major, strides ldz / ldt / lds;
 * logdet [M] may be NULL.  t and s come from the conditioner MLP (usf_linear_f32 launches).  HBM-bound.
 */
int usf_affine_coupling_apply_f32(float* z, int64_t ldz, const float* t, int64_t ldt, const float* s, int64_t lds, int64_t M,
                                  int64_t n, float bound, int32_t inverse, float* logdet, usf_stream_t stream);

/*
 * BlockAffineTransform on image-shaped inputs (in_dims = [C, H, W]): the 1 x 1 convolution of
 * transforms.py:904-962 (F.conv2d with the C x C block matrix viewed [C, C, 1, 1]) on NCHW-contiguous data,
 *   y[b, c, p] = sum_c' W[c, c'] * (x[b, c', p] - pre_sub[c']) + bias[c],   p < P = H * W,  1 <= C <= 64
 * forward: W = M, bias = b; backward: W = M^-1, pre_sub = b (each optional).  x != y.  HBM-bound (8 bytes per
 * element for 2 C flops): every element is read once, coalesced along p.
 */
int usf_channel_affine_f32(const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* W,
                           const float* pre_sub, const float* bias, usf_stream_t stream);

/*
 * Elementwise pieces of the image-shaped coupling layer, one pass each over contiguous [B, C, P] fp32 tensors (P = H * W):
 *   usf_layernorm_channels_f32: y = (a - mean_c a) / sqrt(var_c a + eps) * gamma + beta with a = act(x) (act = USF_ACT_NONE or
 *     USF_ACT_LEAKY_RELU with slope; slope 0 = ReLU): LayerNormChannels.forward (networks.py:40-58; biased variance over
 *     the channel axis) with the nonlinearity ConvNet2D puts in front of it (networks.py:480-493).  C <= 64.
 *   usf_gated_residual_f32:     y = x + vg[:, :C] * sigmoid(vg[:, C:]), vg [B, 2C, P]: GatedConv.forward (networks.py:108-122);
 *     CP = C * P.
 *   usf_masked_residual_f32:    y = x + sign * one_minus_mask * t, one_minus_mask [C * P] broadcast over the batch:
 *     MaskedCoupling.forward (+1) / backward (-1) on image-shaped inputs (transforms.py:277-306).  x may be NULL (taken
 *     as zeros: y = sign * one_minus_mask * t, the gradient of the layer with respect to t, or any broadcast mask product).
 */
int usf_layernorm_channels_f32(const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* gamma,
                               const float* beta, float eps, int32_t act, float slope, usf_stream_t stream);
int usf_gated_residual_f32(const float* x, const float* vg, float* y, int64_t B, int64_t CP, usf_stream_t stream);
int usf_masked_residual_f32(const float* x, const float* t, const float* one_minus_mask, float sign, float* y, int64_t B,
                            int64_t CP, usf_stream_t stream);

/*
 * Pointwise (1 x 1) convolution with few input channels on the vector ALUs, contiguous [B, cin, P] fp32 -> [B, cout, P]
 * (plain) or [B, cout / 2, P] (gated), P = H * W; W [cout, cin] row-major (nn.Conv2d weight [cout, cin, 1, 1]), bias [cout] or NULL:
 *   plain (gate_x == NULL): y[b, co, p] = out_act(bias[co] + sum_ci W[co, ci] * in_act(x[b, ci, p]))
 *   plain with out_act == USF_ACT_GATE (a data gradient; gate_x [B, cout, P] = the forward layer's input):
 *     y[b, co, p] = (bias[co] + W[co] . in_act(x)) * (gate_x[b, co, p] > 0 ? 1 : out_slope)
 *   gated (gate_x [B, C, P], cout = 2 C): y[b, c, p] = gate_x[b, c, p] + (bias[c] + W[c] . a) * sigmoid(bias[C + c] + W[C + c] . a)
 *     -- GatedConv.forward's second convolution with `x + val * sigmoid(gate)` (networks.py:108-122) in one pass.
 *   gated + layer norm (ln_gamma / ln_beta [C] non-NULL; needs cout == 2 cin, cin <= 32): the gated result r, then
 *     y = (a' - mean_c a') / sqrt(var_c a' + ln_eps) * ln_gamma + ln_beta with a' = out_act(r) -- the nonlinearity and the
 *     LayerNormChannels that follow a GatedConv in ConvNet2D (networks.py:480-493, :40-58) joined to the same pass.
 * Exact fp32 FMAs (plain: sums over ci in ascending order; gated: an even-ci and an odd-ci partial sum per row, packed FMAs,
 * added at the end).  cin in {8, 16, 24, 32, 48, 64}, cout <= 256
 * (usf_pointwise_conv_supported: 1 if the shape is served); HBM-bound: 4 (cin + cout) bytes per pixel.
 */
int usf_pointwise_conv_supported(int64_t cin, int64_t cout, int32_t gated);
int usf_pointwise_conv_f32(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t P, const float* W,
                           const float* bias, int32_t in_act, float in_slope, int32_t out_act, float out_slope,
                           const float* gate_x, const float* ln_gamma, const float* ln_beta, float ln_eps, usf_stream_t stream);

/*
 * Conv2d of the CNN conditioner (networks.py:405-510 ConvNet2D, :61-122 GatedConv): stride 1, dilation 1, "same" zero
 * padding, kernel ks = 1 or 3, on contiguous NCHW fp32 tensors, fp32-equivalent bf16x3 arithmetic on the matrix cores:
 *   y[b, co, p] = out_act( bias[co] + sum_{tap, ci} W[co, ci, tap] * (in_act(x[b, ci, p + tap]) * in_mul[ci, p + tap]) )
 * in_act / out_act: USF_ACT_NONE or USF_ACT_LEAKY_RELU (slope 0 = ReLU); in_mul [cin * H * W] or NULL (the coupling mask
 * in front of a conditioner's first convolution); bias [cout] or NULL.  Channels <= 64, H * W <= 256.
 * w_planes: three bf16 planes [3][coutp][kp] of the weight with W1 + W2 + W3 == W (round-to-nearest residual split),
 * coutp = ceil16(cout), K order tap-major / channel-minor with the channels padded to cp = ceil8(cin):
 * element [co][tap * cp + ci] = W[co, ci, tap / ks, tap % ks]; kp = ceil32(ks * ks * cp); zeros in all padding
 * (usf_conv2d_weight_elems(cin, cout, ks) = 3 * coutp * kp elements).
 * Gated mode (gate_x != NULL, gate_channels = C): the convolution is GatedConv's second one (networks.py:108-122,
 * 2C output channels: C values, then C gates) fused with the gate: y [B, C, H, W] = gate_x + value * sigmoid(gate), and
 * the [B, 2C, H, W] tensor never exists.  The caller packs rows (and bias) interleaved in tiles of 16: packed row
 * 32 t + r = value channel 16 t + r for r < 16, gate channel 16 t + r - 16 (original row C + 16 t + r - 16) for r >= 16,
 * zero rows where 16 t + r >= C; cout = 32 * ceil(C / 16) is passed; out_act must be USF_ACT_NONE.
 */
int64_t usf_conv2d_weight_elems(int64_t cin, int64_t cout, int64_t ks);
/* The planes above from an fp32 nn.Conv2d weight w [cout, cin, ks, ks] on the device, one launch.  transposed == 0: planes of
 * this convolution (usf_conv2d_weight_elems(cin, cout, ks) bf16 elements).  transposed == 1: planes of the convolution that
 * computes its DATA gradient -- cout -> cin channels with W'[ci, co, ky, kx] = w[co, ci, ks-1-ky, ks-1-kx]
 * (usf_conv2d_weight_elems(cout, cin, ks) elements).  transposed == 2: both, back to back in `planes` (the first set, then the
 * second), in one launch -- what a training step needs of every convolution.  Not for the gated row packing (the caller
 * packs that on the host side). */
int usf_conv2d_weight_planes_f32(const float* w, void* planes, int64_t cin, int64_t cout, int64_t ks, int32_t transposed,
                                 usf_stream_t stream);
/* > 0 (samples per LDS group) when usf_conv2d_same_f32 serves these sizes: weight planes + one padded sample must fit 158 KB of LDS */
int usf_conv2d_same_fits(int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks);
int usf_conv2d_same_f32(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                        const void* w_planes, const float* bias, const float* in_mul, int32_t in_act, float in_slope,
                        int32_t out_act, float out_slope, const float* gate_x, int64_t gate_channels, usf_stream_t stream);
/* The LAST convolution of a coupling layer's conditioner with MaskedCoupling's masked residual (transforms.py:277-306) in
 * its output stream:  y = res_x + res_sign * (res_mul * conv(in_act(x) * in_mul)),  res_x [B, cout, H, W] (the coupling's
 * input), res_mul [cout * H * W] = 1 - mask -- the arithmetic of usf_conv2d_same_f32 followed by usf_masked_residual_f32,
 * without the conditioner's output tensor ever reaching HBM.  Returns 0 when done, 1 when this shape is not served by the
 * fused form (3 x 3 kernel, 16 / 32 input channels, 16 / 32 / 64 output channels, H W <= 64, 16-byte aligned tensors):
 * the caller then runs the two entry points one after the other; < 0: error. */
int usf_conv2d_same_res_f32(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                            const void* w_planes, const float* bias, const float* in_mul, int32_t in_act, float in_slope,
                            const float* res_x, const float* res_mul, float res_sign, usf_stream_t stream);

/*
 * Gradients of the image-shaped coupling layer's pieces: what torch autograd computes for networks.py:40-122, 405-510 and the
 * 1 x 1 convolution of transforms.py:904-962 when Flow.fit (flows.py:113-210) trains an image flow.  All tensors contiguous
 * fp32; sums over the batch are deterministic (per-wave partial sums in the caller's workspace, added in a fixed order).
 *
 * usf_conv_wgrad_f32: weight and bias gradient of a stride-1 "same" convolution with kernel ks = 1 or 3,
 *     dW[co, ci, ky, kx] = sum_{b, p} dy[b, co, p] * xin[b, ci, p + (ky - ks/2, kx - ks/2)]     (nn.Conv2d weight layout [cout, cin, ks, ks])
 *     db[co]             = sum_{b, p} dy[b, co, p]                                             (db may be NULL)
 *   with xin = in_act(x - pre_sub[ci]) * in_mul -- the input transforms of usf_conv2d_same_f32 (in_act, in_mul) and of
 *   usf_channel_affine_f32 (pre_sub), each optional -- and zeros outside the image.  Exact fp32 products and sums on
 *   v_mfma_f32_16x16x4_f32.  Served: cin, cout multiples of 16 up to 64 (kernel 3: cin * cout <= 1536), H * W <= 64, W >= 2, 16-byte
 *   aligned tensors; returns 1 (nothing written) for other shapes, 0 when done, < 0 on error.  workspace: at least
 *   usf_conv_wgrad_workspace(...) floats (0 = shape not served).  The DATA gradient of these layers is the forward entry point
 *   on the flipped, transposed weight (usf_conv2d_same_f32 / usf_pointwise_conv_f32 / usf_channel_affine_f32).
 * usf_layernorm_channels_bwd_f32: backward of usf_layernorm_channels_f32 (same x, gamma, eps, act, slope):
 *     dx [B, C, P]; dgamma_dbeta [2 C] = (sum dy * xhat, sum dy); workspace >= usf_layernorm_channels_bwd_workspace(B, C, P) floats.
 * usf_gated_residual_bwd_f32: backward of usf_gated_residual_f32 with respect to vg: dvg [B, 2C, P] =
 *     (dy * sigmoid(gate), dy * val * sigmoid(gate) * (1 - sigmoid(gate))); the gradient with respect to x is dy itself.
 */
/* Deferred sums of per-wave partial slots (the last stage of usf_conv_wgrad_f32) and many of them in ONE launch.
 * usf_conv_wgrad_deferred_f32 = usf_conv_wgrad_f32 that stops in front of that stage: job[0 .. 1] (HOST memory, two entries)
 * then describe what remains -- job[1] the final round that writes dW / db, job[0] the first round into the workspace's
 * scratch rows when there are more than 64 slots (job[0].nparts == 0: a single round).  dW / db stay UNWRITTEN until
 * usf_partial_sum_jobs_f32 launches containing first job[0] (if any), then job[1] have run in this stream order; the workspace
 * must stay alive until then.
 * usf_partial_sum_jobs_f32: jobs / block_job are DEVICE arrays: job j owns the blocks [first_block, first_block +
 * ceil(n / 64) * rows) -- ceil(n / 256) * rows when vec4 != 0 (mode 0, n a multiple of 4, 16-byte aligned part / out: four
 * columns per thread) -- and block_job[b] names block b's job (n_blocks entries).  Row r of a job sums the slots
 * [r * per, min((r + 1) * per, nparts)) in the order usf_conv_wgrad_f32's own rounds use: same bits.
 * Why: at the reference's training batch (32 rows, experiments/mnist/mnist.yaml:34) a backward pass of the live MNIST
 * configuration ends ~165 weight gradients with one or two such launches of a few microseconds each, all on the chain of
 * dependent launches that bounds the step; queued, they are two launches behind the pass. */
typedef struct usf_psum_job {
  const float* part; float* out; float* out2;
  int32_t nparts, n, mode, cin, cout, CIT, T, ntile;
  int32_t first_block, per, rows, vec4;
} usf_psum_job;
int usf_conv_wgrad_deferred_f32(const float* x, const float* dy, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                                const float* in_mul, const float* pre_sub, int32_t in_act, float in_slope, float* dW, float* db,
                                float* workspace, int64_t workspace_floats, usf_psum_job* job, usf_stream_t stream);
int usf_partial_sum_jobs_f32(const usf_psum_job* jobs, const int32_t* block_job, int64_t n_blocks, usf_stream_t stream);
/* The weight-gradient kernel itself queued as well (small batches: one weight gradient occupies an eighth of the chip).
 * usf_conv_wgrad_plan_f32 = usf_conv_wgrad_deferred_f32 that launches NOTHING when the shape runs on the LDS-staged kernel: it
 * fills *wjob (HOST memory; blocks > 0) with that kernel's arguments -- the caller later runs all queued jobs of equal
 * (CIT, COT, T) with ONE usf_conv_wgrad_jobs_f32 launch (jobs / block_job DEVICE arrays as for usf_partial_sum_jobs_f32: job j
 * owns the blocks [first_block, first_block + blocks), first_block set by the caller; lds_bytes = the largest of the jobs')
 * and then the sums job[0 .. 1] describe.  x, dy, in_mul, pre_sub and the workspace must stay alive and unchanged until then.
 * wjob->blocks == 0 on return: the shape runs on the direct kernel-1 form, which HAS been launched (only the sums remain). */
typedef struct usf_wgrad_job {
  unsigned char args[192];                      /* the kernel's arguments (opaque) */
  int32_t CIT, COT, T, blocks, lds_bytes, first_block;
} usf_wgrad_job;
int usf_conv_wgrad_plan_f32(const float* x, const float* dy, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                            const float* in_mul, const float* pre_sub, int32_t in_act, float in_slope, float* dW, float* db,
                            float* workspace, int64_t workspace_floats, usf_psum_job* job, usf_wgrad_job* wjob, usf_stream_t stream);
int usf_conv_wgrad_jobs_f32(const usf_wgrad_job* jobs, const int32_t* block_job, int64_t n_blocks, int32_t CIT, int32_t COT, int32_t T,
                            int32_t lds_bytes, usf_stream_t stream);
int64_t usf_conv_wgrad_workspace(int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks);
int usf_conv_wgrad_f32(const float* x, const float* dy, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                       const float* in_mul, const float* pre_sub, int32_t in_act, float in_slope, float* dW, float* db,
                       float* workspace, int64_t workspace_floats, usf_stream_t stream);
int64_t usf_layernorm_channels_bwd_workspace(int64_t B, int64_t C, int64_t P);
int usf_layernorm_channels_bwd_f32(const float* x, const float* dy, float* dx, int64_t B, int64_t C, int64_t P, const float* gamma,
                                   float eps, int32_t act, float slope, float* dgamma_dbeta, float* workspace,
                                   int64_t workspace_floats, usf_stream_t stream);
int usf_gated_residual_bwd_f32(const float* dy, const float* vg, float* dvg, int64_t B, int64_t CP, usf_stream_t stream);

/* usf_conv2d_weight_planes_f32(transposed = 2) for MANY weights in one launch.  jobs / block_job are DEVICE arrays (as for
 * usf_partial_sum_jobs_f32): job j splits the fp32 weight w [cout, cin, ks, ks] into its planes followed by the planes of its
 * data-gradient convolution, at planes_base + out_off (bf16 elements; usf_conv2d_weight_elems(cin, cout, ks) +
 * usf_conv2d_weight_elems(cout, cin, ks) of them), and owns the blocks [first_block, first_block + ceil(max of the two
 * [rows x K] sizes / 256)); block_job[b] names block b's job.  Same bits as the single launches. */
typedef struct usf_wplanes_job {
  const float* w; int64_t out_off;
  int32_t cin, cout, ks, first_block;
} usf_wplanes_job;
int usf_conv2d_weight_planes_batch_f32(const usf_wplanes_job* jobs, const int32_t* block_job, int64_t n_blocks, void* planes_base,
                                       usf_stream_t stream);

/* The tail of a GatedConv layer of ConvNet2D (reference networks.py:108-122 with the nonlinearity and LayerNormChannels that
 * follow it, networks.py:40-58, 480-493) at training batches, forward and backward ONE launch each:
 *     a = in_act(h); [val, gate] = W a + bias (W [2 C, C], bias [2 C] or NULL); r = x + val * sigmoid(gate);
 *     y = LayerNormChannels(post_act(r); gamma, beta, eps)    (ln_gamma == NULL: y = r, post_act must be USF_ACT_NONE)
 * on contiguous [B, C, P] fp32 tensors, C in {8, 16, 24, 32} (usf_gated_tail_supported).  It replaces the chain
 * usf_pointwise_conv_f32 -> usf_gated_residual_f32 -> usf_layernorm_channels_f32 and, backward, usf_layernorm_channels_bwd_f32
 * -> usf_gated_residual_bwd_f32 -> usf_pointwise_conv_f32 on a transposed copy of W -> usf_conv_wgrad_f32 (kernel 1): the
 * backward recomputes val / gate / r from (h, x) and writes dx [B, C, P] (the skip branch), dh [B, C, P] and
 * dparams = [dW (2 C C) | dbias (2 C) | dgamma (C) | dbeta (C)] (the last two only with a layer norm); dvg [B, 2 C, P] =
 * d[val, gate] is written when the pointer is not NULL.  workspace >= usf_gated_tail_workspace floats.
 * job == NULL: dparams is complete when the call's launches have run; else job[0 .. 1] (HOST memory) describe its last
 * sum for usf_partial_sum_jobs_f32 as usf_conv_wgrad_deferred_f32 does (nparts == 0: nothing to do).
 * Eight lanes share a pixel: made for few pixels (a 32-row training batch); HBM traffic 4 C (3 + 2) bytes per pixel backward. */
int usf_gated_tail_supported(int64_t C);
int64_t usf_gated_tail_workspace(int64_t B, int64_t C, int64_t P);
int usf_gated_tail_f32(const float* h, const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* W, const float* bias,
                       int32_t in_act, float in_slope, int32_t post_act, float post_slope, const float* ln_gamma,
                       const float* ln_beta, float ln_eps, usf_stream_t stream);
int usf_gated_tail_bwd_f32(const float* h, const float* x, const float* dy, float* dx, float* dh, float* dvg, int64_t B, int64_t C,
                           int64_t P, const float* W, const float* bias, int32_t in_act, float in_slope, int32_t post_act,
                           float post_slope, const float* ln_gamma, const float* ln_beta, float ln_eps, float* dparams,
                           float* workspace, int64_t workspace_floats, struct usf_psum_job* job, usf_stream_t stream);

/* A data-gradient convolution with the factors of the layer's INPUT transforms in its output stream:
 *   y = conv(x) * (gate_h > 0 ? 1 : gate_slope) * gate_mul          (gate_add == NULL)
 *   y = gate_add + conv(x) * (gate_h > 0 ? 1 : gate_slope)          (gate_add != NULL; then gate_mul must be NULL)
 * gate_h [B, cout, H, W] = the forward layer's input (its (Leaky)ReLU's derivative; gate_slope 0 = ReLU, 1 = no nonlinearity),
 * gate_mul [cout * H * W] or NULL = the forward layer's input mask, gate_add [B, cout, H, W] or NULL = the gradient that
 * reaches the same tensor along another branch (the forward input forks: GatedConv's skip connection, networks.py:108-122).
 * The arithmetic of usf_conv2d_same_f32 followed by usf_act_grad_f32 and the mask product / the sum, in one pass.  Returns 0
 * when done, 1 when the shape is not served by this form (as usf_conv2d_same_res_f32), < 0 on error. */
int usf_conv2d_same_gate_f32(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                             const void* w_planes, const float* gate_h, float gate_slope, const float* gate_mul,
                             const float* gate_add, usf_stream_t stream);

/* column gather/scatter between the user's natural layout and the engine's segment layout:
 * dst[m, j] = src[m, idx[j]] for j < n (idx: int32 device array); idx[j] < 0 writes 0. */
int usf_gather_cols_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int64_t n,
                        const int32_t* idx, usf_stream_t stream);

/*
 * ---- "Planes pipeline" (large batches) ---------------------------------------------------
 * Between the dense layers of a flow the activations travel as three bf16 planes (x == p1 + p2 + p3 exactly,
 * round-to-nearest residual split -- the operand form of the bf16x3 arithmetic above) in the order the consumer's
 * MFMA operand wants them, so that a GEMM's K loop is loads + MFMAs only (no split, no LDS transposition):
 *
 *   planes buffer of M rows and nkb 32-feature blocks: ceil(M/16) row panels x nkb blocks x 3 planes of 1 KiB
 *   chunks, chunk(p, kb, q) at ((p * nkb + kb) * 3 + q) * 1024 bytes; inside a chunk 64 lines of 16 bytes
 *   (8 bf16): line L = 16 g + j holds row 16 p + j, slots 8 g .. 8 g + 7 of block kb; slot s = 8 g + u holds the
 *   block's feature 16 (u >> 2) + 4 g + (u & 3).  Rows >= M of the last panel are padding (any value).
 *
 * The "logical layout" of a buffer (which feature sits at logical position 32 kb + f) is the caller's business
 * (the engine uses its segment layout [mask==0 features | mask==1 features]).
 */

/* fp32 row-major -> planes:  logical position l of row m  <-  (src[m, idx[l]] / pre_div[l]) - pre_sub[l]
 * (idx[l] < 0: zero; pre_div / pre_sub optional).  Head of Flow.log_prob (ScaleTransform.backward + the tail affine
 * layer's bias subtraction, transforms.py:116-125, 960) and of Flow.sample (base noise). */
#define USF_PLANES_BF16X3 0 /* three bf16 planes: 24 significant bits, fp32's exponent range; six MFMAs per product */
#define USF_PLANES_F16X2 1  /* two fp16 planes: 22 significant bits; three MFMAs per product (a1 w1 + a1 w2 + a2 w1);
                             * values must stay inside fp16's range: see range_flag.  Weight planes of this format:
                             * usf_pack_weight_f32 with bit 1 of `transpose` set (two planes instead of three) */
typedef struct usf_pack_planes_desc {
  const float* src; int64_t ld;         /* [M, ld] */
  int64_t M;
  int64_t nkb;                          /* blocks per panel of the planes buffer */
  const int32_t* idx;                   /* [32 * nkb] device */
  const float* pre_div;                 /* [32 * nkb] or NULL */
  const float* pre_sub;                 /* [32 * nkb] or NULL */
  void* planes;
  int32_t format, reserved;             /* USF_PLANES_* */
  int32_t* range_flag;                  /* F16X2 only, may be NULL: set to 1 when a value is NaN or |x| >= 65000 (cannot
                                           travel as fp16): the caller must then redo the pass in BF16X3 */
  /* ABI 33 */
  int64_t src_cols;                     /* > 0: every idx[l] < src_cols -- lets the kernel read whole rows coalesced and gather out
                                           of LDS (16 (src_cols + 1) + 96 nkb floats must fit 64 KB); 0: unknown (one gather per element) */
  /* optional, src_cols > 0 only: the source is transformed on the way in -- the head of the TRAINING backward pass,
   * src = the latent z and the planes receive  g[m, c] = row_weight[m] * d/dz base_c(z[m, c])  (usf_base_logprob_grad_f32's
   * formulas for USF_BASE_LAPLACE / USF_BASE_NORMAL with loc / scale [src_cols]; grad_base = 1 + base id, 0 = plain copy) */
  const float* row_weight; const float* loc; const float* scale;
  int32_t grad_base, reserved2;
} usf_pack_planes_desc;
int usf_pack_planes_f32(const usf_pack_planes_desc* d, usf_stream_t stream);

/*
 * Dense layer on planes:  C = epilogue(A[:, K range] @ W^T + bias)   (F.linear of BlockAffineTransform,
 * transforms.py:913-962, and of the conditioner MLP, networks.py:739-751; MaskedCoupling's residual add / subtract,
 * transforms.py:277-306, in the epilogue).  Arithmetic: bf16x3 (six v_mfma_f32_16x16x32_bf16 per product, fp32
 * accumulation) -- the same as usf_linear_f32 with W_split.
 *   A        planes buffer with a_nkb blocks per panel; the K range is blocks a_kb0 .. a_kb0 + nk - 1
 *   W_planes three bf16 planes [w_rows, ldw] (plane q at + q * w_plane_stride elements), ldw >= 32 nk; column
 *            32 kb + s of a row multiplies SLOT s of block a_kb0 + kb (i.e. the K axis carries the slot permutation
 *            above); rows beyond the wanted outputs must be zero (w_rows: a multiple of 4, >= 32 c_kbn resp. >= N)
 *   bias, post_mul  [w_rows] fp32 or NULL (post_mul: fp32 output only)
 *   planes output (C_planes != NULL): output feature n lands at logical position 32 c_kb0 + n of a buffer with c_nkb
 *            blocks per panel, for n < 32 c_kbn;  v = act(acc + bias);  with residual (a planes buffer of C's
 *            geometry, may alias C_planes): v = residual + res_sign * v
 *   fp32 output (C_f32 != NULL): C_f32[m, n] = act(acc + bias) * post_mul, n < N, row-major with stride ldc
 *   base density in the epilogue (base_part != NULL; fp32-output form, C_f32 may then be NULL: the rows are not stored):
 *            the layer is the LAST of Flow.log_prob (flows.py:225-234) and its result z only feeds
 *            base_distribution.log_prob: base_part[m, j] = sum over the columns n of column block j (32 TN columns, TN as
 *            usf_gemm_planes_variant reports; at most 8 blocks: N <= 1024) of the Laplace / Normal terms of
 *            usf_base_logprob_f32 at z[m, n], with 1 / scale and the per-feature constant taken from base_tab
 *            (usf_base_tables_f32).  base_part is [M, 8] fp32; usf_base_logprob_f32(USF_BASE_ROWSUM) finishes the rows.
 */
typedef struct usf_gemm_planes_desc {
  const void* A; int64_t a_nkb, a_kb0, nk;
  const void* W_planes; int64_t ldw, w_plane_stride, w_rows;
  const float* bias;
  const float* post_mul;
  const void* residual;
  void* C_planes; int64_t c_nkb, c_kb0, c_kbn;
  float* C_f32; int64_t ldc, N;
  int64_t M;
  float res_sign, slope;
  int32_t act;
  int32_t format;                       /* USF_PLANES_*: of A, W_planes, residual and C_planes alike */
  int32_t* range_flag;                  /* F16X2 only, may be NULL: set to 1 when an output that has to travel as fp16
                                           planes is NaN or |x| >= 65000, or an fp32 output is not finite (an overflowed
                                           plane upstream) -- the results of the pass are void, redo it in BF16X3 */
  /* ABI 35 (appended): base density in the epilogue, see above; all zero = off */
  const float* base_tab; int64_t base_tab_stride;
  float* base_part;
  int32_t base, reserved;
} usf_gemm_planes_desc;
int usf_gemm_planes_bf16x3(const usf_gemm_planes_desc* d, usf_stream_t stream);
/* which instantiation serves the descriptor (nothing is launched): 5000 + 10 TN + (1: fp32 output, 0: planes output),
 * TN = column-block width in 32-feature blocks (4 or 5, whichever pads the output less); 0 for empty extents */
int usf_gemm_planes_variant(const usf_gemm_planes_desc* d);

/*
 * Fused additive coupling layer on a planes buffer: MaskedCoupling.forward / backward (transforms.py:277-306) with its
 * whole conditioner MLP (networks.py:739-751) in ONE launch, in place on z:
 *   z[:, transformed blocks] += sign * (W_out act(W_h act(W_in z[:, conditioning blocks] + b_in) + b_h) + b_out)
 * z: planes buffer with z_nkb blocks per panel; the conditioning features are blocks kb_p0 .. kb_p0 + nk_p - 1, the
 * transformed ones kb_t0 .. kb_t0 + nk_t - 1 (the ranges may share a straddling block: weights / rows of the other
 * set's features are zero, those values are rewritten unchanged).  Weight planes (format as z; usf_pack_weight_f32):
 *   W_in  [256, ldw_in >= 32 nk_p]   K axis = slots of the conditioning blocks;  b_in [256]
 *   W_hid[i] [256, ldw_hid >= 256]   K axis = slots of the previous hidden layer (n_hidden - 1 of them);  b_hid[i] [256]
 *   W_out [32 nk_t, ldw_out >= 256]  rows = logical positions of the transformed blocks;  b_out [32 nk_t]
 * hidden widths are padded to hidden_padded = 256 with zero rows / columns (1 <= n_hidden <= 3).  Same arithmetic as
 * usf_gemm_planes_bf16x3 applied layer by layer; hidden activations never leave the registers.  range_flag: as there.
 */
typedef struct usf_coupling_planes_desc {
  void* z; int64_t z_nkb; int64_t M;
  int64_t kb_p0, nk_p, kb_t0, nk_t;
  int32_t n_hidden, hidden_padded;
  const void* W_in; int64_t ldw_in, w_in_plane; const float* b_in;
  const void* W_hid[2]; const float* b_hid[2]; int64_t ldw_hid, w_hid_plane;
  const void* W_out; int64_t ldw_out, w_out_plane; const float* b_out;
  float sign, slope;
  int32_t act, format;
  int32_t* range_flag;
  /* Training (ABI 33; USF_PLANES_BF16X3, n_hidden <= 2).  Planes buffers with 8 blocks per panel (hidden width 256):
   *   hidden_out[l] (optional, all n_hidden or none): receives the activations of hidden layer l -- the lane-local splits
   *     the kernel makes anyway, i.e. the operands of the conditioner's weight gradients (usf_wgrad_blocked_f32) and the
   *     gates of the backward launch.
   *   act == USF_ACT_GATE (needs hidden_out and gate[l] for every layer): the launch runs the conditioner's data-gradient
   *     chain -- the caller passes the transposed weight images in reverse order, zero biases and swaps the block ranges:
   *     z = the gradient buffer, g[:, pass] += sign * MLP^T(g[:, trans]); layer l's nonlinearity is leaky_relu_backward
   *     from the saved activations gate[l] (v * (h > 0 ? 1 : slope); only plane 0 of gate[l] is read) and hidden_out[l]
   *     receives the gated values (the gradients at the pre-activations of forward hidden layer n_hidden - 1 - l).
   * Replaces autograd's backward of MaskedCoupling + its conditioner (transforms.py:277-306, networks.py:739-751) under
   * Flow.fit (flows.py:196-203) at batches of thousands of rows. */
  void* hidden_out[2];
  const void* gate[2];
} usf_coupling_planes_desc;
int usf_coupling_planes(const usf_coupling_planes_desc* d, usf_stream_t stream);

/*
 * Row pass of the vector ConvNet conditioner's blocks (networks.py:222-245 GatedMLP, :206-219 LayerNormVector, vector
 * branch of ConvNet.__init__ :287-308), rows of [M, C] fp32:
 *     r = skip + vg[:, :C] * sigmoid(vg[:, gate_off : gate_off + C])      (vg == NULL: r = skip)
 *     y = (r - mean r) / sqrt(var r + eps) * gamma + beta                  (gamma == beta == NULL: y = r; biased variance)
 *     out = y, out_act = act(y)                                            (either may be NULL, not both)
 * Columns [C, c_pad) of out / out_act are written as zeros (operand padding of usf_linear_f32).  out may alias skip.
 * 1 <= C <= c_pad <= 4096.
 */
typedef struct usf_gated_norm_desc {
  const float* skip;    int64_t ld_skip;
  const float* vg;      int64_t ld_vg;    int64_t gate_off;
  const float* gamma;   const float* beta;
  float*       out;     int64_t ld_out;
  float*       out_act; int64_t ld_act;
  int64_t M, C, c_pad;
  float eps, slope;
  int32_t act, reserved;
} usf_gated_norm_desc;
int usf_gated_norm_rows_f32(const usf_gated_norm_desc* d, usf_stream_t stream);
/* The backward twin (ABI 34): what torch.autograd derives from GatedMLP's gate (networks.py:222-245) and LayerNormVector
 * (:206-219) in Flow.fit, rows of [M, C] fp32, r / mean / variance recomputed from (skip, vg) as the forward computes them:
 *     g = dy * gamma;  dr = (g - mean_c g - xh * mean_c(g xh)) / sqrt(var r + eps)        (gamma == NULL: dr = dy)
 *     d_skip = dr;  d_vg[:, :C] = dr * sigmoid(gate);  d_vg[:, gate_off:] = dr * val * sigmoid(gate) (1 - sigmoid(gate))
 *     dy_xh = dy * xh  (optional, needs gamma: dgamma = its column sums, dbeta = the column sums of dy -- usf_colsum_f32)
 * Columns [C, c_pad) of every output are written as zeros (operand padding of the GEMMs that follow); gate_off >= c_pad. */
typedef struct usf_gated_norm_bwd_desc {
  const float* skip;   int64_t ld_skip;
  const float* vg;     int64_t ld_vg;    int64_t gate_off;
  const float* gamma;
  const float* dy;     int64_t ld_dy;
  float*       d_skip; int64_t ld_d_skip;
  float*       d_vg;   int64_t ld_d_vg;
  float*       dy_xh;  int64_t ld_dy_xh;
  int64_t M, C, c_pad;
  float eps, reserved;
} usf_gated_norm_bwd_desc;
int usf_gated_norm_rows_bwd_f32(const usf_gated_norm_bwd_desc* d, usf_stream_t stream);

/*
 * Run a prebuilt list of ops with ONE call (the whole flow: ~2K+2 launches). op.kind selects the
 * member of the union; pointers inside may be patched by the caller between calls.
 */
#define USF_OP_LINEAR 1
#define USF_OP_COUPLING 2
#define USF_OP_PACK_PLANES 5
#define USF_OP_GEMM_PLANES 6
#define USF_OP_COUPLING_PLANES 7
#define USF_OP_GATED_NORM 9
#define USF_OP_CALL 10
/* One call of an elementwise / image-path entry point inside an op list: the function and its arguments in the order of the
 * C prototype without the stream, every argument one 64-bit word (pointers and integers as they are, a float as its bit
 * pattern in the low 32 bits).  This is how a layer loop that is not a chain of the dense ops above (image-shaped flows:
 * Flow.log_prob / backward over usf_scale_f32, usf_channel_affine_f32, usf_conv2d_same_f32 / _res_f32, usf_pointwise_conv_f32,
 * the elementwise passes, usf_base_logprob_f32) runs as ONE usf_run_ops call: the host records the loop's calls once per
 * (input shape, parameter version) and replays the list. */
#define USF_FN_SCALE 1
#define USF_FN_CHANNEL_AFFINE 2
#define USF_FN_LAYERNORM_CHANNELS 3
#define USF_FN_GATED_RESIDUAL 4
#define USF_FN_MASKED_RESIDUAL 5
#define USF_FN_POINTWISE_CONV 6
#define USF_FN_CONV2D_SAME 7
#define USF_FN_CONV2D_SAME_RES 8
#define USF_FN_BASE_LOGPROB 9
#define USF_FN_RADIAL_LOGPROB 10
#define USF_FN_GATED_TAIL 11
#define USF_CALL_MAX_ARGS 20
typedef struct usf_call_desc {
  int32_t fn;
  int32_t n_args;
  uint64_t a[USF_CALL_MAX_ARGS];
} usf_call_desc;
typedef struct usf_op {
  int32_t kind;
  int32_t reserved;
  union {
    usf_linear_desc linear;
    usf_coupling_desc coupling;
    usf_pack_planes_desc pack_planes;
    usf_gemm_planes_desc gemm_planes;
    usf_coupling_planes_desc coupling_planes;
    usf_gated_norm_desc gated_norm;
    usf_call_desc call;
  } u;
} usf_op;

int usf_run_ops(const usf_op* ops, int32_t n_ops, usf_stream_t stream);

/*
 * ---- parameter prep on the device ---------------------------------------------------
 * The reference re-derives every matrix from the raw parameters on each call with ATen CPU ops; these entry
 * points do it in a handful of batched fp64 launches over all affine blocks of a flow (usf_prep.hip).
 * fp64 results; the caller rounds once to fp32 through usf_pack_weight_f32.
 *
 * usf_lu_prepare_f64 -- LUTransform.L / .U / .matrix / .inverse_matrix / .log_abs_det_jacobian
 *                       (transforms.py:1271-1320) for n blocks at once:
 *   tri[2i]     = L_i = tril(L_raw_i,-1) + I          tri[2i+1]     = U_i^T = triu(U_raw_i)^T   (both lower-triangular)
 *   tri_inv[2i] = L_i^-1                              tri_inv[2i+1] = (U_i^-1)^T
 *   M[i] = L_i U_i,  Minv[i] = U_i^-1 L_i^-1 (either may be NULL),  ladj[i] = sum log|diag U_i| (may be NULL)
 * L_raw / U_raw are HOST arrays of n DEVICE pointers to [D,D] fp32 row-major parameters; tri, tri_inv and the
 * scratch `work` are [2n,D,D] fp64, M / Minv [n,D,D] fp64, all caller-owned device memory.
 */
typedef struct usf_lu_prep_desc {
  int64_t n, D;
  const float* const* L_raw;
  const float* const* U_raw;
  double* tri;
  double* tri_inv;
  double* work;
  double* M;
  double* Minv;
  double* ladj;
} usf_lu_prep_desc;
int usf_lu_prepare_f64(const usf_lu_prep_desc* d, usf_stream_t stream);

/* Batched fp64 GEMM on the f64 MFMA: C[b] = alpha * op(A[b]) op(B[b]) + beta * C[b], row-major, op = identity or
 * transpose (transX != 0: X is stored [K,M] resp. [N,K]); A[b] = A + b*strideA etc.  tri, low 3 bits = k-range hint
 * (the skipped products must be exact zeros): 0 = full K, 1 = op(A) lower- and op(B) upper-triangular (k <= min(i,j)),
 * 2 = op(A) upper- and op(B) lower-triangular (k >= max(i,j)), 3 = op(A) lower-triangular (k <= i), 4 = op(B)
 * upper-triangular (k <= j); + 8: only the 64x64 output tiles that touch the upper triangle (diagonal included) are
 * computed, + 16: only those that touch the lower triangle -- the other tiles of C are left untouched (for results
 * whose other triangle is discarded, e.g. the triu / tril of the LU gradients).
 * SequentialAffineTransform.matrix / .inverse_matrix (transforms.py:1457-1469) and the matrix gradients of the
 * training path are chains of these. */
int usf_gemm_f64(const double* A, int64_t lda, int64_t strideA, int32_t transA, const double* B, int64_t ldb,
                 int64_t strideB, int32_t transB, double* C, int64_t ldc, int64_t strideC, int64_t M, int64_t N,
                 int64_t K, int64_t batch, double alpha, double beta, int32_t tri, usf_stream_t stream);

/* The last step of LUTransform's parameter gradients under Flow.fit (what autograd derives through transforms.py:1271-1320's
 * tril(L_raw, -1) + I and triu(U_raw), and the log-det term sum log|diag U|), for n blocks at once, fp64 in / fp32 out:
 *     dL_out[i] = tril(dL[i] (+ TL[i]), -1)                     dU_out[i] = triu(dU[i] (+ TU[i])) + diag(c[i] / diag(U_i))
 * dL / dU [n, D, D]: the chain-rule products (only the wanted triangle has to be valid); TL / TU: the products of the M = L U
 * usages, or NULL; c [n]: coefficient of the log-det term; tri [2n, D, D] as usf_lu_prepare_f64 leaves it (U_i^T at 2i + 1).
 * One pass instead of triu / tril / diagonal add / sums / converting copies over [n, D, D] tensors. */
int usf_lu_grad_finish_f64(const double* dL, const double* dU, const double* TL, const double* TU, const double* c,
                           const double* tri, int64_t n, int64_t D, float* dL_out, float* dU_out, usf_stream_t stream);

/* HouseholderTransform._construct_householder_permutation (transforms.py:795-809) as row-local rank-1 updates:
 * out = w_0 prod_k (I - 2 v_k v_k^T / v_k.v_k); w_0 [D,D] fp32, vk [nvs,D] fp32, out [D,D] fp64. */
int usf_householder_f64(const float* w_0, const float* vk, int64_t nvs, int64_t D, double* out, usf_stream_t stream);

/* fp64 (or, src_is_f32 != 0, fp32) matrix -> the fp32 weight image the kernels read, rounded once:
 *   W[o, c] = (float) src[out_idx[o], in_idx[c]]   (transpose & 1: src[in_idx[c], out_idx[o]]),   0 where an index is < 0
 * for o < n_out, c < n_in (idx: int32 device arrays or NULL = identity), and, if planes != NULL, the bf16x3 planes of
 * the same values ([3][n_out][ld_planes] bf16, zero for c >= n_in; usf_linear_desc.W_split) or, with transpose & 2,
 * the fp16x2 planes ([2][n_out][ld_planes] fp16: hi = fp16(w), lo = fp16(w - hi); USF_PLANES_F16X2).  W may be NULL.
 * Also builds the mask-aware, zero-padded (and, for the split planes, k-permuted) conditioner weights of
 * usf_coupling_desc from the nn.Linear parameters (networks.py:711-737) and, with n_out == 1, permuted vectors. */
int usf_pack_weight_f32(const void* src, int32_t src_is_f32, int64_t ld_src, int32_t transpose, const int32_t* out_idx, int64_t n_out,
                        const int32_t* in_idx, int64_t n_in, float* W, int64_t ldw, void* planes, int64_t ld_planes,
                        int64_t plane_stride, usf_stream_t stream);

/* The same for many images in ONE launch: `jobs` is a DEVICE array of n_jobs descriptors (the arguments of
 * usf_pack_weight_f32); max_rows / max_cols bound n_out and the written columns over all jobs.  A parameter update
 * refreshes hundreds of small images; batched, that is one dispatch instead of hundreds of dependent ones. */
typedef struct usf_pack_job {
  const void* src;
  const int32_t* out_idx;
  const int32_t* in_idx;
  float* W;
  void* planes;
  int64_t ld_src, n_out, n_in, ldw, ld_planes, plane_stride;
  int32_t src_is_f32, transpose;
} usf_pack_job;
int usf_pack_weights_f32(const usf_pack_job* jobs, int64_t n_jobs, int64_t max_rows, int64_t max_cols,
                         usf_stream_t stream);
/* The same for jobs with the transpose bit set (the data-gradient images W^T of every layer): same results, the source read
 * along its rows through a 32 x 32 LDS tile instead of one cache line per lane. */
int usf_pack_weights_t_f32(const usf_pack_job* jobs, int64_t n_jobs, int64_t max_rows, int64_t max_cols,
                           usf_stream_t stream);

/* out[o] = alpha * sum_k src[idx[o], k] * b[k] (0 where idx[o] < 0; idx NULL = identity), fp64 accumulation; out32 and/or
 * out64 receive the result.  Bias folding c = -(Minv b) and SequentialAffineTransform.bias (transforms.py:1471-1476). */
int usf_matvec_f64(const double* src, int64_t ld_src, int64_t K, const int32_t* idx, int64_t n_out, const double* b,
                   double alpha, float* out32, double* out64, usf_stream_t stream);

/*
 * ---- backward pass of the training step -----------------------------------------------
 * Flow.fit (flows.py:196-199) differentiates -log_prob(batch).mean(); these are the batch-sized pieces of that
 * backward pass (usf_train.hip).  Data gradients of the linear layers are usf_linear_f32 launches on the transposed
 * weight image (usf_pack_weight_f32 with transpose = 1).
 *
 * usf_wgrad_f32: G[n,k] = alpha * sum_m Y[m,n] * A[m,k] + beta * G[n,k]   -- the weight gradient of F.linear
 *   (Y = gradient at the layer's output [M,N], A = the layer's input [M,K]); exact-f32 MFMA, the batch is cut into
 *   row ranges whose partial products are summed in a fixed order (bitwise reproducible).  Y / A rows must be 16-byte
 *   aligned (ld % 4 == 0).  workspace: at least usf_wgrad_workspace_floats(M,N,K) floats.  mode 0: exact-f32 MFMA;
 *   mode 1: the bf16x3 split of DESIGN.md 3.1b (fp32-equivalent accuracy on the bf16 matrix cores; used from M >= 2048;
 *   from 8192 rows with enough output tiles to fill the chip, and while (M + 448) * max(ldy, lda) * 4 < 2^32, the
 *   loader-wave kernel: it reads Y / A with 16-byte loads up to the end of the row extent ((M-1) * ld + N resp. K
 *   floats from the base pointer) -- the same memory the contract above names).
 * usf_wgrad_variant: which kernel such a call launches -- 0 exact-f32, 1 bf16x3 (256 threads), 2 bf16x3 with loader
 *   waves (introspection for the parity tests).
 * usf_colsum_f32: out[n] = alpha * sum_m Y[m,n] + beta * out[n]           -- the bias gradient; workspace
 *   (ceil(M/256) + ceil(M/65536) + 2) * N floats (partials of the 256-row levels).
 */
int usf_wgrad_f32(const float* Y, int64_t ldy, const float* A, int64_t lda, int64_t M, int64_t N, int64_t K, float* G,
                  int64_t ldg, float alpha, float beta, int32_t mode, float* workspace, int64_t workspace_floats,
                  usf_stream_t stream);
int64_t usf_wgrad_workspace_floats(int64_t M, int64_t N, int64_t K);
int usf_wgrad_variant(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t lda, int32_t mode);
/* usf_wgrad_f32 and the layer's bias gradient in one pass (ABI 32): colsum_out[n] = cs_alpha * sum_m Y[m,n] + cs_beta *
 * colsum_out[n] (what usf_colsum_f32 computes), from the operand fragments the bf16x3 kernels hold anyway -- three more
 * MFMAs per fragment row against an operand of ones in the blocks of tile column 0; partial sums in a fixed order.  Only
 * where usf_wgrad_bias_ok says 1 (usf_wgrad_variant >= 1, i.e. mode 1 from 2048 rows, and K >= 64); the same workspace
 * as usf_wgrad_f32. */
int usf_wgrad_bias_f32(const float* Y, int64_t ldy, const float* A, int64_t lda, int64_t M, int64_t N, int64_t K, float* G,
                       int64_t ldg, float alpha, float beta, int32_t mode, float* colsum_out, float cs_alpha, float cs_beta,
                       float* workspace, int64_t workspace_floats, usf_stream_t stream);
int usf_wgrad_bias_ok(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t lda, int32_t mode);
int usf_colsum_f32(const float* Y, int64_t ldy, int64_t M, int64_t N, float* out, float alpha, float beta,
                   float* workspace, int64_t workspace_floats, usf_stream_t stream);

/*
 * The weight gradient from PRE-SPLIT operands (ABI 32).  usf_wgrad_f32's loader waves split every fp32 operand value
 * into its three bf16 planes again in each of the blocks that share its rows (seven times at 784 x 784); here Y and A
 * arrive as the planes the layer's own GEMMs already made of them (usf_linear_desc::A_planes_out: the forward GEMM
 * splits the layer input, the data-gradient GEMM the output gradient), or as usf_split_planes_f32 writes them:
 *   plane p of an operand at base + p * plane_stride elements, each [ceil32(M), ld] bf16 row-major, ld % 8 == 0,
 *   plane_stride % 8 == 0, plane_stride >= ceil32(M) * ld, 16-byte aligned base, rows [M, ceil32(M)) zero,
 *   the three planes of one operand below 4 GiB; x == p0 + p1 + p2 exactly (round-to-nearest residual split).
 * usf_wgrad_planes_f32: G[n,k] = alpha * sum_m Y[m, y_off + n] * A[m, a_off + k] + beta * G[n,k]  (y_off, a_off % 8 == 0)
 *   -- the same six products per value pair in the same order as usf_wgrad_f32 mode 1, one block per CU over row ranges
 *   of equal length, partial sums added in a fixed order (bitwise reproducible).  workspace: at least
 *   usf_wgrad_planes_workspace_floats(M, N, K) floats.  usf_wgrad_planes_ok: 1 where the kernel pays (the loader-wave
 *   kernel's cross-over: M >= 8192 and enough tiles), else 0 -- callers then keep usf_wgrad_f32.
 *   colsum_out (may be NULL; needs usf_wgrad_planes_colsum_ok = K >= 64): colsum_out[n] = cs_alpha * sum_m Y[m, y_off + n]
 *   + cs_beta * colsum_out[n] -- the layer's bias gradient (usf_colsum_f32) from the fragments the kernel holds anyway:
 *   three more MFMAs per fragment row against an operand of ones in the blocks of tile column 0.
 * usf_split_planes_f32: P[p][m][c] for m < ceil32(M), c < ldp: the planes of X[m, c] (zeros for m >= M or c >= N).
 * Replaces: the weight-gradient half of autograd's F.linear backward (flows.py:196-203, transforms.py:913-962,
 * networks.py:739-751) at training batches of thousands of rows.
 */
int usf_wgrad_planes_f32(const void* Y_planes, int64_t ldyp, int64_t y_plane_stride, int64_t y_off, const void* A_planes,
                         int64_t ldap, int64_t a_plane_stride, int64_t a_off, int64_t M, int64_t N, int64_t K, float* G,
                         int64_t ldg, float alpha, float beta, float* colsum_out, float cs_alpha, float cs_beta, float* workspace,
                         int64_t workspace_floats, usf_stream_t stream);
int64_t usf_wgrad_planes_workspace_floats(int64_t M, int64_t N, int64_t K);
int usf_wgrad_planes_ok(int64_t M, int64_t N, int64_t K);
int usf_wgrad_planes_colsum_ok(int64_t M, int64_t N, int64_t K);
int usf_split_planes_f32(const float* X, int64_t ldx, int64_t M, int64_t N, void* planes, int64_t ldp, int64_t plane_stride,
                         usf_stream_t stream);
/* The same weight gradient with both operands in the BLOCKED planes format of the planes pipeline (ABI 33) -- the buffers the
 * training forward's usf_gemm_planes_bf16x3 / usf_coupling_planes launches leave behind and the backward's launches write:
 *   G[n,k] = alpha * sum_m Y[m, 32 y_kb0 + n] * A[m, 32 a_kb0 + k] + beta * G[n,k],   n < N, k < K  (LOGICAL positions),
 * Y / A planes buffers of ceil(M/16) panels with y_nkb / a_nkb blocks per panel (USF_PLANES_BF16X3), N <= 32 (y_nkb - y_kb0),
 * K <= 32 (a_nkb - a_kb0).  Rows [M, 16 ceil(M/16)) of Y must hold zeros (they do in every buffer whose producer chain
 * starts at usf_pack_planes_f32 and has no bias); those of A must be finite.  Same kernel, schedule, order of products,
 * workspace (usf_wgrad_planes_workspace_floats) and colsum_out semantics as usf_wgrad_planes_f32; a buffer must stay below
 * 2 GiB.  Loader waves copy whole 1-KiB chunks; the MFMA waves' transposing reads un-do the slot order, so G comes out in
 * logical order. */
int usf_wgrad_blocked_f32(const void* Y_planes, int64_t y_nkb, int64_t y_kb0, const void* A_planes, int64_t a_nkb, int64_t a_kb0,
                          int64_t M, int64_t N, int64_t K, float* G, int64_t ldg, float alpha, float beta, float* colsum_out,
                          float cs_alpha, float cs_beta, float* workspace, int64_t workspace_floats, usf_stream_t stream);
/* The reduction of usf_wgrad_blocked_f32 queued (round 5; large-batch training: 129 weight gradients per step end with a reduction of
 * ~18 us each, one after the other although only the parameter update waits for them).  usf_wgrad_blocked_plan_f32 = the same call
 * that launches the multiply kernel only and fills *job (HOST memory): G / colsum_out stay unwritten and the workspace stays in use
 * until a usf_wgrad_reduce_jobs_f32 launch containing the job has run (jobs / block_job DEVICE arrays as for
 * usf_partial_sum_jobs_f32: job j owns the blocks [first_block, first_block + blocks), first_block set by the caller).  Same
 * additions in the same order as the undeferred call: same bits. */
typedef struct usf_wreduce_job {
  const float* part; float* out; const float* cs_part; float* cs_out;
  int64_t rows, cols, ldo;
  float alpha, beta, cs_alpha, cs_beta;
  int32_t first_block, blocks;
  unsigned char sched[64];                      /* the schedule's tile classes (opaque) */
} usf_wreduce_job;
int usf_wgrad_blocked_plan_f32(const void* Y_planes, int64_t y_nkb, int64_t y_kb0, const void* A_planes, int64_t a_nkb, int64_t a_kb0,
                          int64_t M, int64_t N, int64_t K, float* G, int64_t ldg, float alpha, float beta, float* colsum_out,
                          float cs_alpha, float cs_beta, float* workspace, int64_t workspace_floats, usf_wreduce_job* job, usf_stream_t stream);
int usf_wgrad_reduce_jobs_f32(const usf_wreduce_job* jobs, const int32_t* block_job, int64_t n_blocks, usf_stream_t stream);

/* Many small weight / bias gradients in ONE launch.  At the reference's training batch (32 rows, tests/explib/mnist.yaml:34)
 * Flow.fit's backward pass (flows.py:196-199) asks for one weight and one bias gradient per F.linear on the path -- some
 * hundreds of launches of a few microseconds whose dispatch, not their work, bounds the step.  `jobs` is a DEVICE array;
 * `block_job` a DEVICE array with the job index of every block of the launch (n_blocks entries; job j owns the blocks
 * first_block .. first_block + its own count - 1, in order):
 *   A != NULL: G[n,k] = alpha * sum_m Y[m,n] A[m,k] + beta * G[n,k], ceil(N/128) * ceil(K/128) blocks, exact-f32 MFMA over
 *              one row range -- bit-identical to usf_wgrad_f32 (mode 0) on the same operands for M <= 256; the alignment
 *              rules of usf_wgrad_f32 apply;
 *   A == NULL: G[n] = alpha * sum_m Y[m,n] + beta * G[n], ceil(N/64) blocks (fixed summation order: reproducible).
 * Meant for M <= 256; jobs of one launch must not write what another job of the same launch reads or writes. */
typedef struct usf_grad_job {
  const float* Y;
  const float* A;
  float* G;
  int64_t ldy, lda, ldg;
  int32_t M, N, K, first_block;
  float alpha, beta;
} usf_grad_job;
int usf_grad_jobs_f32(const usf_grad_job* jobs, const int32_t* block_job, int64_t n_blocks, usf_stream_t stream);

/*
 * SophiaG over all parameter tensors of a model in one launch (sophia.py:39-58 update_hessian, 151-199
 * _single_tensor_sophiag -- the optimiser Flow.fit defaults to, flows.py:116).  `chunks` is a DEVICE array; a chunk is
 * one block's share (any length; the host side cuts tensors into pieces of 16 384 elements) of one fp32 parameter
 * tensor p with its gradient g, momentum m (exp_avg) and Hessian estimate h.
 *   usf_sophiag_step_f32:    p *= decay (= 1 - lr * weight_decay);  m = m * beta1 + g * one_minus_beta1 (g negated with
 *                            maximize);  ratio = min(|m| / (rho_bs * h + 1e-15), 1) (rho_bs = rho * bs);
 *                            p += neg_lr * sign(m) * ratio
 *   usf_sophiag_hessian_f32: h = h * beta2 + one_minus_beta2 * g * g
 */
typedef struct usf_mt_chunk {
  float* p; const float* g; float* m; float* h;
  int32_t n; int32_t reserved;
} usf_mt_chunk;
int usf_sophiag_step_f32(const usf_mt_chunk* chunks, int64_t n_chunks, float decay, float beta1, float one_minus_beta1,
                         float rho_bs, float neg_lr, int32_t maximize, usf_stream_t stream);
int usf_sophiag_hessian_f32(const usf_mt_chunk* chunks, int64_t n_chunks, float beta2, float one_minus_beta2,
                            usf_stream_t stream);

/* (Leaky)ReLU backward from the saved layer OUTPUT h: d[m,j] *= (h[m,j] > 0 ? 1 : slope), slope >= 0
 * (ATen leaky_relu_backward on the pre-activation; sign(h) == sign(pre-activation)). networks.py:745-749 */
int usf_act_grad_f32(float* d, int64_t ldd, const float* h, int64_t ldh, int64_t M, int64_t H, int32_t act, float slope,
                     usf_stream_t stream);

/* Backward of usf_base_logprob_f32: g[m,d] = g_lp[m] * d/dz base_d(z[m,d]) for d < D, 0 for D <= d < ldg
 * (flows.py:245 through torch Laplace.log_prob / Normal.log_prob).  For the LPNORM* ids g_lp is the gradient at the
 * radius r[m] = ||z[m,:] - loc||_p (RadialDistribution.log_prob, distributions.py:501-505) and `scale` carries that
 * radius vector [M] (the forward kernel's output). */
int usf_base_logprob_grad_f32(const float* z, int64_t ldz, const float* g_lp, int64_t M, int64_t D, int32_t base,
                              const float* loc, const float* scale, float* g, int64_t ldg, usf_stream_t stream);

/*
 * ---- parameter maps of the image flows' affine blocks, batched over the blocks of a flow (usf_affine_prep.hip) -----------
 * n transforms of equal structure: an LUTransform (transforms.py:1271-1320: L = tril(L_raw, -1) + I, U = triu(U_raw),
 * M_lu = L U, M_lu^-1 = U^-1 L^-1, log|det| = sum log|U_jj|), optionally followed by a HouseholderTransform of nvs vectors
 * (transforms.py:795-809: H = w_0 prod_k (I - 2 v_k v_k^T / v_k.v_k)) inside a SequentialAffineTransform (:1457-1476):
 *     M = M_lu H,   Minv = H^T M_lu^-1,   b = bias H,   c = -Minv b   (nvs == 0: H = I; vk / w0 may be NULL)
 * (c: the shift of the backward direction, x = Minv (y - b) = Minv y + c.)
 * All operands fp32, contiguous: L_raw, U_raw, w0, M, Minv [n, C, C]; bias, b, c [n, C]; vk [n, nvs, C]; ladj [n];
 * save [n, 7, C, C] receives the factors the backward pass reads.  1 <= C <= 64, nvs <= 8.  One launch each.
 * usf_affine_prep_bwd_f32: from the forward pass's Minv and b and (dM, dMinv, db, dc, dladj) -- zeros where an output was
 * not used -- the gradients of the
 * parameters: dL_raw (strictly lower triangle, zeros elsewhere: the reference's gradient mask, transforms.py:1262-1268),
 * dU_raw (upper triangle), dbias [n, C], dvk [n, nvs, C].  What autograd derives from the reference's matrix() /
 * inverse_matrix() / bias() / log_abs_det_jacobian() chains, in two launches instead of some hundreds.
 */
int usf_affine_prep_f32(const float* L_raw, const float* U_raw, const float* bias, const float* vk, const float* w0, int64_t n,
                        int32_t C, int32_t nvs, float* M, float* Minv, float* b, float* c, float* ladj, float* save,
                        usf_stream_t stream);
int usf_affine_prep_bwd_f32(const float* save, const float* bias, const float* vk, const float* w0, const float* Minv,
                            const float* b, const float* dM, const float* dMinv, const float* db, const float* dc,
                            const float* dladj, int64_t n, int32_t C, int32_t nvs, float* dL_raw, float* dU_raw, float* dbias,
                            float* dvk, usf_stream_t stream);

/* Trainable Laplace / Normal base (the reference's distributions.Laplace / Normal modules, distributions.py:199-238, as a flow's
 * base distribution under Flow.fit, flows.py:196-203) -- ABI 33:
 *   d_loc_scale[d]     = sum_m g_lp[m] * d/dloc_d   base_d(z[m,d]; loc_d, scale_d)
 *   d_loc_scale[D + d] = sum_m g_lp[m] * d/dscale_d base_d(...)          (scale = the CONSTRAINED scale the density uses;
 * the caller applies softplus' for the modules' scale_unconstrained).  Laplace: sign(t)/b, |t|/b^2 - 1/b; Normal: t/s^2,
 * t^2/s^3 - 1/s (t = z - loc).  Row ranges of 256 summed in a fixed order (bit-reproducible).  workspace: at least
 * (ceil(M/256) + ceil(M/65536) + 4) * 2 D floats. */
int usf_base_param_grad_f32(const float* z, int64_t ldz, const float* g_lp, int64_t M, int64_t D, int32_t base, const float* loc,
                            const float* scale, float* d_loc_scale, float* workspace, int64_t workspace_floats, usf_stream_t stream);

/* Measurement aid (bench.py: roofline.sustained_peak): ONE launch of a register-only loop of the planes GEMM's matrix-core
 * instruction mix (v_mfma_f32_16x16x32_bf16, 10 x 2 accumulator tiles, six products per fp32-equivalent product; 512 threads,
 * two waves per SIMD; no LDS, no memory traffic in the loop) -- `iters` slabs of 120 MFMAs per wave on `blocks` blocks
 * (0: two per CU).  src1024: 1024 finite floats (device); sink: one float (device, never written); *flops_out (host, may be
 * NULL): the bf16 MFMA flops of the launch (fp32-equivalent: / 6).  The caller times the launch with events on `stream`. */
int usf_mfma_probe(const float* src1024, float* sink, int64_t iters, int64_t blocks, double* flops_out, usf_stream_t stream);
/* Measurement aid (bench.py: roofline.clock_mhz): while dev_buf2 (device memory, two 64-bit counters, zeroed by the caller) is
 * set, every block of usf_gemm_planes_bf16x3's kernel and of usf_mfma_probe adds its lifetime to it -- [0] in shader-clock cycles
 * (s_memtime), [1] in ticks of the constant 100 MHz counter (s_memrealtime): 100 MHz x [0] / [1] is the clock the matrix cores ran
 * at under that kernel (the nominal peaks assume 2400 MHz).  NULL: off (the default; the kernels then read no counter).
 * One process-wide setting (not per device, not synchronised with launches in flight): set it, launch, synchronise, clear it. */
int usf_set_clock_buffer(unsigned long long* dev_buf2);

/* Tuning knobs of the kernels' host code (A/B switches, cross-overs): named integers, preset on first use from the environment
 * variable USFLOWS_AMD_TUNE ("name=value,..."), changed at run time here.  usf_get_tuning(name, dflt): the value in force. */
int usf_set_tuning(const char* name, int64_t value);
int64_t usf_get_tuning(const char* name, int64_t dflt);

int usf_abi_version(void);
int usf_sizeof_desc(int32_t kind);      /* sizeof(usf_linear_desc|usf_coupling_desc|usf_op|usf_lu_prep_desc|usf_pack_job) for kind 1|2|0|3|4;
                                           usf_pack_planes_desc|usf_gemm_planes_desc|usf_coupling_planes_desc|usf_mt_chunk|usf_gated_norm_desc for 5|6|7|8|9,
                                           usf_call_desc|usf_grad_job for 10|11:
                                           binding self-check */
const char* usf_last_error(void);
const char* usf_build_info(void);       /* "gfx950 ..." */

#ifdef __cplusplus
}
#endif
#endif /* USFLOWS_HIP_H */
