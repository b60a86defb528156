// extern "C" surface of libusflows_hip.so (declared in include/usflows_hip.h).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "usf_common.h"

namespace usf {

static thread_local char g_err[512] = "";

// ---- the ONE table of tuning knobs (usf_common.h: usf::tuning) ------------------------------------------------------------
// Every A/B switch of the kernels' host code is a named integer read through tuning(name, default); the table is preset, on
// first use, from the environment variable USFLOWS_AMD_TUNE ("name=value,name=value": the same variable usflows_amd/config.py
// parses for the Python side's knobs; names nobody asks for are ignored) and changed at run time with usf_set_tuning.  The
// only getenv of the library lives here.  Not synchronised against concurrent usf_set_tuning calls (a tuning aid).
struct TuneEntry { char name[40]; long long value; };
static TuneEntry g_tune[96];
static int g_ntune = 0;
static bool g_tune_init = false;

static void tune_store(const char* name, size_t len, long long value) {
  if (len == 0 || len >= sizeof(g_tune[0].name)) return;
  for (int i = 0; i < g_ntune; ++i)
    if (strlen(g_tune[i].name) == len && strncmp(g_tune[i].name, name, len) == 0) { g_tune[i].value = value; return; }
  if (g_ntune < (int)(sizeof(g_tune) / sizeof(g_tune[0]))) {
    memcpy(g_tune[g_ntune].name, name, len);
    g_tune[g_ntune].name[len] = 0;
    g_tune[g_ntune].value = value;
    ++g_ntune;
  }
}
static void tune_init() {
  if (g_tune_init) return;
  g_tune_init = true;
  const char* e = getenv("USFLOWS_AMD_TUNE");
  while (e && *e) {
    const char* end = strchr(e, ',');
    const size_t n = end ? (size_t)(end - e) : strlen(e);
    const char* eq = (const char*)memchr(e, '=', n);
    if (eq) tune_store(e, (size_t)(eq - e), atoll(eq + 1));
    e = end ? end + 1 : nullptr;
  }
}
unsigned long long* g_clock_buffer = nullptr;
unsigned long long* clock_buffer() { return g_clock_buffer; }
long long tuning(const char* name, long long dflt) {
  tune_init();
  for (int i = 0; i < g_ntune; ++i)
    if (strcmp(g_tune[i].name, name) == 0) return g_tune[i].value;
  return dflt;
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int linear_dispatch(const usf_linear_desc* d, hipStream_t stream);
int linear_variant(const usf_linear_desc* d);
int coupling_dispatch(const usf_coupling_desc* d, hipStream_t stream);
int coupling_variant(const usf_coupling_desc* d);
int coupling_max_width();
int coupling_padded_width(int h);
int lu_grad_finish(const double* dL, const double* dU, const double* TL, const double* TU, const double* c, const double* tri,
                   int64_t n, int64_t D, float* oL, float* oU, hipStream_t stream);
int base_tables(int32_t base, const float* loc, const float* scale, int64_t D, float* tab, int64_t stride, hipStream_t stream);
int base_logprob(const float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc,
                 const float* scale, float logdet_const, const double* logdet_dev, float* logp, double* sum_out,
                 hipStream_t stream);
int base_sample(float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc, const float* scale,
                uint64_t seed, uint64_t offset, int64_t row_offset, hipStream_t stream);
int radial_sample(float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc, const float* r,
                  uint64_t seed, uint64_t offset, int64_t row_offset, hipStream_t stream);
int variates_from_bits(const uint32_t* bits, int64_t n, float* u, float* laplace, float* exponential, hipStream_t stream);
int scale(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t D, const float* s, int32_t divide,
          hipStream_t stream);
int affine_coupling_apply(float* z, int64_t ldz, const float* t, int64_t ldt, const float* s, int64_t lds_, int64_t M,
                          int64_t n, float bound, int32_t inverse, float* logdet, hipStream_t stream);
int channel_affine(const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* W, const float* pre_sub,
                   const float* bias, hipStream_t stream);
int gather_cols(const float* src, int64_t lds_, float* dst, int64_t ldd, int64_t M, int64_t n, const int32_t* idx,
                hipStream_t stream);

int pack_planes(const usf_pack_planes_desc* d, hipStream_t stream);
int gemm_planes(const usf_gemm_planes_desc* d, hipStream_t stream);
int gemm_planes_variant(const usf_gemm_planes_desc* d);
int coupling_planes(const usf_coupling_planes_desc* d, hipStream_t stream);
int lu_prepare(const usf_lu_prep_desc* d, hipStream_t stream);
int gemm_f64(const double* A, int64_t lda, int64_t sA, int transA, const double* B, int64_t ldb, int64_t sB, int transB,
             double* C, int64_t ldc, int64_t sC, int64_t M, int64_t N, int64_t K, int64_t batch, double alpha,
             double beta, int32_t tri, hipStream_t stream);
int householder(const float* w0, const float* vk, int64_t nvs, int64_t D, double* out, hipStream_t stream);
int pack_weight(const void* src, int32_t src_is_f32, int64_t lds_, int32_t transpose, const int32_t* out_idx, int64_t n_out,
                const int32_t* in_idx, int64_t n_in, float* W, int64_t ldw, void* planes, int64_t ldp,
                int64_t plane_stride, hipStream_t stream);
int pack_jobs(const usf_pack_job* jobs, int64_t n_jobs, int64_t max_rows, int64_t max_cols, hipStream_t stream);
int pack_jobs_t(const usf_pack_job* jobs, int64_t n_jobs, int64_t max_rows, int64_t max_cols, hipStream_t stream);
int matvec_rows(const double* src, int64_t lds_, int64_t K, const int32_t* idx, int64_t n_out, const double* b,
                double alpha, float* out32, double* out64, hipStream_t stream);

int wgrad_workspace_floats(int64_t M, int64_t N, int64_t K, int64_t* out);
int wgrad_variant(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t lda, int32_t mode);
int64_t conv2d_weight_elems(int64_t cin, int64_t cout, int64_t ks);
int conv2d_same_fits(int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks);
int conv2d_same_res(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                    const void* wplanes, const float* bias, const float* in_mul, int32_t in_act, float in_slope,
                    const float* res_x, const float* res_mul, float res_sign, hipStream_t stream);
int conv2d_same(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                const void* wplanes, const float* bias, const float* in_mul, int32_t in_act, float in_slope,
                int32_t out_act, float out_slope, const float* gate_x, int64_t gate_channels, hipStream_t stream);
int conv2d_same_gate(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                     const void* wplanes, const float* gate_h, float gate_slope, const float* gate_mul, const float* gate_add,
                     hipStream_t stream);
int layernorm_channels(const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* gamma, const float* beta,
                       float eps, int32_t act, float slope, hipStream_t stream);
int gated_residual(const float* x, const float* vg, float* y, int64_t B, int64_t CP, hipStream_t stream);
int gated_norm_rows(const usf_gated_norm_desc* d, hipStream_t stream);
int gated_norm_rows_bwd(const usf_gated_norm_bwd_desc* d, hipStream_t stream);
int pointwise_conv_supported(int64_t cin, int64_t cout, int32_t gated);
int pointwise_conv(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t P, const float* W, const float* bias,
                   int32_t in_act, float in_slope, int32_t out_act, float out_slope, const float* gate_x,
                   const float* ln_gamma, const float* ln_beta, float ln_eps, hipStream_t stream);
int masked_residual(const float* x, const float* t, const float* om, float sign, float* y, int64_t B, int64_t CP,
                    hipStream_t stream);
int64_t conv_wgrad_workspace(int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks);
int conv_wgrad(const float* x, const float* dy, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
               const float* in_mul, const float* pre_sub, int32_t in_act, float in_slope, float* dW, float* db, float* workspace,
               int64_t workspace_floats, usf_psum_job* job, usf_wgrad_job* wjob, hipStream_t stream);
int conv_wgrad_jobs(const usf_wgrad_job* jobs, const int32_t* block_job, int64_t n_blocks, int32_t CIT, int32_t COT, int32_t T,
                    int32_t lds_bytes, hipStream_t stream);
int partial_sum_jobs(const usf_psum_job* jobs, const int32_t* block_job, int64_t n_blocks, hipStream_t stream);
int conv2d_weight_planes_batch(const usf_wplanes_job* jobs, const int32_t* block_job, int64_t n_blocks, void* planes_base,
                               hipStream_t stream);
int gated_tail_supported(int64_t C);
int64_t gated_tail_workspace(int64_t B, int64_t C, int64_t P);
int gated_tail_fwd(const float* h, const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* W, const float* bias,
                   int32_t in_act, float in_slope, int32_t post_act, float post_slope, const float* gamma, const float* beta,
                   float eps, hipStream_t stream);
int gated_tail_bwd(const float* h, const float* x, const float* dy, float* dx, float* dh, float* dvg, int64_t B, int64_t C, int64_t P,
                   const float* W, const float* bias, int32_t in_act, float in_slope, int32_t post_act, float post_slope,
                   const float* gamma, const float* beta, float eps, float* dparams, float* workspace, int64_t workspace_floats,
                   usf_psum_job* job, hipStream_t stream);
int64_t layernorm_channels_bwd_workspace(int64_t B, int64_t C, int64_t P);
int layernorm_channels_bwd(const float* x, const float* dy, float* dx, int64_t B, int64_t C, int64_t P, const float* gamma, float eps,
                           int32_t act, float slope, float* dgamma, float* dbeta, float* workspace, int64_t workspace_floats,
                           hipStream_t stream);
int gated_residual_bwd(const float* dy, const float* vg, float* dvg, int64_t B, int64_t CP, hipStream_t stream);
int conv2d_weight_planes(const float* w, void* planes, int64_t cin, int64_t cout, int64_t ks, int32_t transposed, hipStream_t stream);
int sophiag_step(const usf_mt_chunk* chunks, int64_t n_chunks, float decay, float beta1, float one_minus_beta1, float rho_bs,
                 float neg_lr, int32_t maximize, hipStream_t stream);
int sophiag_hessian(const usf_mt_chunk* chunks, int64_t n_chunks, float beta2, float one_minus_beta2, hipStream_t stream);
int wgrad(const float* Y, int64_t ldy, const float* A, int64_t lda, int64_t M, int64_t N, int64_t K, float* G,
          int64_t ldg, float alpha, float beta, int32_t mode, float* workspace, int64_t workspace_floats,
          hipStream_t stream, float* colsum_out = nullptr, float cs_alpha = 0.f, float cs_beta = 0.f);
int wgrad_bias_ok(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t lda, int32_t mode);
int affine_prep(const float* Lr, const float* Ur, const float* bias, const float* vk, const float* w0, int64_t n, int32_t C,
                int32_t nvs, float* M, float* Minv, float* b, float* cvec, float* ladj, float* save, hipStream_t stream);
int affine_prep_bwd(const float* save, const float* bias, const float* vk, const float* w0, const float* Minv, const float* b,
                    const float* dM, const float* dMinv, const float* db, const float* dc, const float* dladj, int64_t n,
                    int32_t C, int32_t nvs, float* dLr, float* dUr, float* dbias, float* dvk, hipStream_t stream);
int grad_jobs(const usf_grad_job* jobs, const int32_t* block_job, int64_t n_blocks, hipStream_t stream);
int split_planes(const float* X, int64_t ldx, int64_t M, int64_t N, void* P, int64_t ldp, int64_t plane_stride, hipStream_t stream);
int wgrad_planes_ok(int64_t M, int64_t N, int64_t K);
int64_t wgrad_planes_workspace_floats(int64_t M, int64_t N, int64_t K);
int wgrad_planes(const void* Yp, int64_t ldyp, int64_t ystride, int64_t y_off, const void* Ap, int64_t ldap, int64_t astride,
                 int64_t a_off, int64_t M, int64_t N, int64_t K, float* G, int64_t ldg, float alpha, float beta, float* colsum_out,
                 float cs_alpha, float cs_beta, float* workspace, int64_t workspace_floats, hipStream_t stream);
int wgrad_planes_colsum_ok(int64_t M, int64_t N, int64_t K);
int wgrad_blocked(const void* Yp, int64_t y_nkb, int64_t y_kb0, const void* Ap, int64_t a_nkb, int64_t a_kb0, int64_t M, int64_t N,
                  int64_t K, float* G, int64_t ldg, float alpha, float beta, float* colsum_out, float cs_alpha, float cs_beta,
                  float* workspace, int64_t workspace_floats, usf_wreduce_job* job, hipStream_t stream);
int wgrad_reduce_jobs(const usf_wreduce_job* jobs, const int32_t* block_job, int64_t n_blocks, hipStream_t stream);
int base_param_grad(const float* z, int64_t ldz, const float* g_lp, int64_t M, int64_t D, int32_t base, const float* loc,
                    const float* scale, float* d_loc_scale, float* workspace, int64_t workspace_floats, hipStream_t stream);
int mfma_probe(const float* src1024, float* sink, int64_t iters, int64_t blocks, double* flops_out, hipStream_t stream);
int colsum(const float* Y, int64_t ldy, int64_t M, int64_t N, float* out, float alpha, float beta, float* workspace,
           int64_t workspace_floats, hipStream_t stream);
int act_grad(float* d, int64_t ldd, const float* h, int64_t ldh, int64_t M, int64_t H, int32_t act, float slope,
             hipStream_t stream);
int base_grad(const float* z, int64_t ldz, const float* g_lp, int64_t M, int64_t D, int32_t base, const float* loc,
              const float* scale, float* g, int64_t ldg, hipStream_t stream);
int radial_logprob(const float* z, int64_t ldz, int64_t M, int64_t D, int32_t p_id, const float* loc, int32_t norm, int32_t K,
                   const float* par_a, const float* par_b, const float* logits, double logdv_const, float logdet_const,
                   const double* logdet_dev, float* logp, float* r_out, double* sum_out, hipStream_t stream);
int64_t radial_grad_workspace(int64_t M, int64_t D);
int radial_grad(const float* z, int64_t ldz, const float* r, const float* g_lp, int64_t M, int64_t D, int32_t p_id,
                const float* loc, int32_t norm, int32_t K, const float* par_a, const float* par_b, const float* logits, float* g,
                int64_t ldg, float* d_loc, float* d_a, float* d_b, float* d_logits, void* workspace, int64_t workspace_bytes,
                hipStream_t stream);

}  // namespace usf

extern "C" {

int usf_abi_version(void) { return USF_ABI_VERSION; }
int usf_set_tuning(const char* name, int64_t value) {
  if (!name || !*name || strlen(name) >= sizeof(usf::g_tune[0].name)) { usf::set_error("usf_set_tuning: bad name"); return -1; }
  usf::tune_init();
  usf::tune_store(name, strlen(name), (long long)value);
  return 0;
}
int64_t usf_get_tuning(const char* name, int64_t dflt) { return name ? (int64_t)usf::tuning(name, (long long)dflt) : dflt; }
int usf_sizeof_desc(int32_t kind) {
  switch (kind) {
    case USF_OP_LINEAR: return (int)sizeof(usf_linear_desc);
    case USF_OP_COUPLING: return (int)sizeof(usf_coupling_desc);
    case 0: return (int)sizeof(usf_op);
    case 3: return (int)sizeof(usf_lu_prep_desc);
    case 4: return (int)sizeof(usf_pack_job);
    case USF_OP_PACK_PLANES: return (int)sizeof(usf_pack_planes_desc);
    case USF_OP_GEMM_PLANES: return (int)sizeof(usf_gemm_planes_desc);
    case USF_OP_COUPLING_PLANES: return (int)sizeof(usf_coupling_planes_desc);
    case 8: return (int)sizeof(usf_mt_chunk);
    case USF_OP_GATED_NORM: return (int)sizeof(usf_gated_norm_desc);
    case USF_OP_CALL: return (int)sizeof(usf_call_desc);
    case 11: return (int)sizeof(usf_grad_job);
    case 12: return (int)sizeof(usf_psum_job);
    default: return -1;
  }
}
const char* usf_last_error(void) { return usf::g_err; }
const char* usf_build_info(void) { return "libusflows_hip gfx950 f32-mfma (" __DATE__ " " __TIME__ ")"; }

int usf_linear_f32(const usf_linear_desc* d, usf_stream_t stream) {
  return usf::linear_dispatch(d, (hipStream_t)stream);
}

int usf_linear_variant(const usf_linear_desc* d) { return usf::linear_variant(d); }

int usf_coupling_additive_f32(const usf_coupling_desc* d, usf_stream_t stream) {
  return usf::coupling_dispatch(d, (hipStream_t)stream);
}

int usf_pack_planes_f32(const usf_pack_planes_desc* d, usf_stream_t stream) { return usf::pack_planes(d, (hipStream_t)stream); }
int usf_gemm_planes_bf16x3(const usf_gemm_planes_desc* d, usf_stream_t stream) { return usf::gemm_planes(d, (hipStream_t)stream); }

int usf_coupling_planes(const usf_coupling_planes_desc* d, usf_stream_t stream) { return usf::coupling_planes(d, (hipStream_t)stream); }
int usf_gemm_planes_variant(const usf_gemm_planes_desc* d) { return usf::gemm_planes_variant(d); }
int usf_coupling_variant(const usf_coupling_desc* d) { return usf::coupling_variant(d); }
int usf_coupling_max_width(void) { return usf::coupling_max_width(); }
int usf_coupling_padded_width(int h) { return usf::coupling_padded_width(h); }

int usf_base_tables_f32(int32_t base, const float* loc, const float* scale, int64_t D, float* tab, int64_t stride, usf_stream_t stream) {
  return usf::base_tables(base, loc, scale, D, tab, stride, (hipStream_t)stream);
}
int usf_base_logprob_f32(const float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc,
                         const float* scale, float logdet_const, const double* logdet_dev, float* logp, double* sum_out,
                         usf_stream_t stream) {
  return usf::base_logprob(z, ldz, M, D, base, loc, scale, logdet_const, logdet_dev, logp, sum_out, (hipStream_t)stream);
}

int usf_base_sample_f32(float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc,
                        const float* scale, uint64_t seed, uint64_t offset, int64_t row_offset, usf_stream_t stream) {
  return usf::base_sample(z, ldz, M, D, base, loc, scale, seed, offset, row_offset, (hipStream_t)stream);
}

int usf_radial_sample_f32(float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc, const float* r,
                          uint64_t seed, uint64_t offset, int64_t row_offset, usf_stream_t stream) {
  return usf::radial_sample(z, ldz, M, D, base, loc, r, seed, offset, row_offset, (hipStream_t)stream);
}

int usf_variates_from_bits_f32(const uint32_t* bits, int64_t n, float* u, float* laplace, float* exponential,
                               usf_stream_t stream) {
  return usf::variates_from_bits(bits, n, u, laplace, exponential, (hipStream_t)stream);
}

int usf_scale_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t D, const float* s,
                  int32_t divide, usf_stream_t stream) {
  return usf::scale(x, ldx, y, ldy, M, D, s, divide, (hipStream_t)stream);
}

int usf_affine_coupling_apply_f32(float* z, int64_t ldz, const float* t, int64_t ldt, const float* s, int64_t lds, int64_t M,
                                  int64_t n, float bound, int32_t inverse, float* logdet, usf_stream_t stream) {
  return usf::affine_coupling_apply(z, ldz, t, ldt, s, lds, M, n, bound, inverse, logdet, (hipStream_t)stream);
}

int usf_channel_affine_f32(const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* W,
                           const float* pre_sub, const float* bias, usf_stream_t stream) {
  return usf::channel_affine(x, y, B, C, P, W, pre_sub, bias, (hipStream_t)stream);
}

int usf_gather_cols_f32(const float* src, int64_t lds_, float* dst, int64_t ldd, int64_t M, int64_t n,
                        const int32_t* idx, usf_stream_t stream) {
  return usf::gather_cols(src, lds_, dst, ldd, M, n, idx, (hipStream_t)stream);
}

int usf_lu_prepare_f64(const usf_lu_prep_desc* d, usf_stream_t stream) { return usf::lu_prepare(d, (hipStream_t)stream); }

int usf_gemm_f64(const double* A, int64_t lda, int64_t strideA, int32_t transA, const double* B, int64_t ldb,
                 int64_t strideB, int32_t transB, double* C, int64_t ldc, int64_t strideC, int64_t M, int64_t N,
                 int64_t K, int64_t batch, double alpha, double beta, int32_t tri, usf_stream_t stream) {
  return usf::gemm_f64(A, lda, strideA, transA, B, ldb, strideB, transB, C, ldc, strideC, M, N, K, batch, alpha, beta,
                       tri, (hipStream_t)stream);
}

int usf_lu_grad_finish_f64(const double* dL, const double* dU, const double* TL, const double* TU, const double* c,
                           const double* tri, int64_t n, int64_t D, float* dL_out, float* dU_out, usf_stream_t stream) {
  return usf::lu_grad_finish(dL, dU, TL, TU, c, tri, n, D, dL_out, dU_out, (hipStream_t)stream);
}
int usf_householder_f64(const float* w_0, const float* vk, int64_t nvs, int64_t D, double* out, usf_stream_t stream) {
  return usf::householder(w_0, vk, nvs, D, out, (hipStream_t)stream);
}

int usf_pack_weight_f32(const void* src, int32_t src_is_f32, int64_t ld_src, int32_t transpose, const int32_t* out_idx, int64_t n_out,
                        const int32_t* in_idx, int64_t n_in, float* W, int64_t ldw, void* planes, int64_t ld_planes,
                        int64_t plane_stride, usf_stream_t stream) {
  return usf::pack_weight(src, src_is_f32, ld_src, transpose, out_idx, n_out, in_idx, n_in, W, ldw, planes, ld_planes, plane_stride,
                          (hipStream_t)stream);
}

int usf_pack_weights_f32(const usf_pack_job* jobs, int64_t n_jobs, int64_t max_rows, int64_t max_cols,
                         usf_stream_t stream) {
  return usf::pack_jobs(jobs, n_jobs, max_rows, max_cols, (hipStream_t)stream);
}
int usf_pack_weights_t_f32(const usf_pack_job* jobs, int64_t n_jobs, int64_t max_rows, int64_t max_cols, usf_stream_t stream) {
  return usf::pack_jobs_t(jobs, n_jobs, max_rows, max_cols, (hipStream_t)stream);
}

int usf_matvec_f64(const double* src, int64_t ld_src, int64_t K, const int32_t* idx, int64_t n_out, const double* b,
                   double alpha, float* out32, double* out64, usf_stream_t stream) {
  return usf::matvec_rows(src, ld_src, K, idx, n_out, b, alpha, out32, out64, (hipStream_t)stream);
}

int usf_wgrad_f32(const float* Y, int64_t ldy, const float* A, int64_t lda, int64_t M, int64_t N, int64_t K, float* G,
                  int64_t ldg, float alpha, float beta, int32_t mode, float* workspace, int64_t workspace_floats,
                  usf_stream_t stream) {
  return usf::wgrad(Y, ldy, A, lda, M, N, K, G, ldg, alpha, beta, mode, workspace, workspace_floats, (hipStream_t)stream);
}
int usf_sophiag_step_f32(const usf_mt_chunk* chunks, int64_t n_chunks, float decay, float beta1, float one_minus_beta1,
                         float rho_bs, float neg_lr, int32_t maximize, usf_stream_t stream) {
  return usf::sophiag_step(chunks, n_chunks, decay, beta1, one_minus_beta1, rho_bs, neg_lr, maximize, (hipStream_t)stream);
}
int usf_sophiag_hessian_f32(const usf_mt_chunk* chunks, int64_t n_chunks, float beta2, float one_minus_beta2,
                            usf_stream_t stream) {
  return usf::sophiag_hessian(chunks, n_chunks, beta2, one_minus_beta2, (hipStream_t)stream);
}
int usf_layernorm_channels_f32(const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* gamma,
                               const float* beta, float eps, int32_t act, float slope, usf_stream_t stream) {
  return usf::layernorm_channels(x, y, B, C, P, gamma, beta, eps, act, slope, (hipStream_t)stream);
}
int usf_gated_residual_f32(const float* x, const float* vg, float* y, int64_t B, int64_t CP, usf_stream_t stream) {
  return usf::gated_residual(x, vg, y, B, CP, (hipStream_t)stream);
}
int usf_pointwise_conv_supported(int64_t cin, int64_t cout, int32_t gated) { return usf::pointwise_conv_supported(cin, cout, gated); }
int usf_pointwise_conv_f32(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t P, const float* W,
                           const float* bias, int32_t in_act, float in_slope, int32_t out_act, float out_slope,
                           const float* gate_x, const float* ln_gamma, const float* ln_beta, float ln_eps, usf_stream_t stream) {
  return usf::pointwise_conv(x, y, B, cin, cout, P, W, bias, in_act, in_slope, out_act, out_slope, gate_x, ln_gamma, ln_beta,
                             ln_eps, (hipStream_t)stream);
}
int usf_gated_norm_rows_f32(const usf_gated_norm_desc* d, usf_stream_t stream) {
  return usf::gated_norm_rows(d, (hipStream_t)stream);
}
int usf_gated_norm_rows_bwd_f32(const usf_gated_norm_bwd_desc* d, usf_stream_t stream) {
  return usf::gated_norm_rows_bwd(d, (hipStream_t)stream);
}
int usf_masked_residual_f32(const float* x, const float* t, const float* one_minus_mask, float sign, float* y, int64_t B,
                            int64_t CP, usf_stream_t stream) {
  return usf::masked_residual(x, t, one_minus_mask, sign, y, B, CP, (hipStream_t)stream);
}
int usf_conv2d_same_gate_f32(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                             const void* w_planes, const float* gate_h, float gate_slope, const float* gate_mul, const float* gate_add,
                             usf_stream_t stream) {
  return usf::conv2d_same_gate(x, y, B, cin, cout, H, W, ks, w_planes, gate_h, gate_slope, gate_mul, gate_add, (hipStream_t)stream);
}
int64_t usf_conv_wgrad_workspace(int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks) {
  return usf::conv_wgrad_workspace(B, cin, cout, H, W, ks);
}
int usf_conv_wgrad_f32(const float* x, const float* dy, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                       const float* in_mul, const float* pre_sub, int32_t in_act, float in_slope, float* dW, float* db,
                       float* workspace, int64_t workspace_floats, usf_stream_t stream) {
  return usf::conv_wgrad(x, dy, B, cin, cout, H, W, ks, in_mul, pre_sub, in_act, in_slope, dW, db, workspace, workspace_floats,
                         nullptr, nullptr, (hipStream_t)stream);
}
int usf_conv_wgrad_deferred_f32(const float* x, const float* dy, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                                const float* in_mul, const float* pre_sub, int32_t in_act, float in_slope, float* dW, float* db,
                                float* workspace, int64_t workspace_floats, usf_psum_job* job, usf_stream_t stream) {
  if (!job) { usf::set_error("usf_conv_wgrad_deferred_f32: job is NULL"); return -1; }
  return usf::conv_wgrad(x, dy, B, cin, cout, H, W, ks, in_mul, pre_sub, in_act, in_slope, dW, db, workspace, workspace_floats, job,
                         nullptr, (hipStream_t)stream);
}
int usf_conv_wgrad_plan_f32(const float* x, const float* dy, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                            const float* in_mul, const float* pre_sub, int32_t in_act, float in_slope, float* dW, float* db,
                            float* workspace, int64_t workspace_floats, usf_psum_job* job, usf_wgrad_job* wjob, usf_stream_t stream) {
  if (!job || !wjob) { usf::set_error("usf_conv_wgrad_plan_f32: job / wjob is NULL"); return -1; }
  return usf::conv_wgrad(x, dy, B, cin, cout, H, W, ks, in_mul, pre_sub, in_act, in_slope, dW, db, workspace, workspace_floats, job,
                         wjob, (hipStream_t)stream);
}
int usf_conv_wgrad_jobs_f32(const usf_wgrad_job* jobs, const int32_t* block_job, int64_t n_blocks, int32_t CIT, int32_t COT, int32_t T,
                            int32_t lds_bytes, usf_stream_t stream) {
  return usf::conv_wgrad_jobs(jobs, block_job, n_blocks, CIT, COT, T, lds_bytes, (hipStream_t)stream);
}
int usf_partial_sum_jobs_f32(const usf_psum_job* jobs, const int32_t* block_job, int64_t n_blocks, usf_stream_t stream) {
  return usf::partial_sum_jobs(jobs, block_job, n_blocks, (hipStream_t)stream);
}
int64_t usf_layernorm_channels_bwd_workspace(int64_t B, int64_t C, int64_t P) { return usf::layernorm_channels_bwd_workspace(B, C, P); }
int usf_layernorm_channels_bwd_f32(const float* x, const float* dy, float* dx, int64_t B, int64_t C, int64_t P, const float* gamma,
                                   float eps, int32_t act, float slope, float* dgamma_dbeta, float* workspace,
                                   int64_t workspace_floats, usf_stream_t stream) {
  return usf::layernorm_channels_bwd(x, dy, dx, B, C, P, gamma, eps, act, slope, dgamma_dbeta, dgamma_dbeta ? dgamma_dbeta + C : nullptr,
                                     workspace, workspace_floats, (hipStream_t)stream);
}
int usf_conv2d_weight_planes_f32(const float* w, void* planes, int64_t cin, int64_t cout, int64_t ks, int32_t transposed,
                                 usf_stream_t stream) {
  return usf::conv2d_weight_planes(w, planes, cin, cout, ks, transposed, (hipStream_t)stream);
}
int usf_gated_residual_bwd_f32(const float* dy, const float* vg, float* dvg, int64_t B, int64_t CP, usf_stream_t stream) {
  return usf::gated_residual_bwd(dy, vg, dvg, B, CP, (hipStream_t)stream);
}
int usf_conv2d_weight_planes_batch_f32(const usf_wplanes_job* jobs, const int32_t* block_job, int64_t n_blocks, void* planes_base,
                                       usf_stream_t stream) {
  return usf::conv2d_weight_planes_batch(jobs, block_job, n_blocks, planes_base, (hipStream_t)stream);
}
int usf_gated_tail_supported(int64_t C) { return usf::gated_tail_supported(C); }
int64_t usf_gated_tail_workspace(int64_t B, int64_t C, int64_t P) { return usf::gated_tail_workspace(B, C, P); }
int usf_gated_tail_f32(const float* h, const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* W, const float* bias,
                       int32_t in_act, float in_slope, int32_t post_act, float post_slope, const float* ln_gamma,
                       const float* ln_beta, float ln_eps, usf_stream_t stream) {
  return usf::gated_tail_fwd(h, x, y, B, C, P, W, bias, in_act, in_slope, post_act, post_slope, ln_gamma, ln_beta, ln_eps,
                             (hipStream_t)stream);
}
int usf_gated_tail_bwd_f32(const float* h, const float* x, const float* dy, float* dx, float* dh, float* dvg, int64_t B, int64_t C,
                           int64_t P, const float* W, const float* bias, int32_t in_act, float in_slope, int32_t post_act,
                           float post_slope, const float* ln_gamma, const float* ln_beta, float ln_eps, float* dparams,
                           float* workspace, int64_t workspace_floats, usf_psum_job* job, usf_stream_t stream) {
  return usf::gated_tail_bwd(h, x, dy, dx, dh, dvg, B, C, P, W, bias, in_act, in_slope, post_act, post_slope, ln_gamma, ln_beta,
                             ln_eps, dparams, workspace, workspace_floats, job, (hipStream_t)stream);
}
int64_t usf_conv2d_weight_elems(int64_t cin, int64_t cout, int64_t ks) { return usf::conv2d_weight_elems(cin, cout, ks); }
int usf_conv2d_same_fits(int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks) { return usf::conv2d_same_fits(cin, cout, H, W, ks); }
int usf_conv2d_same_f32(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                        const void* w_planes, const float* bias, const float* in_mul, int32_t in_act, float in_slope,
                        int32_t out_act, float out_slope, const float* gate_x, int64_t gate_channels, usf_stream_t stream) {
  return usf::conv2d_same(x, y, B, cin, cout, H, W, ks, w_planes, bias, in_mul, in_act, in_slope, out_act, out_slope,
                          gate_x, gate_channels, (hipStream_t)stream);
}
int usf_conv2d_same_res_f32(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                            const void* w_planes, const float* bias, const float* in_mul, int32_t in_act, float in_slope,
                            const float* res_x, const float* res_mul, float res_sign, usf_stream_t stream) {
  return usf::conv2d_same_res(x, y, B, cin, cout, H, W, ks, w_planes, bias, in_mul, in_act, in_slope, res_x, res_mul, res_sign,
                              (hipStream_t)stream);
}
int usf_wgrad_bias_f32(const float* Y, int64_t ldy, const float* A, int64_t lda, int64_t M, int64_t N, int64_t K, float* G,
                       int64_t ldg, float alpha, float beta, int32_t mode, float* colsum_out, float cs_alpha, float cs_beta,
                       float* workspace, int64_t workspace_floats, usf_stream_t stream) {
  if (!colsum_out) { usf::set_error("usf_wgrad_bias_f32: colsum_out is NULL (usf_wgrad_f32 is the call without it)"); return -1; }
  return usf::wgrad(Y, ldy, A, lda, M, N, K, G, ldg, alpha, beta, mode, workspace, workspace_floats, (hipStream_t)stream, colsum_out,
                    cs_alpha, cs_beta);
}
int usf_wgrad_bias_ok(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t lda, int32_t mode) {
  return usf::wgrad_bias_ok(M, N, K, ldy, lda, mode);
}
int usf_wgrad_variant(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t lda, int32_t mode) {
  return usf::wgrad_variant(M, N, K, ldy, lda, mode);
}
int64_t usf_wgrad_workspace_floats(int64_t M, int64_t N, int64_t K) {
  int64_t out = 0;
  return usf::wgrad_workspace_floats(M, N, K, &out) == 0 ? out : -1;
}
int usf_wgrad_planes_f32(const void* Y_planes, int64_t ldyp, int64_t y_plane_stride, int64_t y_off, const void* A_planes,
                         int64_t ldap, int64_t a_plane_stride, int64_t a_off, int64_t M, int64_t N, int64_t K, float* G,
                         int64_t ldg, float alpha, float beta, float* colsum_out, float cs_alpha, float cs_beta, float* workspace,
                         int64_t workspace_floats, usf_stream_t stream) {
  return usf::wgrad_planes(Y_planes, ldyp, y_plane_stride, y_off, A_planes, ldap, a_plane_stride, a_off, M, N, K, G, ldg, alpha, beta,
                           colsum_out, cs_alpha, cs_beta, workspace, workspace_floats, (hipStream_t)stream);
}
int usf_wgrad_blocked_f32(const void* Y_planes, int64_t y_nkb, int64_t y_kb0, const void* A_planes, int64_t a_nkb, int64_t a_kb0,
                          int64_t M, int64_t N, int64_t K, float* G, int64_t ldg, float alpha, float beta, float* colsum_out,
                          float cs_alpha, float cs_beta, float* workspace, int64_t workspace_floats, usf_stream_t stream) {
  return usf::wgrad_blocked(Y_planes, y_nkb, y_kb0, A_planes, a_nkb, a_kb0, M, N, K, G, ldg, alpha, beta, colsum_out, cs_alpha,
                            cs_beta, workspace, workspace_floats, nullptr, (hipStream_t)stream);
}
int usf_wgrad_blocked_plan_f32(const void* Y_planes, int64_t y_nkb, int64_t y_kb0, const void* A_planes, int64_t a_nkb, int64_t a_kb0,
                               int64_t M, int64_t N, int64_t K, float* G, int64_t ldg, float alpha, float beta, float* colsum_out,
                               float cs_alpha, float cs_beta, float* workspace, int64_t workspace_floats, usf_wreduce_job* job,
                               usf_stream_t stream) {
  if (!job) { usf::set_error("usf_wgrad_blocked_plan_f32: job is NULL"); return -1; }
  return usf::wgrad_blocked(Y_planes, y_nkb, y_kb0, A_planes, a_nkb, a_kb0, M, N, K, G, ldg, alpha, beta, colsum_out, cs_alpha,
                            cs_beta, workspace, workspace_floats, job, (hipStream_t)stream);
}
int usf_wgrad_reduce_jobs_f32(const usf_wreduce_job* jobs, const int32_t* block_job, int64_t n_blocks, usf_stream_t stream) {
  return usf::wgrad_reduce_jobs(jobs, block_job, n_blocks, (hipStream_t)stream);
}
int usf_base_param_grad_f32(const float* z, int64_t ldz, const float* g_lp, int64_t M, int64_t D, int32_t base, const float* loc,
                            const float* scale, float* d_loc_scale, float* workspace, int64_t workspace_floats, usf_stream_t stream) {
  return usf::base_param_grad(z, ldz, g_lp, M, D, base, loc, scale, d_loc_scale, workspace, workspace_floats, (hipStream_t)stream);
}
int usf_set_clock_buffer(unsigned long long* dev_buf2) { usf::g_clock_buffer = dev_buf2; return 0; }
int usf_mfma_probe(const float* src1024, float* sink, int64_t iters, int64_t blocks, double* flops_out, usf_stream_t stream) {
  return usf::mfma_probe(src1024, sink, iters, blocks, flops_out, (hipStream_t)stream);
}
int usf_wgrad_planes_colsum_ok(int64_t M, int64_t N, int64_t K) { return usf::wgrad_planes_colsum_ok(M, N, K); }
int64_t usf_wgrad_planes_workspace_floats(int64_t M, int64_t N, int64_t K) { return usf::wgrad_planes_workspace_floats(M, N, K); }
int usf_wgrad_planes_ok(int64_t M, int64_t N, int64_t K) { return usf::wgrad_planes_ok(M, N, K); }
int usf_split_planes_f32(const float* X, int64_t ldx, int64_t M, int64_t N, void* planes, int64_t ldp, int64_t plane_stride,
                         usf_stream_t stream) {
  return usf::split_planes(X, ldx, M, N, planes, ldp, plane_stride, (hipStream_t)stream);
}
int usf_colsum_f32(const float* Y, int64_t ldy, int64_t M, int64_t N, float* out, float alpha, float beta,
                   float* workspace, int64_t workspace_floats, usf_stream_t stream) {
  return usf::colsum(Y, ldy, M, N, out, alpha, beta, workspace, workspace_floats, (hipStream_t)stream);
}
int usf_affine_prep_f32(const float* L_raw, const float* U_raw, const float* bias, const float* vk, const float* w0, int64_t n,
                        int32_t C, int32_t nvs, float* M, float* Minv, float* b, float* c, float* ladj, float* save,
                        usf_stream_t stream) {
  return usf::affine_prep(L_raw, U_raw, bias, vk, w0, n, C, nvs, M, Minv, b, c, ladj, save, (hipStream_t)stream);
}
int usf_affine_prep_bwd_f32(const float* save, const float* bias, const float* vk, const float* w0, const float* Minv,
                            const float* b, const float* dM, const float* dMinv, const float* db, const float* dc,
                            const float* dladj, int64_t n, int32_t C, int32_t nvs, float* dL_raw, float* dU_raw, float* dbias,
                            float* dvk, usf_stream_t stream) {
  return usf::affine_prep_bwd(save, bias, vk, w0, Minv, b, dM, dMinv, db, dc, dladj, n, C, nvs, dL_raw, dU_raw, dbias, dvk,
                              (hipStream_t)stream);
}
int usf_grad_jobs_f32(const usf_grad_job* jobs, const int32_t* block_job, int64_t n_blocks, usf_stream_t stream) {
  return usf::grad_jobs(jobs, block_job, n_blocks, (hipStream_t)stream);
}
int usf_act_grad_f32(float* d, int64_t ldd, const float* h, int64_t ldh, int64_t M, int64_t H, int32_t act, float slope,
                     usf_stream_t stream) {
  return usf::act_grad(d, ldd, h, ldh, M, H, act, slope, (hipStream_t)stream);
}
int usf_base_logprob_grad_f32(const float* z, int64_t ldz, const float* g_lp, int64_t M, int64_t D, int32_t base,
                              const float* loc, const float* scale, float* g, int64_t ldg, usf_stream_t stream) {
  return usf::base_grad(z, ldz, g_lp, M, D, base, loc, scale, g, ldg, (hipStream_t)stream);
}

int usf_radial_logprob_f32(const float* z, int64_t ldz, int64_t M, int64_t D, int32_t p_id, const float* loc, int32_t norm,
                           int32_t K, const float* par_a, const float* par_b, const float* logits, double logdv_const,
                           float logdet_const, const double* logdet_dev, float* logp, float* r_out, double* sum_out,
                           usf_stream_t stream) {
  return usf::radial_logprob(z, ldz, M, D, p_id, loc, norm, K, par_a, par_b, logits, logdv_const, logdet_const, logdet_dev, logp,
                             r_out, sum_out, (hipStream_t)stream);
}
int64_t usf_radial_logprob_grad_workspace(int64_t M, int64_t D) { return usf::radial_grad_workspace(M, D); }
int usf_radial_logprob_grad_f32(const float* z, int64_t ldz, const float* r, const float* g_lp, int64_t M, int64_t D, int32_t p_id,
                                const float* loc, int32_t norm, int32_t K, const float* par_a, const float* par_b,
                                const float* logits, float* g, int64_t ldg, float* d_loc, float* d_a, float* d_b, float* d_logits,
                                void* workspace, int64_t workspace_bytes, usf_stream_t stream) {
  return usf::radial_grad(z, ldz, r, g_lp, M, D, p_id, loc, norm, K, par_a, par_b, logits, g, ldg, d_loc, d_a, d_b, d_logits,
                          workspace, workspace_bytes, (hipStream_t)stream);
}

// USF_OP_CALL: the recorded arguments back into the entry point's prototype
static int run_call(const usf_call_desc* c, usf_stream_t stream) {
  const uint64_t* a = c->a;
#define P(i) (reinterpret_cast<const float*>((uintptr_t)a[i]))
#define Q(i) (reinterpret_cast<float*>((uintptr_t)a[i]))
#define I(i) ((int64_t)a[i])
#define J(i) ((int32_t)a[i])
  auto F = [&](int i) { float f; uint32_t u = (uint32_t)a[i]; memcpy(&f, &u, 4); return f; };
  auto Dbl = [&](int i) { double d; uint64_t u = a[i]; memcpy(&d, &u, 8); return d; };
  static const int nargs[] = {0, 8, 8, 10, 5, 7, 16, 17, 16, 11, 17, 15};
  if (c->fn < 1 || c->fn > USF_FN_GATED_TAIL || c->n_args != nargs[c->fn]) {
    usf::set_error("usf_run_ops: call op with unknown function %d or %d arguments", c->fn, c->n_args);
    return -2;
  }
  switch (c->fn) {
    case USF_FN_SCALE: return usf_scale_f32(P(0), I(1), Q(2), I(3), I(4), I(5), P(6), J(7), stream);
    case USF_FN_CHANNEL_AFFINE: return usf_channel_affine_f32(P(0), Q(1), I(2), I(3), I(4), P(5), P(6), P(7), stream);
    case USF_FN_LAYERNORM_CHANNELS: return usf_layernorm_channels_f32(P(0), Q(1), I(2), I(3), I(4), P(5), P(6), F(7), J(8), F(9), stream);
    case USF_FN_GATED_RESIDUAL: return usf_gated_residual_f32(P(0), P(1), Q(2), I(3), I(4), stream);
    case USF_FN_MASKED_RESIDUAL: return usf_masked_residual_f32(P(0), P(1), P(2), F(3), Q(4), I(5), I(6), stream);
    case USF_FN_POINTWISE_CONV:
      return usf_pointwise_conv_f32(P(0), Q(1), I(2), I(3), I(4), I(5), P(6), P(7), J(8), F(9), J(10), F(11), P(12), P(13), P(14), F(15), stream);
    case USF_FN_GATED_TAIL:
      return usf_gated_tail_f32(P(0), P(1), Q(2), I(3), I(4), I(5), P(6), P(7), J(8), F(9), J(10), F(11), P(12), P(13), F(14), stream);
    case USF_FN_CONV2D_SAME:
      return usf_conv2d_same_f32(P(0), Q(1), I(2), I(3), I(4), I(5), I(6), I(7), (const void*)(uintptr_t)a[8], P(9), P(10), J(11), F(12),
                                 J(13), F(14), P(15), I(16), stream);
    case USF_FN_CONV2D_SAME_RES: {
      const int rc = usf_conv2d_same_res_f32(P(0), Q(1), I(2), I(3), I(4), I(5), I(6), I(7), (const void*)(uintptr_t)a[8], P(9), P(10), J(11),
                                             F(12), P(13), P(14), F(15), stream);
      if (rc == 1) usf::set_error("usf_run_ops: recorded usf_conv2d_same_res_f32 call is not served by the fused form any more");
      return rc;
    }
    case USF_FN_BASE_LOGPROB:
      return usf_base_logprob_f32(P(0), I(1), I(2), I(3), J(4), P(5), P(6), F(7), reinterpret_cast<const double*>((uintptr_t)a[8]), Q(9),
                                  reinterpret_cast<double*>((uintptr_t)a[10]), stream);
    case USF_FN_RADIAL_LOGPROB:
      return usf_radial_logprob_f32(P(0), I(1), I(2), I(3), J(4), P(5), J(6), J(7), P(8), P(9), P(10), Dbl(11), F(12),
                                    reinterpret_cast<const double*>((uintptr_t)a[13]), Q(14), Q(15),
                                    reinterpret_cast<double*>((uintptr_t)a[16]), stream);
  }
#undef P
#undef Q
#undef I
#undef J
  return -2;
}

int usf_run_ops(const usf_op* ops, int32_t n_ops, usf_stream_t stream) {
  if (n_ops < 0 || (n_ops > 0 && !ops)) { usf::set_error("usf_run_ops: bad op list"); return -1; }
  for (int32_t i = 0; i < n_ops; ++i) {
    int rc;
    switch (ops[i].kind) {
      case USF_OP_LINEAR: rc = usf::linear_dispatch(&ops[i].u.linear, (hipStream_t)stream); break;
      case USF_OP_COUPLING: rc = usf::coupling_dispatch(&ops[i].u.coupling, (hipStream_t)stream); break;
      case USF_OP_PACK_PLANES: rc = usf::pack_planes(&ops[i].u.pack_planes, (hipStream_t)stream); break;
      case USF_OP_GEMM_PLANES: rc = usf::gemm_planes(&ops[i].u.gemm_planes, (hipStream_t)stream); break;
      case USF_OP_COUPLING_PLANES: rc = usf::coupling_planes(&ops[i].u.coupling_planes, (hipStream_t)stream); break;
      case USF_OP_GATED_NORM: rc = usf::gated_norm_rows(&ops[i].u.gated_norm, (hipStream_t)stream); break;
      case USF_OP_CALL: rc = run_call(&ops[i].u.call, stream); break;
      default: usf::set_error("usf_run_ops: op %d has unknown kind %d", i, ops[i].kind); return -2;
    }
    if (rc != 0) return rc;
  }
  return 0;
}

}  // extern "C"
