// usf_coupling_additive_f32 for TINY layers at launch-bound batches: the live flat configuration of the reference
// (experiments/synthetic/gaussian_mixture.yaml:50-93: D = 2 .. 100, DenseNN [32, 32], Flow.fit at batch 32) spends a replayed
// training step on ~200 dependent launches of a few microseconds each, three of them per coupling layer and direction.  Here the
// whole layer -- conditioner MLP, (Leaky)ReLU or the backward gates, residual -- is ONE launch whose duration is one global-memory
// round trip: every weight matrix of the layer is staged into LDS up front (they fit: <= 64 KB together), the rows' activations
// stay in LDS between the layers, plain fp32 FMAs (a 32 x 32 layer is 1024 dot products of 32 terms: not matrix-core work).
// Serves the training forms of the descriptor too: hidden_out (the saved activations) and USF_ACT_GATE (the conditioner's
// data-gradient chain on transposed weights), which the fp32 MFMA kernel (usf_coupling.hip) does not.
#include "usf_common.h"

namespace usf {

constexpr int TINY_ROWS = 32;          // batch rows per block
constexpr int TINY_MAXW = 128;         // widest column segment
constexpr int TINY_MAXH = 64;          // widest hidden layer
constexpr int TINY_MAX_M = 256;        // above this the MFMA kernels fill the chip

struct TinyArgs {
  const float* z; float* out; int64_t ldz;
  int M, off_pass, n_pass, off_trans, n_trans, nh;
  int rows[4], K[4];                   // layer l = 0 .. nh (nh: the output layer): W[l] is [rows[l], K[l]]
  const float* W[4]; int64_t ldw[4]; const float* b[4];
  const float* ctx; const float* W_ctx; const float* b_ctx;
  float* hout[3]; int64_t ld_hout;
  const float* gate[3]; int64_t ld_gate;
  float sign, slope; int act;
  int woff[4];                         // float offsets of the layers' weight images in LDS (row stride K + 1)
  int xoff, haoff, hboff;              // ... of the input rows [32, n_pass + 1] and the two activation buffers [32, 65]
};

__global__ __launch_bounds__(256) void coupling_tiny_kernel(const TinyArgs p) {
  extern __shared__ __attribute__((aligned(16))) float tiny_lds[];
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * TINY_ROWS;
  // ---- everything the layer reads, in one wave of loads: the weight images and the rows' conditioning half ----
#pragma unroll 1
  for (int l = 0; l <= p.nh; ++l) {
    const int R = p.rows[l], K = p.K[l];
    float* dst = tiny_lds + p.woff[l];
    for (int idx = tid; idx < R * K; idx += 256) {
      const int r = idx / K, k = idx - r * K;
      dst[r * (K + 1) + k] = p.W[l][(int64_t)r * p.ldw[l] + k];
    }
  }
  float* x0 = tiny_lds + p.xoff;
  for (int idx = tid; idx < TINY_ROWS * p.n_pass; idx += 256) {
    const int r = idx / p.n_pass, k = idx - r * p.n_pass;
    x0[r * (p.n_pass + 1) + k] = (row0 + r < p.M) ? p.z[(int64_t)(row0 + r) * p.ldz + p.off_pass + k] : 0.f;
  }
  __syncthreads();
  float* hin = x0;
  int ldin = p.n_pass + 1;
  float* hbuf[2] = {tiny_lds + p.haoff, tiny_lds + p.hboff};
  // ---- hidden layers: h = act(W h_prev + b [+ context branch]); gate mode: (W h_prev) * leaky_relu'(saved activation) ----
#pragma unroll 1
  for (int l = 0; l < p.nh; ++l) {
    const int H = p.rows[l], K = p.K[l];
    const float* Wl = tiny_lds + p.woff[l];
    float* hout = hbuf[l & 1];
    for (int idx = tid; idx < TINY_ROWS * H; idx += 256) {
      const int r = idx / H, u = idx - r * H;
      const int row = row0 + r;
      float acc = p.b[l][u];                                   // (accumulators start at the bias, as in usf_coupling.hip)
      const float* w = Wl + u * (K + 1);
      const float* x = hin + r * ldin;
      for (int k = 0; k < K; ++k) acc = fmaf(w[k], x[k], acc);
      float v;
      if (p.act == USF_ACT_GATE) {
        const float gt = (row < p.M) ? p.gate[l][(int64_t)row * p.ld_gate + u] : 0.f;
        v = acc * (gt > 0.f ? 1.f : p.slope);
      } else {
        if (l == 0 && p.ctx != nullptr && row < p.M) acc = acc + (p.ctx[row] * p.W_ctx[u] + p.b_ctx[u]);   // networks.py:741-743
        v = act_apply(acc, p.act, p.slope);
      }
      hout[r * (TINY_MAXH + 1) + u] = v;
      if (p.hout[l] != nullptr && row < p.M) p.hout[l][(int64_t)row * p.ld_hout + u] = v;
    }
    __syncthreads();
    hin = hout;
    ldin = TINY_MAXH + 1;
  }
  // ---- output layer + residual: out[:, trans] = z[:, trans] + sign * (W_out h + b_out) ----
  {
    const int N = p.rows[p.nh], K = p.K[p.nh];
    const float* Wl = tiny_lds + p.woff[p.nh];
    for (int idx = tid; idx < TINY_ROWS * N; idx += 256) {
      const int r = idx / N, n = idx - r * N;
      const int row = row0 + r;
      if (row >= p.M) continue;
      float acc = 0.f;
      const float* w = Wl + n * (K + 1);
      const float* x = hin + r * ldin;
      for (int k = 0; k < K; ++k) acc = fmaf(w[k], x[k], acc);
      const int64_t o = (int64_t)row * p.ldz + p.off_trans + n;
      p.out[o] = p.z[o] + p.sign * (acc + p.b[p.nh][n]);
    }
  }
}

// the descriptor has passed coupling_dispatch's checks (usf_coupling.hip)
bool coupling_tiny_eligible(const usf_coupling_desc* d) {
  if (!tuning("coupling_tiny", 1) || d->M > TINY_MAX_M || d->n_pass > TINY_MAXW || d->n_trans > TINY_MAXW) return false;
  int64_t floats = 0, k = d->n_pass;
  for (int i = 0; i < d->n_hidden; ++i) {
    if (d->hidden[i] < 1 || d->hidden[i] > TINY_MAXH) return false;
    floats += (int64_t)d->hidden[i] * (k + 1);
    k = d->hidden[i];
  }
  floats += d->n_trans * (k + 1) + TINY_ROWS * (d->n_pass + 1) + 2 * TINY_ROWS * (TINY_MAXH + 1);
  return floats * 4 <= 64 * 1024;
}

int coupling_tiny_dispatch(const usf_coupling_desc* d, hipStream_t stream) {
  if (d->off_pass < 0 || d->off_trans < 0 || d->off_pass + d->n_pass > d->ldz || d->off_trans + d->n_trans > d->ldz) {
    set_error("usf_coupling_additive_f32: column segments outside the rows (ldz %lld)", (long long)d->ldz);
    return -2;
  }
  TinyArgs a;
  a.z = d->z; a.out = d->out; a.ldz = d->ldz;
  a.M = (int)d->M; a.off_pass = (int)d->off_pass; a.n_pass = (int)d->n_pass; a.off_trans = (int)d->off_trans; a.n_trans = (int)d->n_trans;
  a.nh = d->n_hidden;
  const bool want_h = d->hidden_out[0] != nullptr, gated = d->act == USF_ACT_GATE;
  int k = (int)d->n_pass, off = 0;
  for (int l = 0; l <= a.nh; ++l) {
    const bool last = l == a.nh;
    a.rows[l] = last ? (int)d->n_trans : d->hidden[l];
    a.K[l] = k;
    a.W[l] = last ? d->W_out : (l == 0 ? d->W_in : d->W_hid[l - 1]);
    a.ldw[l] = last ? d->ldw_out : (l == 0 ? d->ldw_in : d->ldw_hid[l - 1]);
    a.b[l] = last ? d->b_out : (l == 0 ? d->b_in : d->b_hid[l - 1]);
    if (!a.W[l] || !a.b[l] || a.ldw[l] < k) { set_error("usf_coupling_additive_f32: bad layer %d (null pointer or ld < K)", l); return -2; }
    a.woff[l] = off;
    off += a.rows[l] * (k + 1);
    k = a.rows[l];
  }
  for (int l = a.nh + 1; l < 4; ++l) { a.rows[l] = a.K[l] = a.woff[l] = 0; a.W[l] = nullptr; a.b[l] = nullptr; a.ldw[l] = 0; }
  a.xoff = off; off += TINY_ROWS * (a.n_pass + 1);
  a.haoff = off; off += TINY_ROWS * (TINY_MAXH + 1);
  a.hboff = off; off += TINY_ROWS * (TINY_MAXH + 1);
  for (int l = 0; l < 3; ++l) {
    a.hout[l] = (want_h && l < a.nh) ? d->hidden_out[l] : nullptr;
    a.gate[l] = (gated && l < a.nh) ? d->gate[l] : nullptr;
    if (want_h && l < a.nh && !d->hidden_out[l]) { set_error("usf_coupling_additive_f32: hidden_out needs a buffer for every hidden layer"); return -1; }
    if (gated && l < a.nh && !d->gate[l]) { set_error("usf_coupling_additive_f32: USF_ACT_GATE needs gate[l] for every hidden layer"); return -1; }
  }
  a.ld_hout = d->ld_hidden_out; a.ld_gate = d->ld_gate;
  int hmax = 0;
  for (int l = 0; l < a.nh; ++l) hmax = d->hidden[l] > hmax ? d->hidden[l] : hmax;
  if ((want_h && a.ld_hout < hmax) || (gated && a.ld_gate < hmax)) { set_error("usf_coupling_additive_f32: ld_hidden_out / ld_gate below the hidden width"); return -2; }
  a.ctx = gated ? nullptr : d->context; a.W_ctx = d->W_ctx; a.b_ctx = d->b_ctx;
  if (a.ctx && (!a.W_ctx || !a.b_ctx)) { set_error("usf_coupling_additive_f32: context needs W_ctx and b_ctx"); return -1; }
  a.sign = d->sign; a.slope = d->slope; a.act = d->act;
  const dim3 grid((unsigned)((d->M + TINY_ROWS - 1) / TINY_ROWS)), block(256);
  hipLaunchKernelGGL(coupling_tiny_kernel, grid, block, (size_t)off * sizeof(float), stream, a);
  return check_launch("usf_coupling_additive_f32(tiny)");
}

}  // namespace usf
