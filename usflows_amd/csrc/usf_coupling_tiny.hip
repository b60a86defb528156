// usf_coupling_additive_f32 for TINY layers at launch-bound batches: the live flat configuration of the reference
// (experiments/synthetic/gaussian_mixture.yaml:50-93: D = 2 .. 100, DenseNN [32, 32], Flow.fit at batch 32) spends a replayed
// training step on ~200 dependent launches of a few microseconds each, three of them per coupling layer and direction.  Here the
// whole layer -- conditioner MLP, (Leaky)ReLU or the backward gates, residual -- is ONE launch built around a single exposed
// global-memory round trip: everything the layer reads -- every weight image, bias, the rows' conditioning half and residual, the
// backward gates -- is fetched in ONE batch of loads and parked in LDS (<= 64 KB); the rows' activations stay in LDS between the
// layers, which run on v_mfma_f32_16x16x4_f32.  Serves the training forms of the descriptor too: hidden_out (the saved
// activations) and USF_ACT_GATE (the conditioner's data-gradient chain on transposed weights), which the fp32 MFMA kernel
// (usf_coupling.hip) does not.
#include "usf_common.h"

namespace usf {

constexpr int TINY_ROWS = 32;          // batch rows per block
constexpr int TINY_MAXW = 64;          // widest column segment / hidden layer
constexpr int TINY_MAX_M = 256;        // above this the MFMA kernels fill the chip
constexpr int TINY_HS = TINY_MAXW + 4; // row stride of the activation buffers
constexpr int TINY_NV4 = 4;            // 16-byte groups per thread of the largest weight image (64 x 64 floats)

struct TinySeg {                       // a [rows, cols] piece of global memory and where it goes in LDS
  const float* src; int64_t ld;
  int n, cols;                         // n = rows * cols elements
  float inv_cols;                      // 1 / cols (row of element e = floor((e + 0.5) * inv_cols))
  int dst, dld;                        // LDS float offset and row stride
  int batch;                           // rows are batch rows (offset by the block's first row, zero beyond M)
};

struct TinyArgs {
  float* out; const float* z; int64_t ldz;
  int M, off_trans, n_trans, nh;
  int rows[4], K4[4];                  // layer l = 0 .. nh (nh: the output layer): rows[l] outputs, K4[l] = K rounded up to 4
  int woff[4], wld[4], boff[4], goff[3];
  TinySeg w[4], b[4], g[3], x0, zres, ctx, wctx, bctx;
  int xoff, xld, haoff, hboff, zoff, coff, wcoff, bcoff;
  float* hout[3]; int64_t ld_hout;
  float sign, slope; int act, has_ctx;
};

// element e of a segment -> (row, column): floor((e + 0.5) / cols) through the reciprocal (exact for these sizes)
template <int NV>
__device__ __forceinline__ void tiny_load(const TinySeg& s, int row0, int M, float (&v)[NV]) {
  const int cnt = (s.n + 255) >> 8;                       // block-uniform: iterations beyond it are skipped by a scalar branch
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = 0.f;
    if (i < cnt) {
      const int e = threadIdx.x + 256 * i;
      if (e < s.n) {
        const int r = (int)(((float)e + 0.5f) * s.inv_cols);
        const int c = e - r * s.cols;
        const int gr = s.batch ? row0 + r : r;
        if (!s.batch || gr < M) v[i] = s.src[(int64_t)gr * s.ld + c];
      }
    }
  }
}
template <int NV>
__device__ __forceinline__ void tiny_store(const TinySeg& s, float* lds, const float (&v)[NV]) {
  const int cnt = (s.n + 255) >> 8;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < cnt) {
      const int e = threadIdx.x + 256 * i;
      if (e < s.n) {
        const int r = (int)(((float)e + 0.5f) * s.inv_cols);
        lds[s.dst + r * s.dld + (e - r * s.cols)] = v[i];
      }
    }
  }
}
// weight images: whole 16-byte groups (the source rows are 16-byte aligned and zero beyond K up to the next multiple of 4)
template <int NV>
__device__ __forceinline__ void tiny_load4(const TinySeg& s, f32x4 (&v)[NV]) {
  const int cnt = (s.n + 255) >> 8;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < cnt) {
      const int e = threadIdx.x + 256 * i;
      if (e < s.n) {
        const int r = (int)(((float)e + 0.5f) * s.inv_cols);
        v[i] = *reinterpret_cast<const f32x4*>(s.src + (int64_t)r * s.ld + 4 * (e - r * s.cols));
      }
    }
  }
}
template <int NV>
__device__ __forceinline__ void tiny_store4(const TinySeg& s, float* lds, const f32x4 (&v)[NV]) {
  const int cnt = (s.n + 255) >> 8;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < cnt) {
      const int e = threadIdx.x + 256 * i;
      if (e < s.n) {
        const int r = (int)(((float)e + 0.5f) * s.inv_cols);
        *reinterpret_cast<f32x4*>(lds + s.dst + r * s.dld + 4 * (e - r * s.cols)) = v[i];
      }
    }
  }
}

// One wave per SIMD and nothing to hide behind: the kernel's time is its instruction count.  The layers run on
// v_mfma_f32_16x16x4_f32 (a 32-row x 16-unit tile per wave: K / 4 matrix instructions and two LDS words per lane each), the weight
// images are staged as 16-byte groups, and every staging loop runs only the iterations its segment has.
// NW: 16-byte groups per thread of the largest weight image, NS: elements per thread of the largest row segment (inputs, residual,
// gates) -- the staging code is straight-line and runs once per launch, cold: the small instantiation (<= 32 x 32 layers, the
// live flat configuration) is a third of the large one's code
template <int NW, int NS>
__global__ __launch_bounds__(256) void coupling_tiny_kernel(const TinyArgs p) {
  extern __shared__ __attribute__((aligned(16))) float tl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lj = lane & 15, lg = lane >> 4;
  const int row0 = blockIdx.x * TINY_ROWS;
  const bool gated = p.act == USF_ACT_GATE;
  // ---- everything the layer reads, ONE batch of loads: the arithmetic of a layer is far too short to hide a second round trip ----
  f32x4 vw[4][NW];
  float vx[NS], vz[NS], vb[4][1], vg[3][NS], vc[1], vwc[1], vbc[1];
#pragma unroll
  for (int l = 0; l < 4; ++l)
    if (l <= p.nh) { tiny_load4<NW>(p.w[l], vw[l]); tiny_load<1>(p.b[l], row0, p.M, vb[l]); }
  tiny_load<NS>(p.x0, row0, p.M, vx);
  tiny_load<NS>(p.zres, row0, p.M, vz);
  if (gated) {
#pragma unroll
    for (int l = 0; l < 3; ++l)
      if (l < p.nh) tiny_load<NS>(p.g[l], row0, p.M, vg[l]);
  }
  if (p.has_ctx) { tiny_load<1>(p.ctx, row0, p.M, vc); tiny_load<1>(p.wctx, row0, p.M, vwc); tiny_load<1>(p.bctx, row0, p.M, vbc); }
  // zero the padded images while the loads fly (weight rows beyond the layer's width, inputs beyond K, activations beyond the widths)
  for (int e = tid; e < (p.zoff >> 2); e += 256) reinterpret_cast<f32x4*>(tl)[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
#pragma unroll
  for (int l = 0; l < 4; ++l)
    if (l <= p.nh) { tiny_store4<NW>(p.w[l], tl, vw[l]); tiny_store<1>(p.b[l], tl, vb[l]); }
  tiny_store<NS>(p.x0, tl, vx);
  tiny_store<NS>(p.zres, tl, vz);
  if (gated) {
#pragma unroll
    for (int l = 0; l < 3; ++l)
      if (l < p.nh) tiny_store<NS>(p.g[l], tl, vg[l]);
  }
  if (p.has_ctx) { tiny_store<1>(p.ctx, tl, vc); tiny_store<1>(p.wctx, tl, vwc); tiny_store<1>(p.bctx, tl, vbc); }
  __syncthreads();
  const float* hin = tl + p.xoff;
  int ldin = p.xld;
#pragma unroll 1
  for (int l = 0; l <= p.nh; ++l) {
    const bool last = l == p.nh;
    const int H = p.rows[l], K4 = p.K4[l];
    const int wld = p.wld[l];
    const float* bl = tl + p.boff[l];
    float* hout = tl + ((l & 1) ? p.hboff : p.haoff);
    const int ntile = 2 * ((H + 15) >> 4);                      // (row tile, unit tile) pairs: tile t = 2 ut + rt
    for (int t = wave; t < ntile; t += 4) {                     // wave-uniform
      const int rt = t & 1, ut = t >> 1;
      // D[unit, row] += W[unit, k] x[row, k]: A lane (unit lj, k lg), B lane (k lg, row lj); D lane: row lj, units 4 lg + i
      const float* wa = tl + p.woff[l] + (ut * 16 + lj) * wld + lg;
      const float* xb = hin + (rt * 16 + lj) * ldin + lg;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < K4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 * k], xb[4 * k], acc, 0, 0, 0);
      const int r = rt * 16 + lj, row = row0 + r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = ut * 16 + 4 * lg + i;
        if (u >= H) {                                             // the next layer reads whole groups of four: zeros, never leftovers
          if (!last && u < ((H + 3) & ~3)) hout[r * TINY_HS + u] = 0.f;
          continue;
        }
        if (!last) {
          float v = acc[i] + bl[u];
          if (gated) {
            v = v * (tl[p.goff[l] + r * H + u] > 0.f ? 1.f : p.slope);
          } else {
            if (l == 0 && p.has_ctx) v = v + (tl[p.coff + r] * tl[p.wcoff + u] + tl[p.bcoff + u]);          // networks.py:741-743
            v = act_apply(v, p.act, p.slope);
          }
          hout[r * TINY_HS + u] = v;
          if (p.hout[l] != nullptr && row < p.M) p.hout[l][(int64_t)row * p.ld_hout + u] = v;
        } else if (row < p.M) {
          p.out[(int64_t)row * p.ldz + p.off_trans + u] = tl[p.zoff + r * p.n_trans + u] + p.sign * (acc[i] + bl[u]);
        }
      }
    }
    if (last) break;
    __syncthreads();
    hin = hout;
    ldin = TINY_HS;
  }
}

static inline int r4(int64_t v) { return (int)((v + 3) / 4 * 4); }

// LDS floats of a descriptor's layout (see coupling_tiny_dispatch); the engine restates it (FlowEngine.tiny_coupling)
static int64_t tiny_lds_floats(const usf_coupling_desc* d) {
  int64_t f = 0, k = d->n_pass, rows_sum = 0, gate_sum = 0;
  for (int i = 0; i <= d->n_hidden; ++i) {
    const int64_t rows = i == d->n_hidden ? d->n_trans : d->hidden[i];
    f += ((rows + 15) / 16 * 16) * (r4(k) + 4);
    rows_sum += rows;
    if (i < d->n_hidden) gate_sum += TINY_ROWS * rows;
    k = rows;
  }
  return f + TINY_ROWS * (r4(d->n_pass) + 4) + 2 * TINY_ROWS * TINY_HS + rows_sum + TINY_ROWS * d->n_trans + gate_sum + TINY_ROWS + 2 * TINY_MAXW;
}

// the descriptor has passed coupling_dispatch's basic checks (usf_coupling.hip)
bool coupling_tiny_eligible(const usf_coupling_desc* d) {
  if (!tuning("coupling_tiny", 1) || d->M > TINY_MAX_M || d->n_pass > TINY_MAXW || d->n_trans > TINY_MAXW) return false;
  for (int i = 0; i < d->n_hidden; ++i)
    if (d->hidden[i] < 1 || d->hidden[i] > TINY_MAXW) return false;
  // the weight images are staged as 16-byte groups: aligned rows, and (the padding contract of the header) zeros beyond K up to
  // the next multiple of 4
  int64_t k = d->n_pass;
  for (int i = 0; i <= d->n_hidden; ++i) {
    const float* W = i == d->n_hidden ? d->W_out : (i == 0 ? d->W_in : d->W_hid[i - 1]);
    const int64_t ldw = i == d->n_hidden ? d->ldw_out : (i == 0 ? d->ldw_in : d->ldw_hid[i - 1]);
    if (!W || !aligned16(W) || (ldw & 3) || ldw < r4(k)) return false;
    k = i == d->n_hidden ? d->n_trans : d->hidden[i];
  }
  return tiny_lds_floats(d) * 4 <= 64 * 1024;
}

int coupling_tiny_dispatch(const usf_coupling_desc* d, hipStream_t stream) {
  if (d->off_pass < 0 || d->off_trans < 0 || d->off_pass + d->n_pass > d->ldz || d->off_trans + d->n_trans > d->ldz) {
    set_error("usf_coupling_additive_f32: column segments outside the rows (ldz %lld)", (long long)d->ldz);
    return -2;
  }
  TinyArgs a;
  a.out = d->out; a.z = d->z; a.ldz = d->ldz;
  a.M = (int)d->M; a.off_trans = (int)d->off_trans; a.n_trans = (int)d->n_trans; a.nh = d->n_hidden;
  const bool want_h = d->hidden_out[0] != nullptr, gated = d->act == USF_ACT_GATE;
  auto seg = [](const float* src, int64_t ld, int rows, int cols, int dst, int dld, int batch) {
    TinySeg s;
    s.src = src; s.ld = ld; s.n = src ? rows * cols : 0; s.cols = cols > 0 ? cols : 1; s.inv_cols = 1.0f / (float)s.cols;
    s.dst = dst; s.dld = dld; s.batch = batch;
    return s;
  };
  // LDS layout: [weight images | input rows | two activation buffers]  (zero-filled: p.zoff floats)  | residual | biases | gates | ctx
  int k = (int)d->n_pass, off = 0;
  for (int l = 0; l <= a.nh; ++l) {
    const bool last = l == a.nh;
    const int rows = last ? (int)d->n_trans : d->hidden[l];
    const float* W = last ? d->W_out : (l == 0 ? d->W_in : d->W_hid[l - 1]);
    const int64_t ldw = last ? d->ldw_out : (l == 0 ? d->ldw_in : d->ldw_hid[l - 1]);
    const float* b = last ? d->b_out : (l == 0 ? d->b_in : d->b_hid[l - 1]);
    if (!W || !b || ldw < k) { set_error("usf_coupling_additive_f32: bad layer %d (null pointer or ld < K)", l); return -2; }
    a.rows[l] = rows; a.K4[l] = r4(k) / 4; a.woff[l] = off; a.wld[l] = r4(k) + 4;
    a.w[l] = seg(W, ldw, rows, r4(k) / 4, off, a.wld[l], 0);        // (cols = 16-byte groups per row)
    off += ((rows + 15) / 16 * 16) * a.wld[l];
    k = rows;
  }
  for (int l = a.nh + 1; l < 4; ++l) { a.rows[l] = a.K4[l] = a.woff[l] = a.wld[l] = a.boff[l] = 0; a.w[l] = seg(nullptr, 0, 0, 1, 0, 0, 0); a.b[l] = a.w[l]; }
  a.xoff = off; a.xld = r4(d->n_pass) + 4;
  a.x0 = seg(d->z + d->off_pass, d->ldz, TINY_ROWS, (int)d->n_pass, off, a.xld, 1);
  off += TINY_ROWS * a.xld;
  a.haoff = off; off += TINY_ROWS * TINY_HS;
  a.hboff = off; off += TINY_ROWS * TINY_HS;
  a.zoff = off;
  a.zres = seg(d->z + d->off_trans, d->ldz, TINY_ROWS, (int)d->n_trans, off, (int)d->n_trans, 1);
  off += TINY_ROWS * (int)d->n_trans;
  for (int l = 0; l <= a.nh; ++l) {
    const float* b = l == a.nh ? d->b_out : (l == 0 ? d->b_in : d->b_hid[l - 1]);
    a.boff[l] = off;
    a.b[l] = seg(b, 0, 1, a.rows[l], off, a.rows[l], 0);
    off += a.rows[l];
  }
  int hmax = 0;
  for (int l = 0; l < 3; ++l) {
    const bool used = l < a.nh;
    a.hout[l] = (want_h && used) ? d->hidden_out[l] : nullptr;
    if (want_h && used && !d->hidden_out[l]) { set_error("usf_coupling_additive_f32: hidden_out needs a buffer for every hidden layer"); return -1; }
    if (gated && used && !d->gate[l]) { set_error("usf_coupling_additive_f32: USF_ACT_GATE needs gate[l] for every hidden layer"); return -1; }
    a.goff[l] = off;
    a.g[l] = seg((gated && used) ? d->gate[l] : nullptr, d->ld_gate, TINY_ROWS, used ? d->hidden[l] : 1, off, used ? d->hidden[l] : 1, 1);
    if (gated && used) off += TINY_ROWS * d->hidden[l];
    if (used && d->hidden[l] > hmax) hmax = d->hidden[l];
  }
  a.ld_hout = d->ld_hidden_out;
  if ((want_h && a.ld_hout < hmax) || (gated && d->ld_gate < hmax)) { set_error("usf_coupling_additive_f32: ld_hidden_out / ld_gate below the hidden width"); return -2; }
  a.has_ctx = (!gated && d->context != nullptr) ? 1 : 0;
  if (a.has_ctx && (!d->W_ctx || !d->b_ctx)) { set_error("usf_coupling_additive_f32: context needs W_ctx and b_ctx"); return -1; }
  a.coff = off; a.ctx = seg(a.has_ctx ? d->context : nullptr, 1, TINY_ROWS, 1, off, 1, 1); off += TINY_ROWS;
  a.wcoff = off; a.wctx = seg(a.has_ctx ? d->W_ctx : nullptr, 0, 1, d->hidden[0], off, d->hidden[0], 0); off += TINY_MAXW;
  a.bcoff = off; a.bctx = seg(a.has_ctx ? d->b_ctx : nullptr, 0, 1, d->hidden[0], off, d->hidden[0], 0); off += TINY_MAXW;
  a.sign = d->sign; a.slope = d->slope; a.act = d->act;
  if ((int64_t)off * 4 > 64 * 1024) { set_error("usf_coupling_additive_f32: tiny-layer layout exceeds 64 KB of LDS"); return -3; }
  const dim3 grid((unsigned)((d->M + TINY_ROWS - 1) / TINY_ROWS)), block(256);
  bool small = a.x0.n <= 1024 && a.zres.n <= 1024;
  for (int l = 0; l <= a.nh; ++l) small = small && a.w[l].n <= 256;
  for (int l = 0; l < a.nh; ++l) small = small && a.g[l].n <= 1024;
  if (small) hipLaunchKernelGGL((coupling_tiny_kernel<1, 4>), grid, block, (size_t)off * sizeof(float), stream, a);
  else hipLaunchKernelGGL((coupling_tiny_kernel<TINY_NV4, 8>), grid, block, (size_t)off * sizeof(float), stream, a);
  return check_launch("usf_coupling_additive_f32(tiny)");
}

}  // namespace usf
