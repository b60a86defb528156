// usf_coupling_additive_f32: one launch = one MaskedCoupling layer (transforms.py:277-306) with
// its dense (leaky-)ReLU conditioner (networks.py:739-751), hidden activations never leaving
// the register file.
//
//   out[:, trans] = z[:, trans] + sign * ( W_out . act( W_h . act( W_in . z[:, pass] + b_in [+ ctx] ) + b_h ) + b_out )
//
// Design (DESIGN.md "fused coupling kernel"):
//  * One wave owns 32 batch rows for the whole layer; a 256-thread block = 128 rows.
//  * The MLP is evaluated TRANSPOSED: X1 = W_in . Z^T, X2 = W_h . X1, ... so that every hidden
//    activation tile is an MFMA *accumulator* with the batch row on the lane and the hidden unit in
//    the register index.  On gfx950 the 32x32 f32 accumulator layout (row = (r&3) + 8*(r>>2) +
//    4*(lane>>5)) is exactly the k-permutation this library feeds its f32 MFMAs with (lane half h
//    owns k = 8q+4h..+3), so an accumulator register IS the next layer's B operand: no LDS round
//    trip, no shuffles, no conversion.  The last product flips orientation (X as the A operand)
//    so that its output has the feature on the lane -> coalesced residual read-modify-write.
//  * Only the weights travel through LDS: a unified stream of 32 KB stages (a [H x 32] k-slab of
//    W_in / W_h, or a [32 x H] n-tile of W_out), register-staged and double-buffered, the loads
//    of stage g+1 pinned in front of the MFMA block of stage g.
//  * z fragments (phase 1) and residual values (phase 3) go global -> registers directly; each
//    element is needed by exactly one wave.
#include "usf_common.h"

namespace usf {

constexpr int CPL_HT = 8;                  // hidden tiles of 32 -> hidden widths up to 256
constexpr int CPL_BK = 32;                 // k per weight slab
constexpr int CPL_LDW = CPL_BK + 4;        // slab row stride (odd number of 16-B slots)
constexpr int CPL_H = CPL_HT * 32;
constexpr int CPL_LDW3 = CPL_H + 4;        // n-tile row stride
constexpr int CPL_BUF = (CPL_H * CPL_LDW > 32 * CPL_LDW3) ? CPL_H * CPL_LDW : 32 * CPL_LDW3;
constexpr int CPL_NST = 8;                 // float4 staged per thread per stage (2048 / 256)

struct CplArgs {
  const float* z; float* out; int64_t ldz;
  int M, off_pass, n_pass, off_trans, n_trans;
  int nh; int h[USF_MAX_HIDDEN];
  const float* W_in; int64_t ldw_in; const float* b_in;
  const float* W_hid[2]; const float* b_hid[2]; int64_t ldw_hid[2];
  const float* W_out; int64_t ldw_out; const float* b_out;
  const float* ctx; const float* W_ctx; const float* b_ctx;
  float sign, slope; int act;
};

template <int NH>
__global__ __launch_bounds__(256, 1) void coupling_kernel(const CplArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[2][CPL_BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int wrow0 = blockIdx.x * 128 + wave * 32;
  const int rowc = min(wrow0 + li, p.M - 1);           // rows >= M: valid garbage, never stored
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- unified weight-stage stream --------------------------------------------------------
  const int nS1 = (p.n_pass + CPL_BK - 1) / CPL_BK;
  int nS2[2] = {0, 0};
#pragma unroll
  for (int l = 0; l + 1 < NH; ++l) nS2[l] = (p.h[l] + 31) / 32;
  const int nS3 = (p.n_trans + 31) / 32;
  const int g2 = nS1;                                   // first stage of hidden layer 0 -> 1
  const int g2b = g2 + nS2[0];                          // first stage of hidden layer 1 -> 2
  const int g3 = g2b + nS2[1];                          // first stage of the output product
  const int G = g3 + nS3;

  struct Src { const float* base; int64_t ld; int nrows, ncols, row0, k0, shift; };
  auto stage_src = [&](int g) -> Src {
    g = min(g, G - 1);
    Src s;
    if (g < g2) {                 // W_in k-slab: rows = hidden units, cols = pass-through features
      s.base = p.W_in; s.ld = p.ldw_in; s.nrows = p.h[0]; s.ncols = p.n_pass; s.row0 = 0; s.k0 = g * CPL_BK; s.shift = 3;
    } else if (g < g2b) {
      s.base = p.W_hid[0]; s.ld = p.ldw_hid[0]; s.nrows = p.h[NH > 1 ? 1 : 0]; s.ncols = p.h[0]; s.row0 = 0;
      s.k0 = (g - g2) * CPL_BK; s.shift = 3;
    } else if (g < g3) {
      s.base = p.W_hid[1]; s.ld = p.ldw_hid[1]; s.nrows = p.h[NH > 2 ? 2 : 0]; s.ncols = p.h[NH > 1 ? 1 : 0]; s.row0 = 0;
      s.k0 = (g - g2b) * CPL_BK; s.shift = 3;
    } else {                      // W_out n-tile: 32 output features x all of the last hidden layer
      s.base = p.W_out; s.ld = p.ldw_out; s.nrows = p.n_trans; s.ncols = p.h[NH - 1]; s.row0 = (g - g3) * 32;
      s.k0 = 0; s.shift = 6;
    }
    return s;
  };
  // loads are unconditional (clamped); the zero-fill happens at LDS-store time
  auto issue_stage = [&](int g, f32x4 (&st)[CPL_NST]) {
    const Src s = stage_src(g);
    const int cmask = (1 << s.shift) - 1;
#pragma unroll
    for (int i = 0; i < CPL_NST; ++i) {
      const int idx = tid + i * 256;
      const int row = min(s.row0 + (idx >> s.shift), s.nrows - 1);
      const int k = min(s.k0 + 4 * (idx & cmask), s.ncols - 4);
      st[i] = *reinterpret_cast<const f32x4*>(s.base + (int64_t)row * s.ld + k);
    }
  };
  auto store_stage = [&](int g, int buf, const f32x4 (&st)[CPL_NST]) {
    const Src s = stage_src(g);
    const int cmask = (1 << s.shift) - 1;
    const int ldl = (s.shift == 3) ? CPL_LDW : CPL_LDW3;
#pragma unroll
    for (int i = 0; i < CPL_NST; ++i) {
      const int idx = tid + i * 256;
      const int r = idx >> s.shift, c = idx & cmask;
      const bool ok = (s.row0 + r < s.nrows) && (s.k0 + 4 * c < s.ncols);
      *reinterpret_cast<f32x4*>(&lds[buf][r * ldl + 4 * c]) = ok ? st[i] : zero4;
    }
  };

  f32x16 X1[CPL_HT], X2[CPL_HT];
#pragma unroll
  for (int t = 0; t < CPL_HT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) X1[t][r] = 0.f;

  f32x4 st[CPL_NST];
  f32x4 zc[4], zn[4];
  const float* zrow = p.z + (int64_t)rowc * p.ldz + p.off_pass;
  auto issue_z = [&](int k0, f32x4 (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f32x4*>(zrow + min(k0 + 8 * q + 4 * lh, p.n_pass - 4));
  };
  auto finish_z = [&](int k0, f32x4 (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = (k0 + 8 * q + 4 * lh < p.n_pass) ? dst[q] : zero4;
  };

  int g = 0;                                            // current stage
  issue_stage(0, st);
  issue_z(0, zc);
  store_stage(0, 0, st);
  finish_z(0, zc);
  __syncthreads();

  // ================= phase 1: X1[h][row] += W_in[h][k] * z[row][k] ============================
  const int nht0 = (p.h[0] + 31) / 32;
  for (int s = 0; s < nS1; ++s, ++g) {
    const int buf = g & 1;
    issue_stage(g + 1, st);
    issue_z((s + 1) * CPL_BK, zn);
    __builtin_amdgcn_sched_barrier(0);
    const float* wl = &lds[buf][li * CPL_LDW + 4 * lh];
    const int nq = min(4, (p.n_pass - s * CPL_BK + 7) / 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < nq) {
#pragma unroll
        for (int ht = 0; ht < CPL_HT; ++ht) {
          if (ht < nht0) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(wl + ht * 32 * CPL_LDW + 8 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) X1[ht] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], zc[q][t], X1[ht], 0, 0, 0);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    store_stage(g + 1, buf ^ 1, st);
    finish_z((s + 1) * CPL_BK, zn);
#pragma unroll
    for (int q = 0; q < 4; ++q) zc[q] = zn[q];
    __syncthreads();
  }

  // bias (+ context branch) + activation on an accumulator array, in registers
  auto bias_act = [&](f32x16 (&X)[CPL_HT], const float* bias, int h, bool with_ctx) {
    const float cv = with_ctx ? p.ctx[rowc] : 0.f;
#pragma unroll
    for (int ht = 0; ht < CPL_HT; ++ht) {
      __builtin_amdgcn_sched_barrier(0);               // one tile's bias loads in flight at a time
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hi = min(ht * 32 + 8 * q + 4 * lh, h - 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + hi);
        f32x4 wc = zero4, bc = zero4;
        if (with_ctx) {
          wc = *reinterpret_cast<const f32x4*>(p.W_ctx + hi);
          bc = *reinterpret_cast<const f32x4*>(p.b_ctx + hi);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float v = X[ht][4 * q + t] + b[t];                  // layers[0](x)
          if (with_ctx) v = v + (cv * wc[t] + bc[t]);         // + layers[1](context), networks.py:741-743
          X[ht][4 * q + t] = act_apply(v, p.act, p.slope);
        }
      }
    }
  };
  bias_act(X1, p.b_in, p.h[0], p.ctx != nullptr);

  // ================= phase 2: Xout[h2][row] += W_h[h2][h1] * Xin[h1][row] ====================
  auto hidden_layer = [&](f32x16 (&Xin)[CPL_HT], f32x16 (&Xout)[CPL_HT], int l) {
    const int nkt = (p.h[l] + 31) / 32;
    const int nho = (p.h[l + 1] + 31) / 32;
#pragma unroll
    for (int kt = 0; kt < CPL_HT; ++kt) {
      if (kt < nkt) {
        const int buf = g & 1;
        issue_stage(g + 1, st);
        __builtin_amdgcn_sched_barrier(0);
        const float* wl = &lds[buf][li * CPL_LDW + 4 * lh];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int ht = 0; ht < CPL_HT; ++ht) {
            if (ht < nho) {
              const f32x4 a = *reinterpret_cast<const f32x4*>(wl + ht * 32 * CPL_LDW + 8 * q);
#pragma unroll
              for (int t = 0; t < 4; ++t)
                Xout[ht] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], Xin[kt][4 * q + t], Xout[ht], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_stage(g + 1, buf ^ 1, st);
        __syncthreads();
        ++g;
      }
    }
    bias_act(Xout, p.b_hid[l], p.h[l + 1], false);
  };
  if (NH >= 2) {
#pragma unroll
    for (int t = 0; t < CPL_HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) X2[t][r] = 0.f;
    hidden_layer(X1, X2, 0);
  }
  if (NH >= 3) {
#pragma unroll
    for (int t = 0; t < CPL_HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) X1[t][r] = 0.f;
    hidden_layer(X2, X1, 1);
  }

  // ================= phase 3: T[row][n] = sum_h Xlast[h][row] * W_out[n][h]; residual ==========
  auto output_layer = [&](f32x16 (&X)[CPL_HT]) {
    const int nkt = (p.h[NH - 1] + 31) / 32;
    for (int nt = 0; nt < nS3; ++nt, ++g) {
      const int buf = g & 1;
      issue_stage(g + 1, st);
      const int col = nt * 32 + li;
      const int colc = min(col, p.n_trans - 1);
      float res[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(wrow0 + (r & 3) + 8 * (r >> 2) + 4 * lh, p.M - 1);
        res[r] = p.z[(int64_t)row * p.ldz + p.off_trans + colc];
      }
      const float bo = p.b_out[colc];
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* wl = &lds[buf][li * CPL_LDW3 + 4 * lh];
#pragma unroll
      for (int kt = 0; kt < CPL_HT; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(wl + kt * 32 + 8 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(X[kt][4 * q + t], b[t], acc, 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      store_stage(g + 1, buf ^ 1, st);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = res[r] + p.sign * (acc[r] + bo);
        if (row < p.M && col < p.n_trans) p.out[(int64_t)row * p.ldz + p.off_trans + col] = v;
      }
      __syncthreads();
    }
  };
  if (NH == 2) output_layer(X2); else output_layer(X1);
}

int coupling_max_width() { return CPL_H; }

int coupling_dispatch(const usf_coupling_desc* d, hipStream_t stream) {
  if (!d) { set_error("usf_coupling_additive_f32: null descriptor"); return -1; }
  if (d->M < 0 || d->M > 0x7fffffff || d->n_pass <= 0 || d->n_trans <= 0 || d->n_hidden < 1 || d->n_hidden > 3) {
    set_error("usf_coupling_additive_f32: bad sizes (M=%lld n_pass=%lld n_trans=%lld n_hidden=%d; fused kernel "
              "supports 1..3 hidden layers)", (long long)d->M, (long long)d->n_pass, (long long)d->n_trans, d->n_hidden);
    return -2;
  }
  if (d->M == 0) return 0;
  if (!d->z || !d->out || !d->W_in || !d->b_in || !d->W_out || !d->b_out) { set_error("usf_coupling_additive_f32: null pointer"); return -1; }
  if (d->out != d->z || d->ldo != d->ldz) { set_error("usf_coupling_additive_f32: this version works in place (out == z)"); return -2; }
  if ((d->n_pass & 3) || (d->off_pass & 3) || (d->ldz & 3) || (d->ldw_in & 3) || (d->ldw_out & 3) || !aligned16(d->z) ||
      !aligned16(d->W_in) || !aligned16(d->W_out) || !aligned16(d->b_in)) {
    set_error("usf_coupling_additive_f32: n_pass/off_pass/ldz/ldw must be multiples of 4 and pointers 16-byte aligned");
    return -2;
  }
  if (d->act != USF_ACT_NONE && d->act != USF_ACT_LEAKY_RELU) { set_error("usf_coupling_additive_f32: bad act"); return -2; }
  CplArgs a;
  a.z = d->z; a.out = d->out; a.ldz = d->ldz;
  a.M = (int)d->M; a.off_pass = (int)d->off_pass; a.n_pass = (int)d->n_pass; a.off_trans = (int)d->off_trans; a.n_trans = (int)d->n_trans;
  a.nh = d->n_hidden;
  for (int i = 0; i < USF_MAX_HIDDEN; ++i) a.h[i] = (i < d->n_hidden) ? d->hidden[i] : 4;
  for (int i = 0; i < d->n_hidden; ++i) {
    if (a.h[i] < 4 || (a.h[i] & 3) || a.h[i] > CPL_H) {
      set_error("usf_coupling_additive_f32: hidden width %d must be a multiple of 4 in [4, %d]", a.h[i], CPL_H);
      return -2;
    }
  }
  a.W_in = d->W_in; a.ldw_in = d->ldw_in; a.b_in = d->b_in;
  for (int i = 0; i < 2; ++i) {
    const bool used = i + 1 < d->n_hidden;
    a.W_hid[i] = used ? d->W_hid[i] : d->W_in;
    a.b_hid[i] = used ? d->b_hid[i] : d->b_in;
    a.ldw_hid[i] = used ? d->ldw_hid[i] : d->ldw_in;
    if (used && (!d->W_hid[i] || !d->b_hid[i] || (d->ldw_hid[i] & 3) || !aligned16(d->W_hid[i]) || !aligned16(d->b_hid[i]))) {
      set_error("usf_coupling_additive_f32: bad hidden layer %d", i);
      return -2;
    }
  }
  a.W_out = d->W_out; a.ldw_out = d->ldw_out; a.b_out = d->b_out;
  a.ctx = d->context; a.W_ctx = d->W_ctx; a.b_ctx = d->b_ctx;
  if (a.ctx && (!a.W_ctx || !a.b_ctx || !aligned16(a.b_ctx) || !aligned16(a.W_ctx))) { set_error("usf_coupling_additive_f32: context needs W_ctx and b_ctx"); return -1; }
  a.sign = d->sign; a.slope = d->slope; a.act = d->act;
  const dim3 grid((unsigned)((d->M + 127) / 128)), block(256);
  switch (d->n_hidden) {
    case 1: hipLaunchKernelGGL((coupling_kernel<1>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((coupling_kernel<2>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((coupling_kernel<3>), grid, block, 0, stream, a); break;
  }
  return check_launch("usf_coupling_additive_f32");
}

}  // namespace usf
