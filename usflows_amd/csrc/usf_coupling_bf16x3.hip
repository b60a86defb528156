// Split-precision (bf16x3) variant of the fused additive-coupling kernel: same structure as
// usf_coupling.hip -- hidden activations live in MFMA accumulators, only weights travel through LDS -- but
// every product runs on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16) with both operands split three
// ways (x = x1 + x2 + x3 exactly, six MFMAs per fp32-equivalent product; see usf_linear_bf16x3.hip).
//
//  * 512-thread block = 8 waves x 16 batch rows; one block per CU (2 waves per SIMD).  A weight stage is
//    [256 x 32 k] (k-slab of W_in / W_h) or [32 n x 256 k] (n-tile of W_out), three bf16 planes = 48 KB,
//    double-buffered; the 8 waves share it, so a stage is staged once per 128 batch rows.
//  * accumulator -> operand: the 16x16 f32 accumulator has hidden unit 4*(lane>>4)+r of batch row lane&15
//    in register r.  Two consecutive tiles (hidden 0-15, 16-31 of a 32-k step) give each lane 8 values
//    whose k indices are {4g..4g+3, 16+4g..16+4g+3}; the hidden-layer weights are stored with that
//    k-permutation (pack time), so the split accumulators ARE the B-operand fragments of the next layer.
//  * output orientation as in the f32 kernel: a lane ends with 4 consecutive output features of its row
//    -> 16-byte residual load / store.
#include "usf_common.h"

namespace usf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C3_ROWS = 128;               // batch rows per block (8 waves x 16)
constexpr int C3_NT = 512;
constexpr int C3_HMAX = 256;

struct Cpl3Args {
  const float* z; float* out; int64_t ldz;
  int M, off_pass, n_pass, off_trans, n_trans, n_trans4;
  const __bf16* Win; int64_t ld_in, pl_in; const float* b_in;
  const __bf16* Whid[2]; int64_t ld_hid, pl_hid; const float* b_hid[2];
  const __bf16* Wout; int64_t ld_out, pl_out; const float* b_out;
  const float* ctx; const float* W_ctx; const float* b_ctx;
  float sign, slope; int act;
};

__device__ __forceinline__ void c3_split3(const f32x4 x0, const f32x4 x1, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = (j < 4) ? x0[j & 3] : x1[j & 3];
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;
    const __bf16 m = (__bf16)r;
    const float r2 = r - (float)m;
    p1[j] = h; p2[j] = m; p3[j] = (__bf16)r2;
  }
}

// six-term product, smallest terms first
#define C3_MFMA6(ACC, W1, W2, W3, A1, A2, A3)                                   \
  do {                                                                          \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W3, A1, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W2, A2, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W1, A3, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W2, A1, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W1, A2, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W1, A1, ACC, 0, 0, 0);        \
  } while (0)

template <int NH, int T>
__global__ __launch_bounds__(C3_NT, 2) void coupling_bf16x3_kernel(const Cpl3Args p) {
  constexpr int HP = 16 * T;               // padded hidden width
  constexpr int KS = T / 2;                // 32-k steps over a hidden layer
  constexpr int SLOTS = 3 * 4 * HP;        // 16-B slots per stage (k-slab: 3 planes x 4 chunks x HP rows;
                                           //                       n-tile: 3 planes x (HP/8) chunks x 32 rows)
  constexpr int NST = SLOTS / C3_NT;       // float4 staged per thread per stage
  static_assert(SLOTS % C3_NT == 0, "stage shape");
  __shared__ __attribute__((aligned(16))) float lds[2][SLOTS * 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lj = lane & 15;
  const int lg = lane >> 4;
  const int wrow0 = blockIdx.x * C3_ROWS + wave * 16;
  const int rowc = min(wrow0 + lj, p.M - 1);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- weight stages: slot(plane, chunk, row) = (plane * NC + chunk) * NRW + row --------------------------
  // k-slab: NC = 4, NRW = HP; thread -> row = (tid&7) + 8*(tid>>5) + RS*i, chunk = (tid>>3)&3, RS = 128
  // n-tile: NC = HP/8, NRW = 32; thread idx = tid + 512 i -> row = (idx&7) + 8*(idx / (8*NC)), chunk = (idx>>3) % NC
  constexpr int NPP = NST / 3;             // staged float4 per thread per plane
  const int kr0 = (tid & 7) + 8 * (tid >> 5), kc = (tid >> 3) & 3;
  auto issue_k = [&](const __bf16* W, int64_t ld, int64_t pl, int k0, f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i)
        st[q * NPP + i] = *reinterpret_cast<const f32x4*>(W + q * pl + (int64_t)(kr0 + 128 * i) * ld + k0 + 8 * kc);
  };
  auto store_k = [&](int buf, const f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i)
        *reinterpret_cast<f32x4*>(&lds[buf][4 * ((q * 4 + kc) * HP + kr0 + 128 * i)]) = st[q * NPP + i];
  };
  constexpr int NC = HP / 8;
  auto n_row = [&](int i) { return ((tid + C3_NT * i) & 7) + 8 * ((tid + C3_NT * i) / (8 * NC)); };
  auto n_chunk = [&](int i) { return ((tid + C3_NT * i) >> 3) % NC; };
  auto issue_n = [&](const __bf16* W, int64_t ld, int64_t pl, int n0, f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i)
        st[q * NPP + i] = *reinterpret_cast<const f32x4*>(W + q * pl + (int64_t)(n0 + n_row(i)) * ld + 8 * n_chunk(i));
  };
  auto store_n = [&](int buf, const f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i)
        *reinterpret_cast<f32x4*>(&lds[buf][4 * ((q * NC + n_chunk(i)) * 32 + n_row(i))]) = st[q * NPP + i];
  };

  const int nS1 = (p.n_pass + 31) / 32;
  const int nS3 = (p.n_trans + 31) / 32;

  // accumulators start at the layer bias
  f32x4 X1[T], X2[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    X1[t] = *reinterpret_cast<const f32x4*>(p.b_in + t * 16 + 4 * lg);
    if (NH >= 2) X2[t] = *reinterpret_cast<const f32x4*>(p.b_hid[0] + t * 16 + 4 * lg);
  }

  f32x4 st[NST];
  f32x4 zc[2], zn[2];
  const float* zrow = p.z + (int64_t)rowc * p.ldz + p.off_pass;
  auto issue_z = [&](int k0, f32x4 (&dst)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) dst[u] = *reinterpret_cast<const f32x4*>(zrow + min(k0 + 8 * lg + 4 * u, p.n_pass - 4));
  };
  auto finish_z = [&](int k0, f32x4 (&dst)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) dst[u] = (k0 + 8 * lg + 4 * u < p.n_pass) ? dst[u] : zero4;
  };

  // one 32-k step over all hidden tiles: X[ht] += W(ht) . B, B given as 3 planes; W fragments from a k-slab
  auto mfma_slab = [&](int buf, f32x4 (&X)[T], const bf16x8 b1, const bf16x8 b2, const bf16x8 b3) {
    const float* wl = &lds[buf][4 * (lg * HP + lj)];
#pragma unroll
    for (int ht = 0; ht < T; ++ht) {
      const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((0 * 4) * HP + ht * 16));
      const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((1 * 4) * HP + ht * 16));
      const bf16x8 w3 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((2 * 4) * HP + ht * 16));
      C3_MFMA6(X[ht], w1, w2, w3, b1, b2, b3);
    }
    // fragment reads two tiles ahead of their MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int ht = 0; ht < T; ++ht) {
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      if (ht + 2 < T) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    }
  };
  auto issue_after_input = [&](f32x4 (&s_)[NST]) {
    if (NH >= 2) issue_k(p.Whid[0], p.ld_hid, p.pl_hid, 0, s_); else issue_n(p.Wout, p.ld_out, p.pl_out, 0, s_);
  };
  auto store_after_input = [&](int buf, const f32x4 (&s_)[NST]) {
    if (NH >= 2) store_k(buf, s_); else store_n(buf, s_);
  };

  int g = 0;
  issue_k(p.Win, p.ld_in, p.pl_in, 0, st);
  issue_z(0, zc);
  store_k(0, st);
  finish_z(0, zc);
  __syncthreads();

  // ================= phase 1: X1[h][row] += W_in[h][k] * z[row][k] ============================
  for (int s = 0; s + 1 < nS1; ++s, ++g) {
    const int buf = g & 1;
    issue_k(p.Win, p.ld_in, p.pl_in, (s + 1) * 32, st);
    issue_z((s + 1) * 32, zn);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 z1, z2, z3;
    c3_split3(zc[0], zc[1], z1, z2, z3);
    mfma_slab(buf, X1, z1, z2, z3);
    __builtin_amdgcn_sched_barrier(0);
    store_k(buf ^ 1, st);
    finish_z((s + 1) * 32, zn);
    zc[0] = zn[0]; zc[1] = zn[1];
    __syncthreads();
  }
  {
    const int buf = g & 1;
    issue_after_input(st);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 z1, z2, z3;
    c3_split3(zc[0], zc[1], z1, z2, z3);
    mfma_slab(buf, X1, z1, z2, z3);
    __builtin_amdgcn_sched_barrier(0);
    store_after_input(buf ^ 1, st);
    __syncthreads();
    ++g;
  }

  auto ctx_act = [&](f32x4 (&X)[T], bool with_ctx) {
    const float cv = with_ctx ? p.ctx[rowc] : 0.f;
#pragma unroll
    for (int ht = 0; ht < T; ++ht) {
      f32x4 wc = zero4, bc = zero4;
      if (with_ctx) {
        wc = *reinterpret_cast<const f32x4*>(p.W_ctx + ht * 16 + 4 * lg);
        bc = *reinterpret_cast<const f32x4*>(p.b_ctx + ht * 16 + 4 * lg);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = X[ht][t];
        if (with_ctx) v = v + (cv * wc[t] + bc[t]);
        X[ht][t] = act_apply(v, p.act, p.slope);
      }
    }
  };
  ctx_act(X1, p.ctx != nullptr);

  // ================= phase 2: Xout[h2][row] += W_h[h2][h1'] * Xin[h1'][row] ====================
  auto hidden_layer = [&](f32x4 (&Xin)[T], f32x4 (&Xout)[T], int l) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int buf = g & 1;
      const bool last = (ks + 1 == KS);
      const bool next_is_hidden = (l + 2 < NH);
      if (!last) issue_k(p.Whid[l], p.ld_hid, p.pl_hid, (ks + 1) * 32, st);
      else if (next_is_hidden) issue_k(p.Whid[l + 1], p.ld_hid, p.pl_hid, 0, st);
      else issue_n(p.Wout, p.ld_out, p.pl_out, 0, st);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 x1, x2, x3;
      c3_split3(Xin[2 * ks], Xin[2 * ks + 1], x1, x2, x3);     // k order {4g.., 16+4g..}: the weights' pack-time order
      mfma_slab(buf, Xout, x1, x2, x3);
      __builtin_amdgcn_sched_barrier(0);
      if (!last || next_is_hidden) store_k(buf ^ 1, st); else store_n(buf ^ 1, st);
      __syncthreads();
      ++g;
    }
    ctx_act(Xout, false);
  };
  if (NH >= 2) hidden_layer(X1, X2, 0);
  if (NH >= 3) {
#pragma unroll
    for (int t = 0; t < T; ++t) X1[t] = *reinterpret_cast<const f32x4*>(p.b_hid[1] + t * 16 + 4 * lg);
    hidden_layer(X2, X1, 1);
  }

  // ================= phase 3: out[row][n] = z[row][n] + sign * (b_out[n] + sum_h X[h][row] W_out[n][h']) =====
  auto output_layer = [&](f32x4 (&X)[T]) {
    // split the final hidden activations once: KS steps x 3 planes
    bf16x8 xp[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) c3_split3(X[2 * ks], X[2 * ks + 1], xp[ks][0], xp[ks][1], xp[ks][2]);
    const int orow = min(wrow0 + lj, p.M - 1);
    for (int nt = 0; nt < nS3; ++nt, ++g) {
      const int buf = g & 1;
      issue_n(p.Wout, p.ld_out, p.pl_out, min(nt + 1, nS3 - 1) * 32, st);
      int col[2];
      f32x4 res[2], acc[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        col[u] = nt * 32 + u * 16 + 4 * lg;
        res[u] = *reinterpret_cast<const f32x4*>(p.z + (int64_t)orow * p.ldz + p.off_trans + min(col[u], p.n_trans4 - 4));
        acc[u] = *reinterpret_cast<const f32x4*>(p.b_out + col[u]);     // b_out is padded to 32 * nS3
      }
      __builtin_amdgcn_sched_barrier(0);
      const float* wl = &lds[buf][4 * (lg * 32 + lj)];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((0 * NC + 4 * ks) * 32 + u * 16));
          const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((1 * NC + 4 * ks) * 32 + u * 16));
          const bf16x8 w3 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((2 * NC + 4 * ks) * 32 + u * 16));
          C3_MFMA6(acc[u], w1, w2, w3, xp[ks][0], xp[ks][1], xp[ks][2]);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        if (i + 2 < 2 * KS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      store_n(buf ^ 1, st);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = res[u][r] + p.sign * acc[u][r];
        float* dst = p.out + (int64_t)(wrow0 + lj) * p.ldz + p.off_trans + col[u];
        if (wrow0 + lj < p.M) {
          if (col[u] + 3 < p.n_trans) {
            *reinterpret_cast<f32x4*>(dst) = v;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (col[u] + r < p.n_trans) dst[r] = v[r];
          }
        }
      }
      __syncthreads();
    }
  };
  if (NH == 2) output_layer(X2); else output_layer(X1);
}

bool coupling_bf16x3_eligible(const usf_coupling_desc* d) {
  if (!d->split_in || !d->split_out || d->n_hidden < 1 || d->n_hidden > 3) return false;
  for (int i = 0; i + 1 < d->n_hidden; ++i)
    if (!d->split_hid[i]) return false;
  int hmax = 0;
  for (int i = 0; i < d->n_hidden; ++i) hmax = d->hidden[i] > hmax ? d->hidden[i] : hmax;
  return hmax > 128 && hmax <= C3_HMAX && d->M >= 1024;     // instantiated for the 256-wide tile set only
}

int coupling_bf16x3_dispatch(const usf_coupling_desc* d, hipStream_t stream) {
  Cpl3Args a;
  a.z = d->z; a.out = d->out; a.ldz = d->ldz;
  a.M = (int)d->M; a.off_pass = (int)d->off_pass; a.n_pass = (int)d->n_pass; a.off_trans = (int)d->off_trans;
  a.n_trans = (int)d->n_trans; a.n_trans4 = (int)((d->n_trans + 3) / 4 * 4);
  a.Win = reinterpret_cast<const __bf16*>(d->split_in); a.ld_in = d->split_in_ld; a.pl_in = d->split_in_plane; a.b_in = d->b_in;
  for (int i = 0; i < 2; ++i) {
    const bool used = i + 1 < d->n_hidden;
    a.Whid[i] = reinterpret_cast<const __bf16*>(used ? d->split_hid[i] : d->split_in);
    a.b_hid[i] = used ? d->b_hid[i] : d->b_in;
  }
  a.ld_hid = d->split_hid_ld; a.pl_hid = d->split_hid_plane;
  a.Wout = reinterpret_cast<const __bf16*>(d->split_out); a.ld_out = d->split_out_ld; a.pl_out = d->split_out_plane; a.b_out = d->b_out;
  a.ctx = d->context; a.W_ctx = d->W_ctx; a.b_ctx = d->b_ctx;
  a.sign = d->sign; a.slope = d->slope; a.act = d->act;
  const int64_t kp = ((d->n_pass + 31) / 32) * 32;
  if (d->split_in_ld < kp || d->split_hid_ld < C3_HMAX * (d->n_hidden > 1) || d->split_out_ld < C3_HMAX || (d->split_in_ld & 7) ||
      (d->split_out_ld & 7) || !aligned16(d->split_in) || !aligned16(d->split_out)) {
    set_error("usf_coupling_additive_f32(bf16x3): split-plane padding contract violated");
    return -2;
  }
  const dim3 grid((unsigned)((d->M + C3_ROWS - 1) / C3_ROWS)), block(C3_NT);
  switch (d->n_hidden) {
    case 1: hipLaunchKernelGGL((coupling_bf16x3_kernel<1, 16>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((coupling_bf16x3_kernel<2, 16>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((coupling_bf16x3_kernel<3, 16>), grid, block, 0, stream, a); break;
  }
  return check_launch("usf_coupling_additive_f32(bf16x3)");
}

}  // namespace usf
