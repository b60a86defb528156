// usf_coupling_additive_f32: one launch = one MaskedCoupling layer (transforms.py:277-306) with
// its dense (leaky-)ReLU conditioner (networks.py:739-751), hidden activations never leaving
// the register file.
//
//   out[:, trans] = z[:, trans] + sign * ( W_out . act( W_h . act( W_in . z[:, pass] + b_in [+ ctx] ) + b_h ) + b_out )
//
// Design (DESIGN.md "fused coupling kernel"):
//  * One wave owns 16 batch rows for the whole layer (v_mfma_f32_16x16x4_f32); a 256-thread block
//    = 64 rows, two blocks per CU (<= 256 VGPRs, 64 KB LDS each) so that one block's barriers and
//    load waits are covered by the other block's MFMAs.
//  * The MLP is evaluated TRANSPOSED: X1 = W_in . Z^T, X2 = W_h . X1, ... so that every hidden
//    activation tile is an MFMA *accumulator* with the batch row on the lane and the hidden unit in
//    the register index.  On gfx950 the 16x16 f32 accumulator layout (row = 4*(lane>>4) + r) is
//    exactly the k-permutation this library feeds its f32 MFMAs with (lane group g owns
//    k = 16q+4g..+3), so an accumulator register IS the next layer's B operand: no LDS round trip,
//    no shuffles, no conversion.  The output product keeps that orientation: a lane ends up with 4
//    consecutive output features of its row -> one 16-byte residual load and one 16-byte store.
//  * Only the weights travel through LDS: a unified stream of 32 KB stages (a [256 x 32] k-slab of
//    W_in / W_h, or a [32 x 256] n-tile of W_out), register-staged and double-buffered, the loads
//    of stage g+1 pinned in front of the MFMA block of stage g.  LDS image is k-chunk-major
//    (slot = chunk * rows + row): fragment reads and staging writes are both conflict-free.
//  * z fragments (phase 1) and residual values (phase 3) go global -> registers directly; each
//    element is needed by exactly one wave.
#include "usf_common.h"

namespace usf {

constexpr int CPL_BK = 32;                 // k per weight slab
constexpr int CPL_ROWS = 64;               // batch rows per block (4 waves x 16)
constexpr int CPL_HMAX = 256;              // widest hidden layer the kernel is instantiated for

struct CplArgs {
  const float* z; float* out; int64_t ldz;
  int M, off_pass, n_pass, off_trans, n_trans, n_trans4;
  const float* W_in; int64_t ldw_in; const float* b_in;
  const float* W_hid[2]; const float* b_hid[2]; int64_t ldw_hid[2];
  const float* W_out; int64_t ldw_out; const float* b_out;
  const float* ctx; const float* W_ctx; const float* b_ctx;
  float sign, slope; int act;
  unsigned long long* dbg;              // tuning builds only (USF_STAMP)
};

// NH = number of hidden layers, T = hidden tiles of 16 kept in registers: every hidden layer is
// treated as Hp = 16*T wide (the caller zero-pads the weights to that width, see the header), so
//  * no per-tile branches: each MFMA block is straight-line code and the LDS fragment reads are
//    software-pipelined two tiles ahead of the MFMAs that consume them;
//  * no clamps / selects on the weight loads: a thread's NST loads of a stage sit at constant
//    strides from one per-thread pointer.
template <int NH, int T>
__global__ __launch_bounds__(256, 2) void coupling_kernel(const CplArgs p) {
  constexpr int HP = 16 * T;               // padded hidden width
  constexpr int NST = T / 2;               // float4 staged per thread per stage (HP*32/4/256)
  constexpr int BUF = HP * CPL_BK;         // floats per stage buffer
  __shared__ __attribute__((aligned(16))) float lds[2][BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lj = lane & 15;
  const int lg = lane >> 4;
  const int wrow0 = blockIdx.x * CPL_ROWS + wave * 16;
  const int rowc = min(wrow0 + lj, p.M - 1);           // rows >= M: valid garbage, never stored
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- weight stages ---------------------------------------------------------------------------
  // LDS image: slot(row, chunk) = chunk * nr + row (16-B slots).  k-slab: nr = HP rows x 8 chunks;
  // n-tile: nr = 32 rows x NCH chunks.  Thread -> (row, chunk): 8 consecutive lanes take 8 consecutive
  // rows of one chunk (conflict-free ds_write_b128); a wave covers 8 rows x 8 chunks = 8 whole 128-B
  // lines of the weight matrix (coalesced).  Load i of a thread is at a constant stride from load 0.
  // The per-load address is re-derived from ONE live pointer (opaque bump) -- left to itself hipcc
  // precomputes all NST 64-bit addresses of every weight matrix, keeps them live across the whole
  // kernel and spills inside the MFMA loops.
  const int kr0 = (tid & 7) + 8 * (tid >> 6), kc = (tid >> 3) & 7;          // k-slab: row = kr0 + 32 i
  auto issue_k = [&](const float* W, int64_t ld, int k0, f32x4 (&st)[NST]) {
    int64_t off = (int64_t)kr0 * ld + (k0 + 4 * kc);       // opaque OFFSET (an opaque pointer would turn
    const int64_t step = 32 * ld;                          // the loads into flat_load: lgkmcnt + vmcnt)
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      st[i] = *reinterpret_cast<const f32x4*>(W + off);
      off += step;
      asm volatile("" : "+v"(off));
    }
  };
  auto store_k = [&](int buf, const f32x4 (&st)[NST]) {
    float* dst = &lds[buf][4 * (kc * HP + kr0)];
#pragma unroll
    for (int i = 0; i < NST; ++i) *reinterpret_cast<f32x4*>(dst + 4 * 32 * i) = st[i];
  };
  // n-tile: idx = tid + 256 i -> row = (idx & 7) + 8 * (idx / (8 * NCH)), chunk = (idx >> 3) % NCH, which
  // separates into a per-thread base and a compile-time function of i:
  //   T=4 : row = (tid&7) + 8*(tid>>7) + 16 i, chunk = (tid>>3)&15
  //   T=8 : row = (tid&7) + 8 i,               chunk = (tid>>3)
  //   T=16: row = (tid&7) + 8 (i>>1),          chunk = (tid>>3) + 32 (i&1)
  const int nr0 = (T == 4) ? (tid & 7) + 8 * (tid >> 7) : (tid & 7);
  const int nc0 = (T == 4) ? ((tid >> 3) & 15) : (tid >> 3);
  auto n_drow = [](int i) { return (T == 4) ? 16 * i : ((T == 8) ? 8 * i : 8 * (i >> 1)); };
  auto n_dchunk = [](int i) { return (T == 16) ? 32 * (i & 1) : 0; };
  auto issue_n = [&](const float* W, int64_t ld, int n0, f32x4 (&st)[NST]) {
    int64_t off = (int64_t)(n0 + nr0) * ld + 4 * nc0;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int i = 0; i < NST; ++i)
      st[i] = *reinterpret_cast<const f32x4*>(W + off + (int64_t)n_drow(i) * ld + 4 * n_dchunk(i));
  };
  auto store_n = [&](int buf, const f32x4 (&st)[NST]) {
    float* dst = &lds[buf][4 * (nc0 * 32 + nr0)];
#pragma unroll
    for (int i = 0; i < NST; ++i) *reinterpret_cast<f32x4*>(dst + 4 * (n_dchunk(i) * 32 + n_drow(i))) = st[i];
  };

  const int nS1 = (p.n_pass + CPL_BK - 1) / CPL_BK;
  const int nS3 = (p.n_trans + 31) / 32;

  // Accumulators start at the layer's bias (bias + sum_k): the bias loads are issued at kernel start and
  // land under the first weight stages; loading them between the phases exposed a memory latency (and,
  // at the register limit, spills) while both co-resident blocks sat idle -- 16 % of the kernel.
  // X[ht][t] of lane (j, g) is hidden unit ht*16 + 4g + t of batch row j.
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
    for (int q = 0; q < 2; ++q) dst[q] = *reinterpret_cast<const f32x4*>(zrow + min(k0 + 16 * q + 4 * lg, p.n_pass - 4));
  };
  auto finish_z = [&](int k0, f32x4 (&dst)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) dst[q] = (k0 + 16 * q + 4 * lg < p.n_pass) ? dst[q] : zero4;
  };

  // One 16-k step over all hidden tiles: X[ht] += Wfrag(ht) (x) B.  Tiles go in pairs (two
  // independent accumulators back to back: 16x16x4 has a 40-cycle dependent latency on a 32-cycle
  // issue) and the fragments of the next pair are read while the current pair multiplies.
  auto mfma_block = [&](const float* wq, f32x4 (&X)[T], const f32x4 bop) {
    f32x4 a[T];
#pragma unroll
    for (int ht = 0; ht < T; ++ht) a[ht] = *reinterpret_cast<const f32x4*>(wq + 4 * (ht * 16));
#pragma unroll
    for (int ht = 0; ht < T; ht += 2) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        X[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ht][t], bop[t], X[ht], 0, 0, 0);
        X[ht + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ht + 1][t], bop[t], X[ht + 1], 0, 0, 0);
      }
    }
    // schedule: 4 fragment reads up front, then {8 MFMA, 2 reads} -- reads run two tile pairs ahead
    __builtin_amdgcn_sched_group_barrier(0x100, (T >= 4) ? 4 : T, 0);
#pragma unroll
    for (int i = 0; i < T / 2 - 2; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
  };
  // first stage of the phase that follows hidden layer `l` input (compile-time kind)
  auto issue_after_input = [&](f32x4 (&s_)[NST]) {
    if (NH >= 2) issue_k(p.W_hid[0], p.ldw_hid[0], 0, s_); else issue_n(p.W_out, p.ldw_out, 0, s_);
  };
  auto store_after_input = [&](int buf, const f32x4 (&s_)[NST]) {
    if (NH >= 2) store_k(buf, s_); else store_n(buf, s_);
  };

  int g = 0;                                            // stage counter: buffer = g & 1
  issue_k(p.W_in, p.ldw_in, 0, st);
  issue_z(0, zc);
  store_k(0, st);
  finish_z(0, zc);
  __syncthreads();

#ifdef USF_STAMP
#define CSTAMP(v) unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define CSTAMP(v)
#endif
  CSTAMP(c0);
  // ================= phase 1: X1[h][row] += W_in[h][k] * z[row][k] ============================
  auto phase1_compute = [&](int buf, int s) {
    const float* wl = &lds[buf][4 * (lg * HP + lj)];
    const int nq = min(2, (p.n_pass - s * CPL_BK + 15) / 16);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q < nq) mfma_block(wl + 4 * (q * 4 * HP), X1, zc[q]);     // uniform, per 16-k step
    }
  };
  for (int s = 0; s + 1 < nS1; ++s, ++g) {
    const int buf = g & 1;
    issue_k(p.W_in, p.ldw_in, (s + 1) * CPL_BK, st);
    issue_z((s + 1) * CPL_BK, zn);
    __builtin_amdgcn_sched_barrier(0);
    phase1_compute(buf, s);
    __builtin_amdgcn_sched_barrier(0);
    store_k(buf ^ 1, st);
    finish_z((s + 1) * CPL_BK, zn);
#pragma unroll
    for (int q = 0; q < 2; ++q) zc[q] = zn[q];
    __syncthreads();
  }
  {   // last k-slab of W_in: prefetch the first stage of the next phase
    const int buf = g & 1;
    issue_after_input(st);
    __builtin_amdgcn_sched_barrier(0);
    phase1_compute(buf, nS1 - 1);
    __builtin_amdgcn_sched_barrier(0);
    store_after_input(buf ^ 1, st);
    __syncthreads();
    ++g;
  }

  // (context branch +) activation on an accumulator array, in registers
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
        float v = X[ht][t];                                 // layers[0](x): bias already inside
        if (with_ctx) v = v + (cv * wc[t] + bc[t]);         // + layers[1](context), networks.py:741-743
        X[ht][t] = act_apply(v, p.act, p.slope);
      }
    }
  };
  CSTAMP(c1);
  ctx_act(X1, p.ctx != nullptr);
  CSTAMP(c2);

  // ================= phase 2: Xout[h2][row] += W_h[h2][h1] * Xin[h1][row] ====================
  auto hidden_layer = [&](f32x4 (&Xin)[T], f32x4 (&Xout)[T], int l) {
#pragma unroll
    for (int ks = 0; ks < T / 2; ++ks) {
      const int buf = g & 1;
      const bool last = (ks + 1 == T / 2);
      const bool next_is_hidden = (l + 2 < NH);
      if (!last) issue_k(p.W_hid[l], p.ldw_hid[l], (ks + 1) * CPL_BK, st);
      else if (next_is_hidden) issue_k(p.W_hid[l + 1], p.ldw_hid[l + 1], 0, st);
      else issue_n(p.W_out, p.ldw_out, 0, st);
      __builtin_amdgcn_sched_barrier(0);
      const float* wl = &lds[buf][4 * (lg * HP + lj)];
#pragma unroll
      for (int q = 0; q < 2; ++q) mfma_block(wl + 4 * (q * 4 * HP), Xout, Xin[2 * ks + q]);
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

  CSTAMP(c3);
  // ================= phase 3: T[row][n] = sum_h Xlast[h][row] * W_out[n][h]; residual ==========
  // two 16-wide n tiles per stage, interleaved (16x16x4 needs 2 independent accumulators)
  auto output_layer = [&](f32x4 (&X)[T]) {
    for (int nt = 0; nt < nS3; ++nt, ++g) {
      const int buf = g & 1;
      issue_n(p.W_out, p.ldw_out, min(nt + 1, nS3 - 1) * 32, st);
      // output tile u (16 features) of this stage: lane (j, g) ends up with features 4g..4g+3 of row j
      const int orow = min(wrow0 + lj, p.M - 1);
      int col[2];
      f32x4 res[2], bo[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        col[u] = nt * 32 + u * 16 + 4 * lg;
        // rows of z are padded to 4 floats and n_trans segments start 16-B aligned: a float4 never straddles the buffer
        res[u] = *reinterpret_cast<const f32x4*>(p.z + (int64_t)orow * p.ldz + p.off_trans + min(col[u], p.n_trans4 - 4));
        bo[u] = *reinterpret_cast<const f32x4*>(p.b_out + col[u]);     // b_out is padded to 32 * nS3
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc0 = zero4, acc1 = zero4;
      const float* wl = &lds[buf][4 * (lg * 32 + lj)];
      f32x4 b0[T], b1[T];
#pragma unroll
      for (int kt = 0; kt < T; ++kt) {
        b0[kt] = *reinterpret_cast<const f32x4*>(wl + 4 * (kt * 4 * 32));
        b1[kt] = *reinterpret_cast<const f32x4*>(wl + 4 * (kt * 4 * 32 + 16));
      }
#pragma unroll
      for (int kt = 0; kt < T; ++kt) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[kt][t], X[kt][t], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[kt][t], X[kt][t], acc1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int i = 0; i < T - 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      __builtin_amdgcn_sched_barrier(0);
      store_n(buf ^ 1, st);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 a = (u == 0) ? acc0 : acc1;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = res[u][r] + p.sign * (a[r] + bo[u][r]);
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
#ifdef USF_STAMP
  CSTAMP(c4);
  if (p.dbg && lane == 0) {
    unsigned long long* o = p.dbg + (size_t)((blockIdx.x % 1024) * 4 + wave) * 8;
    o[0] = c1 - c0; o[1] = c2 - c1; o[2] = c3 - c2; o[3] = c4 - c3; o[4] = c4 - c0; o[5] = 1;
  }
#endif
}

#ifdef USF_STAMP
unsigned long long* g_cdbg = nullptr;
#endif

static int padded_width(int h) { return h <= 64 ? 64 : (h <= 128 ? 128 : 256); }

int coupling_padded_width(int h) { return (h < 1 || h > CPL_HMAX) ? -1 : padded_width(h); }

int coupling_max_width() { return CPL_HMAX; }

bool coupling_bf16x3_eligible(const usf_coupling_desc* d);
int coupling_bf16x3_dispatch(const usf_coupling_desc* d, hipStream_t stream);
bool coupling_tiny_eligible(const usf_coupling_desc* d);       // usf_coupling_tiny.hip
int coupling_tiny_dispatch(const usf_coupling_desc* d, hipStream_t stream);

int coupling_variant(const usf_coupling_desc* d) {
  if (!d) return 0;
  if (d->M >= 0 && d->n_pass > 0 && d->n_trans > 0 && d->n_hidden >= 1 && d->n_hidden <= 3 && coupling_tiny_eligible(d)) return 3;
  return coupling_bf16x3_eligible(d) ? 2 : 1;
}

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
  if (d->act != USF_ACT_NONE && d->act != USF_ACT_LEAKY_RELU && d->act != USF_ACT_GATE) { set_error("usf_coupling_additive_f32: bad act"); return -2; }
  // tiny layers at launch-bound batches (M <= 256: disjoint from the bf16x3 kernel's M >= 1024): scalar accesses, no alignment rules
  if (coupling_tiny_eligible(d)) return coupling_tiny_dispatch(d, stream);
  if ((d->n_pass & 3) || (d->off_pass & 3) || (d->off_trans & 3) || (d->ldz & 3) || d->off_trans + ((d->n_trans + 3) / 4) * 4 > d->ldz || (d->ldw_in & 3) || (d->ldw_out & 3) || !aligned16(d->z) ||
      !aligned16(d->W_in) || !aligned16(d->W_out) || !aligned16(d->b_in)) {
    set_error("usf_coupling_additive_f32: n_pass/off_pass/ldz/ldw must be multiples of 4 and pointers 16-byte aligned");
    return -2;
  }
  if (d->act != USF_ACT_NONE && d->act != USF_ACT_LEAKY_RELU && d->act != USF_ACT_GATE) { set_error("usf_coupling_additive_f32: bad act"); return -2; }
  if (coupling_bf16x3_eligible(d)) return coupling_bf16x3_dispatch(d, stream);
  if (d->hidden_out[0] || d->act == USF_ACT_GATE) { set_error("usf_coupling_additive_f32: hidden_out / USF_ACT_GATE are served by the bf16x3 kernel (split planes, hidden width in (128, 256], M >= 1024) and by the tiny-layer kernel (M <= 256, segments <= 128, hidden <= 64) only"); return -2; }
  CplArgs a;
  a.z = d->z; a.out = d->out; a.ldz = d->ldz;
  a.M = (int)d->M; a.off_pass = (int)d->off_pass; a.n_pass = (int)d->n_pass; a.off_trans = (int)d->off_trans; a.n_trans = (int)d->n_trans; a.n_trans4 = (int)((d->n_trans + 3) / 4 * 4);
  int hmax = 0;
  for (int i = 0; i < d->n_hidden; ++i) {
    if (d->hidden[i] < 1 || d->hidden[i] > CPL_HMAX) {
      set_error("usf_coupling_additive_f32: hidden width %d must be in [1, %d]", d->hidden[i], CPL_HMAX);
      return -2;
    }
    hmax = d->hidden[i] > hmax ? d->hidden[i] : hmax;
  }
  const int hp = padded_width(hmax);
  const int64_t kp = ((d->n_pass + 31) / 32) * 32;
  if (d->ldw_in < kp || d->ldw_out < hp) {
    set_error("usf_coupling_additive_f32: padding contract violated (ldw_in %lld < %lld or ldw_out %lld < %d)",
              (long long)d->ldw_in, (long long)kp, (long long)d->ldw_out, hp);
    return -2;
  }
  a.W_in = d->W_in; a.ldw_in = d->ldw_in; a.b_in = d->b_in;
  for (int i = 0; i < 2; ++i) {
    const bool used = i + 1 < d->n_hidden;
    a.W_hid[i] = used ? d->W_hid[i] : d->W_in;
    a.b_hid[i] = used ? d->b_hid[i] : d->b_in;
    a.ldw_hid[i] = used ? d->ldw_hid[i] : d->ldw_in;
    if (used && (!d->W_hid[i] || !d->b_hid[i] || (d->ldw_hid[i] & 3) || d->ldw_hid[i] < hp || !aligned16(d->W_hid[i]) ||
                 !aligned16(d->b_hid[i]))) {
      set_error("usf_coupling_additive_f32: bad hidden layer %d", i);
      return -2;
    }
  }
  a.W_out = d->W_out; a.ldw_out = d->ldw_out; a.b_out = d->b_out;
  a.ctx = d->context; a.W_ctx = d->W_ctx; a.b_ctx = d->b_ctx;
  if (a.ctx && (!a.W_ctx || !a.b_ctx || !aligned16(a.b_ctx) || !aligned16(a.W_ctx))) { set_error("usf_coupling_additive_f32: context needs W_ctx and b_ctx"); return -1; }
  a.sign = d->sign; a.slope = d->slope; a.act = d->act;
  a.dbg = nullptr;
#ifdef USF_STAMP
  a.dbg = g_cdbg;
#endif
  const dim3 grid((unsigned)((d->M + CPL_ROWS - 1) / CPL_ROWS)), block(256);
#define USF_CPL(NHV, TV) hipLaunchKernelGGL((coupling_kernel<NHV, TV>), grid, block, 0, stream, a)
#define USF_CPL_T(NHV) do { if (hmax <= 64) USF_CPL(NHV, 4); else if (hmax <= 128) USF_CPL(NHV, 8); else USF_CPL(NHV, 16); } while (0)
  switch (d->n_hidden) {
    case 1: USF_CPL_T(1); break;
    case 2: USF_CPL_T(2); break;
    default: USF_CPL_T(3); break;
  }
#undef USF_CPL_T
#undef USF_CPL
  return check_launch("usf_coupling_additive_f32");
}

}  // namespace usf
