// usf_linear_f32: fused dense layer  C = epi( pro(A) @ W^T )  on exact-f32 MFMA (gfx950).
//
// Design (DESIGN.md section "linear kernel"):
//  * v_mfma_f32_32x32x2_f32: MFMA-A = activations (i = batch row), MFMA-B = weights (j = output
//    feature), so the accumulator has the output feature on the lane -> coalesced C rows.
//  * k-permutation: for every 8 consecutive k, lane (i, h=lane>>5) owns k = 8q+4h .. 8q+4h+3 and
//    feeds them to 4 successive MFMAs.  Both operands are K-contiguous in memory (torch Linear
//    layout [N,K]; activations [M,K]), so every fragment is ONE 16-byte access per lane.
//  * waves tile M only (WM x 1): a wave owns its rows, so activation fragments go global ->
//    registers directly (prefetched one K-slab ahead, prologue applied in registers); only the
//    weight slab, shared by all waves, is staged through LDS (register-staged, double-buffered,
//    rows padded to an odd number of 16-B slots => conflict-free ds_read_b128).
//  * XCD-aware block map: the column-blocks of one row panel run on one XCD so the panel's
//    activations are served from that XCD's L2 after the first touch.
#include "usf_common.h"

namespace usf {

struct LinArgs {
  const float* A; const float* W; const float* bias; const float* pre_div; const float* pre_sub;
  const float* residual; const float* addend; const float* post_mul; float* C;
  int64_t lda, ldw, ldr, ldadd, ldc;
  int M, N, K;
  int nbm, nbn, vec_ok, bias_vec;
  float res_sign, slope;
  int act;
};

template <int TM, int TN, int WM, int BK, bool PRO, int EPI>
__global__ __launch_bounds__(WM * 64, 2) void linear_kernel(const LinArgs p) {
  constexpr int NT = WM * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = TN * 32;
  constexpr int LDS_LD = BK + 4;            // odd number of 16-B slots per row
  constexpr int QS = BK / 8;                // 8-k steps per slab
  constexpr int WCH = BK / 4;               // float4 chunks per weight row per slab
  constexpr int NWV = (BN * WCH + NT - 1) / NT;  // float4 staged per thread per slab
  static_assert((LDS_LD / 4) % 2 == 1, "row stride must be an odd number of 16-B slots");

  constexpr int BN_LDS = (NWV * NT) / WCH;   // >= BN: every thread stages exactly NWV float4, no guards
  static_assert((NWV * NT) % WCH == 0 && BN_LDS >= BN, "staging shape");
  __shared__ __attribute__((aligned(16))) float lds[2][BN_LDS * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;

  // ---- block -> (row panel, column block); blocks b and b+8 share an XCD -----------------
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int seq = bid >> 3;
  const int panel = (seq / p.nbn) * 8 + xcd;
  const int bn = seq % p.nbn;
  if (panel >= p.nbm) return;               // uniform per block: no barrier was reached yet
  const int row0 = panel * BM + wave * (TM * 32);
  const int n0 = bn * BN;

  // ---- per-lane activation fragment sources -----------------------------------------------
  // All global loads below are UNCONDITIONAL (clamped addresses + selects): a branch around a
  // load makes hipcc drain vmcnt(0) in front of the MFMA block and the prefetch is lost.
  const float* aptr[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int r = min(row0 + tm * 32 + li, p.M - 1);     // rows >= M: valid garbage, never stored
    aptr[tm] = p.A + (int64_t)r * p.lda;
  }
  const float* pdiv = p.pre_div ? p.pre_div : p.pre_sub;   // PRO only: at least one is non-null
  const float* psub = p.pre_sub ? p.pre_sub : p.pre_div;
  const bool has_div = p.pre_div != nullptr, has_sub = p.pre_sub != nullptr;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};

  // Loads are split in two halves so that the ISSUE can be pinned in front of the MFMA block
  // (sched_barrier) and everything that consumes the data (selects, prologue math, LDS stores)
  // behind it: left alone, hipcc sinks the loads to the end of the block and waits for them there.
  constexpr int NPRO = PRO ? QS : 1;
  auto issue_a = [&](int k0, f32x4 (&dst)[TM][QS], f32x4 (&dv)[NPRO], f32x4 (&sv)[NPRO]) {
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const int kc = min(k0 + 8 * q + 4 * lh, p.K - 4);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) dst[tm][q] = *reinterpret_cast<const f32x4*>(aptr[tm] + kc);
      if (PRO) {
        dv[q] = *reinterpret_cast<const f32x4*>(pdiv + kc);
        sv[q] = *reinterpret_cast<const f32x4*>(psub + kc);
      }
    }
  };
  auto finish_a = [&](int k0, f32x4 (&dst)[TM][QS], const f32x4 (&dv)[NPRO], const f32x4 (&sv)[NPRO]) {
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const bool kin = (k0 + 8 * q + 4 * lh) < p.K;
      if (PRO) {
        const f32x4 d = has_div ? dv[q] : one4;
        const f32x4 sb = has_sub ? sv[q] : zero4;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) dst[tm][q] = dst[tm][q] / d - sb;
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) dst[tm][q] = kin ? dst[tm][q] : zero4;
    }
  };

  // ---- weight slab staging (global -> regs -> LDS) ----------------------------------------
  auto issue_w = [&](int k0, f32x4 (&dst)[NWV]) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / WCH;
      const int c = idx % WCH;
      const int n = min(n0 + r, p.N - 1);                  // columns >= N: garbage, never stored
      dst[i] = *reinterpret_cast<const f32x4*>(p.W + (int64_t)n * p.ldw + min(k0 + 4 * c, p.K - 4));
    }
  };
  auto store_w = [&](int buf, int k0, const f32x4 (&src)[NWV]) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / WCH;
      const int c = idx % WCH;
      *reinterpret_cast<f32x4*>(&lds[buf][r * LDS_LD + 4 * c]) = (k0 + 4 * c < p.K) ? src[i] : zero4;
    }
  };

  // MFMA-A = weights (i = output feature), MFMA-B = activations (j = batch row): the accumulator holds
  // C^T -- batch row = lane&31, output feature = (r&3) + 8*(r>>2) + 4*(lane>>5) -- i.e. every group of 4
  // registers is 4 consecutive output features of one row.  Accumulators start at the bias (all bias
  // loads of the tile in flight together, landing under the first slab).
  const bool has_bias = p.bias != nullptr, has_pm = p.post_mul != nullptr;
  const float* biasp = has_bias ? p.bias : p.W;           // any valid address
  const float* pmp = has_pm ? p.post_mul : p.W;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int col = n0 + tn * 32 + 8 * g4 + 4 * lh;
      f32x4 bv = zero4;
      if (has_bias) {
        if (p.bias_vec) {                                // uniform: bias 16-B aligned and N % 4 == 0
          bv = *reinterpret_cast<const f32x4*>(biasp + min(col, p.N - 4));
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[j] = biasp[min(col + j, p.N - 1)];
        }
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tm][tn][4 * g4 + j] = bv[j];
    }

  f32x4 a_cur[TM][QS], a_nxt[TM][QS];
  f32x4 dvr[NPRO], svr[NPRO];
  f32x4 wst[NWV];

  const int nslab = (p.K + BK - 1) / BK;
  issue_w(0, wst);
  issue_a(0, a_cur, dvr, svr);
  finish_a(0, a_cur, dvr, svr);
  store_w(0, 0, wst);
  __syncthreads();

  const float* wl0 = &lds[0][li * LDS_LD + 4 * lh];
  auto compute = [&](int buf, int qn) {
    const float* wl = wl0 + buf * (BN_LDS * LDS_LD);
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      if (q < qn) {
        f32x4 b[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          b[tn] = *reinterpret_cast<const f32x4*>(wl + tn * 32 * LDS_LD + 8 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[tn][t], a_cur[tm][q][t], acc[tm][tn], 0, 0, 0);
      }
    }
  };

  // steady state: every slab but the last prefetches its successor (no branch around the loads)
  for (int s = 0; s + 1 < nslab; ++s) {
    const int buf = s & 1;
    const int k1 = (s + 1) * BK;
    issue_w(k1, wst);
    issue_a(k1, a_nxt, dvr, svr);
    __builtin_amdgcn_sched_barrier(0);
    compute(buf, QS);
    __builtin_amdgcn_sched_barrier(0);
    store_w(buf ^ 1, k1, wst);
    finish_a(k1, a_nxt, dvr, svr);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int q = 0; q < QS; ++q) a_cur[tm][q] = a_nxt[tm][q];
    __syncthreads();
  }
  // last slab: skip the all-zero 8-k steps of the K tail (uniform)
  compute((nslab - 1) & 1, (p.K - (nslab - 1) * BK + 7) / 8);

  // ---- epilogue -------------------------------------------------------------------------------
  int row0e = row0;                       // opaque copy: keeps the epilogue's address math from being
  asm volatile("" : "+v"(row0e));         // hoisted above the K loop (it would live across it and spill)
  const float* extra = (EPI == 1) ? p.residual : p.addend;
  const int64_t ldx = (EPI == 1) ? p.ldr : p.ldadd;
  if (EPI == 0 && p.vec_ok) {
    // no per-element loads: transpose each 32x32 tile through a per-wave LDS scratch (the weight
    // staging buffers are dead by now) so that one store instruction writes 8 rows x 128 contiguous
    // bytes = whole cache lines (the store issue rate bounds this phase)
    constexpr int TLD = 36;               // odd number of 16-B slots per scratch row
    static_assert(2 * BN_LDS * LDS_LD >= WM * 32 * TLD, "transpose scratch must fit the staging buffers");
    float* tw = &lds[0][0] + wave * (32 * TLD);
    const int rr = lane >> 3, cc = 4 * (lane & 7);
    __syncthreads();                      // every wave is done reading the last weight slab
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int col = n0 + tn * 32 + 8 * g4 + 4 * lh;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = act_apply(acc[tm][tn][4 * g4 + j], p.act, p.slope);
            if (has_pm) x = x * pmp[min(col + j, p.N - 1)];
            v[j] = x;
          }
          *reinterpret_cast<f32x4*>(tw + li * TLD + 8 * g4 + 4 * lh) = v;       // [row = lane&31][col in tile]
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                    // same wave wrote and reads: no barrier needed
          const int r = rr + 8 * i;
          const f32x4 v = *reinterpret_cast<const f32x4*>(tw + r * TLD + cc);
          const int row = row0e + tm * 32 + r;
          const int col = n0 + tn * 32 + cc;
          float* dst = p.C + (int64_t)row * p.ldc + col;
          if (row < p.M) {
            if (col + 3 < p.N) {
              *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (col + j < p.N) dst[j] = v[j];
            }
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = row0e + tm * 32 + li;
    const int rowc = min(row, p.M - 1);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int col = n0 + tn * 32 + 8 * g4 + 4 * lh;
        float pm[4], ex[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c2 = min(col + j, p.N - 1);
          pm[j] = has_pm ? pmp[c2] : 1.f;
          ex[j] = (EPI != 0) ? extra[(int64_t)rowc * ldx + c2] : 0.f;
        }
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = acc[tm][tn][4 * g4 + j];
          if (EPI == 2 && p.act == USF_ACT_GATE) x = gate_apply(x, ex[j], p.slope);
          else {
            if (EPI == 2) x = x + ex[j];
            x = act_apply(x, p.act, p.slope);
          }
          if (EPI == 1) x = ex[j] + p.res_sign * x;
          v[j] = x * pm[j];
        }
        float* dst = p.C + (int64_t)row * p.ldc + col;
        if (row < p.M) {
          if (p.vec_ok && col + 3 < p.N) {
            *reinterpret_cast<f32x4*>(dst) = v;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (col + j < p.N) dst[j] = v[j];
          }
        }
      }
    }
  }
}

template <int TM, int TN, int WM, int BK>
static int launch_linear(const LinArgs& a0, hipStream_t stream) {
  const bool pro = a0.pre_div != nullptr || a0.pre_sub != nullptr;
  LinArgs a = a0;
  constexpr int BM = WM * TM * 32, BN = TN * 32;
  a.nbm = (a.M + BM - 1) / BM;
  a.nbn = (a.N + BN - 1) / BN;
  const int64_t panels8 = ((int64_t)(a.nbm + 7) / 8) * 8;
  const int64_t grid = panels8 * a.nbn;
  if (grid > 0x7fffffffLL) { set_error("usf_linear_f32: grid too large"); return -3; }
  const int epi = a.residual ? 1 : (a.addend ? 2 : 0);
  const dim3 g((unsigned)grid), b(WM * 64);
#define USF_LIN(P, E) hipLaunchKernelGGL((linear_kernel<TM, TN, WM, BK, P, E>), g, b, 0, stream, a)
  if (pro) {
    if (epi == 0) USF_LIN(true, 0); else if (epi == 1) USF_LIN(true, 1); else USF_LIN(true, 2);
  } else {
    if (epi == 0) USF_LIN(false, 0); else if (epi == 1) USF_LIN(false, 1); else USF_LIN(false, 2);
  }
#undef USF_LIN
  return check_launch("usf_linear_f32");
}

int split_planes(const float* X, int64_t ldx, int64_t M, int64_t N, void* P, int64_t ldp, int64_t plane_stride, hipStream_t stream);
bool linear_bf16x3_eligible(const usf_linear_desc* d);
int linear_bf16x3_dispatch(const usf_linear_desc* d, hipStream_t stream);
bool linear_skinny_eligible(const usf_linear_desc* d);
int linear_skinny_dispatch(const usf_linear_desc* d, hipStream_t stream);

int linear_bf16x3_variant(int M, int N);

// which kernel family / instantiation usf_linear_f32 would launch for this descriptor (no launch): 1000 = small-batch
// kernel (usf_linear_skinny.hip), 2000 + 100 TM + 10 TN + WM = exact-f32 tile, 3000 + 100 TN + 10 WM + NB = bf16x3 tile
int linear_variant(const usf_linear_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  if (linear_skinny_eligible(d)) return 1000;
  if (linear_bf16x3_eligible(d)) return linear_bf16x3_variant((int)d->M, (int)d->N);
  if (d->M <= 64 || d->N <= 64) return 2122;
  const int pad160 = (((int)d->N + 159) / 160) * 160 - (int)d->N;
  const int pad128 = (((int)d->N + 127) / 128) * 128 - (int)d->N;
  return pad160 < pad128 ? 2254 : 2244;
}

int linear_dispatch(const usf_linear_desc* d, hipStream_t stream) {
  if (!d) { set_error("usf_linear_f32: null descriptor"); return -1; }
  if (d->M < 0 || d->N <= 0 || d->K <= 0 || d->M > 0x7fffffff || d->N > 0x7fffffff || d->K > 0x7fffffff) {
    set_error("usf_linear_f32: bad sizes M=%lld N=%lld K=%lld", (long long)d->M, (long long)d->N, (long long)d->K);
    return -2;
  }
  if (d->M == 0) return 0;
  if (!d->A || !d->W || !d->C) { set_error("usf_linear_f32: null A/W/C"); return -1; }
  if ((d->K & 3) || (d->lda & 3) || (d->ldw & 3) || d->lda < d->K || d->ldw < d->K || d->ldc < d->N ||
      (d->residual && d->ldr < d->N) || (d->addend && d->ldadd < d->N)) {
    set_error("usf_linear_f32: K/lda/ldw must be multiples of 4 and strides >= extents "
              "(K=%lld lda=%lld ldw=%lld ldc=%lld)", (long long)d->K, (long long)d->lda, (long long)d->ldw,
              (long long)d->ldc);
    return -2;
  }
  if (!aligned16(d->A) || !aligned16(d->W) || (d->pre_div && !aligned16(d->pre_div)) ||
      (d->pre_sub && !aligned16(d->pre_sub))) {
    set_error("usf_linear_f32: A/W/pre_div/pre_sub must be 16-byte aligned");
    return -2;
  }
  if (d->act != USF_ACT_NONE && d->act != USF_ACT_LEAKY_RELU && d->act != USF_ACT_GATE) { set_error("usf_linear_f32: bad act"); return -2; }
  if (d->act == USF_ACT_GATE && (!d->addend || d->residual)) { set_error("usf_linear_f32: USF_ACT_GATE reads the gate from `addend` (required) and takes no residual"); return -2; }
  if (d->residual && d->addend) { set_error("usf_linear_f32: residual and addend are mutually exclusive"); return -2; }
  if (d->A_planes_out) {
    const int64_t rows_pad = (d->M + 31) / 32 * 32, kp = (d->K + 31) / 32 * 32;
    if (d->pre_div || d->pre_sub) { set_error("usf_linear_f32: A_planes_out takes no pre_div / pre_sub"); return -2; }
    if (d->ldp_out < kp || (d->ldp_out & 7) || (d->planes_out_stride & 7) || d->planes_out_stride < rows_pad * d->ldp_out ||
        !aligned16(d->A_planes_out) || 3 * d->planes_out_stride * 2 >= (1LL << 31)) {
      set_error("usf_linear_f32: A_planes_out needs ldp_out >= ceil32(K), a multiple of 8, planes_out_stride >= ceil32(M) * ldp_out "
                "and planes below 2 GiB");
      return -2;
    }
    if (linear_skinny_eligible(d) || !linear_bf16x3_eligible(d)) {
      // the kernels that do not split their operand: the planes come from a pass of their own
      const int rc = split_planes(d->A, d->lda, d->M, d->K, d->A_planes_out, d->ldp_out, d->planes_out_stride, stream);
      if (rc) return rc;
    }
  }
  if (linear_skinny_eligible(d)) return linear_skinny_dispatch(d, stream);     // small batches: latency, not FLOPs
  if (linear_bf16x3_eligible(d)) return linear_bf16x3_dispatch(d, stream);
  LinArgs a;
  a.A = d->A; a.W = d->W; a.bias = d->bias; a.pre_div = d->pre_div; a.pre_sub = d->pre_sub;
  a.residual = d->residual; a.addend = d->addend; a.post_mul = d->post_mul; a.C = d->C;
  a.lda = d->lda; a.ldw = d->ldw; a.ldr = d->ldr; a.ldadd = d->ldadd; a.ldc = d->ldc;
  a.M = (int)d->M; a.N = (int)d->N; a.K = (int)d->K;
  a.nbm = a.nbn = 0;
  a.vec_ok = ((d->ldc & 3) == 0 && aligned16(d->C)) ? 1 : 0;   // 16-byte row stores possible
  a.bias_vec = (d->bias && aligned16(d->bias) && (d->N & 3) == 0) ? 1 : 0;
  a.res_sign = d->res_sign; a.slope = d->slope; a.act = d->act;

  // tile choice: 256-row panels; 160- or 128-wide column blocks, whichever pads N less
  // (784 -> 5 x 160 = 800; 256 -> 2 x 128). Small problems take the 64x64 tile.
  if (a.M <= 64 || a.N <= 64) return launch_linear<1, 2, 2, 16>(a, stream);
  const int pad160 = ((a.N + 159) / 160) * 160 - a.N;
  const int pad128 = ((a.N + 127) / 128) * 128 - a.N;
  if (pad160 < pad128) return launch_linear<2, 5, 4, 16>(a, stream);
  return launch_linear<2, 4, 4, 16>(a, stream);
}

}  // namespace usf
