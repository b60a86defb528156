// Parameter maps of the image-shaped flows' affine blocks and their backward pass, batched over the blocks of a flow
// (SURVEY rows N2 / N4).  A BlockAffineTransform of an image flow carries a C x C matrix (C = channels <= 64) built from
// an LUTransform, optionally followed by a HouseholderTransform (SequentialAffineTransform, transforms.py:1457-1476):
//     L = tril(L_raw, -1) + I,  U = triu(U_raw)                                  transforms.py:1271-1281
//     M_lu = L U,  M_lu^-1 = U^-1 L^-1,  log|det| = sum log|U_jj|                 transforms.py:1283-1320
//     H = w_0 prod_k (I - 2 v_k v_k^T / v_k.v_k)   (w_0 a fixed permutation)      transforms.py:795-809
//     M = M_lu H,  M^-1 = H^T M_lu^-1,  b = b_lu H
// The reference (and the mirror's batched torch formulation) evaluates these with a few dozen torch ops on C x C tensors
// per call, and autograd differentiates them op by op: at the reference's training batch of 32 rows the MNIST model's step
// was ~340 launches of ~4 us of which ~200 were these parameter chains.  Here: ONE launch forward and ONE backward for all
// blocks of equal structure -- a 256-thread block per transform, every matrix in LDS, fp32 (as the reference computes).
//
//   affine_prep_kernel      (L_raw, U_raw, bias, v, w_0) -> (M, M^-1, b, c = -M^-1 b, log|det|) + the factors the backward needs
//   affine_prep_bwd_kernel  (dM, dM^-1, db, dc, dlog|det|) -> (dL_raw, dU_raw, dbias, dv):
//       c = -M^-1 b:       dM^-1 -= dc (x) b,         db -= M^-T dc
//       M = M_lu H:        dM_lu = dM H^T,            dH  = M_lu^T dM
//       M^-1 = H^T M_lu^-1: dM_lu^-1 = H dM^-1,       dH += M_lu^-1 (dM^-1)^T
//       b = b_lu H:        db_lu = H db,              dH += b_lu (x) db
//       M_lu = L U:        dL = dM_lu U^T,            dU = L^T dM_lu
//       M_lu^-1 = U^-1 L^-1: dU -= U^-T (dM_lu^-1 L^-T) U^-T,   dL -= L^-T (U^-T dM_lu^-1) L^-T
//       log|det|:          dU_jj += dlog|det| / U_jj
//       masks:             dL_raw = tril(dL, -1),  dU_raw = triu(dU)      (the reference's gradient hooks: :1262-1268)
//       H = P_{k-1} R_k:   dR_k = P_{k-1}^T dP_k,  dv = -2 (A + A^T) v / s + 4 (v^T A v) v / s^2  (A = dR_k, s = v.v),
//                          dP_{k-1} = dP_k R_k
#include "usf_common.h"

namespace usf {

constexpr int AP_NT = 256;
constexpr int AP_MAXC = 64;
constexpr int AP_SAVE = 7;       // matrices kept for the backward pass: L, U, L^-1, U^-1, M_lu, M_lu^-1, H

struct APArgs {
  const float* Lr; const float* Ur; const float* bias; const float* vk; const float* w0;
  float* M; float* Minv; float* b; float* cvec; float* ladj; float* save;
  int C, nvs;
};

struct APBArgs {
  const float* save; const float* bias; const float* vk; const float* w0;
  const float* Minv; const float* b;            // the forward pass's outputs (for the gradient through c = -Minv b)
  const float* dM; const float* dMinv; const float* db; const float* dc; const float* dladj;
  float* dLr; float* dUr; float* dbias; float* dvk;
  int C, nvs, stage;
};

// O[i][j] (op)= alpha * sum_k opA(A)[i][k] opB(B)[k][j];  ACC 0: set, 1: add.  All threads of the block; no barrier inside.
template <bool TA, bool TB, int ACC>
__device__ __forceinline__ void ap_mm(const float* A, int lda, const float* B, int ldb, float* O, int ldo, int C, float alpha) {
  for (int idx = threadIdx.x; idx < C * C; idx += AP_NT) {
    const int i = idx / C, j = idx - i * C;
    float s = 0.f;
    for (int k = 0; k < C; ++k) {
      const float a = TA ? A[k * lda + i] : A[i * lda + k];
      const float bb = TB ? B[j * ldb + k] : B[k * ldb + j];
      s = fmaf(a, bb, s);
    }
    if (ACC) O[i * ldo + j] += alpha * s;
    else O[i * ldo + j] = alpha * s;
  }
}

// X <- X R with R = I - 2 v v^T / s (a Householder reflection; symmetric):  X -= (2 / s) (X v) v^T.
// `tmp` holds C floats; barriers inside (all threads must call).
__device__ __forceinline__ void ap_reflect(float* X, int ld, const float* v, float* tmp, int C) {
  float s = 0.f;
  for (int k = 0; k < C; ++k) s = fmaf(v[k], v[k], s);
  if ((int)threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < C; ++k) t = fmaf(X[threadIdx.x * ld + k], v[k], t);
    tmp[threadIdx.x] = t;
  }
  __syncthreads();
  const float f = 2.f / s;
  for (int idx = threadIdx.x; idx < C * C; idx += AP_NT) {
    const int i = idx / C, j = idx - i * C;
    X[i * ld + j] -= f * tmp[i] * v[j];
  }
  __syncthreads();
}

__global__ __launch_bounds__(AP_NT) void affine_prep_kernel(const APArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ap_lds[];
  const int C = a.C, ld = C + 1, msz = C * ld, tid = threadIdx.x;
  const int64_t blk = blockIdx.x;
  float* sL = ap_lds;
  float* sU = sL + msz;
  float* sLi = sU + msz;
  float* sUi = sLi + msz;
  float* sMlu = sUi + msz;
  float* sMilu = sMlu + msz;
  float* sH = sMilu + msz;
  float* sv = sH + msz;            // [C] current Householder vector
  float* stmp = sv + AP_MAXC;      // [C]
  const float* Lr = a.Lr + blk * C * C;
  const float* Ur = a.Ur + blk * C * C;
  for (int idx = tid; idx < C * C; idx += AP_NT) {
    const int i = idx / C, j = idx - i * C;
    sL[i * ld + j] = j < i ? Lr[idx] : (i == j ? 1.f : 0.f);
    sU[i * ld + j] = j >= i ? Ur[idx] : 0.f;
    sLi[i * ld + j] = 0.f;
    sUi[i * ld + j] = 0.f;
  }
  __syncthreads();
  // triangular inverses by substitution, one thread per column (wave 0: L^-1, wave 1: U^-1)
  if (tid < C) {
    const int j = tid;
    sLi[j * ld + j] = 1.f;
    for (int i = j + 1; i < C; ++i) {
      float s = 0.f;
      for (int k = j; k < i; ++k) s = fmaf(sL[i * ld + k], sLi[k * ld + j], s);
      sLi[i * ld + j] = -s;
    }
  } else if (tid >= 64 && tid < 64 + C) {
    const int j = tid - 64;
    sUi[j * ld + j] = 1.f / sU[j * ld + j];
    for (int i = j - 1; i >= 0; --i) {
      float s = 0.f;
      for (int k = i + 1; k <= j; ++k) s = fmaf(sU[i * ld + k], sUi[k * ld + j], s);
      sUi[i * ld + j] = -s / sU[i * ld + i];
    }
  }
  __syncthreads();
  ap_mm<false, false, 0>(sL, ld, sU, ld, sMlu, ld, C, 1.f);
  ap_mm<false, false, 0>(sUi, ld, sLi, ld, sMilu, ld, C, 1.f);
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < C; ++i) s += logf(fabsf(sU[i * ld + i]));
    a.ladj[blk] = s;
  }
  __syncthreads();
  float* M = a.M + blk * C * C;
  float* Mi = a.Minv + blk * C * C;
  if (a.nvs > 0) {
    const float* w0 = a.w0 + blk * C * C;
    for (int idx = tid; idx < C * C; idx += AP_NT) sH[(idx / C) * ld + idx % C] = w0[idx];
    __syncthreads();
    for (int q = 0; q < a.nvs; ++q) {
      if (tid < C) sv[tid] = a.vk[(blk * a.nvs + q) * C + tid];
      __syncthreads();
      ap_reflect(sH, ld, sv, stmp, C);
    }
    ap_mm<false, false, 0>(sMlu, ld, sH, ld, M, C, C, 1.f);
    ap_mm<true, false, 0>(sH, ld, sMilu, ld, Mi, C, C, 1.f);
    if (tid < C) {
      float s = 0.f;
      for (int i = 0; i < C; ++i) s = fmaf(a.bias[blk * C + i], sH[i * ld + tid], s);
      a.b[blk * C + tid] = s;
      sv[tid] = s;
    }
  } else {
    for (int idx = tid; idx < C * C; idx += AP_NT) {
      const int i = idx / C, j = idx - i * C;
      M[idx] = sMlu[i * ld + j];
      Mi[idx] = sMilu[i * ld + j];
      sH[i * ld + j] = i == j ? 1.f : 0.f;
    }
    if (tid < C) {
      const float s = a.bias[blk * C + tid];
      a.b[blk * C + tid] = s;
      sv[tid] = s;
    }
  }
  // c = -M^-1 b = -H^T (M_lu^-1 b): the shift of the backward direction y = M^-1 x + c (sH is the identity without a factor)
  __syncthreads();
  if (tid < C) {
    float t = 0.f;
    for (int j = 0; j < C; ++j) t = fmaf(sMilu[tid * ld + j], sv[j], t);
    stmp[tid] = t;
  }
  __syncthreads();
  if (tid < C) {
    float t = 0.f;
    for (int i = 0; i < C; ++i) t = fmaf(sH[i * ld + tid], stmp[i], t);
    a.cvec[blk * C + tid] = -t;
  }
  float* sv_out = a.save + blk * AP_SAVE * C * C;
  for (int m = 0; m < AP_SAVE; ++m) {
    const float* src = ap_lds + m * msz;
    for (int idx = tid; idx < C * C; idx += AP_NT) sv_out[m * C * C + idx] = src[(idx / C) * ld + idx % C];
  }
}

__global__ __launch_bounds__(AP_NT) void affine_prep_bwd_kernel(const APBArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ap_lds[];
  const int C = a.C, ld = C + 1, msz = C * ld, tid = threadIdx.x, CC = C * C;
  const int64_t blk = blockIdx.x;
  float* sdMlu = ap_lds;
  float* sdMilu = sdMlu + msz;
  float* sdH = sdMilu + msz;
  float* sdL = sdH + msz;
  float* sdU = sdL + msz;
  float* sT1 = sdU + msz;
  float* sT2 = sT1 + msz;
  float* sv = sT2 + msz;           // [C]
  float* stmp = sv + AP_MAXC;      // [C]
  float* stmp2 = stmp + AP_MAXC;   // [C]
  const float* sav = a.save + blk * AP_SAVE * CC;
  const float* gL = sav;
  const float* gU = sav + CC;
  const float* gLi = sav + 2 * CC;
  const float* gUi = sav + 3 * CC;
  const float* gMlu = sav + 4 * CC;
  const float* gMilu = sav + 5 * CC;
  const float* gH = sav + 6 * CC;
  float* sdMi = stmp2 + AP_MAXC;   // [C][ld]  dM^-1 with the contribution of dc folded in
  float* sdb = sdMi + msz;         // [C]      db likewise
  const float* dM = a.dM + blk * CC;          // (re-pointed to the LDS copy below when staged)
  // c = -M^-1 b:  dM^-1 -= dc (x) b,  db -= (M^-1)^T dc
  {
    const float* gMi = a.Minv + blk * CC;
    const float* gb = a.b + blk * C;
    const float* dc = a.dc + blk * C;
    for (int idx = tid; idx < CC; idx += AP_NT) {
      const int i = idx / C, j = idx - i * C;
      sdMi[i * ld + j] = a.dMinv[blk * CC + idx] - dc[i] * gb[j];
    }
    if (tid < C) {
      float t = 0.f;
      for (int i = 0; i < C; ++i) t = fmaf(gMi[i * C + tid], dc[i], t);
      sdb[tid] = a.db[blk * C + tid] - t;
    }
  }
  // C <= 48: the eight read-only matrices (the saved factors and dM) fit the LDS next to the temporaries -- every product
  // below then reads LDS instead of global memory (21 blocks of C = 48: 232 -> ~60 us)
  int lg = C;
  if (a.stage) {
    float* st = sdb + AP_MAXC;
    for (int m = 0; m < AP_SAVE + 1; ++m) {
      const float* src = m < AP_SAVE ? sav + m * CC : dM;
      for (int idx = tid; idx < CC; idx += AP_NT) st[m * msz + (idx / C) * ld + idx % C] = src[idx];
    }
    gL = st; gU = st + msz; gLi = st + 2 * msz; gUi = st + 3 * msz; gMlu = st + 4 * msz; gMilu = st + 5 * msz; gH = st + 6 * msz;
    dM = st + 7 * msz;
    lg = ld;
  }
  __syncthreads();
  const float* dMi = sdMi;
  const float* db = sdb;
  const int ldi = ld;
  const float* pMlu = dM;          // dM_lu / dM_lu^-1: the incoming gradients themselves without a Householder factor
  const float* pMilu = dMi;
  int ldg = lg, ldgi = ldi;
  if (a.nvs > 0) {
    ap_mm<false, true, 0>(dM, lg, gH, lg, sdMlu, ld, C, 1.f);          // dM H^T
    ap_mm<false, false, 0>(gH, lg, dMi, ldi, sdMilu, ld, C, 1.f);     // H dM^-1
    ap_mm<true, false, 0>(gMlu, lg, dM, lg, sdH, ld, C, 1.f);          // M_lu^T dM
    __syncthreads();
    ap_mm<false, true, 1>(gMilu, lg, dMi, ldi, sdH, ld, C, 1.f);      // + M_lu^-1 (dM^-1)^T
    __syncthreads();
    for (int idx = tid; idx < CC; idx += AP_NT) {
      const int i = idx / C, j = idx - i * C;
      sdH[i * ld + j] += a.bias[blk * C + i] * db[j];               // + b_lu (x) db
    }
    if (tid < C) {
      float s = 0.f;
      for (int j = 0; j < C; ++j) s = fmaf(gH[tid * lg + j], db[j], s);
      a.dbias[blk * C + tid] = s;                                    // H db
    }
    pMlu = sdMlu; pMilu = sdMilu; ldg = ld; ldgi = ld;
  } else if (tid < C) {
    a.dbias[blk * C + tid] = db[tid];
  }
  __syncthreads();
  // ---- LU factors
  ap_mm<false, true, 0>(pMlu, ldg, gU, lg, sdL, ld, C, 1.f);           // dL = dM_lu U^T
  ap_mm<true, false, 0>(gL, lg, pMlu, ldg, sdU, ld, C, 1.f);           // dU = L^T dM_lu
  ap_mm<false, true, 0>(pMilu, ldgi, gLi, lg, sT1, ld, C, 1.f);        // T1 = dM_lu^-1 L^-T          (= dU^-1)
  __syncthreads();
  ap_mm<true, false, 0>(gUi, lg, sT1, ld, sT2, ld, C, 1.f);            // T2 = U^-T T1
  __syncthreads();
  ap_mm<false, true, 1>(sT2, ld, gUi, lg, sdU, ld, C, -1.f);           // dU -= T2 U^-T
  ap_mm<true, false, 0>(gUi, lg, pMilu, ldgi, sT1, ld, C, 1.f);        // T1 = U^-T dM_lu^-1          (= dL^-1)
  __syncthreads();
  ap_mm<true, false, 0>(gLi, lg, sT1, ld, sT2, ld, C, 1.f);            // T2 = L^-T T1
  __syncthreads();
  ap_mm<false, true, 1>(sT2, ld, gLi, lg, sdL, ld, C, -1.f);           // dL -= T2 L^-T
  __syncthreads();
  const float dl = a.dladj[blk];
  for (int idx = tid; idx < CC; idx += AP_NT) {
    const int i = idx / C, j = idx - i * C;
    float du = sdU[i * ld + j];
    if (i == j) du += dl / gU[i * lg + j];
    a.dLr[blk * CC + idx] = j < i ? sdL[i * ld + j] : 0.f;
    a.dUr[blk * CC + idx] = j >= i ? du : 0.f;
  }
  if (a.nvs <= 0) return;
  // ---- Householder vectors, last to first: dP = dH, P_{q} = w_0 R_0 .. R_{q-1}
  const float* w0 = a.w0 + blk * CC;
  __syncthreads();
  for (int q = a.nvs - 1; q >= 0; --q) {
    // prefix product P_q -> sT1
    for (int idx = tid; idx < CC; idx += AP_NT) sT1[(idx / C) * ld + idx % C] = w0[idx];
    __syncthreads();
    for (int r = 0; r < q; ++r) {
      if (tid < C) sv[tid] = a.vk[(blk * a.nvs + r) * C + tid];
      __syncthreads();
      ap_reflect(sT1, ld, sv, stmp, C);
    }
    if (tid < C) sv[tid] = a.vk[(blk * a.nvs + q) * C + tid];
    ap_mm<true, false, 0>(sT1, ld, sdH, ld, sT2, ld, C, 1.f);         // A = P_q^T dP
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(sv[k], sv[k], s);
    if (tid < C) {
      float av = 0.f, atv = 0.f;
      for (int k = 0; k < C; ++k) {
        av = fmaf(sT2[tid * ld + k], sv[k], av);
        atv = fmaf(sT2[k * ld + tid], sv[k], atv);
      }
      stmp[tid] = av;
      stmp2[tid] = atv;
    }
    __syncthreads();
    if (tid < C) {
      float vav = 0.f;
      for (int k = 0; k < C; ++k) vav = fmaf(sv[k], stmp[k], vav);
      a.dvk[(blk * a.nvs + q) * C + tid] = -2.f * (stmp[tid] + stmp2[tid]) / s + 4.f * vav * sv[tid] / (s * s);
    }
    __syncthreads();
    if (q > 0) ap_reflect(sdH, ld, sv, stmp, C);                      // dP_{q-1} = dP_q R_q
  }
}

static size_t ap_lds_bytes(int C) { return (size_t)((AP_SAVE + 1) * C * (C + 1) + 4 * AP_MAXC) * sizeof(float); }
// backward: the read-only matrices staged in LDS as well when 16 padded matrices fit 160 KB (C <= 48)
static int ap_bwd_stage(int C) { return 2 * ap_lds_bytes(C) <= 160 * 1024 ? 1 : 0; }
static size_t ap_bwd_lds_bytes(int C) { return ap_lds_bytes(C) + (ap_bwd_stage(C) ? (size_t)(AP_SAVE + 1) * C * (C + 1) * sizeof(float) : 0); }

static int ap_check(const char* what, int64_t n, int32_t C, int32_t nvs) {
  if (n < 0 || C < 1 || C > AP_MAXC || nvs < 0 || nvs > 8) {
    set_error("%s: need n >= 0, 1 <= C <= %d, 0 <= nvs <= 8", what, AP_MAXC);
    return -1;
  }
  return 0;
}

int affine_prep(const float* Lr, const float* Ur, const float* bias, const float* vk, const float* w0, int64_t n, int32_t C,
                int32_t nvs, float* M, float* Minv, float* b, float* cvec, float* ladj, float* save, hipStream_t stream) {
  if (ap_check("usf_affine_prep_f32", n, C, nvs)) return -1;
  if (n == 0) return 0;
  if (!Lr || !Ur || !bias || !M || !Minv || !b || !cvec || !ladj || !save || (nvs > 0 && (!vk || !w0))) {
    set_error("usf_affine_prep_f32: bad arguments");
    return -1;
  }
  static bool attr_done[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (!attr_done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&affine_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ap_lds_bytes(AP_MAXC)) != hipSuccess) {
      set_error("usf_affine_prep_f32: cannot reserve LDS");
      return -5;
    }
    attr_done[dev] = true;
  }
  const APArgs a{Lr, Ur, bias, vk, w0, M, Minv, b, cvec, ladj, save, C, nvs};
  affine_prep_kernel<<<(unsigned)n, AP_NT, ap_lds_bytes(C), stream>>>(a);
  return check_launch("usf_affine_prep_f32");
}

int affine_prep_bwd(const float* save, const float* bias, const float* vk, const float* w0, const float* Minv, const float* b,
                    const float* dM, const float* dMinv, const float* db, const float* dc, const float* dladj, int64_t n, int32_t C, int32_t nvs, float* dLr, float* dUr, float* dbias,
                    float* dvk, hipStream_t stream) {
  if (ap_check("usf_affine_prep_bwd_f32", n, C, nvs)) return -1;
  if (n == 0) return 0;
  if (!save || !bias || !Minv || !b || !dM || !dMinv || !db || !dc || !dladj || !dLr || !dUr || !dbias || (nvs > 0 && (!vk || !w0 || !dvk))) {
    set_error("usf_affine_prep_bwd_f32: bad arguments");
    return -1;
  }
  static bool attr_done[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (!attr_done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&affine_prep_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      set_error("usf_affine_prep_bwd_f32: cannot reserve LDS");
      return -5;
    }
    attr_done[dev] = true;
  }
  const int stage = ap_bwd_stage(C);
  const APBArgs a{save, bias, vk, w0, Minv, b, dM, dMinv, db, dc, dladj, dLr, dUr, dbias, dvk, C, nvs, stage};
  affine_prep_bwd_kernel<<<(unsigned)n, AP_NT, ap_bwd_lds_bytes(C), stream>>>(a);
  return check_launch("usf_affine_prep_bwd_f32");
}

}  // namespace usf
