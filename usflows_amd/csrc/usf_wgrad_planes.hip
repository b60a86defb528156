// Weight gradient of F.linear from PRE-SPLIT operands (round 4):
//      G[n,k] = alpha * sum_m Y[m, y_off + n] * A[m, a_off + k] + beta * G[n,k]
// what autograd derives for every F.linear of the flow when Flow.fit differentiates -log_prob(batch).mean()
// (flows.py:196-203; BlockAffineTransform transforms.py:913-962, conditioner Linear layers networks.py:739-751).
//
// usf_wgrad_f32's loader-wave kernel (usf_train.hip) reads fp32 rows and splits every value into its three bf16 planes
// again in each of the blocks that share its rows -- seven times at 784 x 784, 5.25 vector instructions per value beside
// the MFMAs of the same SIMD: that split bound it (r02_tuning_experiments.md section 4).  Here Y and A arrive as the
// planes the layer's own GEMMs already made of them (usf_linear_desc::A_planes_out), row-major bf16
// [3][ceil32(M)][ld], and
//   * the loader waves only COPY: 16-byte buffer loads -> ds_write_b128 into an image that keeps the rows as they
//     are in HBM ([32 batch rows][128 columns] per plane, 16-byte chunk ch of row r at ch ^ (((r & 3) << 2) | ((r >> 2) & 3)));
//   * the MFMA waves read their operand fragments (8 consecutive batch rows of one column per lane) with the
//     transposing read ds_read_b64_tr_b16, two per fragment, conflict-free on that image;
//   * 784 = 6 x 128 + 16: a remainder of up to 16 columns rides in the last 128-wide tile (a 144-wide tile: 16 more
//     columns in a small extension image, a fifth fragment column for one wave pair) instead of making 13 edge tiles of
//     49 -- an edge tile streams a whole 128-column operand that nobody shares with it and costs 0.7 of a full tile
//     (measured) for 1/8 of its products.  Only a 144 x 144 tile does not fit the LDS ring: the tile row of the wide
//     tiles keeps the 16-column edge as a tile of its own;
//   * the grid is ONE block per CU and every block gets one item = (tile, row range): every tile is cut into the same
//     floor(CUs / tiles) row ranges, items are numbered class by class and row range by row range, and each XCD gets an
//     eighth of every class -- the blocks of an XCD work on the same rows and share them in its L2.  (The plain grid of
//     (tile, row range) blocks, 49 x 16 = 784 of them on 256 CUs, left CUs idle beside the last blocks: 3.7 block times of
//     makespan for 2.6 of work.  Row ranges sized by a cost model per tile class -- shorter ranges for the wide tiles,
//     whose fifth fragment column makes them ~1.25 x slower -- were measured too: better balance, but blocks of one XCD
//     then walk different rows and the L2 hit rate pays for it: 0.466 vs 0.438 ms.)
// Same six products per value pair in the same order as usf_wgrad_f32 mode 1; partial sums added in a fixed order
// (bitwise reproducible).  Rows [M, ceil32(M)) of the planes must be zero; columns beyond N / K read whatever follows
// (padding, the next row, zeros beyond the buffer) and only reach outputs that are not stored.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "usf_common.h"

namespace usf {

namespace {

typedef __bf16 wp_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wp_s16x4 __attribute__((ext_vector_type(4)));
typedef short wp_s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wp_u32x4 __attribute__((ext_vector_type(4)));
typedef float wp_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wp_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int WP_T = 128;      // normal tile width
constexpr int WP_S = 32;       // batch rows per slab (one MFMA k-extent)
constexpr int WP_FOLD = 16;    // a remainder of up to this many columns rides in the last normal tile

// tile types per direction: 0 normal (128 wide), 1 wide (128 + remainder), 2 edge (the remainder alone)
struct WpSched {
  int items;
  int per_xcd;                 // grid / 8
  int fN, fK, rN, rK;          // N = 128 fN + rN, K = 128 fK + rK
  int foldN, foldK;            // the remainder rides in the last normal tile
  // classes c = 3 * (type in n) + (type in k); (1, 1) does not exist (the tile row of the wide tiles has no wide tile)
  int T[9], nseg[9], rows[9];
  short start[9][8], cnt[9][8];   // items of class c on XCD x: numbers start[c][x] .. + cnt[c][x] inside the class
};

struct WpArgs {
  const __bf16* Yp; int64_t ldyp, ystride;     // planes [3][rows][ld]: plane stride in elements
  const __bf16* Ap; int64_t ldap, astride;
  int y_col0, a_col0;                          // first column of the operands inside their planes (multiples of 8)
  unsigned ybytes, abytes;                     // readable bytes from Yp / Ap (buffer bounds: beyond reads as zeros)
  float* part;
  int M, N, K;
  unsigned long long* dbg;                     // tuning builds (-DUSF_STAMP) only
  float* cs_part;                              // [row ranges][N] partial column sums of Y, or NULL (see wgrad_planes)
  int y_nkb, y_kb0, a_nkb, a_kb0;              // BLOCKED operands (usf_wgrad_blocked_f32): blocks per panel, first block
  WpSched sched;
};

struct WpShared {
  uint4 img[3][2][3][512];                     // [ring][Y / A][plane][32 rows x 16 chunks, swizzled]
  uint4 ext[3][3][64];                         // [ring][plane][32 rows x 2 chunks]: columns 128 .. 143 of the wide operand
};

__device__ __forceinline__ int wp_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
// extension image: row r (32 bytes) at position r ^ 4 where bit 3 of r is set -- the two 4-row blocks a 32-lane half
// reads (rows 8 g + 4 hh .., g = 0, 1) land in different halves of the 64 banks
__device__ __forceinline__ int wp_ext_row(int r) { return r ^ (((r >> 3) & 1) << 2); }

// ---- BLOCKED operands: the planes buffers of the planes pipeline (include/usflows_hip.h: chunk(panel, block, plane) = 1 KiB,
// line 16 g' + j = row j of the panel, slots 8 g' .. 8 g' + 7; slot 8 g' + u = position 16 (u >> 2) + 4 g' + (u & 3) of the
// block).  A slab's image per plane = 2 panels x 4 blocks x 64 lines, the chunks copied as they are except that line
// 16 g' + j of a chunk sits at 16 g' + (j ^ 4 (g' >> 1)).  A transposing read wants, per 16-lane group, 4 rows x 4 pieces of
// 4 consecutive columns; here piece p of sub-tile tt (0 / 1) of a block is the half (p & 1) ^ tt of group g' = p's line: the
// pieces of a 32-lane half land in 32 different (16-byte bank group, half) places -- conflict-free -- and the sub-tile's
// column c = 4 p + e is position wp_blk_pos(c, tt) of the block.
__device__ __forceinline__ int wp_blk_line(int ph, int blk, int L) {
  const int gq = L >> 4, j = L & 15;
  return (ph * 4 + blk) * 64 + 16 * gq + (j ^ ((gq >> 1) << 2));
}
__device__ __forceinline__ int wp_blk_off(int r, int blk, int pp, int tt) {
  return 16 * wp_blk_line(r >> 4, blk, 16 * pp + (r & 15)) + 8 * ((pp & 1) ^ tt);
}
__host__ __device__ __forceinline__ int wp_blk_pos(int c, int tt) { return 16 * (((c >> 2) & 1) ^ tt) + 4 * (c >> 2) + (c & 3); }

__device__ __forceinline__ wp_bf16x8 wp_frag(const char* tile, int o0, int o1) {
#ifdef USF_WP_X_B128                            // tuning build (wrong results): ONE 16-byte read per fragment instead of two transposing reads
  (void)o1;
  return *reinterpret_cast<const wp_bf16x8*>(tile + (o0 & ~15));
#endif
  typedef __attribute__((address_space(3))) wp_s16x4 lds_v;
  const wp_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(tile + o0));
  const wp_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(tile + o1));
  const wp_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(wp_bf16x8, v);
}

// item of this block: tile origin / widths, row range, partial slot
__device__ __forceinline__ bool wp_item(const WpSched& sc, int M, int& n0, int& k0, int& wn, int& wk, int& m_begin, int& m_end, int& seg) {
  const int b = blockIdx.x;
  const int x = b & 7;
  int slot = b >> 3, cls = 0, le = -1;
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const int n = sc.cnt[c][x];
    if (le < 0) {
      if (slot < n) { le = sc.start[c][x] + slot; cls = c; }
      else slot -= n;
    }
  }
  if (le < 0) return false;
  seg = le / sc.T[cls];
  const int t = le - seg * sc.T[cls];
  const int tn = cls / 3, tk = cls - 3 * tn;
  // tiles of the class: (count in n) x (count in k), k fastest
  const int ck = tk == 0 ? ((tn == 1 || !sc.foldK) ? sc.fK : sc.fK - 1) : 1;
  const int in = t / ck, ik = t - in * ck;
  n0 = (tn == 0 ? in : (tn == 1 ? sc.fN - 1 : sc.fN)) * WP_T;
  k0 = (tk == 0 ? ik : (tk == 1 ? sc.fK - 1 : sc.fK)) * WP_T;
  wn = tn == 0 ? WP_T : (tn == 1 ? WP_T + sc.rN : sc.rN);
  wk = tk == 0 ? WP_T : (tk == 1 ? WP_T + sc.rK : sc.rK);
  m_begin = seg * sc.rows[cls];
  m_end = (m_begin + sc.rows[cls] < M) ? m_begin + sc.rows[cls] : M;
  return m_begin < m_end;
}

#ifdef USF_STAMP
#define WP_STAMP() __builtin_amdgcn_s_memtime()
#define WP_Q(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = WP_STAMP()
#else
#define WP_Q(v)
#endif

// the MFMA waves' main loop for a patch of NI x NJ live 16 x 16 sub-tiles; NI == 5 / NJ == 5: the fifth fragment is the
// extension image's (columns 128 .. 143 of a wide tile)
// CS: the wave also sums its Y fragments over the batch -- three more MFMAs per fragment row and slab against a B operand of
// ones (every column of the 16 x 16 result is the column sum of Y: the bias gradient of the layer, for free)
// BLK: the image holds the operands in the BLOCKED planes format (see wgrad_planes_kernel); sub-tile t of a wave's patch is
// then one half of block t >> 1 whose 16 columns come out in the order wp_blk_pos gives.
template <int NI, int NJ, bool CS = false, bool BLK = false>
__device__ __forceinline__ void wp_mfma_loop(WpShared& sh, f32x4 (&acc)[5][5], int nslab, int nslab4, int wvn, int wvk, int lane,
                                             unsigned long long* dbg_slot, f32x4 (&cs)[5]) {
  static_assert(!(NI == 5 && NJ == 5), "a 144 x 144 tile does not fit the LDS ring");
  // lane 16 g + 4 q + p supplies row 8 g + 4 hh + q, columns 16 t + 4 p .. + 3 of the sub-tile (hh = 0, 1: the two reads)
  const int lg = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  int yo[NI][2], ao[NJ][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int r = 8 * lg + 4 * hh + q;
    const int eo = 32 * wp_ext_row(r) + 16 * (pp >> 1) + 8 * (pp & 1);
#pragma unroll
    for (int t = 0; t < NI; ++t)
      yo[t][hh] = t == 4 ? eo : (BLK ? wp_blk_off(r, 2 * wvn + (t >> 1), pp, t & 1)
                                     : 256 * r + 16 * ((8 * wvn + 2 * t + (pp >> 1)) ^ wp_swz(r)) + 8 * (pp & 1));
#pragma unroll
    for (int t = 0; t < NJ; ++t)
      ao[t][hh] = t == 4 ? eo : (BLK ? wp_blk_off(r, 2 * wvk + (t >> 1), pp, t & 1)
                                     : 256 * r + 16 * ((8 * wvk + 2 * t + (pp >> 1)) ^ wp_swz(r)) + 8 * (pp & 1));
  }
  wp_bf16x8 yp[NI][3], ap[2][3];
  wp_bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  auto read_y1 = [&](int ring, int t) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      yp[t][pl] = wp_frag(reinterpret_cast<const char*>(t == 4 ? &sh.ext[ring][pl][0] : &sh.img[ring][0][pl][0]), yo[t][0], yo[t][1]);
  };
  auto read_a = [&](int ring, int j, wp_bf16x8 (&f)[3]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      f[pl] = wp_frag(reinterpret_cast<const char*>(j == 4 ? &sh.ext[ring][pl][0] : &sh.img[ring][1][pl][0]), ao[j][0], ao[j][1]);
  };
  // B0: which of the two A-fragment registers holds column 0 of this slab (alternates from slab to slab when NJ is odd);
  // one set of Y fragments: in the slab's last column each row's registers are refilled with the next slab's fragments as
  // soon as the row's products are issued
  auto slab = [&](int ring, int ring_next, auto b0) {
    constexpr int B0 = decltype(b0)::value;
#define USF_WP(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yp[i][P], ap[(j + B0) & 1][Q], acc[i][j], 0, 0, 0)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j + 1 < NJ) {
        read_a(ring, j + 1, ap[(j + 1 + B0) & 1]);
      } else {                                  // the next slab's first fragments (its image is complete since the last barrier)
        read_a(ring_next, 0, ap[(j + 1 + B0) & 1]);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#ifdef USF_WP_X_SKIP5                           // tuning build (wrong results): the wide tiles without their fifth fragment's products
        if (i == 4 || j == 4) continue;
#endif
        USF_WP(2, 0); USF_WP(1, 1); USF_WP(0, 2); USF_WP(1, 0); USF_WP(0, 1); USF_WP(0, 0);   // smallest terms first (wgrad_lw_kernel's order)
        if (CS && j == 0) {
#pragma unroll
          for (int pl = 2; pl >= 0; --pl) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yp[i][pl], ones, cs[i], 0, 0, 0);
        }
        if (j + 1 == NJ) read_y1(ring_next, i);
      }
    }
#undef USF_WP
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      constexpr int XC = CS ? 3 : 0;                            // column 0's extra MFMAs per fragment row
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      if (j + 1 == NJ) {
#pragma unroll
        for (int t = 0; t < NI; ++t) {
          if (j == 0) __builtin_amdgcn_sched_group_barrier(0x008, 4 + XC, 0);
          else __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        }
      } else {
        if (j == 0) __builtin_amdgcn_sched_group_barrier(0x008, (4 + XC) * NI, 0);
        else __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI, 0);
      }
    }
  };
  typedef std::integral_constant<int, 0> C0;
  typedef std::integral_constant<int, NJ & 1> C1;      // after an odd number of columns the roles of ap[0] / ap[1] swap
#pragma unroll
  for (int t = 0; t < NI; ++t) read_y1(0, t);
  read_a(0, 0, ap[0]);
#ifdef USF_STAMP
  unsigned long long tw = 0, tb = 0;
#endif
  int s = 0, ring = 0;
  auto nxt = [](int r) { return r == 2 ? 0 : r + 1; };
  for (; s < nslab4; s += 2) {
    WP_Q(q0);
    if (s < nslab) slab(ring, nxt(ring), C0());
    WP_Q(q1);
    __syncthreads();
    WP_Q(q2);
    ring = nxt(ring);
    if (s + 1 < nslab) slab(ring, nxt(ring), C1());
    WP_Q(q3);
    __syncthreads();
    ring = nxt(ring);
#ifdef USF_STAMP
    tw += (q1 - q0) + (q3 - q2); tb += (q2 - q1) + (WP_STAMP() - q3);
#endif
  }
#ifdef USF_STAMP
  if (dbg_slot) { dbg_slot[0] = tw; dbg_slot[1] = tb; dbg_slot[2] = nslab; dbg_slot[3] = 1; }
#endif
}

// 512 threads: waves 0 .. 3 (one per SIMD) multiply a (64 | 80) x (64 | 80) patch each, waves 4, 5 copy Y, waves 6, 7 copy A
// BLK: both operands in the BLOCKED planes format (usf_wgrad_blocked_f32); tiles are 4 blocks wide, the folded remainder is
// the first 16 positions of a fifth block.
template <bool BLK>
__global__ __launch_bounds__(512) void wgrad_planes_kernel(WpArgs a) {
  __shared__ __attribute__((aligned(16))) WpShared sh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n0, k0, wn, wk, m_begin, m_end, split;
  if (!wp_item(a.sched, a.M, n0, k0, wn, wk, m_begin, m_end, split)) return;
  const int nslab = (m_end - m_begin + WP_S - 1) / WP_S;
  const int nslab4 = (nslab + 3) & ~3;        // iterations every wave runs (barrier count): see the loader loop
  unsigned long long* dbg_slot = nullptr;
#ifdef USF_STAMP
  if (a.dbg && lane == 0) dbg_slot = a.dbg + (size_t)((blockIdx.x % 1024) * 8 + wave) * 4;
#endif

  if (wave >= 4) {
    // ------------------------------- loader waves -------------------------------
    const bool isA = wave >= 6;                 // wave-uniform (the buffer resource must sit in scalar registers)
    const int u = (tid - 256) & 127;
    const unsigned ld = (unsigned)(isA ? a.ldap : a.ldyp);
    const unsigned pstride = (unsigned)(isA ? a.astride : a.ystride) * 2u;           // bytes
    const unsigned c0 = (unsigned)(isA ? k0 + a.a_col0 : n0 + a.y_col0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(isA ? a.Ap : a.Yp), 0, (int)(isA ? a.abytes : a.ybytes), 0x00020000);
    // chunks beyond the tile's live columns are not fetched: their lanes ask for an offset beyond the buffer (answered
    // with zeros, no memory access); the MFMA waves never read those columns
    const int live = isA ? wk : wn;
    const int live_ch = 2 * (((live < WP_T ? live : WP_T) + 15) >> 4);
    const bool wide = live > WP_T;
    unsigned vo[4], eo;                        // thread = chunks u, u + 128, u + 256, u + 384 of the 32 x 16 image; one of the extension
    int vdst[4];                               // ... and where they go in the image
    // extension: 64 chunks per plane; the operand's first loader wave copies planes 0 and 1, its second wave plane 2
    const bool second = (wave & 1) != 0;        // scalar
    const int er = (u & 63) >> 1, eh = u & 1;
    const int epos = 2 * wp_ext_row(er) + eh;
    // BLOCKED: bytes per panel / between the slabs' panels, the tile's first block
    const unsigned nkb = (unsigned)(isA ? a.a_nkb : a.y_nkb);
    const unsigned kb_t = (unsigned)(isA ? a.a_kb0 + (k0 >> 5) : a.y_kb0 + (n0 >> 5));
    if (BLK) {
      const int live_blk = ((live < WP_T ? live : WP_T) + 31) >> 5;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int c = u + 128 * h, ph = c >> 8, blk = (c >> 6) & 3, L = c & 63;
        vo[h] = blk < live_blk ? ((((unsigned)(m_begin >> 4) + (unsigned)ph) * nkb + kb_t + (unsigned)blk) * 3072u + 16u * (unsigned)L) : 0x80000000u;
        vdst[h] = wp_blk_line(ph, blk, L);
      }
      // extension chunk (er, eh): pieces p = 2 eh, 2 eh + 1 = the first 8 bytes of lines 16 p + (er & 15) of the fifth block
      eo = wide ? ((((unsigned)(m_begin >> 4) + (unsigned)(er >> 4)) * nkb + kb_t + 4u) * 3072u + 16u * (unsigned)(32 * eh + (er & 15))) : 0x80000000u;
    } else {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int c = u + 128 * h, r = c >> 4, ch = (c & 15) ^ wp_swz(r);
        vo[h] = ch < live_ch ? (((unsigned)m_begin + (unsigned)r) * ld + c0 + 8u * (unsigned)ch) * 2u : 0x80000000u;
        vdst[h] = c;
      }
      eo = wide ? (((unsigned)m_begin + (unsigned)er) * ld + c0 + (unsigned)WP_T + 8u * (unsigned)eh) * 2u : 0x80000000u;
    }
    // per slab and thread: 12 loads of the main image + 2 of the extension
    auto fetch = [&](int sl, wp_u32x4 (&v)[14]) {
#ifdef USF_WP_X_NOLOAD
      if (sl > 8) return;                      // tuning build: what the kernel takes without its operand traffic (wrong results)
#endif
      const unsigned adv = BLK ? (unsigned)sl * 2u * nkb * 3072u : (unsigned)sl * (WP_S * 2u) * ld;
      const unsigned pst = BLK ? 1024u : pstride;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int h = 0; h < 4; ++h)
          v[4 * pl + h] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo[h] + adv), (int)((unsigned)pl * pst), 0);
      if (BLK) {
        // two 8-byte pieces per extension chunk (lines 16 apart = 256 bytes)
        typedef unsigned wp_u32x2 __attribute__((ext_vector_type(2)));
        const unsigned e1 = eo + adv, e2 = second ? 0x80000000u : eo + adv;
        const wp_u32x2 a0 = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)e1, (int)((second ? 2u : 0u) * pst), 0);
        const wp_u32x2 a1 = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(e1 + 256u), (int)((second ? 2u : 0u) * pst), 0);
        const wp_u32x2 b0 = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)e2, (int)pst, 0);
        const wp_u32x2 b1 = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(e2 + 256u), (int)pst, 0);
        v[12] = (wp_u32x4){a0[0], a0[1], a1[0], a1[1]};
        v[13] = (wp_u32x4){b0[0], b0[1], b1[0], b1[1]};
      } else {
        v[12] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(eo + adv), (int)((second ? 2u : 0u) * pstride), 0);
        v[13] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(second ? 0x80000000u : eo + adv), (int)pstride, 0);
      }
    };
    auto store = [&](int ring, const wp_u32x4 (&v)[14]) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int h = 0; h < 4; ++h)
          sh.img[ring][isA ? 1 : 0][pl][vdst[h]] = __builtin_bit_cast(uint4, v[4 * pl + h]);
      if (wide) {                               // wave-uniform
        sh.ext[ring][second ? 2 : 0][epos] = __builtin_bit_cast(uint4, v[12]);
        if (!second) sh.ext[ring][1][epos] = __builtin_bit_cast(uint4, v[13]);
      }
    };
#ifndef USF_WP_DEPTH
#define USF_WP_DEPTH 2
#endif
    constexpr int DEPTH = USF_WP_DEPTH;         // slabs in flight per thread (2 or 4: the slab count is rounded up to a multiple of four)
    wp_u32x4 vs[DEPTH][14];
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) fetch(t, vs[t]);
    store(0, vs[0]); fetch(DEPTH, vs[0]);
    store(1, vs[1 % DEPTH]); fetch(DEPTH + 1, vs[1 % DEPTH]);
    __syncthreads();
    // iteration s: slab s is multiplied out of ring s % 3 while slab s + 2 is written into ring (s + 2) % 3 and slab
    // s + 2 + DEPTH is fetched
    int ring2 = 2;
#ifdef USF_STAMP
    unsigned long long tw = 0, tb = 0;
#endif
    for (int s0 = 0; s0 < nslab4; s0 += DEPTH) {
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        WP_Q(q0);
        store(ring2, vs[(k + 2) % DEPTH]);
        fetch(s0 + k + 2 + DEPTH, vs[(k + 2) % DEPTH]);
        WP_Q(q1);
        __syncthreads();
        ring2 = ring2 == 2 ? 0 : ring2 + 1;
#ifdef USF_STAMP
        tw += q1 - q0; tb += WP_STAMP() - q1;
#endif
      }
    }
#ifdef USF_STAMP
    if (dbg_slot) { dbg_slot[0] = tw; dbg_slot[1] = tb; dbg_slot[2] = nslab; dbg_slot[3] = 1; }
#endif
    return;
  }

  // ------------------------------- MFMA waves -------------------------------
  const int wvn = wave >> 1, wvk = wave & 1;
  f32x4 acc[5][5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // live 16-wide sub-tiles of this wave's patch (wave-uniform): 64 columns each, the second wave also the extension
  const int rem_n = wn - wvn * 64, rem_k = wk - wvk * 64;
  int ni = rem_n <= 0 ? 0 : (wvn == 0 ? (rem_n >= 64 ? 4 : (rem_n + 15) / 16) : (rem_n + 15) / 16);
  int nj = rem_k <= 0 ? 0 : (wvk == 0 ? (rem_k >= 64 ? 4 : (rem_k + 15) / 16) : (rem_k + 15) / 16);
  if (BLK) {                                    // a block's two sub-tiles interleave its positions: whole blocks are live
    if (ni > 0 && ni < 4) ni = (ni + 1) & ~1;
    if (nj > 0 && nj < 4) nj = (nj + 1) & ~1;
  }
  const bool do_cs = a.cs_part != nullptr && k0 == 0 && wvk == 0;      // wave-uniform
  f32x4 cs[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  // patch size: live sub-tiles rounded up to {1, 2, 4, 5}; a wave whose patch lies outside the tile only keeps the barriers
  if (ni == 0 || nj == 0) {
    for (int s = 0; s < nslab4; ++s) __syncthreads();
  } else {
#define WP_GO(NI_, NJ_) wp_mfma_loop<NI_, NJ_, false, BLK>(sh, acc, nslab, nslab4, wvn, wvk, lane, dbg_slot, cs)
#define WP_ROW(NI_) do { if (nj > 4) WP_GO(NI_, 5); else if (nj > 2) WP_GO(NI_, 4); else if (nj > 1) WP_GO(NI_, 2); else WP_GO(NI_, 1); } while (0)
#define WP_ROW4(NI_) do { if (nj > 2) WP_GO(NI_, 4); else if (nj > 1) WP_GO(NI_, 2); else WP_GO(NI_, 1); } while (0)
#define WP_CS(NI_) wp_mfma_loop<NI_, 4, true, BLK>(sh, acc, nslab, nslab4, wvn, wvk, lane, dbg_slot, cs)
    if (do_cs && nj == 4) {                     // (the host asks for column sums only where K >= 64: nj == 4 in wave column 0)
      if (ni > 4) WP_CS(5); else if (ni > 2) WP_CS(4); else if (ni > 1) WP_CS(2); else WP_CS(1);
    } else if (ni > 4) WP_ROW4(5); else if (ni > 2) WP_ROW(4); else if (ni > 1) WP_ROW(2); else WP_ROW(1);
#undef WP_CS
#undef WP_ROW4
#undef WP_ROW
#undef WP_GO
  }
  float* out = a.part + (int64_t)split * a.N * a.K;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int tn = wvn * 64 + i * 16 + 4 * (lane >> 4) + r;
        int tk = wvk * 64 + j * 16 + (lane & 15);
        if (BLK) {                              // (the extension sub-tile, index 4 of the second wave, is in position order)
          if (i < 4) tn = wvn * 64 + (i >> 1) * 32 + wp_blk_pos(4 * (lane >> 4) + r, i & 1);
          if (j < 4) tk = wvk * 64 + (j >> 1) * 32 + wp_blk_pos(lane & 15, j & 1);
        }
        if (tn < wn && tk < wk && i < ni && j < nj) out[(int64_t)(n0 + tn) * a.K + k0 + tk] = acc[i][j][r];
      }
  if (do_cs && (lane & 15) == 0) {
    float* co = a.cs_part + (int64_t)split * a.N;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int tn = wvn * 64 + i * 16 + 4 * (lane >> 4) + r;
        if (BLK && i < 4) tn = wvn * 64 + (i >> 1) * 32 + wp_blk_pos(4 * (lane >> 4) + r, i & 1);
        if (tn < wn && i < ni) co[n0 + tn] = cs[i][r];
      }
  }
}
#undef WP_Q

// what the reduction needs of a schedule
struct WpRed { int fN, fK, foldN, foldK; int nseg[9]; };
static WpRed wp_red_of(const WpSched& s) {
  WpRed r{s.fN, s.fK, s.foldN, s.foldK, {}};
  for (int i = 0; i < 9; ++i) r.nseg[i] = s.nseg[i];
  return r;
}

// element (r, c) sums the nseg[class of its tile] partials its tile wrote; block bx of nb (its launch's blocks, or its job's)
__device__ __forceinline__ void reduce_partials_cls_body(const float* __restrict__ part, const WpRed& sc, int64_t rows, int64_t cols,
                                                         float* __restrict__ out, int64_t ldo, float alpha, float beta,
                                                         const float* __restrict__ cs_part, float* __restrict__ cs_out,
                                                         float cs_alpha, float cs_beta, int64_t bx, int64_t nb) {
  const int64_t total = rows * cols;
  if (cs_out)                                   // the column sums of Y: the partials of the tile in tile column 0
    for (int64_t r = bx * 256 + threadIdx.x; r < rows; r += nb * 256) {
      const int tn = sc.foldN ? (r >= (int64_t)(sc.fN - 1) * WP_T ? 1 : 0) : (r >= (int64_t)sc.fN * WP_T ? 2 : 0);
      const int tk0 = sc.fK == 0 ? 2 : ((tn != 1 && sc.fK == 1 && sc.foldK) ? 1 : 0);
      const int n = sc.nseg[3 * tn + tk0];
      float s = 0.f;
#pragma unroll 16
      for (int p = 0; p < n; ++p) s += cs_part[(int64_t)p * rows + r];
      float v = cs_alpha * s;
      if (cs_beta != 0.f) v += cs_beta * cs_out[r];
      cs_out[r] = v;
    }
  for (int64_t e = bx * 256 + threadIdx.x; e < total; e += nb * 256) {
    const int64_t r = e / cols, c = e - r * cols;
    const int tn = sc.foldN ? (r >= (int64_t)(sc.fN - 1) * WP_T ? 1 : 0) : (r >= (int64_t)sc.fN * WP_T ? 2 : 0);
    const int tk = tn == 1 ? (c >= (int64_t)sc.fK * WP_T ? 2 : 0)
                           : (sc.foldK ? (c >= (int64_t)(sc.fK - 1) * WP_T ? 1 : 0) : (c >= (int64_t)sc.fK * WP_T ? 2 : 0));
    const int n = sc.nseg[3 * tn + tk];
    float s = 0.f;
    for (int p = 0; p < n; ++p) s += part[(int64_t)p * total + e];
    float v = alpha * s;
    if (beta != 0.f) v += beta * out[r * ldo + c];
    out[r * ldo + c] = v;
  }
}

__global__ __launch_bounds__(256) void reduce_partials_cls_kernel(const float* __restrict__ part, WpRed sc, int64_t rows, int64_t cols,
                                                                  float* __restrict__ out, int64_t ldo, float alpha, float beta,
                                                                  const float* __restrict__ cs_part, float* __restrict__ cs_out,
                                                                  float cs_alpha, float cs_beta) {
  reduce_partials_cls_body(part, sc, rows, cols, out, ldo, alpha, beta, cs_part, cs_out, cs_alpha, cs_beta, (int64_t)blockIdx.x,
                           (int64_t)gridDim.x);
}

// MANY of these reductions in ONE launch (usf_wgrad_reduce_jobs_f32): block b works on job block_job[b] as block b - first_block of
// that job's own launch would -- same additions in the same order, same bits.  A training step of the cfg2 model at 65 536 rows ends
// 129 weight gradients with such a reduction of ~18 us each, one after the other in stream order although only the parameter
// update waits for them: 2.3 ms of the step; queued they are one launch that fills the chip.
static_assert(sizeof(WpRed) <= sizeof(((usf_wreduce_job*)nullptr)->sched), "usf_wreduce_job.sched too small");
__global__ __launch_bounds__(256) void reduce_partials_cls_jobs_kernel(const usf_wreduce_job* __restrict__ jobs,
                                                                       const int32_t* __restrict__ block_job) {
  const usf_wreduce_job* j = jobs + block_job[blockIdx.x];
  const WpRed sc = *reinterpret_cast<const WpRed*>(j->sched);
  reduce_partials_cls_body(j->part, sc, j->rows, j->cols, j->out, j->ldo, j->alpha, j->beta, j->cs_part, j->cs_out, j->cs_alpha,
                           j->cs_beta, (int64_t)blockIdx.x - j->first_block, (int64_t)j->blocks);
}

// fp32 -> three bf16 planes for 8 values, two at a time (v_cvt_pk_bf16_f32 / v_pk_add_f32)
__device__ __forceinline__ void wp_split(const float (&x)[8], wp_bf16x8& p1, wp_bf16x8& p2, wp_bf16x8& p3) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const wp_f32x2 v = {x[2 * t], x[2 * t + 1]};
    const wp_bf16x2 h = __builtin_convertvector(v, wp_bf16x2);
    const wp_f32x2 r = v - __builtin_convertvector(h, wp_f32x2);            // exact
    const wp_bf16x2 m = __builtin_convertvector(r, wp_bf16x2);
    const wp_f32x2 r2 = r - __builtin_convertvector(m, wp_f32x2);           // exact
    const wp_bf16x2 l = __builtin_convertvector(r2, wp_bf16x2);
    p1[2 * t] = h[0]; p1[2 * t + 1] = h[1];
    p2[2 * t] = m[0]; p2[2 * t + 1] = m[1];
    p3[2 * t] = l[0]; p3[2 * t + 1] = l[1];
  }
}

// fp32 rows -> three row-major bf16 planes (round-to-nearest residual split, x = p1 + p2 + p3 exactly): P[pl][m][c] for
// m < rows_pad, c < ldp; zeros for m >= M or c >= N.  One thread = 8 columns of a row.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ X, int64_t ldx, int M, int N, __bf16* __restrict__ P,
                                                           int64_t ldp, int64_t pstride, int rows_pad, int vec) {
  const int cpr = (int)(ldp >> 3);
  const int64_t total = (int64_t)rows_pad * cpr;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / cpr), c = (int)(e - (int64_t)m * cpr) * 8;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = 0.f;
    if (m < M) {
      if (c + 8 <= N && vec) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(X + (int64_t)m * ldx + c), v1 = *reinterpret_cast<const f32x4*>(X + (int64_t)m * ldx + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = v0[j]; x[4 + j] = v1[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (c + j < N) x[j] = X[(int64_t)m * ldx + c + j];
      }
    }
    wp_bf16x8 p1, p2, p3;
    wp_split(x, p1, p2, p3);
    __bf16* d = P + (int64_t)m * ldp + c;
    *reinterpret_cast<wp_bf16x8*>(d) = p1;
    *reinterpret_cast<wp_bf16x8*>(d + pstride) = p2;
    *reinterpret_cast<wp_bf16x8*>(d + 2 * pstride) = p3;
  }
}

// ---- host: the schedule --------------------------------------------------------------------------------------------
int wp_cus() {
  static int n = -1;
  if (n < 0) {
    int dev = 0; hipDeviceProp_t pr;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    const long long e = tuning("wgrad_cus", 0);
    if (e > 0) n = (int)e;
    n &= ~7;
    if (n < 8) n = 8;
  }
  return n;
}
int wp_env(const char* name, int dflt) { return (int)tuning(name, dflt); }

// balanced = false: the plain grid -- no folding, every tile cut into the same `plain_splits` row ranges (the partial sums
// are then those of usf_wgrad_f32's loader-wave kernel, bit for bit)
bool wp_schedule(int64_t M, int64_t N, int64_t K, bool balanced, int plain_splits, WpSched& sc) {
  memset(&sc, 0, sizeof(sc));
  const int cus = wp_cus();
  const int S = (int)((M + WP_S - 1) / WP_S);                  // slabs
  sc.per_xcd = cus / 8;
  sc.fN = (int)(N / WP_T); sc.fK = (int)(K / WP_T);
  sc.rN = (int)(N % WP_T); sc.rK = (int)(K % WP_T);
  const bool fold = balanced && wp_env("wgrad_fold", 1) != 0;
  sc.foldN = (fold && sc.fN >= 1 && sc.rN > 0 && sc.rN <= WP_FOLD) ? 1 : 0;
  sc.foldK = (fold && sc.fK >= 1 && sc.rK > 0 && sc.rK <= WP_FOLD) ? 1 : 0;
  // tile counts per type in n; in k they depend on the row's type (the wide row keeps its edge)
  const int cn[3] = {sc.foldN ? sc.fN - 1 : sc.fN, sc.foldN, (!sc.foldN && sc.rN) ? 1 : 0};
  int tiles = 0;
  for (int tn = 0; tn < 3; ++tn)
    for (int tk = 0; tk < 3; ++tk) {
      const int c = 3 * tn + tk;
      int ck;
      if (tn == 1) ck = tk == 0 ? sc.fK : (tk == 2 ? (sc.rK ? 1 : 0) : 0);
      else ck = tk == 0 ? (sc.foldK ? sc.fK - 1 : sc.fK) : (tk == 1 ? sc.foldK : ((!sc.foldK && sc.rK) ? 1 : 0));
      sc.T[c] = cn[tn] * ck;
      tiles += sc.T[c];
    }
  if (tiles == 0 || S < 1) return false;
  int ns;
  if (balanced) {
    if (tiles > cus || S < 4) return false;
    // SEVERAL items per CU where that fills the chip better: with one (37 tiles x 6 row ranges at 784 x 784 = 222 items) 34 of
    // the 256 CUs idle and every row range is walked by two XCDs; with three (37 x 20 = 740 items for 768 slots) the grid is a
    // plain one the hardware deals out block by block -- block b runs on XCD b & 7, its items numbered so that the blocks an
    // XCD runs side by side are the tiles of ONE row range -- at the price of 20 instead of 6 partial images to sum.  The
    // number of row ranges minimises rounds x (slabs per item + ~6 slabs' worth of ring prologue / partial-image store).
    const int max_rounds = wp_env("wgrad_rounds", 3);
    const int ns_max = (max_rounds > 1 ? max_rounds : 1) * cus / tiles;
    ns = cus / tiles < 1 ? 1 : cus / tiles;
    long best = -1;
    for (int c = 1; c <= ns_max && c <= S && c <= 256; ++c) {
      const int slabs = (S + c - 1) / c, nseg = (S + slabs - 1) / slabs;
      const long items = (long)tiles * nseg, rounds = (items + cus - 1) / cus;
      const long cost = rounds * (slabs + 6);
      if (best < 0 || cost < best) { best = cost; ns = c; }
    }
  } else {
    ns = plain_splits < 1 ? 1 : plain_splits;
  }
  if (ns > S) ns = S;
  if (ns > 256) ns = 256;
  for (int c = 0; c < 9; ++c) sc.nseg[c] = sc.T[c] ? ns : 0;
  sc.items = 0;
  for (int c = 0; c < 9; ++c) {
    if (sc.T[c]) {
      const int slabs = (S + sc.nseg[c] - 1) / sc.nseg[c];
      sc.rows[c] = slabs * WP_S;
      sc.nseg[c] = (S + slabs - 1) / slabs;
    }
    sc.items += sc.T[c] * sc.nseg[c];
  }
  // class by class, an eighth to every XCD; the remainders go to the XCDs with the fewest items so far
  int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = 0; c < 9; ++c) {
    const int n = sc.T[c] * sc.nseg[c];
    int cnx[8];
    for (int x = 0; x < 8; ++x) cnx[x] = n / 8;
    for (int r = 0; r < n % 8; ++r) {
      int bx = 0;
      for (int x = 1; x < 8; ++x) if (load[x] + cnx[x] < load[bx] + cnx[bx]) bx = x;
      ++cnx[bx];
    }
    int st = 0;
    for (int x = 0; x < 8; ++x) {
      if (st > 32000 || cnx[x] > 32000) return false;
      sc.start[c][x] = (short)st; sc.cnt[c][x] = (short)cnx[x]; st += cnx[x]; load[x] += cnx[x];
    }
  }
  int mx = 0;
  for (int x = 0; x < 8; ++x) if (load[x] > mx) mx = load[x];
  sc.per_xcd = mx;                                             // as many blocks per XCD as its share of the items
  return true;
}
int wp_max_parts(const WpSched& sc) {
  int m = 1;
  for (int c = 0; c < 9; ++c) if (sc.nseg[c] > m) m = sc.nseg[c];
  return m;
}
// the loader-wave kernel's number of row ranges (usf_train.hip: pick_splits(lw = true))
int wp_plain_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((N + WP_T - 1) / WP_T) * ((K + WP_T - 1) / WP_T);
  int blocks = wp_env("wgrad_blocks", 768);
  if (blocks < 1) blocks = 768;
  int64_t s = (blocks + tiles - 1) / tiles;
  const int64_t smax = (M + 1023) / 1024;
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

}  // namespace

#ifdef USF_STAMP
unsigned long long* g_wpdbg = nullptr;
#endif

// usf_split_planes_f32 / usf_wgrad_planes_f32: see include/usflows_hip.h
int split_planes(const float* X, int64_t ldx, int64_t M, int64_t N, void* P, int64_t ldp, int64_t plane_stride, hipStream_t stream) {
  if ((!X && M > 0) || !P || M < 0 || N <= 0 || ldx < N || ldp < N || (ldp & 7) || !aligned16(P) || (plane_stride & 7)) {
    set_error("usf_split_planes_f32: bad arguments (ldp and plane_stride multiples of 8, 16-byte aligned planes)");
    return -1;
  }
  const int64_t rows_pad = (M + WP_S - 1) / WP_S * WP_S;
  if (M > 0x7fffffff - WP_S || plane_stride < rows_pad * ldp) { set_error("usf_split_planes_f32: plane_stride < ceil32(M) * ldp"); return -2; }
  if (rows_pad == 0) return 0;
  int64_t nb = (rows_pad * (ldp >> 3) + 255) / 256;
  if (nb > 65536) nb = 65536;
  split_planes_kernel<<<(unsigned)nb, 256, 0, stream>>>(X, ldx, (int)M, (int)N, (__bf16*)P, ldp, plane_stride, (int)rows_pad,
                                                      (aligned16(X) && !(ldx & 3)) ? 1 : 0);
  return check_launch("usf_split_planes_f32");
}

int wgrad_planes_ok(int64_t M, int64_t N, int64_t K) {
  // where the loader-wave kernel of usf_wgrad_f32 is chosen (its cross-over: the same block shape)
  const int64_t tiles = ((N + WP_T - 1) / WP_T) * ((K + WP_T - 1) / WP_T);
  return M >= 8192 && M * tiles >= 160000;
}

int64_t wgrad_planes_workspace_floats(int64_t M, int64_t N, int64_t K) {
  if (M < 0 || N <= 0 || K <= 0) return -1;
  int m = wp_plain_splits(M, N, K);
  WpSched sc;
  if (wp_schedule(M, N, K, true, 0, sc) && wp_max_parts(sc) > m) m = wp_max_parts(sc);
  return (int64_t)m * N * (K + 1);            // + the partial column sums
}

// column sums ride along where the first wave column of tile column 0 is full (64 columns: the instantiations that carry them)
int wgrad_planes_colsum_ok(int64_t M, int64_t N, int64_t K) { return (M > 0 && N > 0 && K >= 64) ? 1 : 0; }

int wgrad_planes(const void* Yp, int64_t ldyp, int64_t ystride, int64_t y_off, const void* Ap, int64_t ldap, int64_t astride,
                 int64_t a_off, int64_t M, int64_t N, int64_t K, float* G, int64_t ldg, float alpha, float beta, float* colsum_out,
                 float cs_alpha, float cs_beta, float* workspace, int64_t workspace_floats, hipStream_t stream) {
  if (colsum_out && !wgrad_planes_colsum_ok(M, N, K)) {
    set_error("usf_wgrad_planes_f32: colsum_out needs K >= 64 (usf_wgrad_planes_colsum_ok)");
    return -2;
  }
  if (!Yp || !Ap || !G || !workspace || M <= 0 || N <= 0 || K <= 0 || ldg < K || y_off < 0 || a_off < 0 || ldyp < y_off + N ||
      ldap < a_off + K) {
    set_error("usf_wgrad_planes_f32: bad arguments");
    return -1;
  }
  if ((ldyp & 7) || (ldap & 7) || (y_off & 7) || (a_off & 7) || (ystride & 7) || (astride & 7) || !aligned16(Yp) || !aligned16(Ap)) {
    set_error("usf_wgrad_planes_f32: row strides, plane strides and column offsets must be multiples of 8 elements, planes 16-byte aligned");
    return -3;
  }
  const int64_t rows_pad = (M + WP_S - 1) / WP_S * WP_S;
  if (ystride < rows_pad * ldyp || astride < rows_pad * ldap) { set_error("usf_wgrad_planes_f32: plane stride < ceil32(M) * ld"); return -2; }
  const int64_t yb = (2 * ystride + rows_pad * ldyp) * 2, ab = (2 * astride + rows_pad * ldap) * 2;
  if (M > 0x7fffffff - 4096 || N > (1 << 20) || K > (1 << 20) || yb >= (1LL << 31) || ab >= (1LL << 31)) {
    set_error("usf_wgrad_planes_f32: size out of range (the planes of one operand must stay below 2 GiB)");
    return -2;
  }
  WpArgs a{(const __bf16*)Yp, ldyp, ystride, (const __bf16*)Ap, ldap, astride, (int)y_off, (int)a_off, (unsigned)yb, (unsigned)ab,
           workspace, (int)M, (int)N, (int)K, nullptr, nullptr, 0, 0, 0, 0};
#ifdef USF_STAMP
  a.dbg = g_wpdbg;
#endif
  const bool balanced = wp_env("wgrad_sched", 1) != 0 && wp_schedule(M, N, K, true, 0, a.sched) &&
                        (int64_t)wp_max_parts(a.sched) * N * (K + 1) <= workspace_floats;
  if (!balanced && !wp_schedule(M, N, K, false, wp_plain_splits(M, N, K), a.sched)) {
    set_error("usf_wgrad_planes_f32: no schedule for M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    return -2;
  }
  if ((int64_t)wp_max_parts(a.sched) * N * (K + (colsum_out ? 1 : 0)) > workspace_floats) {
    set_error("usf_wgrad_planes_f32: workspace too small (%lld < %lld floats)", (long long)workspace_floats,
              (long long)wp_max_parts(a.sched) * N * (K + 1));
    return -4;
  }
  if (colsum_out) a.cs_part = workspace + (int64_t)wp_max_parts(a.sched) * N * K;
  wgrad_planes_kernel<false><<<(unsigned)(a.sched.per_xcd * 8), 512, 0, stream>>>(a);
  int64_t rb = (N * K + 255) / 256;
  if (rb > 4096) rb = 4096;
  reduce_partials_cls_kernel<<<(unsigned)rb, 256, 0, stream>>>(workspace, wp_red_of(a.sched), N, K, G, ldg, alpha, beta, a.cs_part, colsum_out,
                                                                cs_alpha, cs_beta);
  return check_launch("usf_wgrad_planes_f32");
}

// usf_wgrad_blocked_f32: see include/usflows_hip.h
int wgrad_blocked(const void* Yp, int64_t y_nkb, int64_t y_kb0, const void* Ap, int64_t a_nkb, int64_t a_kb0, int64_t M, int64_t N,
                  int64_t K, float* G, int64_t ldg, float alpha, float beta, float* colsum_out, float cs_alpha, float cs_beta,
                  float* workspace, int64_t workspace_floats, usf_wreduce_job* job, hipStream_t stream) {
  if (job) job->blocks = 0;
  if (colsum_out && !wgrad_planes_colsum_ok(M, N, K)) { set_error("usf_wgrad_blocked_f32: colsum_out needs K >= 64"); return -2; }
  if (!Yp || !Ap || !G || !workspace || M <= 0 || N <= 0 || K <= 0 || ldg < K || y_kb0 < 0 || a_kb0 < 0 || y_nkb <= 0 || a_nkb <= 0 ||
      y_kb0 * 32 + N > y_nkb * 32 || a_kb0 * 32 + K > a_nkb * 32 || !aligned16(Yp) || !aligned16(Ap)) {
    set_error("usf_wgrad_blocked_f32: bad arguments (block ranges inside the buffers, 16-byte aligned planes)");
    return -1;
  }
  const int64_t npanels = (M + 15) / 16;
  const int64_t yb = npanels * y_nkb * 3072, ab = npanels * a_nkb * 3072;
  if (M > 0x7fffffff - 4096 || N > (1 << 20) || K > (1 << 20) || yb >= (1LL << 31) || ab >= (1LL << 31)) {
    set_error("usf_wgrad_blocked_f32: size out of range (a planes buffer must stay below 2 GiB)");
    return -2;
  }
  WpArgs a{(const __bf16*)Yp, 0, 0, (const __bf16*)Ap, 0, 0, 0, 0, (unsigned)yb, (unsigned)ab,
           workspace, (int)M, (int)N, (int)K, nullptr, nullptr, (int)y_nkb, (int)y_kb0, (int)a_nkb, (int)a_kb0};
  // the same schedules as usf_wgrad_planes_f32 (one block per CU where the tiles allow it, else the plain grid)
  const bool balanced = wp_env("wgrad_sched", 1) != 0 && wp_schedule(M, N, K, true, 0, a.sched) &&
                        (int64_t)wp_max_parts(a.sched) * N * (K + 1) <= workspace_floats;
  if (!balanced && !wp_schedule(M, N, K, false, wp_plain_splits(M, N, K), a.sched)) {
    set_error("usf_wgrad_blocked_f32: no schedule for M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    return -2;
  }
  if ((int64_t)wp_max_parts(a.sched) * N * (K + (colsum_out ? 1 : 0)) > workspace_floats) {
    set_error("usf_wgrad_blocked_f32: workspace too small (%lld < %lld floats)", (long long)workspace_floats,
              (long long)wp_max_parts(a.sched) * N * (K + 1));
    return -4;
  }
  if (colsum_out) a.cs_part = workspace + (int64_t)wp_max_parts(a.sched) * N * K;
  wgrad_planes_kernel<true><<<(unsigned)(a.sched.per_xcd * 8), 512, 0, stream>>>(a);
  int64_t rb = (N * K + 255) / 256;
  if (rb > 4096) rb = 4096;
  if (job) {
    // the reduction is handed to the caller (usf_wgrad_reduce_jobs_f32): G / colsum_out stay unwritten, the workspace stays in use
    memset(job, 0, sizeof(*job));
    const WpRed red = wp_red_of(a.sched);
    memcpy(job->sched, &red, sizeof(red));
    job->part = workspace; job->out = G; job->cs_part = a.cs_part; job->cs_out = colsum_out;
    job->rows = N; job->cols = K; job->ldo = ldg; job->alpha = alpha; job->beta = beta; job->cs_alpha = cs_alpha; job->cs_beta = cs_beta;
    job->blocks = (int32_t)rb;
    return check_launch("usf_wgrad_blocked_f32");
  }
  reduce_partials_cls_kernel<<<(unsigned)rb, 256, 0, stream>>>(workspace, wp_red_of(a.sched), N, K, G, ldg, alpha, beta, a.cs_part, colsum_out,
                                                                cs_alpha, cs_beta);
  return check_launch("usf_wgrad_blocked_f32");
}

// tuning aid (tools/exp_wgradp.hip): the schedule in words
void wgrad_planes_describe(int64_t M, int64_t N, int64_t K, char* buf, size_t n) {
  WpSched sc;
  const bool ok = wp_env("wgrad_sched", 1) != 0 && wp_schedule(M, N, K, true, 0, sc);
  if (!ok) wp_schedule(M, N, K, false, wp_plain_splits(M, N, K), sc);
  size_t o = (size_t)snprintf(buf, n, "%s, %d items on %d blocks, fold n/k %d/%d;", ok ? "balanced" : "plain grid", sc.items, sc.per_xcd * 8, sc.foldN, sc.foldK);
  static const char* nm[3] = {"n", "w", "e"};
  for (int c = 0; c < 9 && o < n; ++c)
    if (sc.T[c]) o += (size_t)snprintf(buf + o, n - o, " %s%s: %d tiles x %d ranges of %d rows;", nm[c / 3], nm[c % 3], sc.T[c], sc.nseg[c], sc.rows[c]);
}

int wgrad_reduce_jobs(const usf_wreduce_job* jobs, const int32_t* block_job, int64_t n_blocks, hipStream_t stream) {
  if (n_blocks < 0 || n_blocks > 0x7fffffff || (n_blocks > 0 && (!jobs || !block_job))) {
    set_error("usf_wgrad_reduce_jobs_f32: bad arguments");
    return -1;
  }
  if (n_blocks == 0) return 0;
  reduce_partials_cls_jobs_kernel<<<(unsigned)n_blocks, 256, 0, stream>>>(jobs, block_job);
  return check_launch("usf_wgrad_reduce_jobs_f32");
}

}  // namespace usf
