// The target's PREFILL on the kernels (model/dflash.py:218-225; SURVEY.md §8f-1, VERDICT r2 "missing" #1): P prompt
// rows through the target's layers once, K/V written into the preallocated cache, taps kept on the way.
//
// Unlike every other GEMM of this library this one is compute-shaped (M = P = 1024 rows at BASELINE's prefix: 14 TFLOP
// per 36-layer pass), so it is an LDS-tiled MFMA GEMM — but it runs on the SAME operands as the decode path:
//   * weights in the packed layout [N/16][K/32][64 lanes][8 bf16] (dfl_pack_weight): one (column tile, k-step) fragment
//     is 1 KiB, contiguous, and already in v_mfma_f32_16x16x32_bf16 A-operand order;
//   * activations as frag16 row tiles [P/16][K/32][64][8]: the matching B-operand fragments.
// Both are therefore staged by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, lane-linear image) and read
// back with conflict-free ds_read_b128: no transpose, no swizzle, no register staging.  The target keeps ONE copy of its
// weights for prefill and verify (NativeTarget(keep_hf = False)).
//
// k_pgemm: 256 threads = 4 waves (2 x 2), block tile 128 columns x 128 rows, 64-deep K steps, two LDS stages (64 KiB;
// three of 24 KiB for the 64-row blocks),
// each wave 4 x 4 MFMA tiles (64 accumulator VGPRs); the 8 row blocks of a column block share blockIdx % 8, i.e. an XCD
// and its L2, so the weights leave HBM once.  Epilogues: bf16 rows / residual add (+ tap copy) / SiLU(gate) * up -> frag16.
#include "gemm_rows.h"
#include "moe_route.h"

namespace {

enum { PEPI_ROWS = 0, PEPI_RESID = 1, PEPI_SILU = 2, PEPI_SCALE32 = 3 };

struct PGemmArgs {
  const bf16x8 *wp;  // [ntiles][KS][64]
  const bf16x8 *xf;  // [mtiles][KS][64]
  int KS, ntiles, mtiles, P;
  bf16_t *out;       // ROWS: out[m][n]; RESID: the residual rows h[m][n], updated in place
  int64_t ldo;
  bf16_t *tap;       // RESID: optional copy of the new rows
  int64_t ldtap;
  bf16x8 *act;       // SILU: frag16 row tiles of I columns [mtiles][KSo][64]
  int KSo;
  // grouped form (the experts of a sparse-MoE layer, rows gathered per expert into 16-row tiles): the workgroup's
  // row block comes from a device-built work list of (expert, first gathered tile, tiles <= 4) items
  const int32_t *items, *n_items;
  int64_t w_expert_stride;  // bf16x8 units between the experts' packed weights
  const int32_t *src_row;   // optional: xf holds the UNGATHERED row tiles and gathered row g is source row src_row[g] (< 0: a
  const bf16x8 *zeros;      //   padding row, read from this 1 KiB of zeros) — the gather happens in the LDS-DMA addresses
  float *out32;             // SCALE32: out32[gathered row][n] = routing weight of the row x sum (fp32, no rounding)
  const float *row_w;       // [gathered row]
  int64_t ld32;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// The epilogues of k_pgemm for one wave: acc[i][j] = n tile nb * 8 + wn * 4 + i x row tile tile0 + wm * MW + j.
template <int EPI, int MW, bool GRP>
__device__ __forceinline__ void pgemm_epilogue(const PGemmArgs &a, f32x4 (&acc)[4][MW], int nb, int wn, int wm, int tile0, int ntl,
                                               int l) {
  // D layout (A = W rows n, B = x^T columns m): lane L, register r = column 4 (L >> 4) + r of the n tile, row L & 15
  const int fm = l & 15, fg = l >> 4;
#pragma unroll
  for (int j = 0; j < MW; ++j) {
    if (GRP && wm * MW + j >= ntl) continue;
    const int mt = tile0 + wm * MW + j;
    const int m = mt * 16 + fm;
    if (EPI == PEPI_SILU) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {  // (gate, up) tile pairs: tf:modeling_qwen3.py:82, rounded where torch rounds
        const int pair = (nb * 8 + wn * 4) / 2 + q;
        const int n0 = pair * 16 + 4 * fg;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gb = rbf(acc[2 * q][j][r]), ub = rbf(acc[2 * q + 1][j][r]);
          const float act = rbf(gb / (1.f + __expf(-gb)));
          o[r] = f2bf(act * ub);
        }
        bf16_t *dst = reinterpret_cast<bf16_t *>(a.act + ((size_t)mt * a.KSo + (n0 >> 5)) * 64 + ((n0 >> 3) & 3) * 16 + fm) + (n0 & 7);
        *reinterpret_cast<bf16x4 *>(dst) = o;
      }
    } else if (EPI == PEPI_SCALE32) {  // an expert's down projection: scaled by the row's routing weight, kept fp32
      const float rw = a.row_w[m];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n0 = (nb * 8 + wn * 4 + i) * 16 + 4 * fg;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = rw * acc[i][j][r];
        *reinterpret_cast<f32x4 *>(a.out32 + (int64_t)m * a.ld32 + n0) = o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n0 = (nb * 8 + wn * 4 + i) * 16 + 4 * fg;
        if (m < a.P) {
          bf16_t *dst = a.out + (int64_t)m * a.ldo + n0;
          bf16x4 o;
          if (EPI == PEPI_RESID) {  // model/dflash.py:140,144 form of the residual add: bf16 + bf16 -> bf16
            const bf16x4 hv = *reinterpret_cast<const bf16x4 *>(dst);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(bf2f(hv[r]) + rbf(acc[i][j][r])));
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[i][j][r]);
          }
          *reinterpret_cast<bf16x4 *>(dst) = o;
          if (EPI == PEPI_RESID && a.tap) *reinterpret_cast<bf16x4 *>(a.tap + (int64_t)m * a.ldtap + n0) = o;
        }
      }
    }
  }
}

// MW = row tiles per wave: 4 -> block tile 128 columns x 128 rows; 2 -> 128 x 64 (twice the workgroups: the N = 4096
// projections o_proj / down_proj have only 32 column blocks, and ONE 4-wave workgroup per CU leaves the matrix pipe idle
// whenever it waits for its own LDS-DMA: 115 -> see DESIGN.md for the measured step)
// NST = LDS stages: NST - 1 in flight while one is in the MFMAs (two workgroups per CU up to 80 KiB each).
// GRP: the grouped form: the work items are dealt round-robin to the XCDs, an item's column blocks stay on one.
// (Tried for the grouped form and dropped, commit history: rings of their own for the two operands — five 16 KB weight
// stages from HBM and three activation stages from L2, fed by different waves so that the in-order vmcnt of a wave
// tracks one ring, ONE workgroup per CU: 607 us per 30B-A3B layer against 449 for this kernel with two workgroups per CU.)
template <int EPI, int MW, int NST, bool GRP = false>
__global__ __launch_bounds__(256) void k_pgemm(PGemmArgs a) {
  constexpr int MB = 2 * MW;  // row tiles per block
  constexpr int PER = (16 + 2 * MB) / 4;  // LDS-DMA instructions per wave and stage
  static_assert(PER == 8 || PER == 6, "vmcnt immediates below");
  static_assert(NST == 2 || NST == 3, "stage counts measured");
  __shared__ bf16x8 lds[NST][16 + 2 * MB][64];  // [stage][slot: 0..15 = W (n tile, k-step), 16.. = X (m tile, k-step)][lane]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nbx = a.ntiles >> 3, nby = a.mtiles / MB;
  // column block / row block of this workgroup: the nby row blocks of a column block are 8 apart in blockIdx
  int nb, mb = 0, tile0 = 0, ntl = MB;
  const bf16x8 *wbase = a.wp;
  if (GRP) {
    // the column blocks of an item share blockIdx % 8, i.e. an XCD and its L2: the item's gathered rows leave HBM once
    // (first form, blockIdx = item * nbx + nb: the 12 column blocks of an item sat on 8 XCDs and every one of those
    // L2s fetched the rows for itself)
    const int seq = blockIdx.x >> 3, item = (seq / nbx) * 8 + (blockIdx.x & 7);
    nb = seq % nbx;
    if (item >= *a.n_items) return;  // (uniform: the grid is sized for the worst case)
    wbase += (int64_t)a.items[3 * item] * a.w_expert_stride;
    tile0 = a.items[3 * item + 1];
    ntl = a.items[3 * item + 2];
  } else {
    const int b = blockIdx.x, per = 8 * nby, g = b / per;
    if (g < (nbx >> 3)) {
      const int rem = b - g * per;
      nb = g * 8 + (rem & 7);
      mb = rem >> 3;
    } else {
      const int tail = nbx & 7, rem = b - (nbx >> 3) * per;
      nb = (nbx >> 3) * 8 + rem % tail;
      mb = rem / tail;
    }
    tile0 = mb * MB;
  }
  // (row tile j of the block: tile0 + j; a grouped item with fewer than MB tiles reads its last tile again)
  auto tile_of = [&](int j) { return tile0 + (GRP && j >= ntl ? ntl - 1 : j); };
  const int wn = w & 1, wm = w >> 1;  // the wave's 64 columns x 64 rows inside the block tile
  const int KT = a.KS >> 1;           // 64-deep steps

  // Gather in the addresses (grouped form, a.src_row): this wave stages the X fragments of tiles 2 q + (w >> 1), k-step
  // w & 1 (slot = 4 i + w below); lane (g = l >> 4, row r = l & 15) of such a fragment is 16 bytes of SOURCE row
  // m = src_row[tile * 16 + r], i.e. lane g * 16 + (m & 15) of fragment (m >> 4, k-step) of the ungathered tiles.
  int64_t xsrc[MB / 2];  // element offset (bf16x8 units) of this lane's piece at k-step 0; < 0: a padding row
#pragma unroll
  for (int q = 0; q < MB / 2; ++q) {
    xsrc[q] = -1;
    if (GRP && a.src_row) {
      const int m = a.src_row[tile_of(2 * q + (w >> 1)) * 16 + (l & 15)];
      if (m >= 0) xsrc[q] = ((int64_t)(m >> 4) * a.KS) * 64 + (l >> 4) * 16 + (m & 15);
    }
  }
  // stage kt -> buffer: 16 weight fragments + 2 MB activation fragments of 1 KiB, dealt round-robin to the 4 waves
  auto stage = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < (16 + 2 * MB) / 4; ++i) {
      const int slot = i * 4 + w;  // wave-uniform
      const int s16 = slot < 16 ? slot : slot - 16;  // fragment of its operand: (tile s16 >> 1, k-step s16 & 1)
      const bf16x8 *src = (slot < 16 ? wbase + ((size_t)(nb * 8 + (s16 >> 1)) * a.KS + kt * 2 + (s16 & 1)) * 64
                                     : a.xf + ((size_t)tile_of(s16 >> 1) * a.KS + kt * 2 + (s16 & 1)) * 64) + l;
      if (GRP && i >= 4 && a.src_row) {
        const int64_t o = xsrc[i >= 4 ? i - 4 : 0];
        src = o < 0 ? a.zeros + l : a.xf + o + (kt * 2 + (w & 1)) * 64;
      }
      __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)&lds[buf][slot][0], 16, 0, 0);
    }
  };

  f32x4 acc[4][MW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < KT) stage(i, i);
  int buf = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // this wave's share of stage kt has landed in LDS (vmcnt is in order: min(NST - 2, KT - 1 - kt) later stages may
    // still be in flight)
    const int later = KT - 1 - kt;
    if (NST >= 3 && later >= 1) {
      if (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ... and everyone's; every wave has finished reading the buffer of iteration kt - 1 (its MFMAs have consumed the
    // ds_reads).  A bare s_barrier: __syncthreads() carries a fence, i.e. vmcnt(0) — the stages in flight.
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NST - 1 < KT) stage(kt + NST - 1, buf == 0 ? NST - 1 : buf - 1);  // into the buffer of iteration kt - 1
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bq[MW];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = lds[buf][(wn * 4 + i) * 2 + ks][l];
#pragma unroll
      for (int j = 0; j < MW; ++j) bq[j] = lds[buf][16 + (wm * MW + j) * 2 + ks][l];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bq[j], acc[i][j], 0, 0, 0);
    }
    buf = buf == NST - 1 ? 0 : buf + 1;
  }

  pgemm_epilogue<EPI, MW, GRP>(a, acc, nb, wn, wm, tile0, ntl, l);
}

// rows [P][H] -> (RMSNorm) -> frag16 row tiles.  One WAVE per row (grid = row tiles x 4, four rows per workgroup): the
// row's chunks stay in registers between the sum of squares and the normalisation (one pass over memory; the first
// form, one workgroup per 16-row tile with 16 threads per row, took 22 us for 1024 x 4096: 64 workgroups, two passes).
// norm_w == nullptr: pack only.  Rows >= P give zero fragments.  H <= 64 * 8 * PN_MAXC.
constexpr int PN_MAXC = 16;  // H <= 8192 per call (a row's chunks stay in registers)
// ks_total / ks0: the tiles' k-step count and the k-step these H columns start at (a column chunk of wider rows).
__global__ __launch_bounds__(256) void k_pnorm_pack(const bf16_t *h, int64_t ldh, int P, int H, const bf16_t *norm_w,
                                                    float eps, bf16x8 *xf, int ks_total, int ks0) {
  const int mt = blockIdx.x, l = threadIdx.x & 63;
  const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int row = mt * 16 + m;
  const int nch = H >> 3;
  const bf16_t *src = h + (int64_t)row * ldh;
  bf16x8 v[PN_MAXC];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < PN_MAXC; ++i) {
    const int c = l + 64 * i;
    v[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (row < P && c < nch) v[i] = *reinterpret_cast<const bf16x8 *>(src + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += bf2f(v[i][j]) * bf2f(v[i][j]);
  }
  float rstd = 1.f;
  if (norm_w) rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
  bf16x8 *dst = xf + ((size_t)mt * ks_total + ks0) * 64;
#pragma unroll
  for (int i = 0; i < PN_MAXC; ++i) {
    const int c = l + 64 * i;
    if (c < nch) {
      bf16x8 o = v[i];
      if (norm_w) {  // Qwen3RMSNorm: weight * bf16(x * rstd), tf:modeling_qwen3.py:59-64
        const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(norm_w + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wv[j]) * rbf(bf2f(v[i][j]) * rstd));
      }
      dst[(size_t)c * 16 + m] = o;  // chunk c = k-step c >> 2, lane (c & 3) * 16 + m
    }
  }
}

// q/k-norm + RoPE over P prompt rows (tf:modeling_qwen3.py:205-215 in the order model/dflash.py's target runs it): one
// wave per (row, head) item of 128 values, lane owns d = l and l + 64 (rotate_half pairs).  q rows are rewritten in
// place inside the qkv row buffer; k (normed, rotated) and v go to cache rows row0 + row.
struct PRopeArgs {
  bf16_t *qkv;
  int64_t ld;
  int P, q_col, k_col, v_col, n_q, n_kv;
  const bf16_t *q_w, *k_w;
  float eps;
  const bf16_t *cos_tab, *sin_tab;
  int max_pos, pos0;
  bf16_t *kcache, *vcache;
  int cache_rows, row0;
};

__global__ __launch_bounds__(256) void k_pqk_rope(PRopeArgs a) {
  const int l = threadIdx.x & 63;
  const int per_row = a.n_q + 2 * a.n_kv;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int row = (int)(item / per_row), hh = (int)(item % per_row);
  if (row >= a.P) return;
  const int kind = hh < a.n_q ? 0 : (hh < a.n_q + a.n_kv ? 1 : 2);
  const int head = kind == 0 ? hh : (kind == 1 ? hh - a.n_q : hh - a.n_q - a.n_kv);
  bf16_t *src = a.qkv + (int64_t)row * a.ld + (kind == 0 ? a.q_col : kind == 1 ? a.k_col : a.v_col) + head * 128;
  const float x1 = bf2f(src[l]), x2 = bf2f(src[l + 64]);
  const int crow = a.row0 + row;
  if (kind == 2) {
    if (crow < a.cache_rows) {
      bf16_t *dst = a.vcache + ((int64_t)head * a.cache_rows + crow) * 128;
      dst[l] = f2bf(x1);
      dst[l + 64] = f2bf(x2);
    }
    return;
  }
  const bf16_t *nw = kind == 0 ? a.q_w : a.k_w;
  float n1 = x1, n2 = x2;
  if (nw) {
    const float ss = wave_sum(x1 * x1 + x2 * x2);
    const float rstd = rsqrtf(ss * (1.f / 128.f) + a.eps);
    n1 = rbf(bf2f(nw[l]) * rbf(x1 * rstd));
    n2 = rbf(bf2f(nw[l + 64]) * rbf(x2 * rstd));
  }
  int pos = a.pos0 + row;
  pos = pos < a.max_pos ? pos : a.max_pos - 1;
  const float c = bf2f(a.cos_tab[(int64_t)pos * 64 + l]);
  const float sn = bf2f(a.sin_tab[(int64_t)pos * 64 + l]);
  const float o1 = rbf(rbf(n1 * c) + rbf(-n2 * sn));
  const float o2 = rbf(rbf(n2 * c) + rbf(n1 * sn));
  if (kind == 0) {
    src[l] = f2bf(o1);
    src[l + 64] = f2bf(o2);
  } else if (crow < a.cache_rows) {
    bf16_t *dst = a.kcache + ((int64_t)head * a.cache_rows + crow) * 128;
    dst[l] = f2bf(o1);
    dst[l + 64] = f2bf(o2);
  }
}

// Causal attention of the prompt over itself (tf:modeling_qwen3.py eager / SDPA attention of the target's prefill
// forward, model/dflash.py:218-225): softmax(q k^T * scale, causal) v per query head, GQA (kernel: k_pattn below).
// q: post-norm, post-RoPE bf16 rows (dfl_prefill_qk_rope rewrote them in place); K/V: the cache rows it wrote.
// Output: frag16 row tiles of n_q * 128 columns — o_proj's operand, no pack step in between.
struct PAttnArgs {
  const bf16_t *q;
  int64_t ldq;
  int q_col;
  const bf16_t *kc, *vc;
  int cache_rows, P, G;
  float scale_log2;
  bf16x8 *out_frag;
  int KSo;  // n_q * 128 / 32
  int hq;   // query heads per workgroup (1, 2 or 4, all of one kv head): its waves = hq heads x PA_NW / hq query tiles
};

typedef __attribute__((address_space(3))) bf16x4 plds_bf16x4;
__device__ __forceinline__ int pv_swz(int row, int ch) { return row * 256 + ((((ch >> 1) ^ (row & 7)) << 5) | ((ch & 1) << 4)); }
__device__ __forceinline__ void pswap16(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void pswap32(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float pg_max(float v) {  // max over the four 16-lane rows of the wave (csrc/attn_head.hip)
  float a = v, b = v;
  pswap16(a, b);
  a = b = fmaxf(a, b);
  pswap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float pg_sum(float v) {
  float a = v, b = v;
  pswap16(a, b);
  a = b = a + b;
  pswap32(a, b);
  return a + b;
}

// Causal attention of the prompt over itself, flash-style: a workgroup = 4 waves = hq query heads of ONE kv head x 4 / hq
// 16-row query tiles, walking the 32-key tiles of that kv head together.  A tile (K 8 KB + V 8 KB) enters LDS once per
// workgroup by LDS-DMA into a ring of PA_NS stages, three tiles in flight under the MFMAs of the current one (first form:
// every wave fetched its own K / V tile into registers, one tile ahead — 2.5 us per tile and wave at P = 1024, the L2
// latency).  The DMA writes a lane-linear image, so the bank swizzles are applied on the SOURCE side: lane j of a 1 KiB
// request fetches the 16-byte chunk that belongs in slot j.  K rows: chunk c of key k sits at k * 256 + ((c ^ (k & 15)) << 4)
// (conflict-free ds_read_b128 of the MFMA A fragments: lane (key qi, g) reads chunk 4 s + g); V rows: pv_swz (the layout
// ds_read_b64_tr_b16 wants, csrc/attn_head.hip).  S^T = K Q^T, O^T += V^T P^T, base-2 online softmax, P in bf16.
constexpr int PA_NS = 4, PA_NW = 8;  // ring stages; waves per workgroup (hq heads x PA_NW / hq query tiles)
__global__ __launch_bounds__(64 * PA_NW) void k_pattn(PAttnArgs a) {
  __shared__ __attribute__((aligned(1024))) char lds[PA_NS][2][8192];  // [stage][K | V]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, qi = l & 15, g = l >> 4;
  // blockIdx.x = head group (the fast index: with 8 head groups a kv head's workgroups share an XCD and its L2),
  // blockIdx.y = query tile group
  const int head = blockIdx.x * a.hq + w % a.hq;
  const int nqt = (a.P + 15) >> 4;
  const int t_top = nqt - 1 - (int)blockIdx.y * (PA_NW / a.hq);  // the workgroup's longest query tile (late tiles first)
  const int t = t_top - w / a.hq;                                // this wave's; < 0: no tile (it still feeds the ring)
  const bf16_t *K = a.kc + (int64_t)(head / a.G) * a.cache_rows * 128;
  const bf16_t *V = a.vc + (int64_t)(head / a.G) * a.cache_rows * 128;
  const int qrow = t * 16 + qi, qrow_c = qrow < 0 ? 0 : (qrow < a.P ? qrow : a.P - 1);
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *reinterpret_cast<const bf16x8 *>(a.q + (int64_t)qrow_c * a.ldq + a.q_col + head * 128 + 32 * s + 8 * g);
  const int kend_wg = (t_top * 16 + 16 < a.P ? t_top * 16 + 16 : a.P);  // keys [0, kend) are visible to some row of the tile
  const int ntile_wg = (kend_wg + 31) >> 5;
  const int kend = t < 0 ? 0 : (t * 16 + 16 < a.P ? t * 16 + 16 : a.P);
  const int ntile = (kend + 31) >> 5;
  // this wave's share of a stage: the 1 KiB requests (4 rows each) w, w + PA_NW, .. < 8 of the K tile and of the V tile; rows past
  // the last prompt row are clamped (their scores are masked); past the last tile the last tile is requested again, so
  // that the number of requests in flight is the same in every iteration
  auto stage = [&](int tile) {
    const int tl = tile < ntile_wg ? tile : ntile_wg - 1;
    char *base = &lds[tile % PA_NS][0][0];
#pragma unroll
    for (int j = 0; j < 8 / PA_NW; ++j) {
      const int rq = w + j * PA_NW;  // request index within the tile
      const int rl = rq * 4 + (l >> 4), pc = l & 15;
      int row = tl * 32 + rl;
      row = row < a.P ? row : a.P - 1;
      const int ck = pc ^ (rl & 15);
      const int cv = ((((pc >> 1) ^ (rl & 7)) << 1) | (pc & 1));
      __builtin_amdgcn_global_load_lds((glb_void *)(K + (int64_t)row * 128 + ck * 8), (lds_void *)(base + rq * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void *)(V + (int64_t)row * 128 + cv * 8), (lds_void *)(base + 8192 + rq * 1024), 16,
                                       0, 0);
    }
  };
  f32x4 o[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  auto compute = [&](const char *kt, const char *vt, int key0) {
    const int qq = qi >> 2, p4 = l & 3;
    const int r0 = 4 * g + qq, r1 = 16 + 4 * g + qq;
    f32x4 sc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      sc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(kt + (u * 16 + qi) * 256 + (((s * 4 + g) ^ qi) << 4));
        sc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], sc[u], 0, 0, 0);
      }
    }
    float mx = -INFINITY;  // lane (q = qi, g): sc[u][r] is key key0 + u * 16 + 4 g + r
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + u * 16 + 4 * g + r;
        const float v = (key <= qrow && key < a.P) ? sc[u][r] * a.scale_log2 : -INFINITY;
        sc[u][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = pg_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float m_ref = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_ref);
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pb;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(sc[u][r] - m_ref);
        psum += p;
        pb[u * 4 + r] = f2bf(p);
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] *= alpha;
    // The transposed V reads go through inline asm: behind the builtin hipcc puts s_waitcnt vmcnt(0) in front of every
    // ds_read_b64_tr_b16 (an LDS read that "may alias" the LDS-DMA requests in flight — the ring), i.e. the tiles in
    // flight.  The asm issues the reads of four d tiles and waits for them itself; "memory" keeps the compiler's own LDS
    // reads (whose lgkmcnt it counts itself) on their side of the block.
    // (rows past the prompt hold copies of its last row: finite values under P = 0)
    const unsigned vb = (unsigned)(uintptr_t)(plds_bf16x4 *)vt;
    const unsigned a0 = vb + r0 * 256 + (p4 << 3), a1 = vb + r1 * 256 + (p4 << 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x4 v0[4], v1[4];
      unsigned ad0[4], ad1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ad0[i] = a0 + (((h * 4 + i) ^ (r0 & 7)) << 5);
        ad1[i] = a1 + (((h * 4 + i) ^ (r1 & 7)) << 5);
      }
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %12\n\t"
          "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %13\n\t"
          "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %14\n\t"
          "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %15\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v0[0]), "=&v"(v1[0]), "=&v"(v0[1]), "=&v"(v1[1]), "=&v"(v0[2]), "=&v"(v1[2]), "=&v"(v0[3]), "=&v"(v1[3])
          : "v"(ad0[0]), "v"(ad0[1]), "v"(ad0[2]), "v"(ad0[3]), "v"(ad1[0]), "v"(ad1[1]), "v"(ad1[2]), "v"(ad1[3])
          : "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 va = {v0[i][0], v0[i][1], v0[i][2], v0[i][3], v1[i][0], v1[i][1], v1[i][2], v1[i][3]};
        o[h * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[h * 4 + i], 0, 0, 0);
      }
    }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the q fragments: from here on vmcnt counts ring requests only
#pragma unroll
  for (int i = 0; i < PA_NS - 1; ++i) stage(i);
  for (int tc = 0; tc < ntile_wg; ++tc) {
    // this wave's requests of tile tc have landed (16 / PA_NW per stage, PA_NS - 2 later stages may still be in flight) ...
    if (PA_NW == 8)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    static_assert(PA_NS == 4 && (PA_NW == 4 || PA_NW == 8), "vmcnt immediates above = (16 / PA_NW) * (PA_NS - 2)");
    // ... and every wave's; every wave is done with tile tc - 1, whose stage the next request overwrites.  A bare
    // s_barrier: __syncthreads() carries a fence, i.e. vmcnt(0) — the tiles in flight.
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage(tc + PA_NS - 1);
    if (tc < ntile) compute(&lds[tc % PA_NS][0][0], &lds[tc % PA_NS][1][0], tc * 32);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (requests of the tail still target this workgroup's LDS)
  if (t < 0) return;
  // O^T tile dt: lane (q = qi, g) holds d = dt * 16 + 4 g + r  ->  frag16: column k = head * 128 + d of row tile t
  const float inv = 1.f / pg_sum(l_run);
  bf16x8 *base = a.out_frag + (size_t)t * a.KSo * 64;
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) {
    bf16x4 ov;
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = f2bf(qrow < a.P ? o[dt][r] * inv : 0.f);
    bf16_t *dst = reinterpret_cast<bf16_t *>(base + (head * 4 + (dt >> 1)) * 64 + ((dt & 1) * 2 + (g >> 1)) * 16 + qi) + (g & 1) * 4;
    *reinterpret_cast<bf16x4 *>(dst) = ov;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Sparse-MoE MLP over the P prompt rows (Qwen3MoeSparseMoeBlock, tf:models/qwen3_moe/modeling_qwen3_moe.py; the
// reference runs it inside the HF forward of model/dflash.py:218-225, a Python loop over the experts).  The (row, slot)
// pairs of a layer are sorted by expert on the device and each expert's rows gathered into 16-row frag16 tiles, so
// that every expert's weights are read ONCE by MFMA row blocks of its own rows (the grouped form of k_pgemm):
//   k_pmoe_route    one wave per row: fp32 softmax / top-k / renormalise (moe_route.h) -> (expert, weight) per slot
//   k_pmoe_plan     one workgroup: counts and ranks per expert, tile offsets, the work list (expert, first tile,
//                   tiles <= 4) of the grouped GEMMs, source row and routing weight of every gathered row
//   k_pmoe_gather   one workgroup per gathered tile: the source rows' normalised fragments (zero for padding rows)
//   k_pgemm<SILU, grouped>, k_pgemm<SCALE32, grouped>: act = silu(x Wg_e^T) * (x Wu_e^T); out32 = w * (act Wd_e^T)
//   k_pmoe_combine  one wave per row: the k slot rows summed in fp32 (slot order), rounded once, residual add (+ tap) —
//                   the rounding points of the decode-side kernels (moe.hip), fewer than HF's per-expert bf16 adds.
__global__ __launch_bounds__(256) void k_pmoe_route(const bf16_t *logits, int64_t ld, int P, int E, int top_k, int norm_topk,
                                                    int32_t *pair_e, float *pair_w) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (m >= P) return;
  float sel_v[8], tot;
  int sel_i[8];
  route_row(logits + m * ld, E, top_k, l, sel_v, sel_i, tot);
  if (l == 0) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r < top_k) {
        const float w = norm_topk ? sel_v[r] / tot : sel_v[r];
        pair_e[m * 8 + r] = sel_i[r];
        pair_w[m * 8 + r] = rbf(w);  // the routing weights are cast to the hidden dtype (Qwen3MoeTopKRouter.forward)
      }
  }
}

// ONE workgroup: per-expert counts and each pair's rank among its expert's rows (LDS atomics: the order inside an
// expert's tiles is arbitrary, the arithmetic of a row does not depend on it), tile offsets (prefix over the experts),
// the work list of the grouped GEMMs, and for every gathered row its source row and routing weight (-1 / 0: padding).
__global__ __launch_bounds__(1024) void k_pmoe_plan(const int32_t *pair_e, const float *pair_w, int P, int E, int top_k,
                                                    int32_t *cnt_out, int32_t *tile_off, int32_t *items, int32_t *n_items,
                                                    int32_t *posmap, int32_t *src_row, float *row_w, int max_rows, int tpi) {
  __shared__ int cnt[256], toff[256], ioff[256];
  const int tid = threadIdx.x;
  if (tid < 256) cnt[tid] = 0;
  for (int i = tid; i < max_rows; i += 1024) {
    src_row[i] = -1;
    row_w[i] = 0.f;
  }
  __syncthreads();
  const int npair = P * top_k;
  // (ranks are kept in posmap until the offsets are known; eight pairs per thread and pass, their loads requested
  // together.  The kernel stays at 13 - 19 us for the 8192 pairs of a 1k-row prompt either way: one workgroup, and its LDS
  // atomics meet on 128 counters)
  for (int i0 = tid; i0 < npair; i0 += 8 * 1024) {
    int e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 1024, m = i / top_k, r = i - m * top_k;
      e[u] = i < npair ? pair_e[m * 8 + r] : 0;
      e[u] = (unsigned)e[u] < (unsigned)E ? e[u] : 0;  // (route_row keeps its indices < E even for NaN rows; never index LDS by a foreign value)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 1024, m = i / top_k, r = i - m * top_k;
      if (i < npair) posmap[m * 8 + r] = atomicAdd(&cnt[e[u]], 1);
    }
  }
  __syncthreads();
  if (tid < 64) {  // one wave: exclusive prefix of tiles and of work items over the experts, 4 experts per lane
    int nt[4], ni[4], st = 0, si = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid * 4 + j;
      nt[j] = e < E ? (cnt[e] + 15) >> 4 : 0;
      ni[j] = (nt[j] + tpi - 1) / tpi;
      st += nt[j];
      si += ni[j];
    }
    int pt = st, pi = si;  // inclusive scan over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int ot = __shfl_up(pt, o, 64), oi = __shfl_up(pi, o, 64);
      if (tid >= o) {
        pt += ot;
        pi += oi;
      }
    }
    int bt = pt - st, bi = pi - si;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid * 4 + j;
      if (e < 256) {
        toff[e] = bt;
        ioff[e] = bi;
      }
      bt += nt[j];
      bi += ni[j];
    }
    if (tid == 63) {
      n_items[0] = pi;
      n_items[1] = pt;
    }
  }
  __syncthreads();
  if (tid < E) {
    cnt_out[tid] = cnt[tid];
    tile_off[tid] = toff[tid];
    const int nt = (cnt[tid] + 15) >> 4;
    for (int b = 0, w = ioff[tid]; b < nt; b += tpi, ++w) {
      items[3 * w] = tid;
      items[3 * w + 1] = toff[tid] + b;
      items[3 * w + 2] = nt - b < tpi ? nt - b : tpi;
    }
  }
  for (int i0 = tid; i0 < npair; i0 += 8 * 1024) {
    int e[8], rk[8];
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 1024, m = i / top_k, r = i - m * top_k, k = i < npair ? m * 8 + r : 0;
      e[u] = pair_e[k];
      e[u] = (unsigned)e[u] < (unsigned)E ? e[u] : 0;
      rk[u] = posmap[k];
      wv[u] = pair_w[k];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 1024, m = i / top_k, r = i - m * top_k;
      if (i < npair) {
        const int pos = toff[e[u]] * 16 + rk[u];
        posmap[m * 8 + r] = pos;
        src_row[pos] = m;
        row_w[pos] = wv[u];
      }
    }
  }
}

// one workgroup per gathered tile: chunk c (8 values) of its 16 rows is one contiguous 256-byte run of the tile
__global__ __launch_bounds__(256) void k_pmoe_gather(const bf16x8 *xf, int KS, const int32_t *src_row, const int32_t *n_items,
                                                     bf16x8 *xg) {
  const int t = blockIdx.x;
  if (t >= n_items[1]) return;
  const int r = threadIdx.x & 15;
  const int m = src_row[t * 16 + r];
  const bf16x8 *src = xf + (size_t)((m < 0 ? 0 : m) >> 4) * KS * 64 + ((m < 0 ? 0 : m) & 15);
  bf16x8 *dst = xg + (size_t)t * KS * 64 + r;
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = threadIdx.x >> 4; c < KS * 4; c += 16) dst[c * 16] = m < 0 ? z : src[c * 16];
}

// one workgroup per row, its four waves a quarter of the columns each
// sum_out (optional): the rows' fp32 sums are stored there ([P][H]) and h is left alone — the ragged-batch decode path adds
// them to the residual stream in its next norm launch (one rounding there).
__global__ __launch_bounds__(256) void k_pmoe_combine(const float *out32, int64_t ld32, const int32_t *posmap, int P, int H,
                                                      int top_k, bf16_t *h, int64_t ldh, bf16_t *tap, int64_t ldtap,
                                                      float *sum_out) {
  const int m = blockIdx.x;
  int pos[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) pos[r] = r < top_k ? posmap[m * 8 + r] : 0;
  for (int c = threadIdx.x; c < H / 4; c += 256) {
    f32x4 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      v[r] = r < top_k ? *reinterpret_cast<const f32x4 *>(out32 + (int64_t)pos[r] * ld32 + c * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += v[r][j];  // slot order (the padding slots of top_k < 8 add zero)
      sum[j] = s;
    }
    if (sum_out) {
      *reinterpret_cast<f32x4 *>(sum_out + (int64_t)m * H + c * 4) = sum;
      continue;
    }
    bf16_t *hp = h + (int64_t)m * ldh + c * 4;
    const bf16x4 hv = *reinterpret_cast<const bf16x4 *>(hp);
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f2bf(rbf(bf2f(hv[j]) + rbf(sum[j])));
    *reinterpret_cast<bf16x4 *>(hp) = o;
    if (tap) *reinterpret_cast<bf16x4 *>(tap + (int64_t)m * ldtap + c * 4) = o;
  }
}

bool pmoe_gemm_fill(PGemmArgs &a, const void *wp, int64_t w_expert_stride, const void *xg, const int32_t *items,
                    const int32_t *n_items, int max_items, int N, int K, const char *who) {
  if (!wp || !xg || !items || !n_items) {
    dfl_set_error("%s: null pointer", who);
    return false;
  }
  if (max_items < 1 || N <= 0 || K <= 0 || N % 128 || K % 64 || w_expert_stride != (int64_t)N * K) {
    dfl_set_error("%s: need N%%128==0, K%%64==0, expert stride = N*K (N=%d K=%d)", who, N, K);
    return false;
  }
  a.wp = (const bf16x8 *)wp;
  a.xf = (const bf16x8 *)xg;
  a.KS = K / 32;
  a.ntiles = N / 16;
  a.mtiles = 4;
  a.P = 0;
  a.items = items;
  a.n_items = n_items;
  a.w_expert_stride = w_expert_stride / 8;
  return true;
}

bool pgemm_fill(PGemmArgs &a, const void *wp, const void *xf, int P, int N, int K, const char *who) {
  if (!wp || !xf) {
    dfl_set_error("%s: null pointer", who);
    return false;
  }
  if (P < 1 || N <= 0 || K <= 0 || N % 128 || K % 64) {
    dfl_set_error("%s: need P >= 1, N%%128==0, K%%64==0 (P=%d N=%d K=%d)", who, P, N, K);
    return false;
  }
  a.wp = (const bf16x8 *)wp;
  a.xf = (const bf16x8 *)xf;
  a.KS = K / 32;
  a.ntiles = N / 16;
  a.mtiles = (P + 127) / 128 * 8;
  a.P = P;
  return true;
}

template <int EPI>
void pgemm_launch(const PGemmArgs &a, hipStream_t st) {
  // 128-row blocks unless that leaves fewer than two workgroups per CU
  // LDS stages, measured at P = 1024 on the 8B shapes (scripts/bench_prefill.py, whole prefill, same box):
  // 128-row blocks 2 / 3 / 4 stages (64 / 96 / 128 KiB): 20.9 / 22.5 / 22.5 ms — two workgroups per CU beat a deeper
  // queue; 64-row blocks 2 / 3 / 4 stages (48 / 72 / 96 KiB): 21.4 / 20.9 / 22.6 ms.  The wave tile (64 x 64) reads
  // 16 KiB of LDS per 32 MFMAs, i.e. the whole LDS bandwidth at the MFMA peak: the kernel is LDS-bound, not latency-bound.
  if ((a.ntiles / 8) * (a.mtiles / 8) >= 512)
    hipLaunchKernelGGL((k_pgemm<EPI, 4, 2>), dim3((a.ntiles / 8) * (a.mtiles / 8)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_pgemm<EPI, 2, 3>), dim3((a.ntiles / 8) * (a.mtiles / 4)), dim3(256), 0, st, a);
}

}  // namespace

extern "C" int64_t dfl_prefill_rows_padded(int P) { return (int64_t)(P + 127) / 128 * 128; }

extern "C" int dfl_prefill_gemm_rows(const void *wp, const void *x_frag, int P, int N, int K, void *out, int64_t ldo,
                                     void *stream) {
  PGemmArgs a{};
  DFL_REQUIRE(out && ldo >= N && ldo % 4 == 0, "dfl_prefill_gemm_rows: bad output");
  if (!pgemm_fill(a, wp, x_frag, P, N, K, "dfl_prefill_gemm_rows")) return DFL_EINVAL;
  a.out = (bf16_t *)out;
  a.ldo = ldo;
  pgemm_launch<PEPI_ROWS>(a, (hipStream_t)stream);
  DFL_CHECK_LAUNCH("dfl_prefill_gemm_rows");
  return DFL_OK;
}

extern "C" int dfl_prefill_gemm_resid(const void *wp, const void *x_frag, int P, int N, int K, void *h_io, int64_t ldh,
                                      void *tap, int64_t ldtap, void *stream) {
  PGemmArgs a{};
  DFL_REQUIRE(h_io && ldh >= N && ldh % 4 == 0 && (!tap || (ldtap >= N && ldtap % 4 == 0)), "dfl_prefill_gemm_resid: bad rows");
  if (!pgemm_fill(a, wp, x_frag, P, N, K, "dfl_prefill_gemm_resid")) return DFL_EINVAL;
  a.out = (bf16_t *)h_io;
  a.ldo = ldh;
  a.tap = (bf16_t *)tap;
  a.ldtap = ldtap;
  pgemm_launch<PEPI_RESID>(a, (hipStream_t)stream);
  DFL_CHECK_LAUNCH("dfl_prefill_gemm_resid");
  return DFL_OK;
}

extern "C" int dfl_prefill_gemm_silu(const void *wp_gateup, const void *x_frag, int P, int I, int K, void *act_frag,
                                     void *stream) {
  PGemmArgs a{};
  DFL_REQUIRE(act_frag && I > 0 && I % 64 == 0, "dfl_prefill_gemm_silu: need I%%64==0");
  if (!pgemm_fill(a, wp_gateup, x_frag, P, 2 * I, K, "dfl_prefill_gemm_silu")) return DFL_EINVAL;
  a.act = (bf16x8 *)act_frag;
  a.KSo = I / 32;
  pgemm_launch<PEPI_SILU>(a, (hipStream_t)stream);
  DFL_CHECK_LAUNCH("dfl_prefill_gemm_silu");
  return DFL_OK;
}

extern "C" int dfl_prefill_norm_pack(const void *h, int64_t ldh, int P, int H, const void *norm_w, float eps,
                                     void *x_frag, void *stream) {
  DFL_REQUIRE(h && x_frag && P >= 1 && H > 0 && H % 32 == 0 && H <= 64 * 8 * PN_MAXC && ldh >= H && ldh % 8 == 0,
              "dfl_prefill_norm_pack: bad shape (H %% 32, H <= %d)", 64 * 8 * PN_MAXC);
  const int mtiles = (P + 127) / 128 * 8;  // the padded row tiles are written too (zero fragments)
  hipLaunchKernelGGL(k_pnorm_pack, dim3(mtiles, 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)h, ldh, P, H,
                     (const bf16_t *)norm_w, eps, (bf16x8 *)x_frag, H / 32, 0);
  DFL_CHECK_LAUNCH("dfl_prefill_norm_pack");
  return DFL_OK;
}

extern "C" int dfl_prefill_pack_rows(const void *rows, int64_t ld, int P, int K, void *x_frag, void *stream) {
  DFL_REQUIRE(rows && x_frag && P >= 1 && K > 0 && K % 32 == 0 && ld >= K && ld % 8 == 0, "dfl_prefill_pack_rows: bad shape (K %% 32)");
  const int mtiles = (P + 127) / 128 * 8, chunk = 64 * 8 * PN_MAXC;
  for (int c0 = 0; c0 < K; c0 += chunk) {  // (a wave keeps at most `chunk` columns of its row in registers)
    const int hc = K - c0 < chunk ? K - c0 : chunk;
    hipLaunchKernelGGL(k_pnorm_pack, dim3(mtiles, 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)rows + c0, ld, P, hc,
                       (const bf16_t *)nullptr, 0.f, (bf16x8 *)x_frag, K / 32, c0 / 32);
  }
  DFL_CHECK_LAUNCH("dfl_prefill_pack_rows");
  return DFL_OK;
}

extern "C" int dfl_prefill_qk_rope(void *qkv_rows, int64_t ld, int P, int q_col, int k_col, int v_col, int n_q, int n_kv,
                                   const void *q_norm_w, const void *k_norm_w, float eps, const void *cos_tab,
                                   const void *sin_tab, int max_pos, int pos0, void *kcache, void *vcache,
                                   int cache_rows, int row0, void *stream) {
  DFL_REQUIRE(qkv_rows && cos_tab && sin_tab && kcache && vcache, "dfl_prefill_qk_rope: null pointer");
  DFL_REQUIRE(P >= 1 && n_q >= 0 && n_kv >= 1 && max_pos >= 1 && pos0 >= 0 && row0 >= 0 && row0 + P <= cache_rows,
              "dfl_prefill_qk_rope: bad lengths (P=%d row0=%d cache_rows=%d)", P, row0, cache_rows);
  PRopeArgs a{(bf16_t *)qkv_rows, ld, P, q_col, k_col, v_col, n_q, n_kv, (const bf16_t *)q_norm_w,
              (const bf16_t *)k_norm_w, eps, (const bf16_t *)cos_tab, (const bf16_t *)sin_tab, max_pos, pos0,
              (bf16_t *)kcache, (bf16_t *)vcache, cache_rows, row0};
  const int64_t items = (int64_t)P * (n_q + 2 * n_kv);
  hipLaunchKernelGGL(k_pqk_rope, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_prefill_qk_rope");
  return DFL_OK;
}

extern "C" int dfl_prefill_attn(const void *q_rows, int64_t ldq, int q_col, const void *kcache, const void *vcache,
                                int cache_rows, int P, int n_q, int n_kv, float scale, void *out_frag, void *stream) {
  DFL_REQUIRE(q_rows && kcache && vcache && out_frag, "dfl_prefill_attn: null pointer");
  DFL_REQUIRE(P >= 1 && P <= cache_rows && n_q >= 1 && n_kv >= 1 && n_q % n_kv == 0 && ldq % 8 == 0 && q_col % 8 == 0,
              "dfl_prefill_attn: bad shape (P=%d cache_rows=%d n_q=%d n_kv=%d)", P, cache_rows, n_q, n_kv);
  const int G = n_q / n_kv;
  int hq = G % 4 == 0 ? 4 : (G % 2 == 0 ? 2 : 1);
  // (measured, 8B shapes, whole prefill, with the first form of k_pattn: P = 1024 20.85 ms either way; P = 4096, 12 layers:
  // 33.0 -> 28.3 ms; then the shared K / V ring: 19.7 / 24.8 ms with 4-wave workgroups, 19.3 / 23.6 ms with 8 waves)
  PAttnArgs a{(const bf16_t *)q_rows, ldq, q_col, (const bf16_t *)kcache, (const bf16_t *)vcache, cache_rows, P, G,
              scale * 1.4426950408889634f, (bf16x8 *)out_frag, n_q * 4, hq};
  const int nqt = (P + 15) / 16, per = PA_NW / hq;
  hipLaunchKernelGGL(k_pattn, dim3(n_q / hq, (nqt + per - 1) / per), dim3(64 * PA_NW), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_prefill_attn");
  return DFL_OK;
}

// ---- sparse-MoE MLP of the prefill (see k_pmoe_route above).  Scratch, all caller-owned: pair_e / posmap
// int32 [P][8], pair_w float [P][8], cnt / tile_off int32 [E], items int32 [3 * max_items], n_items int32 [2]
// (items, tiles), src_row int32 / row_w float [max_tiles * 16], xg bf16 [max_tiles * 16 * H],
// with max_items = dfl_prefill_moe_max_items(P, top_k, E) and max_tiles = dfl_prefill_moe_max_tiles(P, top_k, E).
extern "C" int64_t dfl_prefill_moe_max_tiles(int P, int top_k, int E) { return ((int64_t)P * top_k + 15) / 16 + E; }
extern "C" int64_t dfl_prefill_moe_max_items(int P, int top_k, int E) { return ((int64_t)P * top_k + 63) / 64 + E; }
// rows_per_item: 64 or 128 rows (4 or 8 gathered tiles) per work item of the grouped GEMMs, the same value for the
// route call and both GEMMs of a layer.  128 reads an expert's weights once where it serves 65..128 rows (30B-A3B widths,
// P = 1024, ~64 rows per expert: 566 -> 539 us per layer); 64 wastes less on experts with few rows.

extern "C" int dfl_prefill_moe_route(const void *logits, int64_t ld, int P, int E, int top_k, int norm_topk, int32_t *pair_e,
                                     float *pair_w, int32_t *cnt, int32_t *tile_off, int32_t *items, int32_t *n_items,
                                     int32_t *posmap, int32_t *src_row, float *row_w, int rows_per_item, void *stream) {
  DFL_REQUIRE(rows_per_item == 64 || rows_per_item == 128, "dfl_prefill_moe_route: rows_per_item must be 64 or 128");
  DFL_REQUIRE(logits && pair_e && pair_w && cnt && tile_off && items && n_items && posmap && src_row && row_w,
              "dfl_prefill_moe_route: null pointer");
  DFL_REQUIRE(P >= 1 && E >= 1 && E <= 256 && top_k >= 1 && top_k <= 8 && top_k <= E && ld >= E,
              "dfl_prefill_moe_route: need 1 <= top_k <= 8, top_k <= E <= 256, ld >= E (E=%d top_k=%d)", E, top_k);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pmoe_route, dim3((P + 3) / 4), dim3(256), 0, st, (const bf16_t *)logits, ld, P, E, top_k, norm_topk,
                     pair_e, pair_w);
  hipLaunchKernelGGL(k_pmoe_plan, dim3(1), dim3(1024), 0, st, pair_e, pair_w, P, E, top_k, cnt, tile_off, items, n_items, posmap,
                     src_row, row_w, (int)dfl_prefill_moe_max_tiles(P, top_k, E) * 16, rows_per_item / 16);
  DFL_CHECK_LAUNCH("dfl_prefill_moe_route");
  return DFL_OK;
}

extern "C" int dfl_prefill_moe_gather(const void *x_frag, int P, int H, int top_k, int E, const int32_t *src_row,
                                      const int32_t *n_items, void *xg, void *stream) {
  DFL_REQUIRE(x_frag && src_row && n_items && xg, "dfl_prefill_moe_gather: null pointer");
  DFL_REQUIRE(P >= 1 && H > 0 && H % 32 == 0 && top_k >= 1 && top_k <= 8 && E >= 1, "dfl_prefill_moe_gather: bad shape");
  hipLaunchKernelGGL(k_pmoe_gather, dim3((int)dfl_prefill_moe_max_tiles(P, top_k, E)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16x8 *)x_frag, H / 32, src_row, n_items, (bf16x8 *)xg);
  DFL_CHECK_LAUNCH("dfl_prefill_moe_gather");
  return DFL_OK;
}

extern "C" int dfl_prefill_moe_gemm_silu(const void *wp_gateup_e, int64_t w_expert_stride, const void *xg, const int32_t *items,
                                         const int32_t *n_items, int max_items, int I, int K, void *act_g, int rows_per_item,
                                         const int32_t *src_row, const void *zeros_1k, void *stream) {
  PGemmArgs a{};
  DFL_REQUIRE(!src_row == !zeros_1k, "dfl_prefill_moe_gemm_silu: src_row and zeros_1k come together");
  a.src_row = src_row;
  a.zeros = (const bf16x8 *)zeros_1k;
  DFL_REQUIRE(rows_per_item == 64 || rows_per_item == 128, "dfl_prefill_moe_gemm_silu: rows_per_item must be 64 or 128");
  DFL_REQUIRE(act_g && I > 0 && I % 64 == 0, "dfl_prefill_moe_gemm_silu: need I%%64==0");
  if (!pmoe_gemm_fill(a, wp_gateup_e, w_expert_stride, xg, items, n_items, max_items, 2 * I, K, "dfl_prefill_moe_gemm_silu"))
    return DFL_EINVAL;
  a.act = (bf16x8 *)act_g;
  a.KSo = I / 32;
  const dim3 grid((max_items + 7) / 8 * 8 * (a.ntiles / 8));
  if (rows_per_item == 128)
    hipLaunchKernelGGL((k_pgemm<PEPI_SILU, 4, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((k_pgemm<PEPI_SILU, 2, 3, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_prefill_moe_gemm_silu");
  return DFL_OK;
}

extern "C" int dfl_prefill_moe_gemm_down(const void *wp_down_e, int64_t w_expert_stride, const void *act_g, const int32_t *items,
                                         const int32_t *n_items, int max_items, int H, int I, const float *row_w, float *out32,
                                         int rows_per_item, void *stream) {
  PGemmArgs a{};
  DFL_REQUIRE(rows_per_item == 64 || rows_per_item == 128, "dfl_prefill_moe_gemm_down: rows_per_item must be 64 or 128");
  DFL_REQUIRE(row_w && out32, "dfl_prefill_moe_gemm_down: null pointer");
  if (!pmoe_gemm_fill(a, wp_down_e, w_expert_stride, act_g, items, n_items, max_items, H, I, "dfl_prefill_moe_gemm_down"))
    return DFL_EINVAL;
  a.row_w = row_w;
  a.out32 = out32;
  a.ld32 = H;
  const dim3 grid((max_items + 7) / 8 * 8 * (a.ntiles / 8));
  if (rows_per_item == 128)
    hipLaunchKernelGGL((k_pgemm<PEPI_SCALE32, 4, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((k_pgemm<PEPI_SCALE32, 2, 3, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_prefill_moe_gemm_down");
  return DFL_OK;
}

extern "C" int dfl_prefill_moe_combine(const float *out32, const int32_t *posmap, int P, int H, int top_k, void *h_io,
                                       int64_t ldh, void *tap, int64_t ldtap, float *sum_out, void *stream) {
  DFL_REQUIRE(out32 && posmap && (h_io || sum_out), "dfl_prefill_moe_combine: null pointer");
  DFL_REQUIRE(P >= 1 && H > 0 && H % 4 == 0 && top_k >= 1 && top_k <= 8 && (sum_out || (ldh >= H && ldh % 4 == 0)) &&
                  (!tap || (ldtap >= H && ldtap % 4 == 0)),
              "dfl_prefill_moe_combine: bad shape");
  hipLaunchKernelGGL(k_pmoe_combine, dim3(P), dim3(256), 0, (hipStream_t)stream, out32, (int64_t)H, posmap, P, H, top_k,
                     (bf16_t *)h_io, ldh, (bf16_t *)tap, ldtap, sum_out);
  DFL_CHECK_LAUNCH("dfl_prefill_moe_combine");
  return DFL_OK;
}
