// Block attention, second form (round 2): the attention stage of one block in ONE launch, reading
// the q/k/v projections as FINISHED bf16 Linear outputs (dfl_gemm_resid with add_residual = 0)
// instead of fp32 K-split partials.  Replaces model/dflash.py:71-99 (q/k-norm, local RoPE of :22-28,
// cache append, attention over cached prefix + context + block rows) and, with causal = 1, the same
// stage of the target's verify forward (:249-255).
//
// Why a second form (profiles/r1_*: k_attn_fused = 19.6 us per launch for 4.3 MB of K/V): the first
// form's critical path was  q prologue 4.4 -> new K/V rows 2.4 -> 4 LDS-staged tiles of one wave per
// SIMD 5.8 -> release fence + ticket 2.3 -> 9-way merge of 32 KB partials by ONE workgroup 5.4 us.
// This form removes each of those:
//   * grid (kv head, G query heads x splits): a workgroup owns ONE query head, its 8 waves own
//     disjoint 32-key tiles — no LDS staging shared between waves, no barrier in the tile loop:
//     K fragments go HBM/L2 -> VGPR directly in MFMA A-operand order (a lane's 16 B = 8 consecutive
//     d of one key; 16 keys x 64 B per load instruction, whole 256-B rows over the four k-steps),
//     V rows go through a wave-private 8 KB LDS tile for the transposed read (ds_read_b64_tr_b16);
//   * the G workgroups of a kv head have equal blockIdx.x = kv head: with 8 kv heads they share an
//     XCD under round-robin placement, so the K/V rows they all read come from HBM once and from that
//     XCD's L2 afterwards (speed only, never correctness);
//   * the 8 waves' (m, l, O) meet in LDS (64 KB), so a workgroup publishes ONE 8 KB partial per
//     query tile; at 1k keys that is 6 partials per head instead of 9 x 32 KB per kv head, merged by
//     the last arriver of the HEAD (32 mergers side by side instead of 8);
//   * partials are published with write-through (sc1) stores + drain + relaxed ticket and read back
//     with sc1 loads: no release fence (L2 write-back) and no acquire (L1 invalidate) on the path
//     (MI355X_MICROARCH.md, hand-off table row 1);
//   * this cycle's new rows (context + block: <= 64) are a split of their own: its workgroups turn
//     the rows into K/V (norm, RoPE) in LDS, attend over them from there, and the hh = 0 workgroup
//     also appends them to the cache; no workgroup reads a cache row written in this launch;
//   * lengths may come as immediates (host-driven loop) instead of a dependent scalar load.
// MFMA: S^T = K Q^T and O^T += V^T P^T on v_mfma_f32_16x16x32_bf16 as in attn_block.hip (same
// fragment layouts, same base-2 online softmax in fp32, P rounded to bf16 for the PV product).
#include "gemm_rows.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct HeadAttnArgs {
  const bf16_t *xq;  // block rows [bs][ldq]: q | k | v column blocks
  int64_t ldq;
  int q_col, k_col, v_col;
  const bf16_t *xc;  // context rows [tau][ldc] (k, v only); may be null when tau == 0
  int64_t ldc;
  int ck_col, cv_col;
  const bf16_t *q_w, *k_w;
  float eps;
  const bf16_t *cos_tab, *sin_tab;
  int max_pos;
  bf16_t *kc, *vc;
  int cache_rows;
  int n_q, n_kv, G;
  float scale_log2;
  int causal;
  const int32_t *dyn;
  int S, tau, bs, pos0;  // used when dyn == nullptr
  bf16x8 *out_frag;
  int64_t out_tile_stride;  // bf16x8 units between the frag16 buffers of the two query tiles
  float *o_part;   // [ns][n_q][QT*16][128]
  float *ml_part;  // [ns][n_q][QT*16][2]
  int *tickets;    // [n_q]
  int ns_old;      // splits over the cached (old) keys; split ns_old owns the new rows
  // candidate blocks (grid.z = candidate, SURVEY.md §8f-4: several drafts of one block verified against ONE cached
  // prefix): candidate c reads its block rows at xq + c * xq_cand_stride, writes its frag16 output at
  // out_frag + c * out_cand_stride, its partials / tickets at + c * ws_cand_stride floats, and its NEW K/V rows
  // not into the cache but into kv_out rows [0, bs): [c][n_kv][out_rows][128] — the caller copies the winner's.
  // The same dimension carries the REQUESTS of a ragged batch (dfl_attn_head_batch): then every request also has its
  // own length record (dyn + c * dyn_cand_stride ints) and its own cache (+ c * cache_cand_stride elements).
  int64_t xq_cand_stride, out_cand_stride, ws_cand_stride, cache_cand_stride;
  int dyn_cand_stride;
  // XF32 instantiations (round 4: dfl_attn_head_batch_f32): the block rows arrive as the fp32 K-PART SUMS of the qkv GEMM
  // (dfl_gemm_f32_batch: [part][tile rows][ldq]) instead of finished bf16 rows; a row value is bf16(part 0 + part 1) —
  // the Linear's bf16 output, summed in the order the GEMM's own combine used.  The K parts of q/k/v then meet HERE (these
  // loads are a few KB per workgroup) and the qkv launch carries no slabs, ticket or combine phase (18.3 -> 14.7 us at 4
  // tiles).  ldq / the column offsets / xq_cand_stride count floats of one part.
  const float *xq32;
  int64_t part_stride32;  // floats between the parts
  int nparts32;           // 1 or 2
  bf16_t *k_out, *v_out;   // null: the new rows go to the cache at rows S + rel
  int64_t kv_out_cand_stride;
  int out_rows;
};

// 64-lane butterflies on VALU: v_permlane16_swap / v_permlane32_swap (gfx950) instead of
// ds_bpermute round trips (~100 cycles each on a lone wave).  swap16 exchanges the odd 16-lane rows of
// its first operand with the even rows of the second; swap32 the upper half of the first with the lower
// half of the second: started from two copies of x, the two results are x's rows {0,0,2,2} and
// {1,1,3,3} (resp. halves {lo,lo} and {hi,hi}), whose max / sum is the xor-16 (xor-32) butterfly.
// Inline asm, not __builtin_amdgcn_permlane{16,32}_swap: hipcc (ROCm 7.2) treats the builtin's two
// results as equal when its operands are (it reasons per lane) and folds max(r0, r1) to r0.  The
// s_nop covers the VALU-write -> permlane-swap-read hazard the compiler pads the same way.
__device__ __forceinline__ void swap16(float &a, float &b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float &a, float &b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float wave_g_max(float v) {  // max over the four 16-lane rows, in every lane
  float a = v, b = v;
  swap16(a, b);
  a = b = fmaxf(a, b);
  swap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float wave_g_sum(float v) {
  float a = v, b = v;
  swap16(a, b);
  a = b = a + b;
  swap32(a, b);
  return a + b;
}

// NP items per 16-lane group, an item = 128 values of one (row, head): lane c = l & 15 owns
// d = 8c .. 8c+7.  Loads first (all in flight), then Linear-output values -> per-head RMSNorm
// (Qwen3RMSNorm over head_dim, model/dflash.py:72,79; skipped when nw == nullptr) -> RoPE
// (rotate_half pairs lane c with lane c^8: DPP row_ror:8), each product and sum rounded to bf16
// where torch rounds (model/dflash.py:22-28).  src[p] == nullptr marks an absent item.
template <int NP>
struct RopeLoads {
  bf16x8 xv[NP], cs[NP], sn[NP], wv;
  f32x4 pa[NP][2], pb[NP][2];  // XF32: the item's 8 values of K part 0 / part 1 (untouched, i.e. optimised away, otherwise)
};

// issue: the loads only (row values, cos/sin rows, norm weight).  A wave's vector loads return in issue
// order, so these few KB are requested BEFORE the K/V tile burst and arrive within one round trip.
template <int NP>
__device__ __forceinline__ void rope_issue(const bf16_t *const (&src)[NP], const int (&pos)[NP], const bf16_t *nw,
                                           const bf16_t *cos_tab, const bf16_t *sin_tab, int max_pos, const bf16_t *safe,
                                           int l, RopeLoads<NP> &ld) {
  const int d0 = (l & 15) * 8;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    ld.xv[p] = *reinterpret_cast<const bf16x8 *>((src[p] ? src[p] : safe) + d0);
    int pp = pos[p] < max_pos ? pos[p] : max_pos - 1;
    pp = pp < 0 ? 0 : pp;
    ld.cs[p] = *reinterpret_cast<const bf16x8 *>(cos_tab + (int64_t)pp * 64 + (d0 & 63));
    ld.sn[p] = *reinterpret_cast<const bf16x8 *>(sin_tab + (int64_t)pp * 64 + (d0 & 63));
  }
  // unconditional (a load under a branch makes hipcc wait vmcnt(0) at the join): no norm -> 16 B nobody uses
  ld.wv = *reinterpret_cast<const bf16x8 *>((nw ? nw : safe) + d0);
}

// XF32: the same requests for items given as fp32 K-part sums (src[p]: part 0; part 1 at + part_stride; nparts 1: part 1
// is a second read of part 0 that the finish ignores — no load under a branch).  rope_sum32 turns them into the bf16
// Linear outputs rope_finish expects.
template <int NP>
__device__ __forceinline__ void rope_issue32(const float *const (&src)[NP], const int (&pos)[NP], const bf16_t *nw,
                                             const bf16_t *cos_tab, const bf16_t *sin_tab, int max_pos, const float *safe32,
                                             const bf16_t *safe, int64_t part_stride, int nparts, int l, RopeLoads<NP> &ld) {
  const int d0 = (l & 15) * 8;
  const int64_t ps = nparts > 1 ? part_stride : 0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float *b = (src[p] ? src[p] : safe32) + d0;
    ld.pa[p][0] = *reinterpret_cast<const f32x4 *>(b);
    ld.pa[p][1] = *reinterpret_cast<const f32x4 *>(b + 4);
    ld.pb[p][0] = *reinterpret_cast<const f32x4 *>(b + ps);
    ld.pb[p][1] = *reinterpret_cast<const f32x4 *>(b + ps + 4);
    int pp = pos[p] < max_pos ? pos[p] : max_pos - 1;
    pp = pp < 0 ? 0 : pp;
    ld.cs[p] = *reinterpret_cast<const bf16x8 *>(cos_tab + (int64_t)pp * 64 + (d0 & 63));
    ld.sn[p] = *reinterpret_cast<const bf16x8 *>(sin_tab + (int64_t)pp * 64 + (d0 & 63));
  }
  ld.wv = *reinterpret_cast<const bf16x8 *>((nw ? nw : safe) + d0);
}
template <int NP>
__device__ __forceinline__ void rope_sum32(RopeLoads<NP> &ld, int nparts) {
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s0 = ld.pa[p][j >> 2][j & 3], s1 = ld.pb[p][j >> 2][j & 3];
      ld.xv[p][j] = f2bf(nparts > 1 ? s0 + s1 : s0);  // fixed part order, one rounding: the Linear's bf16 output
    }
}

template <int NP>
__device__ __forceinline__ void rope_finish(const RopeLoads<NP> &ld, const bool (&rope)[NP], bool has_norm, float eps, int l,
                                            bf16x8 (&out)[NP]) {
  const int c = l & 15;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    float x[8], n[8];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x[j] = bf2f(ld.xv[p][j]);
      ss += x[j] * x[j];
    }
    if (has_norm) {
      ss += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ss), 0xB1, 0xF, 0xF, true));
      ss += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ss), 0x4E, 0xF, 0xF, true));
      ss += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ss), 0x141, 0xF, 0xF, true));
      ss += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ss), 0x140, 0xF, 0xF, true));
      const float rstd = rsqrtf(ss * (1.f / 128.f) + eps);
#pragma unroll
      for (int j = 0; j < 8; ++j) n[j] = rbf(bf2f(ld.wv[j]) * rbf(x[j] * rstd));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) n[j] = x[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float pn =
          __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, n[j]), 0x128, 0xF, 0xF, true));
      const float cj = bf2f(ld.cs[p][j]), sj = bf2f(ld.sn[p][j]);
      const float r = rbf(rbf(n[j] * cj) + rbf((c < 8 ? -pn : pn) * sj));
      out[p][j] = f2bf(rope[p] ? r : x[j]);
    }
  }
}

template <int NP>
__device__ __forceinline__ void rope_items(const bf16_t *const (&src)[NP], const int (&pos)[NP], const bool (&rope)[NP],
                                           const bf16_t *nw, float eps, const bf16_t *cos_tab, const bf16_t *sin_tab,
                                           int max_pos, const bf16_t *safe, int l, bf16x8 (&out)[NP]) {
  RopeLoads<NP> ld;
  rope_issue<NP>(src, pos, nw, cos_tab, sin_tab, max_pos, safe, l, ld);
  rope_finish<NP>(ld, rope, nw != nullptr, eps, l, out);
}

__device__ __forceinline__ int k_swz(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
__device__ __forceinline__ int v_swz(int row, int ch) {
  return row * 256 + ((((ch >> 1) ^ (row & 7)) << 5) | ((ch & 1) << 4));
}

#ifdef DFL_ATTN_STAMPS  // diagnostic build only (scripts/dbg_attn_head_stamps.py): 100 MHz wall stamps of lane 0 of
// workgroup (kv head 0, query head 0) of the first old-key split [0] and of the new-row split [1]
__device__ unsigned long long g_hstamps[2][8];
#define HSTAMP(i)                                                                                  \
  do {                                                                                             \
    if (tid == 0 && kvh == 0 && hh == 0 && (split == 0 || is_new))                                 \
      g_hstamps[is_new ? 1 : 0][i] = __builtin_amdgcn_s_memrealtime();                             \
  } while (0)
// ... and of the first and the last o_proj workgroup of k_attn_oproj [0], [1] (scripts/dbg_attn_oproj_stamps.py)
__device__ unsigned long long g_ostamps[2][8];
// [split]: latest ticket-time of any attention workgroup of that key split; [15]: latest head-done signal
__device__ unsigned long long g_amax[16];
#define OSTAMP(i)                                                                                  \
  do {                                                                                             \
    if (tid == 0 && (t == 0 || t == p.ntiles - 1)) g_ostamps[t ? 1 : 0][i] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define HSTAMP(i)
#define OSTAMP(i)
#endif

template <int QT, int NW>
struct HeadLds {
  static_assert(NW * 4 >= QT * 16 && NW * 8192 >= 32768, "q rows: 4 per wave; new K / V: two 32-row tiles each");
  static constexpr int kVPriv = NW * 8192;           // wave-private V tiles (old splits) | new K, new V (new split)
  static constexpr int kQ = kVPriv;                  // q rows, swizzled like K: QT * 4 KB
  static constexpr int kLoop = kVPriv + QT * 4096;
  // after the loop: NW waves x QT x [16][kRow] fp32, rows padded 128 -> 132 floats: the 16 query rows of a wave's
  // ds_write_b128 (and of the merge's ds_read_b128) then fall on different banks (unpadded: 8-way conflicts,
  // 2.3 us of the stage)
  static constexpr int kRow = 132;
  static constexpr int kMergeO = 0;
  static constexpr int kMergeML = NW * QT * 16 * kRow * 4;  // NW waves x QT x [16][2] fp32
  static constexpr int kMerge = kMergeML + NW * QT * 128;
  static constexpr int kRaw = kLoop > kMerge ? kLoop : kMerge;
  // 8 waves: > 80 KB, one workgroup per CU, the geometry the sc1 hand-off is measured for
  static constexpr int kBytes = NW < 8 ? kRaw : (kRaw > 84 * 1024 ? kRaw : 84 * 1024);
};

// The stage as a device function of an NW-wave workgroup: k_attn_head runs it alone on 8 waves; k_attn_oproj runs it on 4,
// beside the o_proj workgroups that wait for it (SIGNAL: the workgroup that writes a head's final output stores it
// write-through and then counts the head on *done_ctr).  (kvh, by, cand) = blockIdx.x / .y / .z of k_attn_head:
// cand = candidate of a multi-candidate verify / request of a ragged batch (0 otherwise).
// HP (with QT = 2): the two 16-row "tiles" are not two query tiles of one head but the same <= 16 rows of TWO query heads
// of the kv group (head, head + 1): every K/V tile a wave fetches serves both.  Long prefixes (the host switches at
// ~5k keys): the G heads of a kv group otherwise pull the same K/V through their XCD's L2 G times (S = 8192: 134 MB of
// L2 reads per launch for 33.5 MB of cache, PMC: HBM fetch 34 MB), and with half the workgroups per split there are
// twice the splits.
template <int QT, int NW, bool SIGNAL, bool HP = false, bool XF32 = false>
__device__ __forceinline__ void attn_head_body(const HeadAttnArgs &a, char *lds, int *s_last_p, const int kvh, const int by,
                                               const int cand, int *done_ctr) {
  static_assert(!HP || QT == 2, "head pairs use the two-tile register layout");
  using L = HeadLds<QT, NW>;
  int &s_last = *s_last_p;

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const int hpg = HP ? a.G >> 1 : a.G;  // workgroups per kv head and split
  const int hh = by % hpg, split = by / hpg;
  const int head = kvh * a.G + (HP ? 2 * hh : hh);
  const int ns = a.ns_old + 1;
  const bool is_new = split == a.ns_old;
  int S = a.S, tau = a.tau, bs = a.bs, pos0 = a.pos0;
  if (a.dyn) {
    const int32_t *dp = a.dyn + cand * a.dyn_cand_stride;
    S = dp[DFL_DYN_S];
    tau = dp[DFL_DYN_TAU];
    bs = dp[DFL_DYN_BS];
    pos0 = dp[DFL_DYN_POS0];
  }
  const int n_new = tau + bs;  // <= 64
  const int qi = l & 15, g = l >> 4;
  HSTAMP(0);
  // per-candidate / per-request views as plain locals: the argument block stays untouched
  const bf16_t *const xq = XF32 ? a.cos_tab : a.xq + cand * a.xq_cand_stride;  // (XF32: only the dummy loads' safe address)
  const float *const xq32 = XF32 ? a.xq32 + cand * a.xq_cand_stride : nullptr;
  bf16x8 *const out_frag = a.out_frag + cand * a.out_cand_stride;
  float *const o_part = a.o_part + cand * a.ws_cand_stride;
  float *const ml_part = a.ml_part + cand * a.ws_cand_stride;
  int *const tickets = a.tickets + cand * a.ws_cand_stride;
  bf16_t *const kc = a.kc + cand * a.cache_cand_stride, *const vc = a.vc + cand * a.cache_cand_stride;
  bf16_t *const k_new = a.k_out ? a.k_out + cand * a.kv_out_cand_stride : kc;
  bf16_t *const v_new = a.k_out ? a.v_out + cand * a.kv_out_cand_stride : vc;
  const int new_rows_cap = a.k_out ? a.out_rows : a.cache_rows;
  const int new_row0 = a.k_out ? 0 : S;
  const bf16_t *kbase = kc + (int64_t)kvh * a.cache_rows * 128;
  const bf16_t *vbase = vc + (int64_t)kvh * a.cache_rows * 128;


  // ---- old-key tiles of this split; wave w walks t0 + w, t0 + w + NW, ...
  int t0 = 0, t1 = 0;
  if (!is_new && a.ns_old > 0) {
    const int nt = (S + 31) >> 5;
    const int tps = (nt + a.ns_old - 1) / a.ns_old;
    t0 = split * tps;
    t1 = t0 + tps;
    t0 = t0 > nt ? nt : t0;
    t1 = t1 > nt ? nt : t1;
  }
  const int old_last = S > 0 ? S - 1 : 0;
  bf16x8 kA[2][4], vA[8], kB[2][4], vB[8];
  // all loads of a tile are unconditional with clamped rows (they batch; rows >= S are masked
  // below and never touch memory another workgroup writes in this launch)
  // (lim = old_last: a real tile; lim = 0: a DUMMY fetch — every lane reads row 0, one or two cache lines per
  // instruction, already in L1/L2 — issued where the loops below have nothing left to prefetch, see there)
  auto fetch_k = [&](bf16x8(&kf)[2][4], int t, int lim) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int row = t * 32 + u * 16 + qi;
      row = row < lim ? row : lim;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        kf[u][s] = *reinterpret_cast<const bf16x8 *>(kbase + (int64_t)row * 128 + s * 32 + g * 8);
    }
  };
  auto fetch_v = [&](bf16x8(&vr)[8], int t, int lim) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = l + 64 * i;
      int row = t * 32 + (c >> 4);
      row = row < lim ? row : lim;
      vr[i] = *reinterpret_cast<const bf16x8 *>(vbase + (int64_t)row * 128 + (c & 15) * 8);
    }
  };
  auto fetch = [&](bf16x8(&kf)[2][4], bf16x8(&vr)[8], int t, int lim) {
    fetch_k(kf, t, lim);
    fetch_v(vr, t, lim);
  };
  // ---- q rows of this head: QT*16 items, 4 per wave (waves 0 .. 4*QT-1), written swizzled like a K tile.
  // Their loads go out first, the first K/V tile's behind them (vmcnt is in order), the arithmetic after both.
  char *q_lds = lds + L::kQ;
  const int jq = 4 * w + g;  // this 16-lane group's q item (QT = 1: waves 0..3 hold rows, the others a dummy)
  const int jrow = HP ? jq & 15 : jq;            // its block row ...
  const int jhead = HP ? head + (jq >> 4) : head;  // ... and query head
  RopeLoads<1> qld;
  {
    const int pos[1] = {pos0 + tau + jrow};
    if constexpr (XF32) {
      const float *src[1] = {jrow < bs ? xq32 + (int64_t)jrow * a.ldq + a.q_col + jhead * 128 : nullptr};
      rope_issue32<1>(src, pos, a.q_w, a.cos_tab, a.sin_tab, a.max_pos, xq32, xq, a.part_stride32, a.nparts32, l, qld);
    } else {
      const bf16_t *src[1] = {jrow < bs ? xq + (int64_t)jrow * a.ldq + a.q_col + jhead * 128 : nullptr};
      rope_issue<1>(src, pos, a.q_w, a.cos_tab, a.sin_tab, a.max_pos, xq, l, qld);
    }
  }
  // (compiler fences: without them hipcc hoists the K/V burst above the q loads and sinks the norm-weight load
  // into a branch of the arithmetic, and the q rows wait for the whole burst after all)
  asm volatile("" ::: "memory");
  auto finish_q = [&]() {
    if (jq < QT * 16) {
      const bool rp[1] = {true};
      bf16x8 ov[1];
      if constexpr (XF32) rope_sum32<1>(qld, a.nparts32);
      rope_finish<1>(qld, rp, a.q_w != nullptr, a.eps, l, ov);
      const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      *reinterpret_cast<bf16x8 *>(q_lds + (jq >> 4) * 4096 + k_swz(jq & 15, qi)) = jrow < bs ? ov[0] : z;
    }
  };
  int tcur = t0 + w;
  char *new_k = lds, *new_v = lds + 16384;  // up to two 32-row tiles each
  if (!is_new) {
    // the first tile's loads are issued by EVERY wave of an old split, with or without a tile (clamped rows of
    // valid cache memory): under a per-wave branch the compiler loses count of what is in flight and makes the q
    // arithmetic wait for the whole K/V burst instead of for its own four loads
    fetch(kA, vA, tcur < t1 ? tcur : 0, old_last);
    asm volatile("" ::: "memory");
    finish_q();
  } else {
    // ---- new split: this cycle's K / V rows of the kv head -> LDS tiles (and the cache, hh == 0).  Its loads go
    // out right behind the q loads (no K/V tile burst in this workgroup), then q, then the rows.
    auto sweep = [&](auto np_tag) {
      constexpr int NP = decltype(np_tag)::value;
      const bf16_t *src[NP];
      const float *src32[NP];
      int pos[NP], rel[NP];
      bool rp[NP], isv[NP];
      bf16x8 ov[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int it = (p * NW + w) * 4 + g;  // 4 NW items per pass
        isv[p] = it >= n_new;
        rel[p] = isv[p] ? it - n_new : it;
        const bool ok = it < 2 * n_new;
        // (arithmetic, not a select between two argument FIELDS: hipcc turns such a select into a select of their
        // addresses plus a per-lane global load with vmcnt(0) — four dependent round trips in this prologue)
        const int vsel = isv[p] ? 1 : 0;
        if constexpr (XF32) {  // (no context rows in this form: the host admits tau == 0 only)
          src32[p] = ok ? xq32 + (int64_t)rel[p] * a.ldq + a.k_col + vsel * (a.v_col - a.k_col) + kvh * 128 : nullptr;
          src[p] = ok ? xq : nullptr;   // presence flag for the stores below
        } else {
          const bf16_t *row = rel[p] < tau ? a.xc + (int64_t)rel[p] * a.ldc + a.ck_col + vsel * (a.cv_col - a.ck_col)
                                           : xq + (int64_t)(rel[p] - tau) * a.ldq + a.k_col + vsel * (a.v_col - a.k_col);
          src[p] = ok ? row + kvh * 128 : nullptr;
          src32[p] = nullptr;
        }
        pos[p] = pos0 + rel[p];
        rp[p] = !isv[p];
      }
      RopeLoads<NP> kld;
      if constexpr (XF32)
        rope_issue32<NP>(src32, pos, a.k_w, a.cos_tab, a.sin_tab, a.max_pos, xq32, xq, a.part_stride32, a.nparts32, l, kld);
      else
        rope_issue<NP>(src, pos, a.k_w, a.cos_tab, a.sin_tab, a.max_pos, xq, l, kld);
      asm volatile("" ::: "memory");
      finish_q();
      if constexpr (XF32) rope_sum32<NP>(kld, a.nparts32);
      rope_finish<NP>(kld, rp, a.k_w != nullptr, a.eps, l, ov);
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if (src[p]) {
          const int jt = rel[p] >> 5, r = rel[p] & 31;
          *reinterpret_cast<bf16x8 *>((isv[p] ? new_v + jt * 8192 + v_swz(r, qi) : new_k + jt * 8192 + k_swz(r, qi))) = ov[p];
          const int crow = new_row0 + rel[p];
          if (hh == 0 && crow < new_rows_cap)
            *reinterpret_cast<bf16x8 *>((isv[p] ? v_new : k_new) + ((int64_t)kvh * new_rows_cap + crow) * 128 + qi * 8) = ov[p];
        }
    };
    if (2 * n_new <= NW * 4)
      sweep(std::integral_constant<int, 1>{});
    else if (2 * n_new <= NW * 8)
      sweep(std::integral_constant<int, 2>{});
    else  // (the host function of a 4-wave launch admits tau + bs <= 32)
      sweep(std::integral_constant<int, 4>{});
  }
  HSTAMP(1);
  __syncthreads();

  // Q^T B-fragments: lane (q = l&15, g) holds Q[q][32 s + 8 g .. +8]
  bf16x8 qf[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qt][s] = *reinterpret_cast<const bf16x8 *>(q_lds + qt * 4096 + k_swz(qi, s * 4 + g));

  f32x4 o[QT][8];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -INFINITY;
    l_run[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // one 32-key tile: K fragments in registers, V tile (swizzled rows) at vt in LDS.
  // key0 = index of the tile's first key in its own numbering (old: cache row; new: rel);
  // nvalid = keys of that numbering that exist; vrows = rows of the V tile holding real data.
  auto compute = [&](const bf16x8(&kf)[2][4], const char *vt, int key0, int nvalid, int vrows, bool new_rows) {
    const int qq = qi >> 2, p4 = l & 3;
    int r0 = 4 * g + qq, r1 = 16 + 4 * g + qq;
    r0 = r0 < vrows ? r0 : vrows - 1;  // rows past the data hold stale LDS bytes: P is 0 there, but 0 x NaN is not
    r1 = r1 < vrows ? r1 : vrows - 1;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 sc[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        sc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) sc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u][s], qf[qt][s], sc[u], 0, 0, 0);
      }
      // lane (q, g): sc[u][r] is key key0 + u*16 + 4g + r
      float mx = -INFINITY;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + u * 16 + 4 * g + r;
          // causal (target verify): query row j sees new row rel <= tau + j; cached rows always
          const bool vis = key < nvalid && (!new_rows || !a.causal || key <= tau + (HP ? 0 : qt * 16) + qi);
          const float v = vis ? sc[u][r] * a.scale_log2 : -INFINITY;
          sc[u][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = wave_g_max(mx);
      const float m_new = fmaxf(m_run[qt], mx);
      const float m_ref = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_ref);
      m_run[qt] = m_new;
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
      l_run[qt] = l_run[qt] * alpha + psum;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) o[qt][dt] *= alpha;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const bf16x4 v0 =
            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(vt + r0 * 256 + (((dt ^ (r0 & 7)) << 5) | (p4 << 3))));
        const bf16x4 v1 =
            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(vt + r1 * 256 + (((dt ^ (r1 & 7)) << 5) | (p4 << 3))));
        const bf16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[qt][dt], 0, 0, 0);
      }
    }
  };

  if (is_new) {
    const int ntile = (n_new + 31) >> 5;  // 1 or 2
    if (w < ntile) {
      bf16x8 kf[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          kf[u][s] = *reinterpret_cast<const bf16x8 *>(new_k + w * 8192 + k_swz(u * 16 + qi, s * 4 + g));
      int vrows = n_new - w * 32;
      vrows = vrows > 32 ? 32 : vrows;
      compute(kf, new_v + w * 8192, w * 32, n_new, vrows, true);
    }
  } else {
    char *my_v = lds + w * 8192;
    auto put_v = [&](const bf16x8(&vr)[8]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = l + 64 * i;
        *reinterpret_cast<bf16x8 *>(my_v + v_swz(c >> 4, c & 15)) = vr[i];
      }
    };
    // the wave's own LDS tile: its ds_write -> ds_read order is program order (lgkmcnt), no barrier
    // A wave with ONE tile (S ~ 1k: every wave) just computes it.  A wave with several walks them with the next tile
    // in flight, and there every fetch is UNCONDITIONAL: behind a conditional fetch hipcc cannot count the loads in
    // flight and waits vmcnt(0) before the next ds_write / MFMA, i.e. for the tile it has just requested — one full
    // memory round trip per tile (2.9 us per tile at S = 8192, where a wave walks six).  Where nothing is left to
    // prefetch the fetch degenerates to a dummy (lim = 0).
    if (tcur + NW >= t1) {
      if (tcur < t1) {
        put_v(vA);
        compute(kA, my_v, tcur * 32, S, 32, false);
      }
    } else if (QT == 1) {
      for (; tcur < t1; tcur += 2 * NW) {
        fetch(kB, vB, tcur + NW, tcur + NW < t1 ? old_last : 0);
        put_v(vA);
        compute(kA, my_v, tcur * 32, S, 32, false);
        fetch(kA, vA, tcur + 2 * NW, tcur + 2 * NW < t1 ? old_last : 0);
        if (tcur + NW < t1) {
          put_v(vB);
          compute(kB, my_v, (tcur + NW) * 32, S, 32, false);
        }
      }
    } else {
      // two query tiles / a head pair: 64 accumulator + 32 q registers more, no room for a second K/V tile in flight
      // (a head pair with K double-buffered and its q fragments re-read from LDS sat at 254 VGPRs and ran SLOWER:
      // 4.98 against 4.68 ms per cycle at S = 8192).  (First form of the two-tile kernel: hipcc spilled
      // each V fragment right behind its load: eight serial round trips per tile.)  K of the next tile is requested
      // once the current one's QK^T is done with it, V once it sits in LDS
      for (; tcur < t1; tcur += NW) {
        const int lim = tcur + NW < t1 ? old_last : 0;
        put_v(vA);
        fetch_v(vA, tcur + NW, lim);
        compute(kA, my_v, tcur * 32, S, 32, false);
        fetch_k(kA, tcur + NW, lim);
      }
    }
  }

  // ---- the NW waves meet in LDS
  HSTAMP(2);
  __syncthreads();
  HSTAMP(3);
  {
    float *mo = reinterpret_cast<float *>(lds + L::kMergeO);
    float *mml = reinterpret_cast<float *>(lds + L::kMergeML);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float lt = wave_g_sum(l_run[qt]);
      float *op = mo + ((w * QT + qt) * 16 + qi) * L::kRow;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4 *>(op + dt * 16 + 4 * g) = o[qt][dt];
      if (g == 0) {
        mml[((w * QT + qt) * 16 + qi) * 2] = m_run[qt];
        mml[((w * QT + qt) * 16 + qi) * 2 + 1] = lt;
      }
    }
  }
  __syncthreads();
  const bool has_item = tid < QT * 256;  // QT = 1 on 8 waves: the upper 4 have no item (they still join the barriers)
  const int qt = tid >> 8, q = (tid >> 4) & 15, dg = tid & 15;
  float M = -INFINITY, Lsum = 0.f, acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (has_item) {
    const float *mo = reinterpret_cast<const float *>(lds + L::kMergeO);
    const float *mml = reinterpret_cast<const float *>(lds + L::kMergeML);
    float ms[NW], ls[NW];
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      ms[ww] = mml[((ww * QT + qt) * 16 + q) * 2];
      ls[ww] = mml[((ww * QT + qt) * 16 + q) * 2 + 1];
      M = fmaxf(M, ms[ww]);
    }
    const float mref = M == -INFINITY ? 0.f : M;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {  // fixed wave order: reproducible sums
      const float wgt = __builtin_amdgcn_exp2f(ms[ww] - mref);  // exp2(-inf) = 0: a wave without tiles
      const float *op = mo + ((ww * QT + qt) * 16 + q) * L::kRow + dg * 8;
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(op), a1 = *reinterpret_cast<const f32x4 *>(op + 4);
      Lsum += wgt * ls[ww];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] += wgt * a0[j];
        acc[4 + j] += wgt * a1[j];
      }
    }
  }
  auto emit = [&](float Lt, const float(&v)[8]) {  // frag16 chunk (n>>3 = head*16 + dg, row q) of query tile qt
    const float inv = Lt > 0.f ? 1.f / Lt : 0.f;
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f2bf(v[j] * inv);
    if (SIGNAL && q >= bs) r = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};  // the GEMM behind it takes rows >= bs as zero
    bf16x8 *dst = HP ? &out_frag[((head + qt) * 16 + dg) * 16 + q] : &out_frag[qt * a.out_tile_stride + (head * 16 + dg) * 16 + q];
    if (SIGNAL) {  // read by other workgroups of THIS launch: write-through
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 16, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r), rd, 0, 0, 16);
    } else {
      *dst = r;
    }
  };
  // SIGNAL: every storing wave drains, the workgroup meets, one lane counts the head as done (hand-off row 1 of the
  // table in MI355X_MICROARCH.md: sc1 stores, drain, barrier, agent-scope add; consumers poll, barrier, sc1 loads)
  auto signal = [&]() {
    if (SIGNAL) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // one wave instruction, 32 lanes: +1 on each of the 32 replicas of the counter (a line of its own each), so that a
      // replica has 8 pollers, not 256
      if (tid < 32) __hip_atomic_fetch_add(done_ctr + tid * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef DFL_ATTN_STAMPS
      if (tid == 0) atomicMax(&g_amax[15], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    }
  };
  HSTAMP(4);
  if (ns == 1) {
    if (has_item) emit(Lsum, acc);
    signal();
    return;
  }

  // ---- publish the workgroup's partial write-through, drain, ticket; the head's last arriver merges
  const size_t item_row = HP ? (size_t)(head + qt) * 16 + q : (size_t)head * (QT * 16) + qt * 16 + q;
  const size_t rows_per_split = (size_t)a.n_q * (HP ? 16 : QT * 16);
  const __amdgpu_buffer_rsrc_t ro =
      __builtin_amdgcn_make_buffer_rsrc(o_part, 0, (int)(rows_per_split * ns * 128 * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rm =
      __builtin_amdgcn_make_buffer_rsrc(ml_part, 0, (int)(rows_per_split * ns * 2 * sizeof(float)), 0x00020000);
  if (has_item) {
    const int off = (int)((((size_t)split * rows_per_split + item_row) * 128 + dg * 8) * sizeof(float));
    const f32x4 s0 = {acc[0], acc[1], acc[2], acc[3]}, s1 = {acc[4], acc[5], acc[6], acc[7]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s0), ro, off, 0, 16);  // sc1
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s1), ro, off + 16, 0, 16);
    if (dg == 0) {
      const int offm = (int)((((size_t)split * rows_per_split + item_row) * 2) * sizeof(float));
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, M), rm, offm, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, Lsum), rm, offm + 4, 0, 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  HSTAMP(5);
  if (tid == 0) {
    const int ticket = __hip_atomic_fetch_add(&tickets[head], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == ns - 1;
    if (last) __hip_atomic_store(&tickets[head], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
    s_last = last;
  }
  __syncthreads();
  HSTAMP(6);
#ifdef DFL_ATTN_STAMPS
  if (tid == 0) atomicMax(&g_amax[split < 15 ? split : 14], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
  if (!s_last) return;

  // every load of the handed-off bytes is an sc1 load (L2-served): no acquire needed
  if (has_item) {
    float Mg = -INFINITY, Lg = 0.f, ag[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ag[j] = 0.f;
    constexpr int U = 8;  // at 1k keys ns = 5 or 6: one round trip
    for (int s = 0; s < ns; s += U) {
      float ms[U], ls[U];
      f32x4 a0[U], a1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int sc = s + u < ns ? s + u : ns - 1;
        const int off = (int)((((size_t)sc * rows_per_split + item_row) * 128 + dg * 8) * sizeof(float));
        const int offm = (int)((((size_t)sc * rows_per_split + item_row) * 2) * sizeof(float));
        ms[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, offm, 0, 16));
        ls[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, offm + 4, 0, 16));
        a0[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ro, off, 0, 16));
        a1[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ro, off + 16, 0, 16));
      }
      float mnew = Mg;
#pragma unroll
      for (int u = 0; u < U; ++u) mnew = fmaxf(mnew, s + u < ns ? ms[u] : -INFINITY);
      const float mref = mnew == -INFINITY ? 0.f : mnew;
      const float scale = __builtin_amdgcn_exp2f(Mg - mref);
      Mg = mnew;
      Lg *= scale;
#pragma unroll
      for (int j = 0; j < 8; ++j) ag[j] *= scale;
#pragma unroll
      for (int u = 0; u < U; ++u) {  // fixed split order
        const float wgt = s + u < ns ? __builtin_amdgcn_exp2f(ms[u] - mref) : 0.f;
        Lg += wgt * ls[u];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ag[j] += wgt * a0[u][j];
          ag[4 + j] += wgt * a1[u][j];
        }
      }
    }
    emit(Lg, ag);
  }
  signal();
#ifdef DFL_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  HSTAMP(7);
}

template <int QT>
__global__ __launch_bounds__(512) void k_attn_head(HeadAttnArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[HeadLds<QT, 8>::kBytes];
  __shared__ int s_last;
  attn_head_body<QT, 8, false>(a, lds, &s_last, blockIdx.x, blockIdx.y, blockIdx.z, nullptr);
}

__global__ __launch_bounds__(512) void k_attn_head_pair(HeadAttnArgs a) {  // two query heads per workgroup
  __shared__ __attribute__((aligned(16))) char lds[HeadLds<2, 8>::kBytes];
  __shared__ int s_last;
  attn_head_body<2, 8, false, true>(a, lds, &s_last, blockIdx.x, blockIdx.y, blockIdx.z, nullptr);
}

// the same two kernels on fp32 K-part sums of the qkv GEMM (HeadAttnArgs::xq32)
__global__ __launch_bounds__(512) void k_attn_head32(HeadAttnArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[HeadLds<1, 8>::kBytes];
  __shared__ int s_last;
  attn_head_body<1, 8, false, false, true>(a, lds, &s_last, blockIdx.x, blockIdx.y, blockIdx.z, nullptr);
}
__global__ __launch_bounds__(512) void k_attn_head_pair32(HeadAttnArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[HeadLds<2, 8>::kBytes];
  __shared__ int s_last;
  attn_head_body<2, 8, false, true, true>(a, lds, &s_last, blockIdx.x, blockIdx.y, blockIdx.z, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// Attention stage + o_proj in ONE launch (round 2; VERDICT r1 item 4: a stage pair kept inside a launch where the
// hand-off buys more than the boundary it replaces).  The attention stage leaves HBM idle for ~10 us per layer, and
// o_proj's weights (Qwen3-8B: 33.5 MB = 131 KB per CU) fit the chip's register files.  So the launch carries, behind
// the attention workgroups, one workgroup per 16-column output tile of o_proj that requests its whole weight slice
// (32 k-steps per wave = 128 VGPRs) the moment it starts — the weights stream from HBM WHILE the attention runs —
// then waits for the heads' final outputs, loads them (sc1), and is 32 MFMAs per wave, a 4-wave LDS reduction and the
// residual epilogue away from done.
// MEASURED (profiles/r2_attn_oproj_stamps.txt): 23.2 us per launch against 10.5 + 9.8 us for the two
// launches it replaces, so the callers keep two launches (fuse_oproj = False) and this stays an opt-in, tested variant.
// The weights do land early (7 us) and every workgroup is resident, but each cross-XCD hand-off costs ~2 us at either
// end (write-through drain, then loads served from beyond L2): ticket 11.5 -> heads merged 16.5 -> seen 17.2 ->
// 128 KB of activations per o_proj workgroup loaded 21.5 -> done 22.5, and the weight stream slows the attention
// itself by 3 us.  The boundary it removes is priced at 1.8 us.
//   * every workgroup has 4 waves with the 256-VGPR budget (__launch_bounds__(256, 2)): a CU holds two, whichever
//     kind; with <= 256 attention workgroups + H/16 = 256 o_proj workgroups the whole grid is resident at once;
//   * no attention workgroup ever waits; an o_proj workgroup waits only for attention workgroups, which have LOWER
//     linear ids (dispatched first), so progress never depends on co-residency; the wait is BOUNDED all the same
//     (2 ms, then *fail = 1 and the workgroup leaves: wrong numbers and a raised error, never a hang);
//   * hand-off: the workgroup that merges a head stores its frag16 output write-through, drains, meets, and adds 1 to
//     each of 32 replicas of `done`; one lane per o_proj workgroup polls ITS replica (relaxed agent-scope load +
//     s_sleep), the workgroup barrier spreads the news, an agent-scope acquire fence precedes the activation loads;
//   * the last o_proj workgroup past its wait re-arms both counters: the launch is capturable and re-launchable as is.
struct AttnOArgs {
  HeadAttnArgs at;
  int n_attn;         // attention workgroups = n_kv * G * (ns_old + 1); workgroup b < n_attn: kv head b % n_kv, y = b / n_kv
  const bf16x8 *wo;   // packed o_proj weight [ntiles][KS][64]
  int KS, ntiles;     // q_dim / 32 (<= 128), H / 16
  bf16_t *h_io;       // residual stream [16][ldh]: h <- bf16(h + bf16(attn . Wo^T))
  int64_t ldh;
  float *ss_out;      // [ntiles][16] partial sums of squares of the new rows (next GEMM's RMSNorm)
  int *done;          // 32 replicas of the heads-done counter, 32 ints (one 128-B line) apart
  int *o_done, *fail;
  int done_target;    // = n_q: heads whose output must be complete
};

__global__ __launch_bounds__(256, 2) void k_attn_oproj(AttnOArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[HeadLds<1, 4>::kBytes];
  __shared__ int s_last;
  const int b = blockIdx.x;
  if (b < p.n_attn) {
    attn_head_body<1, 4, true>(p.at, lds, &s_last, b % p.at.n_kv, b / p.at.n_kv, 0, p.done);
    return;
  }
  const int t = b - p.n_attn;  // output column tile
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  OSTAMP(0);
  // ---- the whole weight slice of the tile, now: wave w holds k-steps [32 w, 32 w + 32)
  bf16x8 wv[4][8];
  {
    const bf16x8 *base = p.wo + ((size_t)t * p.KS + 32 * w) * 64;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int nf = p.KS - 32 * w - 8 * c;
      nf = nf < 0 ? 0 : (nf > 8 ? 8 : nf);
      load_ksteps<8>(wv[c], base + c * 8 * 64, nf, l);
    }
  }
#ifdef DFL_ATTN_STAMPS
  OSTAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  OSTAMP(2);  // weights landed
#endif
  // ---- wait (bounded) until every head's output is complete
  if (tid == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int *mine = p.done + (t & 31) * 32;  // 8 pollers per replica
    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.done_target) {
      __builtin_amdgcn_s_sleep(8);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) {  // 2 ms of the 100 MHz clock
        *p.fail = 1;
        break;
      }
    }
  }
  OSTAMP(3);  // heads done (lane 0's poll)
  __syncthreads();
  // ---- the activation fragments (frag16 of the 16 rows: k-step ks = 1 KiB at out_frag + 64 ks), sc1 loads: 128 KB per
  // workgroup, 4.3 us; an agent-scope acquire fence (L2 invalidate) + plain loads instead took 11 us
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  {
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(p.at.out_frag), 0, p.KS * 1024, 0x00020000);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x8 xv[8];
#pragma unroll
      for (int f = 0; f < 8; ++f)  // (k-steps past KS: out of the descriptor's range, read as zeros)
        xv[f] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, ((32 * w + 8 * c + f) * 64 + l) * 16, 0, 16));
#pragma unroll
      for (int f = 0; f < 8; ++f) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[c][f], xv[f], acc, 0, 0, 0);
    }
  }
  float *red = reinterpret_cast<float *>(lds);  // [4 waves][256]
  *reinterpret_cast<f32x4 *>(&red[w * 256 + l * 4]) = acc;
  OSTAMP(4);  // activations loaded, MFMAs done (wave 0)
  __syncthreads();
  OSTAMP(5);
  {  // thread (row m, column nl) of the tile; D layout of the MFMA as in gemm_skinny.hip
    const int m = tid >> 4, nl = tid & 15;
    const int idx = 4 * (m + 16 * (nl >> 2)) + (nl & 3);
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) s += red[ww * 256 + idx];
    const int n = t * 16 + nl;
    const float v = rbf(s);  // the Linear's bf16 output (model/dflash.py:101)
    bf16_t *hp = p.h_io + (int64_t)m * p.ldh + n;
    const float hn = rbf(bf2f(*hp) + v);  // residual add (:140)
    *hp = f2bf(hn);
    const float qs = row_sum16(hn * hn);
    if (nl == 0 && p.ss_out) p.ss_out[t * 16 + m] = qs;
  }
#ifdef DFL_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  OSTAMP(6);
  // ---- the last o_proj workgroup past its wait re-arms the counters for the next launch
  if (tid == 0) {
    const int k = __hip_atomic_fetch_add(p.o_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == p.ntiles - 1) {
      for (int r = 0; r < 32; ++r) __hip_atomic_store(p.done + r * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.o_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

#ifdef DFL_ATTN_STAMPS
extern "C" int dfl_debug_read_head_stamps(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_hstamps), sizeof(unsigned long long) * 16);
}
extern "C" int dfl_debug_read_attn_max(unsigned long long *host_out) {  // reads and clears
  const int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_amax), sizeof(unsigned long long) * 16);
  unsigned long long z[16] = {};
  return rc ? rc : (int)hipMemcpyToSymbol(HIP_SYMBOL(g_amax), z, sizeof(z));
}
extern "C" int dfl_debug_read_oproj_stamps(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ostamps), sizeof(unsigned long long) * 16);
}
#endif

extern "C" int64_t dfl_attn_head_ws_bytes(int n_q, int max_splits, int q_tiles) {
  const int64_t b = (int64_t)max_splits * n_q * q_tiles * 16 * (128 + 2) * sizeof(float) + (int64_t)n_q * sizeof(int) + 64;
  return (b + 15) / 16 * 16;   // per-candidate blocks of a multi-candidate launch stay 16-byte aligned
}

namespace {
bool pair_ok_forced(int knob, bool has_tail, int q_tiles, int G, int bs) {
  return knob == 1 && !has_tail && q_tiles == 1 && G % 2 == 0 && bs <= 16;
}

// o_proj behind the attention stage in the same launch (dfl_attn_head_oproj)
struct OprojTail {
  const void *wo;
  int q_dim, H;
  void *h_io;
  int64_t ldh;
  float *ss_out;
  int32_t *sync;
};

int attn_head_launch(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, const void *xc, int64_t ldc, int ck_col,
                     int cv_col, int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w, float eps,
                     const void *cos_tab, const void *sin_tab, int max_pos, void *kcache, void *vcache, int cache_rows,
                     float scale, int causal, const int32_t *dyn, int S, int tau, int bs, int pos0, int q_tiles, void *ws,
                     int max_splits, void *out_frag, int64_t out_tile_stride, int n_cand, int64_t xq_cand_stride,
                     int64_t out_cand_stride, void *k_out, void *v_out, int64_t kv_out_cand_stride, int out_rows,
                     int dyn_cand_stride, int64_t cache_cand_stride, void *stream, const OprojTail *tail = nullptr,
                     int nparts32 = 0, int64_t part_stride32 = 0) {
  // nparts32 > 0: xq points at fp32 K-part sums (HeadAttnArgs::xq32; strides in floats)
  DFL_REQUIRE(xq && cos_tab && sin_tab && kcache && vcache && out_frag && ws, "dfl_attn_head: null pointer");
  DFL_REQUIRE(nparts32 >= 0 && nparts32 <= 2 && (nparts32 == 0 || (!tail && q_tiles == 1 && tau == 0 && !xc && part_stride32 % 4 == 0)),
              "dfl_attn_head: the fp32-partials form takes 1 or 2 K parts, one query tile, no context rows");
  DFL_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), "dfl_attn_head: give both norm weights or neither");
  DFL_REQUIRE(n_q > 0 && n_kv > 0 && n_q % n_kv == 0, "dfl_attn_head: bad head counts (n_q=%d n_kv=%d)", n_q, n_kv);
  DFL_REQUIRE(q_tiles == 1 || q_tiles == 2, "dfl_attn_head: q_tiles must be 1 or 2");
  DFL_REQUIRE(ldq > 0 && ldq % 8 == 0 && q_col >= 0 && k_col >= 0 && v_col >= 0 && q_col % 8 == 0 && k_col % 8 == 0 &&
                  v_col % 8 == 0 && max_pos > 0,
              "dfl_attn_head: bad block-row layout");
  DFL_REQUIRE(S >= 0 && tau >= 0 && tau <= 32 && bs >= 1 && bs <= 16 * q_tiles && tau + bs <= 64,
              "dfl_attn_head: lengths S=%d tau=%d bs=%d outside the kernel's range (q_tiles=%d)", S, tau, bs, q_tiles);
  DFL_REQUIRE(tau == 0 || (xc && ldc > 0 && ldc % 8 == 0 && ck_col >= 0 && cv_col >= 0 && ck_col % 8 == 0 && cv_col % 8 == 0),
              "dfl_attn_head: context rows without a context source");
  DFL_REQUIRE(S + (k_out ? 0 : tau + bs) <= cache_rows, "dfl_attn_head: S + tau + bs = %d exceeds cache_rows = %d", S + tau + bs,
              cache_rows);
  DFL_REQUIRE(max_splits >= 1 && out_tile_stride >= 0 && out_tile_stride % 8 == 0, "dfl_attn_head: bad max_splits / out_tile_stride");
  DFL_REQUIRE(n_cand >= 1 && n_cand <= 64, "dfl_attn_head: n_cand outside 1..64");
  DFL_REQUIRE(n_cand == 1 || (xq_cand_stride % 8 == 0 && out_cand_stride % 8 == 0 && cache_cand_stride % 8 == 0 &&
                              ((k_out && v_out) || cache_cand_stride >= (int64_t)n_kv * cache_rows * 128)),
              "dfl_attn_head: several blocks per launch need 8-element strides and either a K/V staging area "
              "(candidates on one cache) or a cache per request");
  DFL_REQUIRE(!k_out == !v_out && (!k_out || (out_rows >= tau + bs && kv_out_cand_stride >= (int64_t)n_kv * out_rows * 128)),
              "dfl_attn_head: bad K/V staging area");
  // Old-key splits: a workgroup's 8 waves take one 32-key tile each per round, so up to 8 tiles
  // per split cost one round; beyond ~224 workgroups per launch the splits grow instead
  // (S here is the bound the caller sized the launch for when the lengths come from dyn).
  const int G = n_q / n_kv;
  const int nt = (S + 31) / 32;
  // (DFL_ATTN_HEAD_TILES / DFL_ATTN_HEAD_WGS: tuning knobs read once from the environment, for
  // scripts/dbg_attn_head_stamps.py; the defaults are the measured choice)
  static const int knob_tiles = [] { const char *e = getenv("DFL_ATTN_HEAD_TILES"); return e ? atoi(e) : 8; }();
  static const int knob_wgs = [] { const char *e = getenv("DFL_ATTN_HEAD_WGS"); return e ? atoi(e) : 224; }();
  // (several blocks per launch — requests of a ragged batch, candidates: ONE round of workgroups, 256; with 448 the
  // second half-round of a 4-request launch was a tail: 6.10 -> 5.99 ms per 4-request cycle)
  static const int knob_wgsm = [] { const char *e = getenv("DFL_ATTN_HEAD_WGS_MULTI"); return e ? atoi(e) : 256; }();
  static const int knob_wgs4 = [] { const char *e = getenv("DFL_ATTN_OPROJ_WGS"); return e ? atoi(e) : 256; }();
  static const int knob_tiles4 = [] { const char *e = getenv("DFL_ATTN_OPROJ_TILES"); return e ? atoi(e) : 4; }();
  // (with the o_proj tail: 4-wave workgroups, two per CU; 256 attention + H/16 o_proj workgroups fill the 512 slots)
  const int tiles = tail ? (knob_tiles4 < 1 ? 1 : knob_tiles4) : (knob_tiles < 1 ? 1 : knob_tiles);
  int ns_old = (nt + tiles - 1) / tiles;
  int budget = (tail ? knob_wgs4 : (n_cand > 1 ? knob_wgsm : knob_wgs)) / (n_q * n_cand) - 1;
  budget = budget < 1 ? 1 : budget;
  ns_old = ns_old > budget ? budget : ns_old;
  // Head pairs (k_attn_head_pair): when the splits the budget allows would leave a wave more than one tile, two heads
  // per workgroup halve the K/V pulled through L2 and double the splits.  DFL_ATTN_HEAD_PAIR=0/1 forces it off / on.
  static const int knob_pair = [] { const char *e = getenv("DFL_ATTN_HEAD_PAIR"); return e ? atoi(e) : -1; }();
  // Measured on the 8B shapes (cycle, ms): S = 2048 4.28 -> 4.40 (worse), 4096 4.48 -> 4.49, 8192 4.78 -> 4.68: on from ~5k keys.
  // Several blocks per launch (requests of a ragged batch, candidates): the one round of workgroups leaves a single
  // old-key split per head from two requests on, i.e. ~4 tiles per wave at S = 1k — pairs on whenever the budget
  // leaves a wave more than one tile (round 3, same box: 4 requests 6.03 -> 5.81 ms per cycle, 3 requests 5.97 -> 5.69;
  // 2 requests 5.22 -> 5.29: from three blocks on).
  bool pair = !tail && q_tiles == 1 && G % 2 == 0 && bs <= 16 && nt > tiles * ns_old && (nt > 160 || n_cand > 2);
  if (knob_pair >= 0) pair = pair_ok_forced(knob_pair, tail != nullptr, q_tiles, G, bs);
  if (pair) {
    int budget2 = (n_cand > 1 ? knob_wgsm : knob_wgs) / ((n_q / 2) * n_cand) - 1;
    budget2 = budget2 < 1 ? 1 : budget2;
    ns_old = (nt + tiles - 1) / tiles;
    ns_old = ns_old > budget2 ? budget2 : ns_old;
  }
  ns_old = ns_old > max_splits - 1 ? max_splits - 1 : ns_old;
  if (nt == 0 && !dyn) ns_old = 0;
  HeadAttnArgs a{};
  a.xq = (const bf16_t *)xq;
  a.ldq = ldq;
  a.q_col = q_col;
  a.k_col = k_col;
  a.v_col = v_col;
  a.xc = (const bf16_t *)xc;
  a.ldc = ldc;
  a.ck_col = ck_col;
  a.cv_col = cv_col;
  a.q_w = (const bf16_t *)q_norm_w;
  a.k_w = (const bf16_t *)k_norm_w;
  a.eps = eps;
  a.cos_tab = (const bf16_t *)cos_tab;
  a.sin_tab = (const bf16_t *)sin_tab;
  a.max_pos = max_pos;
  a.kc = (bf16_t *)kcache;
  a.vc = (bf16_t *)vcache;
  a.cache_rows = cache_rows;
  a.n_q = n_q;
  a.n_kv = n_kv;
  a.G = G;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.causal = causal ? 1 : 0;
  a.dyn = dyn;
  a.S = S;
  a.tau = tau;
  a.bs = bs;
  a.pos0 = pos0;
  a.out_frag = (bf16x8 *)out_frag;
  a.out_tile_stride = out_tile_stride / 8;
  const int64_t rows = (int64_t)max_splits * n_q * q_tiles * 16;
  a.o_part = (float *)ws;
  a.ml_part = (float *)ws + rows * 128;
  a.tickets = (int *)((float *)ws + rows * 130);
  a.ns_old = ns_old;
  a.xq_cand_stride = xq_cand_stride;
  a.out_cand_stride = out_cand_stride / 8;
  a.ws_cand_stride = dfl_attn_head_ws_bytes(n_q, max_splits, q_tiles) / 4;
  a.k_out = (bf16_t *)k_out;
  a.v_out = (bf16_t *)v_out;
  a.kv_out_cand_stride = kv_out_cand_stride;
  a.out_rows = out_rows;
  a.dyn_cand_stride = dyn_cand_stride;
  a.cache_cand_stride = cache_cand_stride;
  const dim3 grid(n_kv, (pair ? G / 2 : G) * (ns_old + 1), n_cand);
  hipStream_t st = (hipStream_t)stream;
  if (nparts32 > 0) {
    a.xq = nullptr;
    a.xq32 = (const float *)xq;
    a.nparts32 = nparts32;
    a.part_stride32 = part_stride32;
    if (pair)
      hipLaunchKernelGGL(k_attn_head_pair32, grid, dim3(512), 0, st, a);
    else
      hipLaunchKernelGGL(k_attn_head32, grid, dim3(512), 0, st, a);
    DFL_CHECK_LAUNCH("dfl_attn_head_batch_f32");
    return DFL_OK;
  }
  if (pair) {
    hipLaunchKernelGGL(k_attn_head_pair, grid, dim3(512), 0, st, a);
    DFL_CHECK_LAUNCH("dfl_attn_head");
    return DFL_OK;
  }
  if (tail) {
    AttnOArgs p{};
    p.at = a;
    p.n_attn = n_kv * G * (ns_old + 1);
    p.wo = (const bf16x8 *)tail->wo;
    p.KS = tail->q_dim / 32;
    p.ntiles = tail->H / 16;
    p.h_io = (bf16_t *)tail->h_io;
    p.ldh = tail->ldh;
    p.ss_out = tail->ss_out;
    p.done = tail->sync;
    p.o_done = tail->sync + 1024;
    p.fail = tail->sync + 1025;
    p.done_target = n_q;
    hipLaunchKernelGGL(k_attn_oproj, dim3(p.n_attn + p.ntiles), dim3(256), 0, st, p);
    DFL_CHECK_LAUNCH("dfl_attn_head_oproj");
    return DFL_OK;
  }
  if (q_tiles == 1)
    hipLaunchKernelGGL(k_attn_head<1>, grid, dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL(k_attn_head<2>, grid, dim3(512), 0, st, a);
  DFL_CHECK_LAUNCH("dfl_attn_head");
  return DFL_OK;
}
}  // namespace

extern "C" int dfl_attn_head(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, const void *xc, int64_t ldc,
                             int ck_col, int cv_col, int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w,
                             float eps, const void *cos_tab, const void *sin_tab, int max_pos, void *kcache,
                             void *vcache, int cache_rows, float scale, int causal, const int32_t *dyn, int S, int tau,
                             int bs, int pos0, int q_tiles, void *ws, int max_splits, void *out_frag,
                             int64_t out_tile_stride, void *stream) {
  return attn_head_launch(xq, ldq, q_col, k_col, v_col, xc, ldc, ck_col, cv_col, n_q, n_kv, q_norm_w, k_norm_w, eps, cos_tab,
                          sin_tab, max_pos, kcache, vcache, cache_rows, scale, causal, dyn, S, tau, bs, pos0, q_tiles, ws,
                          max_splits, out_frag, out_tile_stride, 1, 0, 0, nullptr, nullptr, 0, 0, 0, 0, stream);
}

extern "C" int dfl_attn_head_oproj(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, const void *xc, int64_t ldc,
                                   int ck_col, int cv_col, int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w,
                                   float eps, const void *cos_tab, const void *sin_tab, int max_pos, void *kcache,
                                   void *vcache, int cache_rows, float scale, int causal, const int32_t *dyn, int S, int tau,
                                   int bs, int pos0, void *ws, int max_splits, void *attn_frag, const void *wo_packed, int H,
                                   void *h_io, int64_t ldh, float *ss_out, int32_t *sync, void *stream) {
  DFL_REQUIRE(wo_packed && h_io && sync, "dfl_attn_head_oproj: null pointer");
  DFL_REQUIRE(H > 0 && H % 16 == 0 && ldh >= H && n_q * 128 <= 4096 && bs <= 16 && tau + bs <= 32,
              "dfl_attn_head_oproj: H=%d q_dim=%d tau=%d bs=%d outside the kernel's range (q_dim <= 4096, tau + bs <= 32)", H,
              n_q * 128, tau, bs);
  const OprojTail tail{wo_packed, n_q * 128, H, h_io, ldh, ss_out, sync};
  return attn_head_launch(xq, ldq, q_col, k_col, v_col, xc, ldc, ck_col, cv_col, n_q, n_kv, q_norm_w, k_norm_w, eps, cos_tab,
                          sin_tab, max_pos, kcache, vcache, cache_rows, scale, causal, dyn, S, tau, bs, pos0, 1, ws, max_splits,
                          attn_frag, 0, 1, 0, 0, nullptr, nullptr, 0, 0, 0, 0, stream, &tail);
}

extern "C" int dfl_attn_head_cand(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, int n_cand,
                                  int64_t xq_cand_stride, int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w,
                                  float eps, const void *cos_tab, const void *sin_tab, int max_pos, const void *kcache,
                                  const void *vcache, int cache_rows, float scale, int S, int bs, void *ws, int max_splits,
                                  void *out_frag, int64_t out_cand_stride, void *k_out, void *v_out,
                                  int64_t kv_out_cand_stride, int out_rows, void *stream) {
  DFL_REQUIRE(n_cand >= 1 && k_out && v_out, "dfl_attn_head_cand: needs the K/V staging area");
  return attn_head_launch(xq, ldq, q_col, k_col, v_col, nullptr, 0, 0, 0, n_q, n_kv, q_norm_w, k_norm_w, eps, cos_tab, sin_tab,
                          max_pos, const_cast<void *>(kcache), const_cast<void *>(vcache), cache_rows, scale, 1, nullptr, S, 0,
                          bs, S, 1, ws, max_splits, out_frag, 0, n_cand, xq_cand_stride, out_cand_stride, k_out, v_out,
                          kv_out_cand_stride, out_rows, 0, 0, stream);
}

extern "C" int dfl_attn_head_batch(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, int R, int64_t xq_req_stride,
                                   int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w, float eps,
                                   const void *cos_tab, const void *sin_tab, int max_pos, void *kcache, void *vcache,
                                   int cache_rows, int64_t cache_req_stride, float scale, int causal, const int32_t *dyn,
                                   int kv_len_max, void *ws, int max_splits, void *out_frag, int64_t out_req_stride,
                                   void *stream) {
  DFL_REQUIRE(dyn, "dfl_attn_head_batch: the requests' lengths come from dyn (R records)");
  DFL_REQUIRE(kv_len_max >= 16 && kv_len_max <= cache_rows, "dfl_attn_head_batch: kv_len_max=%d outside 16..cache_rows", kv_len_max);
  // lengths from the device records (block form: tau = 0, bs <= 16); kv_len_max - 16 bounds every request's S
  return attn_head_launch(xq, ldq, q_col, k_col, v_col, nullptr, 0, 0, 0, n_q, n_kv, q_norm_w, k_norm_w, eps, cos_tab, sin_tab,
                          max_pos, kcache, vcache, cache_rows, scale, causal, dyn, kv_len_max - 16, 0, 16, 0, 1, ws, max_splits,
                          out_frag, 0, R, xq_req_stride, out_req_stride, nullptr, nullptr, 0, 0, DFL_DYN_WORDS, cache_req_stride,
                          stream);
}

// Blocks of 17..32 rows in the ragged batch: request r = TWO consecutive 16-row tiles of xq / out_frag (strides per
// request), one cache, one length record (bs = 17..32 counts both tiles).
extern "C" int dfl_attn_head_batch_t(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, int R, int64_t xq_req_stride,
                                     int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w, float eps,
                                     const void *cos_tab, const void *sin_tab, int max_pos, void *kcache, void *vcache,
                                     int cache_rows, int64_t cache_req_stride, float scale, int causal, const int32_t *dyn,
                                     int kv_len_max, void *ws, int max_splits, void *out_frag, int64_t out_req_stride,
                                     int64_t out_tile_stride, int q_tiles, void *stream) {
  DFL_REQUIRE(dyn, "dfl_attn_head_batch_t: the requests' lengths come from dyn (R records)");
  DFL_REQUIRE(q_tiles == 1 || q_tiles == 2, "dfl_attn_head_batch_t: q_tiles must be 1 or 2");
  DFL_REQUIRE(kv_len_max >= 16 * q_tiles && kv_len_max <= cache_rows, "dfl_attn_head_batch_t: kv_len_max=%d outside range", kv_len_max);
  return attn_head_launch(xq, ldq, q_col, k_col, v_col, nullptr, 0, 0, 0, n_q, n_kv, q_norm_w, k_norm_w, eps, cos_tab, sin_tab,
                          max_pos, kcache, vcache, cache_rows, scale, causal, dyn, kv_len_max - 16 * q_tiles, 0, 16 * q_tiles, 0,
                          q_tiles, ws, max_splits, out_frag, out_tile_stride, R, xq_req_stride, out_req_stride, nullptr, nullptr, 0,
                          0, DFL_DYN_WORDS, cache_req_stride, stream);
}

// Candidate blocks of 17..32 rows: candidate c = TWO consecutive 16-row tiles of xq / out_frag; staging rows 0..bs-1.
extern "C" int dfl_attn_head_cand_t(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, int n_cand,
                                    int64_t xq_cand_stride, int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w,
                                    float eps, const void *cos_tab, const void *sin_tab, int max_pos, const void *kcache,
                                    const void *vcache, int cache_rows, float scale, int S, int bs, void *ws, int max_splits,
                                    void *out_frag, int64_t out_cand_stride, int64_t out_tile_stride, int q_tiles, void *k_out,
                                    void *v_out, int64_t kv_out_cand_stride, int out_rows, void *stream) {
  DFL_REQUIRE(n_cand >= 1 && k_out && v_out, "dfl_attn_head_cand_t: needs the K/V staging area");
  DFL_REQUIRE(q_tiles == 1 || q_tiles == 2, "dfl_attn_head_cand_t: q_tiles must be 1 or 2");
  return attn_head_launch(xq, ldq, q_col, k_col, v_col, nullptr, 0, 0, 0, n_q, n_kv, q_norm_w, k_norm_w, eps, cos_tab, sin_tab,
                          max_pos, const_cast<void *>(kcache), const_cast<void *>(vcache), cache_rows, scale, 1, nullptr, S, 0,
                          bs, S, q_tiles, ws, max_splits, out_frag, out_tile_stride, n_cand, xq_cand_stride, out_cand_stride, k_out,
                          v_out, kv_out_cand_stride, out_rows, 0, 0, stream);
}

// dfl_attn_head_batch on the fp32 K-PART SUMS of the qkv projection (dfl_gemm_f32_batch's output: [part][R tiles x 16
// rows][ldq] floats) instead of finished bf16 rows: request r's rows of part k at xq_parts + k * part_stride +
// r * xq_req_stride (floats); a value is bf16(part 0 + part 1).  The K parts of the projection meet in this launch's row
// loads, so the GEMM in front of it needs no slab / ticket / combine phase of its own.
extern "C" int dfl_attn_head_batch_f32(const float *xq_parts, int nparts, int64_t part_stride, int64_t ldq, int q_col, int k_col,
                                       int v_col, int R, int64_t xq_req_stride, int n_q, int n_kv, const void *q_norm_w,
                                       const void *k_norm_w, float eps, const void *cos_tab, const void *sin_tab, int max_pos,
                                       void *kcache, void *vcache, int cache_rows, int64_t cache_req_stride, float scale,
                                       int causal, const int32_t *dyn, int kv_len_max, void *ws, int max_splits,
                                       void *out_frag, int64_t out_req_stride, void *stream) {
  DFL_REQUIRE(dyn, "dfl_attn_head_batch_f32: the requests' lengths come from dyn (R records)");
  DFL_REQUIRE(nparts == 1 || nparts == 2, "dfl_attn_head_batch_f32: 1 or 2 K parts (got %d)", nparts);
  DFL_REQUIRE(kv_len_max >= 16 && kv_len_max <= cache_rows, "dfl_attn_head_batch_f32: kv_len_max=%d outside 16..cache_rows",
              kv_len_max);
  return attn_head_launch(xq_parts, ldq, q_col, k_col, v_col, nullptr, 0, 0, 0, n_q, n_kv, q_norm_w, k_norm_w, eps, cos_tab,
                          sin_tab, max_pos, kcache, vcache, cache_rows, scale, causal, dyn, kv_len_max - 16, 0, 16, 0, 1, ws,
                          max_splits, out_frag, 0, R, xq_req_stride, out_req_stride, nullptr, nullptr, 0, 0, DFL_DYN_WORDS,
                          cache_req_stride, stream, nullptr, nparts, part_stride);
}
