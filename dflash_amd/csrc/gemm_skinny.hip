// Skinny (M <= 32 rows) weight-streaming GEMM for gfx950.
//
// Shape of the problem: every nn.Linear of the draft step and of the
// target's verify sees at most 16 block rows (+ <=16 context rows), so each weight byte
// is used for <= 32 rows and the kernel is bound by HBM.  Design:
//   * weights are pre-packed (dfl_pack_weight) so a wave's 16-B-per-lane load is one
//     contiguous 1 KiB run AND one MFMA 16x16x32 A fragment: HBM -> VGPR -> matrix core,
//     no LDS, no shuffles (guide: "GEMV / M<=16: load straight to VGPRs, deep unroll,
//     late vmcnt");
//   * a 1024-thread workgroup = 16 waves splits K 16 ways and walks the workgroup's
//     column tiles, prefetching the next item's weights while the current one is in the
//     MFMA; per tile the 16 partial 16x16 tiles meet in LDS (double-buffered, one
//     barrier per tile) and 256 threads finish it;
//   * the activation operand comes from a row source (dfl_rows): ready-made frag16
//     fragments, plain bf16 rows, or the residual stream + per-tile sums of squares, in
//     which case the wave applies the RMSNorm itself while building its fragments —
//     the norm between two GEMMs is then not a launch of its own;
//   * epilogues: fp32 partials (K split over grid.y) / SiLU(gate)*up -> frag16 /
//     running bf16 argmax (lm_head) / residual add -> h, taps, sums of squares;
//   * K beyond 16 waves x 8 steps: either grid.y partials, or (CHUNKED) the workgroup
//     loops over K chunks itself so that the epilogue sees finished sums.
// MFMA utilisation is a few percent by construction; the roofline is HBM: every launch
// costs ~3.6 us + bytes / 6.5 TB/s (scripts/bench_gemm.py).
#include "gemm_rows.h"

namespace {

struct GemmArgs {
  // what the first weight request needs comes first: one s_load of the head of the kernel-argument block
  const bf16x8 *wp;  // packed weights [ntiles][KS][64]
  int KS;      // K / 32
  int ntiles;  // N / 16
  int nfr;     // k-steps per wave per chunk (<= FR)
  int nch;     // K chunks walked inside the workgroup (1 unless CHUNKED)
  // tiles [0, ntiles) are walked whole (workgroup b: b, b+G, ...); tiles [ntiles, ntiles + nhalf/2) are cut in two
  // 8-column halves, one per workgroup b < nhalf (tile ntiles + b/2, half b&1), as that workgroup's LAST item:
  // 384 tiles (qkv of an 8B model) are 1.5 tiles for each of 256 workgroups instead of 2 for each of 192
  int nhalf;
  const int32_t *dyn;
  RowSrc src[2];
  // EPI_F32
  float *out;  // [ksplit][MT*16][ldo]
  int ldo;
  // EPI_SILU
  bf16_t *act;  // frag16 [I/8][16][8]
  // EPI_SILU over the ACTIVE experts of a sparse-MoE layer (dfl_gemm_silu_mul_experts): the experts' packed gate/up
  // weights and frag16 outputs lie back to back, so (expert e, pair p) is pair e * npp + p of one tall matrix; the
  // workgroups share out the pairs of the n_active[0] experts listed in elist
  const int32_t *elist, *n_active;
  int npp;
  // EPI_ARGMAX
  int row0, nrows;
  int nrows_word;
  float *best_val;  // [gridDim.x][16]
  int *best_idx;
  float *best2_val;  // [gridDim.x][16] runner-up value (top-2 logit margin), optional
  bf16_t *logits;  // optional [16][N]
  int N;
  // EPI_RESID
  bf16_t *h_io;  // [16][ldh]: h <- bf16(h + bf16(acc)) (add_resid) or bf16(acc)
  int64_t ldh;
  int add_resid;
  bf16_t *tap;  // optional second copy of the new rows, row stride ldtap
  int64_t ldtap;
  float *ss_out;  // [ntiles][16] sum over the tile's 16 columns of h_new^2
};

#ifdef DFL_GEMM_STAMPS  // diagnostic build only (scripts/dbg_gemm_stamps.py): 100 MHz wall stamps of workgroup 0,
// and the start / first-item / end stamps of EVERY workgroup (is the tail a few late workgroups or all of them?)
__device__ unsigned long long g_gstamps[8];
__device__ unsigned long long g_wgstamps[256][4];
#define GSTAMP(i)                                                                         \
  do {                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_gstamps[i] = __builtin_amdgcn_s_memrealtime(); \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 256 && ((i) == 0 || (i) == 4 || (i) == 5))            \
      g_wgstamps[blockIdx.x][(i) == 0 ? 0 : (i) - 3] = __builtin_amdgcn_s_memrealtime();                           \
  } while (0)
#else
#define GSTAMP(i)
#endif

constexpr int gemm_fr(int MT, bool CHUNKED) { return CHUNKED ? 4 : (MT == 1 ? 8 : 4); }

// One launch.  Order of the prologue (round 3; measured with scripts/dbg_gemm_stamps.py, scripts/ab_prof.sh and against
// the pure-stream floor of scripts/probes/l2_prefetch_probe.hip — a stream of the same bytes in the same four launches
// per layer takes 62 us where the round-2 kernels took 80):
//   0. the lengths (dyn words): dependent SCALAR loads.  Requested behind the first weight burst they came back 5 - 9 us
//      later — most of what round 2 called the "rstd prologue".  Asked for before any vector load they cost nothing.
//   1. the activation side: the wave's row fragments (8 KB, from L2), for a normalised source one chunk of the RMSNorm
//      weight (staged ONCE per workgroup in LDS instead of 32 VGPRs per lane) and the producer's partial sums of
//      squares.  A wave's vector loads return in issue order and, at a launch's start, everything requested behind the
//      first weight burst queues behind ~32-64 MB in the memory system: asked for first, the rows are back after
//      ~1.5 us and the sum of squares -> rstd -> normalise chain (1-2 us of VALU on 16 waves) runs UNDER the weight
//      latency.  (Weights first: the rows came back 8 us later and the chain ran with HBM idle — gate/up 34.8 -> 36.6 us,
//      qkv 14.3 -> 16.4 us in the cycle.)  Launches without a normalised source put a workgroup barrier here, so that
//      every wave's row requests are in the CU's in-order memory pipe before any wave's weight request.
//   2. then item 0's weights and (residual launches without a norm) the residual value of the first tile;
//   3. only then the waits: rstd across the 16 waves, normalise, and the item loop (item i+1 requested at its top).
// Every load of the prologue is unconditional (clamped addresses, zero-length descriptors): a load under a branch
// costs a vmcnt(0) at the join, i.e. the whole burst.
// NORM: the launch has a normalised source (mode 2); the other instantiation carries none of that code or its LDS.
// Variants built, measured on the same box and dropped (DESIGN.md section 5 has the numbers; the code is in the history,
// commits dc953f3 and 5c920f1): item 1's weights requested in the prologue as well; the workgroup summing the squares
// of its own rows (v_dot2c) instead of loading the producer's partials; waiting for the rows before the first weight
// request; an 8-wave, three-buffer form of the K-chunked kernel.
template <int MT, bool CHUNKED, int EPI, bool NORM>
__device__ __forceinline__ void gemm_body(const GemmArgs &a) {
  GSTAMP(0);
  constexpr int FR = gemm_fr(MT, CHUNKED);
  // red[buf][wave][mt][256]: lane l owns floats 4l..4l+3 (its MFMA D regs)
  __shared__ float red[2][16][MT][256];
  __shared__ float ssred[NORM ? MT : 1][16][16];
  __shared__ bf16x8 nwl[NORM ? MT : 1][NORM ? 512 : 1];  // mode 2: the norm weight of K <= 4096, chunk c = elements 8c .. 8c+7

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;

  auto ks0_of = [&](int c) { return ((blockIdx.y * a.nch + c) * 16 + w) * a.nfr; };
  auto nf_of = [&](int c) {
    int nf = a.KS - ks0_of(c);
    return nf < 0 ? 0 : (nf > a.nfr ? a.nfr : nf);
  };

  // ---- tile sequence of this workgroup.  F32/ARGMAX/RESID: tiles bx, bx+G, ...
  // SILU: the packed weight interleaves (gate tile p, up tile p) and the sequence walks
  // pairs p = bx, bx+G, ... as gate,up,gate,up: the finishing thread meets a pair's two
  // sums in consecutive positions.
  const int stride = gridDim.x;
  constexpr bool SILU = EPI == EPI_SILU || EPI == EPI_SILU_E;
  constexpr bool moe = EPI == EPI_SILU_E;  // the tile sequence comes from the active-expert list (dependent scalar loads)
  const int nwhole = (!SILU && (int)blockIdx.x < a.ntiles) ? (a.ntiles - 1 - (int)blockIdx.x) / stride + 1 : 0;  // whole tiles of this workgroup
  const bool has_half = !SILU && (int)blockIdx.x < a.nhalf;
  const int myhalf = (int)blockIdx.x & 1;
  auto half_of = [&](int j) -> int { return (!SILU && j >= nwhole) ? myhalf : -1; };  // -1: a whole tile
  auto tile_of = [&](int j) -> int {
    if (!SILU) return j < nwhole ? (int)blockIdx.x + j * stride : a.ntiles + ((int)blockIdx.x >> 1);
    int p = (int)blockIdx.x + (j >> 1) * stride;
    if (moe) p = a.elist[p / a.npp] * a.npp + p % a.npp;  // scalar loads: p is uniform over the workgroup
    return 2 * p + (j & 1);
  };

  bf16x8 wA[FR], wB[FR];
  bf16x8 xA[MT][FR], xB[MT][FR];
  bf16x8 xr[MT][FR];  // activations of the (single) chunk stay in registers for the launch

  const int nf0 = nf_of(0);
  const int ngroups = SILU ? a.ntiles >> 1 : a.ntiles;
  int nseq;
  if (SILU) {
    const int npairs = moe ? a.n_active[0] * a.npp : a.ntiles >> 1;
    nseq = (int)blockIdx.x < npairs ? 2 * ((npairs - 1 - (int)blockIdx.x) / stride + 1) : 0;
  } else {
    nseq = nwhole + (has_half ? 1 : 0);
  }
  const int nitems = nseq * a.nch;

  int nv[MT];
  auto read_nv = [&]() {  // row validity (rows >= dyn[valid_word] count as zero): scalar loads, on their own counter
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const RowSrc &s0 = a.src[mt];
      nv[mt] = (s0.valid_word >= 0 && a.dyn) ? a.dyn[s0.valid_word] : 16;
    }
  };

  // finishing thread f < MT*256: row m = (f&255)>>4, column nl = f&15 of the tile
  float best = -INFINITY;  // running argmax
  int bestn = 0x7fffffff;
  float second = -INFINITY;  // runner-up VALUE as torch.topk(2) defines it: a tie with the best counts
  float gate_sum = 0.f;    // SILU: the pair's gate sum, kept across one position
  int arg_rows = 0;

  // chunked kernels: item (tile t, chunk c) = the wave's weights of that chunk AND its activation fragments.  Only the
  // REQUESTS are made here; the mask (rows beyond the valid count, k-steps beyond K) is applied in process(), right in
  // front of the MFMAs.  Round 2 masked right behind the requests: a wait for the activation loads — which return after
  // the weights requested in front of them — so every wave waited for the item it had JUST asked for and no two items
  // of a wave were ever in flight (down_proj at 4.9 TB/s, its workgroups ending 3.4 us apart).
  auto load_item_x = [&](bf16x8(&xb)[MT][FR], int c) {
    const int ks0 = ks0_of(c);
    int ks[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) ks[f] = ks0 + f < a.KS ? ks0 + f : a.KS - 1;
    bf16x8 wv[MT][FR];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int f = 0; f < FR; ++f) wv[mt][f] = xb[mt][f] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) issue_x<FR, false>(a.src[mt], ks, l, nv[mt], xb[mt], wv[mt]);  // no mode 2 here
  };
  auto load_item = [&](bf16x8(&wr)[FR], bf16x8(&xb)[MT][FR], int t, int c, int half) {
    load_ksteps<FR>(wr, a.wp + ((size_t)t * a.KS + ks0_of(c)) * 64, nf_of(c), l, half);
    if (CHUNKED) load_item_x(xb, c);  // the weights first (HBM), the activation fragments (L2) behind them
  };

  f32x4 acc[MT];
  // EPI_RESID without a normalised source (o_proj, down_proj: one tile per workgroup): this thread's residual value
  // of the FIRST tile, requested in the prologue
  constexpr bool PREF = EPI == EPI_RESID && !NORM;
  bf16_t resid0 = (bf16_t)0.f;

  auto finish = [&](int t, int pos) {
    const int buf = pos & 1;
    const int hf = half_of(pos);  // a half tile: the other 8 columns belong to the neighbouring workgroup
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4 *>(&red[buf][w][mt][l * 4]) = acc[mt];
    __syncthreads();
    // D layout of mfma 16x16x32 (A = W rows n, B = x^T cols m): lane L reg r holds
    // n_local = 4*(L>>4) + r, m = L & 15, i.e. float 4L + r of the wave's slot.
    if (tid < MT * 256) {
      const int mt = tid >> 8, ff = tid & 255;
      const int m = ff >> 4, nl = ff & 15;
      const int idx = 4 * (m + 16 * (nl >> 2)) + (nl & 3);
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < 16; ++ww) s += red[buf][ww][mt][idx];
      const bool colok = hf < 0 || (nl >> 3) == hf;
      if (EPI == EPI_F32) {
        if (colok) a.out[((size_t)(blockIdx.y * MT + mt) * 16 + m) * a.ldo + t * 16 + nl] = s;
      } else if (SILU) {
        if (buf == 0) {
          gate_sum = s;
        } else {
          // tf:modeling_qwen3.py:82: gate/up Linear outputs are bf16, silu is evaluated in
          // fp32 and rounded, the product is rounded again.
          const float gb = rbf(gate_sum), ub = rbf(s);
          const float act = rbf(gb / (1.f + __expf(-gb)));
          const int n = (t >> 1) * 16 + nl;
          a.act[((size_t)(n >> 3) * 16 + m) * 8 + (n & 7)] = f2bf(act * ub);
        }
      } else if (EPI == EPI_ARGMAX) {
        const int n = t * 16 + nl;
        const float vb = rbf(s);  // lm_head output is bf16 before argmax (model/dflash.py:238,247)
        const bool live = (m >= a.row0) && (m < a.row0 + arg_rows) && colok;
        if (a.logits && live) a.logits[(size_t)m * a.N + n] = f2bf(s);
        // n grows along the sequence for a fixed thread: strict '>' keeps the first maximum
        if (live && (vb > best || bestn == 0x7fffffff)) {
          second = best;
          best = vb;
          bestn = n;
        } else if (live) {
          second = fmaxf(second, vb);
        }
      } else {  // EPI_RESID: model/dflash.py:140,144 residual adds (bf16 + bf16 -> bf16)
        const int n = t * 16 + nl;
        const float v = rbf(s);  // the Linear's bf16 output
        bf16_t *hp = a.h_io + (int64_t)m * a.ldh + n;
        // the first tile's old value came with the prologue's loads (a dependent ~1 us load at the tail of the
        // one-tile launches o_proj / down_proj otherwise); later tiles of a workgroup read theirs here
        float hn = v;
        if (a.add_resid) hn = rbf(bf2f((PREF && pos == 0) ? resid0 : *hp) + v);
        if (colok) {
          *hp = f2bf(hn);
          if (a.tap) a.tap[(int64_t)m * a.ldtap + n] = f2bf(hn);
        }
        const float q = row_sum16(hn * hn);  // the 16 threads of row m are one DPP row
        if (nl == 0 && a.ss_out) a.ss_out[t * 16 + m] = q;
      }
    }
  };

  auto process = [&](bf16x8(&wr)[FR], bf16x8(&xb)[MT][FR], int t, int c, int pos) {
    if (!CHUNKED || c == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int ks0c = ks0_of(c);
#pragma unroll
    for (int f = 0; f < FR; ++f)  // k-steps past the wave's share carry zero weights AND zero activations
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bf16x8 xv = xr[mt][f];
        if (CHUNKED) {
          const bool keep = ks0c + f < a.KS && (a.src[mt].mode == 0 || (l & 15) < nv[mt]);
          xv = keep ? xb[mt][f] : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[f], xv, acc[mt], 0, 0, 0);
      }
    if (!CHUNKED || c == a.nch - 1) finish(t, pos);
  };

  // item 0 of this workgroup (F32 / ARGMAX / RESID): whole tile blockIdx.x, or — a launch cut entirely in halves
  // (plan_tiles: at most 128 tiles) — half blockIdx.x & 1 of tile blockIdx.x >> 1
  const int first_tile = (SILU || nwhole > 0) ? (int)blockIdx.x : a.ntiles + ((int)blockIdx.x >> 1);
  const int first_half = (SILU || nwhole > 0) ? -1 : myhalf;
  const bool first_live = SILU ? (int)blockIdx.x < ngroups : (nwhole > 0 || has_half);
  // the residual value of this thread's element of the FIRST tile (every thread asks: ff = tid & 255 — no branch)
  auto ask_resid0 = [&]() {
    if (PREF) {
      const int ff = tid & 255;
      resid0 = a.h_io[(int64_t)(ff >> 4) * a.ldh + first_tile * 16 + (ff & 15)];
    }
  };

  // ---- (0) the lengths: dependent SCALAR loads (argument block -> dyn -> word), on their own counter and out of order
  // with the vector loads — but through the same L2: requested behind the first weight burst they came back 5 - 9 us
  // later (the "rstd prologue" of round 2 was mostly this wait).  Asked for before any vector load they cost nothing.
  read_nv();
  if (EPI == EPI_ARGMAX) {
    arg_rows = a.nrows;
    if (a.dyn && a.nrows_word >= 0) arg_rows = a.dyn[a.nrows_word] - a.row0;
  }
  __builtin_amdgcn_sched_barrier(0);
  float rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) rstd[mt] = 1.f;
  if (!CHUNKED) {
    // ---- (1) the activation side, all unconditional: norm weight chunk (mode 2; any other mode reads 16 B of its own
    // operand and ignores them), partial sums of squares, the wave's activation fragments, the residual value
    bf16x8 nwv[MT];
    float ssv[MT][4];
    bool any_norm = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const RowSrc &s = a.src[mt];
#pragma unroll
      for (int u = 0; u < 4; ++u) ssv[mt][u] = 0.f;
      nwv[mt] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (NORM) {
        const int c = min(tid & 511, a.KS * 4 - 1);
        const char *nb = s.mode == 2 ? reinterpret_cast<const char *>(s.nw)
                                     : (s.mode == 0 ? reinterpret_cast<const char *>(s.frag) : reinterpret_cast<const char *>(s.rows));
        nwv[mt] = *reinterpret_cast<const bf16x8 *>(nb + c * 16);
      }
      if (NORM && s.mode == 2) {  // uniform (a kernel argument); the loads inside are clamped, not guarded
        // the nss partial sums of squares of a row are summed by the 16 waves together:
        // wave w takes partials w, w+16, ... (four loads in flight per lane per 256)
        const int m = l & 15, part = l >> 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = w + 16 * (part + 4 * u);
          ssv[mt][u] = s.ss[(i < s.nss ? i : s.nss - 1) * 16 + m];
        }
        any_norm = true;
      }
    }
    int ks[FR];
    bool take[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      take[f] = f < nf0;
      ks[f] = take[f] ? ks0_of(0) + f : 0;
    }
    bf16x8 raw[MT][FR], wv0[MT][FR];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int f = 0; f < FR; ++f) wv0[mt][f] = raw[mt][f] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) issue_x<FR, false>(a.src[mt], ks, l, 16, raw[mt], wv0[mt]);
    __builtin_amdgcn_sched_barrier(0);  // the ORDER of issue is the point: hipcc moved the weight requests in front
    if (!NORM) {  // every wave's row requests enter the CU's in-order memory pipe before any weight request
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- (2) item 0's weights.  No guards on the k-step count anywhere: a runtime guard makes hipcc branch around
    // every fragment load and wait vmcnt(0) after each (49 full waits instead of 16 in the SILU kernel, 34 -> 41 us).
    // load_ksteps clips at the wave's share through the buffer descriptor instead (zero weights, no traffic), and
    // activations past the share are zero as well.  (The expert list of an MoE launch is a dependent scalar load: that
    // launch asks for its first weights once it knows the tile.)
    if (!moe)
      load_ksteps<FR>(wA, a.wp + ((size_t)(SILU ? 2 * blockIdx.x : first_tile) * a.KS + ks0_of(0)) * 64,
                      first_live ? nf0 : 0, l, first_half);
    else if (nitems > 0)
      load_ksteps<FR>(wA, a.wp + ((size_t)tile_of(0) * a.KS + ks0_of(0)) * 64, nf0, l);
    GSTAMP(6);
    __builtin_amdgcn_sched_barrier(0);
    ask_resid0();
    GSTAMP(1);
    // ---- (3) rstd, normalise
    if (NORM && any_norm) {  // uniform over the workgroup: the modes are kernel arguments
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const RowSrc &s = a.src[mt];
        if (NORM) nwl[mt][tid & 511] = nwv[mt];  // both halves of the workgroup store the same 512 chunks
        if (NORM && s.mode == 2) {
          const int m = l & 15, part = l >> 4;
          float v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = (w + 16 * (part + 4 * u)) < s.nss ? ssv[mt][u] : 0.f;
          float t = (v[0] + v[1]) + (v[2] + v[3]);
          for (int base = 256; base < s.nss; base += 256) {  // more than 256 partials: rare, late loads
            float v2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int i = base + w + 16 * (part + 4 * u);
              v2[u] = s.ss[(i < s.nss ? i : s.nss - 1) * 16 + m];
              v2[u] = i < s.nss ? v2[u] : 0.f;
            }
            t += (v2[0] + v2[1]) + (v2[2] + v2[3]);
          }
          t += __shfl_xor(t, 16, 64);
          t += __shfl_xor(t, 32, 64);
          if (part == 0) ssred[mt][w][m] = t;
        }
      }
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if (a.src[mt].mode == 2) {
          float t = 0.f;
#pragma unroll
          for (int ww = 0; ww < 16; ++ww) t += ssred[mt][ww][l & 15];
          rstd[mt] = rsqrtf(t / (float)(a.KS * 32) + a.src[mt].eps);
        }
    }
    GSTAMP(2);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (NORM)
        finish_xl<FR>(a.src[mt], ks, take, l, nv[mt], rstd[mt], raw[mt], nwl[mt], xr[mt]);
      else
        finish_x<FR, false>(a.src[mt], take, l, nv[mt], 1.f, raw[mt], wv0[mt], xr[mt]);
    }
  } else {
    // chunked kernels build their activation fragments per item: item 0's behind the weights already requested
    if (nitems > 0 || !moe)
      load_ksteps<FR>(wA, a.wp + ((size_t)first_tile * a.KS + ks0_of(0)) * 64, first_live ? nf0 : 0, l, first_half);
    GSTAMP(6);
    load_item_x(xA, 0);
    ask_resid0();
    GSTAMP(1);
    GSTAMP(2);
  }

  GSTAMP(3);
  if (nitems > 0) {
    int j = 0, c = 0;  // current item
    for (int i = 0;; i += 2) {
      int jn = j, cn = c + 1;  // item i+1
      if (cn == a.nch) {
        cn = 0;
        ++jn;
      }
      // (requesting unconditionally — a zero-length dummy past the last item, so that hipcc can count the loads in
      // flight — was tried twice in this 16-wave kernel, rounds 2 and 3: 20.4 -> 20.8 us on down_proj; not kept)
      if (i + 1 < nitems) load_item(wB, xB, tile_of(jn), cn, half_of(jn));
      process(wA, xA, tile_of(j), c, j);
      if (i == 0) GSTAMP(4);
      if (i + 1 >= nitems) break;
      int j2 = jn, c2 = cn + 1;  // item i+2
      if (c2 == a.nch) {
        c2 = 0;
        ++j2;
      }
      if (i + 2 < nitems) load_item(wA, xA, tile_of(j2), c2, half_of(j2));
      process(wB, xB, tile_of(jn), cn, jn);
      if (i + 2 >= nitems) break;
      j = j2;
      c = c2;
    }
  }

  GSTAMP(5);
  if (EPI == EPI_ARGMAX) {
    // the 16 threads of row m are consecutive lanes: shuffle down to lane nl == 0
    if (tid < 256) {
#pragma unroll
      for (int o = 1; o <= 8; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bestn, o, 64);
        const float os = __shfl_xor(second, o, 64);
        second = fmaxf(fmaxf(second, os), fminf(best, ov));  // runner-up of the union
        if (ov > best || (ov == best && oi < bestn)) {
          best = ov;
          bestn = oi;
        }
      }
      if ((tid & 15) == 0) {
        a.best_val[blockIdx.x * 16 + (tid >> 4)] = best;
        a.best_idx[blockIdx.x * 16 + (tid >> 4)] = bestn;
        if (a.best2_val) a.best2_val[blockIdx.x * 16 + (tid >> 4)] = second;
      }
    }
  }
}

template <int MT, bool CHUNKED, int EPI, bool NORM>
__global__ __launch_bounds__(1024) void k_gemm(GemmArgs a) {
  gemm_body<MT, CHUNKED, EPI, NORM>(a);
}

// the instantiation for this launch's row sources (a normalised source needs the NORM kernels)
template <int MT, bool CHUNKED, int EPI>
void launch_gemm(const GemmArgs &a, dim3 grid, hipStream_t stream) {
  bool norm = false;
  for (int mt = 0; mt < MT; ++mt) norm |= a.src[mt].mode == 2;
  if constexpr (!CHUNKED) {  // (the chunked form takes no normalised source: its callers reject one)
    if (norm) {
      hipLaunchKernelGGL((k_gemm<MT, false, EPI, true>), grid, dim3(1024), 0, stream, a);
      return;
    }
  }
  hipLaunchKernelGGL((k_gemm<MT, CHUNKED, EPI, false>), grid, dim3(1024), 0, stream, a);
}

// Cross-workgroup finish of the fused argmax: one wave per row.  margin_out (optional):
// top-1 minus top-2 logit of the row — the reference's only confidence statistic
// (benchmark_candidate_solutions.py:296-302: topk(2) values, a tie gives 0).
__global__ __launch_bounds__(64) void k_argmax_finish(const float *best_val, const int *best_idx, const float *best2_val,
                                                      int nblk, int row0, int nrows, const int32_t *dyn, int nrows_word,
                                                      int64_t *out_ids, int out_off, float *margin_out) {
  int rows = nrows;
  if (dyn && nrows_word >= 0) rows = dyn[nrows_word] - row0;
  const int m = row0 + blockIdx.x;
  if ((int)blockIdx.x >= rows) return;
  float bv = -INFINITY, sv = -INFINITY;
  int bi = 0x7fffffff;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    const float ov = best_val[b * 16 + m];
    const int oi = best_idx[b * 16 + m];
    const float os = best2_val ? best2_val[b * 16 + m] : -INFINITY;
    if (oi != 0x7fffffff) sv = fmaxf(fmaxf(sv, os), bi == 0x7fffffff ? -INFINITY : fminf(bv, ov));
    if (ov > bv || (ov == bv && oi < bi) || bi == 0x7fffffff) {
      bv = ov;
      bi = oi;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    const float os = __shfl_xor(sv, o, 64);
    const bool have = bi != 0x7fffffff, ohave = oi != 0x7fffffff;
    sv = fmaxf(sv, os);
    if (have && ohave) sv = fmaxf(sv, fminf(bv, ov));
    if (ohave && (!have || ov > bv || (ov == bv && oi < bi))) {
      bv = ov;
      bi = oi;
    }
  }
  if (threadIdx.x == 0) {
    out_ids[out_off + blockIdx.x] = (int64_t)bi;
    if (margin_out) margin_out[out_off + blockIdx.x] = bv - sv;
  }
}

// ---- one-time packing ---------------------------------------------------------
// out chunk index c = ((t*KS + ks)*64 + l): 16 B from W[t*16 + (l&15)][ks*32 + (l>>4)*8 ...]
__global__ void k_pack_weight(const bf16x8 *__restrict__ w, bf16x8 *__restrict__ wp, int ntiles, int KS,
                              int tile_mul, int tile_add) {
  const size_t total = (size_t)ntiles * KS * 64;
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < total; c += (size_t)gridDim.x * blockDim.x) {
    const int l = c & 63;
    const size_t tk = c >> 6;
    const int ks = tk % KS;
    const size_t t = tk / KS;
    const size_t row = t * 16 + (l & 15);
    const size_t src = row * (size_t)(KS * 4) + (size_t)ks * 4 + (l >> 4);  // in 16-B units
    const size_t dt = t * tile_mul + tile_add;
    wp[(dt * KS + ks) * 64 + l] = w[src];
  }
}

// h[m] = embed[ids[m]] (bf16 rows) and the rows' sums of squares (one partial per row)
__global__ __launch_bounds__(256) void k_embed_rows(const bf16_t *embed, const int64_t *ids, bf16_t *h, int H,
                                                    float *ss_out, const int32_t *dyn, int dyn_word) {
  __shared__ float wsum[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int nv = dyn ? dyn[dyn_word] : 16;
  float ss = 0.f;
  if (m < nv) {
    const bf16_t *src = embed + ids[m] * (int64_t)H;
    for (int c = tid; c < (H >> 3); c += 256) {
      const bf16x8 v = *reinterpret_cast<const bf16x8 *>(src + c * 8);
      *reinterpret_cast<bf16x8 *>(h + (int64_t)m * H + c * 8) = v;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += bf2f(v[j]) * bf2f(v[j]);
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) ss_out[m] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

}  // namespace

#ifdef DFL_GEMM_STAMPS
extern "C" int dfl_debug_read_gemm_stamps(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gstamps), sizeof(unsigned long long) * 8);
}
extern "C" int dfl_debug_read_wg_stamps(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wgstamps), sizeof(unsigned long long) * 256 * 4);
}
#endif

// Grid and tile split of a launch over `ntiles` column tiles (single K pass).  When the tiles do not divide over the
// 256 CUs and the remainder is at most half a round (r <= 128), the last r tiles are cut in 8-column halves, one for
// each of 2r workgroups: every CU streams within half a tile of the same byte count.  (qkv of an 8B model: 384 tiles =
// 1.5 per workgroup on 256 CUs instead of 2 on 192 of them; lm_head: 9496 = 37 x 256 + 24.)
static void plan_tiles(GemmArgs &a, int ntiles, bool allow_half, int &gx) {
  gx = grid_x_for(ntiles);
  a.ntiles = ntiles;
  a.nhalf = 0;
  const int r = ntiles % 256;
  // (round 4) at most 128 tiles: ALL of them in halves, one per workgroup — twice the CUs stream half the bytes each
  // (o_proj / down_proj of a hidden-2048 model: 128 workgroups of one 16-column tile were half the chip).
  // DFL_HALF_SMALL=0: whole tiles, the round-3 plan.
  static const bool half_small = [] { const char *e = getenv("DFL_HALF_SMALL"); return !(e && e[0] == '0'); }();
  if (allow_half && half_small && ntiles <= 128) {
    gx = 2 * ntiles;
    a.ntiles = 0;
    a.nhalf = 2 * ntiles;
    return;
  }
  if (!allow_half || ntiles <= 256 || r == 0 || r > 128) return;
  gx = 256;
  a.ntiles = ntiles - r;
  a.nhalf = 2 * r;
}

extern "C" int dfl_pack_weight(const void *w, void *wp, int N, int K, void *stream) {
  DFL_REQUIRE(w && wp, "dfl_pack_weight: null pointer");
  DFL_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "dfl_pack_weight: need N%%16==0, K%%32==0 (N=%d K=%d)", N, K);
  hipLaunchKernelGGL(k_pack_weight, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const bf16x8 *)w, (bf16x8 *)wp,
                     N / 16, K / 32, 1, 0);
  DFL_CHECK_LAUNCH("dfl_pack_weight");
  return DFL_OK;
}

extern "C" int dfl_pack_weight_gateup(const void *gate, const void *up, void *wp, int I, int K, void *stream) {
  DFL_REQUIRE(gate && up && wp, "dfl_pack_weight_gateup: null pointer");
  DFL_REQUIRE(I > 0 && K > 0 && I % 16 == 0 && K % 32 == 0, "dfl_pack_weight_gateup: need I%%16==0, K%%32==0");
  hipLaunchKernelGGL(k_pack_weight, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const bf16x8 *)gate,
                     (bf16x8 *)wp, I / 16, K / 32, 2, 0);
  hipLaunchKernelGGL(k_pack_weight, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const bf16x8 *)up, (bf16x8 *)wp,
                     I / 16, K / 32, 2, 1);
  DFL_CHECK_LAUNCH("dfl_pack_weight_gateup");
  return DFL_OK;
}

extern "C" int dfl_embed_rows(const void *embed, const int64_t *ids, void *h_out, int H, float *ss_out,
                              const int32_t *dyn, int dyn_word, void *stream) {
  DFL_REQUIRE(embed && ids && h_out && ss_out, "dfl_embed_rows: null pointer");
  DFL_REQUIRE(H > 0 && H % 8 == 0, "dfl_embed_rows: H%%8 != 0");
  hipLaunchKernelGGL(k_embed_rows, dim3(16), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)embed, ids,
                     (bf16_t *)h_out, H, ss_out, dyn, dyn_word);
  DFL_CHECK_LAUNCH("dfl_embed_rows");
  return DFL_OK;
}

extern "C" int dfl_gemm_f32(const void *wp, const dfl_rows *x0, const dfl_rows *x1, int mt, int N, int K, int ksplit,
                            float *out, const int32_t *dyn, void *stream) {
  DFL_REQUIRE(wp && out, "dfl_gemm_f32: null pointer");
  DFL_REQUIRE(mt == 1 || mt == 2, "dfl_gemm_f32: mt must be 1 or 2");
  DFL_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "dfl_gemm_f32: need N%%16==0, K%%32==0 (N=%d K=%d)", N, K);
  const int KS = K / 32;
  const int fr_max = mt == 1 ? 8 : 4;
  DFL_REQUIRE(ksplit >= pick_ksplit_min(KS, fr_max) && ksplit <= 64, "dfl_gemm_f32: ksplit=%d too small for K=%d (need >= %d)",
              ksplit, K, pick_ksplit_min(KS, fr_max));
  GemmArgs a{};
  if (!fill_src(a.src[0], x0, K, "dfl_gemm_f32")) return DFL_EINVAL;
  if (mt == 2 && !fill_src(a.src[1], x1, K, "dfl_gemm_f32")) return DFL_EINVAL;
  a.wp = (const bf16x8 *)wp;
  a.dyn = dyn;
  a.KS = KS;
  a.ntiles = N / 16;
  a.nfr = (KS + 16 * ksplit - 1) / (16 * ksplit);
  a.nch = 1;
  a.out = out;
  a.ldo = N;
  int gx1 = 0;
  if (ksplit == 1) plan_tiles(a, N / 16, true, gx1);
  dim3 grid(ksplit == 1 ? gx1 : grid_x_for(a.ntiles, ksplit), ksplit);
  if (mt == 1)
    launch_gemm<1, false, EPI_F32>(a, grid, (hipStream_t)stream);
  else
    launch_gemm<2, false, EPI_F32>(a, grid, (hipStream_t)stream);
  DFL_CHECK_LAUNCH("dfl_gemm_f32");
  return DFL_OK;
}

extern "C" int dfl_gemm_silu_mul(const void *wp_gateup, const dfl_rows *x, int I, int K, void *act_frag,
                                 const int32_t *dyn, void *stream) {
  DFL_REQUIRE(wp_gateup && act_frag, "dfl_gemm_silu_mul: null pointer");
  DFL_REQUIRE(I > 0 && K > 0 && I % 16 == 0 && K % 32 == 0, "dfl_gemm_silu_mul: need I%%16==0, K%%32==0");
  const int KS = K / 32;
  DFL_REQUIRE(KS <= 16 * 8, "dfl_gemm_silu_mul: K=%d needs a K split, which the fused activation cannot take", K);
  GemmArgs a{};
  if (!fill_src(a.src[0], x, K, "dfl_gemm_silu_mul")) return DFL_EINVAL;
  a.wp = (const bf16x8 *)wp_gateup;
  a.dyn = dyn;
  a.KS = KS;
  a.ntiles = 2 * (I / 16);
  a.nfr = (KS + 15) / 16;
  a.nch = 1;
  a.act = (bf16_t *)act_frag;
  launch_gemm<1, false, EPI_SILU>(a, dim3(grid_x_for(I / 16), 1), (hipStream_t)stream);
  DFL_CHECK_LAUNCH("dfl_gemm_silu_mul");
  return DFL_OK;
}

// One launch for the gate/up projection of EVERY active expert of a sparse-MoE layer (Qwen3-MoE:
// tf:models/qwen3_moe/modeling_qwen3_moe.py, Qwen3MoeExperts.forward: act_fn(gate) * up per expert).  The 16 block rows
// are one MFMA tile whatever subset of them an expert serves, so an expert costs its weight bytes once.  One workgroup
// per CU; each builds its activation fragments ONCE and walks its share of the (active expert, gate/up pair) items,
// the next item's weights prefetched under the current one — round 2, first form: grid.z = expert, 8 workgroups per
// expert, every one with its own prologue and no overlap between successive workgroups of a CU: 4.7 TB/s at 82
// active experts (109.6 us per layer of a 30B-A3B-shaped target).
extern "C" int dfl_gemm_silu_mul_experts(const void *wp_gateup, int64_t wp_expert_stride, const dfl_rows *x, int E, int I, int K,
                                         void *act_frag, int64_t act_expert_stride, const int32_t *list,
                                         const int32_t *n_active, const int32_t *dyn, void *stream) {
  DFL_REQUIRE(wp_gateup && act_frag && list && n_active, "dfl_gemm_silu_mul_experts: null pointer");
  DFL_REQUIRE(E >= 1 && E <= 1024 && I > 0 && K > 0 && I % 16 == 0 && K % 32 == 0, "dfl_gemm_silu_mul_experts: bad shape");
  DFL_REQUIRE(wp_expert_stride == (int64_t)2 * I * K && act_expert_stride == (int64_t)16 * I,
              "dfl_gemm_silu_mul_experts: the experts' weights and outputs must lie back to back (strides %lld, %lld)",
              (long long)wp_expert_stride, (long long)act_expert_stride);
  const int KS = K / 32;
  DFL_REQUIRE(KS <= 16 * 8, "dfl_gemm_silu_mul_experts: K=%d exceeds 4096", K);
  GemmArgs a{};
  if (!fill_src(a.src[0], x, K, "dfl_gemm_silu_mul_experts")) return DFL_EINVAL;
  a.wp = (const bf16x8 *)wp_gateup;
  a.dyn = dyn;
  a.KS = KS;
  a.ntiles = 2 * (I / 16) * E;
  a.nfr = (KS + 15) / 16;
  a.nch = 1;
  a.act = (bf16_t *)act_frag;
  a.elist = list;
  a.n_active = n_active;
  a.npp = I / 16;
  launch_gemm<1, false, EPI_SILU_E>(a, dim3(256, 1), (hipStream_t)stream);
  DFL_CHECK_LAUNCH("dfl_gemm_silu_mul_experts");
  return DFL_OK;
}

extern "C" int64_t dfl_argmax_ws_bytes(void) { return 256 * 16 * (int64_t)(2 * sizeof(float) + sizeof(int)); }

namespace {
int gemm_argmax_impl(const void *wp, const dfl_rows *x, int V, int K, int row0, int nrows, const int32_t *dyn,
                     int nrows_dyn_word, void *ws, int64_t *out_ids, int out_off, void *logits, float *margin_out,
                     hipEvent_t ev0, hipEvent_t ev1, void *stream) {
  DFL_REQUIRE(wp && ws && out_ids, "dfl_gemm_argmax: null pointer");
  DFL_REQUIRE(V > 0 && K > 0 && V % 16 == 0 && K % 32 == 0, "dfl_gemm_argmax: need V%%16==0, K%%32==0 (V=%d K=%d)", V, K);
  DFL_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= 16, "dfl_gemm_argmax: rows [%d,%d) outside the 16-row tile", row0,
              row0 + nrows);
  const int KS = K / 32;
  DFL_REQUIRE(KS <= 16 * 8, "dfl_gemm_argmax: K=%d exceeds 4096 (argmax needs finished sums)", K);
  GemmArgs a{};
  if (!fill_src(a.src[0], x, K, "dfl_gemm_argmax")) return DFL_EINVAL;
  a.wp = (const bf16x8 *)wp;
  a.dyn = dyn;
  a.KS = KS;
  a.ntiles = V / 16;
  a.nfr = (KS + 15) / 16;
  a.nch = 1;
  a.row0 = row0;
  a.nrows = nrows;
  a.nrows_word = nrows_dyn_word;
  a.best_val = (float *)ws;
  a.best_idx = (int *)((char *)ws + 256 * 16 * sizeof(float));
  a.best2_val = margin_out ? (float *)((char *)ws + 256 * 16 * (sizeof(float) + sizeof(int))) : nullptr;
  a.logits = (bf16_t *)logits;
  a.N = V;
  int gx = 0;
  plan_tiles(a, V / 16, true, gx);
  if (ev0) (void)hipEventRecord(ev0, (hipStream_t)stream);
  launch_gemm<1, false, EPI_ARGMAX>(a, dim3(gx, 1), (hipStream_t)stream);
  if (ev1) (void)hipEventRecord(ev1, (hipStream_t)stream);
  hipLaunchKernelGGL(k_argmax_finish, dim3(16), dim3(64), 0, (hipStream_t)stream, a.best_val, a.best_idx, a.best2_val, gx,
                     row0, nrows, dyn, nrows_dyn_word, out_ids, out_off, margin_out);
  DFL_CHECK_LAUNCH("dfl_gemm_argmax");
  return DFL_OK;
}
}  // namespace

extern "C" int dfl_gemm_argmax(const void *wp, const dfl_rows *x, int V, int K, int row0, int nrows,
                               const int32_t *dyn, int nrows_dyn_word, void *ws, int64_t *out_ids, int out_off,
                               void *logits, float *margin_out, void *stream) {
  return gemm_argmax_impl(wp, x, V, K, row0, nrows, dyn, nrows_dyn_word, ws, out_ids, out_off, logits, margin_out, nullptr,
                          nullptr, stream);
}

extern "C" int dfl_gemm_argmax_timed(const void *wp, const dfl_rows *x, int V, int K, int row0, int nrows,
                                     const int32_t *dyn, int nrows_dyn_word, void *ws, int64_t *out_ids, int out_off,
                                     void *logits, float *margin_out, void *ev_start, void *ev_end, void *stream) {
  return gemm_argmax_impl(wp, x, V, K, row0, nrows, dyn, nrows_dyn_word, ws, out_ids, out_off, logits, margin_out,
                          (hipEvent_t)ev_start, (hipEvent_t)ev_end, stream);
}


extern "C" int dfl_gemm_resid(const void *wp, const dfl_rows *x, int N, int K, void *h_io, int64_t ldh,
                              int add_residual, void *tap, int64_t ldtap, float *ss_out, const int32_t *dyn,
                              void *stream) {
  DFL_REQUIRE(wp && h_io, "dfl_gemm_resid: null pointer");
  DFL_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "dfl_gemm_resid: need N%%16==0, K%%32==0 (N=%d K=%d)", N, K);
  DFL_REQUIRE(ldh >= N && (!tap || ldtap >= N), "dfl_gemm_resid: row strides shorter than N");
  const int KS = K / 32;
  GemmArgs a{};
  if (!fill_src(a.src[0], x, K, "dfl_gemm_resid")) return DFL_EINVAL;
  a.wp = (const bf16x8 *)wp;
  a.dyn = dyn;
  a.KS = KS;
  a.ntiles = N / 16;
  a.h_io = (bf16_t *)h_io;
  a.ldh = ldh;
  a.add_resid = add_residual ? 1 : 0;
  a.tap = (bf16_t *)tap;
  a.ldtap = ldtap;
  a.ss_out = ss_out;
  int gx = 0;
  // (a half tile has no slot of its own in ss_out: launches that leave sums of squares keep whole tiles)
  plan_tiles(a, N / 16, KS <= 16 * 8 && !ss_out, gx);
  const dim3 grid(gx, 1);
  if (KS <= 16 * 8) {  // the whole K fits the 16 waves x 8 steps of one pass
    a.nfr = (KS + 15) / 16;
    a.nch = 1;
    launch_gemm<1, false, EPI_RESID>(a, grid, (hipStream_t)stream);
  } else {  // walk K in chunks of 16 waves x 4 steps inside the workgroup
    DFL_REQUIRE(a.src[0].mode != 2, "dfl_gemm_resid: a normalised source needs K <= 4096");
    a.nfr = 4;
    a.nch = (KS + 63) / 64;  // 64 k-steps per chunk: 16 waves x 4
    launch_gemm<1, true, EPI_RESID>(a, grid, (hipStream_t)stream);
  }
  DFL_CHECK_LAUNCH("dfl_gemm_resid");
  return DFL_OK;
}
