// Skinny GEMM over a ragged batch of requests: R <= 4 requests, one 16-row tile each,
// share ONE pass over the weights (BASELINE.json configs[2]: 4 requests per GPU;
// SURVEY.md §8e "within a GPU the requests are a ragged batch for the kernels").
//
// Same dataflow as gemm_skinny.hip — packed weights HBM -> VGPR -> MFMA, a 16-wave
// workgroup splitting K, the 16 partial tiles meeting in LDS — with one difference that
// the register file forces: the activations of MT tiles over K = 4096 are MT x 128 KB,
// i.e. the whole 512 KB VGPR file of a CU at MT = 4.  A workgroup therefore keeps only a
// 2048-wide K slice of every tile resident (8 waves x 8 k-steps, 128 VGPRs at MT = 4) and
// K is cut over grid.y.  The fused epilogues need finished sums, so the K parts of a
// tile column meet through HBM-resident fp32 partials (9 % of the weight bytes at
// MT = 4) and the LAST workgroup of a column group to arrive — an agent-scope ticket, no
// spinning, nothing to deadlock — sums them in a fixed order and runs the epilogue.
// Request r reads its rows at base + r * stride and its lengths from dyn + 8 r.
#include "gemm_rows.h"
#include "gemm_ring.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct GemmBArgs {
  const bf16x8 *wp;  // packed weights [ntiles][KS][64]
  RowSrc src;        // request 0; request r at the strides below
  int64_t frag_stride, rows_stride, ss_stride;  // in elements of the respective buffers
  const int32_t *dyn;  // [R][DFL_DYN_WORDS]
  int KS, ntiles, nfr;
  float *part;   // [ksplit][ntiles][MT][256] fp32 slabs, D layout (fused epilogues, ksplit > 1)
  int64_t part_bytes;
  int *tickets;  // [gridDim.x], zero between launches
  // EPI_F32
  float *out;  // [ksplit][MT*16][ldo]
  int ldo;
  // EPI_SILU
  bf16_t *act;  // frag16 [I/8][16][8] per request
  int64_t act_stride;
  // EPI_ARGMAX
  int row0, nrows, nrows_word;
  float *best_val;  // [gridDim.x][MT][16]
  int *best_idx;
  bf16_t *logits;  // optional [16][N] per request
  int64_t logits_stride;
  int N;
  // EPI_RESID
  bf16_t *h_io;
  int64_t ldh, h_stride;
  int add_resid;
  bf16_t *tap;
  int64_t ldtap, tap_stride;
  float *ss_out;  // [ntiles][16] per request
  int64_t ss_out_stride;
};

// NW waves split the workgroup's K part; each keeps FR k-steps of every request's tile in
// registers.  8 waves x 8 k-steps (not 16 x 4): a 512-thread workgroup alone on a CU gives
// each wave 256 VGPRs — 128 for the activations of 4 tiles, 96 for THREE rotating weight
// buffers, so two items (128 KB per CU) are in flight while one is in the MFMA.
//
// Finishing a tile: wave mt < MT sums the NW partial tiles of request mt in the MFMA D
// layout (lane L holds floats 4L..4L+3 = row m = L & 15, columns 4 (L >> 4) .. +3):
// conflict-free ds_read_b128, and every global access of the epilogue is 8 or 16 bytes wide.
// The other waves run ahead into the next item.
//
// K parts (grid.y > 1, fused epilogues): each part's finished 16x16 sums go to HBM-resident
// slabs with WRITE-THROUGH (sc1) 16-byte stores — no release fence, which would write back
// the whole XCD L2 (MI355X_MICROARCH.md § visibility: 1.7-6.5 us per workgroup; the first
// version of this kernel, with plain 4-byte slab stores + release, ran the ticketed GEMMs at
// 1.0-2.6 TB/s) — then drain, barrier, one relaxed agent-scope ticket.  The last part to
// arrive acquires (L1 invalidate only) and ALL its waves combine the slabs, 8 (tile, request)
// items per wave in flight, in a fixed part order.
template <int MT, int EPI>
__global__ __launch_bounds__(512) void k_gemm_b(GemmBArgs a) {
  constexpr int FR = 8, NW = 8;
  constexpr int TPU = EPI == EPI_SILU ? 2 : 1;  // tiles per epilogue unit (SILU: gate + up)
  // MT = 4 lives at the 256-VGPR limit in its main loop (128 activations + 96 weights + 16
  // accumulators): the epilogue code is kept out of that loop — even a single K part goes
  // through its slab and the combine phase, where the streaming registers are dead — or hipcc
  // spills a weight fragment right after loading it (vmcnt(0) in the prefetch ring).
  constexpr bool INLINE_EPI = MT < 4;
  __shared__ float red[2][NW][MT][256];
  __shared__ int s_last;

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int ky = blockIdx.y, nky = gridDim.y;
  const int fm = l & 15, fg = l >> 4;  // finishing lane: row, column group

  RowSrc src[MT];
  int nv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    src[mt] = a.src;
    if (a.src.frag) src[mt].frag = a.src.frag + (mt * a.frag_stride) / 8;
    if (a.src.rows) src[mt].rows = a.src.rows + mt * a.rows_stride;
    nv[mt] = (a.src.valid_word >= 0 && a.dyn) ? a.dyn[mt * DFL_DYN_WORDS + a.src.valid_word] : 16;
  }

  const int ks0 = (ky * NW + w) * a.nfr;
  int nf0 = a.KS - ks0;
  nf0 = nf0 < 0 ? 0 : (nf0 > a.nfr ? a.nfr : nf0);

  // tile sequence (see gemm_skinny.hip): tiles bx, bx+G, ...; SILU walks (gate, up) pairs
  const int stride = gridDim.x;
  int nseq;
  if (EPI == EPI_SILU) {
    const int npairs = a.ntiles >> 1;
    nseq = (int)blockIdx.x < npairs ? 2 * ((npairs - 1 - (int)blockIdx.x) / stride + 1) : 0;
  } else {
    nseq = (int)blockIdx.x < a.ntiles ? (a.ntiles - 1 - (int)blockIdx.x) / stride + 1 : 0;
  }
  auto tile_of = [&](int j) -> int {
    return EPI == EPI_SILU ? 2 * ((int)blockIdx.x + (j >> 1) * stride) + (j & 1) : (int)blockIdx.x + j * stride;
  };

  // Weights through buffer loads clipped at the wave's share (gemm_rows.h).  With global loads
  // the MT = 4 kernels sat at 256 VGPRs and hipcc spilled the eighth fragment of a weight buffer
  // right after loading it: `s_waitcnt vmcnt(0); scratch_store` in the main loop, which
  // serialises the whole prefetch ring.
  // (live = false: a dummy request — empty descriptor, zeros, no traffic — see the main loop)
  auto load_item = [&](bf16x8(&wr)[FR], int t, bool live = true) {
    load_ksteps<FR>(wr, a.wp + ((size_t)t * a.KS + ks0) * 64, live ? nf0 : 0, l);
  };

  // ---- prologue: the first two weight items leave for HBM, then this wave's K slice of every
  // request's tile (it stays in registers for the launch).  Unlike the single-request kernel the
  // activations go SECOND here: at 4 tiles they are 256 KB per workgroup, and asking for them
  // first delays the weight stream by more than it saves (6.47 -> 6.69 ms per 4-request cycle).
  bf16x8 xr[MT][FR];
  bf16x8 wA[FR], wB[FR], wC[FR];
  if (nseq > 0) load_item(wA, tile_of(0));
  if (nseq > 1) load_item(wB, tile_of(1));
  {
    int ks[FR];
    bool take[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      take[f] = f < nf0;
      ks[f] = take[f] ? ks0 + f : 0;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) build_x<FR, false>(src[mt], ks, take, l, nv[mt], 1.f, xr[mt]);
  }

  // ---- epilogue state of this lane, per request
  float best[MT];
  int bestn[MT], arg_rows[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    best[mt] = -INFINITY;
    bestn[mt] = 0x7fffffff;
    arg_rows[mt] = 0;
    if (EPI == EPI_ARGMAX) {
      arg_rows[mt] = a.nrows;
      if (a.dyn && a.nrows_word >= 0) arg_rows[mt] = a.dyn[mt * DFL_DYN_WORDS + a.nrows_word] - a.row0;
    }
  }
  f32x4 gate_sum = {0.f, 0.f, 0.f, 0.f};

  // The fused epilogues on the finished sums s[r] of (request mt, row fm, columns
  // t*16 + 4 fg + r).  SILU: called with the gate tile's sums in g4 and the up tile's in s.
  auto epilogue = [&](int t, int mt, const f32x4 &s, const f32x4 &g4) {
    const int n0 = (EPI == EPI_SILU ? (t >> 1) : t) * 16 + 4 * fg;
    if (EPI == EPI_SILU) {  // tf:modeling_qwen3.py:82, rounded where torch rounds
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gb = rbf(g4[r]), ub = rbf(s[r]);
        const float act = rbf(gb / (1.f + __expf(-gb)));
        o[r] = f2bf(act * ub);
      }
      *reinterpret_cast<bf16x4 *>(a.act + mt * a.act_stride + ((size_t)(n0 >> 3) * 16 + fm) * 8 + (n0 & 7)) = o;
    } else if (EPI == EPI_ARGMAX) {
      const bool live = (fm >= a.row0) && (fm < a.row0 + arg_rows[mt]);
      if (a.logits && live) {
        bf16x4 o = {f2bf(s[0]), f2bf(s[1]), f2bf(s[2]), f2bf(s[3])};
        *reinterpret_cast<bf16x4 *>(a.logits + mt * a.logits_stride + (size_t)fm * a.N + n0) = o;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float vb = rbf(s[r]);  // n ascends with r, with the tile and with the wave's items: first maximum kept
        if (live && (vb > best[mt] || bestn[mt] == 0x7fffffff)) {
          best[mt] = vb;
          bestn[mt] = n0 + r;
        }
      }
    } else if (EPI == EPI_RESID) {  // model/dflash.py:140,144 residual adds (bf16 + bf16 -> bf16)
      bf16_t *hp = a.h_io + mt * a.h_stride + (int64_t)fm * a.ldh + n0;
      const bf16x4 hv = *reinterpret_cast<const bf16x4 *>(hp);
      bf16x4 o;
      float q = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = rbf(s[r]);
        const float hn = a.add_resid ? rbf(bf2f(hv[r]) + v) : v;
        o[r] = f2bf(hn);
        q += hn * hn;
      }
      *reinterpret_cast<bf16x4 *>(hp) = o;
      if (a.tap) *reinterpret_cast<bf16x4 *>(a.tap + mt * a.tap_stride + (int64_t)fm * a.ldtap + n0) = o;
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (fg == 0 && a.ss_out) a.ss_out[mt * a.ss_out_stride + t * 16 + fm] = q;
    }
  };

  // slab of (K part ky, tile t, request mt): 256 floats in D layout, lane l owns 4l..4l+3
  const __amdgpu_buffer_rsrc_t part_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(a.part, 0, a.part ? (int)a.part_bytes : 0, 0x00020000);
  auto slab_off = [&](int k, int t, int mt) -> int { return ((((k * a.ntiles + t) * MT + mt) * 256) + 4 * l) * 4; };

  f32x4 acc[MT];
  auto process = [&](bf16x8(&wr)[FR], int t, int pos) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < FR; ++f)  // k-steps past the wave's share: zero weights and zero activations
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[f], xr[mt][f], acc[mt], 0, 0, 0);
    const int buf = pos & 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4 *>(&red[buf][w][mt][l * 4]) = acc[mt];
    __syncthreads();
    if (w < MT) {  // wave mt finishes request mt's tile
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) s += *reinterpret_cast<const f32x4 *>(&red[buf][ww][w][l * 4]);
      if (EPI == EPI_F32) {
        *reinterpret_cast<f32x4 *>(a.out + ((size_t)(ky * MT + w) * 16 + fm) * a.ldo + t * 16 + 4 * fg) = s;
      } else if (nky > 1 || !INLINE_EPI) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), part_rsrc, slab_off(ky, t, w), 0, 16);  // sc1
      } else if (EPI == EPI_SILU) {
        if ((pos & 1) == 0)
          gate_sum = s;
        else
          epilogue(t, w, s, gate_sum);
      } else {
        epilogue(t, w, s, s);
      }
    }
  };

  // three rotating weight buffers: while one item is in the MFMA, the next two are in flight.  Every request is
  // UNCONDITIONAL (a dummy past the last item) and pinned in front of the MFMAs: behind `if (more) load_item` hipcc
  // cannot count the loads in flight at the join and made the CURRENT item's MFMAs wait vmcnt(0) — for the item just
  // requested — so that only one item (64 KB per CU with 8 waves) was ever in flight.
  for (int j = 0; j < nseq; j += 3) {
    load_item(wC, tile_of(j + 2 < nseq ? j + 2 : j), j + 2 < nseq);
    __builtin_amdgcn_sched_barrier(0);
    process(wA, tile_of(j), j);
    if (j + 1 >= nseq) break;
    __builtin_amdgcn_sched_barrier(0);
    load_item(wA, tile_of(j + 3 < nseq ? j + 3 : j), j + 3 < nseq);
    __builtin_amdgcn_sched_barrier(0);
    process(wB, tile_of(j + 1), j + 1);
    if (j + 2 >= nseq) break;
    __builtin_amdgcn_sched_barrier(0);
    load_item(wB, tile_of(j + 4 < nseq ? j + 4 : j), j + 4 < nseq);
    __builtin_amdgcn_sched_barrier(0);
    process(wC, tile_of(j + 2), j + 2);
  }

  if (EPI != EPI_F32 && (nky > 1 || !INLINE_EPI)) {
    // ---- the K parts of this column group meet: sc1 slabs are drained by every storing wave,
    // then ONE relaxed agent-scope ticket; the last part to arrive combines
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(&a.tickets[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == nky - 1;
      if (last) __hip_atomic_store(&a.tickets[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // items = (unit, request), dealt round-robin to the 8 waves; UB of a wave's items in flight.
    // (Tried: batches sized to the wave's item count and two K parts per round — slower, 16.8 ->
    // 19.5 us on o_proj at 4 tiles; the simple form stays.)
    constexpr int UB = 8;
    // Hand-off of the slabs (MI355X_MICROARCH.md, "Valid forms", table row 1): every slab byte was stored sc1 (write-
    // through), every storing wave drained (vmcnt(0)) before the workgroup barrier in front of its ONE ticket add, the
    // consumer's waves load behind the barrier that follows the returned add — and EVERY load of the handed-off bytes
    // is an sc1 load (L1 bypassed), so no agent-scope acquire (buffer_inv sc1: ~1.7 us per launch, rounds 1-3) is needed.
    const int nitems = (nseq / TPU) * MT;
    for (int it0 = w; it0 < nitems; it0 += NW * UB) {
      f32x4 sv[UB][TPU];
#pragma unroll
      for (int b = 0; b < UB; ++b)
#pragma unroll
        for (int tp = 0; tp < TPU; ++tp) sv[b][tp] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < nky; ++k) {
#pragma unroll
        for (int b = 0; b < UB; ++b) {
          int it = it0 + b * NW;
          it = it < nitems ? it : nitems - 1;
          const int u = it / MT, mt = it - u * MT;
#pragma unroll
          for (int tp = 0; tp < TPU; ++tp)  // sc1 loads (L2-served): see the hand-off note above the combine
            sv[b][tp] += __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(part_rsrc, slab_off(k, tile_of(u * TPU + tp), mt), 0, 16));
        }
      }
#pragma unroll
      for (int b = 0; b < UB; ++b) {
        const int it = it0 + b * NW;
        if (it < nitems) {
          const int u = it / MT, mt = it - u * MT;
          epilogue(tile_of(u * TPU + TPU - 1), mt, sv[b][TPU - 1], sv[b][0]);
        }
      }
    }
  }

  if (EPI == EPI_ARGMAX) {
    // every wave leaves its candidates: in the main loop wave mt saw request mt only, in the
    // combine any wave may have seen any request
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float bv = best[mt];
      int bn = bestn[mt];
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bn, o, 64);
        if (ov > bv || (ov == bv && oi < bn)) {
          bv = ov;
          bn = oi;
        }
      }
      if (fg == 0) {
        a.best_val[(((size_t)blockIdx.x * NW + w) * MT + mt) * 16 + fm] = bv;
        a.best_idx[(((size_t)blockIdx.x * NW + w) * MT + mt) * 16 + fm] = bn;
      }
    }
  }
}

// Cross-workgroup finish of the fused argmax: one wave per (row, request).
__global__ __launch_bounds__(64) void k_argmax_finish_b(const float *best_val, const int *best_idx, int nblk, int MT,
                                                        int row0, int nrows, const int32_t *dyn, int nrows_word,
                                                        int64_t *out_ids, int64_t out_stride, int out_off) {
  const int r = blockIdx.y;
  int rows = nrows;
  if (dyn && nrows_word >= 0) rows = dyn[r * DFL_DYN_WORDS + nrows_word] - row0;
  const int m = row0 + blockIdx.x;
  if ((int)blockIdx.x >= rows) return;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    const float ov = best_val[((size_t)b * MT + r) * 16 + m];
    const int oi = best_idx[((size_t)b * MT + r) * 16 + m];
    if (ov > bv || (ov == bv && oi < bi) || bi == 0x7fffffff) {
      bv = ov;
      bi = oi;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if (threadIdx.x == 0) out_ids[r * out_stride + out_off + blockIdx.x] = (int64_t)bi;
}

// h[r][m] = embed[ids[r][m]] and the rows' sums of squares, grid (16, R)
__global__ __launch_bounds__(256) void k_embed_rows_b(const bf16_t *embed, const int64_t *ids, int64_t ids_stride,
                                                      bf16_t *h, int64_t h_stride, int H, float *ss_out,
                                                      int64_t ss_stride, const int32_t *dyn, int dyn_word) {
  __shared__ float wsum[4];
  const int m = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  const int nv = dyn ? dyn[r * DFL_DYN_WORDS + dyn_word] : 16;
  float ss = 0.f;
  if (m < nv) {
    const bf16_t *src = embed + ids[r * ids_stride + m] * (int64_t)H;
    bf16_t *dst = h + r * h_stride + (int64_t)m * H;
    for (int c = tid; c < (H >> 3); c += 256) {
      const bf16x8 v = *reinterpret_cast<const bf16x8 *>(src + c * 8);
      *reinterpret_cast<bf16x8 *>(dst + c * 8) = v;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += bf2f(v[j]) * bf2f(v[j]);
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) ss_out[r * ss_stride + m] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Residual add + RMSNorm -> frag16 for R requests, grid (16, R):
//   h[r][m] <- part ? bf16(h + bf16(sum_k part[k][r*16+m][:])) : h     (model/dflash.py:140,144)
//   tap[r][m] <- the new h row (optional: a tapped target layer, model/utils.py:16-25)
//   frag[r]  <- norm_w * bf16(h * rsqrt(mean(h^2) + eps))               (tf:modeling_qwen3.py:59-64)
// rows >= dyn valid count: frag zeroed, h untouched.  `part` = the fp32 K-part sums of the
// o_proj / down_proj GEMM that ran just before (dfl_gemm_f32_batch): their K parts meet HERE, at
// the launch boundary — this kernel reads every h row anyway — instead of through slabs + ticket
// inside the GEMM (o_proj 16.8 -> ~12 us, down 33.6 -> ~27 us at 4 tiles; scripts/bench_gemm_batch.py).
// Batched GEMMs take normalised rows from here rather than normalising in their prologue: that
// prologue runs in every workgroup (128x redundant) and at 4 request tiles costs ~10 us of VALU
// per launch (qkv 25.6 us with the normed source vs 15.2 us from frag16).
// Round 3: every load of the row (residual chunks, the K-part sums of all parts) is requested up front, before the
// row-validity word — a dependent scalar load — is looked at; the new row stays in registers across the barrier
// instead of being re-read (6.5 -> see DESIGN.md section 6b us per launch, 82 launches per 4-request cycle).
// MAXC chunks of 8 columns per thread: H <= 2048 * MAXC.
template <int MAXC>
__global__ __launch_bounds__(256) void k_norm_frag_b(bf16_t *h, int64_t h_stride, int64_t ldh, const float *part,
                                                     int nsplit, int64_t part_split, int ldp, bf16_t *tap,
                                                     int64_t ldtap, int64_t tap_stride, const bf16_t *nw, float eps,
                                                     bf16x8 *frag, int64_t frag_stride8, int H, const int32_t *dyn,
                                                     int dyn_word) {
  __shared__ float wsum[4];
  const int m = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  const int nv = dyn ? dyn[r * DFL_DYN_WORDS + dyn_word] : 16;  // (used only after the loads below have been requested)
  const int nchunks = H >> 3;
  bf16x8 *out = frag + r * frag_stride8;
  bf16_t *row = h + r * h_stride + (int64_t)m * ldh;
  const float *prow = part ? part + (int64_t)(r * 16 + m) * ldp : nullptr;
  bf16x8 v[MAXC], wv[MAXC];
  float acc[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256, cc = c < nchunks ? c : nchunks - 1;  // clamped: no branch around a load
    v[i] = *reinterpret_cast<const bf16x8 *>(row + cc * 8);
    wv[i] = *reinterpret_cast<const bf16x8 *>(nw + cc * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  }
  if (prow) {  // uniform
    for (int k = 0; k < nsplit; ++k) {  // fixed part order
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = tid + i * 256, cc = c < nchunks ? c : nchunks - 1;
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(prow + k * part_split + cc * 8);
        const f32x4 p1 = *reinterpret_cast<const f32x4 *>(prow + k * part_split + cc * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] += p0[j];
          acc[i][4 + j] += p1[j];
        }
      }
    }
  }
  if (m >= nv) {  // uniform per workgroup
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = tid; c < nchunks; c += 256) out[c * 16 + m] = z;
    return;
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    if (c < nchunks) {
      if (prow) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = f2bf(bf2f(v[i][j]) + rbf(acc[i][j]));  // Linear output in bf16, then the add
        *reinterpret_cast<bf16x8 *>(row + c * 8) = v[i];
      }
      if (tap) *reinterpret_cast<bf16x8 *>(tap + r * tap_stride + (int64_t)m * ldtap + c * 8) = v[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += bf2f(v[i][j]) * bf2f(v[i][j]);
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  const float rstd = rsqrtf((wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    if (c < nchunks) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wv[i][j]) * rbf(bf2f(v[i][j]) * rstd));
      out[c * 16 + m] = o;
    }
  }
}

constexpr int64_t WS_TICKETS = 1024;                                  // 256 ints
constexpr int64_t WS_CAND = 256 * 16 * 4 * 16;                         // argmax candidates [wg][wave <= 16][req][row]
constexpr int64_t WS_ARGMAX = WS_CAND * (int64_t)(sizeof(float) + sizeof(int));
constexpr int64_t WS_HEAD = WS_TICKETS + WS_ARGMAX;

// DFL_BATCH_GEMM=slab: the round-1..3 form (k_gemm_b: K cut over workgroups, fp32 slabs + ticket) for every batched
// GEMM — A/B measurement and the second implementation the tests compare with.  Default: the ring form (gemm_ring.h)
// where it applies (gate/up, lm_head from fragment sources).
bool use_ring() {
  static const bool v = [] {
    const char *e = getenv("DFL_BATCH_GEMM");
    return !(e && e[0] == 's');
  }();
  return v;
}

// workgroups, units per workgroup and pass, passes: every workgroup walks ceil(nunits / gx) units in passes of <= upp_max
void ring_plan(GemmRArgs &a, int nunits, int upp_max, int &gx) {
  gx = nunits < 256 ? nunits : 256;
  const int per_wg = (nunits + gx - 1) / gx;
  a.nunits = nunits;
  a.npass = (per_wg + upp_max - 1) / upp_max;
  a.upp = (per_wg + a.npass - 1) / a.npass;
}

template <int MT, int TPU, int KQ, int NW, int A, int EPI, int CK = 8>
void launch_ring(const GemmRArgs &a, int gx, hipStream_t st, int gy = 1) {
  constexpr int lds = ring_lds_bytes<MT, TPU, KQ, NW, A, CK>();
  static bool attr_set = false;  // more than 64 KB of dynamic LDS needs the attribute (set once per instantiation)
  void (*kern)(GemmRArgs) = dfl_k_gemm_r<MT, TPU, KQ, NW, A, EPI, CK>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(NW * 64), lds, st, a);
}

// DFL_RING_VARIANT=<n>: A/B knob for the lm_head launch (0: 8-k-step slots, two chunks ahead; 1: 16-k-step slots, one ahead)
int ring_variant() {
  static const int v = [] {
    const char *e = getenv("DFL_RING_VARIANT");
    return e ? atoi(e) : 0;
  }();
  return v;
}

// DFL_RING_ROT=m: workgroup b starts its K walk at chunk (b * m) % chunks.  Default 0 (every workgroup at chunk 0):
// measured on one box, 4 tiles, gate/up 37.3 us at m = 0 against 39.8 / 38.9 / 39.5 at m = 1 / 3 / 5, lm_head 221 - 222 us
// either way (profiles/r4_ring_microbench.txt) — HBM channel camping is not what limits the walk, and workgroups in
// step share each activation chunk's L2 lines.
int ring_rot() {
  static const int v = [] {
    const char *e = getenv("DFL_RING_ROT");
    return e ? atoi(e) : 0;
  }();
  return v;
}

void fill_ring(GemmRArgs &a, const void *wp, const dfl_rows_batch *x, int N, int K, const int32_t *dyn) {
  a.rot_mul = ring_rot();
  a.wp = (const bf16x8 *)wp;
  a.xf = (const bf16x8 *)x->r0.frag;
  a.frag_stride8 = x->frag_stride / 8;
  a.KS = K / 32;
  a.ntiles = N / 16;
  a.dyn = dyn;
}

int batch_ksplit(int K) { return (K / 32 + 63) / 64; }
int mt_of(int R) { return R <= 2 ? 2 : 4; }

bool fill_batch(GemmBArgs &a, const void *wp, const dfl_rows_batch *x, int R, int N, int K, const int32_t *dyn,
                void *ws, const char *who) {
  if (!wp || !x) {
    dfl_set_error("%s: null pointer", who);
    return false;
  }
  if (R < 1 || R > 4 || N <= 0 || K <= 0 || N % 16 || K % 32) {
    dfl_set_error("%s: need 1 <= R <= 4, N%%16==0, K%%32==0 (R=%d N=%d K=%d)", who, R, N, K);
    return false;
  }
  if (batch_ksplit(K) > 16) {
    dfl_set_error("%s: K=%d beyond 16 K parts of 2048", who, K);
    return false;
  }
  if (!fill_src(a.src, &x->r0, K, who)) return false;
  if (x->r0.mode == 2) {  // 128x redundant in every workgroup and ~10 us of VALU at 4 tiles: not offered here
    dfl_set_error("%s: the batched GEMMs take normalised rows from dfl_norm_frag_batch (mode 0), not mode 2", who);
    return false;
  }
  if ((x->r0.valid_word >= 0 && !dyn) || x->frag_stride % 8) {
    dfl_set_error("%s: row validity needs dyn; frag_stride must be a multiple of 8", who);
    return false;
  }
  a.wp = (const bf16x8 *)wp;
  a.frag_stride = x->frag_stride;
  a.rows_stride = x->rows_stride;
  a.ss_stride = x->ss_stride;
  a.dyn = dyn;
  a.KS = K / 32;
  a.ntiles = N / 16;
  const int ksplit = batch_ksplit(K);
  a.nfr = (a.KS + 8 * ksplit - 1) / (8 * ksplit);  // k-steps per wave: 8 waves split a K part
  if (ws) {
    a.tickets = (int *)ws;
    a.best_val = (float *)((char *)ws + WS_TICKETS);
    a.best_idx = (int *)((char *)ws + WS_TICKETS + WS_CAND * sizeof(float));
    a.part = (float *)((char *)ws + WS_HEAD);
    a.part_bytes = (int64_t)ksplit * a.ntiles * mt_of(R) * 256 * sizeof(float);
  }
  return true;
}

template <int EPI>
void launch_b(int R, dim3 grid, hipStream_t st, const GemmBArgs &a) {
  // MT is the compiled tile count: R = 3 runs as 4 with an empty fourth request, R = 1 as 2
  if (R <= 2)
    hipLaunchKernelGGL((k_gemm_b<2, EPI>), grid, dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL((k_gemm_b<4, EPI>), grid, dim3(512), 0, st, a);
}

}  // namespace

extern "C" int dfl_batch_ksplit(int K) { return batch_ksplit(K); }
extern "C" int dfl_batch_tiles(int R) { return mt_of(R); }

extern "C" int64_t dfl_gemm_batch_ws_bytes(int N, int K) {
  return WS_HEAD + (int64_t)batch_ksplit(K) * (N / 16) * 4 * 256 * sizeof(float);
}

extern "C" int dfl_gemm_f32_batch(const void *wp, const dfl_rows_batch *x, int R, int N, int K, float *out,
                                  const int32_t *dyn, void *stream) {
  GemmBArgs a{};
  DFL_REQUIRE(out, "dfl_gemm_f32_batch: null pointer");
  // Ring form with K cut over grid.y (round 4, DFL_RING_F32=1; OFF by default): the same part count and output layout as
  // k_gemm_b (the consumer adds dfl_batch_ksplit(K) parts), but a workgroup's 2048-wide K part of the activations streams
  // through the LDS ring instead of being loaded into registers in front of the first MFMA.  Applies where a workgroup
  // holds >= 6 tiles of a part (bytes in flight = tiles x look-ahead): down_proj (7 tiles x 6 parts).  MEASURED SLOWER
  // there than the register-resident form: 23.8 against 22.7 us per launch, 5.25 against 5.11 ms per 4-request cycle
  // (profiles/r4_batch4_ab.txt) — with a 2048-wide K part resident in registers the workgroup has ALL its activations
  // after one burst and 3 x 64 KB weight items in flight; the ring holds 24 k-steps of look-ahead.  Kept as the measured
  // variant; o_proj in four K parts on the same form was tried as well (12.0 against 11.9 us, and the norm launch
  // behind it reads four parts: 5.23 against 5.16 ms) and removed.
  static const int ring_f32 = [] { const char *e = getenv("DFL_RING_F32"); return e ? atoi(e) : 0; }();
  {
    const int ksplit = batch_ksplit(K), nt = N / 16;
    const int gxm = 256 / ksplit > 0 ? 256 / ksplit : 1;
    const int U = (nt + gxm - 1) / gxm;   // tiles per workgroup and part
    if (ring_f32 && use_ring() && wp && ring_ok(x, K) && R >= 3 && R <= 4 && N % 16 == 0 && x->frag_stride % 8 == 0 && U >= 6 &&
        U <= 8 && ksplit >= 2) {
      GemmRArgs r{};
      fill_ring(r, wp, x, N, K, dyn);
      r.out = out;
      r.ldo = N;
      r.ksp = 8 * ((K / 32 + 8 * ksplit - 1) / (8 * ksplit));   // the k-steps of k_gemm_b's parts: 8 waves x nfr
      r.nunits = nt;
      r.npass = 1;
      r.upp = U;
      const int gx = (nt + U - 1) / U;
      launch_ring<4, 1, 2, 16, 3, EPI_F32>(r, gx, (hipStream_t)stream, ksplit);
      DFL_CHECK_LAUNCH("dfl_gemm_f32_batch");
      return DFL_OK;
    }
  }
  if (!fill_batch(a, wp, x, R, N, K, dyn, nullptr, "dfl_gemm_f32_batch")) return DFL_EINVAL;
  a.out = out;
  a.ldo = N;
  const int ksplit = batch_ksplit(K);
  launch_b<EPI_F32>(R, dim3(grid_x_for(a.ntiles, ksplit), ksplit), (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_gemm_f32_batch");
  return DFL_OK;
}

extern "C" int dfl_gemm_silu_mul_batch(const void *wp_gateup, const dfl_rows_batch *x, int R, int I, int K,
                                       void *act_frag, int64_t act_stride, void *ws, const int32_t *dyn,
                                       void *stream) {
  GemmBArgs a{};
  DFL_REQUIRE(act_frag && ws, "dfl_gemm_silu_mul_batch: null pointer");
  DFL_REQUIRE(act_stride >= 16 * (int64_t)I, "dfl_gemm_silu_mul_batch: act_stride < 16*I");
  if (use_ring() && wp_gateup && ring_ok(x, K) && R >= 1 && R <= 4 && I > 0 && I % 16 == 0 && x->frag_stride % 8 == 0) {
    // ring form: a wave owns a (gate, up) tile pair and a quarter of every K chunk; no K cut over workgroups
    GemmRArgs r{};
    fill_ring(r, wp_gateup, x, 2 * I, K, dyn);
    r.act = (bf16_t *)act_frag;
    r.act_stride = act_stride;
    int gx = 0;
    const int per_wg = (I / 16 + 255) / 256;
    const bool w16 = per_wg % 4 == 0 || per_wg > 6;  // units per pass: 3 (12 waves) or 4 (16 waves)
    ring_plan(r, I / 16, w16 ? 4 : 3, gx);
    if (R <= 2) {
      if (w16)
        launch_ring<2, 2, 4, 16, 3, EPI_SILU>(r, gx, (hipStream_t)stream);
      else
        launch_ring<2, 2, 4, 12, 3, EPI_SILU>(r, gx, (hipStream_t)stream);
    } else {
      if (w16)
        launch_ring<4, 2, 4, 16, 2, EPI_SILU>(r, gx, (hipStream_t)stream);
      else if (ring_variant() & 2)   // A/B: four chunks ahead (160 KB of ring)
        launch_ring<4, 2, 4, 12, 4, EPI_SILU>(r, gx, (hipStream_t)stream);
      else if (ring_variant() & 4)   // A/B: three chunks ahead
        launch_ring<4, 2, 4, 12, 3, EPI_SILU>(r, gx, (hipStream_t)stream);
      else   // two chunks ahead (16 chunks at K = 4096: no padding iterations): in the cycle 35.0 us against 37.3 (three)
        launch_ring<4, 2, 4, 12, 2, EPI_SILU>(r, gx, (hipStream_t)stream);
    }
    DFL_CHECK_LAUNCH("dfl_gemm_silu_mul_batch");
    return DFL_OK;
  }
  if (!fill_batch(a, wp_gateup, x, R, 2 * I, K, dyn, ws, "dfl_gemm_silu_mul_batch")) return DFL_EINVAL;
  a.act = (bf16_t *)act_frag;
  a.act_stride = act_stride;
  const int ksplit = batch_ksplit(K);
  launch_b<EPI_SILU>(R, dim3(grid_x_for(I / 16, ksplit), ksplit), (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_gemm_silu_mul_batch");
  return DFL_OK;
}

extern "C" int dfl_gemm_resid_batch(const void *wp, const dfl_rows_batch *x, int R, int N, int K, void *h_io,
                                    int64_t ldh, int64_t h_stride, int add_residual, void *tap, int64_t ldtap,
                                    int64_t tap_stride, float *ss_out, int64_t ss_stride, void *ws,
                                    const int32_t *dyn, void *stream) {
  GemmBArgs a{};
  DFL_REQUIRE(h_io && ws, "dfl_gemm_resid_batch: null pointer");
  DFL_REQUIRE(ldh >= N && (!tap || ldtap >= N), "dfl_gemm_resid_batch: row strides shorter than N");
  // DFL_RING_RESID=1|2 (A/B knob, default off): the ring form for the small-N projections too — finished sums, no K cut.
  // What bounds it there is the ring itself: bytes of weights in flight = tiles per workgroup x look-ahead k-steps, and the
  // look-ahead is capped by LDS (4 KiB of activations per k-step at 4 tiles): DESIGN.md section 6b.
  static const int ring_resid = [] { const char *e = getenv("DFL_RING_RESID"); return e ? atoi(e) : 0; }();
  if (ring_resid && use_ring() && wp && ring_ok(x, K) && R >= 3 && R <= 4 && !ss_out && N % 16 == 0 && K <= 4096 &&
      x->frag_stride % 8 == 0) {
    GemmRArgs r{};
    fill_ring(r, wp, x, N, K, dyn);
    r.h_io = (bf16_t *)h_io;
    r.ldh = ldh;
    r.h_stride = h_stride;
    r.add_resid = add_residual ? 1 : 0;
    r.tap = (bf16_t *)tap;
    r.ldtap = ldtap;
    r.tap_stride = tap_stride;
    r.nunits = N / 16;
    r.npass = 1;
    if (ring_resid == 1) {  // 3 tiles per workgroup, 4 waves per tile
      r.upp = 3;
      const int gx = (r.nunits + 2) / 3;
      launch_ring<4, 1, 4, 12, 4, EPI_RESID>(r, gx, (hipStream_t)stream);
    } else {                // 2 tiles per workgroup, 8 waves per tile
      r.upp = 2;
      const int gx = (r.nunits + 1) / 2;
      launch_ring<4, 1, 8, 16, 4, EPI_RESID>(r, gx, (hipStream_t)stream);
    }
    DFL_CHECK_LAUNCH("dfl_gemm_resid_batch");
    return DFL_OK;
  }
  if (!fill_batch(a, wp, x, R, N, K, dyn, ws, "dfl_gemm_resid_batch")) return DFL_EINVAL;
  a.h_io = (bf16_t *)h_io;
  a.ldh = ldh;
  a.h_stride = h_stride;
  a.add_resid = add_residual ? 1 : 0;
  a.tap = (bf16_t *)tap;
  a.ldtap = ldtap;
  a.tap_stride = tap_stride;
  a.ss_out = ss_out;
  a.ss_out_stride = ss_stride;
  const int ksplit = batch_ksplit(K);
  launch_b<EPI_RESID>(R, dim3(grid_x_for(a.ntiles, ksplit), ksplit), (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_gemm_resid_batch");
  return DFL_OK;
}

extern "C" int dfl_gemm_argmax_batch(const void *wp, const dfl_rows_batch *x, int R, int V, int K, int row0, int nrows,
                                     const int32_t *dyn, int nrows_dyn_word, void *ws, int64_t *out_ids,
                                     int64_t out_stride, int out_off, void *logits, int64_t logits_stride,
                                     void *stream) {
  GemmBArgs a{};
  DFL_REQUIRE(ws && out_ids, "dfl_gemm_argmax_batch: null pointer");
  DFL_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= 16, "dfl_gemm_argmax_batch: rows [%d,%d) outside the tile", row0,
              row0 + nrows);
  if (use_ring() && wp && ring_ok(x, K) && R >= 1 && R <= 4 && V > 0 && V % 16 == 0 && x->frag_stride % 8 == 0) {
    // ring form: a wave owns a column tile over the whole K, the workgroup walks its tiles in passes of <= 16
    GemmRArgs r{};
    fill_ring(r, wp, x, V, K, dyn);
    r.row0 = row0;
    r.nrows = nrows;
    r.nrows_word = nrows_dyn_word;
    r.best_val = (float *)((char *)ws + WS_TICKETS);
    r.best_idx = (int *)((char *)ws + WS_TICKETS + WS_CAND * sizeof(float));
    r.logits = (bf16_t *)logits;
    r.logits_stride = logits_stride;
    r.N = V;
    int gx = 0;
    ring_plan(r, V / 16, 16, gx);
    if (R <= 2)
      launch_ring<2, 1, 1, 16, 2, EPI_ARGMAX>(r, gx, (hipStream_t)stream);
    else if (ring_variant() & 1)
      launch_ring<4, 1, 1, 16, 1, EPI_ARGMAX, 16>(r, gx, (hipStream_t)stream);
    else
      launch_ring<4, 1, 1, 16, 2, EPI_ARGMAX>(r, gx, (hipStream_t)stream);
    hipLaunchKernelGGL(k_argmax_finish_b, dim3(16, R), dim3(64), 0, (hipStream_t)stream, r.best_val, r.best_idx, gx,
                       mt_of(R), row0, nrows, dyn, nrows_dyn_word, out_ids, out_stride, out_off);
    DFL_CHECK_LAUNCH("dfl_gemm_argmax_batch");
    return DFL_OK;
  }
  if (!fill_batch(a, wp, x, R, V, K, dyn, ws, "dfl_gemm_argmax_batch")) return DFL_EINVAL;
  a.row0 = row0;
  a.nrows = nrows;
  a.nrows_word = nrows_dyn_word;
  a.logits = (bf16_t *)logits;
  a.logits_stride = logits_stride;
  a.N = V;
  const int ksplit = batch_ksplit(K);
  const int gx = grid_x_for(a.ntiles, ksplit);
  launch_b<EPI_ARGMAX>(R, dim3(gx, ksplit), (hipStream_t)stream, a);
  // candidates per (column group, wave): only the finishing workgroup of a group writes them
  hipLaunchKernelGGL(k_argmax_finish_b, dim3(16, R), dim3(64), 0, (hipStream_t)stream, a.best_val, a.best_idx, gx * 8,
                     mt_of(R), row0, nrows, dyn, nrows_dyn_word, out_ids, out_stride, out_off);
  DFL_CHECK_LAUNCH("dfl_gemm_argmax_batch");
  return DFL_OK;
}

extern "C" int dfl_embed_rows_batch(const void *embed, const int64_t *ids, int64_t ids_stride, int R, void *h_out,
                                    int64_t h_stride, int H, float *ss_out, int64_t ss_stride, const int32_t *dyn,
                                    int dyn_word, void *stream) {
  DFL_REQUIRE(embed && ids && h_out && ss_out, "dfl_embed_rows_batch: null pointer");
  DFL_REQUIRE(H > 0 && H % 8 == 0 && R >= 1 && R <= 4, "dfl_embed_rows_batch: H%%8 != 0 or R outside 1..4");
  hipLaunchKernelGGL(k_embed_rows_b, dim3(16, R), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)embed, ids,
                     ids_stride, (bf16_t *)h_out, h_stride, H, ss_out, ss_stride, dyn, dyn_word);
  DFL_CHECK_LAUNCH("dfl_embed_rows_batch");
  return DFL_OK;
}

extern "C" int dfl_norm_frag_batch(void *h, int64_t h_stride, int64_t ldh, int R, const float *part, int nsplit,
                                   int64_t part_split, int ldp, void *tap, int64_t ldtap, int64_t tap_stride,
                                   const void *norm_w, float eps, void *frag, int64_t frag_stride, int H,
                                   const int32_t *dyn, int dyn_word, void *stream) {
  DFL_REQUIRE(h && norm_w && frag, "dfl_norm_frag_batch: null pointer");
  DFL_REQUIRE(H > 0 && H % 8 == 0 && ldh >= H && ldh % 8 == 0 && frag_stride % 8 == 0 && frag_stride >= 16 * (int64_t)H,
              "dfl_norm_frag_batch: bad shape / strides");
  DFL_REQUIRE(R >= 1 && R <= 4, "dfl_norm_frag_batch: R outside 1..4");
  DFL_REQUIRE(!part || (nsplit >= 1 && ldp >= H && ldp % 4 == 0), "dfl_norm_frag_batch: bad partial layout");
  DFL_REQUIRE(!tap || (ldtap >= H && ldtap % 8 == 0), "dfl_norm_frag_batch: bad tap layout");
  DFL_REQUIRE(H <= 2048 * 8, "dfl_norm_frag_batch: H = %d exceeds 16384", H);
#define DFL_NORM_B(MAXC)                                                                                                  \
  hipLaunchKernelGGL((k_norm_frag_b<MAXC>), dim3(16, R), dim3(256), 0, (hipStream_t)stream, (bf16_t *)h, h_stride, ldh,   \
                     part, nsplit, part_split, ldp, (bf16_t *)tap, ldtap, tap_stride, (const bf16_t *)norm_w, eps,       \
                     (bf16x8 *)frag, frag_stride / 8, H, dyn, dyn_word)
  if (H <= 2048 * 2)
    DFL_NORM_B(2);
  else if (H <= 2048 * 4)
    DFL_NORM_B(4);
  else
    DFL_NORM_B(8);
#undef DFL_NORM_B
  DFL_CHECK_LAUNCH("dfl_norm_frag_batch");
  return DFL_OK;
}
