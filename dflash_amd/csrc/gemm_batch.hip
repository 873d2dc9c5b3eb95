// Skinny GEMM over a ragged batch of requests: R <= 4 requests, one 16-row tile each,
// share ONE pass over the weights (BASELINE.json configs[2]: 4 requests per GPU;
// SURVEY.md §8e "within a GPU the requests are a ragged batch for the kernels").
//
// Same dataflow as gemm_skinny.hip — packed weights HBM -> VGPR -> MFMA, a 16-wave
// workgroup splitting K, the 16 partial tiles meeting in LDS — with one difference that
// the register file forces: the activations of MT tiles over K = 4096 are MT x 128 KB,
// i.e. the whole 512 KB VGPR file of a CU at MT = 4.  A workgroup therefore keeps only a
// 2048-wide K slice of every tile resident (4 k-steps per wave, 64 VGPRs at MT = 4) and
// K is cut over grid.y.  The fused epilogues need finished sums, so the K parts of a
// tile column meet through HBM-resident fp32 partials (9 % of the weight bytes at
// MT = 4) and the LAST workgroup of a column group to arrive — an agent-scope ticket, no
// spinning, nothing to deadlock — sums them in a fixed order and runs the epilogue.
// Request r reads its rows at base + r * stride and its lengths from dyn + 8 r.
#include "gemm_rows.h"

namespace {

struct GemmBArgs {
  const bf16x8 *wp;  // packed weights [ntiles][KS][64]
  RowSrc src;        // request 0; request r at the strides below
  int64_t frag_stride, rows_stride, ss_stride;  // in elements of the respective buffers
  const int32_t *dyn;  // [R][DFL_DYN_WORDS]
  int KS, ntiles, nfr;
  float *part;   // [ksplit][ntiles][MT][256] fp32 partial tiles (fused epilogues, ksplit > 1)
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

template <int MT, int EPI>
__global__ __launch_bounds__(1024) void k_gemm_b(GemmBArgs a) {
  constexpr int FR = 4;
  __shared__ float red[2][16][MT][256];
  __shared__ float ssred[MT][16][16];
  __shared__ int s_last;

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int ky = blockIdx.y, nky = gridDim.y;

  RowSrc src[MT];
  int nv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    src[mt] = a.src;
    if (a.src.frag) src[mt].frag = a.src.frag + (mt * a.frag_stride) / 8;
    if (a.src.rows) src[mt].rows = a.src.rows + mt * a.rows_stride;
    if (a.src.ss) src[mt].ss = a.src.ss + mt * a.ss_stride;
    nv[mt] = (a.src.valid_word >= 0 && a.dyn) ? a.dyn[mt * DFL_DYN_WORDS + a.src.valid_word] : 16;
  }

  const int ks0 = (ky * 16 + w) * a.nfr;
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

  auto load_item = [&](bf16x8(&wr)[FR], int t) {
    const bf16x8 *base = a.wp + ((size_t)t * a.KS + ks0) * 64 + l;
#pragma unroll
    for (int f = 0; f < FR; ++f)
      if (f < nf0) wr[f] = ld_stream(base + (size_t)f * 64);
  };

  bf16x8 wA[FR], wB[FR];
  if (nseq > 0) load_item(wA, tile_of(0));  // the first weights leave for HBM before the prologue

  // ---- (mode 2) rstd of every request's rows: the nss partial sums of squares of a row are
  // summed by the 16 waves together, then exchanged through LDS in a fixed order
  float rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) rstd[mt] = 1.f;
  if (a.src.mode == 2) {
    const int m = l & 15, part = l >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float t = 0.f;
      for (int base = 0; base < a.src.nss; base += 256) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = base + w + 16 * (part + 4 * u);
          v[u] = src[mt].ss[(i < a.src.nss ? i : a.src.nss - 1) * 16 + m];
          v[u] = i < a.src.nss ? v[u] : 0.f;
        }
        t += (v[0] + v[1]) + (v[2] + v[3]);
      }
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (part == 0) ssred[mt][w][m] = t;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < 16; ++ww) t += ssred[mt][ww][l & 15];
      rstd[mt] = rsqrtf(t / (float)(a.KS * 32) + a.src.eps);
    }
  }

  // ---- this wave's K slice of every request's tile stays in registers for the launch
  bf16x8 xr[MT][FR];
  {
    int ks[FR];
    bool take[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      take[f] = f < nf0;
      ks[f] = take[f] ? ks0 + f : 0;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) build_x<FR>(src[mt], ks, take, l, nv[mt], rstd[mt], xr[mt]);
  }

  // finishing thread: request mt = tid >> 8, row m, column nl of the tile
  const int fmt = tid >> 8, ff = tid & 255;
  const int fm = ff >> 4, fnl = ff & 15;
  float best = -INFINITY;
  int bestn = 0x7fffffff;
  float gate_sum = 0.f;
  int arg_rows = 0;
  if (EPI == EPI_ARGMAX && fmt < MT) {
    arg_rows = a.nrows;
    if (a.dyn && a.nrows_word >= 0) arg_rows = a.dyn[fmt * DFL_DYN_WORDS + a.nrows_word] - a.row0;
  }

  // the fused epilogues, on a finished sum s of (request fmt, row fm, column t*16 + fnl);
  // `pos` = position of the tile in this workgroup's sequence
  auto epilogue = [&](int t, int pos, float s) {
    if (EPI == EPI_SILU) {
      if ((pos & 1) == 0) {
        gate_sum = s;
      } else {  // tf:modeling_qwen3.py:82, rounded where torch rounds
        const float gb = rbf(gate_sum), ub = rbf(s);
        const float act = rbf(gb / (1.f + __expf(-gb)));
        const int n = (t >> 1) * 16 + fnl;
        a.act[fmt * a.act_stride + ((size_t)(n >> 3) * 16 + fm) * 8 + (n & 7)] = f2bf(act * ub);
      }
    } else if (EPI == EPI_ARGMAX) {
      const int n = t * 16 + fnl;
      const float vb = rbf(s);
      const bool live = (fm >= a.row0) && (fm < a.row0 + arg_rows);
      if (a.logits && live) a.logits[fmt * a.logits_stride + (size_t)fm * a.N + n] = f2bf(s);
      if (live && (vb > best || bestn == 0x7fffffff)) {  // n ascends along the sequence: first maximum kept
        best = vb;
        bestn = n;
      }
    } else if (EPI == EPI_RESID) {
      const int n = t * 16 + fnl;
      const float v = rbf(s);
      bf16_t *hp = a.h_io + fmt * a.h_stride + (int64_t)fm * a.ldh + n;
      const float hn = a.add_resid ? rbf(bf2f(*hp) + v) : v;
      *hp = f2bf(hn);
      if (a.tap) a.tap[fmt * a.tap_stride + (int64_t)fm * a.ldtap + n] = f2bf(hn);
      const float q = row_sum16(hn * hn);
      if (fnl == 0 && a.ss_out) a.ss_out[fmt * a.ss_out_stride + t * 16 + fm] = q;
    }
  };

  f32x4 acc[MT];
  auto process = [&](bf16x8(&wr)[FR], int t, int pos) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < FR; ++f)
      if (f < nf0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[f], xr[mt][f], acc[mt], 0, 0, 0);
      }
    const int buf = pos & 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4 *>(&red[buf][w][mt][l * 4]) = acc[mt];
    __syncthreads();
    if (tid < MT * 256) {
      const int idx = 4 * (fm + 16 * (fnl >> 2)) + (fnl & 3);  // D layout, see gemm_skinny.hip
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < 16; ++ww) s += red[buf][ww][fmt][idx];
      if (EPI == EPI_F32)
        a.out[((size_t)(ky * MT + fmt) * 16 + fm) * a.ldo + t * 16 + fnl] = s;
      else if (nky == 1)
        epilogue(t, pos, s);
      else
        a.part[(((size_t)ky * a.ntiles + t) * MT + fmt) * 256 + ff] = s;
    }
  };

  for (int j = 0; j < nseq; j += 2) {
    if (j + 1 < nseq) load_item(wB, tile_of(j + 1));
    process(wA, tile_of(j), j);
    if (j + 1 >= nseq) break;
    if (j + 2 < nseq) load_item(wA, tile_of(j + 2));
    process(wB, tile_of(j + 1), j + 1);
  }

  if (EPI != EPI_F32 && nky > 1) {
    // ---- the K parts of this column group meet: last arriver finishes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(&a.tickets[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == nky - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&a.tickets[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid < MT * 256) {
      constexpr int U = 4;  // tiles whose partial loads are in flight together
      for (int j0 = 0; j0 < nseq; j0 += U) {
        float sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          sv[u] = 0.f;
          const int j = j0 + u < nseq ? j0 + u : nseq - 1;
          const float *pp = a.part + ((size_t)tile_of(j) * MT + fmt) * 256 + ff;
          for (int k = 0; k < nky; ++k) sv[u] += __builtin_nontemporal_load(pp + (size_t)k * a.ntiles * MT * 256);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (j0 + u < nseq) epilogue(tile_of(j0 + u), j0 + u, sv[u]);
      }
    }
  }

  if (EPI == EPI_ARGMAX) {
    if (tid < MT * 256) {
#pragma unroll
      for (int o = 1; o <= 8; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bestn, o, 64);
        if (ov > best || (ov == best && oi < bestn)) {
          best = ov;
          bestn = oi;
        }
      }
      if (fnl == 0) {
        a.best_val[((size_t)blockIdx.x * MT + fmt) * 16 + fm] = best;
        a.best_idx[((size_t)blockIdx.x * MT + fmt) * 16 + fm] = bestn;
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

constexpr int64_t WS_TICKETS = 1024;                                  // 256 ints
constexpr int64_t WS_ARGMAX = 256 * 4 * 16 * (int64_t)(sizeof(float) + sizeof(int));  // best_val + best_idx
constexpr int64_t WS_HEAD = WS_TICKETS + WS_ARGMAX;

int batch_ksplit(int K) { return (K / 32 + 63) / 64; }

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
  a.nfr = (a.KS + 16 * ksplit - 1) / (16 * ksplit);
  if (ws) {
    a.tickets = (int *)ws;
    a.best_val = (float *)((char *)ws + WS_TICKETS);
    a.best_idx = (int *)((char *)ws + WS_TICKETS + 256 * 4 * 16 * sizeof(float));
    a.part = (float *)((char *)ws + WS_HEAD);
  }
  return true;
}

template <int EPI>
void launch_b(int R, dim3 grid, hipStream_t st, const GemmBArgs &a) {
  // MT is the compiled tile count: R = 3 runs as 4 with an empty fourth request, R = 1 as 2
  if (R <= 2)
    hipLaunchKernelGGL((k_gemm_b<2, EPI>), grid, dim3(1024), 0, st, a);
  else
    hipLaunchKernelGGL((k_gemm_b<4, EPI>), grid, dim3(1024), 0, st, a);
}
int mt_of(int R) { return R <= 2 ? 2 : 4; }

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
  // only the finishing workgroup of a column group writes best_*: every x index has exactly one
  hipLaunchKernelGGL(k_argmax_finish_b, dim3(16, R), dim3(64), 0, (hipStream_t)stream, a.best_val, a.best_idx, gx,
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
