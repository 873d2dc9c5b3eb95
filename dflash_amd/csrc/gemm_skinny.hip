// Skinny (M <= 32 rows) weight-streaming GEMM for gfx950.
//
// Shape of the problem: every nn.Linear of the draft step sees
// at most 16 block rows (+ <=16 context rows), so each weight byte is used for
// <= 32 rows and the kernel is bound by HBM: 3.3 GB of weights per cycle against
// 8 TB/s.  Design:
//   * weights are pre-packed (dfl_pack_weight) so a wave's 16-B-per-lane load is
//     one contiguous 1 KiB run AND one MFMA 16x16x32 A fragment: HBM -> VGPR ->
//     matrix core, no LDS, no shuffles (guide: "GEMV / M<=16: load straight to
//     VGPRs, deep unroll, late vmcnt");
//   * a 1024-thread workgroup = 16 waves splits K 16 ways; each wave keeps its
//     K-slice of the activations (frag16 layout, written by the producer kernel)
//     in registers for the whole launch and walks the workgroup's column tiles,
//     prefetching the next tile's weights while the current one is in the MFMA;
//   * per tile the 16 partial 16x16 tiles meet in LDS (double-buffered, one
//     barrier per tile) and 256 threads finish: fp32 partial store, fused
//     SiLU(gate)*up -> frag16, or the lm_head's running bf16 argmax;
//   * grid.y splits K further when K/32 > 128 steps (fc K=5H, down K=I).
// MFMA utilisation is a few percent by construction; the roofline is HBM.
#include "dfl_common.h"

namespace {

enum { EPI_F32 = 0, EPI_SILU = 1, EPI_ARGMAX = 2 };

struct GemmArgs {
  const bf16x8 *wp;     // packed weights [ntiles][KS][64]
  const bf16x8 *xf[2];  // frag16 activations per row tile [KS][64]
  int KS;               // K / 32
  int ntiles;           // N / 16
  int nfr;              // k-steps per wave (<= FR)
  // EPI_F32
  float *out;           // [ksplit][MT*16][ldo]
  int ldo;
  // EPI_SILU
  bf16_t *act;          // frag16 [I/8][16][8]
  // EPI_ARGMAX
  int row0, nrows;      // rows [row0, row0+nrows) take part
  const int32_t *dyn;
  int nrows_word;
  float *best_val;      // [gridDim.x][16]
  int *best_idx;        // [gridDim.x][16]
  bf16_t *logits;       // optional [16][N]
  int N;
};

__device__ __forceinline__ bf16x8 ld_stream(const bf16x8 *p) { return __builtin_nontemporal_load(p); }

template <int MT, int FR, int EPI>
__global__ __launch_bounds__(1024) void k_gemm(GemmArgs a) {
  // red[buf][wave][mt][256]: lane l owns floats 4l..4l+3 (its MFMA D regs)
  __shared__ float red[2][16][MT][256];

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int ks0 = (blockIdx.y * 16 + w) * a.nfr;
  int nf = a.KS - ks0;
  nf = nf < 0 ? 0 : (nf > a.nfr ? a.nfr : nf);

  bf16x8 xr[MT][FR];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      xr[mt][f] = (f < nf) ? a.xf[mt][(size_t)(ks0 + f) * 64 + l] : z;
    }

  // Tile sequence of this workgroup.  F32/ARGMAX: tiles bx, bx+G, bx+2G, ...
  // SILU: the packed weight interleaves (gate tile p, up tile p), and the sequence
  // walks pairs p = bx, bx+G, ... as gate,up,gate,up so that the finishing thread
  // meets a pair's two sums in consecutive iterations.
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

  // finishing-thread state (threads < 256 only)
  float best = -INFINITY;  // running argmax
  int bestn = 0x7fffffff;
  float gate_sum = 0.f;    // SILU: the pair's gate sum, kept across one iteration
  int arg_rows = 0;
  if (EPI == EPI_ARGMAX) {
    arg_rows = a.nrows;
    if (a.dyn && a.nrows_word >= 0) arg_rows = a.dyn[a.nrows_word] - a.row0;
  }

  auto load_tile = [&](bf16x8(&wr)[FR], int t) {
    const bf16x8 *base = a.wp + ((size_t)t * a.KS + ks0) * 64 + l;
#pragma unroll
    for (int f = 0; f < FR; ++f)
      if (f < nf) wr[f] = ld_stream(base + (size_t)f * 64);
  };

  // `buf` doubles as the position parity in the sequence (0 = gate, 1 = up for SILU)
  auto compute = [&](bf16x8(&wr)[FR], int t, const int buf) {
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < FR; ++f)
      if (f < nf) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[f], xr[mt][f], acc[mt], 0, 0, 0);
      }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4 *>(&red[buf][w][mt][l * 4]) = acc[mt];
    __syncthreads();

    // D layout of mfma 16x16x32 (A = W rows n, B = x^T cols m): lane L reg r ->
    // n_local = 4*(L>>4) + r, m = L & 15.  Thread `oo` finishes float oo of a tile.
    if (tid < MT * 256) {
      const int mt = tid >> 8, oo = tid & 255;
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < 16; ++ww) s += red[buf][ww][mt][oo];
      const int L = oo >> 2, r = oo & 3;
      const int m = L & 15, nl = 4 * (L >> 4) + r;
      if (EPI == EPI_F32) {
        a.out[((size_t)(blockIdx.y * MT + mt) * 16 + m) * a.ldo + t * 16 + nl] = s;
      } else if (EPI == EPI_SILU) {
        if (buf == 0) {
          gate_sum = s;
        } else {
          // tf:...modeling_qwen3.py:82: gate/up Linear outputs are bf16, silu is
          // evaluated in fp32 and rounded, the product is rounded again.
          const float gb = rbf(gate_sum), ub = rbf(s);
          const float act = rbf(gb / (1.f + __expf(-gb)));
          const int n = (t >> 1) * 16 + nl;
          a.act[((size_t)(n >> 3) * 16 + m) * 8 + (n & 7)] = f2bf(act * ub);
        }
      } else {
        const int n = t * 16 + nl;
        const float vb = rbf(s);  // lm_head output is bf16 before argmax (model/dflash.py:238,247)
        const bool live = (m >= a.row0) && (m < a.row0 + arg_rows);
        if (a.logits && live) a.logits[(size_t)m * a.N + n] = f2bf(s);
        // n grows along the sequence for a fixed thread: strict '>' keeps the first maximum
        if (live && (vb > best || bestn == 0x7fffffff)) {
          best = vb;
          bestn = n;
        }
      }
    }
  };

  bf16x8 wA[FR], wB[FR];
  if (nseq > 0) {
    load_tile(wA, tile_of(0));
    for (int j = 0;; j += 2) {
      if (j + 1 < nseq) load_tile(wB, tile_of(j + 1));
      compute(wA, tile_of(j), 0);
      if (j + 1 >= nseq) break;
      if (j + 2 < nseq) load_tile(wA, tile_of(j + 2));
      compute(wB, tile_of(j + 1), 1);
      if (j + 2 >= nseq) break;
    }
  }

  if (EPI == EPI_ARGMAX) {
    // threads 4m+r (+64*wave) of waves 0..3 share row m: reduce 4 lanes, then 4 waves
    __syncthreads();
    float *sv = &red[0][0][0][0];
    int *si = reinterpret_cast<int *>(&red[1][0][0][0]);
    if (tid < 256) {
#pragma unroll
      for (int o = 1; o <= 2; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bestn, o, 64);
        if (ov > best || (ov == best && oi < bestn)) {
          best = ov;
          bestn = oi;
        }
      }
      if ((l & 3) == 0) {
        sv[w * 16 + (l >> 2)] = best;
        si[w * 16 + (l >> 2)] = bestn;
      }
    }
    __syncthreads();
    if (tid < 16) {
      float bv = sv[tid];
      int bi = si[tid];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) {
        const float ov = sv[ww * 16 + tid];
        const int oi = si[ww * 16 + tid];
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      a.best_val[blockIdx.x * 16 + tid] = bv;
      a.best_idx[blockIdx.x * 16 + tid] = bi;
    }
  }
}

// Cross-workgroup finish of the fused argmax: one wave per row.
__global__ __launch_bounds__(64) void k_argmax_finish(const float *best_val, const int *best_idx, int nblk, int row0,
                                                      int nrows, const int32_t *dyn, int nrows_word,
                                                      int64_t *out_ids, int out_off) {
  int rows = nrows;
  if (dyn && nrows_word >= 0) rows = dyn[nrows_word] - row0;
  const int m = row0 + blockIdx.x;
  if ((int)blockIdx.x >= rows) return;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    const float ov = best_val[b * 16 + m];
    const int oi = best_idx[b * 16 + m];
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
  if (threadIdx.x == 0) out_ids[out_off + blockIdx.x] = (int64_t)bi;
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

int pick_ksplit(int KS, int fr_max) { return (KS + 16 * fr_max - 1) / (16 * fr_max); }

}  // namespace

extern "C" int dfl_pack_weight(const void *w, void *wp, int N, int K, void *stream) {
  DFL_REQUIRE(w && wp, "dfl_pack_weight: null pointer");
  DFL_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "dfl_pack_weight: need N%%16==0, K%%32==0 (N=%d K=%d)", N, K);
  const int ntiles = N / 16, KS = K / 32;
  hipLaunchKernelGGL(k_pack_weight, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const bf16x8 *)w, (bf16x8 *)wp,
                     ntiles, KS, 1, 0);
  DFL_CHECK_LAUNCH("dfl_pack_weight");
  return DFL_OK;
}

extern "C" int dfl_pack_weight_gateup(const void *gate, const void *up, void *wp, int I, int K, void *stream) {
  DFL_REQUIRE(gate && up && wp, "dfl_pack_weight_gateup: null pointer");
  DFL_REQUIRE(I > 0 && K > 0 && I % 16 == 0 && K % 32 == 0, "dfl_pack_weight_gateup: need I%%16==0, K%%32==0");
  const int ntiles = I / 16, KS = K / 32;
  hipLaunchKernelGGL(k_pack_weight, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const bf16x8 *)gate,
                     (bf16x8 *)wp, ntiles, KS, 2, 0);
  hipLaunchKernelGGL(k_pack_weight, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const bf16x8 *)up, (bf16x8 *)wp,
                     ntiles, KS, 2, 1);
  DFL_CHECK_LAUNCH("dfl_pack_weight_gateup");
  return DFL_OK;
}

// Workgroups along x for `ngroups` tile groups when the K axis is cut `ksplit` ways: at
// most 256 workgroups in all (one 16-wave workgroup per CU), every workgroup walking
// the same number of groups so that no CU streams twice as long as its neighbour.
static int grid_x_for(int ngroups, int ksplit = 1) {
  int gx_max = 256 / ksplit;
  if (gx_max < 1) gx_max = 1;
  const int per_wg = (ngroups + gx_max - 1) / gx_max;
  return (ngroups + per_wg - 1) / per_wg;
}

extern "C" int dfl_gemm_f32(const void *wp, const void *xf0, const void *xf1, int mt, int N, int K, int ksplit,
                            float *out, void *stream) {
  DFL_REQUIRE(wp && xf0 && out, "dfl_gemm_f32: null pointer");
  DFL_REQUIRE(mt == 1 || (mt == 2 && xf1), "dfl_gemm_f32: mt must be 1 or 2 (with xf1)");
  DFL_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "dfl_gemm_f32: need N%%16==0, K%%32==0 (N=%d K=%d)", N, K);
  const int KS = K / 32;
  const int fr_max = mt == 1 ? 8 : 4;
  DFL_REQUIRE(ksplit >= pick_ksplit(KS, fr_max) && ksplit <= 64, "dfl_gemm_f32: ksplit=%d too small for K=%d (need >= %d)",
              ksplit, K, pick_ksplit(KS, fr_max));
  GemmArgs a{};
  a.wp = (const bf16x8 *)wp;
  a.xf[0] = (const bf16x8 *)xf0;
  a.xf[1] = (const bf16x8 *)xf1;
  a.KS = KS;
  a.ntiles = N / 16;
  a.nfr = (KS + 16 * ksplit - 1) / (16 * ksplit);
  a.out = out;
  a.ldo = N;
  dim3 grid(grid_x_for(a.ntiles, ksplit), ksplit);
  if (mt == 1)
    hipLaunchKernelGGL((k_gemm<1, 8, EPI_F32>), grid, dim3(1024), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((k_gemm<2, 4, EPI_F32>), grid, dim3(1024), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_gemm_f32");
  return DFL_OK;
}

extern "C" int dfl_gemm_silu_mul(const void *wp_gateup, const void *xf, int I, int K, void *act_frag, void *stream) {
  DFL_REQUIRE(wp_gateup && xf && act_frag, "dfl_gemm_silu_mul: null pointer");
  DFL_REQUIRE(I > 0 && K > 0 && I % 16 == 0 && K % 32 == 0, "dfl_gemm_silu_mul: need I%%16==0, K%%32==0");
  const int KS = K / 32;
  DFL_REQUIRE(KS <= 16 * 8, "dfl_gemm_silu_mul: K=%d needs a K split, which the fused activation cannot take", K);
  GemmArgs a{};
  a.wp = (const bf16x8 *)wp_gateup;
  a.xf[0] = (const bf16x8 *)xf;
  a.KS = KS;
  a.ntiles = 2 * (I / 16);
  a.nfr = (KS + 15) / 16;
  a.act = (bf16_t *)act_frag;
  dim3 grid(grid_x_for(I / 16), 1);
  hipLaunchKernelGGL((k_gemm<1, 8, EPI_SILU>), grid, dim3(1024), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_gemm_silu_mul");
  return DFL_OK;
}

extern "C" int64_t dfl_argmax_ws_bytes(void) { return 256 * 16 * (int64_t)(sizeof(float) + sizeof(int)); }

extern "C" int dfl_gemm_argmax(const void *wp, const void *xf, int V, int K, int row0, int nrows, const int32_t *dyn,
                               int nrows_dyn_word, void *ws, int64_t *out_ids, int out_off, void *logits,
                               void *stream) {
  DFL_REQUIRE(wp && xf && ws && out_ids, "dfl_gemm_argmax: null pointer");
  DFL_REQUIRE(V > 0 && K > 0 && V % 16 == 0 && K % 32 == 0, "dfl_gemm_argmax: need V%%16==0, K%%32==0 (V=%d K=%d)", V, K);
  DFL_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= 16, "dfl_gemm_argmax: rows [%d,%d) outside the 16-row tile", row0,
              row0 + nrows);
  const int KS = K / 32;
  DFL_REQUIRE(KS <= 16 * 8, "dfl_gemm_argmax: K=%d exceeds 4096 (argmax needs finished sums)", K);
  GemmArgs a{};
  a.wp = (const bf16x8 *)wp;
  a.xf[0] = (const bf16x8 *)xf;
  a.KS = KS;
  a.ntiles = V / 16;
  a.nfr = (KS + 15) / 16;
  a.row0 = row0;
  a.nrows = nrows;
  a.dyn = dyn;
  a.nrows_word = nrows_dyn_word;
  a.best_val = (float *)ws;
  a.best_idx = (int *)((char *)ws + 256 * 16 * sizeof(float));
  a.logits = (bf16_t *)logits;
  a.N = V;
  const int gx = grid_x_for(a.ntiles);
  hipLaunchKernelGGL((k_gemm<1, 8, EPI_ARGMAX>), dim3(gx, 1), dim3(1024), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(k_argmax_finish, dim3(16), dim3(64), 0, (hipStream_t)stream, a.best_val, a.best_idx, gx, row0,
                     nrows, dyn, nrows_dyn_word, out_ids, out_off);
  DFL_CHECK_LAUNCH("dfl_gemm_argmax");
  return DFL_OK;
}
