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
// k_pgemm: 256 threads = 4 waves (2 x 2), block tile 128 columns x 128 rows, 64-deep K steps, two LDS stages (64 KiB),
// each wave 4 x 4 MFMA tiles (64 accumulator VGPRs); the 8 row blocks of a column block share blockIdx % 8, i.e. an XCD
// and its L2, so the weights leave HBM once.  Epilogues: bf16 rows / residual add (+ tap copy) / SiLU(gate) * up -> frag16.
#include "gemm_rows.h"

namespace {

enum { PEPI_ROWS = 0, PEPI_RESID = 1, PEPI_SILU = 2 };

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
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// MW = row tiles per wave: 4 -> block tile 128 columns x 128 rows; 2 -> 128 x 64 (twice the workgroups: the N = 4096
// projections o_proj / down_proj have only 32 column blocks, and ONE 4-wave workgroup per CU leaves the matrix pipe idle
// whenever it waits for its own LDS-DMA: 115 -> see DESIGN.md for the measured step)
template <int EPI, int MW>
__global__ __launch_bounds__(256) void k_pgemm(PGemmArgs a) {
  constexpr int MB = 2 * MW;  // row tiles per block
  __shared__ bf16x8 lds[2][16 + 2 * MB][64];  // [stage][slot: 0..15 = W (n tile, k-step), 16.. = X (m tile, k-step)][lane]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nbx = a.ntiles >> 3, nby = a.mtiles / MB;
  // column block / row block of this workgroup: the nby row blocks of a column block are 8 apart in blockIdx
  int nb, mb;
  {
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
  }
  const int wn = w & 1, wm = w >> 1;  // the wave's 64 columns x 64 rows inside the block tile
  const int KT = a.KS >> 1;           // 64-deep steps

  // stage kt -> buffer: 16 weight fragments + 2 MB activation fragments of 1 KiB, dealt round-robin to the 4 waves
  auto stage = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < (16 + 2 * MB) / 4; ++i) {
      const int slot = i * 4 + w;  // wave-uniform
      const int s16 = slot < 16 ? slot : slot - 16;  // fragment of its operand: (tile s16 >> 1, k-step s16 & 1)
      const bf16x8 *src = (slot < 16 ? a.wp + ((size_t)(nb * 8 + (s16 >> 1)) * a.KS + kt * 2 + (s16 & 1)) * 64
                                     : a.xf + ((size_t)(mb * MB + (s16 >> 1)) * a.KS + kt * 2 + (s16 & 1)) * 64) + l;
      __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)&lds[buf][slot][0], 16, 0, 0);
    }
  };

  f32x4 acc[4][MW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage kt has landed in LDS
    __syncthreads();  // ... and everyone's; every wave has finished reading buffer buf ^ 1 (iteration kt - 1)
    if (kt + 1 < KT) stage(kt + 1, buf ^ 1);
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
  }

  // D layout (A = W rows n, B = x^T columns m): lane L, register r = column 4 (L >> 4) + r of the n tile, row L & 15
  const int fm = l & 15, fg = l >> 4;
#pragma unroll
  for (int j = 0; j < MW; ++j) {
    const int mt = mb * MB + wm * MW + j;
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

// rows [P][H] -> (RMSNorm) -> frag16 row tiles.  One WAVE per row (grid = row tiles x 4, four rows per workgroup): the
// row's chunks stay in registers between the sum of squares and the normalisation (one pass over memory; the first
// form, one workgroup per 16-row tile with 16 threads per row, took 22 us for 1024 x 4096: 64 workgroups, two passes).
// norm_w == nullptr: pack only.  Rows >= P give zero fragments.  H <= 64 * 8 * PN_MAXC.
constexpr int PN_MAXC = 8;
__global__ __launch_bounds__(256) void k_pnorm_pack(const bf16_t *h, int64_t ldh, int P, int H, const bf16_t *norm_w,
                                                    float eps, bf16x8 *xf) {
  const int mt = blockIdx.x, l = threadIdx.x & 63;
  const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int row = mt * 16 + m;
  const int KS = H >> 5, nch = H >> 3;
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
  bf16x8 *dst = xf + (size_t)mt * KS * 64;
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
  if ((a.ntiles / 8) * (a.mtiles / 8) >= 512)
    hipLaunchKernelGGL((k_pgemm<EPI, 4>), dim3((a.ntiles / 8) * (a.mtiles / 8)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_pgemm<EPI, 2>), dim3((a.ntiles / 8) * (a.mtiles / 4)), dim3(256), 0, st, a);
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
                     (const bf16_t *)norm_w, eps, (bf16x8 *)x_frag);
  DFL_CHECK_LAUNCH("dfl_prefill_norm_pack");
  return DFL_OK;
}

extern "C" int dfl_prefill_qk_rope(void *qkv_rows, int64_t ld, int P, int q_col, int k_col, int v_col, int n_q, int n_kv,
                                   const void *q_norm_w, const void *k_norm_w, float eps, const void *cos_tab,
                                   const void *sin_tab, int max_pos, int pos0, void *kcache, void *vcache,
                                   int cache_rows, int row0, void *stream) {
  DFL_REQUIRE(qkv_rows && cos_tab && sin_tab && kcache && vcache, "dfl_prefill_qk_rope: null pointer");
  DFL_REQUIRE(P >= 1 && n_q >= 1 && n_kv >= 1 && max_pos >= 1 && pos0 >= 0 && row0 >= 0 && row0 + P <= cache_rows,
              "dfl_prefill_qk_rope: bad lengths (P=%d row0=%d cache_rows=%d)", P, row0, cache_rows);
  PRopeArgs a{(bf16_t *)qkv_rows, ld, P, q_col, k_col, v_col, n_q, n_kv, (const bf16_t *)q_norm_w,
              (const bf16_t *)k_norm_w, eps, (const bf16_t *)cos_tab, (const bf16_t *)sin_tab, max_pos, pos0,
              (bf16_t *)kcache, (bf16_t *)vcache, cache_rows, row0};
  const int64_t items = (int64_t)P * (n_q + 2 * n_kv);
  hipLaunchKernelGGL(k_pqk_rope, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_prefill_qk_rope");
  return DFL_OK;
}
