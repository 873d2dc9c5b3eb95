// Pieces shared by the skinny GEMM kernels (gemm_skinny.hip: one request;
// gemm_batch.hip: a ragged batch of requests over one weight stream).
#pragma once
#include "dfl_common.h"

namespace {

enum { EPI_F32 = 0, EPI_SILU = 1, EPI_ARGMAX = 2, EPI_RESID = 3, EPI_SILU_E = 4 /* SILU over a list of active experts */ };

struct RowSrc {
  const bf16x8 *frag;  // mode 0: frag16 [KS][64]
  const bf16_t *rows;  // mode 1/2: row-major [16][K], row stride ld
  int64_t ld;
  const float *ss;     // mode 2: [nss][16] partial sums of squares per row
  int nss;
  const bf16_t *nw;    // mode 2: RMSNorm weight [K]
  float eps;
  int valid_word;      // dyn word with the number of valid rows, < 0: all 16
  int mode;
};

__device__ __forceinline__ bf16x8 ld_stream(const bf16x8 *p) { return __builtin_nontemporal_load(p); }

// Up to NF consecutive k-steps (1 KiB each, wave-wide) of a packed column tile, starting at
// `first` (= tile base + ks0 * 64), as buffer loads: the descriptor (first, nf KiB) sits in
// SGPRs, the lane offset is one constant VGPR, k-step offsets go to the immediate / scalar
// offset fields — no 64-bit VGPR address per fragment.  A k-step >= nf (past the wave's share
// or past K) is outside the descriptor's range: it returns ZERO and costs no memory traffic,
// so neither the loads nor the MFMAs that consume them need a guard (guards around loads make
// hipcc wait vmcnt(0) per fragment; zero weights add zero).
// `half` (0 / 1; -1 = the whole tile): only the lanes of columns 8*half .. 8*half+7 of the 16-column tile load (lane l
// holds column l & 15); the others get an offset past the descriptor, i.e. zeros and no traffic — branch-free.
template <int NF, int AUX = 2 /* nt: streamed once; 0 for fragments many workgroups re-read from L2 */>
__device__ __forceinline__ void load_ksteps(bf16x8 (&wr)[NF], const bf16x8 *first, int nf, int l, int half = -1) {
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(first), 0, (nf < 0 ? 0 : nf) * 1024, 0x00020000);
  const int voff = (half < 0 || ((l >> 3) & 1) == half) ? l * 16 : 0x40000000;
#pragma unroll
  for (int f = 0; f < NF; ++f)
    wr[f] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff + (f & 3) * 1024, (f >> 2) * 4096, AUX));
}

// sum over the 16 lanes of a DPP row, result in each of them
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

// B-operand fragments of k-steps ks[0..NF) for lane l (m = l&15, kq = l>>4):
// x[m][ks*32 + kq*8 .. +8], in two steps so that the caller can put its weight loads BETWEEN
// them: a wave's vector loads return in issue order (vmcnt), so activation loads issued after
// a weight burst are held back until the burst has landed — and the rstd / normalise chain
// behind them with it (scripts/dbg_gemm_stamps.py: the prologue then ran AFTER the first
// burst with HBM idle, 4-5 us per launch).  issue_x: the loads (raw rows or fragments, norm
// weights); finish_x: mask / normalise.  `take[f]` false -> zero fragment (k-step past the
// wave's share or past K; ks[f] is then any valid step).  The source mode is switched
// OUTSIDE the fragment loops so that the loads of one call are issued together; branching
// per fragment serialised them with a vmcnt(0) each (+6.5 us per launch, measured).
template <int NF, bool NORM = true>
__device__ __forceinline__ void issue_x(const RowSrc &s, const int (&ks)[NF], int l, int nv, bf16x8 (&raw)[NF],
                                        bf16x8 (&wv)[NF]) {
  // one branch-free load sequence for all modes (a mode branch around the loads made hipcc keep
  // the fragment arrays in scratch memory): per-lane base + k-step stride, both by select
  // All 16 rows are loaded, valid or not (the caller's buffer covers 16 rows, dflash_hip.h):
  // no address depends on `nv`, so nothing here waits for the scalar load of the lengths.
  (void)nv;
  const int m = l & 15, kq = l >> 4;
  const bool fr = s.mode == 0;
  const char *base = fr ? reinterpret_cast<const char *>(s.frag) + l * 16
                        : reinterpret_cast<const char *>(s.rows) + ((int64_t)m * s.ld + kq * 8) * 2;
  const int kstep = fr ? 1024 : 64;  // bytes per k-step: a 64-lane fragment / 32 bf16 of a row
  // norm weights: only mode 2 has them; the others read (and ignore) 16 B of their own operand
  const char *nwb = s.mode == 2 ? reinterpret_cast<const char *>(s.nw) + kq * 16 : base;
  const int nstep = s.mode == 2 ? 64 : kstep;
#pragma unroll
  for (int f = 0; f < NF; ++f) raw[f] = *reinterpret_cast<const bf16x8 *>(base + (int64_t)ks[f] * kstep);
  if (NORM) {  // compile-time: kernels that never see mode 2 skip these loads
#pragma unroll
    for (int f = 0; f < NF; ++f) wv[f] = *reinterpret_cast<const bf16x8 *>(nwb + (int64_t)ks[f] * nstep);
  }
}

template <int NF, bool NORM = true>
__device__ __forceinline__ void finish_x(const RowSrc &s, const bool (&take)[NF], int l, int nv, float rstd,
                                         const bf16x8 (&raw)[NF], const bf16x8 (&wv)[NF], bf16x8 (&x)[NF]) {
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  if (s.mode == 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) x[f] = take[f] ? raw[f] : z;
    return;
  }
  const int m = l & 15;
  const bool norm = NORM && s.mode == 2;  // Qwen3RMSNorm: weight * bf16(x * rstd), tf:modeling_qwen3.py:59-64
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const bf16x8 r = raw[f];  // whole-vector copies: element access through the array
    bf16x8 o = r;             // reference kept the arrays in scratch memory
    if (norm) {
      const bf16x8 wq = wv[f];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wq[j]) * rbf(bf2f(r[j]) * rstd));
    }
    x[f] = (take[f] && m < nv) ? o : z;
  }
}

// finish_x with the RMSNorm weight read from LDS (nwl[c] = elements 8c .. 8c+7, staged once per workgroup by the
// caller, K <= 4096): no second 32-VGPR fragment array is alive while the activation loads are in flight.
template <int NF>
__device__ __forceinline__ void finish_xl(const RowSrc &s, const int (&ks)[NF], const bool (&take)[NF], int l, int nv,
                                          float rstd, const bf16x8 (&raw)[NF], const bf16x8 *nwl, bf16x8 (&x)[NF]) {
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  if (s.mode == 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) x[f] = take[f] ? raw[f] : z;
    return;
  }
  const int m = l & 15, kq = l >> 4;
  const bool norm = s.mode == 2;  // Qwen3RMSNorm: weight * bf16(x * rstd), tf:modeling_qwen3.py:59-64
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const bf16x8 r = raw[f];
    bf16x8 o = r;
    if (norm) {
      const bf16x8 wq = nwl[(ks[f] * 4 + kq) & 511];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wq[j]) * rbf(bf2f(r[j]) * rstd));
    }
    x[f] = (take[f] && m < nv) ? o : z;
  }
}

template <int NF, bool NORM = true>
__device__ __forceinline__ void build_x(const RowSrc &s, const int (&ks)[NF], const bool (&take)[NF], int l, int nv,
                                        float rstd, bf16x8 (&x)[NF]) {
  bf16x8 raw[NF], wv[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) wv[f] = raw[f] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  issue_x<NF, NORM>(s, ks, l, nv, raw, wv);
  finish_x<NF, NORM>(s, take, l, nv, rstd, raw, wv, x);
}

// ---- host side ----
inline int pick_ksplit_min(int KS, int fr_max) { return (KS + 16 * fr_max - 1) / (16 * fr_max); }

// Workgroups along x for `ngroups` tile groups when the K axis is cut `ksplit` ways: at
// most 256 workgroups in all (one 16-wave workgroup per CU), every workgroup walking the
// same number of groups so that no CU streams twice as long as its neighbour.
inline int grid_x_for(int ngroups, int ksplit = 1) {
  int gx_max = 256 / ksplit;
  if (gx_max < 1) gx_max = 1;
  const int per_wg = (ngroups + gx_max - 1) / gx_max;
  return (ngroups + per_wg - 1) / per_wg;
}

inline bool fill_src(RowSrc &d, const dfl_rows *s, int K, const char *who) {
  if (!s) {
    dfl_set_error("%s: null row source", who);
    return false;
  }
  d.frag = (const bf16x8 *)s->frag;
  d.rows = (const bf16_t *)s->rows;
  d.ld = s->ld;
  d.ss = s->ss;
  d.nss = s->nss;
  d.nw = (const bf16_t *)s->norm_w;
  d.eps = s->eps;
  d.valid_word = s->valid_word;
  d.mode = s->mode;

  const bool ok = (s->mode == 0 && s->frag) || (s->mode == 1 && s->rows && s->ld >= K && s->ld % 8 == 0) ||
                  (s->mode == 2 && s->rows && s->ss && s->nss >= 1 && s->norm_w && s->ld >= K && s->ld % 8 == 0);
  if (!ok) dfl_set_error("%s: bad row source (mode %d)", who, s->mode);
  if (ok && s->mode == 2 && K > 4096) {  // the norm weight is staged in 8 KB of LDS; the in-kernel rstd covers one K pass
    dfl_set_error("%s: a normalised row source needs K <= 4096 (K=%d)", who, K);
    return false;
  }
  return ok;
}


}  // namespace
