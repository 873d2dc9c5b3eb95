// Routing arithmetic of Qwen3MoeTopKRouter.forward (tf:models/qwen3_moe/modeling_qwen3_moe.py): softmax over the gate
// Linear's bf16 logits in fp32, top-k by (probability descending, expert index ascending), optional renormalisation.
// One wavefront per row, E <= 256 (lane l owns experts l, l + 64, l + 128, l + 192).  Shared by the decode-side kernel
// (moe.hip: k_moe_route, the <= 16 rows of a verify block) and the prefill-side one (prefill.hip: k_pmoe_route).
#pragma once
#include "dfl_common.h"

// 64-lane max / min, result in every lane: the same DPP steps (round 4: the LDS-routed butterfly they replace cost the
// MoE router's top-k 96 dependent ~100-cycle steps per row, ~4 us of an 8.7 us launch).  Exact either way.
__device__ __forceinline__ float route_wave_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  const int iv = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48))));
}
__device__ __forceinline__ int route_wave_min(int v) {
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

__device__ __forceinline__ bool route_better(float v, int i, float ov, int oi) { return v > ov || (v == ov && i < oi); }

// logits_row: the row's E bf16 logits.  On return every lane holds the k selected experts (sel_i, in selection order),
// their softmax probabilities (sel_v) and the sum of those (tot).
__device__ __forceinline__ void route_row(const bf16_t *logits_row, int E, int top_k, int l, float (&sel_v)[8],
                                          int (&sel_i)[8], float &tot) {
  float p[4];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = l + 64 * j;
    p[j] = e < E ? bf2f(logits_row[e]) : -INFINITY;
    mx = fmaxf(mx, p[j]);
  }
  mx = route_wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[j] = (l + 64 * j) < E ? __expf(p[j] - mx) : 0.f;
    sum += p[j];
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = p[j] / sum;  // softmax(dtype = float)
  tot = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {   // unrolled with a guard: a runtime index would send sel_v / sel_i to scratch memory
    sel_v[r] = 0.f;
    sel_i[r] = 0;
    if (r >= top_k) continue;
    float bv = -1.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = l + 64 * j;
      if (e < E && route_better(p[j], e, bv, bi)) {
        bv = p[j];
        bi = e;
      }
    }
    // the wave's best: the largest probability, and among the lanes that hold it the lowest expert (two DPP
    // reductions; a NaN row leaves bv = -1 / bi = INT_MAX in every lane, handled below)
    const float mv = route_wave_max(bv);
    bi = route_wave_min(bv == mv ? bi : 0x7fffffff);
    bv = mv;
    if (bi == 0x7fffffff) {  // a NaN row (one NaN logit makes every probability NaN): no comparison succeeded.  The index
      bi = r;                // feeds LDS / global counters downstream (k_pmoe_plan, k_moe_route), so it must stay < E: round
      bv = p[0] + p[1];      // r takes expert r (distinct per round, top_k <= E) with a NaN weight — the NaN propagates
    }                        // through the expert sums as it does through torch.topk + index_add_, nothing faults
    sel_v[r] = bv;
    sel_i[r] = bi;
    tot += bv;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (l + 64 * j == bi) p[j] = -1.f;  // taken
  }
}
