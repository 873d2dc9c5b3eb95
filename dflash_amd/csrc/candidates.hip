// Multi-candidate verify, integer / selection side (reference:
// benchmark_candidate_solutions.py).  The candidate builders (:84-414) need, per drafted position, the
// k <= 8 largest draft logits with their token ids and — for the probability-margin gate and the beam
// scores — the row's log-sum-exp; the loop (:570-618) then verifies every candidate block and keeps the one
// with the longest accepted prefix (ties: higher draft score, then lower candidate index).
//   dfl_topk_rows        top-k + logsumexp of bf16 logit rows (replaces torch.topk / softmax / log_softmax
//                        over 15 x V at :98-101, :119-125, :216, :296, :305-307)
//   dfl_candidate_select acceptance length of every candidate, the lexicographic choice of :592-601 and the
//                        commit of :612-613 (+ stop test and length bookkeeping as dfl_accept_commit)
#include "dfl_common.h"

namespace {

// (value desc, index asc): the order torch.argmax uses; torch.topk's order among equal values is an
// implementation detail (bf16 logits tie often), so THIS order is the documented one here.
__device__ __forceinline__ bool better(float v, int i, float ov, int oi) { return v > ov || (v == ov && i < oi); }

// one 256-thread workgroup per row
__global__ __launch_bounds__(256) void k_topk_rows(const bf16_t *__restrict__ logits, int64_t ld, int V, int k,
                                                   float *__restrict__ out_val, int32_t *__restrict__ out_idx,
                                                   float *__restrict__ out_lse) {
  __shared__ float s_v[256][8];
  __shared__ int s_i[256][8];
  __shared__ float s_rv[4];
  __shared__ int s_ri[4], s_rt[4];
  __shared__ float s_m[4], s_s[4];
  const int tid = threadIdx.x, row = blockIdx.x;
  const bf16_t *x = logits + (int64_t)row * ld;
  float v[8];
  int id[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[j] = -INFINITY;
    id[j] = 0x7fffffff;
  }
  float m = -INFINITY, s = 0.f;  // online log-sum-exp
  auto take = [&](float f, int i) {
    if (f > m) {
      s = s * __expf(m - f) + 1.f;
      m = f;
    } else {
      s += (f == -INFINITY) ? 0.f : __expf(f - m);  // a masked (-inf) entry adds nothing; -inf - -inf would be NaN
    }
    if (better(f, i, v[7], id[7])) {
      v[7] = f;
      id[7] = i;
#pragma unroll
      for (int p = 7; p > 0; --p)
        if (better(v[p], id[p], v[p - 1], id[p - 1])) {
          const float tv = v[p];
          v[p] = v[p - 1];
          v[p - 1] = tv;
          const int ti = id[p];
          id[p] = id[p - 1];
          id[p - 1] = ti;
        }
    }
  };
  const int nvec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? V / 8 : 0;
  for (int c = tid; c < nvec; c += 256) {
    const bf16x8 q = *reinterpret_cast<const bf16x8 *>(x + (int64_t)c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) take(bf2f(q[j]), c * 8 + j);
  }
  for (int i = nvec * 8 + tid; i < V; i += 256) take(bf2f(x[i]), i);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s_v[tid][j] = v[j];
    s_i[tid][j] = id[j];
  }
  // ---- log-sum-exp over the 256 partial (m, s)
  {
    float mm = m, ss = s;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mm, o, 64), os = __shfl_xor(ss, o, 64);
      const float nm = fmaxf(mm, om);
      ss = (mm == -INFINITY ? 0.f : ss * __expf(mm - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
      mm = nm;
    }
    if ((tid & 63) == 0) {
      s_m[tid >> 6] = mm;
      s_s[tid >> 6] = ss;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float mm = -INFINITY;
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, s_m[w]);
    float ss = 0.f;
    for (int w = 0; w < 4; ++w) ss += s_m[w] == -INFINITY ? 0.f : s_s[w] * __expf(s_m[w] - mm);
    out_lse[row] = mm + __logf(ss);
  }
  // ---- k rounds: every thread offers the head of its own sorted list, the workgroup takes the best
  int head = 0;
  for (int r = 0; r < k; ++r) {
    float bv = head < 8 ? s_v[tid][head] : -INFINITY;
    int bi = head < 8 ? s_i[tid][head] : 0x7fffffff;
    int bt = tid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64), ot = __shfl_xor(bt, o, 64);
      if (better(ov, oi, bv, bi)) {
        bv = ov;
        bi = oi;
        bt = ot;
      }
    }
    if ((tid & 63) == 0) {
      s_rv[tid >> 6] = bv;
      s_ri[tid >> 6] = bi;
      s_rt[tid >> 6] = bt;
    }
    __syncthreads();
    float gv = s_rv[0];
    int gi = s_ri[0], gt = s_rt[0];
    for (int w = 1; w < 4; ++w)
      if (better(s_rv[w], s_ri[w], gv, gi)) {
        gv = s_rv[w];
        gi = s_ri[w];
        gt = s_rt[w];
      }
    if (tid == gt) ++head;
    if (tid == 0) {
      out_val[row * 8 + r] = gv;
      out_idx[row * 8 + r] = gi;
    }
    __syncthreads();
  }
}

// one wavefront per candidate (<= 8); wave 0 then chooses and commits
__global__ __launch_bounds__(512) void k_candidate_select(const int64_t *blocks, int64_t blk_stride, const int64_t *post,
                                                          int64_t post_stride, const float *scores, int C, int bs,
                                                          int64_t *output_ids, int64_t output_len, int32_t *dyn,
                                                          const int64_t *stop_ids, int n_stop, int32_t *result) {
  __shared__ int s_acc[8];
  const int c = threadIdx.x >> 6, i = threadIdx.x & 63;
  if (c < C) {
    const int64_t *b = blocks + c * blk_stride, *p = post + c * post_stride;
    const bool cmp = i < bs - 1;
    const bool eq = cmp && (b[i + 1] == p[i]);
    const unsigned long long mism = __ballot(cmp && !eq);
    if (i == 0) s_acc[c] = mism ? (int)__builtin_ctzll(mism) : bs - 1;  // :586-588
  }
  __syncthreads();
  if (c != 0) return;
  // lexicographic choice in fp32 exactly as :596-597 writes it: tau * 1e6 + draft_score - idx * 1e-3, first maximum
  int win = 0;
  float best = -INFINITY;
  for (int k = 0; k < C; ++k) {
    const float comp = ((float)(s_acc[k] + 1) * 1e6f + scores[k]) - (float)k * 1e-3f;
    if (comp > best) {
      best = comp;
      win = k;
    }
  }
  const int acc = s_acc[win];
  const int start = dyn[DFL_DYN_START];
  const int64_t *b = blocks + win * blk_stride, *p = post + win * post_stride;
  int64_t tok = -1;
  if (i <= acc)
    tok = b[i];
  else if (i == acc + 1)
    tok = p[acc];
  if (i <= acc + 1 && start + i < output_len) output_ids[start + i] = tok;  // :612-613
  bool hit = false;
  if (i <= acc + 1)
    for (int s = 0; s < n_stop; ++s) hit |= (tok == stop_ids[s]);
  const bool any_stop = __ballot(hit) != 0ull;
  if (i == 0) {
    const int new_start = start + acc + 1;
    dyn[DFL_DYN_S] = start;
    dyn[DFL_DYN_TAU] = acc + 1;
    dyn[DFL_DYN_POS0] = start;
    dyn[DFL_DYN_START] = new_start;
    dyn[DFL_DYN_STOP] |= any_stop ? 1 : 0;
    dyn[DFL_DYN_CYCLE] += 1;
    result[0] = acc;
    result[1] = new_start;
    result[2] = dyn[DFL_DYN_STOP];
    result[3] = win;
    for (int k = 0; k < 8; ++k) result[4 + k] = k < C ? s_acc[k] : -1;
  }
}

}  // namespace

extern "C" int dfl_topk_rows(const void *logits, int64_t ld, int rows, int V, int k, float *out_val, int32_t *out_idx,
                             float *out_lse, void *stream) {
  DFL_REQUIRE(logits && out_val && out_idx && out_lse, "dfl_topk_rows: null pointer");
  DFL_REQUIRE(rows >= 0 && rows <= 64 && V >= 8 && ld >= V && k >= 1 && k <= 8, "dfl_topk_rows: rows=%d V=%d k=%d outside range",
              rows, V, k);
  if (rows == 0) return DFL_OK;
  hipLaunchKernelGGL(k_topk_rows, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)logits, ld, V, k, out_val,
                     out_idx, out_lse);
  DFL_CHECK_LAUNCH("dfl_topk_rows");
  return DFL_OK;
}

extern "C" int dfl_candidate_select(const int64_t *blocks, int64_t blk_stride, const int64_t *posterior,
                                    int64_t post_stride, const float *scores, int n_cand, int bs, int64_t *output_ids,
                                    int64_t output_len, int32_t *dyn, const int64_t *stop_ids, int n_stop, int32_t *result,
                                    void *stream) {
  DFL_REQUIRE(blocks && posterior && scores && output_ids && dyn && result, "dfl_candidate_select: null pointer");
  DFL_REQUIRE(n_cand >= 1 && n_cand <= 8 && bs >= 1 && bs <= 63, "dfl_candidate_select: n_cand=%d bs=%d outside range", n_cand, bs);
  DFL_REQUIRE(n_stop == 0 || stop_ids, "dfl_candidate_select: n_stop > 0 without stop_ids");
  hipLaunchKernelGGL(k_candidate_select, dim3(1), dim3(512), 0, (hipStream_t)stream, blocks, blk_stride, posterior, post_stride,
                     scores, n_cand, bs, output_ids, output_len, dyn, stop_ids, n_stop, result);
  DFL_CHECK_LAUNCH("dfl_candidate_select");
  return DFL_OK;
}
