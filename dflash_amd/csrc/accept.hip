// Integer side of the hot path: greedy argmax over materialised logits (the
// target's posterior at T=0, model/utils.py:28-29) and the acceptance scan with
// commit, bonus token, stop test and rollback bookkeeping (model/dflash.py:258-268).
#include "dfl_common.h"

namespace {

// One 256-thread block per row.  torch.argmax returns the FIRST maximal index, so
// the reduction is on (value desc, index asc).  Values are compared as fp32 (exact
// for bf16 inputs).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_argmax(const T *__restrict__ logits, int64_t V, int64_t *__restrict__ ids) {
  __shared__ float sv[4];
  __shared__ int64_t si[4];
  const T *row = logits + (int64_t)blockIdx.x * V;
  float best = -INFINITY;
  int64_t bi = INT64_MAX;
  auto take = [&](float v, int64_t i) {
    if (v > best || (v == best && i < bi) || (bi == INT64_MAX)) {
      best = v;
      bi = i;
    }
  };
  const int64_t nvec = ((reinterpret_cast<uintptr_t>(row) % (VEC * sizeof(T))) == 0) ? V / VEC : 0;
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  for (int64_t c = threadIdx.x; c < nvec; c += 256) {
    const vec_t v = *reinterpret_cast<const vec_t *>(row + c * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float f = (float)v[j];
      if (f > best || bi == INT64_MAX) {  // indices ascend within a thread
        best = f;
        bi = c * VEC + j;
      }
    }
  }
  for (int64_t i = nvec * VEC + threadIdx.x; i < V; i += 256) take((float)row[i], i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int64_t oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = best;
    si[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
        best = sv[w];
        bi = si[w];
      }
    ids[blockIdx.x] = bi;
  }
}

// Single wavefront.  Lane i compares draft token i+1 with the target's token for
// the same position; the accepted length is the number of trailing-zero-free ones
// of the ballot (first mismatch by ffs), no host round trip.
// next_block (optional): the NEXT cycle's block, re-armed here so that the host need not copy it out of output_ids
// (model/dflash.py:235: block = output_ids[:, start:start+bs] = the bonus token followed by mask ids) — rearm_n slots.
__global__ __launch_bounds__(64) void k_accept_commit(const int64_t *block_ids, const int64_t *posterior, int bs,
                                                      int64_t *output_ids, int64_t output_len, int32_t *dyn,
                                                      const int64_t *stop_ids, int n_stop, int32_t *result,
                                                      int64_t *next_block, int rearm_n, int64_t mask_id, int32_t *dyn_t) {
  const int i = threadIdx.x;
  const int start = dyn[DFL_DYN_START];
  const bool cmp = i < bs - 1;
  const bool eq = cmp && (block_ids[i + 1] == posterior[i]);
  const unsigned long long mism = __ballot(cmp && !eq);
  int acc = mism ? (int)__builtin_ctzll(mism) : bs - 1;  // leading matches, 0..bs-1
  // commit block[0..acc] at start.., then the target's own token (model/dflash.py:259-260)
  int64_t tok = -1;
  if (i <= acc)
    tok = block_ids[i];
  else if (i == acc + 1)
    tok = posterior[acc];
  if (i <= acc + 1 && start + i < output_len) output_ids[start + i] = tok;
  bool hit = false;
  if (i <= acc + 1)
    for (int s = 0; s < n_stop; ++s) hit |= (tok == stop_ids[s]);
  const bool any_stop = __ballot(hit) != 0ull;
  if (next_block && i < rearm_n) {  // every compared id was read before the ballot above: in-place re-arm is safe
    const int64_t bonus = posterior[acc];
    next_block[i] = i == 0 ? bonus : mask_id;
  }
  if (i == 0) {
    const int new_start = start + acc + 1;
    dyn[DFL_DYN_S] = start;         // draft cache keeps rows [0, start): crop(start), :246
    dyn[DFL_DYN_TAU] = acc + 1;     // next cycle's context rows, :263
    dyn[DFL_DYN_POS0] = start;
    dyn[DFL_DYN_START] = new_start; // :261
    dyn[DFL_DYN_STOP] |= any_stop ? 1 : 0;
    dyn[DFL_DYN_CYCLE] += 1;
    if (dyn_t) {  // the block-form record of the NEXT verify (its block size word stays): rows kept = positions = new start
      dyn_t[DFL_DYN_S] = new_start;
      dyn_t[DFL_DYN_TAU] = 0;
      dyn_t[DFL_DYN_POS0] = new_start;
      dyn_t[DFL_DYN_START] = new_start;
    }
    if (result) {
      // Hand-over a CPU thread may be polling (pinned host memory): the payload words first, then — behind a
      // system-scope release, so that no store can overtake them — the cycle counter as the LAST word.  The host
      // polls word 3 and reads words 0..2 only after it has changed (generate.py); no reliance on a 16-byte store
      // arriving whole.
      result[0] = acc;
      result[1] = new_start;
      result[2] = dyn[DFL_DYN_STOP];
      __hip_atomic_store(&result[3], dyn[DFL_DYN_CYCLE], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Ragged batch: one wavefront per request (grid.x = request).  Besides the draft-form
// lengths dyn_d (S = rows kept in the draft cache, tau = context rows of the next cycle) it
// maintains the block-form lengths dyn_t = {S = new start, tau = 0, pos0 = new start} which
// the target verify AND the draft's block stage read, so that a steady-state cycle needs no
// host-side length update at all.
__global__ __launch_bounds__(64) void k_accept_commit_b(const int64_t *block_ids, int64_t blk_stride,
                                                        const int64_t *posterior, int64_t post_stride,
                                                        int64_t *output_ids, int64_t out_stride, int64_t output_len,
                                                        int32_t *dyn_d, int32_t *dyn_t, const int64_t *stop_ids,
                                                        int n_stop, int32_t *result, int64_t *next_block,
                                                        int64_t mask_id, int rearm_n, int tiles_per_req, int32_t *dyn_dt,
                                                        int32_t *dyn_tt) {
  const int r = blockIdx.x, i = threadIdx.x;
  if (next_block) next_block += r * blk_stride;
  block_ids += r * blk_stride;
  posterior += r * post_stride;
  output_ids += r * out_stride;
  dyn_d += r * DFL_DYN_WORDS;
  dyn_t += r * DFL_DYN_WORDS;
  const int start = dyn_d[DFL_DYN_START];
  const int bs = dyn_d[DFL_DYN_BS];
  const bool cmp = i < bs - 1;
  const bool eq = cmp && (block_ids[i + 1] == posterior[i]);
  const unsigned long long mism = __ballot(cmp && !eq);
  const int acc = mism ? (int)__builtin_ctzll(mism) : bs - 1;
  int64_t tok = -1;
  if (i <= acc)
    tok = block_ids[i];
  else if (i == acc + 1)
    tok = posterior[acc];
  if (bs > 0 && i <= acc + 1 && start + i < output_len) output_ids[start + i] = tok;
  bool hit = false;
  if (bs > 0 && i <= acc + 1)
    for (int s = 0; s < n_stop; ++s) hit |= (tok == stop_ids[s]);
  const bool any_stop = __ballot(hit) != 0ull;
  // the next cycle's block = output_ids[new start .. +bs) = [bonus token, mask, mask, ...]
  // (model/dflash.py:235); all reads of block_ids above precede these writes in the wave
  const int64_t bonus = __shfl(tok, acc + 1, 64);  // posterior[acc]; every lane takes part
  if (next_block && bs > 0 && i < rearm_n) next_block[i] = i == 0 ? bonus : mask_id;
  if (i == 0 && bs > 0) {
    const int new_start = start + acc + 1;
    const int stop = dyn_d[DFL_DYN_STOP] | (any_stop ? 1 : 0);
    const int cyc = dyn_d[DFL_DYN_CYCLE] + 1;
    dyn_d[DFL_DYN_S] = start;
    dyn_d[DFL_DYN_TAU] = acc + 1;
    dyn_d[DFL_DYN_POS0] = start;
    dyn_d[DFL_DYN_START] = new_start;
    dyn_d[DFL_DYN_STOP] = stop;
    dyn_d[DFL_DYN_CYCLE] = cyc;
    dyn_t[DFL_DYN_S] = new_start;
    dyn_t[DFL_DYN_TAU] = 0;
    dyn_t[DFL_DYN_BS] = bs;
    dyn_t[DFL_DYN_POS0] = new_start;
    dyn_t[DFL_DYN_START] = new_start;
    dyn_t[DFL_DYN_STOP] = stop;
    dyn_t[DFL_DYN_CYCLE] = cyc;
    // blocks of 17..32 rows: the request's tiles have length records of their own for the per-tile launches (GEMMs,
    // norms, context K/V append): tile j holds context rows 16 j .. of the tau accepted ones and block rows 16 j ..
    for (int j = 0; dyn_dt && j < tiles_per_req; ++j) {
      int32_t *dd = dyn_dt + (r * tiles_per_req + j) * DFL_DYN_WORDS, *dt = dyn_tt + (r * tiles_per_req + j) * DFL_DYN_WORDS;
      const int tj = acc + 1 - 16 * j, bj = bs - 16 * j;
      dd[DFL_DYN_S] = start + 16 * j;
      dd[DFL_DYN_TAU] = tj < 0 ? 0 : (tj > 16 ? 16 : tj);
      dd[DFL_DYN_POS0] = start + 16 * j;
      dd[DFL_DYN_START] = new_start;
      dt[DFL_DYN_S] = new_start;
      dt[DFL_DYN_TAU] = 0;
      dt[DFL_DYN_BS] = bj < 0 ? 0 : (bj > 16 ? 16 : bj);
      dt[DFL_DYN_POS0] = new_start;
      dt[DFL_DYN_START] = new_start;
    }
    if (result) {
      result[r * 4 + 0] = acc;
      result[r * 4 + 1] = new_start;
      result[r * 4 + 2] = stop;
      result[r * 4 + 3] = cyc;
    }
  }
}

}  // namespace

extern "C" int dfl_argmax(const void *logits, int dtype, int rows, int64_t V, int64_t *ids, void *stream) {
  DFL_REQUIRE(logits && ids, "dfl_argmax: null pointer");
  DFL_REQUIRE(rows >= 0 && V > 0, "dfl_argmax: bad shape rows=%d V=%lld", rows, (long long)V);
  DFL_REQUIRE(dtype == 0 || dtype == 1, "dfl_argmax: dtype must be 0 (bf16) or 1 (fp32)");
  if (rows == 0) return DFL_OK;
  if (dtype == 0)
    hipLaunchKernelGGL((k_argmax<bf16_t, 8>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)logits, V, ids);
  else
    hipLaunchKernelGGL((k_argmax<float, 4>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (const float *)logits, V, ids);
  DFL_CHECK_LAUNCH("dfl_argmax");
  return DFL_OK;
}

extern "C" int dfl_accept_commit(const int64_t *block_ids, const int64_t *posterior, int bs, int64_t *output_ids,
                                 int64_t output_len, int32_t *dyn, const int64_t *stop_ids, int n_stop,
                                 int32_t *result, void *stream) {
  DFL_REQUIRE(block_ids && posterior && output_ids && dyn, "dfl_accept_commit: null pointer");
  // lane acc + 1 <= bs writes the bonus token: bs = 64 would need a 65th lane
  DFL_REQUIRE(bs >= 1 && bs <= 63, "dfl_accept_commit: bs=%d outside 1..63", bs);
  DFL_REQUIRE(n_stop == 0 || stop_ids, "dfl_accept_commit: n_stop>0 without stop_ids");
  hipLaunchKernelGGL(k_accept_commit, dim3(1), dim3(64), 0, (hipStream_t)stream, block_ids, posterior, bs, output_ids,
                     output_len, dyn, stop_ids, n_stop, result, (int64_t *)nullptr, 0, (int64_t)0, (int32_t *)nullptr);
  DFL_CHECK_LAUNCH("dfl_accept_commit");
  return DFL_OK;
}

extern "C" int dfl_accept_commit_rearm(const int64_t *block_ids, const int64_t *posterior, int bs, int64_t *output_ids,
                                       int64_t output_len, int32_t *dyn, const int64_t *stop_ids, int n_stop,
                                       int32_t *result, int64_t *next_block, int rearm_n, int64_t mask_id, void *stream) {
  DFL_REQUIRE(block_ids && posterior && output_ids && dyn && next_block, "dfl_accept_commit_rearm: null pointer");
  DFL_REQUIRE(bs >= 1 && bs <= 63 && rearm_n >= 1 && rearm_n <= 64, "dfl_accept_commit_rearm: bs=%d rearm_n=%d outside range", bs,
              rearm_n);
  DFL_REQUIRE(n_stop == 0 || stop_ids, "dfl_accept_commit_rearm: n_stop>0 without stop_ids");
  hipLaunchKernelGGL(k_accept_commit, dim3(1), dim3(64), 0, (hipStream_t)stream, block_ids, posterior, bs, output_ids,
                     output_len, dyn, stop_ids, n_stop, result, next_block, rearm_n, mask_id, (int32_t *)nullptr);
  DFL_CHECK_LAUNCH("dfl_accept_commit_rearm");
  return DFL_OK;
}

extern "C" int dfl_accept_commit_rearm_t(const int64_t *block_ids, const int64_t *posterior, int bs, int64_t *output_ids,
                                         int64_t output_len, int32_t *dyn, const int64_t *stop_ids, int n_stop,
                                         int32_t *result, int64_t *next_block, int rearm_n, int64_t mask_id, int32_t *dyn_t,
                                         void *stream) {
  DFL_REQUIRE(block_ids && posterior && output_ids && dyn && next_block && dyn_t, "dfl_accept_commit_rearm_t: null pointer");
  DFL_REQUIRE(bs >= 1 && bs <= 63 && rearm_n >= 1 && rearm_n <= 64, "dfl_accept_commit_rearm_t: bs=%d rearm_n=%d outside range", bs,
              rearm_n);
  DFL_REQUIRE(n_stop == 0 || stop_ids, "dfl_accept_commit_rearm_t: n_stop>0 without stop_ids");
  hipLaunchKernelGGL(k_accept_commit, dim3(1), dim3(64), 0, (hipStream_t)stream, block_ids, posterior, bs, output_ids,
                     output_len, dyn, stop_ids, n_stop, result, next_block, rearm_n, mask_id, dyn_t);
  DFL_CHECK_LAUNCH("dfl_accept_commit_rearm_t");
  return DFL_OK;
}

extern "C" int dfl_accept_commit_batch(const int64_t *block_ids, int64_t blk_stride, const int64_t *posterior,
                                       int64_t post_stride, int R, int64_t *output_ids, int64_t out_stride,
                                       int64_t output_len, int32_t *dyn_d, int32_t *dyn_t, const int64_t *stop_ids,
                                       int n_stop, int32_t *result, int64_t *next_block, int64_t mask_id,
                                       void *stream) {
  DFL_REQUIRE(block_ids && posterior && output_ids && dyn_d && dyn_t, "dfl_accept_commit_batch: null pointer");
  DFL_REQUIRE(R >= 1 && R <= 1024, "dfl_accept_commit_batch: R=%d outside 1..1024", R);
  DFL_REQUIRE(n_stop == 0 || stop_ids, "dfl_accept_commit_batch: n_stop>0 without stop_ids");
  hipLaunchKernelGGL(k_accept_commit_b, dim3(R), dim3(64), 0, (hipStream_t)stream, block_ids, blk_stride, posterior,
                     post_stride, output_ids, out_stride, output_len, dyn_d, dyn_t, stop_ids, n_stop, result, next_block,
                     mask_id, 16, 1, (int32_t *)nullptr, (int32_t *)nullptr);
  DFL_CHECK_LAUNCH("dfl_accept_commit_batch");
  return DFL_OK;
}

extern "C" int dfl_accept_commit_batch_t(const int64_t *block_ids, int64_t blk_stride, const int64_t *posterior,
                                         int64_t post_stride, int R, int64_t *output_ids, int64_t out_stride,
                                         int64_t output_len, int32_t *dyn_d, int32_t *dyn_t, const int64_t *stop_ids,
                                         int n_stop, int32_t *result, int64_t *next_block, int64_t mask_id, int tiles_per_req,
                                         int32_t *dyn_d_tiles, int32_t *dyn_t_tiles, void *stream) {
  DFL_REQUIRE(block_ids && posterior && output_ids && dyn_d && dyn_t && dyn_d_tiles && dyn_t_tiles,
              "dfl_accept_commit_batch_t: null pointer");
  DFL_REQUIRE(R >= 1 && R <= 1024 && (tiles_per_req == 1 || tiles_per_req == 2) && blk_stride >= 16 * tiles_per_req,
              "dfl_accept_commit_batch_t: R=%d tiles_per_req=%d outside range", R, tiles_per_req);
  DFL_REQUIRE(n_stop == 0 || stop_ids, "dfl_accept_commit_batch_t: n_stop>0 without stop_ids");
  hipLaunchKernelGGL(k_accept_commit_b, dim3(R), dim3(64), 0, (hipStream_t)stream, block_ids, blk_stride, posterior,
                     post_stride, output_ids, out_stride, output_len, dyn_d, dyn_t, stop_ids, n_stop, result, next_block,
                     mask_id, 16 * tiles_per_req, tiles_per_req, dyn_d_tiles, dyn_t_tiles);
  DFL_CHECK_LAUNCH("dfl_accept_commit_batch_t");
  return DFL_OK;
}
