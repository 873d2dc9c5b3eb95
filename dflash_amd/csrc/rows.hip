// Row-wise stages between the GEMMs: frag16 conversion, residual + RMSNorm,
// per-head q/k RMSNorm + RoPE + KV-cache append.  All are tiny (<= 32 rows) and
// latency-bound; their job is to leave activations in the layout the next GEMM
// streams against (frag16) and K/V in the cache, with the reference's bf16
// rounding points.
#include "dfl_common.h"

namespace {

__global__ void k_set_dyn(int32_t *dyn, int S, int tau, int bs, int pos0) {
  if (threadIdx.x == 0) {
    dyn[DFL_DYN_S] = S;
    dyn[DFL_DYN_TAU] = tau;
    dyn[DFL_DYN_BS] = bs;
    dyn[DFL_DYN_POS0] = pos0;
    dyn[DFL_DYN_START] = pos0 + tau;
    dyn[DFL_DYN_STOP] = 0;
    dyn[DFL_DYN_CYCLE] = 0;
    dyn[7] = 0;
  }
}

// Two records for a block of up to 32 rows handled as two 16-row tiles: record t holds the valid-row counts of
// tile t (rows 16 t ..), everything else as k_set_dyn.
__global__ void k_set_dyn2(int32_t *dyn, int S, int tau, int bs, int pos0) {
  const int t = threadIdx.x;
  if (t < 2) {
    int32_t *d = dyn + t * DFL_DYN_WORDS;
    const int tt = tau - 16 * t, bb = bs - 16 * t;
    d[DFL_DYN_S] = S;
    d[DFL_DYN_TAU] = tt < 0 ? 0 : (tt > 16 ? 16 : tt);
    d[DFL_DYN_BS] = bb < 0 ? 0 : (bb > 16 ? 16 : bb);
    d[DFL_DYN_POS0] = pos0;
    d[DFL_DYN_START] = pos0 + tau;
    d[DFL_DYN_STOP] = 0;
    d[DFL_DYN_CYCLE] = 0;
    d[7] = 0;
  }
}

// one 16-B chunk per thread: frag[(k8*16 + m)*8 ..] = x[m][k8*8 ..]
__global__ void k_pack_rows(const bf16_t *__restrict__ x, int64_t ldx, int rows, int K8, bf16x8 *__restrict__ xf,
                            const int32_t *dyn, int dyn_word) {
  int nv = rows;
  if (dyn) nv = min(rows, dyn[dyn_word]);
  const int total = K8 * 16;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    const int m = c & 15, k8 = c >> 4;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m < nv) v = *reinterpret_cast<const bf16x8 *>(x + (int64_t)m * ldx + (int64_t)k8 * 8);
    xf[c] = v;
  }
}

struct NormArgs {
  const float *part;
  int nsplit;
  int64_t part_split;
  int ldp, row_off;
  const bf16_t *resid_in;
  const bf16_t *embed;
  const int64_t *ids;
  bf16_t *h_out;
  bf16_t *h_out2;  // optional second copy of the residual row (target taps), row stride ld2
  int64_t ld2;
  const bf16_t *norm_w;
  float eps;
  bf16x8 *frag;
  int H;
  const int32_t *dyn;
  int dyn_word;
};

// grid = 16 rows, 256 threads; each thread owns chunks c = tid, tid+256, ... of 8 columns
template <int MAXC>
__global__ __launch_bounds__(256) void k_norm_pack(NormArgs a) {
  __shared__ float wsum[4];
  const int m = blockIdx.x;
  const int tid = threadIdx.x;
  int nv = 16;
  if (a.dyn) nv = a.dyn[a.dyn_word];
  const int nchunks = a.H >> 3;
  if (m >= nv) {
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = tid; c < nchunks; c += 256) a.frag[c * 16 + m] = z;
    return;
  }
  const bf16_t *src = nullptr;
  if (a.embed)
    src = a.embed + a.ids[m] * (int64_t)a.H;
  else if (a.resid_in)
    src = a.resid_in + (int64_t)m * a.H;

  float h[MAXC][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    if (c < nchunks) {
      float v[8];
      if (a.part) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        for (int s = 0; s < a.nsplit; ++s) {
          const float *p = a.part + s * a.part_split + (int64_t)(a.row_off + m) * a.ldp + c * 8;
          const f32x4 p0 = *reinterpret_cast<const f32x4 *>(p);
          const f32x4 p1 = *reinterpret_cast<const f32x4 *>(p + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] += p0[j];
            v[4 + j] += p1[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rbf(v[j]);  // the Linear's bf16 output
      }
      if (src) {
        const bf16x8 r = *reinterpret_cast<const bf16x8 *>(src + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) h[i][j] = a.part ? rbf(bf2f(r[j]) + v[j]) : bf2f(r[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[i][j] = v[j];
      }
      if (a.h_out) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(h[i][j]);
        *reinterpret_cast<bf16x8 *>(a.h_out + (int64_t)m * a.H + c * 8) = o;
        if (a.h_out2) *reinterpret_cast<bf16x8 *>(a.h_out2 + (int64_t)m * a.ld2 + c * 8) = o;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += h[i][j] * h[i][j];
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  const float tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  const float rstd = rsqrtf(tot / (float)a.H + a.eps);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    if (c < nchunks) {
      const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(a.norm_w + c * 8);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wv[j]) * rbf(h[i][j] * rstd));
      a.frag[c * 16 + m] = o;
    }
  }
}

struct RopeArgs {
  const float *qkv;
  int nsplit;
  int64_t split_stride;
  int ld, q_col, k_col, v_col, ctx_row0, blk_row0;
  int n_q, n_kv;
  const bf16_t *q_w, *k_w;
  float eps;
  const bf16_t *cos_tab, *sin_tab;
  int max_pos;
  bf16_t *q_out, *kcache, *vcache;
  int cache_rows;
  const int32_t *dyn;
  int ctx_override, row_base;
  // ragged batch: grid.y = request, grid.z = layer (the draft's context K/V of every layer
  // depend only on the context rows, so all layers are appended in one launch)
  int req_rows;
  int64_t cache_req_stride, cache_layer_stride, kw_layer_stride;
  int col_layer_stride;
  int tiles_per_req;  // grid.y counts 16-row TILES; tiles_per_req consecutive ones belong to one request (one cache)
};

// One wave per (kind, row, head) item of 128 values; lane owns d = l and l + 64,
// the two halves rotate_half pairs up (tf:...modeling_qwen3.py:140-144).
// Items: [0, 16*n_q) q of block rows; then (slot 0..31, kv head) for k; then for v,
// slot < 16 = context row, slot >= 16 = block row.
__global__ __launch_bounds__(256) void k_qknorm_rope(RopeArgs a_in) {
  RopeArgs a = a_in;
  if (blockIdx.y | blockIdx.z) {  // uniform
    const int r = blockIdx.y, ly = blockIdx.z;
    a.dyn += r * DFL_DYN_WORDS;
    a.ctx_row0 += r * a.req_rows;
    if (a.blk_row0 >= 0) a.blk_row0 += r * a.req_rows;
    const int rq = a.tiles_per_req > 1 ? r / a.tiles_per_req : r;
    a.kcache += rq * a.cache_req_stride + ly * a.cache_layer_stride;
    a.vcache += rq * a.cache_req_stride + ly * a.cache_layer_stride;
    a.k_col += ly * a.col_layer_stride;
    a.v_col += ly * a.col_layer_stride;
    if (a.k_w) a.k_w += ly * a.kw_layer_stride;
  }
  const int l = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int S = a.dyn[DFL_DYN_S];
  int tau = a.dyn[DFL_DYN_TAU], bs = a.dyn[DFL_DYN_BS];
  const int pos0 = a.dyn[DFL_DYN_POS0];
  if (a.ctx_override >= 0) {
    tau = a.ctx_override;
    bs = 0;
  }
  if (a.blk_row0 < 0) bs = 0;
  const int nq_items = 16 * a.n_q, nk_items = 32 * a.n_kv;
  int kind, slot, head;
  if (item < nq_items) {
    kind = 0;
    slot = 16 + item / a.n_q;
    head = item % a.n_q;
  } else if (item < nq_items + nk_items) {
    kind = 1;
    slot = (item - nq_items) / a.n_kv;
    head = (item - nq_items) % a.n_kv;
  } else if (item < nq_items + 2 * nk_items) {
    kind = 2;
    slot = (item - nq_items - nk_items) / a.n_kv;
    head = (item - nq_items - nk_items) % a.n_kv;
  } else {
    return;
  }
  if (kind == 0 && a.q_col < 0) return;
  // validity, buffer row, and sequence index relative to the first new row
  int rel, brow;
  if (slot < 16) {
    if (slot >= tau) return;
    rel = a.row_base + slot;
    brow = a.ctx_row0 + slot;
  } else {
    if (slot - 16 >= bs) return;
    rel = tau + (slot - 16);
    brow = a.blk_row0 + (slot - 16);
  }
  const int col = (kind == 0 ? a.q_col : kind == 1 ? a.k_col : a.v_col) + head * 128;
  float x1 = 0.f, x2 = 0.f;
  for (int s = 0; s < a.nsplit; ++s) {
    const float *p = a.qkv + s * a.split_stride + (int64_t)brow * a.ld + col;
    x1 += p[l];
    x2 += p[l + 64];
  }
  x1 = rbf(x1);
  x2 = rbf(x2);  // Linear output in bf16
  const int crow = S + rel;  // cache row
  if (kind == 2) {
    if (crow < a.cache_rows) {
      bf16_t *dst = a.vcache + ((int64_t)head * a.cache_rows + crow) * 128;
      dst[l] = f2bf(x1);
      dst[l + 64] = f2bf(x2);
    }
    return;
  }
  // Qwen3RMSNorm over head_dim (model/dflash.py:72,79)
  const bf16_t *nw = kind == 0 ? a.q_w : a.k_w;
  float n1 = x1, n2 = x2;
  if (nw) {  // Qwen3 has per-head q/k norms, Llama does not
    const float ss = wave_sum(x1 * x1 + x2 * x2);
    const float rstd = rsqrtf(ss * (1.f / 128.f) + a.eps);
    n1 = rbf(bf2f(nw[l]) * rbf(x1 * rstd));
    n2 = rbf(bf2f(nw[l + 64]) * rbf(x2 * rstd));
  }
  // RoPE, model/dflash.py:22-28: (x*cos) + (rotate_half(x)*sin), each product and the
  // sum rounded to bf16 as torch's elementwise bf16 ops do.
  int pos = pos0 + rel;
  pos = pos < a.max_pos ? pos : a.max_pos - 1;
  const float c = bf2f(a.cos_tab[(int64_t)pos * 64 + l]);
  const float sn = bf2f(a.sin_tab[(int64_t)pos * 64 + l]);
  const float o1 = rbf(rbf(n1 * c) + rbf(-n2 * sn));
  const float o2 = rbf(rbf(n2 * c) + rbf(n1 * sn));
  if (kind == 0) {
    bf16_t *dst = a.q_out + ((int64_t)head * 16 + (slot - 16)) * 128;
    dst[l] = f2bf(o1);
    dst[l + 64] = f2bf(o2);
  } else if (crow < a.cache_rows) {
    bf16_t *dst = a.kcache + ((int64_t)head * a.cache_rows + crow) * 128;
    dst[l] = f2bf(o1);
    dst[l + 64] = f2bf(o2);
  }
}

}  // namespace

extern "C" int dfl_set_dyn(int32_t *dyn, int S, int tau, int bs, int pos0, void *stream) {
  DFL_REQUIRE(dyn, "dfl_set_dyn: null pointer");
  DFL_REQUIRE(S >= 0 && tau >= 0 && bs >= 0 && bs <= 63 && pos0 >= 0, "dfl_set_dyn: bad lengths S=%d tau=%d bs=%d pos0=%d", S,
              tau, bs, pos0);
  hipLaunchKernelGGL(k_set_dyn, dim3(1), dim3(64), 0, (hipStream_t)stream, dyn, S, tau, bs, pos0);
  DFL_CHECK_LAUNCH("dfl_set_dyn");
  return DFL_OK;
}

extern "C" int dfl_set_dyn2(int32_t *dyn, int S, int tau, int bs, int pos0, void *stream) {
  DFL_REQUIRE(dyn, "dfl_set_dyn2: null pointer");
  DFL_REQUIRE(S >= 0 && tau >= 0 && tau <= 32 && bs >= 0 && bs <= 32 && pos0 >= 0,
              "dfl_set_dyn2: bad lengths S=%d tau=%d bs=%d pos0=%d", S, tau, bs, pos0);
  hipLaunchKernelGGL(k_set_dyn2, dim3(1), dim3(64), 0, (hipStream_t)stream, dyn, S, tau, bs, pos0);
  DFL_CHECK_LAUNCH("dfl_set_dyn2");
  return DFL_OK;
}

extern "C" int dfl_pack_rows(const void *x, int64_t ldx, int rows, int K, void *xf, const int32_t *dyn, int dyn_word,
                             void *stream) {
  DFL_REQUIRE(x && xf, "dfl_pack_rows: null pointer");
  DFL_REQUIRE(rows >= 0 && rows <= 16 && K > 0 && K % 8 == 0 && ldx % 8 == 0, "dfl_pack_rows: rows<=16, K%%8==0, ldx%%8==0");
  const int total = (K / 8) * 16;
  hipLaunchKernelGGL(k_pack_rows, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, ldx,
                     rows, K / 8, (bf16x8 *)xf, dyn, dyn_word);
  DFL_CHECK_LAUNCH("dfl_pack_rows");
  return DFL_OK;
}

extern "C" int dfl_norm_pack(const float *part, int nsplit, int64_t part_split, int ldp, int row_off,
                             const void *resid_in, const void *embed, const int64_t *ids, void *h_out,
                             void *h_out2, int64_t ld2, const void *norm_w, float eps, void *frag, int H,
                             const int32_t *dyn, int dyn_word, void *stream) {
  DFL_REQUIRE(norm_w && frag, "dfl_norm_pack: null norm_w/frag");
  DFL_REQUIRE(part || resid_in || embed, "dfl_norm_pack: no input");
  DFL_REQUIRE(!(embed && !ids), "dfl_norm_pack: embed without ids");
  DFL_REQUIRE(H > 0 && H % 8 == 0 && H <= 16384, "dfl_norm_pack: H=%d unsupported", H);
  DFL_REQUIRE(!part || (nsplit >= 1 && ldp % 4 == 0), "dfl_norm_pack: bad partial layout");
  DFL_REQUIRE(!h_out2 || (h_out && ld2 % 8 == 0), "dfl_norm_pack: h_out2 needs h_out and ld2%%8==0");
  NormArgs a{part, nsplit, part_split, ldp, row_off, (const bf16_t *)resid_in, (const bf16_t *)embed, ids,
             (bf16_t *)h_out, (bf16_t *)h_out2, ld2, (const bf16_t *)norm_w, eps, (bf16x8 *)frag, H, dyn, dyn_word};
  const int nchunks = H / 8;
  if (nchunks <= 512)
    hipLaunchKernelGGL(k_norm_pack<2>, dim3(16), dim3(256), 0, (hipStream_t)stream, a);
  else if (nchunks <= 1024)
    hipLaunchKernelGGL(k_norm_pack<4>, dim3(16), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_norm_pack<8>, dim3(16), dim3(256), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_norm_pack");
  return DFL_OK;
}

extern "C" int dfl_qknorm_rope_append(const float *qkv, int nsplit, int64_t split_stride, int ld, int q_col, int k_col,
                                      int v_col, int ctx_row0, int blk_row0, int n_q, int n_kv, const void *q_norm_w,
                                      const void *k_norm_w, float eps, const void *cos_tab, const void *sin_tab,
                                      int max_pos, void *q_out, void *kcache, void *vcache, int cache_rows,
                                      const int32_t *dyn, int ctx_rows_override, int row_base, void *stream) {
  DFL_REQUIRE(qkv && cos_tab && sin_tab && kcache && vcache && dyn, "dfl_qknorm_rope_append: null pointer");
  DFL_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), "dfl_qknorm_rope_append: give both norm weights or neither");
  DFL_REQUIRE(q_col < 0 || q_out, "dfl_qknorm_rope_append: q wanted but q_out is null");
  DFL_REQUIRE(nsplit >= 1 && n_q > 0 && n_kv > 0 && n_q % n_kv == 0, "dfl_qknorm_rope_append: bad head counts");
  DFL_REQUIRE(ctx_rows_override <= 16, "dfl_qknorm_rope_append: at most 16 context rows per call");
  DFL_REQUIRE(k_col >= 0 && v_col >= 0 && ld > 0 && max_pos > 0, "dfl_qknorm_rope_append: bad layout");
  RopeArgs a{qkv, nsplit, split_stride, ld, q_col, k_col, v_col, ctx_row0, blk_row0, n_q, n_kv,
             (const bf16_t *)q_norm_w, (const bf16_t *)k_norm_w, eps, (const bf16_t *)cos_tab, (const bf16_t *)sin_tab,
             max_pos, (bf16_t *)q_out, (bf16_t *)kcache, (bf16_t *)vcache, cache_rows, dyn, ctx_rows_override, row_base};
  const int items = 16 * n_q + 2 * 32 * n_kv;
  hipLaunchKernelGGL(k_qknorm_rope, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_qknorm_rope_append");
  return DFL_OK;
}

/* see include/dflash_hip.h */
static int kv_append_batch_impl(const float *kv, int nsplit, int64_t split_stride, int ld, int k_col, int v_col,
                                   int col_layer_stride, int n_layers, int R, int req_rows, int n_kv,
                                   const void *k_norm_w, int64_t kw_layer_stride, float eps, const void *cos_tab,
                                   const void *sin_tab, int max_pos, void *kcache, void *vcache, int cache_rows,
                                   int64_t cache_req_stride, int64_t cache_layer_stride, const int32_t *dyn,
                                   int tiles_per_req, void *stream) {
  DFL_REQUIRE(kv && cos_tab && sin_tab && kcache && vcache && dyn, "dfl_kv_append_batch: null pointer");
  DFL_REQUIRE(nsplit >= 1 && n_kv > 0 && ld > 0 && k_col >= 0 && v_col >= 0 && max_pos > 0, "dfl_kv_append_batch: bad layout");
  DFL_REQUIRE(R >= 1 && R <= 64 && n_layers >= 1 && req_rows >= 16 && tiles_per_req >= 1, "dfl_kv_append_batch: bad batch shape");
  RopeArgs a{kv, nsplit, split_stride, ld, -1, k_col, v_col, 0, -1, /*n_q=*/0, n_kv,
             nullptr, (const bf16_t *)k_norm_w, eps, (const bf16_t *)cos_tab, (const bf16_t *)sin_tab,
             max_pos, nullptr, (bf16_t *)kcache, (bf16_t *)vcache, cache_rows, dyn, -1, 0,
             req_rows, cache_req_stride, cache_layer_stride, kw_layer_stride, col_layer_stride, tiles_per_req};
  // no q items (n_q = 0); of the 32 k / v slots per head only the context ones (< tau) do work
  const int items = 2 * 32 * n_kv;
  hipLaunchKernelGGL(k_qknorm_rope, dim3((items + 3) / 4, R, n_layers), dim3(256), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_kv_append_batch");
  return DFL_OK;
}

extern "C" int dfl_kv_append_batch(const float *kv, int nsplit, int64_t split_stride, int ld, int k_col, int v_col,
                                   int col_layer_stride, int n_layers, int R, int req_rows, int n_kv,
                                   const void *k_norm_w, int64_t kw_layer_stride, float eps, const void *cos_tab,
                                   const void *sin_tab, int max_pos, void *kcache, void *vcache, int cache_rows,
                                   int64_t cache_req_stride, int64_t cache_layer_stride, const int32_t *dyn,
                                   void *stream) {
  return kv_append_batch_impl(kv, nsplit, split_stride, ld, k_col, v_col, col_layer_stride, n_layers, R, req_rows, n_kv, k_norm_w,
                              kw_layer_stride, eps, cos_tab, sin_tab, max_pos, kcache, vcache, cache_rows, cache_req_stride,
                              cache_layer_stride, dyn, 1, stream);
}

extern "C" int dfl_kv_append_batch_t(const float *kv, int nsplit, int64_t split_stride, int ld, int k_col, int v_col,
                                     int col_layer_stride, int n_layers, int n_tiles, int req_rows, int n_kv,
                                     const void *k_norm_w, int64_t kw_layer_stride, float eps, const void *cos_tab,
                                     const void *sin_tab, int max_pos, void *kcache, void *vcache, int cache_rows,
                                     int64_t cache_req_stride, int64_t cache_layer_stride, const int32_t *dyn_tiles,
                                     int tiles_per_req, void *stream) {
  return kv_append_batch_impl(kv, nsplit, split_stride, ld, k_col, v_col, col_layer_stride, n_layers, n_tiles, req_rows, n_kv,
                              k_norm_w, kw_layer_stride, eps, cos_tab, sin_tab, max_pos, kcache, vcache, cache_rows,
                              cache_req_stride, cache_layer_stride, dyn_tiles, tiles_per_req, stream);
}
