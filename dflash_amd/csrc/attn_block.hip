// Block attention of the draft step (model/dflash.py:86-99): the block's 16 query
// rows attend, with no mask, to every cached prefix key plus this cycle's context
// and block keys.  GQA: the G query heads of a kv head share K/V.
//
// gfx950 mapping
//   grid (kv head, key split); workgroup = G waves, wave g owns query head kvh*G+g.
//   K/V tiles of 32 keys are read coalesced from the cache (16 B per lane, whole
//   256-B rows) and staged once per workgroup in LDS: K XOR-swizzled by row for
//   conflict-free ds_read_b128 A-fragments, V swizzled in 32-B chunks for
//   conflict-free ds_read_b64_tr_b16, which hands V^T fragments to the PV MFMA
//   without a transpose pass.  Q^T fragments live in registers for the launch.
//   Scores are computed transposed, S^T = K Q^T (mfma 16x16x32: keys on rows), so a
//   query's scores sit in one lane's registers: the row max / sum need two
//   cross-lane steps, and exp(S^T) is already the B operand of O^T += V^T P^T.
//   Online softmax in fp32 (base-2), P rounded to bf16 for the PV product exactly
//   as flash-style backends do; partial (m, l, O) per split, merged by
//   k_attn_merge which also writes the result as frag16 for o_proj.
// MFMA is used here and only here for attention math; the work is ~0.3 GFLOP per
// layer and 4 MB of K/V at S=1k, i.e. latency-bound, not a roofline kernel.
#include "dfl_common.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct AttnArgs {
  const bf16_t *q;  // [n_q][16][128]
  const bf16_t *kc, *vc;
  int cache_rows;
  int n_q, n_kv, G;
  float scale_log2;
  int causal;  // 1: query row j sees cache rows <= S + tau + j (target verify); 0: no mask (draft)
  const int32_t *dyn;
  float *o_part;   // [nsplit][n_q][16][128]
  float *ml_part;  // [nsplit][n_q][16][2]
};

__global__ __launch_bounds__(512) void k_block_attn(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char lds_k[32 * 256];
  __shared__ __attribute__((aligned(16))) char lds_v[32 * 256];

  const int tid = threadIdx.x, nthr = blockDim.x;
  const int wv = tid >> 6, l = tid & 63;
  const int kvh = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const int head = kvh * a.G + wv;
  const int qbase = a.dyn[DFL_DYN_S] + a.dyn[DFL_DYN_TAU];  // cache row of the block's first query
  const int kv_len = qbase + a.dyn[DFL_DYN_BS];
  const int ntiles = (kv_len + 31) >> 5;
  const int tps = (ntiles + nsplit - 1) / nsplit;
  const int t0 = split * tps;
  const int t1 = min(ntiles, t0 + tps);

  const int qi = l & 15, g = l >> 4;
  // Q^T B-fragments: lane (q = l&15, g) holds Q[q][32 s + 8 g .. +8]
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *reinterpret_cast<const bf16x8 *>(a.q + ((int64_t)head * 16 + qi) * 128 + s * 32 + g * 8);

  f32x4 o[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const bf16_t *kbase = a.kc + (int64_t)kvh * a.cache_rows * 128;
  const bf16_t *vbase = a.vc + (int64_t)kvh * a.cache_rows * 128;

  for (int t = t0; t < t1; ++t) {
    // ---- stage K and V tile (32 keys x 256 B each) ----
    for (int c = tid; c < 512; c += nthr) {
      const int row = c >> 4, ch = c & 15;
      int key = t * 32 + row;
      key = key < kv_len ? key : kv_len - 1;  // tail rows: any valid row, masked below
      const bf16x8 kk = *reinterpret_cast<const bf16x8 *>(kbase + (int64_t)key * 128 + ch * 8);
      const bf16x8 vv = *reinterpret_cast<const bf16x8 *>(vbase + (int64_t)key * 128 + ch * 8);
      *reinterpret_cast<bf16x8 *>(lds_k + row * 256 + ((ch ^ (row & 15)) << 4)) = kk;
      *reinterpret_cast<bf16x8 *>(lds_v + row * 256 + ((((ch >> 1) ^ (row & 7)) << 5) | ((ch & 1) << 4))) = vv;
    }
    __syncthreads();

    // ---- S^T = K Q^T for the two 16-key sub-tiles ----
    f32x4 sc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      sc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int row = u * 16 + qi;  // A fragment: lane (key = l&15, g)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ch = s * 4 + g;
        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(lds_k + row * 256 + ((ch ^ qi) << 4));
        sc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], sc[u], 0, 0, 0);
      }
    }
    // lane (q = l&15, g): sc[u][r] is key t*32 + u*16 + 4g + r
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 32 + u * 16 + 4 * g + r;
        const bool vis = key < kv_len && (!a.causal || key <= qbase + qi);
        const float v = vis ? sc[u][r] * a.scale_log2 : -INFINITY;
        sc[u][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // no mask: every tile in range holds >= 1 valid key, m_new is finite.  causal: a query
    // may see nothing of an early split's tile; keep -inf as the running max but never
    // form (-inf) - (-inf).
    const float m_new = fmaxf(m_run, mx);
    const float m_ref = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_ref);
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pb;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(sc[u][r] - m_ref);
        psum += p;
        pb[u * 4 + r] = f2bf(p);
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] *= alpha;

    // ---- O^T += V^T P^T: A = V^T fragments via transposed LDS reads ----
    // group of 16 lanes (g): block = keys 4g..4g+3 (rows) x 16 d (cols); lane 4qq+p
    // gives the address of row qq, cols 4p..4p+3 and receives column (l&15).
    {
      const int qq = (l & 15) >> 2, p = l & 3;
      const int r0 = 4 * g + qq, r1 = 16 + 4 * g + qq;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (lds_bf16x4 *)(lds_v + r0 * 256 + (((dt ^ (r0 & 7)) << 5) | (p << 3))));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (lds_bf16x4 *)(lds_v + r1 * 256 + (((dt ^ (r1 & 7)) << 5) | (p << 3))));
        const bf16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[dt], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  // lane (q, g): o[dt][r] = O[q][dt*16 + 4g + r]
  float *op = a.o_part + (((int64_t)split * a.n_q + head) * 16 + qi) * 128;
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4 *>(op + dt * 16 + 4 * g) = o[dt];
  if (g == 0) {
    float *ml = a.ml_part + (((int64_t)split * a.n_q + head) * 16 + qi) * 2;
    ml[0] = m_run;
    ml[1] = l_tot;
  }
}

// grid = n_q heads, 256 threads: thread (q = tid>>4, dg = tid&15) owns 8 d values
__global__ __launch_bounds__(256) void k_attn_merge(const float *o_part, const float *ml_part, int nsplit, int n_q,
                                                    bf16x8 *out_frag) {
  const int head = blockIdx.x, q = threadIdx.x >> 4, dg = threadIdx.x & 15;
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, ml_part[(((int64_t)s * n_q + head) * 16 + q) * 2]);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float L = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float *ml = ml_part + (((int64_t)s * n_q + head) * 16 + q) * 2;
    if (ml[1] <= 0.f) continue;  // empty split
    const float wgt = exp2f(ml[0] - M);
    L += wgt * ml[1];
    const float *op = o_part + (((int64_t)s * n_q + head) * 16 + q) * 128 + dg * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(op);
    const f32x4 a1 = *reinterpret_cast<const f32x4 *>(op + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] += wgt * a0[j];
      acc[4 + j] += wgt * a1[j];
    }
  }
  const float inv = L > 0.f ? 1.f / L : 0.f;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j] * inv);
  out_frag[(head * 16 + dg) * 16 + q] = o;  // frag16 chunk (n>>3 = head*16+dg, row q)
}

}  // namespace

extern "C" int64_t dfl_attn_ws_bytes(int n_q, int max_splits) {
  return (int64_t)max_splits * n_q * 16 * (128 + 2) * sizeof(float);
}

extern "C" int dfl_block_attn(const void *q, const void *kcache, const void *vcache, int cache_rows, int n_q, int n_kv,
                              float scale, int causal, const int32_t *dyn, int kv_len_max, void *ws,
                              int max_splits, void *out_frag, void *stream) {
  DFL_REQUIRE(q && kcache && vcache && dyn && ws && out_frag, "dfl_block_attn: null pointer");
  DFL_REQUIRE(n_q > 0 && n_kv > 0 && n_q % n_kv == 0 && n_q / n_kv <= 8, "dfl_block_attn: GQA group must be 1..8 (n_q=%d n_kv=%d)",
              n_q, n_kv);
  DFL_REQUIRE(kv_len_max > 0 && kv_len_max <= cache_rows, "dfl_block_attn: kv_len_max=%d exceeds cache_rows=%d", kv_len_max,
              cache_rows);
  DFL_REQUIRE(max_splits >= 1, "dfl_block_attn: max_splits < 1");
  // ~4 key tiles (128 keys) per split, bounded by the workspace
  int nsplit = (kv_len_max + 127) / 128;
  nsplit = nsplit < 1 ? 1 : (nsplit > max_splits ? max_splits : nsplit);
  AttnArgs a{};
  a.q = (const bf16_t *)q;
  a.kc = (const bf16_t *)kcache;
  a.vc = (const bf16_t *)vcache;
  a.cache_rows = cache_rows;
  a.n_q = n_q;
  a.n_kv = n_kv;
  a.G = n_q / n_kv;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.causal = causal ? 1 : 0;
  a.dyn = dyn;
  a.o_part = (float *)ws;
  a.ml_part = (float *)ws + (int64_t)max_splits * n_q * 16 * 128;
  hipLaunchKernelGGL(k_block_attn, dim3(n_kv, nsplit), dim3(a.G * 64), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(k_attn_merge, dim3(n_q), dim3(256), 0, (hipStream_t)stream, (const float *)a.o_part,
                     (const float *)a.ml_part, nsplit, n_q, (bf16x8 *)out_frag);
  DFL_CHECK_LAUNCH("dfl_block_attn");
  return DFL_OK;
}
