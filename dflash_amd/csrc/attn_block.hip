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
#include <type_traits>

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

// Merge the key splits of NH (head, q, 8-column group) items that differ only in the
// head (consecutive heads): branch-free, every load of a batch of U splits issued before any
// is used (an empty split has m = -inf, l = 0 and gets weight 0).  ONE pass with a running
// maximum (online rescaling, as inside a split) instead of a maximum pass followed by a sum
// pass: every batch is a dependent ~0.65 us round trip to partials other workgroups wrote, and
// 9 splits took 3 + 5 rounds (5.4 us of the stage's 20, scripts/dbg_attn_stamps.py); now 3.
// Fixed split order, so the sums are reproducible.
template <int NH>
__device__ __forceinline__ void merge_items(const float *o_part, const float *ml_part, int nsplit, int n_q, int hh0,
                                            int q, int dg, bf16x8 (&out)[NH]) {
  constexpr int U = 3;  // (U = 4: 247 VGPRs, no faster)
  const int64_t sml = (int64_t)n_q * 16 * 2, so = (int64_t)n_q * 16 * 128;
  const float *ml0 = ml_part + ((int64_t)hh0 * 16 + q) * 2;
  const float *o0 = o_part + ((int64_t)hh0 * 16 + q) * 128 + dg * 8;
  float M[NH], L[NH], acc[NH][8];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    M[h] = -INFINITY;
    L[h] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
  }
  for (int s = 0; s < nsplit; s += U) {
    float ms[U][NH], ls[U][NH];
    f32x4 a0[U][NH], a1[U][NH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sc = s + u < nsplit ? s + u : nsplit - 1;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        ms[u][h] = ml0[sc * sml + h * 32];
        ls[u][h] = ml0[sc * sml + h * 32 + 1];
        a0[u][h] = *reinterpret_cast<const f32x4 *>(o0 + sc * so + h * 2048);
        a1[u][h] = *reinterpret_cast<const f32x4 *>(o0 + sc * so + h * 2048 + 4);
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float mnew = M[h];
#pragma unroll
      for (int u = 0; u < U; ++u) mnew = fmaxf(mnew, s + u < nsplit ? ms[u][h] : -INFINITY);
      const float mref = mnew == -INFINITY ? 0.f : mnew;
      const float scale = exp2f(M[h] - mref);  // first batch: exp2(-inf) = 0 on zero sums
      M[h] = mnew;
      L[h] *= scale;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[h][j] *= scale;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float wgt = s + u < nsplit ? exp2f(ms[u][h] - mref) : 0.f;  // exp2(-inf) = 0: empty split
        L[h] += wgt * ls[u][h];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[h][j] += wgt * a0[u][h][j];
          acc[h][4 + j] += wgt * a1[u][h][j];
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const float inv = L[h] > 0.f ? 1.f / L[h] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[h][j] = f2bf(acc[h][j] * inv);
  }
}

// grid = n_q heads, 256 threads: thread (q = tid>>4, dg = tid&15) owns 8 d values
__global__ __launch_bounds__(256) void k_attn_merge(const float *o_part, const float *ml_part, int nsplit, int n_q,
                                                    bf16x8 *out_frag) {
  const int head = blockIdx.x, q = threadIdx.x >> 4, dg = threadIdx.x & 15;
  bf16x8 r[1];
  merge_items<1>(o_part, ml_part, nsplit, n_q, head, q, dg, r);
  out_frag[(head * 16 + dg) * 16 + q] = r[0];  // frag16 chunk (n>>3 = head*16+dg, row q)
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

// ============================================================================
// Fused attention stage: one launch does what k_qknorm_rope + k_block_attn +
// k_attn_merge do in three (the stage is launch/latency-bound: ~5+9+9 us per layer in
// profiles/r1_cycle_breakdown.txt against ~1 us of actual data movement).
//   grid = (kv heads, key splits), workgroup = G waves (wave g = query head kvh*G+g).
//   phase 0  every workgroup turns its heads' q rows from the QKV GEMM's fp32 partials
//            into bf16 q (Linear rounding, per-head RMSNorm, RoPE) in LDS — all 16 rows
//            of a head in flight at once (unconditional loads, DPP reductions);
//   phase 1  the LAST split does the same for the new K rows and copies the new V rows
//            (tau + bs <= 32 rows of its kv head) into the cache for later cycles AND
//            into LDS, from where its tiles take them.  Split ranges are laid out so
//            that every row >= S belongs to the last split: no other workgroup reads a
//            cache row written in this launch;
//   phase 2  the tile loop of k_block_attn with K/V chunks prefetched two tiles ahead,
//            the first two requested before the prologues;
//   phase 3  (m, l, O) partials; the last workgroup to arrive for a kv head (agent-scope
//            release -> ticket -> acquire, cdna guide §6 G16) merges the splits in a
//            fixed order and writes frag16.  One split: written directly.
namespace {

struct FusedAttnArgs {
  const float *qkv;
  int nsplit_k;
  int64_t split_stride;
  int ld, q_col, k_col, v_col, ctx_row0, blk_row0;
  const bf16_t *q_w, *k_w;
  float eps;
  const bf16_t *cos_tab, *sin_tab;
  int max_pos;
  bf16_t *kc, *vc;
  int cache_rows;
  int n_q, n_kv, G;
  float scale_log2;
  int causal;
  const int32_t *dyn;
  bf16x8 *out_frag;
  float *o_part;   // [nsplit][n_q][16][128]
  float *ml_part;  // [nsplit][n_q][16][2]
  int *tickets;    // [n_kv], zero before the first launch; the merger leaves it zero
  // ragged batch (grid.z = request): request r's rows sit req_rows further down the partial
  // buffer, its cache / output / workspace at the strides below, its lengths at dyn + 8 r
  int req_rows;
  int64_t cache_req_stride, out_req_stride, opart_req_stride, ml_req_stride;
};

// NP passes of 4 (row, head) items each on one wave: the 16 lanes of a DPP row share an
// item, lane c = l&15 owns d = 8c .. 8c+7 (c < 8: first half of the head, c >= 8: second
// half; rotate_half pairs lane c with lane c^8 of the same row).  Per pass a lane issues
// 2*nsplit 16-B loads of partial sums plus three 16-B table loads, the sum of squares is a
// 4-step DPP row reduction and the partner values arrive by row_ror:8 — ~5x fewer
// instructions than one item per wave with lane = d (6 us -> measured below).
// brow < 0 marks an absent item (loads are clamped, outputs garbage, never stored).
template <int NP>
__device__ __forceinline__ void rope_rows(const FusedAttnArgs &a, const int (&brow)[NP], const int (&col)[NP],
                                          const int (&pos)[NP], const bool (&rope)[NP], const bf16_t *nw, int l,
                                          bf16x8 (&out)[NP]) {
  const int c = l & 15, d0 = c * 8;
  float x[NP][8];
  bf16x8 cs[NP], sn[NP];
  const float *rp[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    rp[p] = a.qkv + (int64_t)(brow[p] < 0 ? 0 : brow[p]) * a.ld + col[p] + d0;
    int pp = pos[p] < a.max_pos ? pos[p] : a.max_pos - 1;
    pp = pp < 0 ? 0 : pp;
    cs[p] = *reinterpret_cast<const bf16x8 *>(a.cos_tab + (int64_t)pp * 64 + (d0 & 63));
    sn[p] = *reinterpret_cast<const bf16x8 *>(a.sin_tab + (int64_t)pp * 64 + (d0 & 63));
#pragma unroll
    for (int j = 0; j < 8; ++j) x[p][j] = 0.f;
  }
  bf16x8 wv = {0, 0, 0, 0, 0, 0, 0, 0};
  if (nw) wv = *reinterpret_cast<const bf16x8 *>(nw + d0);
  for (int s = 0; s < a.nsplit_k; ++s) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const f32x4 v0 = *reinterpret_cast<const f32x4 *>(rp[p] + s * a.split_stride);
      const f32x4 v1 = *reinterpret_cast<const f32x4 *>(rp[p] + s * a.split_stride + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[p][j] += v0[j];
        x[p][4 + j] += v1[j];
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x[p][j] = rbf(x[p][j]);  // the Linear's bf16 output
      ss += x[p][j] * x[p][j];
    }
    float n[8];
    if (nw) {  // Qwen3RMSNorm over the 128 values of the head = the 16 lanes of this DPP row
      ss += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ss), 0xB1, 0xF, 0xF, true));
      ss += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ss), 0x4E, 0xF, 0xF, true));
      ss += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ss), 0x141, 0xF, 0xF, true));
      ss += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ss), 0x140, 0xF, 0xF, true));
      const float rstd = rsqrtf(ss * (1.f / 128.f) + a.eps);
#pragma unroll
      for (int j = 0; j < 8; ++j) n[j] = rbf(bf2f(wv[j]) * rbf(x[p][j] * rstd));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) n[j] = x[p][j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // partner value n[d +- 64] sits in lane c^8 of the same row: rotate the row by 8
      const float pn =
          __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, n[j]), 0x128, 0xF, 0xF, true));
      const float cj = bf2f(cs[p][j]), sj = bf2f(sn[p][j]);
      // rotate_half: first half gets -x2, second half +x1 (tf:modeling_qwen3.py:140-144)
      const float r = rbf(rbf(n[j] * cj) + rbf((c < 8 ? -pn : pn) * sj));
      out[p][j] = f2bf(rope[p] ? r : x[p][j]);
    }
  }
}

#ifdef DFL_ATTN_STAMPS  // diagnostic build only (scripts/dbg_attn_stamps.py): 100 MHz wall stamps per phase
__device__ unsigned long long g_stamps[2][8];
#define STAMP(i)                                                                           \
  do {                                                                                     \
    if (tid == 0 && kvh == 0 && (split == 0 || split == nsplit - 1))                       \
      g_stamps[split == 0 ? 0 : 1][i] = __builtin_amdgcn_s_memrealtime();                 \
  } while (0)
#else
#define STAMP(i)
#endif

template <int G>
__global__ __launch_bounds__(G * 64) void k_attn_fused(FusedAttnArgs a_in) {
  FusedAttnArgs a = a_in;
  if (blockIdx.z) {  // uniform: request index of a ragged batch
    const int r = blockIdx.z;
    a.dyn += r * DFL_DYN_WORDS;
    a.ctx_row0 += r * a.req_rows;
    a.blk_row0 += r * a.req_rows;
    a.kc += r * a.cache_req_stride;
    a.vc += r * a.cache_req_stride;
    a.out_frag += r * a.out_req_stride;
    a.o_part += r * a.opart_req_stride;
    a.ml_part += r * a.ml_req_stride;
    a.tickets += r * a.n_kv;
  }
  __shared__ __attribute__((aligned(16))) char lds_k[32 * 256];
  __shared__ __attribute__((aligned(16))) char lds_v[32 * 256];
  __shared__ __attribute__((aligned(16))) bf16_t new_k[32][128];
  __shared__ __attribute__((aligned(16))) bf16_t new_v[32][128];
  __shared__ __attribute__((aligned(16))) bf16_t q_lds[G][16][128];
  __shared__ int s_last;

  constexpr int nthr = G * 64;
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const int kvh = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const int head = kvh * G + wv;
  const int S = a.dyn[DFL_DYN_S], tau = a.dyn[DFL_DYN_TAU], bs = a.dyn[DFL_DYN_BS], pos0 = a.dyn[DFL_DYN_POS0];
  const int qbase = S + tau;
  const int kv_len = qbase + bs;
  const int n_new = tau + bs;  // <= 32
  const int ntiles = (kv_len + 31) >> 5;
  STAMP(0);

  // ---- key-tile range of this split.  The last split owns every tile that holds a row
  // >= S (tiles >= S>>5) and an even share of the rest.
  int t0, t1;
  {
    const int tnew = S >> 5;
    const int tps = (ntiles + nsplit - 1) / nsplit;
    int last0 = ntiles - tps;
    last0 = last0 < 0 ? 0 : (last0 > tnew ? tnew : last0);
    if (split == nsplit - 1) {
      t0 = last0;
      t1 = ntiles;
    } else {
      const int per = (last0 + nsplit - 2) / (nsplit - 1);
      t0 = split * per;
      t1 = t0 + per;
      t0 = t0 > last0 ? last0 : t0;
      t1 = t1 > last0 ? last0 : t1;
    }
  }
  const bool is_last_split = split == nsplit - 1;

  // K/V chunk ownership: chunk c of a tile = (row c>>4, 16-B piece c&15); a thread owns
  // chunks tid, tid+nthr, ...  Old rows come from the cache; rows >= S (last split only)
  // are replaced from new_k / new_v when the tile is staged.  Loads are unconditional
  // (clamped row) so that they batch; the first two tiles are requested before the RoPE
  // prologues so their HBM latency hides under them.
  const bf16_t *kbase = a.kc + (int64_t)kvh * a.cache_rows * 128;
  const bf16_t *vbase = a.vc + (int64_t)kvh * a.cache_rows * 128;
  constexpr int MAXC = 512 / nthr;  // 1, 2, 4 or 8 chunks per thread
  bf16x8 kA[MAXC], vA[MAXC], kB[MAXC], vB[MAXC];
  const int old_max = S > 0 ? S - 1 : 0;
  auto fetch = [&](bf16x8(&pk)[MAXC], bf16x8(&pv)[MAXC], int t) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * nthr;
      const int row = c >> 4, ch = c & 15;
      int key = t * 32 + row;
      key = key < old_max ? key : old_max;
      pk[i] = *reinterpret_cast<const bf16x8 *>(kbase + (int64_t)key * 128 + ch * 8);
      pv[i] = *reinterpret_cast<const bf16x8 *>(vbase + (int64_t)key * 128 + ch * 8);
    }
  };
  auto stage = [&](bf16x8(&pk)[MAXC], bf16x8(&pv)[MAXC], int t) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * nthr;
      const int row = c >> 4, ch = c & 15;
      int key = t * 32 + row;
      key = key < kv_len ? key : kv_len - 1;
      bf16x8 kk = pk[i], vv = pv[i];
      if (key >= S) {
        kk = *reinterpret_cast<const bf16x8 *>(&new_k[key - S][ch * 8]);
        vv = *reinterpret_cast<const bf16x8 *>(&new_v[key - S][ch * 8]);
      }
      *reinterpret_cast<bf16x8 *>(lds_k + row * 256 + ((ch ^ (row & 15)) << 4)) = kk;
      *reinterpret_cast<bf16x8 *>(lds_v + row * 256 + ((((ch >> 1) ^ (row & 7)) << 5) | ((ch & 1) << 4))) = vv;
    }
  };
  if (t0 < t1) fetch(kA, vA, t0);
  if (t0 + 1 < t1) fetch(kB, vB, t0 + 1);

  // ---- phase 0: the 16 q rows of this wave's head: 4 passes x 4 rows, all loads in flight
  {
    const int rsub = l >> 4, d0 = (l & 15) * 8;
    int brow[4], col[4], pos[4];
    bool rp[4];
    bf16x8 ov[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int j = p * 4 + rsub;
      brow[p] = j < bs ? a.blk_row0 + j : -1;
      col[p] = a.q_col + head * 128;
      pos[p] = pos0 + tau + j;
      rp[p] = true;
    }
    rope_rows<4>(a, brow, col, pos, rp, a.q_w, l, ov);
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int j = p * 4 + rsub;
      *reinterpret_cast<bf16x8 *>(&q_lds[wv][j][d0]) = j < bs ? ov[p] : z;
    }
  }
  STAMP(1);
  // ---- phase 1 (last split): new K / V rows of this kv head; items 0..n_new-1 are K rows,
  // n_new..2n_new-1 V rows; a sweep of NP passes covers 4*NP*G items.  Two passes suffice
  // for the target verify (16 new rows), the draft (tau + 16 rows) takes four.
  if (is_last_split) {
    const int rsub = l >> 4, d0 = (l & 15) * 8;
    auto sweep = [&](auto np_tag, int base) {
      constexpr int NP = decltype(np_tag)::value;
      int brow[NP], col[NP], pos[NP], rel[NP];
      bool rp[NP], isv[NP];
      bf16x8 ov[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int it = base + (p * 4 + rsub) * G + wv;
        isv[p] = it >= n_new;
        rel[p] = isv[p] ? it - n_new : it;
        const bool ok = it < 2 * n_new;
        brow[p] = !ok ? -1 : (rel[p] < tau ? a.ctx_row0 + rel[p] : a.blk_row0 + (rel[p] - tau));
        col[p] = (isv[p] ? a.v_col : a.k_col) + kvh * 128;
        pos[p] = pos0 + rel[p];
        rp[p] = !isv[p];
      }
      rope_rows<NP>(a, brow, col, pos, rp, a.k_w, l, ov);
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if (brow[p] >= 0) {
          *reinterpret_cast<bf16x8 *>(isv[p] ? &new_v[rel[p]][d0] : &new_k[rel[p]][d0]) = ov[p];
          const int crow = S + rel[p];
          if (crow < a.cache_rows)
            *reinterpret_cast<bf16x8 *>((isv[p] ? a.vc : a.kc) + ((int64_t)kvh * a.cache_rows + crow) * 128 + d0) =
                ov[p];
        }
    };
    if (2 * n_new <= 8 * G) {
      sweep(std::integral_constant<int, 2>{}, 0);
    } else {
      for (int base = 0; base < 2 * n_new; base += 16 * G) sweep(std::integral_constant<int, 4>{}, base);
    }
  }
  __syncthreads();
  STAMP(2);

  const int qi = l & 15, g = l >> 4;
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8 *>(&q_lds[wv][qi][s * 32 + g * 8]);

  f32x4 o[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // ---- phase 2
  auto compute_tile = [&](int t) {
    f32x4 sc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      sc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int row = u * 16 + qi;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ch = s * 4 + g;
        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(lds_k + row * 256 + ((ch ^ qi) << 4));
        sc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], sc[u], 0, 0, 0);
      }
    }
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
  };

  for (int t = t0; t < t1; t += 2) {
    stage(kA, vA, t);
    __syncthreads();
    if (t + 2 < t1) fetch(kA, vA, t + 2);
    compute_tile(t);
    __syncthreads();
    if (t + 1 < t1) {
      stage(kB, vB, t + 1);
      __syncthreads();
      if (t + 3 < t1) fetch(kB, vB, t + 3);
      compute_tile(t + 1);
      __syncthreads();
    }
  }
  STAMP(3);

  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  bf16_t *outp = reinterpret_cast<bf16_t *>(a.out_frag);

  if (nsplit == 1) {  // lane (q, g): o[dt][r] = O[q][dt*16 + 4g + r]
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const int col = head * 128 + dt * 16 + 4 * g;
      bf16x4 w4 = {f2bf(o[dt][0] * inv), f2bf(o[dt][1] * inv), f2bf(o[dt][2] * inv), f2bf(o[dt][3] * inv)};
      *reinterpret_cast<bf16x4 *>(outp + ((int64_t)(col >> 3) * 16 + qi) * 8 + (col & 7)) = w4;
    }
    return;
  }

  // ---- partials, then the last arriver of this kv head merges
  {
    float *op = a.o_part + (((int64_t)split * a.n_q + head) * 16 + qi) * 128;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4 *>(op + dt * 16 + 4 * g) = o[dt];
    if (g == 0) {
      float *ml = a.ml_part + (((int64_t)split * a.n_q + head) * 16 + qi) * 2;
      ml[0] = m_run;
      ml[1] = l_tot;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  STAMP(4);
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ticket = __hip_atomic_fetch_add(&a.tickets[kvh], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == nsplit - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&a.tickets[kvh], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    s_last = last;
  }
  __syncthreads();
  STAMP(5);
  if (!s_last) return;

  // thread (q, 8-column group) merges NH heads of the group per pass
  {
    constexpr int NH = G >= 4 ? 4 : G;                 // heads per thread per pass
    constexpr int NG = nthr >= 256 ? nthr / 256 : 1;   // head batches handled side by side
    const int q = (tid >> 4) & 15, dg = tid & 15;
    if (nthr >= 256) {
      for (int hb = (tid >> 8) * NH; hb < G; hb += NG * NH) {
        bf16x8 r[NH];
        merge_items<NH>(a.o_part, a.ml_part, nsplit, a.n_q, kvh * G + hb, q, dg, r);
#pragma unroll
        for (int h = 0; h < NH; ++h) a.out_frag[((kvh * G + hb + h) * 16 + dg) * 16 + q] = r[h];
      }
    } else {  // G = 1 or 2: fewer than 256 threads, each covers several (q, dg) pairs
      for (int it = tid; it < 256; it += nthr) {
        bf16x8 r[NH];
        merge_items<NH>(a.o_part, a.ml_part, nsplit, a.n_q, kvh * G, it >> 4, it & 15, r);
#pragma unroll
        for (int h = 0; h < NH; ++h) a.out_frag[((kvh * G + h) * 16 + (it & 15)) * 16 + (it >> 4)] = r[h];
      }
    }
  }
  STAMP(6);
}

}  // namespace

#ifdef DFL_ATTN_STAMPS
extern "C" int dfl_debug_read_stamps(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
}
#endif

extern "C" int64_t dfl_attn_fused_ws_bytes(int n_q, int n_kv, int max_splits) {
  return dfl_attn_ws_bytes(n_q, max_splits) + (int64_t)n_kv * sizeof(int) + 64;
}

namespace {
int attn_fused_launch(const float *qkv, int nsplit, int64_t split_stride, int ld, int q_col, int k_col, int v_col,
                      int ctx_row0, int blk_row0, int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w,
                      float eps, const void *cos_tab, const void *sin_tab, int max_pos, void *kcache, void *vcache,
                      int cache_rows, float scale, int causal, const int32_t *dyn, int kv_len_max, void *ws,
                      int max_splits, void *out_frag, int R, int req_rows, int64_t cache_req_stride,
                      int64_t out_req_stride, void *stream) {
  DFL_REQUIRE(qkv && cos_tab && sin_tab && kcache && vcache && dyn && out_frag && ws, "dfl_attn_fused: null pointer");
  DFL_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), "dfl_attn_fused: give both norm weights or neither");
  DFL_REQUIRE(n_q > 0 && n_kv > 0 && n_q % n_kv == 0, "dfl_attn_fused: bad head counts");
  DFL_REQUIRE(nsplit >= 1 && ld > 0 && q_col >= 0 && k_col >= 0 && v_col >= 0 && blk_row0 >= 0 && max_pos > 0,
              "dfl_attn_fused: bad layout");
  DFL_REQUIRE(kv_len_max > 0 && kv_len_max <= cache_rows && max_splits >= 1, "dfl_attn_fused: bad kv_len_max/max_splits");
  DFL_REQUIRE(R >= 1 && R <= 64, "dfl_attn_fused: R outside 1..64");
  // Key splits: a workgroup's tile loop costs ~1.45 us per 32-key tile, the last arriver's merge
  // ~0.6 us per split (scripts/dbg_attn_stamps.py), so the stage is shortest for ns ~ sqrt(tiles):
  // 8 splits at 1k keys (as with the former tiles/4 rule), 15 instead of 32 at 4k keys, where the
  // stage took 47 us with the linear rule and 35-38 us with this one (constants 1.1-1.55 measure
  // the same within noise).
  const int ntiles = (kv_len_max + 31) / 32;
#ifndef DFL_ATTN_NS_C  // swept in scripts/dbg_attn_stamps.py (-DDFL_ATTN_NS_C=...)
#define DFL_ATTN_NS_C 1.35f
#endif
  int ns = (int)(DFL_ATTN_NS_C * sqrtf((float)ntiles) + 0.5f);
  ns = ns > ntiles ? ntiles : ns;
  ns = ns < 1 ? 1 : (ns > max_splits ? max_splits : ns);
  FusedAttnArgs a{};
  a.qkv = qkv;
  a.nsplit_k = nsplit;
  a.split_stride = split_stride;
  a.ld = ld;
  a.q_col = q_col;
  a.k_col = k_col;
  a.v_col = v_col;
  a.ctx_row0 = ctx_row0;
  a.blk_row0 = blk_row0;
  a.q_w = (const bf16_t *)q_norm_w;
  a.k_w = (const bf16_t *)k_norm_w;
  a.eps = eps;
  a.cos_tab = (const bf16_t *)cos_tab;
  a.sin_tab = (const bf16_t *)sin_tab;
  a.max_pos = max_pos;
  a.kc = (bf16_t *)kcache;
  a.vc = (bf16_t *)vcache;
  a.cache_rows = cache_rows;
  a.n_q = n_q;
  a.n_kv = n_kv;
  a.G = n_q / n_kv;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.causal = causal ? 1 : 0;
  a.dyn = dyn;
  a.out_frag = (bf16x8 *)out_frag;
  // workspace: [R] o_part | [R] ml_part | [R][n_kv] tickets
  a.opart_req_stride = (int64_t)max_splits * n_q * 16 * 128;
  a.ml_req_stride = (int64_t)max_splits * n_q * 16 * 2;
  a.o_part = (float *)ws;
  a.ml_part = (float *)ws + R * a.opart_req_stride;
  a.tickets = (int *)((char *)ws + R * dfl_attn_ws_bytes(n_q, max_splits));
  a.req_rows = req_rows;
  a.cache_req_stride = cache_req_stride;
  a.out_req_stride = out_req_stride / 8;
  const dim3 grid(n_kv, ns, R);
  hipStream_t st = (hipStream_t)stream;
  switch (a.G) {
    case 1: hipLaunchKernelGGL(k_attn_fused<1>, grid, dim3(64), 0, st, a); break;
    case 2: hipLaunchKernelGGL(k_attn_fused<2>, grid, dim3(128), 0, st, a); break;
    case 4: hipLaunchKernelGGL(k_attn_fused<4>, grid, dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL(k_attn_fused<8>, grid, dim3(512), 0, st, a); break;
    default: DFL_REQUIRE(false, "dfl_attn_fused: GQA group %d not in {1,2,4,8}", a.G);
  }
  DFL_CHECK_LAUNCH("dfl_attn_fused");
  return DFL_OK;
}
}  // namespace

extern "C" int dfl_attn_fused(const float *qkv, int nsplit, int64_t split_stride, int ld, int q_col, int k_col,
                              int v_col, int ctx_row0, int blk_row0, int n_q, int n_kv, const void *q_norm_w,
                              const void *k_norm_w, float eps, const void *cos_tab, const void *sin_tab, int max_pos,
                              void *kcache, void *vcache, int cache_rows, float scale, int causal,
                              const int32_t *dyn, int kv_len_max, void *ws, int max_splits, void *out_frag,
                              void *stream) {
  return attn_fused_launch(qkv, nsplit, split_stride, ld, q_col, k_col, v_col, ctx_row0, blk_row0, n_q, n_kv, q_norm_w,
                           k_norm_w, eps, cos_tab, sin_tab, max_pos, kcache, vcache, cache_rows, scale, causal, dyn,
                           kv_len_max, ws, max_splits, out_frag, 1, 0, 0, 0, stream);
}

extern "C" int64_t dfl_attn_fused_batch_ws_bytes(int R, int n_q, int n_kv, int max_splits) {
  return R * (dfl_attn_ws_bytes(n_q, max_splits) + (int64_t)n_kv * sizeof(int)) + 64;
}

extern "C" int dfl_attn_fused_batch(const float *qkv, int nsplit, int64_t split_stride, int ld, int q_col, int k_col,
                                    int v_col, int blk_row0, int req_rows, int R, int n_q, int n_kv,
                                    const void *q_norm_w, const void *k_norm_w, float eps, const void *cos_tab,
                                    const void *sin_tab, int max_pos, void *kcache, void *vcache, int cache_rows,
                                    int64_t cache_req_stride, float scale, int causal, const int32_t *dyn,
                                    int kv_len_max, void *ws, int max_splits, void *out_frag, int64_t out_req_stride,
                                    void *stream) {
  DFL_REQUIRE(req_rows >= 16 && out_req_stride % 8 == 0 && cache_req_stride >= (int64_t)n_kv * cache_rows * 128,
              "dfl_attn_fused_batch: bad request strides");
  return attn_fused_launch(qkv, nsplit, split_stride, ld, q_col, k_col, v_col, 0, blk_row0, n_q, n_kv, q_norm_w,
                           k_norm_w, eps, cos_tab, sin_tab, max_pos, kcache, vcache, cache_rows, scale, causal, dyn,
                           kv_len_max, ws, max_splits, out_frag, R, req_rows, cache_req_stride, out_req_stride, stream);
}
