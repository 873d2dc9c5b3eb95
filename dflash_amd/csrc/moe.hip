// Sparse-MoE MLP of the target's verify forward (BASELINE configs[4]: Qwen3-Coder-30B-A3B; the reference calls the HF
// model, model/dflash.py:249-255; arithmetic: tf:models/qwen3_moe/modeling_qwen3_moe.py — Qwen3MoeTopKRouter.forward,
// Qwen3MoeExperts.forward).  For the <= 16 block rows of a verify:
//   k_moe_route   router logits [16][E] (bf16, the gate Linear's output) -> fp32 softmax -> top-k -> normalised bf16
//                 weights as a dense [16][E] matrix (0 = not routed), the ascending list of experts some row uses
//   (dfl_gemm_silu_mul_experts, gemm_skinny.hip: act_e = silu(x Wg_e^T) * (x Wu_e^T) for every active expert)
//   k_moe_down    out[m][n] = sum over active experts e of w[m][e] * (act_e[m] . Wd_e[n]) as fp32 K-part sums: one MFMA
//                 tile of 16 output columns per workgroup, the active experts dealt to its 16 waves, each wave scaling
//                 its expert's 16x16 product by the rows' routing weights before it adds it up.
// Rounding: HF rounds every expert's down-projection to bf16, scales it in bf16 and accumulates the <= k terms of a
// row in bf16 (index_add_ in expert order); here the sum over experts stays in fp32 and is rounded once, where the
// dense MLP's Linear output is rounded — fewer roundings than the reference, inside the stated bf16 tolerance.
#include "gemm_rows.h"
#include "moe_route.h"

namespace {

// one wavefront per row (16 rows), E <= 256
__global__ __launch_bounds__(1024) void k_moe_route(const bf16_t *logits, int ld, int E, int top_k, int norm_topk,
                                                    bf16_t *wt, int32_t *active, int32_t *list, int32_t *n_active,
                                                    const int32_t *dyn, int dyn_word) {
  const int tid = threadIdx.x, m = tid >> 6, l = tid & 63;
  const int nv = dyn ? dyn[dyn_word] : 16;
  for (int e = tid; e < E; e += 1024) active[e] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = l + 64 * j;
    if (e < E) wt[(int64_t)m * E + e] = (bf16_t)0.f;
  }
  float sel_v[8], tot;
  int sel_i[8];
  route_row(logits + (int64_t)m * ld, E, top_k, l, sel_v, sel_i, tot);
  if (m < nv && l == 0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (r < top_k) {
        const float w = norm_topk ? sel_v[r] / tot : sel_v[r];
        wt[(int64_t)m * E + sel_i[r]] = f2bf(w);
        active[sel_i[r]] = 1;
      }
    }
  }
  __syncthreads();
  // ascending list of the active experts (wave 0)
  if (m == 0) {
    int base = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int e = e0 + l;
      const bool a = e < E && active[e] != 0;
      const unsigned long long b = __ballot(a);
      if (a) list[base + __popcll(b & ((1ull << l) - 1ull))] = e;
      base += __popcll(b);
    }
    if (l == 0) *n_active = base;
  }
}

struct MoeDownArgs {
  const bf16x8 *wd;     // [E][ntiles][KSe][64]
  int64_t wd_stride;    // bf16x8 units per expert
  const bf16x8 *act;    // [E] frag16 [KSe][64]
  int64_t act_stride;   // bf16x8 units per expert
  const bf16_t *wt;     // [16][E]
  const int32_t *list, *n_active;
  int E, KSe, ntiles;
  float *out;           // [nsplit][16][ldo]
  int ldo;
};

__global__ __launch_bounds__(1024) void k_moe_down(MoeDownArgs a) {
  __shared__ float red[16][256];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const int t = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const int n = *a.n_active;
  const int per = (n + nsplit - 1) / nsplit;
  const int p0 = split * per, p1 = min(n, p0 + per);
  // items = (expert of this split, 4-k-step chunk of its K), dealt round-robin to the 16 waves and double-buffered:
  // the next item's weights and activations are in flight while the current one is in the MFMA.  (First form: whole
  // experts dealt to the waves, 8 k-steps loaded, waited for and multiplied at a time: 2.56 experts per wave = 3 rounds
  // of 3 serial round trips, 4.7 TB/s at 82 active experts.)
  const int cpe = (a.KSe + 3) >> 2;
  const int nitems = (p1 > p0 ? p1 - p0 : 0) * cpe;
  f32x4 total = {0.f, 0.f, 0.f, 0.f};
  bf16x8 wA[4], xA[4], wB[4], xB[4];
  bf16_t rA = 0, rB = 0;  // (kept as loaded: converting at the load would make the wave wait for it there)
  // Every request in the loop is UNCONDITIONAL — past the wave's last item a dummy (empty descriptors: zeros, no
  // traffic): behind `if (more) issue` hipcc cannot count the loads in flight and waits vmcnt(0) before the MFMAs of
  // the current item, i.e. for the item it has just requested.
  auto issue = [&](bf16x8(&wv)[4], bf16x8(&xv)[4], bf16_t &wr, int it) {
    const bool live = it < nitems;
    it = live ? it : nitems - 1;
    const int e = a.list[p0 + it / cpe], ks0 = (it % cpe) * 4;
    int nf = a.KSe - ks0;
    // (scalar: a clamp compiled to v_med3 puts the buffer descriptors in VGPRs and every load into a waterfall loop)
    nf = __builtin_amdgcn_readfirstlane(live ? (nf > 4 ? 4 : nf) : 0);
    wr = a.wt[(int64_t)(l & 15) * a.E + e];  // the routing weight of this lane's row; first: loads return in order
    load_ksteps<4, 0>(xv, a.act + e * a.act_stride + (size_t)ks0 * 64, nf, l);  // re-read by every column tile: L2
    load_ksteps<4>(wv, a.wd + e * a.wd_stride + ((size_t)t * a.KSe + ks0) * 64, nf, l);  // past nf: zero fragments
  };
  auto consume = [&](const bf16x8(&wv)[4], const bf16x8(&xv)[4], bf16_t wraw) {
    const float wr = bf2f(wraw);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 4; ++f) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[f], xv[f], acc, 0, 0, 0);
    // D layout: lane L, register r = output column 4 (L >> 4) + r of row L & 15
    total += acc * wr;
  };
  int it = w;
  if (it < nitems) {
    issue(wA, xA, rA, it);
    for (; it < nitems; it += 32) {
      issue(wB, xB, rB, it + 16);
      __builtin_amdgcn_sched_barrier(0);  // (or hipcc sinks the requests below the current item's waits and MFMAs)
      consume(wA, xA, rA);
      __builtin_amdgcn_sched_barrier(0);
      issue(wA, xA, rA, it + 32);
      __builtin_amdgcn_sched_barrier(0);
      if (it + 16 < nitems) consume(wB, xB, rB);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  *reinterpret_cast<f32x4 *>(&red[w][l * 4]) = total;
  __syncthreads();
  if (tid < 256) {
    const int m = tid >> 4, nl = tid & 15;
    const int idx = 4 * (m + 16 * (nl >> 2)) + (nl & 3);
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) s += red[ww][idx];
    a.out[((size_t)split * 16 + m) * a.ldo + t * 16 + nl] = s;
  }
}

// Gate/up projection + SiLU of every ACTIVE expert for K <= 2048 (30B-A3B: K = 2048, I = 768): at this K a 16-column
// tile is only 64 KB, so the skinny GEMM's per-tile LDS meeting of its 16 waves (gemm_skinny.hip) weighs twice what
// it does at K = 4096.  Here an item is a PAIR (gate tile p, up tile p) of an active expert: every wave holds its 4
// k-steps of both tiles, the 16 waves meet ONCE per pair, and the finishing threads have gate and up sums side by
// side for the SiLU epilogue (tf:models/qwen3_moe/modeling_qwen3_moe.py Qwen3MoeExperts.forward; rounding points
// as in dfl_gemm_silu_mul).  One workgroup per CU walks its share of the (expert, pair) items, next item in flight.
struct MoeGuArgs {
  const bf16x8 *wp;      // experts' packed gate/up weights, back to back: [E][2*npp tiles][KS][64]
  const bf16x8 *xfrag;   // frag16 of the 16 normalised rows [KS][64]
  const int32_t *list, *n_active, *dyn;
  int valid_word;
  int KS, npp;           // K / 32 (<= 64), I / 16
  bf16_t *act;           // [E][16 * I] frag16 per expert
};

__global__ __launch_bounds__(1024) void k_moe_gate_up(MoeGuArgs a) {
  __shared__ float red[2][16][2][256];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  int nfw = a.KS - 4 * w;
  nfw = __builtin_amdgcn_readfirstlane(nfw < 0 ? 0 : (nfw > 4 ? 4 : nfw));
  const int nitems = a.n_active[0] * a.npp;
  const int64_t tile = (int64_t)a.KS * 64, expert = 2 * a.npp * tile;   // bf16x8 units
  bf16x8 gA[4], uA[4], gB[4], uB[4], xr[4];
  // (every request unconditional, a dummy past the end: see k_moe_down)
  auto issue = [&](bf16x8(&g)[4], bf16x8(&u)[4], int it) {
    const bool live = it < nitems;
    it = live ? it : nitems - 1;
    const int e = a.list[it / a.npp], p = it % a.npp;
    const bf16x8 *base = a.wp + e * expert + 2 * p * tile + 4 * w * 64;
    const int nf = __builtin_amdgcn_readfirstlane(live ? nfw : 0);
    load_ksteps<4>(g, base, nf, l);
    load_ksteps<4>(u, base + tile, nf, l);
  };
  int it = blockIdx.x;
  if (it >= nitems) return;
  load_ksteps<4, 0>(xr, a.xfrag + 4 * w * 64, nfw, l);   // activations first: a wave's loads return in issue order
  issue(gA, uA, it);
  const int nv = (a.dyn && a.valid_word >= 0) ? a.dyn[a.valid_word] : 16;
  if ((l & 15) >= nv) {
#pragma unroll
    for (int f = 0; f < 4; ++f) xr[f] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
  int pos = 0;
  auto consume = [&](const bf16x8(&g)[4], const bf16x8(&u)[4], int itc) {
    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 4; ++f) ag = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[f], xr[f], ag, 0, 0, 0);
#pragma unroll
    for (int f = 0; f < 4; ++f) au = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[f], xr[f], au, 0, 0, 0);
    const int buf = pos & 1;
    ++pos;
    *reinterpret_cast<f32x4 *>(&red[buf][w][0][l * 4]) = ag;
    *reinterpret_cast<f32x4 *>(&red[buf][w][1][l * 4]) = au;
    __syncthreads();
    if (tid < 256) {  // D layout of the MFMA as in gemm_skinny.hip: thread (row m, column nl) of the pair
      const int m = tid >> 4, nl = tid & 15;
      const int idx = 4 * (m + 16 * (nl >> 2)) + (nl & 3);
      float sg = 0.f, su = 0.f;
#pragma unroll
      for (int ww = 0; ww < 16; ++ww) {
        sg += red[buf][ww][0][idx];
        su += red[buf][ww][1][idx];
      }
      const float gb = rbf(sg), ub = rbf(su);   // the Linears' bf16 outputs
      const float act = rbf(gb / (1.f + __expf(-gb)));
      const int e = a.list[itc / a.npp], n = (itc % a.npp) * 16 + nl;
      a.act[(int64_t)e * (16 * 16 * a.npp) + ((size_t)(n >> 3) * 16 + m) * 8 + (n & 7)] = f2bf(act * ub);
    }
  };
  const int G = gridDim.x;
  for (; it < nitems; it += 2 * G) {
    issue(gB, uB, it + G);
    __builtin_amdgcn_sched_barrier(0);
    consume(gA, uA, it);
    __builtin_amdgcn_sched_barrier(0);
    issue(gA, uA, it + 2 * G);
    __builtin_amdgcn_sched_barrier(0);
    if (it + G < nitems) consume(gB, uB, it + G);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace

extern "C" int dfl_moe_gate_up(const void *wp_gateup, const void *x_frag, int E, int I, int K, void *act_frag,
                               const int32_t *list, const int32_t *n_active, const int32_t *dyn, int valid_word,
                               void *stream) {
  DFL_REQUIRE(wp_gateup && x_frag && act_frag && list && n_active, "dfl_moe_gate_up: null pointer");
  DFL_REQUIRE(E >= 1 && E <= 1024 && I > 0 && I % 16 == 0 && K > 0 && K % 32 == 0 && K <= 2048,
              "dfl_moe_gate_up: E=%d I=%d K=%d outside range (K <= 2048; larger K: dfl_gemm_silu_mul_experts)", E, I, K);
  MoeGuArgs a{};
  a.wp = (const bf16x8 *)wp_gateup;
  a.xfrag = (const bf16x8 *)x_frag;
  a.list = list;
  a.n_active = n_active;
  a.dyn = dyn;
  a.valid_word = valid_word;
  a.KS = K / 32;
  a.npp = I / 16;
  a.act = (bf16_t *)act_frag;
  hipLaunchKernelGGL(k_moe_gate_up, dim3(256), dim3(1024), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_moe_gate_up");
  return DFL_OK;
}

extern "C" int dfl_moe_route(const void *logits, int ld, int E, int top_k, int norm_topk, void *wt, int32_t *active,
                             int32_t *list, int32_t *n_active, const int32_t *dyn, int dyn_word, void *stream) {
  DFL_REQUIRE(logits && wt && active && list && n_active, "dfl_moe_route: null pointer");
  DFL_REQUIRE(E >= 1 && E <= 256 && ld >= E && top_k >= 1 && top_k <= 8 && top_k <= E, "dfl_moe_route: E=%d top_k=%d outside range",
              E, top_k);
  hipLaunchKernelGGL(k_moe_route, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const bf16_t *)logits, ld, E, top_k,
                     norm_topk ? 1 : 0, (bf16_t *)wt, active, list, n_active, dyn, dyn_word);
  DFL_CHECK_LAUNCH("dfl_moe_route");
  return DFL_OK;
}

extern "C" int dfl_moe_down(const void *wp_down, int64_t wp_expert_stride, const void *act_frag, int64_t act_expert_stride,
                            const void *wt, const int32_t *list, const int32_t *n_active, int E, int N, int I, int nsplit,
                            float *out, void *stream) {
  DFL_REQUIRE(wp_down && act_frag && wt && list && n_active && out, "dfl_moe_down: null pointer");
  DFL_REQUIRE(E >= 1 && N > 0 && I > 0 && N % 16 == 0 && I % 32 == 0 && nsplit >= 1 && nsplit <= 16, "dfl_moe_down: bad shape");
  DFL_REQUIRE(wp_expert_stride >= (int64_t)N * I && wp_expert_stride % 8 == 0 && act_expert_stride >= (int64_t)16 * I &&
                  act_expert_stride % 8 == 0,
              "dfl_moe_down: expert strides too short");
  MoeDownArgs a{};
  a.wd = (const bf16x8 *)wp_down;
  a.wd_stride = wp_expert_stride / 8;
  a.act = (const bf16x8 *)act_frag;
  a.act_stride = act_expert_stride / 8;
  a.wt = (const bf16_t *)wt;
  a.list = list;
  a.n_active = n_active;
  a.E = E;
  a.KSe = I / 32;
  a.ntiles = N / 16;
  a.out = out;
  a.ldo = N;
  hipLaunchKernelGGL(k_moe_down, dim3(N / 16, nsplit), dim3(1024), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_moe_down");
  return DFL_OK;
}
