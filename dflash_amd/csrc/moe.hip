// Sparse-MoE MLP of the target's verify forward (BASELINE configs[4]: Qwen3-Coder-30B-A3B; the reference calls the HF
// model, model/dflash.py:249-255; arithmetic: tf:models/qwen3_moe/modeling_qwen3_moe.py — Qwen3MoeTopKRouter.forward,
// Qwen3MoeExperts.forward).  For the <= 16 block rows of a verify:
//   k_moe_route   router logits [16][E] (bf16, the gate Linear's output) -> fp32 softmax -> top-k -> normalised bf16
//                 weights as a dense [16][E] matrix (0 = not routed), the ascending list of experts some row uses
//   k_moe_router  the block's RMSNorm + the gate Linear + that routing in one launch (round 4)
//   (dfl_gemm_silu_mul_experts, gemm_skinny.hip: act_e = silu(x Wg_e^T) * (x Wu_e^T) for every active expert)
//   k_moe_down    out[m][n] = sum over active experts e of w[m][e] * (act_e[m] . Wd_e[n]) as fp32 K-part sums: one MFMA
//                 tile of 16 output columns per workgroup, the active experts dealt to its 16 waves, each wave scaling
//                 its expert's 16x16 product by the rows' routing weights before it adds it up.
// Rounding: HF rounds every expert's down-projection to bf16, scales it in bf16 and accumulates the <= k terms of a
// row in bf16 (index_add_ in expert order); here the sum over experts stays in fp32 and is rounded once, where the
// dense MLP's Linear output is rounded — fewer roundings than the reference, inside the stated bf16 tolerance.
#include "gemm_rows.h"
#include "moe_route.h"
#include <limits.h>
#include <stdlib.h>

namespace {

#ifdef DFL_MOE_STAMPS  // diagnostic build only (scripts/dbg_moe_router_stamps.py): 100 MHz wall stamps of every workgroup
__device__ unsigned long long g_rstamps[16][12];
#define RSTAMP(i)                                                                            \
  do {                                                                                       \
    if (threadIdx.x == 0) g_rstamps[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime();       \
  } while (0)
#else
#define RSTAMP(i)
#endif

// workgroup barrier for LDS traffic only (__syncthreads() also waits for every global store in flight: ~1 us each
// time behind the routing kernels' output stores)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// max / min over the 16 lanes of a DPP row, result in each of them
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  return v;
}
__device__ __forceinline__ long long row_max16_i64(long long v) {
#define DFL_STEP64(ctrl)                                                                                         \
  {                                                                                                              \
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, ctrl, 0xF, 0xF, true);              \
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)((unsigned long long)v >> 32), ctrl, 0xF, 0xF, true); \
    const long long o = (long long)(((unsigned long long)hi << 32) | lo);                                         \
    v = o > v ? o : v;                                                                                            \
  }
  DFL_STEP64(0xB1) DFL_STEP64(0x4E) DFL_STEP64(0x141) DFL_STEP64(0x140)
#undef DFL_STEP64
  return v;
}
__device__ __forceinline__ int row_min16(int v) {
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true));
  return v;
}

// Routing of the 16 rows by a workgroup of >= 4 waves (E <= 16 EJ): a QUARTER wave per row — the 16 lanes of a DPP row
// own experts q, q + 16, ... of row 4 w + (l >> 4), so every reduction of the softmax and of the k selection rounds is
// four DPP steps inside the row, for four rows per instruction.  (First form: one wavefront per row, 16 waves = four per
// SIMD taking turns at ~700 instructions each: 6.8 us of the launch by its stamps, scripts/dbg_moe_router_stamps.py.)
// Arithmetic of route_row (moe_route.h): fp32 softmax of the bf16 logits, k rounds of (largest probability, lowest
// expert), optional renormalisation by the sum taken in selection order; a NaN row gives experts 0 .. k-1 NaN weights.
// lg: the 16 rows' logits, row stride ld (LDS or global).  The active flags meet in LDS (s_act[256]); nothing waits
// for a global store.
template <int EJ>
__device__ __forceinline__ void route_rows(const bf16_t *lg, int ld, int E, int top_k, int norm_topk, int nv, bf16_t *wt,
                                           int32_t *active, int32_t *list, int32_t *n_active, int *s_act) {
  __shared__ float s_sv[16 * 8];
  __shared__ int s_si[16 * 8];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  if (tid < 256) s_act[tid] = 0;
  lds_barrier();
  RSTAMP(8);
  if (w < 4) {
    const int row = 4 * w + (l >> 4), q = l & 15;
    // BRANCH-FREE throughout: one wave alone on its SIMD pays every exec-mask branch in full (the first form's
    // `if (e < E && p > best)` per expert cost 16 branches per round: 0.37 us per round by the stamps)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(wt, 0, 16 * E * (int)sizeof(bf16_t), 0x00020000);
    float p[EJ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < EJ; ++j) {
      const int e = q + 16 * j;
      const int ec = e < E ? e : 0;  // (clamped: no branch around the load)
      const float v = bf2f(lg[(int64_t)row * ld + ec]);
      p[j] = e < E ? v : -INFINITY;
      // zero the row's dense weights (past E: an offset beyond the descriptor, dropped)
      __builtin_amdgcn_raw_buffer_store_b16((unsigned short)0, wr, e < E ? (row * E + e) * (int)sizeof(bf16_t) : 0x40000000, 0, 0);
      mx = fmaxf(mx, p[j]);
    }
    mx = row_max16(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < EJ; ++j) {
      p[j] = (q + 16 * j) < E ? __expf(p[j] - mx) : 0.f;
      sum += p[j];
    }
    sum = row_sum16(sum);
    // One sortable 64-bit key per expert: the probability's bits (>= 0: they order as integers; a NaN canonical and above
    // everything) over INT_MAX - expert, so that a plain integer max IS "largest probability, lowest expert" — a round
    // is 7 maxima in the lane, 4 DPP steps in the row and 8 compares to retire the winner (~70 instructions; the first
    // forms — (value, index) pairs through compare / select trees — were ~250 per round at 4 cycles each for a wave
    // alone on its SIMD: 0.45 us per round by the stamps).  A NaN row selects experts 0 .. k-1 with NaN weights, as
    // route_row does.
    long long key[EJ];
#pragma unroll
    for (int j = 0; j < EJ; ++j) {
      const int e = q + 16 * j;
      const float pj = p[j] / sum;  // softmax(dtype = float)
      const int bits = pj == pj ? __float_as_int(pj) : 0x7fc00000;
      key[j] = e < E ? (long long)(((unsigned long long)(unsigned)bits << 32) | (unsigned)(0x7fffffff - e)) : LLONG_MIN;
    }
    RSTAMP(9);
    float tot = 0.f;
#pragma unroll 1  // (a ROLLED loop: the launch runs this code once, from a cold instruction cache)
    for (int r = 0; r < top_k; ++r) {
      long long best = key[0];
#pragma unroll
      for (int j = 1; j < EJ; ++j) best = key[j] > best ? key[j] : best;
      best = row_max16_i64(best);
      const float bv = __int_as_float((int)(best >> 32));
      tot += bv;  // (selection order, as route_row sums it)
      s_sv[row * 8 + r] = bv;  // (all 16 lanes of the row: the same value to the same word)
      s_si[row * 8 + r] = 0x7fffffff - (int)(unsigned)best;
#pragma unroll
      for (int j = 0; j < EJ; ++j) key[j] = key[j] == best ? LLONG_MIN : key[j];  // taken
    }
    RSTAMP(10);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the row's own lane 0 wrote them: same wave, in order)
    if (row < nv && q < top_k) {  // lane q writes selection q (behind this wave's own zero stores of the row: in order)
      const float v = s_sv[row * 8 + q];
      const int e = s_si[row * 8 + q];
      wt[(int64_t)row * E + e] = f2bf(norm_topk ? v / tot : v);
      s_act[e] = 1;
    }
  }
  lds_barrier();
  RSTAMP(11);
  for (int e = tid; e < E; e += (int)blockDim.x) active[e] = s_act[e];
  if (w == 0) {  // ascending list of the active experts
    int base = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int e = e0 + l;
      const bool on = e < E && s_act[e] != 0;
      const unsigned long long bb = __ballot(on);
      if (on) list[base + __popcll(bb & ((1ull << l) - 1ull))] = e;
      base += __popcll(bb);
    }
    if (l == 0) *n_active = base;
  }
}

__global__ __launch_bounds__(256) void k_moe_route(const bf16_t *logits, int ld, int E, int top_k, int norm_topk,
                                                   bf16_t *wt, int32_t *active, int32_t *list, int32_t *n_active,
                                                   const int32_t *dyn, int dyn_word) {
  __shared__ int s_act[256];
  const int nv = dyn ? dyn[dyn_word] : 16;
  if (E <= 128)
    route_rows<8>(logits, ld, E, top_k, norm_topk, nv, wt, active, list, n_active, s_act);
  else
    route_rows<16>(logits, ld, E, top_k, norm_topk, nv, wt, active, list, n_active, s_act);
}

// ---- RMSNorm + router GEMM + routing of the <= 16 rows in ONE launch (round 4).  The three launches it replaces
// (k_norm_pack 4.5 us, the router's 8-workgroup k_gemm 4.8 us, k_moe_route 8.7 us — 4.7 us with this round's
// routing code — plus two boundaries) are pure latency: 64 KB of rows, 512 KB of router weights, 4 KB of logits —
// 15 % of a 48-layer verify (BASELINE configs[4]).  Stamps of this launch at the 30B-A3B shape
// (profiles/r4_moe_router_stamps.txt, us from its start): rows + weights landed 3.7 · normalised, multiplied, met in
// LDS 5.9 · logits stored, ticket back 7.3 · (last workgroup) logits in LDS 8.9, softmax 10.2, k rounds 12.4,
// list 13.4; 12.8 us by the kernel trace against 13.9 + two boundaries.
// Here workgroup b (one per 16-expert column tile of the gate Linear) normalises the rows itself while it builds its
// MFMA B operands (each wave its own k-steps, straight from the residual rows in fragment order; the row's sum of
// squares meets in LDS), writes its share of the normalised fragments for the expert GEMMs, multiplies with its tile
// of the router weights, rounds the 16 x 16 logits to bf16 (the gate Linear's output) and hands them over: sc1
// 16-byte stores by wave 0, drained, the workgroup's barrier, ONE returning agent-scope add by one lane; the
// workgroup whose add came last loads all logits with sc1 4-byte loads behind its barrier
// (MI355X_MICROARCH.md, "Valid forms", table row 1: the last arriver is told by the returned value; no spinning,
// nothing to deadlock) and routes as k_moe_route does (route_rows above).
struct MoeRouterArgs {
  const bf16_t *h;  // [16][ldh] residual rows to normalise; null: xn holds the normalised fragments already
  int64_t ldh;
  const bf16_t *nw;
  float eps;
  bf16x8 *xn;        // frag16 [KS][64]: written (h != null) or read
  const bf16x8 *wr;  // packed router weights [Ep/16][KS][64]
  int KS, E, top_k, norm_topk;
  bf16_t *rlog;  // [16][ld] bf16 logits (the hand-off buffer; a test reads it)
  int ld;
  bf16_t *wt;
  int32_t *active, *list, *n_active;
  const int32_t *dyn;
  int dyn_word;
  int *ticket;  // zero between launches
};

template <int FR, bool NORM>
__global__ __launch_bounds__(1024) void k_moe_router(MoeRouterArgs a) {
  __shared__ float red[16][256];
  __shared__ float ssw[16][16];
  __shared__ __attribute__((aligned(16))) bf16_t tile[16][16];
  __shared__ int s_last;
  __shared__ int s_act[256];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const int b = blockIdx.x, nwg = gridDim.x;
  const int m = l & 15, kq = l >> 4;
  const int ks0 = w * FR;
  RSTAMP(0);
  int nf = a.KS - ks0;
  nf = __builtin_amdgcn_readfirstlane(nf < 0 ? 0 : (nf > FR ? FR : nf));
  bf16x8 wv[FR], x[FR];
  // the router weights first (K <= 2048); with 8 k-steps per wave they would not fit beside the rows being normalised
  // (128 VGPRs at 1024 threads) and are asked for behind the normalisation instead
  if constexpr (!(NORM && FR > 4)) load_ksteps<FR>(wv, a.wr + ((size_t)b * a.KS + ks0) * 64, nf, l);
  const int nv = a.dyn ? a.dyn[a.dyn_word] : 16;
  if (NORM) {
    bf16x8 g[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      const int ks = f < nf ? ks0 + f : 0;  // (clamped: no branch around a load; all 16 rows are readable memory)
      x[f] = *reinterpret_cast<const bf16x8 *>(a.h + (int64_t)m * a.ldh + ks * 32 + kq * 8);
      if constexpr (FR <= 4) g[f] = *reinterpret_cast<const bf16x8 *>(a.nw + ks * 32 + kq * 8);
    }
    float ss = 0.f;
#pragma unroll
    for (int f = 0; f < FR; ++f)
      if (f < nf)
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += bf2f(x[f][j]) * bf2f(x[f][j]);
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (kq == 0) ssw[w][m] = ss;
    __syncthreads();
  RSTAMP(1);
    float tot = 0.f;
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) tot += ssw[ww][m];
    const float rstd = rsqrtf(tot / (float)(a.KS * 32) + a.eps);  // Qwen3MoeRMSNorm: fp32 mean of squares
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      if constexpr (FR > 4)  // (K > 2048: the norm weights are asked for only now — 8 + 8 + 8 fragments do not fit 128 VGPRs)
        g[f] = *reinterpret_cast<const bf16x8 *>(a.nw + (f < nf ? ks0 + f : 0) * 32 + kq * 8);
      const bool live = f < nf && m < nv;  // absent rows: zero fragments
#pragma unroll
      for (int j = 0; j < 8; ++j) x[f][j] = live ? f2bf(bf2f(g[f][j]) * rbf(bf2f(x[f][j]) * rstd)) : (bf16_t)0.f;
    }
  } else {
    load_ksteps<FR, 0>(x, a.xn + (size_t)ks0 * 64, nf, l);
  }
  if constexpr (NORM && FR > 4) {  // (K > 2048: the fragments leave at once; kept through the routing they spill)
    load_ksteps<FR>(wv, a.wr + ((size_t)b * a.KS + ks0) * 64, nf, l);
#pragma unroll
    for (int f = 0; f < FR; ++f)
      if (f < nf && (ks0 + f) % nwg == b) a.xn[(size_t)(ks0 + f) * 64 + l] = x[f];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < FR; ++f) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[f], x[f], acc, 0, 0, 0);
  *reinterpret_cast<f32x4 *>(&red[w][l * 4]) = acc;
  __syncthreads();
  RSTAMP(2);
  if (tid < 256) {  // D layout: lane L, register r = column 4 (L >> 4) + r of row L & 15
    const int mm = tid >> 4, nl = tid & 15;
    const int idx = 4 * (mm + 16 * (nl >> 2)) + (nl & 3);
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) s += red[ww][idx];
    tile[mm][nl] = f2bf(s);
  }
  __syncthreads();
  RSTAMP(3);
  if (tid < 32) {  // (wave 0) row tid >> 1, experts 16 b + 8 (tid & 1) .. + 7: one 16-byte write-through store
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc(a.rlog, 0, 16 * a.ld * (int)sizeof(bf16_t), 0x00020000);
    const u32x4 v = *reinterpret_cast<const u32x4 *>(&tile[tid >> 1][8 * (tid & 1)]);
    __builtin_amdgcn_raw_buffer_store_b128(v, rr, ((tid >> 1) * a.ld + 16 * b + 8 * (tid & 1)) * (int)sizeof(bf16_t), 0, 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (only wave 0 has stores in flight: the logits)
  RSTAMP(4);
  lds_barrier();
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == nwg - 1;
    if (last) __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
    s_last = last;
  }
  RSTAMP(5);
  // this workgroup's share of the normalised fragments (the expert GEMMs' source) leaves LAST: in front of the hand-off
  // or of the logits' loads its stores would lengthen their waits (a wave's vmcnt counts them all)
  auto store_xn = [&]() {
    if (NORM && FR <= 4) {
#pragma unroll
      for (int f = 0; f < FR; ++f)
        if (f < nf && (ks0 + f) % nwg == b) a.xn[(size_t)(ks0 + f) * 64 + l] = x[f];
    }
  };
  lds_barrier();
  if (!s_last) {
    store_xn();
    return;
  }
  // ---- the last workgroup to arrive routes: every logit through an sc1 load into LDS, then k_moe_route's body
  bf16_t *lg = reinterpret_cast<bf16_t *>(&red[0][0]);  // [16][E] (E <= 256: 8 KB of the 16 KB)
  {
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc(a.rlog, 0, 16 * a.ld * (int)sizeof(bf16_t), 0x00020000);
    const int half = a.E >> 1;  // pairs of experts per row (E even)
    for (int i = tid; i < 16 * half; i += 1024) {
      const int mm = i / half, p = i - mm * half;
      const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rr, (mm * a.ld + 2 * p) * (int)sizeof(bf16_t), 0, 16);
      *reinterpret_cast<unsigned *>(&lg[mm * a.E + 2 * p]) = v;
    }
  }
  lds_barrier();
  RSTAMP(6);
  if (a.E <= 128)
    route_rows<8>(lg, a.E, a.E, a.top_k, a.norm_topk, nv, a.wt, a.active, a.list, a.n_active, s_act);
  else
    route_rows<16>(lg, a.E, a.E, a.top_k, a.norm_topk, nv, a.wt, a.active, a.list, a.n_active, s_act);
  RSTAMP(7);
  store_xn();
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

// CT: column tiles per workgroup.  With CT = 1 every workgroup re-reads every expert's activations (24 KB per expert
// at I = 768: as many bytes from L2 as weight bytes from HBM); CT = 2 shares an item's activation fragments between two
// column tiles.
template <int CT, int NW>
__global__ __launch_bounds__(NW * 64) void k_moe_down(MoeDownArgs a) {
  __shared__ float red[NW][CT][256];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const int t0 = blockIdx.x * CT, split = blockIdx.y, nsplit = gridDim.y;
  const int n = *a.n_active;
  const int per = (n + nsplit - 1) / nsplit;
  const int p0 = split * per, p1 = min(n, p0 + per);
  // items = (expert of this split, 4-k-step chunk of its K), dealt round-robin to the 16 waves and double-buffered:
  // the next item's weights and activations are in flight while the current one is in the MFMA.  (First form: whole
  // experts dealt to the waves, 8 k-steps loaded, waited for and multiplied at a time: 2.56 experts per wave = 3 rounds
  // of 3 serial round trips, 4.7 TB/s at 82 active experts.)
  const int cpe = (a.KSe + 3) >> 2;
  const int nitems = (p1 > p0 ? p1 - p0 : 0) * cpe;
  f32x4 total[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) total[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 wA[CT][4], xA[4], wB[CT][4], xB[4];
  bf16_t rA = 0, rB = 0;  // (kept as loaded: converting at the load would make the wave wait for it there)
  // Every request in the loop is UNCONDITIONAL — past the wave's last item a dummy (empty descriptors: zeros, no
  // traffic): behind `if (more) issue` hipcc cannot count the loads in flight and waits vmcnt(0) before the MFMAs of
  // the current item, i.e. for the item it has just requested.
  auto issue = [&](bf16x8(&wv)[CT][4], bf16x8(&xv)[4], bf16_t &wr, int it) {
    const bool live = it < nitems;
    it = live ? it : nitems - 1;
    const int e = a.list[p0 + it / cpe], ks0 = (it % cpe) * 4;
    int nf = a.KSe - ks0;
    // (scalar: a clamp compiled to v_med3 puts the buffer descriptors in VGPRs and every load into a waterfall loop)
    nf = __builtin_amdgcn_readfirstlane(live ? (nf > 4 ? 4 : nf) : 0);
    wr = a.wt[(int64_t)(l & 15) * a.E + e];  // the routing weight of this lane's row; first: loads return in order
    load_ksteps<4, 0>(xv, a.act + e * a.act_stride + (size_t)ks0 * 64, nf, l);  // re-read by every column group: L2
#pragma unroll
    for (int c = 0; c < CT; ++c)
      load_ksteps<4>(wv[c], a.wd + e * a.wd_stride + ((size_t)(t0 + c) * a.KSe + ks0) * 64, nf, l);  // past nf: zero fragments
  };
  auto consume = [&](const bf16x8(&wv)[CT][4], const bf16x8(&xv)[4], bf16_t wraw) {
    const float wr = bf2f(wraw);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int f = 0; f < 4; ++f) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[c][f], xv[f], acc, 0, 0, 0);
      // D layout: lane L, register r = output column 4 (L >> 4) + r of row L & 15
      total[c] += acc * wr;
    }
  };
  int it = w;
  if (it < nitems) {
    issue(wA, xA, rA, it);
    for (; it < nitems; it += 2 * NW) {
      issue(wB, xB, rB, it + NW);
      __builtin_amdgcn_sched_barrier(0);  // (or hipcc sinks the requests below the current item's waits and MFMAs)
      consume(wA, xA, rA);
      __builtin_amdgcn_sched_barrier(0);
      issue(wA, xA, rA, it + 2 * NW);
      __builtin_amdgcn_sched_barrier(0);
      if (it + NW < nitems) consume(wB, xB, rB);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) *reinterpret_cast<f32x4 *>(&red[w][c][l * 4]) = total[c];
  __syncthreads();
  for (int o = tid; o < 256 * CT; o += NW * 64) {
    const int c = o >> 8, m = (o >> 4) & 15, nl = o & 15;
    const int idx = 4 * (m + 16 * (nl >> 2)) + (nl & 3);
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) s += red[ww][c][idx];
    a.out[((size_t)split * 16 + m) * a.ldo + (t0 + c) * 16 + nl] = s;
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
  hipLaunchKernelGGL(k_moe_route, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)logits, ld, E, top_k,
                     norm_topk ? 1 : 0, (bf16_t *)wt, active, list, n_active, dyn, dyn_word);
  DFL_CHECK_LAUNCH("dfl_moe_route");
  return DFL_OK;
}

extern "C" int dfl_moe_router(const void *h, int64_t ldh, const void *norm_w, float eps, void *xn_frag, const void *wp_router,
                              int K, int E, int top_k, int norm_topk, void *rlog, int ld, void *wt, int32_t *active,
                              int32_t *list, int32_t *n_active, const int32_t *dyn, int dyn_word, int32_t *ticket,
                              void *stream) {
  DFL_REQUIRE(xn_frag && wp_router && rlog && wt && active && list && n_active && ticket, "dfl_moe_router: null pointer");
  DFL_REQUIRE(!h || (norm_w && ldh >= K && ldh % 8 == 0), "dfl_moe_router: rows to normalise need norm_w and ldh >= K, ldh %% 8 == 0");
  DFL_REQUIRE(K >= 32 && K % 32 == 0 && K <= 4096, "dfl_moe_router: K=%d outside 32..4096 (multiples of 32)", K);
  DFL_REQUIRE(E >= 2 && E <= 256 && E % 2 == 0 && top_k >= 1 && top_k <= 8 && top_k <= E, "dfl_moe_router: E=%d top_k=%d outside range", E,
              top_k);
  const int Ep = (E + 15) / 16 * 16;
  DFL_REQUIRE(ld >= Ep && ld % 8 == 0, "dfl_moe_router: ld=%d shorter than the padded expert count %d or not a multiple of 8", ld, Ep);
  MoeRouterArgs a{};
  a.h = (const bf16_t *)h;
  a.ldh = ldh;
  a.nw = (const bf16_t *)norm_w;
  a.eps = eps;
  a.xn = (bf16x8 *)xn_frag;
  a.wr = (const bf16x8 *)wp_router;
  a.KS = K / 32;
  a.E = E;
  a.top_k = top_k;
  a.norm_topk = norm_topk ? 1 : 0;
  a.rlog = (bf16_t *)rlog;
  a.ld = ld;
  a.wt = (bf16_t *)wt;
  a.active = active;
  a.list = list;
  a.n_active = n_active;
  a.dyn = dyn;
  a.dyn_word = dyn_word;
  a.ticket = ticket;
  const dim3 grid(Ep / 16), block(1024);
  if (K <= 2048) {
    if (h)
      hipLaunchKernelGGL((k_moe_router<4, true>), grid, block, 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((k_moe_router<4, false>), grid, block, 0, (hipStream_t)stream, a);
  } else {
    if (h)
      hipLaunchKernelGGL((k_moe_router<8, true>), grid, block, 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((k_moe_router<8, false>), grid, block, 0, (hipStream_t)stream, a);
  }
  DFL_CHECK_LAUNCH("dfl_moe_router");
  return DFL_OK;
}

#ifdef DFL_MOE_STAMPS
extern "C" int dfl_debug_read_router_stamps(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rstamps), sizeof(unsigned long long) * 16 * 12);
}
#endif

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
  // Two column tiles per workgroup share an item's activation fragments (the caller asks for four shares, so that the
  // grid stays at 256 workgroups for N = 2048); DFL_MOE_DOWN_CT=1: one tile per workgroup, the round-2/3 form
  // (same box, BASELINE configs[4]'s cycle: one tile 6.44 - 6.45 ms, two 6.34, four tiles on 8 waves 6.37 - 6.38:
  // profiles/r4_moe_router_ab.txt)
  static const int ct = [] { const char *e = getenv("DFL_MOE_DOWN_CT"); return e ? atoi(e) : 2; }();
  if (ct == 2 && N % 32 == 0)
    hipLaunchKernelGGL((k_moe_down<2, 16>), dim3(N / 32, nsplit), dim3(1024), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((k_moe_down<1, 16>), dim3(N / 16, nsplit), dim3(1024), 0, (hipStream_t)stream, a);
  DFL_CHECK_LAUNCH("dfl_moe_down");
  return DFL_OK;
}
