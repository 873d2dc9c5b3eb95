// Ragged-batch GEMM, second form (round 4): waves OWN column tiles and walk K themselves; the activations of the
// R <= 4 request tiles stream through an LDS ring that every wave of the workgroup reads.
//
// Why: the first form (gemm_batch.hip: k_gemm_b) keeps the activations in registers, which at
// four request tiles allows a 2048-wide K part per workgroup — so K is cut over grid.y and the fused epilogues (SiLU*up,
// argmax) get their finished sums through fp32 slabs in HBM, an arrival ticket and a combine phase in the last workgroup
// to arrive (measured, round 3: lm_head 1.13 x its bytes and 0.65 of the HBM peak, gate/up 1.155 x and 0.60, and every
// launch starts with 256 KB of activation loads per workgroup in front of its first MFMA).  Here:
//   * a wave owns TPU column tiles (one tile, or a gate/up pair) and 1/KQ of every K chunk; its accumulators are the
//     FINISHED sums of (tile, request) over its k-steps, so there is no K cut over workgroups, no slab, no ticket, no
//     combine — at most one KQ-way sum through LDS at the very end (gate/up: KQ = 4);
//   * weights go HBM -> VGPR -> MFMA as before (packed fragments, nt buffer loads clipped at the tile's K), A chunks of
//     the wave's own k-steps in flight;
//   * the activations of a chunk (CK k-steps x MT requests x 1 KiB fragments, already in MFMA B-operand order: the
//     frag16 layout) enter LDS ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4, lane-linear image, conflict-free
//     ds_read_b128), A chunks ahead of their use in a ring of A + 1 slots; every wave issues its share of the pieces.
//     What must be buffered is set by the weights in flight: F bytes of weights over U tiles cover F / U k-steps of K,
//     whose activations (MT KiB per k-step) have to be in LDS — gate/up at 4 tiles: 96..144 KB of weights in flight per
//     CU over 6 tiles = 16..24 k-steps = 64..96 KB of ring (why the small-N projections stay on the
//     register-resident form);
//   * one s_barrier per chunk; every load of the loop is unconditional with a constant count per iteration, so that
//     the in-order vmcnt of a wave can be waited on by immediates: at the top of iteration c the wave's DMA pieces of
//     chunk c are older than exactly A * KPC * TPU weight loads + (A - 1) * SP pieces.
// The compiler must not see the LDS reads of the ring: behind an LDS read that "may alias" LDS-DMA requests in flight
// hipcc waits vmcnt for the youngest of them (SIInsertWaitcnts), i.e. for the chunk just requested.  They are inline
// asm with their own lgkmcnt wait.
#pragma once
#include "gemm_rows.h"

// (External linkage for the argument struct and the kernel template: hipFuncSetAttribute — more than 64 KB of dynamic
// LDS — needs the kernel's address, and the host-side handle of a kernel with internal linkage is not emitted.)
struct GemmRArgs {
  const bf16x8 *wp;  // packed weights [ntiles][KS][64]
  const bf16x8 *xf;  // frag16 activations of request 0 [KS][64]; request r at + r * frag_stride8
  int64_t frag_stride8;
  int KS, ntiles;
  int nunits;  // tiles (TPU = 1) or gate/up pairs (TPU = 2)
  int upp;     // units per workgroup and pass (<= NW / KQ)
  int npass;   // passes over K per workgroup (lm_head: its 37-38 tiles in 3 passes of 13)
  const int32_t *dyn;  // [MT][DFL_DYN_WORDS]
  // EPI_SILU
  bf16_t *act;  // frag16 [I/8][16][8] per request
  int64_t act_stride;
  // EPI_ARGMAX
  int row0, nrows, nrows_word;
  float *best_val;  // [gridDim.x][MT][16]
  int *best_idx;
  bf16_t *logits;  // optional [16][N] per request
  int64_t logits_stride;
  int N;
  // EPI_F32: out[(mt * 16 + m) * ldo + n]
  float *out;
  int ldo;
  int rot_mul;  // workgroup b starts its K walk at chunk (b * rot_mul) % chunks (0: every workgroup at chunk 0)
  int ksp;      // EPI_F32 with K cut over grid.y (0: no cut): k-steps per K part; part y writes out + y * MT * 16 * ldo
  // EPI_RESID: h[r][m][n] <- bf16(sum) or bf16(h + bf16(sum)) (model/dflash.py:140,144), optional second copy (tap)
  bf16_t *h_io;
  int64_t ldh, h_stride;
  int add_resid;
  bf16_t *tap;
  int64_t ldtap, tap_stride;
  float *ss_out;  // optional [ntiles][16] per request: sum over the tile's 16 columns of the new rows' squares (next RMSNorm)
  int64_t ss_stride;
};

namespace {
typedef __attribute__((address_space(3))) void r_lds_void;
typedef __attribute__((address_space(1))) void r_glb_void;

constexpr int R_CK = 8;  // k-steps per ring slot
}  // namespace

// MT request tiles, TPU tiles per unit, KQ waves per unit (each takes CK / KQ k-steps of every chunk), NW waves per
// workgroup, A chunks of look-ahead (weights in flight per wave: A * CK / KQ * TPU KiB; ring: A + 1 slots)
template <int MT, int TPU, int KQ, int NW, int A, int EPI, int CK = 8>
__global__ __launch_bounds__(NW * 64) void dfl_k_gemm_r(GemmRArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass needs the kernel's handle only; with this body it drops the instantiation (no diagnostic)
  constexpr int KPC = CK / KQ, NS = A + 1;  // CK: k-steps per ring slot
  constexpr int NPIECE = CK * MT;                 // 1 KiB pieces per chunk
  constexpr int SP = (NPIECE + NW - 1) / NW;      // pieces per wave and chunk (surplus: a duplicate of the last piece)
  constexpr int VMC = A * KPC * TPU + (A - 1) * SP;  // VMEM ops of a wave younger than its pieces of the chunk it is about to read
  static_assert(CK % KQ == 0 && VMC < 64, "chunk split / vmcnt immediate");
  static_assert(NS * NPIECE * 1024 <= 160 * 1024, "ring exceeds LDS");
  static_assert(NW * TPU * MT * 1024 <= 160 * 1024 || KQ == 1, "the final KQ-way sum reuses the ring's LDS (ring_lds_bytes)");
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];  // [NS][NPIECE][1024]

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int fm = l & 15, fg = l >> 4;  // D layout: row m, columns 4 fg .. 4 fg + 3
  const int u = w / KQ, q = w - u * KQ;
  // K range of this workgroup: the whole K, or (EPI_F32, grid.y parts: fp32 partial sums for a consumer that adds them —
  // dfl_norm_frag_batch) k-steps [ks0, ks0 + KSl)
  const int ks0 = a.ksp ? (int)blockIdx.y * a.ksp : 0;
  const int KSl = a.ksp ? (a.KS - ks0 < a.ksp ? a.KS - ks0 : a.ksp) : a.KS;
  const int nch = (KSl + CK - 1) / CK;
  const int nit = (nch + A - 1) / A * A;  // chunk iterations, rounded up to the unroll (clipped weights: zeros)
  const unsigned ring_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ring;

  int arg_rows[MT];
  float best[MT];
  int bestn[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    best[mt] = -INFINITY;
    bestn[mt] = 0x7fffffff;
    arg_rows[mt] = 0;
    if (EPI == EPI_ARGMAX) {
      arg_rows[mt] = a.nrows;
      if (a.dyn && a.nrows_word >= 0) arg_rows[mt] = a.dyn[mt * DFL_DYN_WORDS + a.nrows_word] - a.row0;
    }
  }

  // this wave's pieces of a chunk: piece i = (k-step f = i / MT, request mt = i % MT) -> LDS slot offset i * 1024.
  // NW % MT == 0: all pieces of a wave belong to ONE request, i.e. one buffer descriptor (clipped at K: a k-step past
  // it reads as zeros and costs no traffic).  The buffer form of the LDS-DMA, not global_load_lds: behind a FLAT-encoded
  // instruction that may touch LDS hipcc flushes vmcnt to ZERO at the next register dependency ("pending flat"),
  // i.e. it would wait for the chunks just requested before the first MFMA of every iteration.
  static_assert(NW % MT == 0, "a wave's pieces share a request");
  const int pmt = w % MT;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16x8 *>(a.xf + pmt * a.frag_stride8 + (size_t)ks0 * 64), 0, KSl * 1024, 0x00020000);
  int pks[SP], poff[SP];
#pragma unroll
  for (int s = 0; s < SP; ++s) {
    int i = w + s * NW;
    i = i < NPIECE ? i : i - NW;  // surplus: the wave's previous piece again (same bytes to the same place)
    pks[s] = i / MT;
    poff[s] = i * 1024;
  }
  // K is walked in chunks of CK k-steps; workgroup b starts at chunk (b * rot_mul) % nch and wraps: at one moment the
  // 256 workgroups x <= 16 waves do NOT all ask HBM for the same offset inside their (128 KB-aligned) column tiles.
  // A pass runs nit >= nch iterations (the unroll's multiple): iterations >= nch are padding (k-steps past K: zeros).
  const int rot = a.rot_mul ? (int)(((unsigned)blockIdx.x * (unsigned)a.rot_mul) % (unsigned)nch) : 0;
  auto kchunk = [&](int c) {
    int k = c + rot;
    k = k >= nch ? k - nch : k;
    return c < nch ? k : nch;
  };
  int slot_w = 0, slot_r = 0;  // ring slots of the next stage() / the next chunk to compute (cycle 0 .. NS - 1)
  auto stage = [&](int kc) {   // data chunk kc -> the next ring slot
    const unsigned slot = ring_base + (unsigned)slot_w * (NPIECE * 1024);
    slot_w = slot_w + 1 == NS ? 0 : slot_w + 1;
#pragma unroll
    for (int s = 0; s < SP; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (r_lds_void *)(uintptr_t)(slot + poff[s]), 16, l * 16, (kc * CK + pks[s]) * 1024, 0, 0);
  };
  // the wave's unit of pass p: weights base of its TPU tiles and their K extent in bytes (0: no unit — zeros, no traffic)
  constexpr bool CONT = KQ == 1;  // passes run as ONE pipeline: the last A iterations of a pass request the next pass's first chunks
  auto unit_of = [&](int p, const bf16x8 *(&base)[TPU], int &bytes, int &g) {
    g = (int)blockIdx.x + (p * a.upp + u) * (int)gridDim.x;
    const bool have = p < a.npass && u < a.upp && g < a.nunits;
#pragma unroll
    for (int tp = 0; tp < TPU; ++tp) base[tp] = a.wp + ((size_t)(have ? TPU * g + tp : 0) * a.KS + ks0) * 64;
    bytes = have ? KSl * 1024 : 0;
  };
  bf16x8 wreg[A][KPC][TPU];
  // weights of data chunk kc for this wave: k-steps kc * CK + q * KPC + j of the unit's tiles
  auto wload1 = [&](bf16x8(&dst)[TPU], const bf16x8 *const(&base)[TPU], int bytes, int kc, int j) {
    const int soff = (kc * CK + q * KPC + j) * 1024;
#pragma unroll
    for (int tp = 0; tp < TPU; ++tp) {
      const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(base[tp]), 0, bytes, 0x00020000);
      dst[tp] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, l * 16, soff, 2));
    }
  };
  auto wload = [&](bf16x8(&dst)[KPC][TPU], const bf16x8 *const(&base)[TPU], int bytes, int kc) {
#pragma unroll
    for (int j = 0; j < KPC; ++j) wload1(dst[j], base, bytes, kc, j);
  };

  const bf16x8 *wcur[TPU], *wnxt[TPU];
  int bcur, bnxt, gcur, gnxt;
  unit_of(0, wcur, bcur, gcur);
  // ---- prologue: the first A chunks, pieces then weights per chunk (the order the immediates below assume)
#pragma unroll
  for (int c = 0; c < A; ++c) {
    stage(kchunk(c));
    __builtin_amdgcn_sched_barrier(0);
    wload(wreg[c], wcur, bcur, kchunk(c));
    __builtin_amdgcn_sched_barrier(0);
  }

  for (int p = 0; p < a.npass; ++p) {
    unit_of(CONT ? p + 1 : a.npass, wnxt, bnxt, gnxt);  // (not CONT: nothing is requested across a pass boundary)
    f32x4 acc[TPU][MT];
#pragma unroll
    for (int tp = 0; tp < TPU; ++tp)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[tp][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < nit; c0 += A) {
#pragma unroll
      for (int ca = 0; ca < A; ++ca) {
        const int c = c0 + ca;
        // this wave's pieces of the chunk it is about to read have landed ...
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");
        // ... and everyone's; every wave has finished reading the chunk before, whose slot the next request takes
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // A chunks ahead: this pass's chunk c + A, or (one pipeline over the passes) the next pass's chunk c + A - nit
        const int cn = c + A;
        const bool nx = cn >= nit;
        const int kcn = nx ? (CONT && p + 1 < a.npass ? kchunk(cn - nit) : nch) : kchunk(cn);
        stage(kcn);
        __builtin_amdgcn_sched_barrier(0);
        // (uniform selects, not a branch: loads under a branch cost the counted waits)
        const bf16x8 *wsel[TPU];
#pragma unroll
        for (int tp = 0; tp < TPU; ++tp) wsel[tp] = nx ? wnxt[tp] : wcur[tp];
        const int bsel = nx ? bnxt : bcur;
        const unsigned rd = ring_base + (unsigned)slot_r * (NPIECE * 1024) + (unsigned)(q * KPC * MT) * 1024 + l * 16;
        slot_r = slot_r + 1 == NS ? 0 : slot_r + 1;
#pragma unroll
        for (int j = 0; j < KPC; ++j) {
          bf16x8 b[MT];
          if constexpr (MT == 4) {
            asm volatile(
                "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                : "v"(rd + j * MT * 1024)
                : "memory");
          } else if constexpr (MT == 2) {
            asm volatile(
                "ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(b[0]), "=&v"(b[1])
                : "v"(rd + j * MT * 1024)
                : "memory");
          } else {
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b[0]) : "v"(rd + j * MT * 1024) : "memory");
          }
#pragma unroll
          for (int tp = 0; tp < TPU; ++tp)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              acc[tp][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ca][j][tp], b[mt], acc[tp][mt], 0, 0, 0);
          // the registers just consumed take the same k-step of the chunk A ahead at once (past K: clipped, zeros, no
          // traffic): the wave keeps A chunks of weights in flight all the time, not A - 1 while it computes
          __builtin_amdgcn_sched_barrier(0);
          wload1(wreg[ca][j], wsel, bsel, kcn, j);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    // ---- finished sums of (unit, request): KQ-way sum through LDS, then the fused epilogue in the MFMA D layout
    auto epilogue = [&](int t_last, int mt, const f32x4 &s, const f32x4 &g4) {
      // s: sums of tile t_last (row fm, columns 4 fg .. + 3); SILU: g4 = the gate tile's sums, s = the up tile's
      const int n0 = (EPI == EPI_SILU ? (t_last >> 1) : t_last) * 16 + 4 * fg;
      if (EPI == EPI_SILU) {  // tf:modeling_qwen3.py:82, rounded where torch rounds
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gb = rbf(g4[r]), ub = rbf(s[r]);
          const float act = rbf(gb / (1.f + __expf(-gb)));
          o[r] = f2bf(act * ub);
        }
        *reinterpret_cast<bf16x4 *>(a.act + mt * a.act_stride + ((size_t)(n0 >> 3) * 16 + fm) * 8 + (n0 & 7)) = o;
      } else if (EPI == EPI_ARGMAX) {
        const bool live = (fm >= a.row0) && (fm < a.row0 + arg_rows[mt]);
        if (a.logits && live) {
          bf16x4 o = {f2bf(s[0]), f2bf(s[1]), f2bf(s[2]), f2bf(s[3])};
          *reinterpret_cast<bf16x4 *>(a.logits + mt * a.logits_stride + (size_t)fm * a.N + n0) = o;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vb = rbf(s[r]);  // lm_head output is bf16 before argmax (model/dflash.py:238,247); first maximum kept
          if (live && (vb > best[mt] || (vb == best[mt] && n0 + r < bestn[mt]) || bestn[mt] == 0x7fffffff)) {
            best[mt] = vb;
            bestn[mt] = n0 + r;
          }
        }
      } else if (EPI == EPI_RESID) {
        bf16_t *hp = a.h_io + mt * a.h_stride + (int64_t)fm * a.ldh + n0;
        bf16x4 o;
        if (a.add_resid) {
          const bf16x4 hv = *reinterpret_cast<const bf16x4 *>(hp);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = f2bf(bf2f(hv[r]) + rbf(s[r]));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = f2bf(s[r]);
        }
        *reinterpret_cast<bf16x4 *>(hp) = o;
        if (a.tap) *reinterpret_cast<bf16x4 *>(a.tap + mt * a.tap_stride + (int64_t)fm * a.ldtap + n0) = o;
        if (a.ss_out) {  // (uniform)
          float q = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) q += bf2f(o[r]) * bf2f(o[r]);
          q += __shfl_xor(q, 16, 64);
          q += __shfl_xor(q, 32, 64);
          if (fg == 0) a.ss_out[mt * a.ss_stride + t_last * 16 + fm] = q;
        }
      } else {  // EPI_F32
        *reinterpret_cast<f32x4 *>(a.out + ((size_t)(blockIdx.y * MT + mt) * 16 + fm) * a.ldo + n0) = s;
      }
    };

    if constexpr (KQ == 1) {
      if (bcur) {  // (the wave has a unit in this pass)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) epilogue(TPU * gcur + TPU - 1, mt, acc[TPU - 1][mt], acc[0][mt]);
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (padding pieces past the last chunk still target the ring)
      __builtin_amdgcn_s_barrier();                     // every wave has left the ring
      asm volatile("" ::: "memory");
      float *red = reinterpret_cast<float *>(ring);     // [NW][TPU][MT][256]
#pragma unroll
      for (int tp = 0; tp < TPU; ++tp)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4 *>(&red[((w * TPU + tp) * MT + mt) * 256 + l * 4]) = acc[tp][mt];
      __syncthreads();
      for (int it = w; it < a.upp * MT; it += NW) {  // item = (unit slot, request)
        const int uu = it / MT, mt = it - uu * MT;
        const int gg = (int)blockIdx.x + (p * a.upp + uu) * (int)gridDim.x;
        if (gg >= a.nunits) continue;
        f32x4 s[TPU];
#pragma unroll
        for (int tp = 0; tp < TPU; ++tp) {
          s[tp] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int qq = 0; qq < KQ; ++qq)  // fixed order: deterministic sums
            s[tp] += *reinterpret_cast<const f32x4 *>(&red[(((uu * KQ + qq) * TPU + tp) * MT + mt) * 256 + l * 4]);
        }
        epilogue(TPU * gg + TPU - 1, mt, s[TPU - 1], s[0]);
      }
      __syncthreads();
    }
    // ---- the next pass
#pragma unroll
    for (int tp = 0; tp < TPU; ++tp) wcur[tp] = wnxt[tp];
    bcur = bnxt;
    gcur = gnxt;
    if (!CONT && p + 1 < a.npass) {  // the ring was reused for the sums: a prologue of its own
      unit_of(p + 1, wcur, bcur, gcur);
      slot_r = slot_w;
#pragma unroll
      for (int c = 0; c < A; ++c) {
        stage(kchunk(c));
        __builtin_amdgcn_sched_barrier(0);
        wload(wreg[c], wcur, bcur, kchunk(c));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  if (EPI == EPI_ARGMAX) {
    // the waves' candidates per (request, row) meet in LDS: ONE entry per workgroup goes out (the finish kernel scans
    // gridDim.x entries per row, not gridDim.x * NW: 30 us -> 5 us at 256 x 16)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (padding pieces past the last chunk still target the ring)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float *cv = reinterpret_cast<float *>(ring);             // [NW][MT][16]
    int *ci = reinterpret_cast<int *>(ring) + NW * MT * 16;  // [NW][MT][16]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float bv = best[mt];
      int bn = bestn[mt];
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {  // lanes of a row (same fm) differ in fg
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bn, o, 64);
        if (ov > bv || (ov == bv && oi < bn)) {
          bv = ov;
          bn = oi;
        }
      }
      if (fg == 0) {
        cv[(w * MT + mt) * 16 + fm] = bv;
        ci[(w * MT + mt) * 16 + fm] = bn;
      }
    }
    __syncthreads();
    if (tid < MT * 16) {
      float bv = -INFINITY;
      int bn = 0x7fffffff;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) {
        const float ov = cv[ww * MT * 16 + tid];
        const int oi = ci[ww * MT * 16 + tid];
        if (oi != 0x7fffffff && (bn == 0x7fffffff || ov > bv || (ov == bv && oi < bn))) {
          bv = ov;
          bn = oi;
        }
      }
      a.best_val[(size_t)blockIdx.x * MT * 16 + tid] = bv;  // [gridDim.x][MT][16]
      a.best_idx[(size_t)blockIdx.x * MT * 16 + tid] = bn;
    }
  }
#endif
}

namespace {

template <int MT, int TPU, int KQ, int NW, int A, int CK = 8>
constexpr int ring_lds_bytes() {
  const int ring = (A + 1) * CK * MT * 1024, red = KQ > 1 ? NW * TPU * MT * 1024 : 0;
  return ring > red ? ring : red;
}

// The ring form takes fragment sources (mode 0) whose K fits whole k-steps; the caller falls back to k_gemm_b otherwise.
inline bool ring_ok(const dfl_rows_batch *x, int K) { return x && x->r0.mode == 0 && x->r0.frag && K % 32 == 0 && K >= 256; }

}  // namespace
