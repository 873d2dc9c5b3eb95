/*
 * dflash_hip.h — C ABI of libdflash_hip.so: the MI355X (gfx950) kernels behind the
 * DFlash block-diffusion speculative-decoding hot path.
 *
 * Drop-in boundary: the reference has no FFI — its hot path is
 * Python calling torch/transformers ops.  Each entry point below replaces the
 * torch op sequence at the cited reference lines (paths relative to the reference
 * repo; `tf:` = transformers 5.15.0).  The Python mirror of the reference's
 * operator API (dflash_amd.DFlashDraftModel.forward / spec_generate, dflash_generate,
 * dflash_generate_policy) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into caller-owned memory (torch storage)
 *    unless named host_*; the library allocates nothing and keeps no state except
 *    the thread-local last-error string;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only
 *    enqueue work, never synchronise, and are capturable into a hipGraph;
 *  - return 0 on success, a negative DFL_E* code on a rejected argument or a
 *    failed launch; dfl_last_error() gives the text;
 *  - bf16 storage everywhere, fp32 accumulation; ids are int64 (torch.long).
 *
 * Device layouts (MI355X-first)
 *  - "packed weight": a row-major [N][K] bf16 nn.Linear weight re-laid as
 *    [N/16][K/32][64 lanes][8] so that one wave-wide 16-B load is one MFMA
 *    16x16x32 A-operand fragment and a column tile streams as one contiguous run;
 *  - "frag16 activations": up to 16 rows x K bf16 stored as [K/8][16][8], i.e. the
 *    matching MFMA B-operand fragments, written directly by the producing kernel;
 *  - "dyn": int32[8] per request on the device: {S, tau, bs, pos0, ...} — draft
 *    cache length before this cycle, context rows, block rows, absolute position
 *    of the first context row.  Kernels read lengths from it so that one captured
 *    graph serves every cycle.
 */
#ifndef DFLASH_HIP_H
#define DFLASH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFL_ABI_VERSION 1

#define DFL_OK 0
#define DFL_EINVAL (-22)   /* bad shape / alignment / null pointer */
#define DFL_ELAUNCH (-5)   /* hipLaunch failed (text in dfl_last_error) */

#define DFL_DYN_WORDS 8
#define DFL_DYN_S 0        /* rows already in the draft KV cache               */
#define DFL_DYN_TAU 1      /* context rows appended this cycle (prev acc+1)     */
#define DFL_DYN_BS 2       /* block (noise) rows this cycle                     */
#define DFL_DYN_POS0 3     /* absolute position id of the first context row     */
#define DFL_DYN_START 4    /* `start` of the decode loop (= pos0 + tau)         */
#define DFL_DYN_STOP 5     /* 1 once a stop token was committed                 */
#define DFL_DYN_CYCLE 6    /* cycles run                                        */

/* Where one 16-row activation tile of a GEMM comes from.
 *  mode 0  frag   : frag16 fragments [K/8][16][8] written by a producer kernel;
 *  mode 1  rows   : plain bf16 rows [16][K] with row stride ld (e.g. the target taps); all 16
 *                   rows must be READABLE memory (they are loaded before the valid count is known
 *                   and masked afterwards), whatever dyn[valid_word] says;
 *  mode 2  normed : rows = the residual stream h (a buffer of 16 readable rows), and the GEMM applies the RMSNorm while
 *                   it builds its fragments: x = norm_w * bf16(h * rsqrt(sum_i ss[i][m] / K
 *                   + eps)) (Qwen3RMSNorm, tf:models/qwen3/modeling_qwen3.py:59-64), where
 *                   ss[nss][16] are partial sums of squares of each row left by the
 *                   producer (dfl_embed_rows: nss = 1; dfl_gemm_resid: nss = N/16).
 * Rows >= dyn[valid_word] (valid_word >= 0) are treated as zero. */
typedef struct dfl_rows {
  const void *frag;
  const void *rows;
  int64_t ld;
  const float *ss;
  int32_t nss;
  const void *norm_w;
  float eps;
  int32_t valid_word;
  int32_t mode;
} dfl_rows;

int dfl_version(void);
const char *dfl_last_error(void);

/* ---- one-time layout conversion (weight load; not on the per-cycle path) ---- */

/* nn.Linear weight [N][K] bf16 row-major -> packed weight.  N%16==0, K%32==0. */
int dfl_pack_weight(const void *w, void *wp, int N, int K, void *stream);
/* gate_proj and up_proj [I][K] each -> one packed weight of 2I rows with tiles
 * interleaved (gate tile t, up tile t) so SiLU(gate)*up fuses into the GEMM
 * epilogue (tf:models/qwen3/modeling_qwen3.py:81-83). */
int dfl_pack_weight_gateup(const void *gate, const void *up, void *wp, int I, int K, void *stream);

/* ---- per-cycle kernels ---- */

/* dyn <- {S, tau, bs, pos0, start=pos0+tau, stop=0, cycle=0}. */
int dfl_set_dyn(int32_t *dyn, int S, int tau, int bs, int pos0, void *stream);
/* Blocks of 17..32 rows (results.md:11-16 sweeps 20 and 24; benchmark_dynamic_schedule.py:44-51 takes any
 * candidate >= 2) run as TWO 16-row tiles: every GEMM / embed launch is issued once per tile on rows 16 t ..
 * with the tile's own record.  dyn holds 2 x DFL_DYN_WORDS ints; record t gets
 * {S, clamp(tau - 16 t, 0, 16), clamp(bs - 16 t, 0, 16), pos0, start, 0, 0}. */
int dfl_set_dyn2(int32_t *dyn, int S, int tau, int bs, int pos0, void *stream);

/* rows x K bf16 (row stride ldx elements) -> frag16; rows beyond n_valid zeroed.
 * n_valid = dyn[dyn_word] if dyn != NULL else rows.  Used for the target taps
 * (model/utils.py:16-25 output) feeding fc. */
int dfl_pack_rows(const void *x, int64_t ldx, int rows, int K, void *xf, const int32_t *dyn, int dyn_word,
                  void *stream);

/* Skinny GEMM, weights streamed once: out[c][mt*16+m][n] = sum_{k in chunk c} x_mt[m][k] * W[n][k]
 * (fp32 partials, summed and rounded to bf16 by the consumer exactly where the
 * reference's nn.Linear output is rounded).  Replaces F.linear at model/dflash.py:70,
 * 73-76 (q/k/v of context and block rows).  mt in {1,2} row tiles (x1 ignored for
 * mt==1), N%16==0, K%32==0, out has ksplit*mt*16*N floats;
 * ksplit >= ceil(K/32 / (16*8/mt)). */
int dfl_gemm_f32(const void *wp, const dfl_rows *x0, const dfl_rows *x1, int mt, int N, int K, int ksplit, float *out,
                 const int32_t *dyn, void *stream);

/* act = bf16(silu(bf16(x Wg^T))) * bf16(x Wu^T) written as frag16 [I/8][16][8]
 * (tf:...modeling_qwen3.py:82 inner expression).  wp from dfl_pack_weight_gateup.
 * K/32 <= 128 (no K split: the activation needs the finished sums). */
int dfl_gemm_silu_mul(const void *wp_gateup, const dfl_rows *x, int I, int K, void *act_frag, const int32_t *dyn,
                      void *stream);

/* lm_head GEMM fused with the greedy unmask: ids[r] = argmax_n bf16(x[r] . W[n])
 * for r in [row0, row0+nrows), first index on ties (model/dflash.py:238-247 +
 * model/utils.py:28-29).  The 15 x V logits are never materialised unless
 * `logits` != NULL (bf16 [16][V], rows outside the range untouched).
 * ws: workspace of dfl_argmax_ws_bytes() bytes.  out_ids int64, written at
 * out_ids[r - row0 + out_off]. nrows_dyn_word >= 0: rows = dyn[word] - row0.
 * margin_out (optional, fp32, same indexing as out_ids): top-1 minus top-2 bf16 logit of the
 * row, the reference's confidence statistic (benchmark_candidate_solutions.py:296-302,
 * torch.topk(2) values: an exact tie gives 0) — the per-position confidence comes with the
 * unmask at no extra pass over the logits. */
int64_t dfl_argmax_ws_bytes(void);
int dfl_gemm_argmax(const void *wp, const dfl_rows *x, int V, int K, int row0, int nrows, const int32_t *dyn,
                    int nrows_dyn_word, void *ws, int64_t *out_ids, int out_off, void *logits, float *margin_out,
                    void *stream);
/* The same, with two hipEvent_t recorded on `stream` right before and right after the GEMM launch (the argmax finish
 * launch comes behind ev_end): bench.py times the lm_head kernel itself with them (roofline.achieved). */
int dfl_gemm_argmax_timed(const void *wp, const dfl_rows *x, int V, int K, int row0, int nrows, const int32_t *dyn,
                          int nrows_dyn_word, void *ws, int64_t *out_ids, int out_off, void *logits, float *margin_out,
                          void *ev_start, void *ev_end, void *stream);

/* GEMM with the residual epilogue (o_proj / down_proj / fc, model/dflash.py:101,140,144,177):
 *   v = bf16(x W^T);  h_io[m][n] <- add_residual ? bf16(h_io[m][n] + v) : v;
 *   tap[m][n] <- the same value (optional: a tapped target layer, model/utils.py:16-25);
 *   ss_out[n/16][m] <- sum over the tile's 16 columns of h_new^2 (optional; feeds the
 *   mode-2 row source of the next GEMM, so the RMSNorm is no launch of its own).
 * Any K: beyond 4096 the workgroup walks K in chunks itself (x must then be mode 0/1). */
int dfl_gemm_resid(const void *wp, const dfl_rows *x, int N, int K, void *h_io, int64_t ldh, int add_residual,
                   void *tap, int64_t ldtap, float *ss_out, const int32_t *dyn, void *stream);

/* h_out[m] = embed[ids[m]] for m < dyn[dyn_word] (model/dflash.py:237) and ss_out[m] =
 * sum of squares of that row (one partial per row: nss = 1 for the next GEMM). */
int dfl_embed_rows(const void *embed, const int64_t *ids, void *h_out, int H, float *ss_out, const int32_t *dyn,
                   int dyn_word, void *stream);

/* Row-wise residual/norm stage that also converts to frag16:
 *   v   = part ? bf16(sum_c part[c][row_off+m][:]) : (none)
 *   h   = embed ? embed[ids[m]] : resid_in ? resid_in[m] : 0       (bf16)
 *   h   = part ? (resid_in||embed ? bf16(h + v) : v) : h            (model/dflash.py:140,144)
 *   h_out[m] = h                                   (if h_out; h_out2[m*ld2 ..] gets the same row:
 *                                                   a tapped target layer, model/utils.py:16-25)
 *   frag[m]  = norm_w * bf16(h * rsqrt(mean(h^2)+eps))  (Qwen3RMSNorm, tf:...:59-64)
 * rows m >= n_valid (dyn[dyn_word], or 16) get zero fragments.  H%8==0, H<=16384.
 * part row stride ldp floats, split stride part_split floats. */
int dfl_norm_pack(const float *part, int nsplit, int64_t part_split, int ldp, int row_off, const void *resid_in,
                  const void *embed, const int64_t *ids, void *h_out, void *h_out2, int64_t ld2, const void *norm_w,
                  float eps, void *frag, int H, const int32_t *dyn, int dyn_word, void *stream);

/* q_norm/k_norm + RoPE + KV append (model/dflash.py:71-85 with the local
 * apply_rotary_pos_emb of :22-28).  qkv: fp32 partials from dfl_gemm_f32,
 * element (split c, buffer row r, column j) at qkv[c*split_stride + r*ld + j];
 * q/k/v column blocks start at q_col/k_col/v_col (q_col < 0: no q wanted);
 * context rows sit at buffer rows ctx_row0.., block rows at blk_row0.. (< 0: none).
 * Writes
 *   q_out  bf16 [n_q][16][128]       (block rows only: q uses the last bs positions)
 *   kcache/vcache bf16 [n_kv][cache_rows][128] at rows S + rel, rel = row_base + i
 *   for context row i, tau + j for block row j; RoPE position = pos0 + rel.
 * cos/sin: bf16 [max_pos][64] tables (tf:...:125-137, computed by the host in fp32
 * then cast, as the reference does).  head_dim must be 128.  q_norm_w == k_norm_w ==
 * NULL skips the per-head norms (Llama-style attention).
 * ctx_rows_override >= 0: that many context rows instead of dyn tau and no block
 * rows — the draft-cache prefill of the prompt's context in 16-row groups, with
 * row_base = index of the group's first row. */
int dfl_qknorm_rope_append(const float *qkv, int nsplit, int64_t split_stride, int ld, int q_col, int k_col,
                           int v_col, int ctx_row0, int blk_row0, int n_q, int n_kv, const void *q_norm_w,
                           const void *k_norm_w, float eps, const void *cos_tab, const void *sin_tab, int max_pos,
                           void *q_out, void *kcache, void *vcache, int cache_rows, const int32_t *dyn,
                           int ctx_rows_override, int row_base, void *stream);

/* GQA attention of the block's 16 query rows over the cached prefix + this cycle's
 * rows: softmax(q k^T * scale) v over kv_len = S + tau + bs keys.  causal = 0: no mask
 * (the draft, is_causal=False, model/dflash.py:86-99).  causal = 1: query row j sees
 * cache rows <= S + tau + j (the target's verify forward over the same block).  MFMA QK^T / PV, K/V tiles
 * staged in LDS, split over the key axis with a log-sum-exp merge.
 * ws: dfl_attn_ws_bytes(n_q, max_splits) bytes.  out: frag16 [n_q*128/8][16][8]. */
int64_t dfl_attn_ws_bytes(int n_q, int max_splits);
int dfl_block_attn(const void *q, const void *kcache, const void *vcache, int cache_rows, int n_q, int n_kv,
                   float scale, int causal, const int32_t *dyn, int kv_len_max, void *ws, int max_splits,
                   void *out_frag, void *stream);

/* The whole attention stage of a block in ONE launch: q/k-norm + RoPE + KV append
 * (model/dflash.py:71-85) + attention (:86-99, causal = 0; target verify, causal = 1)
 * + the merge of the key splits, result as frag16 for o_proj.  Same arithmetic as
 * dfl_qknorm_rope_append + dfl_block_attn (V rows bit-identical, K up to rare 1-ulp flips); the key axis is
 * split over workgroups and merged by the last one to arrive.  tau + bs <= 32 new rows.
 * ws: dfl_attn_fused_ws_bytes(n_q, n_kv, max_splits) bytes, ZEROED once by the caller
 * (it holds the arrival tickets, which every launch leaves at zero again). */
int64_t dfl_attn_fused_ws_bytes(int n_q, int n_kv, int max_splits);
int dfl_attn_fused(const float *qkv, int nsplit, int64_t split_stride, int ld, int q_col, int k_col, int v_col,
                   int ctx_row0, int blk_row0, int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w,
                   float eps, const void *cos_tab, const void *sin_tab, int max_pos, void *kcache, void *vcache,
                   int cache_rows, float scale, int causal, const int32_t *dyn, int kv_len_max, void *ws,
                   int max_splits, void *out_frag, void *stream);

/* The attention stage, second form: same arithmetic as dfl_attn_fused (model/dflash.py:71-99; causal = 1: the
 * target verify's stage, :249-255), but q/k/v arrive as FINISHED bf16 Linear outputs — dfl_gemm_resid(add_residual
 * = 0) of the block rows into xq [bs][ldq] (q | k | v column blocks at q_col / k_col / v_col) and, for the draft,
 * of the context rows into xc [tau][ldc] (k, v at ck_col / cv_col; xc may be NULL when tau == 0) — and the grid is
 * (kv head, query heads of the group x key splits): one query head per workgroup (two, sharing every K/V tile, from
 * ~5k cached keys), 8 waves over disjoint 32-key
 * tiles with no barrier in the loop, wave results merged in LDS, one 8 KB partial per workgroup published
 * write-through and merged by the head's last arriver (csrc/attn_head.hip).  New rows (tau + bs <= 64, tau <= 32)
 * are appended to the cache at rows S.. by the launch itself.  bs <= 16 * q_tiles, q_tiles in {1, 2}: the second
 * query tile's frag16 output lies out_tile_stride bf16 elements after the first.
 * Lengths: dyn == NULL -> the immediates S, tau, bs, pos0; else they are read from dyn and S is the caller's
 * upper bound on dyn[S] (it sizes the key splits, as kv_len_max does for dfl_attn_fused).
 * ws: dfl_attn_head_ws_bytes(n_q, max_splits, q_tiles) bytes, ZEROED once (arrival tickets, left zero). */
int64_t dfl_attn_head_ws_bytes(int n_q, int max_splits, int q_tiles);
int dfl_attn_head(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, const void *xc, int64_t ldc,
                  int ck_col, int cv_col, int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w, float eps,
                  const void *cos_tab, const void *sin_tab, int max_pos, void *kcache, void *vcache, int cache_rows,
                  float scale, int causal, const int32_t *dyn, int S, int tau, int bs, int pos0, int q_tiles,
                  void *ws, int max_splits, void *out_frag, int64_t out_tile_stride, void *stream);

/* dfl_attn_head (q_tiles = 1) AND the o_proj + residual GEMM that follows it (model/dflash.py:101,140; for the
 * target tf:modeling_qwen3.py o_proj) in ONE launch: besides the attention workgroups the grid carries one workgroup
 * per 16-column tile of o_proj, which pulls its weight slice into registers while the attention runs (the stage
 * leaves HBM idle), waits for the heads' outputs (bounded: 2 ms), and finishes the GEMM with dfl_gemm_resid's
 * epilogue: h_io [16][ldh] <- bf16(h_io + bf16(attn . Wo^T)), ss_out [H/16][16] partial sums of squares.
 * Range: q_dim = n_q * 128 <= 4096, bs <= 16, tau + bs <= 32, H % 16 == 0.  attn_frag: frag16 of the attention output
 * (16 * q_dim bf16; rows >= bs are written as zeros).  sync: DFL_ATTN_OPROJ_SYNC_WORDS int32, ZEROED once; sync[1025] != 0 after a launch means an
 * o_proj workgroup gave up waiting and h_io is invalid (the Python layer raises; it cannot happen on an idle GPU). */
#define DFL_ATTN_OPROJ_SYNC_WORDS 1056
int dfl_attn_head_oproj(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, const void *xc, int64_t ldc,
                        int ck_col, int cv_col, int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w, float eps,
                        const void *cos_tab, const void *sin_tab, int max_pos, void *kcache, void *vcache,
                        int cache_rows, float scale, int causal, const int32_t *dyn, int S, int tau, int bs, int pos0,
                        void *ws, int max_splits, void *attn_frag, const void *wo_packed, int H, void *h_io,
                        int64_t ldh, float *ss_out, int32_t *sync, void *stream);

/* ---- multi-candidate verify (benchmark_candidate_solutions.py) ----
 * Several drafts of ONE block are verified against ONE cached prefix and the best is kept (:570-618).
 *
 * dfl_attn_head_cand: dfl_attn_head (causal, no context rows, q_tiles = 1) for n_cand candidate blocks in one launch
 * (grid.z = candidate): candidate c's block rows at xq + c * xq_cand_stride elements, its frag16 output at out_frag +
 * c * out_cand_stride elements; every candidate attends the same cached rows [0, S) plus ITS OWN bs new rows, which go
 * not to the cache but to the staging area k_out / v_out [c][n_kv][out_rows][128] (rows 0..bs-1) — the reference clones
 * and batch-repeats the whole DynamicCache instead (:76-81, :572-575) and selects the winner's copy (:604-608); here the
 * caller copies the winner's bs rows into the cache.  ws: n_cand * dfl_attn_head_ws_bytes(n_q, max_splits, 1), zeroed. */
int dfl_attn_head_cand(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, int n_cand, int64_t xq_cand_stride,
                       int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w, float eps, const void *cos_tab,
                       const void *sin_tab, int max_pos, const void *kcache, const void *vcache, int cache_rows,
                       float scale, int S, int bs, void *ws, int max_splits, void *out_frag, int64_t out_cand_stride,
                       void *k_out, void *v_out, int64_t kv_out_cand_stride, int out_rows, void *stream);
/* The same for candidate blocks of 17..32 rows (q_tiles = 2): candidate c = two consecutive 16-row tiles of xq and of
 * out_frag (out_tile_stride elements apart); ws: n_cand * dfl_attn_head_ws_bytes(n_q, max_splits, q_tiles) bytes. */
int dfl_attn_head_cand_t(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, int n_cand, int64_t xq_cand_stride,
                         int n_q, int n_kv, const void *q_norm_w, const void *k_norm_w, float eps, const void *cos_tab,
                         const void *sin_tab, int max_pos, const void *kcache, const void *vcache, int cache_rows,
                         float scale, int S, int bs, void *ws, int max_splits, void *out_frag, int64_t out_cand_stride,
                         int64_t out_tile_stride, int q_tiles, void *k_out, void *v_out, int64_t kv_out_cand_stride,
                         int out_rows, void *stream);

/* dfl_attn_head for the R requests of a ragged batch in one launch (grid.z = request; replaces dfl_attn_fused_batch):
 * request r's block rows at xq + r * xq_req_stride, its lengths at dyn + r * DFL_DYN_WORDS (block form: the context rows
 * are cached already, dyn tau == 0), its cache at + r * cache_req_stride, its frag16 output at + r * out_req_stride.
 * kv_len_max bounds S + 16 over the requests (it sizes the key splits).  ws: R * dfl_attn_head_ws_bytes(n_q, max_splits,
 * 1) bytes, zeroed once. */
int dfl_attn_head_batch(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, int R, int64_t xq_req_stride, int n_q,
                        int n_kv, const void *q_norm_w, const void *k_norm_w, float eps, const void *cos_tab,
                        const void *sin_tab, int max_pos, void *kcache, void *vcache, int cache_rows,
                        int64_t cache_req_stride, float scale, int causal, const int32_t *dyn, int kv_len_max, void *ws,
                        int max_splits, void *out_frag, int64_t out_req_stride, void *stream);
/* dfl_attn_head_batch on the fp32 K-PART SUMS of the q/k/v projection (dfl_gemm_f32_batch's output) instead of finished
 * bf16 rows (model/dflash.py:70-76: a q/k/v value is the Linear's bf16 output = bf16(part 0 + part 1)): request r's rows of
 * part k at xq_parts + k * part_stride + r * xq_req_stride floats, row stride ldq floats; nparts = dfl_batch_ksplit(hidden)
 * must be 1 or 2.  The K parts of the projection meet in this launch's row loads, so the GEMM in front of it needs no slab /
 * ticket / combine phase (qkv at 4 request tiles: 18.3 -> 14.7 us).  Blocks of <= 16 rows. */
int dfl_attn_head_batch_f32(const float *xq_parts, int nparts, int64_t part_stride, int64_t ldq, int q_col, int k_col,
                            int v_col, int R, int64_t xq_req_stride, int n_q, int n_kv, const void *q_norm_w,
                            const void *k_norm_w, float eps, const void *cos_tab, const void *sin_tab, int max_pos,
                            void *kcache, void *vcache, int cache_rows, int64_t cache_req_stride, float scale, int causal,
                            const int32_t *dyn, int kv_len_max, void *ws, int max_splits, void *out_frag,
                            int64_t out_req_stride, void *stream);
/* The same for blocks of 17..32 rows (q_tiles = 2; benchmark.py's block-size sweep, results.md:11-16, with several requests
 * per GPU): request r is TWO consecutive 16-row tiles of xq and of out_frag (out_tile_stride elements apart), one cache and
 * one length record whose bs counts both tiles.  ws: R * dfl_attn_head_ws_bytes(n_q, max_splits, q_tiles) bytes. */
int dfl_attn_head_batch_t(const void *xq, int64_t ldq, int q_col, int k_col, int v_col, int R, int64_t xq_req_stride, int n_q,
                          int n_kv, const void *q_norm_w, const void *k_norm_w, float eps, const void *cos_tab,
                          const void *sin_tab, int max_pos, void *kcache, void *vcache, int cache_rows,
                          int64_t cache_req_stride, float scale, int causal, const int32_t *dyn, int kv_len_max, void *ws,
                          int max_splits, void *out_frag, int64_t out_req_stride, int64_t out_tile_stride, int q_tiles,
                          void *stream);

/* Per row of bf16 logits [rows][ld] (rows <= 64): the k <= 8 largest values with their indices, ordered (value
 * descending, index ascending) — out_val fp32 [rows][8], out_idx int32 [rows][8] — and the row's log-sum-exp (fp32).
 * What the candidate builders take from the 15 x V draft logits: torch.topk at :216, :296, the top-2 probability
 * margin softmax(x)[top1] - softmax(x)[top2] = exp(v1 - lse) - exp(v2 - lse) at :98-101, :305-307, and
 * log_softmax(x)[token] = x[token] - lse at :119-131.  torch.topk's order among EQUAL values is an implementation
 * detail; this one is fixed (and equals torch.argmax's for k = 1). */
int dfl_topk_rows(const void *logits, int64_t ld, int rows, int V, int k, float *out_val, int32_t *out_idx, float *out_lse,
                  void *stream);

/* Acceptance length of every candidate block against its posterior, the choice of :592-601 — maximise tau, then the
 * draft score, then prefer the lower candidate index, evaluated as the reference does: argmax over the fp32 value
 * tau * 1e6 + draft_score - idx * 1e-3 — and the winner's commit (:612-613) with the stop test and length bookkeeping
 * of dfl_accept_commit.  blocks / posterior int64 [n_cand][stride], scores fp32 [n_cand], n_cand <= 8.
 * result int32 [12]: {acc, new_start, stop, winner, acc of candidate 0..7 (-1 beyond n_cand)}. */
int dfl_candidate_select(const int64_t *blocks, int64_t blk_stride, const int64_t *posterior, int64_t post_stride,
                         const float *scores, int n_cand, int bs, int64_t *output_ids, int64_t output_len, int32_t *dyn,
                         const int64_t *stop_ids, int n_stop, int32_t *result, void *stream);

/* ---- sparse-MoE MLP of a target's verify forward (BASELINE configs[4]; tf:models/qwen3_moe/modeling_qwen3_moe.py) ----
 * The reference calls the HF target (model/dflash.py:249-255); for an MoE target these three replace its
 * Qwen3MoeSparseMoeBlock on the <= 16 block rows:
 *  dfl_moe_route: router logits bf16 [16][ld] (the gate Linear's output: dfl_gemm_resid, no residual) -> softmax in fp32 ->
 *    top_k (<= 8; ties: lower expert first) -> divided by their sum (norm_topk) -> bf16 -> wt bf16 [16][E] (0 = not routed);
 *    active int32 [E]; list = the active experts in ascending order, *n_active their number.  Rows >= dyn[dyn_word] route
 *    nowhere.  E <= 256.
 *  dfl_gemm_silu_mul_experts: dfl_gemm_silu_mul for every ACTIVE expert (list[0 .. *n_active)) in one launch: the
 *    workgroups share out the (expert, gate/up column pair) items; expert e's packed gate/up weight at wp + e *
 *    wp_expert_stride elements, its frag16 output at act_frag + e * act_expert_stride — back to back: the strides must
 *    be exactly 2*I*K and 16*I.
 *  dfl_moe_down: out[c][m][n] = sum over the c-th share of the active experts of wt[m][e] * (act_e[m] . Wd_e[n]) as fp32
 *    (nsplit shares; dfl_norm_pack adds them to the residual stream and rounds once, where the reference rounds every
 *    expert's output and accumulates in bf16: fewer roundings, inside the bf16 tolerance). */
int dfl_moe_route(const void *logits, int ld, int E, int top_k, int norm_topk, void *wt, int32_t *active, int32_t *list,
                  int32_t *n_active, const int32_t *dyn, int dyn_word, void *stream);
/* The block's post-attention RMSNorm (Qwen3MoeRMSNorm), the gate Linear and dfl_moe_route in ONE launch (round 4; they are
 * latency, not bytes: 15 % of a 48-layer verify as three launches).  h != NULL: rows [16][ldh] bf16 are normalised with
 * norm_w / eps and written to xn_frag (frag16 [K/8][16][8], the expert GEMMs' source; rows >= dyn[dyn_word]: zeros);
 * h == NULL: xn_frag already holds the normalised rows.  wp_router: the gate weight [E padded to 16][K] packed
 * (dfl_pack_weight); rlog bf16 [16][ld] receives the logits (ld >= padded E, ld % 8 == 0); wt / active / list / n_active
 * as dfl_moe_route writes them (same arithmetic).  ticket: one int32, zero before the first launch, left zero by every
 * launch.  K <= 4096, E even. */
int dfl_moe_router(const void *h, int64_t ldh, const void *norm_w, float eps, void *xn_frag, const void *wp_router, int K,
                   int E, int top_k, int norm_topk, void *rlog, int ld, void *wt, int32_t *active, int32_t *list,
                   int32_t *n_active, const int32_t *dyn, int dyn_word, int32_t *ticket, void *stream);
int dfl_gemm_silu_mul_experts(const void *wp_gateup, int64_t wp_expert_stride, const dfl_rows *x, int E, int I, int K,
                              void *act_frag, int64_t act_expert_stride, const int32_t *list, const int32_t *n_active,
                              const int32_t *dyn, void *stream);
/* dfl_gemm_silu_mul_experts for K <= 2048 and frag16 rows (x_frag: [K/32][64] fragments): an item is the (gate, up)
 * tile PAIR of an active expert, the 16 waves of a workgroup meet once per pair (at K = 2048 a tile is 64 KB and the
 * per-tile meeting of the general kernel weighs twice what it does at K = 4096).  Same results, same layouts; rows
 * >= dyn[valid_word] count as zero (valid_word < 0 or dyn NULL: all 16). */
int dfl_moe_gate_up(const void *wp_gateup, const void *x_frag, int E, int I, int K, void *act_frag, const int32_t *list,
                    const int32_t *n_active, const int32_t *dyn, int valid_word, void *stream);
int dfl_moe_down(const void *wp_down, int64_t wp_expert_stride, const void *act_frag, int64_t act_expert_stride,
                 const void *wt, const int32_t *list, const int32_t *n_active, int E, int N, int I, int nsplit, float *out,
                 void *stream);

/* First-max-index argmax over the last axis (model/utils.py:28-29).
 * dtype: 0 = bf16, 1 = fp32.  ids int64 [rows]. */
int dfl_argmax(const void *logits, int dtype, int rows, int64_t V, int64_t *ids, void *stream);

/* Acceptance scan + commit + bonus token + stop test + length bookkeeping in one
 * wavefront (model/dflash.py:258-268):
 *   acc = #leading i with block[i+1] == posterior[i]           (0..bs-1)
 *   output_ids[start .. start+acc] = block[0..acc]; output_ids[start+acc+1] = posterior[acc]
 *   dyn: S <- start, tau <- acc+1, pos0 <- start, start <- start+acc+1, cycle++, stop |= any stop id
 *        among the acc+2 tokens just written
 * result (int32[4], may be pinned host memory mapped to the device): {acc, new_start, stop, cycle}; words 0..2 are
 * stored first, the cycle counter (>= 1) last behind a system-scope release: a polling host waits for word 3 to change
 * and reads the others afterwards. */
int dfl_accept_commit(const int64_t *block_ids, const int64_t *posterior, int bs, int64_t *output_ids,
                      int64_t output_len, int32_t *dyn, const int64_t *stop_ids, int n_stop, int32_t *result,
                      void *stream);

/* dfl_accept_commit that also RE-ARMS the next cycle's block on the device (model/dflash.py:235: block =
 * output_ids[:, start:start+bs] = the token just committed at the new start followed by mask ids):
 * next_block[0] <- posterior[acc], next_block[1 .. rearm_n-1] <- mask_id (next_block may be block_ids itself). */
int dfl_accept_commit_rearm(const int64_t *block_ids, const int64_t *posterior, int bs, int64_t *output_ids,
                            int64_t output_len, int32_t *dyn, const int64_t *stop_ids, int n_stop, int32_t *result,
                            int64_t *next_block, int rearm_n, int64_t mask_id, void *stream);
/* The same, and the block-form length record of the NEXT target verify maintained on the device as well (dyn_t:
 * S = POS0 = START = new start, TAU = 0; its BS word is left alone): with it a steady-state cycle needs no host-side length
 * at all — verify and accept can be replayed from a hipGraph (DecodeSession.capture). */
int dfl_accept_commit_rearm_t(const int64_t *block_ids, const int64_t *posterior, int bs, int64_t *output_ids,
                              int64_t output_len, int32_t *dyn, const int64_t *stop_ids, int n_stop, int32_t *result,
                              int64_t *next_block, int rearm_n, int64_t mask_id, int32_t *dyn_t, void *stream);

/* ======================================================================================
 * Target PREFILL on the kernels (model/dflash.py:218-225: target(input_ids, ..., output_hidden_states=True) over the
 * P prompt rows).  Same operands as the decode path: weights as packed by dfl_pack_weight /
 * dfl_pack_weight_gateup, activations as frag16 row tiles — tile t (rows 16t .. 16t+15) of K columns at
 * x_frag + t * 16 * K elements, dfl_prefill_rows_padded(P) / 16 tiles (rows are padded to a multiple of 128; the
 * padding holds zero fragments).  LDS-tiled MFMA GEMM (128 x 128 block tiles, LDS-DMA staging).  N % 128 == 0,
 * K % 64 == 0.  Row buffers must hold dfl_prefill_rows_padded(P) rows; rows >= P are never stored.
 * ====================================================================================== */
int64_t dfl_prefill_rows_padded(int P);

/* out[m][n] = bf16(x[m] . W[n])  (bf16 rows, row stride ldo): the q/k/v projection of the prompt rows. */
int dfl_prefill_gemm_rows(const void *wp, const void *x_frag, int P, int N, int K, void *out, int64_t ldo, void *stream);

/* h_io[m][n] = bf16(h_io[m][n] + bf16(x[m] . W[n])) (o_proj / down_proj + residual add); tap (optional) gets a copy
 * of the new rows (row stride ldtap): a tapped target layer, model/utils.py:16-25. */
int dfl_prefill_gemm_resid(const void *wp, const void *x_frag, int P, int N, int K, void *h_io, int64_t ldh, void *tap,
                           int64_t ldtap, void *stream);

/* act = bf16(silu(gate) * up) over the interleaved gate/up weight (tf:modeling_qwen3.py:82), written as frag16 row
 * tiles of I columns (the down projection's input).  I % 64 == 0. */
int dfl_prefill_gemm_silu(const void *wp_gateup, const void *x_frag, int P, int I, int K, void *act_frag, void *stream);

/* Rows h [P][H] (bf16, row stride ldh) -> Qwen3RMSNorm (tf:modeling_qwen3.py:59-64; norm_w NULL: none) -> frag16 row
 * tiles of H columns; all dfl_prefill_rows_padded(P) / 16 tiles are written (zero fragments beyond row P). */
int dfl_prefill_norm_pack(const void *h, int64_t ldh, int P, int H, const void *norm_w, float eps, void *x_frag,
                          void *stream);
/* rows [P][K] (row stride ld) -> frag16 row tiles of K columns, any K % 32 == 0 (the draft's fc operand: K = 5 H tapped
 * states per prompt row, model/dflash.py:166-171); no norm. */
int dfl_prefill_pack_rows(const void *rows, int64_t ld, int P, int K, void *x_frag, void *stream);

/* Per-head q/k RMSNorm (weights may be NULL: Llama) + RoPE at positions pos0 + row over the P rows of a bf16 q/k/v row
 * buffer (row stride ld): q is rewritten in place, k (normed, rotated) and v go to cache rows row0 + row of
 * kcache / vcache [n_kv][cache_rows][128].  cos/sin: bf16 [max_pos][64]. */
int dfl_prefill_qk_rope(void *qkv_rows, int64_t ld, int P, int q_col, int k_col, int v_col, int n_q, int n_kv,
                        const void *q_norm_w, const void *k_norm_w, float eps, const void *cos_tab, const void *sin_tab,
                        int max_pos, int pos0, void *kcache, void *vcache, int cache_rows, int row0, void *stream);

/* Causal attention of the P prompt rows over themselves (GQA, head_dim 128): q = the rows dfl_prefill_qk_rope left in
 * the q/k/v row buffer (row stride ldq, q columns from q_col), K/V = cache rows [0, P) it wrote.  softmax(q k^T * scale)
 * v in fp32 with P rounded to bf16 for the PV product (as dfl_attn_head).  out_frag: frag16 row tiles of n_q * 128
 * columns (o_proj's operand); the tiles of rows >= P within the last written tile are zero. */
int dfl_prefill_attn(const void *q_rows, int64_t ldq, int q_col, const void *kcache, const void *vcache, int cache_rows,
                     int P, int n_q, int n_kv, float scale, void *out_frag, void *stream);

/* Sparse-MoE MLP of the prefill: Qwen3MoeSparseMoeBlock over the P prompt rows of one layer of an MoE target
 * (tf:models/qwen3_moe/modeling_qwen3_moe.py: Qwen3MoeTopKRouter.forward + Qwen3MoeExperts.forward; the reference
 * reaches it through the HF forward of model/dflash.py:218-225, a Python loop over the experts).  The layer's
 * (row, slot) pairs are sorted by expert on the device and each expert's rows gathered into 16-row frag16 tiles, so
 * every expert's packed weights are read once by MFMA row blocks of its own rows.  Call order per layer:
 *   router logits = dfl_prefill_gemm_rows(router weight padded to a multiple of 128 rows) on the ln2-normalised tiles
 *   dfl_prefill_moe_route   fp32 softmax / top-k (probability desc, index asc) / renormalise per row -> pair_e / pair_w
 *                           [P][8]; then (one workgroup) per-expert counts cnt and tile offsets, the work list items =
 *                           (expert, first gathered tile, tiles <= 4) with n_items[0] = items, [1] = tiles, posmap
 *                           [P][8] = gathered row of each pair, src_row / row_w [gathered row] = its source row (-1:
 *                           padding) and routing weight (bf16 value, as float)
 *   dfl_prefill_moe_gather  the source rows' normalised fragments into the gathered tiles xg (zero for padding rows)
 *   dfl_prefill_moe_gemm_silu   act_g = silu(xg Wg_e^T) * (xg Wu_e^T) per expert (gate/up interleaved as dfl_pack_weight_gateup);
 *                               with src_row (and 1 KiB of zeros for padding rows) `xg` is the UNGATHERED x_frag and the
 *                               gather happens in the kernel's LDS-DMA addresses: no dfl_prefill_moe_gather call, no xg
 *   dfl_prefill_moe_gemm_down   out32[gathered row][H] = row_w * (act_g Wd_e^T), fp32
 *   dfl_prefill_moe_combine     h[m] = bf16(h[m] + bf16(sum over the row's k slots of out32)) (+ tap copy): the sum over
 *                               experts is rounded once (HF: per-expert bf16 adds), as in dfl_moe_down's consumer.
 *                               sum_out != NULL: the fp32 sums [P][H] are stored there instead and h is left alone (the
 *                               ragged-batch decode path adds them in its next dfl_norm_frag_batch).
 * Scratch is caller-owned: cnt / tile_off int32 [E], items int32 [3 * max_items], src_row int32 / row_w float
 * [max_tiles * 16], xg bf16 [max_tiles * 16 * H], act_g bf16 [max_tiles * 16 * I], out32 float [max_tiles * 16 * H].
 * rows_per_item: 64 or 128 rows of one expert per work item (4 or 8 gathered tiles), the same in the three calls of a
 * layer that take it; 128 reads an expert's weights once where it serves up to 128 rows.  E <= 256, top_k <= 8, H % 128 == 0, I % 64 == 0, expert strides = elements per expert. */
int64_t dfl_prefill_moe_max_tiles(int P, int top_k, int E);
int64_t dfl_prefill_moe_max_items(int P, int top_k, int E);
int dfl_prefill_moe_route(const void *logits, int64_t ld, int P, int E, int top_k, int norm_topk, int32_t *pair_e,
                          float *pair_w, int32_t *cnt, int32_t *tile_off, int32_t *items, int32_t *n_items, int32_t *posmap,
                          int32_t *src_row, float *row_w, int rows_per_item, void *stream);
int dfl_prefill_moe_gather(const void *x_frag, int P, int H, int top_k, int E, const int32_t *src_row, const int32_t *n_items,
                           void *xg, void *stream);
int dfl_prefill_moe_gemm_silu(const void *wp_gateup_e, int64_t w_expert_stride, const void *xg, const int32_t *items,
                              const int32_t *n_items, int max_items, int I, int K, void *act_g, int rows_per_item,
                              const int32_t *src_row, const void *zeros_1k, void *stream);
int dfl_prefill_moe_gemm_down(const void *wp_down_e, int64_t w_expert_stride, const void *act_g, const int32_t *items,
                              const int32_t *n_items, int max_items, int H, int I, const float *row_w, float *out32,
                              int rows_per_item, void *stream);
int dfl_prefill_moe_combine(const float *out32, const int32_t *posmap, int P, int H, int top_k, void *h_io, int64_t ldh,
                            void *tap, int64_t ldtap, float *sum_out, void *stream);

/* ======================================================================================
 * Ragged batch of requests on one GPU (BASELINE.json configs[2]; SURVEY.md §8e: "within a
 * GPU the requests are a ragged batch for the kernels: shared weight stream, per-request
 * S and tau").  The reference has no batched form of the path — "batch" there is a Python
 * loop over prompts (benchmark_batched.py:212-243, benchmark.py:445) — so these entry
 * points replace R consecutive passes of model/dflash.py:235-268 by one pass in which the
 * R requests share every weight byte.  Request r (0 <= r < R <= 4 per launch):
 *   - 16-row tile at `base + r * stride` of every activation buffer,
 *   - lengths at dyn + r * DFL_DYN_WORDS,
 *   - KV cache at cache + r * cache_req_stride.
 * The GEMMs are compiled for 2 or 4 tiles: buffers (and dyn) must hold dfl_batch_tiles(R)
 * requests; the unused ones need valid (readable) memory and dyn words of zero.
 * ====================================================================================== */
typedef struct dfl_rows_batch {
  dfl_rows r0;          /* request 0: mode 0 (frag16) or 1 (plain rows); mode 2 is rejected — take
                         * normalised rows from dfl_norm_frag_batch */
  int64_t frag_stride;  /* bf16 elements between the requests' frag16 buffers (mode 0), %8 == 0 */
  int64_t rows_stride;  /* bf16 elements between the requests' row buffers (mode 1, 2) */
  int64_t ss_stride;    /* floats between the requests' sum-of-squares partials (mode 2) */
} dfl_rows_batch;

int dfl_batch_tiles(int R);  /* 2 for R <= 2, else 4 */
int dfl_batch_ksplit(int K); /* K parts of 2048 the batched GEMMs cut K into (grid.y) */
/* workspace of the fused-epilogue batched GEMMs: arrival tickets (ZEROED once by the caller,
 * left zero by every launch), argmax candidates, fp32 partial tiles of the K parts */
int64_t dfl_gemm_batch_ws_bytes(int N, int K);

/* dfl_gemm_f32 for R requests: out[c][r*16+m][n], c < dfl_batch_ksplit(K) partial sums
 * (out holds ksplit * dfl_batch_tiles(R) * 16 * N floats). */
int dfl_gemm_f32_batch(const void *wp, const dfl_rows_batch *x, int R, int N, int K, float *out, const int32_t *dyn,
                       void *stream);
/* dfl_gemm_silu_mul for R requests; request r's frag16 output at act_frag + r * act_stride. */
int dfl_gemm_silu_mul_batch(const void *wp_gateup, const dfl_rows_batch *x, int R, int I, int K, void *act_frag,
                            int64_t act_stride, void *ws, const int32_t *dyn, void *stream);
/* dfl_gemm_resid for R requests (h_io, tap, ss_out advance by their strides per request). */
int dfl_gemm_resid_batch(const void *wp, const dfl_rows_batch *x, int R, int N, int K, void *h_io, int64_t ldh,
                         int64_t h_stride, int add_residual, void *tap, int64_t ldtap, int64_t tap_stride,
                         float *ss_out, int64_t ss_stride, void *ws, const int32_t *dyn, void *stream);
/* dfl_gemm_argmax for R requests: out_ids[r * out_stride + out_off + (row - row0)]. */
int dfl_gemm_argmax_batch(const void *wp, const dfl_rows_batch *x, int R, int V, int K, int row0, int nrows,
                          const int32_t *dyn, int nrows_dyn_word, void *ws, int64_t *out_ids, int64_t out_stride,
                          int out_off, void *logits, int64_t logits_stride, void *stream);
/* dfl_embed_rows for R requests: ids[r * ids_stride + m]. */
int dfl_embed_rows_batch(const void *embed, const int64_t *ids, int64_t ids_stride, int R, void *h_out,
                         int64_t h_stride, int H, float *ss_out, int64_t ss_stride, const int32_t *dyn, int dyn_word,
                         void *stream);

/* Residual add + RMSNorm of R requests' rows straight into frag16 (model/dflash.py:140,144 +
 * Qwen3RMSNorm, tf:models/qwen3/modeling_qwen3.py:59-64):
 *   h[r][m]   <- part ? bf16(h[r][m] + bf16(sum_{k<nsplit} part[k*part_split + (r*16+m)*ldp ..])) : h[r][m]
 *   tap[r][m] <- that row (optional: a tapped target layer, model/utils.py:16-25)
 *   frag[r]   <- norm_w * bf16(h * rstd);  rows >= dyn[r][dyn_word]: frag zeroed, h untouched.
 * `part` = the fp32 K-part sums of the o_proj / down_proj GEMM just before it
 * (dfl_gemm_f32_batch): the parts meet at this launch boundary — the kernel reads every h row
 * anyway — instead of inside the GEMM.  The batched GEMMs read their normalised operand from
 * here (the in-GEMM norm of the single-request path is replicated in every workgroup, which
 * at 4 tiles costs more than this launch). */
int dfl_norm_frag_batch(void *h, int64_t h_stride, int64_t ldh, int R, const float *part, int nsplit,
                        int64_t part_split, int ldp, void *tap, int64_t ldtap, int64_t tap_stride, const void *norm_w,
                        float eps, void *frag, int64_t frag_stride, int H, const int32_t *dyn, int dyn_word,
                        void *stream);

/* Context K/V of ALL draft layers for R requests in one launch (model/dflash.py:73-85, the
 * context half): kv = fp32 partials of the context rows times the concatenated k/v weights
 * of the n_layers layers (layer i's k columns at k_col + i * col_layer_stride, v alike);
 * k_norm + RoPE (position pos0 + row) -> caches at rows S + row, row < dyn tau.  The block
 * stage that follows then sees them as cached rows (lengths in block form, see
 * dfl_accept_commit_batch). */
int dfl_kv_append_batch(const float *kv, int nsplit, int64_t split_stride, int ld, int k_col, int v_col,
                        int col_layer_stride, int n_layers, int R, int req_rows, int n_kv, const void *k_norm_w,
                        int64_t kw_layer_stride, float eps, const void *cos_tab, const void *sin_tab, int max_pos,
                        void *kcache, void *vcache, int cache_rows, int64_t cache_req_stride,
                        int64_t cache_layer_stride, const int32_t *dyn, void *stream);
/* The same over n_tiles 16-row TILES with one length record each (dyn_tiles), tiles_per_req consecutive tiles sharing one
 * request's cache: the context rows of requests that run 17..32-row blocks (up to 32 accepted rows per cycle). */
int dfl_kv_append_batch_t(const float *kv, int nsplit, int64_t split_stride, int ld, int k_col, int v_col,
                          int col_layer_stride, int n_layers, int n_tiles, int req_rows, int n_kv, const void *k_norm_w,
                          int64_t kw_layer_stride, float eps, const void *cos_tab, const void *sin_tab, int max_pos,
                          void *kcache, void *vcache, int cache_rows, int64_t cache_req_stride,
                          int64_t cache_layer_stride, const int32_t *dyn_tiles, int tiles_per_req, void *stream);

/* dfl_attn_fused for R requests (grid.z = request): request r's block rows at partial-buffer
 * rows blk_row0 + r * req_rows, cache at + r * cache_req_stride, frag16 output at
 * + r * out_req_stride; no context rows in this form (dyn tau == 0).
 * ws: dfl_attn_fused_batch_ws_bytes bytes, zeroed once. */
int64_t dfl_attn_fused_batch_ws_bytes(int R, int n_q, int n_kv, int max_splits);
int dfl_attn_fused_batch(const float *qkv, int nsplit, int64_t split_stride, int ld, int q_col, int k_col, int v_col,
                         int blk_row0, int req_rows, int R, int n_q, int n_kv, const void *q_norm_w,
                         const void *k_norm_w, float eps, const void *cos_tab, const void *sin_tab, int max_pos,
                         void *kcache, void *vcache, int cache_rows, int64_t cache_req_stride, float scale, int causal,
                         const int32_t *dyn, int kv_len_max, void *ws, int max_splits, void *out_frag,
                         int64_t out_req_stride, void *stream);

/* dfl_accept_commit for R requests, one wavefront each (model/dflash.py:258-268 per request).
 * bs is read from dyn_d.  Updates dyn_d (draft form: S <- start, tau <- acc+1, pos0 <- start,
 * start <- start+acc+1) and dyn_t (block form: S = pos0 = start = new start, tau = 0), the
 * latter read by the target verify and by the draft's block stage of the next cycle.
 * result int32 [R][4] = {acc, new_start, stop, cycle}.  A request with dyn_d bs == 0 is idle.
 * next_block (optional, may alias block_ids; same stride): re-armed for the next cycle as
 * [bonus token, mask_id x 15] = output_ids[new start .. +16) (model/dflash.py:235). */
int dfl_accept_commit_batch(const int64_t *block_ids, int64_t blk_stride, const int64_t *posterior,
                            int64_t post_stride, int R, int64_t *output_ids, int64_t out_stride, int64_t output_len,
                            int32_t *dyn_d, int32_t *dyn_t, const int64_t *stop_ids, int n_stop, int32_t *result,
                            int64_t *next_block, int64_t mask_id, void *stream);
/* The same for requests of tiles_per_req (1 or 2) 16-row tiles — blocks of up to 32 rows: besides the per-request records
 * it keeps one record per TILE for the per-tile launches (dyn_d_tiles: S / pos0 = start + 16 j, tau = the tile's share of
 * the acc + 1 context rows; dyn_t_tiles: block form with bs = the tile's share of the block rows); the re-armed block has
 * 16 * tiles_per_req slots. */
int dfl_accept_commit_batch_t(const int64_t *block_ids, int64_t blk_stride, const int64_t *posterior,
                              int64_t post_stride, int R, int64_t *output_ids, int64_t out_stride, int64_t output_len,
                              int32_t *dyn_d, int32_t *dyn_t, const int64_t *stop_ids, int n_stop, int32_t *result,
                              int64_t *next_block, int64_t mask_id, int tiles_per_req, int32_t *dyn_d_tiles,
                              int32_t *dyn_t_tiles, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFLASH_HIP_H */
