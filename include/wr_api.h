/*
 * wr_api.h -- C ABI of libwr_mi355x.so: the MI355X (gfx950) implementation of
 * WeNet's CTC / RNN-T loss and transducer-decode hot path.
 *
 * This is the drop-in boundary.  The reference has no
 * native interface on this path: its seams are Python call sites into
 * third-party libraries.  Each entry point below names the reference call site
 * it replaces; INTEGRATION.md shows the ctypes binding a maintainer would add
 * on the reference side.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / C++ types.
 *  - Every pointer named *_d is DEVICE memory owned by the caller (PyTorch's
 *    allocator in practice).  The library never allocates or frees device
 *    memory and never synchronises the device: all work is enqueued on the
 *    caller's stream (`stream` is a hipStream_t passed as void*; NULL = the
 *    default stream).  Entry points are re-entrant; there is no global mutable
 *    state besides a thread-local error string.
 *  - Return value: 0 on success, a negative WR_E* code otherwise;
 *    wr_last_error() returns a human-readable message for the calling thread.
 *    Non-finite losses are values, not errors (the reference skips such steps
 *    upstream: wenet/utils/executor.py:124-125,171).
 *  - dtype codes: WR_F32 (the parity bar), WR_F16, WR_BF16 (AMP; accumulate in
 *    fp32, gradients returned in the input dtype).
 */
#ifndef WR_API_H_
#define WR_API_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an existing signature or struct layout changes (callers built against another version must not
 * bind): 1 = round 1; 2 = round 2 (an `activation` argument inside wr_joint_fwd / wr_joint_bwd_dz / *_split,
 * wr_transducer_weights grew the predictor-variant fields); 3 = this header. */
#define WR_API_VERSION 3

enum wr_dtype { WR_F32 = 0, WR_F16 = 1, WR_BF16 = 2 };
/* joiner activation (TransducerJoint(activation=...), wenet/transducer/joint.py:25 -> wenet/utils/common.py:228-242);
 * "tanh" is the shipped configuration, "swish" is torch.nn.SiLU, "gelu" the erf form (torch.nn.GELU default) */
enum wr_activation { WR_ACT_TANH = 0, WR_ACT_RELU = 1, WR_ACT_HARDTANH = 2, WR_ACT_SELU = 3, WR_ACT_SWISH = 4, WR_ACT_GELU = 5 };

enum wr_status {
    WR_OK = 0,
    WR_EINVAL = -1,      /* bad argument (null pointer, non-positive size, blank out of range ...) */
    WR_EUNSUPPORTED = -2,/* shape outside what the kernels cover (message says which limit) */
    WR_EWORKSPACE = -3,  /* workspace too small */
    WR_ELAUNCH = -4      /* HIP reported a launch error */
};

int wr_api_version(void);
const char *wr_last_error(void);

/* Benchmarking hook: process-wide launch-shape knobs of the streaming kernels.  Results never depend
 * on them; the defaults are the measured best.  key 0: workgroups per CU of the RNN-T row-lse pass,
 * key 1: of the RNN-T gradient pass (0, the default: automatic -- one row per wave in the row-lse pass, about 68 KB of
 * logits per wave in the gradient pass; n > 0: a persistent grid of n workgroups per CU as in round 1), key 2: non-temporal bits (1: gradient-pass loads, 2: gradient-pass
 * stores, 4: row-lse loads; default 7), key 3 / key 4: 16-byte vectors in flight per lane in the gradient (4, 8 or 16;
 * default 16) / row-lse pass (4, 8 or 16; default 16),
 * key 5: exact-fp32 joiner forward (default: fragment-layout operands; 1: the first forward kernel, bit-identical), key 6: decoder GEMM lane
 * tile (0: by occupancy, 1: 32 lanes, 2: 64 lanes; for GEMMs of more than 256 tiles 3: 4-wave workgroups, 4: one wave
 * per tile instead of 128-lane workgroups -- every form gives bit-identical results), key 7: column parts of the split
 * joiner forward (0: automatic),
 * key 8: retired (the 64-cell split dZ tiling was removed), key 9 / key 10: exact dW / dZ tiling (0: 256 x 256 blocks,
 * 1: the first tilings of joint.hip), key 11: greedy / beam micro-step with an LSTM predictor (0: projection folded
 * into pred_ffn -- one launch less, the default; 1: two launches), key 12: single-term (AMP) split joiner forward
 * (0: 64 lattice cells per workgroup, two workgroups per CU, the default; 1: 64 cells, one workgroup per CU; 2: 128
 * cells -- measured slower), key 13: its logit stores (0: cell-major tiles through a per-wave LDS stage as whole lines, the
 * default; 1: transposed tiles stored from registers in 8/16-byte pieces -- measured slower).  Keys 6-13 pick between kernels that compute the
 * same sums; the two dZ tilings are bit-identical, the dW tilings differ in the order of fp32 additions, the folded
 * projection in the rounding of one composed weight matrix (formed in float64). */
int wr_tune_set(int key, int value);

/* ------------------------------------------------------------------------
 * RNN-T loss + gradient w.r.t. the joiner logits (log-softmax fused).
 * Replaces torchaudio.functional.rnnt_loss(logits, targets, logit_lengths,
 * target_lengths, blank, clamp, reduction) as called at
 *   wenet/transducer/transducer.py:142-147  (training, reduction="mean")
 *   wenet/transducer/transducer.py:296-301  (rescoring, reduction='none').
 *
 * Shapes (as torchaudio): logits [B, Tmax, U1max, V] contiguous,
 * targets [B, U1max-1] int32, logit_lengths [B] int32, target_lengths [B] int32.
 * U1max = max target length + 1.
 *
 * The op is split where autograd splits it, so that the logits-sized tensor is
 * touched the algorithmic minimum of three times (read, read, write):
 *   wr_rnnt_loss_fwd : pass 1 (row log-sum-exp, blank/label log-probs) +
 *                      alpha/beta lattice sweeps -> costs[B]; lattice state is
 *                      kept in the caller's workspace.
 *   wr_rnnt_loss_bwd : pass 3, grads = grad_costs[b] * d cost_b / d logits,
 *                      zero outside [0,T_b) x [0,U_b].  `grads_d` may alias
 *                      `logits_d` (in-place).  grad_costs_d may be NULL (= 1).
 * Reduction ("mean" = mean over batch, not length-normalised) is the caller's:
 * it is a B-element operation folded into grad_costs.
 * ---------------------------------------------------------------------- */
size_t wr_rnnt_workspace_bytes(int B, int Tmax, int U1max);

int wr_rnnt_loss_fwd(const void *logits_d, int dtype,
                     const int32_t *targets_d, const int32_t *logit_lengths_d,
                     const int32_t *target_lengths_d,
                     int B, int Tmax, int U1max, int V, int blank,
                     float *costs_d /* [B] out */,
                     void *workspace_d, size_t workspace_bytes, void *stream);

int wr_rnnt_loss_bwd(const void *logits_d, int dtype,
                     const int32_t *targets_d, const int32_t *logit_lengths_d,
                     const int32_t *target_lengths_d,
                     int B, int Tmax, int U1max, int V, int blank, float clamp,
                     const float *grad_costs_d /* [B] or NULL */,
                     void *grads_d /* same shape/dtype as logits; may alias */,
                     const void *workspace_d, size_t workspace_bytes, void *stream);

/* The lattice sweeps alone, for a workspace whose row statistics (denom and the skip / emit log-probabilities) were
 * already produced by the joiner's fused epilogue (wr_joint_fwd_lse / wr_joint_fwd_split_lse below): pass 1 -- one
 * full read of the logits -- is skipped.  The epilogue sums exponentials against the first logit a lane sees instead
 * of a running maximum; if a row spreads over more than 88 nats that sum overflows, the epilogue raises a flag in
 * the workspace and this entry point runs the stand-alone pass 1 over `logits_d` (fp32, as written by the joiner)
 * before the sweeps -- otherwise that kernel returns at once.  costs as wr_rnnt_loss_fwd; wr_rnnt_loss_bwd follows
 * unchanged. */
int wr_rnnt_loss_fwd_from_lse(const float *logits_d, const int32_t *targets_d,
                              const int32_t *logit_lengths_d, const int32_t *target_lengths_d,
                              int B, int Tmax, int U1max, int V, int blank, float *costs_d /* [B] out */,
                              void *workspace_d, size_t workspace_bytes, void *stream);

/* Diagnostic view of the lattice state left in the workspace by wr_rnnt_loss_fwd
 * (tests compare alpha/beta with the oracle): copies alpha and beta into plain
 * [B, Tmax, U1max] float arrays (entries outside the valid region are 0). */
int wr_rnnt_export_lattice(const void *workspace_d, size_t workspace_bytes,
                           const int32_t *logit_lengths_d, const int32_t *target_lengths_d,
                           int B, int Tmax, int U1max,
                           float *alpha_d, float *beta_d, void *stream);

/* ------------------------------------------------------------------------
 * CTC loss + gradient with the log-softmax fused in.
 * Replaces  ys_hat = ys_hat.log_softmax(2); loss = nn.CTCLoss(...)(ys_hat, ys_pad, hlens, ys_lens)
 * at wenet/transformer/ctc.py:60-61 (blank = 0, zero_infinity = False).
 *
 * logits [B, Tmax, V] contiguous = the PRE-softmax output of ctc_lo, batch-major
 * (no transpose needed); targets [B, Smax] int32, entries beyond
 * target_lengths[b] are never read; lengths [B] int32.  nll[b] = -log p(y_b | x_b)
 * per utterance (infeasible alignment -> +inf, a value, not an error);
 * reduction ('sum' then /B in the reference, ctc.py:61-63) is the caller's and
 * is folded into grad_nll.  grads = grad_nll[b] * d nll_b / d logits, zero for
 * t >= input_lengths[b]; grads_d may alias logits_d.  Limits: Smax <= 511,
 * V <= 16384.
 * ---------------------------------------------------------------------- */
size_t wr_ctc_workspace_bytes(int B, int Tmax, int Smax);

int wr_ctc_loss_fwd(const void *logits_d, int dtype,
                    const int32_t *targets_d, const int32_t *input_lengths_d,
                    const int32_t *target_lengths_d,
                    int B, int Tmax, int Smax, int V, int blank,
                    float *nll_d /* [B] out */,
                    void *workspace_d, size_t workspace_bytes, void *stream);

int wr_ctc_loss_bwd(const void *logits_d, int dtype,
                    const int32_t *targets_d, const int32_t *input_lengths_d,
                    const int32_t *target_lengths_d,
                    int B, int Tmax, int Smax, int V, int blank,
                    const float *grad_nll_d /* [B] or NULL */,
                    void *grads_d,
                    const void *workspace_d, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------
 * Transducer joint network (the one dense contraction on the path).
 * Replaces TransducerJoint.forward, wenet/transducer/joint.py:45-70, from the
 * point where the two pre-join projections exist:
 *   ep = enc_ffn(enc) [B, T, J],  pp = pred_ffn(pred) [B, U1, J]   (joint.py:55-58)
 *   out[b,t,u,:] = ffn_out(act(ep[b,t,:] + pp[b,u,:]))              (joint.py:60-69)
 * `activation` is a wr_activation code (tanh in the shipped configuration; the text below writes tanh for it).
 * w_out [V, J] and b_out [V] are ffn_out.weight / .bias in nn.Linear layout.
 * fp32 throughout (exact-fp32 MFMA).  J a multiple of 4, at most 512.
 * The activation tensor tanh(ep+pp) [B,T,U1,J] is never written to HBM in the
 * forward pass.  If both length arrays are given, 64-cell tiles that lie wholly
 * in the padded region (t >= T_b or u > U_b) are skipped and `out` is left
 * untouched there -- the RNN-T loss never reads those cells.
 *
 * wr_joint_bwd_dz: dz[b,t,u,:] = (gout[b,t,u,:] @ w_out) * (1 - tanh(ep+pp)^2),
 * zero in padded cells when lengths are given; h_d (optional, [B,T,U1,J])
 * receives tanh(ep+pp) for the weight gradient (wr_joint_bwd_dw).  d ep = sum_u dz
 * and d pp = sum_t dz are plain library reductions on the host side.
 * ---------------------------------------------------------------------- */
size_t wr_joint_workspace_bytes(int J, int V);

int wr_joint_fwd(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                 const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                 int B, int T, int U1, int J, int V, int activation,
                 float *out_d /* [B,T,U1,V] */,
                 void *workspace_d, size_t workspace_bytes, void *stream);

/* wr_joint_fwd with pass 1 of the RNN-T loss fused into its epilogue (transducer.py:132 + the first pass of :142-147):
 * a forward workgroup owns every column of its 64 lattice cells, so while the logits leave for HBM it also writes
 * denom(t,u) = logsumexp_v and the blank / label log-probabilities of every valid cell into the RNN-T workspace
 * (wr_rnnt_workspace_bytes(B, T, U1)); wr_rnnt_loss_fwd_from_lse then only runs the lattice sweeps.  Both length
 * arrays are required; targets [B, U1-1] as for wr_rnnt_loss_fwd. */
int wr_joint_fwd_lse(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                     const int32_t *logit_lengths_d, const int32_t *target_lengths_d, const int32_t *targets_d,
                     int B, int T, int U1, int J, int V, int activation, int blank,
                     float *out_d /* [B,T,U1,V] */,
                     void *workspace_d, size_t workspace_bytes,
                     void *rnnt_workspace_d, size_t rnnt_workspace_bytes, void *stream);

int wr_joint_bwd_dz(const float *gout_d /* [B,T,U1,V] */, const float *ep_d, const float *pp_d,
                    const float *w_out_d,
                    const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                    int B, int T, int U1, int J, int V, int activation,
                    float *dz_d /* [B,T,U1,J] */, float *h_d /* [B,T,U1,J] or NULL */, void *stream);

/* Split-precision joiner on the bf16 matrix cores (opt-in; the exact-fp32 entry points above stay the default).
 * Same operator and arguments as wr_joint_fwd; every fp32 operand is split into bf16 hi + lo parts and
 *   terms = 3:  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulation -- logits within 1e-4 (relative to
 *               their scale) of the fp32 result, at several times the rate of the exact-fp32 MFMA;
 *   terms = 1:  a_hi*b_hi only: the AMP path (the reference under --use_amp, executor.py:91, runs ffn_out in
 *               fp16 with fp32 accumulation).
 * out_d has out_dtype (WR_F32 / WR_F16 / WR_BF16).  J a multiple of 4, at most 512 (as the exact entry points). */
size_t wr_joint_split_workspace_bytes(int J, int V);

int wr_joint_fwd_split(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                       const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                       int B, int T, int U1, int J, int V, int activation, int terms,
                       void *out_d /* [B,T,U1,V] */, int out_dtype,
                       void *workspace_d, size_t workspace_bytes, void *stream);

/* wr_joint_fwd_split with the RNN-T loss's row statistics fused into the epilogue (see wr_joint_fwd_lse); fp32 logits. */
int wr_joint_fwd_split_lse(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                           const int32_t *logit_lengths_d, const int32_t *target_lengths_d, const int32_t *targets_d,
                           int B, int T, int U1, int J, int V, int activation, int blank, int terms,
                           float *out_d /* [B,T,U1,V] */,
                           void *workspace_d, size_t workspace_bytes,
                           void *rnnt_workspace_d, size_t rnnt_workspace_bytes, void *stream);

/* wr_joint_bwd_dz on the bf16 matrix cores, same split as wr_joint_fwd_split (terms = 3: gout and w_out split into
 * bf16 hi + lo, three MFMA terms, fp32 accumulation; terms = 1: single bf16 product).  Same outputs as
 * wr_joint_bwd_dz (H = tanh(ep+pp) recomputed with the exact tanhf).  V a multiple of 4, at least 32. */
size_t wr_joint_dz_split_workspace_bytes(int J, int V);

int wr_joint_bwd_dz_split(const float *gout_d /* [B,T,U1,V] */, const float *ep_d, const float *pp_d,
                          const float *w_out_d,
                          const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                          int B, int T, int U1, int J, int V, int activation, int terms,
                          float *dz_d /* [B,T,U1,J] */, float *h_d /* [B,T,U1,J] or NULL */,
                          void *workspace_d, size_t workspace_bytes, void *stream);

/* The same with the logits gradient in bf16 -- what the loss hands back for the 16-bit logits of the AMP step
 * (reference: executor.py:91 autocast; torch casts the Linear's incoming gradient the same way).  bf16 values are their
 * own hi parts: no conversion pass, half the gradient bytes, results identical to wr_joint_bwd_dz_split on the same
 * values widened to fp32.  V a multiple of 8, at least 32. */
int wr_joint_bwd_dz_split_bf16(const void *gout_bf16_d /* [B,T,U1,V] bf16 */, const float *ep_d, const float *pp_d,
                               const float *w_out_d,
                               const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                               int B, int T, int U1, int J, int V, int activation, int terms,
                               float *dz_d /* [B,T,U1,J] */, float *h_d /* [B,T,U1,J] or NULL */,
                               void *workspace_d, size_t workspace_bytes, void *stream);

/* Second half of the activation gradient for callers that form dH = gout . W themselves (the AMP step hands that plain
 * bf16 contraction to the vendor GEMM library): dz[cell, j] = dH[cell, j] * act'(ep + pp) in place (zero in padded cells
 * when lengths are given) and, unless h_d is NULL, h[cell, 0..J) = act(ep + pp) (zero in padded cells) as fp32, fp16 or bf16
 * with row stride h_ld >= J (a multiple of 4).  When h_ld > J, column J of a row is 1 in valid cells and 0 in padded
 * ones and columns J+1 .. h_ld-1 are 0: gout^T h then carries the bias gradient in column J.  J a multiple of 4. */
int wr_joint_dz_act(float *dz_d /* [B,T,U1,J] in: dH, out: dZ */, const float *ep_d, const float *pp_d,
                    const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                    int B, int T, int U1, int J, int activation,
                    void *h_d /* [B,T,U1,h_ld] or NULL */, int h_dtype /* WR_F32 | WR_F16 | WR_BF16 */, int h_ld, void *stream);

/* Bias gradient from a bf16 logits gradient: db[v] = sum over lattice cells of gout[cell, v], padded cells excluded when
 * lengths are given (the AMP step, whose weight gradient is a vendor-library GEMM).  V a multiple of 8.  Deterministic. */
size_t wr_joint_db_workspace_bytes(int B, int T, int U1, int V);

int wr_joint_db_bf16(const void *gout_bf16_d /* [B,T,U1,V] bf16 */,
                     const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                     int B, int T, int U1, int V, float *db_d /* [V] */,
                     void *workspace_d, size_t workspace_bytes, void *stream);

/* The same for a float16 gradient (torch.cuda.amp.autocast's default dtype, the reference's --use_amp). */
int wr_joint_db_f16(const void *gout_f16_d /* [B,T,U1,V] fp16 */,
                    const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                    int B, int T, int U1, int V, float *db_d /* [V] */,
                    void *workspace_d, size_t workspace_bytes, void *stream);

/* Weight gradient of ffn_out:  dw[v, :] = sum over lattice cells of gout[cell, v] * h[cell, :],
 * db[v] = sum of gout[cell, v]  (h = tanh(ep+pp) as written by wr_joint_bwd_dz).  With lengths, cells in the
 * padded region do not contribute.  db_d may be NULL.  Deterministic (partial slabs + ordered reduction). */
size_t wr_joint_dw_workspace_bytes(int J, int V);

int wr_joint_bwd_dw(const float *gout_d /* [B,T,U1,V] */, const float *h_d /* [B,T,U1,J] */,
                    const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                    int B, int T, int U1, int J, int V,
                    float *dw_d /* [V,J] */, float *db_d /* [V] or NULL */,
                    void *workspace_d, size_t workspace_bytes, void *stream);

/* wr_joint_bwd_dw on the bf16 matrix cores (terms as in wr_joint_fwd_split): dw = gout^T h, db = column sums of gout
 * (kept in fp32), padded cells excluded when lengths are given.  V and J multiples of 4.  Deterministic. */
size_t wr_joint_dw_split_workspace_bytes(int B, int T, int U1, int J, int V);

int wr_joint_bwd_dw_split(const float *gout_d /* [B,T,U1,V] */, const float *h_d /* [B,T,U1,J] */,
                          const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                          int B, int T, int U1, int J, int V, int terms,
                          float *dw_d /* [V,J] */, float *db_d /* [V] or NULL */,
                          void *workspace_d, size_t workspace_bytes, void *stream);

/* The same with the logits gradient in bf16 (see wr_joint_bwd_dz_split_bf16); workspace as wr_joint_bwd_dw_split. */
int wr_joint_bwd_dw_split_bf16(const void *gout_bf16_d /* [B,T,U1,V] bf16 */, const float *h_d /* [B,T,U1,J] */,
                               const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                               int B, int T, int U1, int J, int V, int terms,
                               float *dw_d /* [V,J] */, float *db_d /* [V] or NULL */,
                               void *workspace_d, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------
 * Transducer decoding: batched greedy search, batched prefix beam search and
 * the predictor step API, with every per-step operation on the device.
 *
 * Replaces
 *   greedy  wenet/transducer/search/greedy_search copy.py:6-63 -- the upstream core loop behind
 *           Transducer.greedy_search (wenet/transducer/transducer.py:515-598)
 *   beam    PrefixBeamSearch.prefix_beam_search, wenet/transducer/search/prefix_beam_search.py:42-148
 *           (called from Transducer.beam_search, transducer.py:332-377)
 *   step    RNNPredictor.forward_step, wenet/transducer/predictor.py:160-200 (EmbeddingPredictor.forward_step
 *           :325-372 and ConvPredictor.forward_step :455-481 with predictor_type 1 / 2), and the joiner for
 *           step shapes (joint.py:45-70), i.e. forward_predictor_step / forward_joint_step
 *           (transducer.py:613-629)
 *
 * Weights are passed in the reference modules' own (nn.Linear / nn.LSTM / nn.Embedding) layouts;
 * wr_decoder_create re-lays them out k-major once into the caller's workspace.
 * The decoder handle owns no device memory; it holds the launch configuration, the captured
 * hipGraphs of the micro-steps and an internal work stream that is ordered against the caller's
 * stream with events (no device-wide synchronisation).  wr_greedy_search synchronises its own work
 * stream once every 16 micro-steps to test the "lanes still decoding" word (the reference
 * synchronises on every step); wr_prefix_beam_search and wr_predictor_step never do.
 * Limits: lanes <= 1024, vocabulary <= 16384, LSTM layers <= 4, beam <= 16, layer widths <= 1024.
 * ---------------------------------------------------------------------- */
#define WR_MAX_LSTM_LAYERS 4

typedef struct wr_transducer_weights {
    int32_t vocab_size;   /* V */
    int32_t enc_dim;      /* E: encoder output size */
    int32_t pred_dim;     /* P: predictor output size (RNNPredictor.projection out) */
    int32_t embed_dim;    /* D */
    int32_t hidden;       /* H: LSTM hidden size */
    int32_t n_layers;     /* L */
    int32_t join_dim;     /* J */
    int32_t activation;   /* wr_activation of the joiner (0 = tanh) */
    const float *embed;                        /* predictor.embed.weight       [V, D] ([embed_rows, D] if embed_rows > 0) */
    const float *w_ih[WR_MAX_LSTM_LAYERS];     /* predictor.rnn.weight_ih_l{k} [4H, D or H], gate order i,f,g,o */
    const float *w_hh[WR_MAX_LSTM_LAYERS];     /* predictor.rnn.weight_hh_l{k} [4H, H] */
    const float *b_ih[WR_MAX_LSTM_LAYERS];     /* predictor.rnn.bias_ih_l{k}   [4H] */
    const float *b_hh[WR_MAX_LSTM_LAYERS];     /* predictor.rnn.bias_hh_l{k}   [4H] */
    const float *proj_w, *proj_b;              /* predictor.projection         [P, H], [P] */
    const float *enc_ffn_w, *enc_ffn_b;        /* joint.enc_ffn                [J, E], [J] */
    const float *pred_ffn_w, *pred_ffn_b;      /* joint.pred_ffn               [J, P], [J] */
    const float *out_w, *out_b;                /* joint.ffn_out                [V, J], [V] */
    /* The stateless predictors of wenet/transducer/predictor.py:203-481 (not in the shipped configuration).  Their
     * state is the embeddings of the last context_size - 1 tokens; it travels through the same cache tensors as the
     * LSTM state: n_layers = context_size - 1 "layers" of width hidden = embed_dim = pred_dim (slot 0 oldest), the
     * cell-state tensors are carried and ignored.  LSTM fields above are unused (may be NULL) for types 1 and 2. */
    int32_t predictor_type;    /* 0 RNNPredictor (LSTM), 1 EmbeddingPredictor, 2 ConvPredictor */
    int32_t context_size;      /* history_size + 1: 2..5 */
    int32_t n_head;            /* type 1: heads of the positional weighting; n_head * context_size <= 64 */
    int32_t pred_activation;   /* wr_activation applied after the LayerNorm */
    float ln_eps;              /* LayerNorm epsilon */
    int32_t embed_rows;        /* rows of predictor.embed.weight when that differs from V (a predictor stepped on its own
                                * through wr_predictor_step has no joiner vocabulary); 0 = vocab_size.  The LSTM path keeps a
                                * table of W_ih(layer 0) . embed[v] with that many rows. */
    const float *pos_w;                        /* type 1: pos_embed.weight     [n_head, D * context_size] (bias unused) */
    const float *ffn_w, *ffn_b;                /* type 1: ffn                  [D, D], [D] */
    const float *norm_w, *norm_b;              /* types 1, 2: norm             [D], [D] */
    const float *conv_w, *conv_b;              /* type 2: conv.weight [D, 1, context_size], conv.bias [D] or NULL */
} wr_transducer_weights;

typedef struct wr_decoder wr_decoder;

size_t wr_decoder_workspace_bytes(const wr_transducer_weights *w, int max_lanes, int max_utt, int Tmax,
                                  int max_hyp, int max_beam);

/* max_lanes: streams decoded together (greedy) or utterances x beam (beam search);
 * max_utt: utterances per call; Tmax: encoder frames; max_hyp: greedy hypothesis capacity. */
int wr_decoder_create(const wr_transducer_weights *w, int max_lanes, int max_utt, int Tmax, int max_hyp,
                      int max_beam, void *workspace_d, size_t workspace_bytes, void *stream,
                      wr_decoder **out);
int wr_decoder_destroy(wr_decoder *h);
/* hipGraph replay of the per-step kernel sequence.  Default: on for greedy search (the host polls a "lanes still
 * active" word once per replay), off for prefix beam search (fixed frame count, plain launches measured faster).
 * enable = 0 / 1 forces both off / on. */
int wr_decoder_set_graph(wr_decoder *h, int enable);
/* Greedy look-ahead: evaluate the joiner for `frames` (1..4) consecutive encoder frames of every stream per
 * micro-step.  A stream's predictor output only changes after a non-blank, so a run of blanks is consumed in one
 * micro-step instead of `frames`; decisions are walked in frame order through the same state machine and stop at
 * the first emission, so the token sequences are those of the one-frame loop (frames = 1).
 * frames = 0 (the default): chosen before every graph replay from the share of blank decisions in the previous one
 * (>= 75 % blank: 4 frames, >= 62 %: 2, else 1; never more than 256 joiner rows per micro-step). */
int wr_decoder_set_lookahead(wr_decoder *h, int frames);

/* enc_out [N, T, E] fp32, enc_lens [N]; hyps [N, max_hyp] / hyp_lens [N] out (tokens beyond
 * max_hyp are counted in hyp_lens but not stored).  Each lane follows the reference loop exactly:
 * predictor stepped only after a non-blank, at most n_steps emissions per frame. */
int wr_greedy_search(wr_decoder *h, const float *enc_out_d, const int32_t *enc_lens_d, int N, int T,
                     int n_steps, int blank, int32_t *hyps_d, int32_t *hyp_lens_d, void *stream);

/* Streaming (chunk-synchronous) greedy search: the stateful reset_cache() / forward_greedy_search(chunk)
 * pair that the reference's C++ runtime drives ("wenet/transducer/transducer ref.py":541-606, consumed at
 * runtime/core/decoder/torch_asr_model.cc:126,313).  reset != 0 starts N fresh streams; otherwise the
 * predictor cache, last token, "predictor must step" flag, per-frame emission counter and predictor
 * output of every stream carry over from the previous call.  hyps/hyp_lens receive the tokens emitted in
 * THIS chunk.  reference_new_cache != 0 reproduces the reference's `new_cache = self.cache` at the top of
 * each chunk (the not-yet-committed predictor state of the previous chunk is dropped); 0 keeps it, which
 * makes chunked decoding identical to decoding the concatenated frames with wr_greedy_search. */
int wr_greedy_search_chunk(wr_decoder *h, const float *enc_chunk_d, const int32_t *chunk_lens_d, int N, int T,
                           int n_steps, int blank, int reset, int reference_new_cache, int32_t *hyps_d,
                           int32_t *hyp_lens_d, void *stream);

/* enc_out [B, T, E], ctc_logp [B, T, V] = log_softmax(ctc_lo(enc_out)) (ctc.py:66-75).
 * Out: hyps [B, beam, Tmax+1] (each begins with the seed blank, padded with -1), hyp_lens [B, beam],
 * scores [B, beam] float64 (best first), n_hyps [B]. */
int wr_prefix_beam_search(wr_decoder *h, const float *enc_out_d, const int32_t *enc_lens_d,
                          const float *ctc_logp_d, int B, int T, int beam, float ctc_weight,
                          float transducer_weight, int blank, int32_t *hyps_d, int32_t *hyp_lens_d,
                          double *scores_d, int32_t *n_hyps_d, void *stream);

/* One predictor step for N lanes: tokens [N], cache_h / cache_c [L, N, H] in;
 * out [N, P], new_h / new_c [L, N, H] out (padding is applied by the caller, predictor.py:9-15). */
int wr_predictor_step(wr_decoder *h, const int32_t *tokens_d, const float *cache_h_d,
                      const float *cache_c_d, int N, float *out_d, float *new_h_d, float *new_c_d,
                      void *stream);

/* ------------------------------------------------------------------------
 * Hot-word greedy search with the gate inside the device step: the fork's default
 * decode path, wenet/transducer/search/greedy_search.py:297-430 (`basic_greedy_search_both`, selected by
 * loss_mode='both', wenet/transducer/transducer.py:43,559-597), around wenet/transformer/context_bias.py::ContextBias.
 *
 * Per predictor step the reference calls ContextBias.forward_predictor_bias (:375-381: multi-head attention of the
 * predictor output over the encoded hot-word list, LayerNorm, Linear over the concatenation, LayerNorm) and
 * ContextBias.forward_hw_pred_both (:388-394: the two-class "is a hot word being spoken" gate), takes the gate's
 * top-1 on the host (.item()), and with context_filter_state == 'on' runs the "go-back" (:365-392): when the gate
 * flips 0 -> 1 the token emitted after the gate-0 step is withdrawn and decoding resumes from that step's frame with
 * biasing forced on until the frame of the flip.  Here all of it lives in the captured micro-step:
 *   - gate: the classifier attends from ONE query to ONE key (the encoder bias feature of frame t), so its softmax
 *     weight is exactly 1 and its output is a function of the frame alone (q / k projections cannot influence it):
 *     the gate of every frame is evaluated up front by one kernel (hw_gate_table_kernel) and looked up per step;
 *   - predictor biasing: one fused kernel per micro-step (query projection, attention over the list, output
 *     projection, LayerNorm, combine Linear, LayerNorm) for the variant -- hot-word list or empty list -- that the gate
 *     selected; the joiner activation picks the matching biased encoder stream;
 *   - gate trace, go-back rewind and the withdrawn token are part of the update kernel's state machine; the host
 *     reads back once per graph replay, exactly as in wr_greedy_search.
 * The loop-invariant tensors are the caller's (the reference computes them before its loop, :327-336, with the same
 * module): bias_hidden of the hot-word list and of the empty list, and the biased encoder outputs.
 *
 * Weights in the reference module's layouts (nn.Linear [out, in], LayerNorm [dim]); LayerNorm eps = 1e-5.
 * dim = embedding_size = encoder output size = predictor output size (the module requires all three equal);
 * dim <= 512, hw_dim <= 256, n_labels <= 8, heads divides dim, lists of at most max_ctx entries.
 * ---------------------------------------------------------------------- */
typedef struct wr_hotword_weights {
    int32_t dim, heads, hw_dim, n_labels;
    /* ContextBias.predictor_bias (MultiHeadedAttention, attention.py:35-45) */
    const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b;          /* linear_q / _k / _v / _out: [dim, dim], [dim] */
    const float *bias_norm_w, *bias_norm_b;                              /* predictor_bias_bias_norm */
    const float *combine_w, *combine_b;                                  /* predictor_bias_combine [dim, 2 dim], [dim] */
    const float *out_norm_w, *out_norm_b;                                /* predictor_bias_out_norm */
    /* gate: hw_output_layer_enc -> hw_bias.linear_v -> hw_bias.linear_out -> hw_bias_norm -> hw_output_layer */
    const float *hw_enc_w, *hw_enc_b;                                    /* [hw_dim, dim], [hw_dim] */
    const float *hw_v_w, *hw_v_b, *hw_o_w, *hw_o_b;                      /* [hw_dim, hw_dim], [hw_dim] */
    const float *hw_norm_w, *hw_norm_b;                                  /* [hw_dim] */
    const float *hw_out_w, *hw_out_b;                                    /* [n_labels, hw_dim], [n_labels] */
} wr_hotword_weights;

size_t wr_hotword_workspace_bytes(const wr_decoder *h, const wr_hotword_weights *hw, int max_ctx);

/* Attach the hot-word module to a decoder handle (weights re-laid k-major once into `workspace_d`, which the caller
 * keeps alive with the handle). */
int wr_decoder_attach_hotword(wr_decoder *h, const wr_hotword_weights *hw, int max_ctx,
                              void *workspace_d, size_t workspace_bytes, void *stream);

/* enc_hot / enc_cold [N, T, dim]: ContextBias.forward_encoder_bias of the encoder output with the hot-word list /
 * the empty list (first return value); enc_feat [N, T, dim]: its second return value for the hot-word list;
 * hidden_hot [n_ctx_hot, dim], hidden_cold [n_ctx_cold, dim]: ContextBias.forward_bias_hidden of the two lists
 * (shared by all N streams).  filter_on = (context_filter_state == 'on').  Out: hyps [N, max_hyp] / hyp_lens [N] as
 * wr_greedy_search; trace [N, trace_cap] / trace_lens [N]: the gate trace (`result` in the reference, whose edit
 * distance to the hot-word labels is the second return value of the reference function). */
int wr_greedy_search_hotword(wr_decoder *h, const float *enc_hot_d, const float *enc_cold_d, const float *enc_feat_d,
                             const int32_t *enc_lens_d, const float *hidden_hot_d, int n_ctx_hot,
                             const float *hidden_cold_d, int n_ctx_cold, int N, int T, int n_steps, int blank,
                             int filter_on, int32_t *hyps_d, int32_t *hyp_lens_d, int32_t *trace_d, int trace_cap,
                             int32_t *trace_lens_d, void *stream);

/* ------------------------------------------------------------------------
 * CTC decode modes, from the ctc_lo output on; the log-softmax is fused.
 * Replaces ASRModel.ctc_greedy_search (wenet/transformer/asr_model.py:281-324) and
 * ASRModel._ctc_prefix_beam_search (:326-409; C++ twin runtime/core/decoder/ctc_prefix_beam_search.cc:107-238,
 * known-answer test runtime/core/test/ctc_prefix_beam_search_test.cc:30-73).
 * logits [B, T, V] pre-softmax, lens [B].
 * Greedy: hyps [B, T] / hyp_lens [B] / scores [B].  As in the reference, frames past an utterance's length are
 * filled with `eos` before duplicate/blank removal, and scores[b] is the maximum over all T frames of the
 * best log-probability.
 * Prefix beam: hyps [B, beam, T] (padded with -1), hyp_lens [B, beam], scores [B, beam] float64 (best first),
 * n_hyps [B]; any number of utterances at once (the reference asserts batch size 1).  beam <= 16.
 * ---------------------------------------------------------------------- */
size_t wr_ctc_decode_workspace_bytes(int B, int T, int beam);

int wr_ctc_greedy_search(const float *logits_d, const int32_t *lens_d, int B, int T, int V, int blank, int eos,
                         int32_t *hyps_d, int32_t *hyp_lens_d, float *scores_d,
                         void *workspace_d, size_t workspace_bytes, void *stream);

int wr_ctc_prefix_beam_search(const float *logits_d, const int32_t *lens_d, int B, int T, int V, int beam,
                              int blank, int32_t *hyps_d, int32_t *hyp_lens_d, double *scores_d,
                              int32_t *n_hyps_d, void *workspace_d, size_t workspace_bytes, void *stream);

/* CTC forced alignment: Viterbi over the T x (2S+1) lattice, replacing
 * forced_align, wenet/utils/ctc_util.py:27-83 (CLI wenet/bin/alignment.py:215).  logits [B, Tmax, V]: pre-softmax
 * ctc_lo output, or log-posteriors if normalized != 0 (the reference is handed ctc.log_softmax(...)).
 * alignment [B, Tmax]: the token (blank or label) aligned to each frame, -1 past input_lengths[b].
 * fp32 scores and first-candidate tie rule as the reference; Smax >= 1. */
size_t wr_ctc_align_workspace_bytes(int B, int Tmax, int Smax);

int wr_ctc_forced_align(const float *logits_d, int normalized, const int32_t *targets_d,
                        const int32_t *input_lengths_d, const int32_t *target_lengths_d,
                        int B, int Tmax, int Smax, int V, int blank, int32_t *alignment_d,
                        void *workspace_d, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WR_API_H_ */
