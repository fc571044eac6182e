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

#define WR_API_VERSION 1

enum wr_dtype { WR_F32 = 0, WR_F16 = 1, WR_BF16 = 2 };

enum wr_status {
    WR_OK = 0,
    WR_EINVAL = -1,      /* bad argument (null pointer, non-positive size, blank out of range ...) */
    WR_EUNSUPPORTED = -2,/* shape outside what the kernels cover (message says which limit) */
    WR_EWORKSPACE = -3,  /* workspace too small */
    WR_ELAUNCH = -4      /* HIP reported a launch error */
};

int wr_api_version(void);
const char *wr_last_error(void);

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
 * t >= input_lengths[b]; grads_d may alias logits_d.  Limits: Smax <= 255,
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
 *   out[b,t,u,:] = ffn_out(tanh(ep[b,t,:] + pp[b,u,:]))             (joint.py:60-69)
 * w_out [V, J] and b_out [V] are ffn_out.weight / .bias in nn.Linear layout.
 * fp32 throughout (exact-fp32 MFMA).  J a multiple of 4, at most 512.
 * The activation tensor tanh(ep+pp) [B,T,U1,J] is never written to HBM in the
 * forward pass.  If both length arrays are given, 64-cell tiles that lie wholly
 * in the padded region (t >= T_b or u > U_b) are skipped and `out` is left
 * untouched there -- the RNN-T loss never reads those cells.
 *
 * wr_joint_bwd_dz: dz[b,t,u,:] = (gout[b,t,u,:] @ w_out) * (1 - tanh(ep+pp)^2),
 * zero in padded cells when lengths are given; h_d (optional, [B,T,U1,J])
 * receives tanh(ep+pp) for the weight-gradient GEMM (dW = gout^T h), which --
 * like d ep = sum_u dz and d pp = sum_t dz -- is a plain library reduction/GEMM
 * on the host side.
 * ---------------------------------------------------------------------- */
size_t wr_joint_workspace_bytes(int J, int V);

int wr_joint_fwd(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                 const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                 int B, int T, int U1, int J, int V,
                 float *out_d /* [B,T,U1,V] */,
                 void *workspace_d, size_t workspace_bytes, void *stream);

int wr_joint_bwd_dz(const float *gout_d /* [B,T,U1,V] */, const float *ep_d, const float *pp_d,
                    const float *w_out_d,
                    const int32_t *logit_lengths_d /* nullable */, const int32_t *target_lengths_d /* nullable */,
                    int B, int T, int U1, int J, int V,
                    float *dz_d /* [B,T,U1,J] */, float *h_d /* [B,T,U1,J] or NULL */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WR_API_H_ */
