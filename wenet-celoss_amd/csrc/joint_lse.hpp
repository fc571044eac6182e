// Row statistics of the RNN-T loss as an epilogue of the joiner forward kernels (joint.hip, joint_split.hip).
//
// A forward workgroup owns 64 lattice cells and ALL V columns of their logits, so it can produce what pass 1 of
// the loss (rnnt_lse_kernel: one more read of the 96.6 GB logits tensor at the BASELINE shape) would compute from
// them: denom(t,u) = logsumexp_v logits, skip = logit[blank] - denom, emit = logit[y_u] - denom -- written straight
// into the RNN-T workspace layout (RnntWs: denom row-major, {skip, emit} skewed by anti-diagonal).  The loss then
// only runs its lattice sweeps (wr_rnnt_loss_fwd_from_lse).  Reference call sites fused here:
// wenet/transducer/transducer.py:132 (joint) and the first pass of :142-147 (rnnt_loss).
//
// Mechanics: in the MFMA C/D layout a lane holds, per 32-column tile, ONE column of 32 rows (2 row tiles x 16
// registers).  The kernels are matrix-core bound and their epilogue does not overlap with the MFMAs of the same
// wave, so the per-logit work is kept to four instructions: each lane keeps, per row, a partial sum
//     s = sum over the columns it has seen of 2^(x*log2e - ref),     ref = the first such x*log2e,
// with no running maximum and no rescaling (an online max would cost two exponentials per logit).  After the last
// column the (ref, s) pairs of a row (32 lanes x `waves`) are merged through LDS (the activation tile's storage,
// dead by then) as log-sum-exp of ref + log2(s).  A later logit more than 2^127 times the first one a lane saw
// overflows s to +inf: such a workgroup raises the workspace's `repair` flag instead of writing garbage, and
// wr_rnnt_loss_fwd_from_lse then runs the stand-alone pass 1 over the logits (never seen with real joiner outputs:
// it takes a spread of more than 88 nats inside one row).  The blank / label logits of a row are read back from
// the logits the workgroup has just written (L2-resident), 128 loads per workgroup.
#pragma once
#include "wr_common.hpp"

namespace wr {

struct JointLse {
    const int32_t *targets;   // [B, U1-1]
    int blank;
    int S;                    // anti-diagonals per utterance (RnntWs::S)
    float2 *lp_skew;          // [B, S, U1]   {skip, emit}
    float *denom;             // [B, T, U1]
    int32_t *repair;          // set to 1 when a partial sum overflowed (RnntWs::flag_off)
};

constexpr int kLseRows = 64;  // cells per workgroup (= kBM = kSM)

// statistics exchange: [64 rows][waves * 32 entries] floats (one log-sum-exp per lane and row), overlaid on the
// activation tile after the k-loops
inline size_t joint_lse_exchange_bytes(int waves) { return (size_t)kLseRows * waves * 32 * sizeof(float); }

// one logit of column `col` (col >= V: padding, contributes nothing); first = this is the lane's first column
__device__ __forceinline__ void joint_lse_add(float &ref, float &s, float x, bool in, bool first)
{
    const float y = x * kLog2e;
    if (first) {
        ref = in ? y : -3.0e38f;
        s = in ? 1.f : 0.f;
    } else {
        s += in ? fast_exp2(y - ref) : 0.f;           // ref = -3e38 (nothing seen yet) gives +inf: caught below
    }
}

// Merge and write.  ref/s: this lane's statistics, index rt * 16 + r <-> row 32 rt + (r&3) + 8 (r>>2) + 4 half.
// `xch` must be free for joint_lse_exchange_bytes(WAVES) bytes and every wave must have finished with whatever
// lived there (the caller synchronises before the call); `out` = the logits this workgroup has stored.
template <int WAVES>
__device__ __forceinline__ void joint_lse_finish(const JointLse &a, float *xch, const float (&ref)[32], const float (&s)[32],
                                                 const int32_t *llens, const int32_t *tlens, const float *out, long m0,
                                                 long M, int T, int U1, int V)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int E = WAVES * 32;                         // entries per row
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int row = 32 * (i >> 4) + (i & 3) + 8 * ((i & 15) >> 2) + 4 * half;
        // this lane's log-sum-exp (log2 domain); a lane that saw no valid column holds s = 0 -> -inf: no weight
        const float e = s[i] > 0.f ? ref[i] + fast_log2(s[i]) : -3.0e38f;
        bad = bad || !(s[i] < 3.0e38f);                   // +inf or NaN: overflowed
        xch[(size_t)row * E + wave * 32 + l31] = e;
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) *a.repair = 1;
    __threadfence_block();                                // this workgroup's logit stores are visible to its own loads below
    __syncthreads();
    constexpr int RPW = kLseRows / WAVES;                 // rows merged by one wave
    constexpr int EPL = (E + 63) / 64;                    // entries per lane
#pragma unroll 1
    for (int q = 0; q < RPW; ++q) {
        const int row = wave * RPW + q;
        float e[EPL];
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int idx = lane + 64 * j;
            e[j] = idx < E ? xch[(size_t)row * E + idx] : -3.0e38f;
            mx = fmaxf(mx, e[j]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) sum += fast_exp2(e[j] - mx);
        sum = wave_sum(sum);
        const long m = m0 + row;
        if (lane == 0 && m < M) {
            const long bt = m / U1;
            const int u = (int)(m - bt * U1);
            const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
            const int U = tlens[b];
            if (t < llens[b] && u <= U) {                 // same rows as rnnt_lse_kernel writes
                const float d = (mx + fast_log2(sum)) * kLn2;
                const float *orow = out + (size_t)m * V;
                const float xb = __builtin_nontemporal_load(orow + a.blank);
                float em = 0.f;
                if (u < U) em = __builtin_nontemporal_load(orow + a.targets[(size_t)b * (U1 - 1) + u]) - d;
                a.denom[m] = d;
                a.lp_skew[((size_t)b * a.S + (t + u)) * U1 + u] = make_float2(xb - d, em);
            }
        }
    }
}

}  // namespace wr
