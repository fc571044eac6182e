// CTC loss + gradient with the log-softmax fused in, for MI355X (gfx950).
// Replaces `ys_hat.log_softmax(2)` + `torch.nn.CTCLoss(reduction=...)` of the
// reference (wenet/transformer/ctc.py:57-63).  See include/wr_api.h.
//
// Input is the *pre-softmax* output of ctc_lo, batch-major [B, Tmax, V] (the
// reference's transpose to (T,B,V) is only nn.CTCLoss's layout requirement).
//
//   pass 1  ctc_lse_kernel    one wave per frame (b,t): streams the V logits once,
//           writes denom(b,t) and the log-probs the lattice needs --
//           lp_blank(b,t) and lp_label(b,t,i) for the S_b labels -- (S+1 values
//           instead of 2S+1: every even state of the extended sequence is blank).
//   pass 2  ctc_sweep_kernel  one workgroup per (utterance, direction), one lane per
//           state of the extended label sequence (ceil((2*Smax+1)/64) waves).
//           alpha_t(s) depends on alpha_{t-1}(s, s-1, s-2): the previous frame's
//           states sit in a double-buffered LDS array; one barrier per frame.
//           State in fp64, the bounded log-sum-exp correction in fp32 (same
//           precision argument as the RNN-T sweep).  Latency-bound: T dependent steps.
//   pass 3  ctc_grad_kernel   one workgroup per frame: softmax row into LDS,
//           subtract the state occupancies exp(alpha+beta+nll-lp) with LDS float
//           atomics (repeated labels and the S+1 blank states collide), write the
//           row out with 16-B stores.  Frames t >= T_b are zero-filled.
//
// Algorithmic traffic: 4*V per valid frame (pass 1) + 2*4*V per frame (pass 3).
#include "row_stream.hpp"
#include "wr_common.hpp"

namespace wr {
namespace {

constexpr int kCtcMaxStates = 1024;     // one lane per state of the extended label sequence: Smax <= 511

struct CtcWs {
    int KS;           // extended-label states per lane
    int SP;           // 2*Smax+1
    size_t denom_off, lpb_off, lpl_off, alpha_off, beta_off, nll_off, dump_off, total;
};

inline CtcWs ctc_ws_layout(int B, int Tmax, int Smax)
{
    CtcWs w;
    w.SP = 2 * Smax + 1;
    w.KS = (w.SP + kWave - 1) / kWave;
    size_t off = 0;
    const size_t frames = (size_t)B * Tmax;
    w.denom_off = off; off = align_up(off + frames * sizeof(float), 256);
    w.lpb_off = off;   off = align_up(off + frames * sizeof(float), 256);
    w.lpl_off = off;   off = align_up(off + frames * (size_t)(Smax > 0 ? Smax : 1) * sizeof(float), 256);
    w.alpha_off = off; off = align_up(off + frames * w.SP * sizeof(double), 256);
    w.beta_off = off;  off = align_up(off + frames * w.SP * sizeof(double), 256);
    w.nll_off = off;   off = align_up(off + (size_t)B * sizeof(double), 256);
    w.dump_off = off;  off = align_up(off + (size_t)B * 2 * kCtcMaxStates * sizeof(double), 256);
    w.total = off;
    return w;
}

// ------------------------------------------------------------------ pass 1 --
// Row log-sum-exp: wave_row_lse of row_stream.hpp (16-byte vectors, 16 in flight per lane: 153 -> 137 us against 8), with default-policy loads --
// the CTC logits (B*T*V) are re-read by the gather below and by pass 3.
__global__ __launch_bounds__(256) void ctc_lse_kernel(
    const float *__restrict__ logits, const int32_t *__restrict__ targets,
    const int32_t *__restrict__ ilens, const int32_t *__restrict__ tlens,
    int B, int Tmax, int Smax, int V, int blank,
    float *__restrict__ denom, float *__restrict__ lp_blank, float *__restrict__ lp_label, int normalized = 0)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = blockDim.x >> 6;
    const long nrows = (long)B * Tmax;
    for (long r = (long)blockIdx.x * wpb + wid; r < nrows; r += (long)gridDim.x * wpb) {
        const int b = (int)(r / Tmax);
        const int t = (int)(r - (long)b * Tmax);
        if (t >= ilens[b]) continue;
        const float *row = logits + (size_t)r * V;
        int S = tlens[b];
        S = S < 0 ? 0 : (S > Smax ? Smax : S);
        // the label logits of the first 256 labels are gathered before the row is streamed, so that the two dependent
        // loads (label, then logit) do not add their latency after it
        constexpr int GQ = 4;
        float xl[GQ];
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int i = lane + kWave * q;
            int lab = i < S ? targets[(size_t)b * Smax + i] : 0;
            lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
            xl[q] = row[lab];
        }
        const float xb = row[blank];
        const float d = normalized ? 0.f : wave_row_lse<float, false, 16>(row, V, lane);
        if (lane == 0) {
            denom[r] = d;
            lp_blank[r] = xb - d;
        }
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int i = lane + kWave * q;
            if (i < S) lp_label[(size_t)r * Smax + i] = xl[q] - d;
        }
        for (int i = lane + kWave * GQ; i < S; i += kWave) {
            int lab = targets[(size_t)b * Smax + i];
            lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
            lp_label[(size_t)r * Smax + i] = row[lab] - d;
        }
    }
}

// ------------------------------------------------------------------ pass 2 --
__device__ __forceinline__ double lse3_d(double a, double b, double c)
{
    const double m = fmax(a, fmax(b, c));
    const float ea = fast_exp2((float)(a - m) * kLog2e);
    const float eb = fast_exp2((float)(b - m) * kLog2e);
    const float ec = fast_exp2((float)(c - m) * kLog2e);
    const float r = kLn2 * fast_log2(ea + eb + ec);
    return (m == (double)kNegInf) ? (double)kNegInf : m + (double)r;
}

// One workgroup per (utterance, direction), one lane per state of the extended label sequence
// (NW = ceil((2*Smax+1)/64) waves).  alpha_t(s) depends on alpha_{t-1}(s), (s-1), (s-2) only, so the previous
// frame's states live in a double-buffered LDS array and a step is: three LDS reads, one fp64/fp32 log-sum-exp,
// one LDS write, one s_barrier.  Log-prob rows are prefetched PF frames ahead with unconditional (clamped) loads;
// idle lanes store to a sink so the loop body is straight-line.
template <int PF>
__global__ __launch_bounds__(kCtcMaxStates) void ctc_sweep_kernel(
    const float *__restrict__ lp_blank, const float *__restrict__ lp_label,
    const int32_t *__restrict__ targets, const int32_t *__restrict__ ilens,
    const int32_t *__restrict__ tlens, int Tmax, int Smax, int SP,
    double *__restrict__ alpha, double *__restrict__ beta, double *__restrict__ nll_ws,
    float *__restrict__ nll_out, double *__restrict__ dump)
{
    constexpr double NEG = (double)kNegInf;
    __shared__ double prev[2][kCtcMaxStates + 4];  // states shifted by 2; two NEG guard cells on each side
    const int b = blockIdx.x;
    const bool backward = blockIdx.y != 0;
    const int s = threadIdx.x;                     // state index
    int T = ilens[b], S = tlens[b];
    T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
    S = S < 0 ? 0 : (S > Smax ? Smax : S);
    const int NS = 2 * S + 1;                      // states of this utterance
    const float *__restrict__ lpb = lp_blank + (size_t)b * Tmax;
    const float *__restrict__ lpl = lp_label + (size_t)b * Tmax * Smax;
    double *__restrict__ out = (backward ? beta : alpha) + (size_t)b * Tmax * SP;
    double *__restrict__ sink = dump + ((size_t)b * 2 + (backward ? 1 : 0)) * kCtcMaxStates + s;

    if (T == 0) {
        // no frames: feasible only for the empty target (ATen: nll = 0 if S == 0 else inf)
        if (s == 0 && backward) {
            const double v = (S == 0) ? 0.0 : (double)__builtin_huge_valf();
            nll_ws[b] = v;
            nll_out[b] = (float)v;
        }
        return;
    }
    const bool live = s < NS;
    // per-state constants: which log-prob this state reads and whether the s-2 (fwd) / s+2 (bwd) skip is allowed
    const bool is_label = live && (s & 1);
    const int li = is_label ? (s >> 1) : 0;
    bool skip_ok = false;
    if (is_label) {
        const int me = targets[(size_t)b * Smax + li];
        if (!backward) { if (li >= 1) skip_ok = (me != targets[(size_t)b * Smax + li - 1]); }
        else { if (li + 1 < S) skip_ok = (me != targets[(size_t)b * Smax + li + 1]); }
    }
    auto load_lp = [&](int t) -> float {
        const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const float lb = lpb[tc];
        const float ll = lpl[(size_t)tc * Smax + li];
        return is_label ? ll : lb;
    };
    for (int i = threadIdx.x; i < 2 * (kCtcMaxStates + 4); i += blockDim.x) (&prev[0][0])[i] = NEG;
    __syncthreads();

    float ring[PF];
    int cur = 0;
    if (!backward) {
        // t = 0: alpha_0(0) = lp(0,blank), alpha_0(1) = lp(0,y_1)
        {
            const double v = (live && s <= 1) ? (double)load_lp(0) : NEG;
            prev[0][s + 2] = v;
            *(live ? out + s : sink) = v;
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = load_lp(1 + i);
        __syncthreads();
        for (int base = 1; base < T; base += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int t = base + i;             // frames t >= T only rewrite the guard-free LDS copy; no global effect
                const float lp = ring[i];
                ring[i] = load_lp(t + PF);
                const double a0 = prev[cur][s + 2];
                const double a1 = prev[cur][s + 1];
                const double a2 = skip_ok ? prev[cur][s] : NEG;
                double v = lse3_d(a0, a1, a2);
                v = (v == NEG) ? NEG : v + (double)lp;
                v = live ? v : NEG;
                const bool on = live && (t < T);
                if (t < T) prev[cur ^ 1][s + 2] = v;
                *(on ? out + (size_t)t * SP + s : sink) = v;
                if (t < T) cur ^= 1;
                __syncthreads();
            }
        }
    } else {
        {
            const double v = (live && s >= NS - 2) ? (double)load_lp(T - 1) : NEG;
            prev[0][s + 2] = v;
            *(live ? out + (size_t)(T - 1) * SP + s : sink) = v;
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = load_lp(T - 2 - i);
        __syncthreads();
        for (int base = 0; base < T - 1; base += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int t = T - 2 - (base + i);
                const float lp = ring[i];
                ring[i] = load_lp(t - PF);
                const double a0 = prev[cur][s + 2];
                const double a1 = prev[cur][s + 3];
                const double a2 = skip_ok ? prev[cur][s + 4] : NEG;
                double v = lse3_d(a0, a1, a2);
                v = (v == NEG) ? NEG : v + (double)lp;
                v = live ? v : NEG;
                const bool on = live && (t >= 0);
                if (t >= 0) prev[cur ^ 1][s + 2] = v;
                *(on ? out + (size_t)(t < 0 ? 0 : t) * SP + s : sink) = v;
                if (t >= 0) cur ^= 1;
                __syncthreads();
            }
        }
        // nll = -logsumexp(beta_0(0), beta_0(1))
        if (s == 0) {
            const double ll = lse3_d(prev[cur][2], NS > 1 ? prev[cur][3] : NEG, NEG);
            nll_ws[b] = -ll;
            nll_out[b] = (float)(-ll);
        }
    }
}

// ------------------------------------------------------------------ pass 3 --
// One workgroup per frame.  The state data of the frame (alpha, beta, log-probs, labels: this thread's states
// tid, tid + 256, ...) is requested first so that its latency hides behind the row stream; the row is read and written
// in 16-byte vectors through an LDS copy whose vector body is 16-byte aligned (index v + pad).
constexpr int kCtcGradThreads = 256;
constexpr int kCtcStatesPerThread = kCtcMaxStates / kCtcGradThreads;

__global__ __launch_bounds__(kCtcGradThreads) void ctc_grad_kernel(
    const float *logits, const int32_t *__restrict__ targets, const int32_t *__restrict__ ilens,
    const int32_t *__restrict__ tlens, int B, int Tmax, int Smax, int SP, int V, int blank,
    const float *__restrict__ denom, const float *__restrict__ lp_blank, const float *__restrict__ lp_label,
    const double *__restrict__ alpha, const double *__restrict__ beta, const double *__restrict__ nll_ws,
    const float *__restrict__ grad_nll, float *grads)
{
    extern __shared__ __attribute__((aligned(16))) float srow_raw[];   // V + 4 floats
    const long r = blockIdx.x;
    const int b = (int)(r / Tmax);
    const int t = (int)(r - (long)b * Tmax);
    const int T = ilens[b];
    int S = tlens[b];
    S = S < 0 ? 0 : (S > Smax ? Smax : S);
    const float *row = logits + (size_t)r * V;
    float *grow = grads + (size_t)r * V;
    const int tid = threadIdx.x;
    constexpr int nt = kCtcGradThreads;
    const RowSplit<float> sp(grow, V);            // logits and grads rows share their alignment when both bases are 16-byte
    const bool same = ((reinterpret_cast<uintptr_t>(row) ^ reinterpret_cast<uintptr_t>(grow)) & 15) == 0;
    const int h = same ? sp.h : V, nv = same ? sp.nv : 0, tail = same ? sp.tail : 0;   // else: scalar path for the whole row
    f32x4 *gbody = reinterpret_cast<f32x4 *>(grow + h);

    if (t >= T) {
        if (tid < h && h < V) grow[tid] = 0.f;
        if (h == V) for (int v = tid; v < V; v += nt) grow[v] = 0.f;
        if (tid < tail) grow[h + 4 * nv + tid] = 0.f;
        for (int i = tid; i < nv; i += nt) gbody[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;
    }
    // ---- state data in flight first
    const int NS = 2 * S + 1;
    const double *al = alpha + (size_t)r * SP;
    const double *be = beta + (size_t)r * SP;
    double a_s[kCtcStatesPerThread], b_s[kCtcStatesPerThread];
    float lp_s[kCtcStatesPerThread];
    int lab_s[kCtcStatesPerThread];
    const float lpb = lp_blank[r];
#pragma unroll
    for (int q = 0; q < kCtcStatesPerThread; ++q) {
        const int s = tid + nt * q;
        a_s[q] = 0.0; b_s[q] = 0.0; lp_s[q] = lpb; lab_s[q] = blank;
        if (s < NS) {
            a_s[q] = al[s];
            b_s[q] = be[s];
            if (s & 1) {
                int lab = targets[(size_t)b * Smax + (s >> 1)];
                lab_s[q] = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
                lp_s[q] = lp_label[(size_t)r * Smax + (s >> 1)];
            }
        }
    }
    const double nll = nll_ws[b];
    const float go = grad_nll ? grad_nll[b] : 1.f;
    const float d2 = denom[r] * kLog2e;
    // ---- softmax row (times the incoming gradient) into LDS
    const int pad = (4 - (h & 3)) & 3;
    float *srow = srow_raw + pad;                         // element v at srow[v]: the body (v = h + 4 i) is 16-byte aligned
    if (h == V) {
        for (int v = tid; v < V; v += nt) srow[v] = go * fast_exp2(fmaf(row[v], kLog2e, -d2));
    } else {
        if (tid < h) srow[tid] = go * fast_exp2(fmaf(row[tid], kLog2e, -d2));
        if (tid < tail) srow[h + 4 * nv + tid] = go * fast_exp2(fmaf(row[h + 4 * nv + tid], kLog2e, -d2));
        const f32x4 *body = reinterpret_cast<const f32x4 *>(row + h);
        f32x4 *sbody = reinterpret_cast<f32x4 *>(srow + h);
        constexpr int UN = 4;
        int i = tid;
        for (; i + (UN - 1) * nt < nv; i += UN * nt) {
            f32x4 x[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) x[q] = body[i + q * nt];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = go * fast_exp2(fmaf(x[q][c], kLog2e, -d2));
                sbody[i + q * nt] = o;
            }
        }
        for (; i < nv; i += nt) {
            const f32x4 x = body[i];
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = go * fast_exp2(fmaf(x[c], kLog2e, -d2));
            sbody[i] = o;
        }
    }
    __syncthreads();
    // ---- subtract the state occupancies exp(alpha + beta - lp + nll)   (alpha and beta both include lp)
#pragma unroll
    for (int q = 0; q < kCtcStatesPerThread; ++q) {
        const int s = tid + nt * q;
        if (s < NS) {
            const float e = (float)(a_s[q] + b_s[q] + nll - (double)lp_s[q]);
            atomicAdd(&srow[lab_s[q]], -go * fast_exp2(e * kLog2e));
        }
    }
    __syncthreads();
    if (h == V) {
        for (int v = tid; v < V; v += nt) grow[v] = srow[v];
    } else {
        if (tid < h) grow[tid] = srow[tid];
        if (tid < tail) grow[h + 4 * nv + tid] = srow[h + 4 * nv + tid];
        const f32x4 *sbody = reinterpret_cast<const f32x4 *>(srow + h);
        for (int i = tid; i < nv; i += nt) gbody[i] = sbody[i];
    }
}

// ------------------------------------------------------- forced alignment --
// Viterbi over the same T x (2S+1) lattice: wenet/utils/ctc_util.py:27-83.
// One workgroup per utterance, one lane per state, fp32 like the reference; ties prefer the first candidate
// [s, s-1, s-2] (torch.argmax).  Quirk kept: for s = 0 the reference reads log_alpha[t-1, s-1] with s-1 = -1,
// i.e. the LAST state (Python negative index) -- reproduced by wrapping.
__global__ __launch_bounds__(kCtcMaxStates) void ctc_viterbi_kernel(
    const float *__restrict__ lp_blank, const float *__restrict__ lp_label, const int32_t *__restrict__ targets,
    const int32_t *__restrict__ ilens, const int32_t *__restrict__ tlens, int Tmax, int Smax, int SP, int blank,
    int16_t *__restrict__ backptr /* [B][Tmax][SP] */, int32_t *__restrict__ align /* [B][Tmax] */)
{
    __shared__ float prev[2][kCtcMaxStates];
    __shared__ int s_state;
    const int b = blockIdx.x, s = threadIdx.x;
    int T = ilens[b], S = tlens[b];
    T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
    S = S < 1 ? 1 : (S > Smax ? Smax : S);
    const int NS = 2 * S + 1;
    const float NEG = -__builtin_huge_valf();
    const float *lpb = lp_blank + (size_t)b * Tmax;
    const float *lpl = lp_label + (size_t)b * Tmax * Smax;
    int16_t *bp = backptr + (size_t)b * Tmax * SP;
    for (int t = s; t < Tmax; t += blockDim.x) align[(size_t)b * Tmax + t] = -1;
    if (T == 0) return;
    const bool live = s < NS;
    const bool is_label = live && (s & 1);
    const int li = is_label ? (s >> 1) : 0;
    const int my_tok = is_label ? targets[(size_t)b * Smax + li] : blank;
    // two-candidate rule of the reference: blank, s < 2, or same label as s-2
    bool two_only = true;
    if (is_label && s >= 2) two_only = (my_tok == blank) || (my_tok == targets[(size_t)b * Smax + li - 1]);
    auto lp_at = [&](int t) -> float { return is_label ? lpl[(size_t)t * Smax + li] : lpb[t]; };
    prev[0][s] = (live && s <= 1) ? lp_at(0) : NEG;
    __syncthreads();
    int cur = 0;
    for (int t = 1; t < T; ++t) {
        if (live) {
            const int s1 = (s >= 1) ? s - 1 : NS - 1;              // s-1 = -1 wraps to the last state
            float best = prev[cur][s];
            int arg = s;
            const float c1 = prev[cur][s1];
            if (c1 > best) { best = c1; arg = s1; }
            if (!two_only) {
                const float c2 = prev[cur][s - 2];
                if (c2 > best) { best = c2; arg = s - 2; }
            }
            prev[cur ^ 1][s] = best + lp_at(t);
            bp[(size_t)t * SP + s] = (int16_t)arg;
        }
        cur ^= 1;
        __syncthreads();
    }
    if (s == 0) {
        const float a = prev[cur][NS - 1], c = prev[cur][NS - 2];
        int st = (c > a) ? NS - 2 : NS - 1;
        for (int t = T - 1; t >= 0; --t) {
            const int tok = (st & 1) ? targets[(size_t)b * Smax + (st >> 1)] : blank;
            align[(size_t)b * Tmax + t] = tok;
            if (t > 0) st = bp[(size_t)t * SP + st];
        }
        s_state = st;
    }
}

int ctc_check(int B, int Tmax, int Smax, int V, int blank)
{
    WR_REQUIRE(B > 0 && Tmax > 0 && Smax >= 0 && V > 0, WR_EINVAL,
               "ctc: B, Tmax, V must be positive and Smax >= 0 (got %d,%d,%d,%d)", B, Tmax, Smax, V);
    WR_REQUIRE(blank >= 0 && blank < V, WR_EINVAL, "ctc: blank %d out of range [0,%d)", blank, V);
    WR_REQUIRE(2 * Smax + 1 <= kCtcMaxStates, WR_EUNSUPPORTED,
               "ctc: Smax=%d exceeds the sweep kernel's limit of %d labels", Smax, (kCtcMaxStates - 1) / 2);
    WR_REQUIRE((size_t)V * sizeof(float) <= 64 * 1024, WR_EUNSUPPORTED,
               "ctc: V=%d does not fit the gradient kernel's LDS row (max 16384)", V);
    return WR_OK;
}

void launch_ctc_sweep(const CtcWs &w, char *ws, const int32_t *targets, const int32_t *ilens, const int32_t *tlens,
                      int B, int Tmax, int Smax, float *nll, hipStream_t st)
{
    hipLaunchKernelGGL((ctc_sweep_kernel<8>), dim3(B, 2), dim3(64 * w.KS), 0, st,
                       reinterpret_cast<const float *>(ws + w.lpb_off), reinterpret_cast<const float *>(ws + w.lpl_off),
                       targets, ilens, tlens, Tmax, Smax > 0 ? Smax : 1, w.SP,
                       reinterpret_cast<double *>(ws + w.alpha_off), reinterpret_cast<double *>(ws + w.beta_off),
                       reinterpret_cast<double *>(ws + w.nll_off), nll, reinterpret_cast<double *>(ws + w.dump_off));
}

}  // namespace
}  // namespace wr

using namespace wr;

extern "C" size_t wr_ctc_workspace_bytes(int B, int Tmax, int Smax)
{
    if (B <= 0 || Tmax <= 0 || Smax < 0) return 0;
    return ctc_ws_layout(B, Tmax, Smax).total;
}

extern "C" int wr_ctc_loss_fwd(const void *logits_d, int dtype, const int32_t *targets_d,
                               const int32_t *input_lengths_d, const int32_t *target_lengths_d, int B, int Tmax,
                               int Smax, int V, int blank, float *nll_d, void *workspace_d, size_t workspace_bytes,
                               void *stream)
{
    if (int rc = ctc_check(B, Tmax, Smax, V, blank)) return rc;
    WR_REQUIRE(logits_d && input_lengths_d && target_lengths_d && nll_d && workspace_d, WR_EINVAL,
               "ctc_loss_fwd: null pointer argument");
    WR_REQUIRE(targets_d || Smax == 0, WR_EINVAL, "ctc_loss_fwd: targets is null");
    WR_REQUIRE(dtype == WR_F32, WR_EUNSUPPORTED, "ctc_loss_fwd: dtype %d not supported (fp32 only)", dtype);
    const CtcWs w = ctc_ws_layout(B, Tmax, Smax);
    WR_REQUIRE(workspace_bytes >= w.total, WR_EWORKSPACE, "ctc_loss_fwd: workspace %zu < required %zu",
               workspace_bytes, w.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_d);
    const long nrows = (long)B * Tmax;
    const int SmaxA = Smax > 0 ? Smax : 1;
    long blocks = (nrows + 3) / 4;
    // one frame per wave, workgroups dispatched in frame order (as the RNN-T row pass; 153 us against 158 us with 2 048
    // persistent workgroups at the BASELINE shape)
    hipLaunchKernelGGL(ctc_lse_kernel, dim3((int)blocks), dim3(256), 0, st, static_cast<const float *>(logits_d),
                       targets_d, input_lengths_d, target_lengths_d, B, Tmax, SmaxA, V, blank,
                       reinterpret_cast<float *>(ws + w.denom_off), reinterpret_cast<float *>(ws + w.lpb_off),
                       reinterpret_cast<float *>(ws + w.lpl_off));
    WR_CHECK_LAUNCH("ctc_lse_kernel");
    launch_ctc_sweep(w, ws, targets_d, input_lengths_d, target_lengths_d, B, Tmax, Smax, nll_d, st);
    WR_CHECK_LAUNCH("ctc_sweep_kernel");
    return WR_OK;
}

extern "C" int wr_ctc_loss_bwd(const void *logits_d, int dtype, const int32_t *targets_d,
                               const int32_t *input_lengths_d, const int32_t *target_lengths_d, int B, int Tmax,
                               int Smax, int V, int blank, const float *grad_nll_d, void *grads_d,
                               const void *workspace_d, size_t workspace_bytes, void *stream)
{
    if (int rc = ctc_check(B, Tmax, Smax, V, blank)) return rc;
    WR_REQUIRE(logits_d && input_lengths_d && target_lengths_d && grads_d && workspace_d, WR_EINVAL,
               "ctc_loss_bwd: null pointer argument");
    WR_REQUIRE(targets_d || Smax == 0, WR_EINVAL, "ctc_loss_bwd: targets is null");
    WR_REQUIRE(dtype == WR_F32, WR_EUNSUPPORTED, "ctc_loss_bwd: dtype %d not supported (fp32 only)", dtype);
    const CtcWs w = ctc_ws_layout(B, Tmax, Smax);
    WR_REQUIRE(workspace_bytes >= w.total, WR_EWORKSPACE, "ctc_loss_bwd: workspace %zu < required %zu",
               workspace_bytes, w.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *ws = static_cast<const char *>(workspace_d);
    const int SmaxA = Smax > 0 ? Smax : 1;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((size_t)(V + 4) * sizeof(float)));      // V = 16384: 16 bytes above the 64 KB default
    hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)((long)B * Tmax)), dim3(kCtcGradThreads), (size_t)(V + 4) * sizeof(float), st,
                       static_cast<const float *>(logits_d), targets_d, input_lengths_d, target_lengths_d, B, Tmax,
                       SmaxA, w.SP, V, blank, reinterpret_cast<const float *>(ws + w.denom_off),
                       reinterpret_cast<const float *>(ws + w.lpb_off), reinterpret_cast<const float *>(ws + w.lpl_off),
                       reinterpret_cast<const double *>(ws + w.alpha_off),
                       reinterpret_cast<const double *>(ws + w.beta_off),
                       reinterpret_cast<const double *>(ws + w.nll_off), grad_nll_d, static_cast<float *>(grads_d));
    WR_CHECK_LAUNCH("ctc_grad_kernel");
    return WR_OK;
}

extern "C" size_t wr_ctc_align_workspace_bytes(int B, int Tmax, int Smax)
{
    if (B <= 0 || Tmax <= 0 || Smax <= 0) return 0;
    const CtcWs w = ctc_ws_layout(B, Tmax, Smax);
    return w.total + align_up((size_t)B * Tmax * w.SP * sizeof(int16_t), 256);
}

extern "C" int wr_ctc_forced_align(const float *logits_d, int normalized, const int32_t *targets_d,
                                   const int32_t *input_lengths_d, const int32_t *target_lengths_d, int B, int Tmax,
                                   int Smax, int V, int blank, int32_t *alignment_d, void *workspace_d,
                                   size_t workspace_bytes, void *stream)
{
    if (int rc = ctc_check(B, Tmax, Smax, V, blank)) return rc;
    WR_REQUIRE(Smax >= 1, WR_EINVAL, "ctc_forced_align: empty label sequences cannot be aligned");
    WR_REQUIRE(logits_d && targets_d && input_lengths_d && target_lengths_d && alignment_d && workspace_d, WR_EINVAL,
               "ctc_forced_align: null pointer argument");
    const CtcWs w = ctc_ws_layout(B, Tmax, Smax);
    const size_t need = w.total + align_up((size_t)B * Tmax * w.SP * sizeof(int16_t), 256);
    WR_REQUIRE(workspace_bytes >= need, WR_EWORKSPACE, "ctc_forced_align: workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_d);
    const long nrows = (long)B * Tmax;
    long blocks = (nrows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ctc_lse_kernel, dim3((int)blocks), dim3(256), 0, st, logits_d, targets_d, input_lengths_d,
                       target_lengths_d, B, Tmax, Smax, V, blank, reinterpret_cast<float *>(ws + w.denom_off),
                       reinterpret_cast<float *>(ws + w.lpb_off), reinterpret_cast<float *>(ws + w.lpl_off), normalized);
    WR_CHECK_LAUNCH("ctc_lse_kernel");
    hipLaunchKernelGGL(ctc_viterbi_kernel, dim3(B), dim3(64 * w.KS), 0, st, reinterpret_cast<const float *>(ws + w.lpb_off),
                       reinterpret_cast<const float *>(ws + w.lpl_off), targets_d, input_lengths_d, target_lengths_d, Tmax,
                       Smax, w.SP, blank, reinterpret_cast<int16_t *>(ws + w.total), alignment_d);
    WR_CHECK_LAUNCH("ctc_viterbi_kernel");
    return WR_OK;
}
