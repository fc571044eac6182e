// CTC decode modes on MI355X (gfx950): greedy search and prefix beam search from the ctc_lo output.
// (SURVEY.md section 8f item 1 -- the callers either side of the hot path.)
//
// Replaces, from the encoder output's CTC projection on,
//   ASRModel.ctc_greedy_search         wenet/transformer/asr_model.py:281-324
//   ASRModel._ctc_prefix_beam_search   wenet/transformer/asr_model.py:326-409
// (the reference's C++ twin, runtime/core/decoder/ctc_prefix_beam_search.cc:107-238, carries the
// only known-answer test on this side of the path: runtime/core/test/ctc_prefix_beam_search_test.cc:30-73).
//
// Input is the PRE-softmax ctc_lo output [B,T,V]; the log-softmax is fused (ctc.py:66-75).
//   ctc_frame_top1_kernel     one wave per frame: row log-sum-exp, argmax (first index on ties), its log-prob
//   ctc_greedy_collapse_kernel  one workgroup per utterance: eos-fill of padded frames, duplicate/blank removal
//   ctc_frame_topk_kernel     one workgroup per frame: log-softmax row in LDS, top-`beam` (value desc, index asc)
//   ctc_prefix_beam_kernel    one workgroup per utterance, all T frames inside one launch: the prefix
//                             dictionary (blank / non-blank ending scores in float64, log_add exactly as
//                             wenet/utils/common.py:268-276), insertion-ordered, stable prune -- one contribution
//                             per thread, all-pairs steps in parallel (see the kernel).
//                             Prefixes are compared by (length, 64-bit rolling hash) and verified token by token.
#include "wr_common.hpp"

namespace wr {
namespace {

constexpr int kMaxCtcBeam = 16;

struct CtcDecWs {
    size_t best_off, top_off, tkv_off, tki_off, seq_off, total;
};

inline CtcDecWs ctc_dec_layout(int B, int T, int beam)
{
    CtcDecWs w;
    size_t off = 0;
    const size_t rows = (size_t)B * T;
    w.best_off = off; off = align_up(off + rows * sizeof(int32_t), 256);
    w.top_off = off;  off = align_up(off + rows * sizeof(float), 256);
    w.tkv_off = off;  off = align_up(off + rows * beam * sizeof(float), 256);
    w.tki_off = off;  off = align_up(off + rows * beam * sizeof(int32_t), 256);
    w.seq_off = off;  off = align_up(off + (size_t)B * 2 * beam * T * sizeof(int32_t), 256);
    w.total = off;
    return w;
}

__global__ __launch_bounds__(256) void ctc_frame_top1_kernel(const float *__restrict__ logits, int rows, int V,
                                                             int32_t *__restrict__ best, float *__restrict__ top)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = blockDim.x >> 6;
    for (long r = (long)blockIdx.x * wpb + wid; r < rows; r += (long)gridDim.x * wpb) {
        const float *row = logits + (size_t)r * V;
        float m = -3.0e38f;
        int mi = 0x7fffffff;
        for (int v = lane; v < V; v += kWave) {
            const float x = row[v];
            if (x > m) { m = x; mi = v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, kWave);
            const int oi = __shfl_xor(mi, o, kWave);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        float s = 0.f;
        for (int v = lane; v < V; v += kWave) s += expf(row[v] - m);
        s = wave_sum(s);
        if (lane == 0) {
            best[r] = mi;
            top[r] = (m - m) - logf(s);              // log_softmax of the maximum: (x - max) - log(sum)
        }
    }
}

__global__ __launch_bounds__(256) void ctc_greedy_collapse_kernel(const int32_t *__restrict__ best,
                                                                  const float *__restrict__ top,
                                                                  const int32_t *__restrict__ lens, int T, int blank, int eos,
                                                                  int32_t *__restrict__ hyps, int32_t *__restrict__ hyp_lens,
                                                                  float *__restrict__ scores)
{
    __shared__ float sv[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = lens[b] < T ? lens[b] : T;
    float mx = -3.0e38f;
    for (int t = tid; t < T; t += 256) mx = fmaxf(mx, top[(size_t)b * T + t]);   // over ALL frames, as the reference
    mx = block_max(mx, sv);
    if (tid == 0) {
        scores[b] = mx;
        int n = 0, prev = -1;
        for (int t = 0; t < T; ++t) {
            const int tok = (t < len) ? best[(size_t)b * T + t] : eos;           // masked_fill_(mask, eos), :319
            if (tok != prev && tok != blank) hyps[(size_t)b * T + n++] = tok;    // remove_duplicates_and_blank
            prev = tok;
        }
        hyp_lens[b] = n;
    }
}

__global__ __launch_bounds__(256) void ctc_frame_topk_kernel(const float *__restrict__ logits, int V, int beam,
                                                             float *__restrict__ tkv, int32_t *__restrict__ tki)
{
    extern __shared__ float lp[];
    __shared__ float sv[4];
    __shared__ int si[4];
    const long r = blockIdx.x;
    const int tid = threadIdx.x;
    const float *x = logits + (size_t)r * V;
    float m = -3.0e38f;
    for (int v = tid; v < V; v += 256) m = fmaxf(m, x[v]);
    m = block_max(m, sv);
    float sum = 0.f;
    for (int v = tid; v < V; v += 256) sum += expf(x[v] - m);
    sum = block_sum(sum, sv);
    const float ls = logf(sum);
    for (int v = tid; v < V; v += 256) lp[v] = (x[v] - m) - ls;
    __syncthreads();
    for (int k = 0; k < beam; ++k) {
        float bestv = -__builtin_huge_valf();
        int bi = 0x7fffffff;
        for (int v = tid; v < V; v += 256) {
            const float val = lp[v];
            if (val > bestv) { bestv = val; bi = v; }
        }
        block_argmax(bestv, bi, sv, si);
        if (tid == 0) {
            tkv[(size_t)r * beam + k] = bestv;
            tki[(size_t)r * beam + k] = (bi < V) ? bi : 0;
            if (bi < V) lp[bi] = -__builtin_huge_valf();
        }
        __syncthreads();
    }
}

// wenet/utils/common.py:268-276 in float64
__device__ __forceinline__ double py_log_add2(double a, double b)
{
    const double ninf = -__builtin_huge_val();
    if (a == ninf && b == ninf) return ninf;
    const double mx = a > b ? a : b;
    return mx + log(exp(a - mx) + exp(b - mx));
}
__device__ __forceinline__ double py_log_add3(double a, double b, double c)
{
    const double ninf = -__builtin_huge_val();
    if (a == ninf && b == ninf && c == ninf) return ninf;
    double mx = a > b ? a : b;
    mx = mx > c ? mx : c;
    return mx + log((exp(a - mx) + exp(b - mx)) + exp(c - mx));    // sum() adds left to right
}

// One workgroup per utterance, all T frames inside one launch.  Per frame the reference visits the (top-k symbol,
// current prefix) pairs in order and adds each pair's contribution(s) to a dictionary keyed by the new prefix.  Here
// every contribution is a SLOT (slot = 2 * (k * ncur + i) + j in the reference's visiting order; j = 1 only for the
// "same symbol again" case, which feeds two keys), handled in parallel:
//   A  a thread fills its slots: key (base prefix, appended token or -1), its length and rolling hash, the operation;
//   B  all pairs of slots: equal keys (length + hash, verified token by token unless trivially equal) -> every slot
//      learns the first slot with its key, its representative; representatives in slot order = the dictionary's
//      insertion order;
//   C  a representative applies its slots' operations in slot order (float64 log_add exactly as
//      wenet/utils/common.py:268-276: the n-ary form sums left to right);
//   D  score = log_add(pb, pnb); stable descending rank among representatives by counting over all pairs;
//   E  the kept prefixes are copied into the other sequence buffer ((prefix, position) pairs in flight together).
constexpr int kMaxSlots = 2 * kMaxCtcBeam * kMaxCtcBeam;
enum CtcOp { kOpNone = 0, kOpPb3 = 1, kOpPnb2 = 2, kOpPnb3 = 3 };

__global__ __launch_bounds__(256) void ctc_prefix_beam_kernel(
    const float *__restrict__ tkv, const int32_t *__restrict__ tki, const int32_t *__restrict__ lens, int T, int beam,
    int blank, int32_t *__restrict__ seqs /* [B][2][beam][T] */, int32_t *__restrict__ hyps /* [B][beam][T] */,
    int32_t *__restrict__ hyp_lens, double *__restrict__ scores, int32_t *__restrict__ n_hyps)
{
    // current beam
    __shared__ int c_len[kMaxCtcBeam], c_last[kMaxCtcBeam];
    __shared__ unsigned long long c_hash[kMaxCtcBeam];
    __shared__ double c_pb[kMaxCtcBeam], c_pnb[kMaxCtcBeam];
    // slots of this frame
    __shared__ int s_base[kMaxSlots], s_tok[kMaxSlots], s_len[kMaxSlots], s_op[kMaxSlots], s_rep[kMaxSlots], s_rank[kMaxSlots];
    __shared__ unsigned long long s_hash[kMaxSlots];
    __shared__ double s_a[kMaxSlots], s_b[kMaxSlots], r_pb[kMaxSlots], r_pnb[kMaxSlots], r_score[kMaxSlots];
    __shared__ int order[kMaxCtcBeam];
    __shared__ int s_nrep, s_wtot[4];
    // next beam (staged so that the current one stays readable while it is built)
    __shared__ int t_len[kMaxCtcBeam], t_last[kMaxCtcBeam], t_base[kMaxCtcBeam], t_tok[kMaxCtcBeam];
    __shared__ unsigned long long t_hash[kMaxCtcBeam];
    __shared__ double t_pb[kMaxCtcBeam], t_pnb[kMaxCtcBeam];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = lens[b] < T ? lens[b] : T;
    const double NINF = -__builtin_huge_val();
    int32_t *sq = seqs + (size_t)b * 2 * beam * T;
    int ncur = 1, sel = 0;
    if (tid == 0) { c_len[0] = 0; c_last[0] = -1; c_hash[0] = 1469598103934665603ULL; c_pb[0] = 0.0; c_pnb[0] = NINF; }
    __syncthreads();
    for (int t = 0; t < len; ++t) {
        const int32_t *__restrict__ cur = sq + (size_t)sel * beam * T;
        int32_t *__restrict__ nxt = sq + (size_t)(1 - sel) * beam * T;
        const size_t row = ((size_t)b * T + t) * beam;
        // ---- A: slots, written densely in the reference's visiting order (one (symbol, prefix) pair per thread) ----
        const int C = beam * ncur;                                    // <= 256
        int tok0 = -1, op0 = kOpNone, tok1 = -1, op1 = kOpNone, pi = 0;
        double a0 = NINF, b0 = NINF, a1 = NINF;
        if (tid < C) {
            const int k = tid / ncur;
            pi = tid - k * ncur;
            const int sym = tki[row + k];
            const double ps = (double)tkv[row + k];                   // logp[s].item()
            const double pb = c_pb[pi], pnb = c_pnb[pi];
            if (sym == blank) { op0 = kOpPb3; a0 = pb + ps; b0 = pnb + ps; }
            else if (sym == c_last[pi]) {
                op0 = kOpPnb2; a0 = pnb + ps;                          // *ss -> *s
                tok1 = sym; op1 = kOpPnb2; a1 = pb + ps;               // *s-s -> *ss
            } else { tok0 = sym; op0 = kOpPnb3; a0 = pb + ps; b0 = pnb + ps; }
        }
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned long long extra_mask = __ballot(op1 != kOpNone);
        if (lane == 0) s_wtot[wave] = __popcll(extra_mask);
        if (tid == 0) s_nrep = 0;
        __syncthreads();
        int dpos = tid + __popcll(extra_mask & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) dpos += s_wtot[w];
        const int nslot = C + s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
        if (tid < C) {
            s_base[dpos] = pi; s_tok[dpos] = tok0; s_op[dpos] = op0; s_a[dpos] = a0; s_b[dpos] = b0;
            s_len[dpos] = c_len[pi] + (tok0 >= 0 ? 1 : 0);
            s_hash[dpos] = tok0 >= 0 ? c_hash[pi] * 1099511628211ULL + (unsigned long long)(tok0 + 1) : c_hash[pi];
            s_rep[dpos] = dpos; s_rank[dpos] = 0;
            if (op1 != kOpNone) {
                const int d1 = dpos + 1;
                s_base[d1] = pi; s_tok[d1] = tok1; s_op[d1] = op1; s_a[d1] = a1; s_b[d1] = NINF;
                s_len[d1] = c_len[pi] + 1;
                s_hash[d1] = c_hash[pi] * 1099511628211ULL + (unsigned long long)(tok1 + 1);
                s_rep[d1] = d1; s_rank[d1] = 0;
            }
        }
        __syncthreads();
        // ---- B: representative = first slot with the same key (four threads per slot i, f < i) ----
        for (int i = tid >> 2; i < nslot; i += 64) {
            const int li_ = s_len[i];
            const unsigned long long hi_ = s_hash[i];
            for (int f = tid & 3; f < i; f += 4) {
                if (s_len[f] != li_ || s_hash[f] != hi_) continue;
                bool same = (s_base[f] == s_base[i] && s_tok[f] == s_tok[i]);
                if (!same) {                                           // different spelling of (maybe) the same prefix
                    const int bi = s_base[i], bf = s_base[f], ti = s_tok[i], tf = s_tok[f];
                    const int li = c_len[bi], lf = c_len[bf];
                    same = true;
                    for (int q = li_ - 1; q >= 0 && same; --q) {
                        const int x = q < li ? cur[(size_t)bi * T + q] : ti;
                        const int y = q < lf ? cur[(size_t)bf * T + q] : tf;
                        same = (x == y);
                    }
                }
                if (same) atomicMin(&s_rep[i], f);
            }
        }
        __syncthreads();
        // ---- C: a representative applies its slots' operations in slot order ----
        for (int i = tid; i < nslot; i += 256) {
            if (s_rep[i] != i) continue;
            double pb = NINF, pnb = NINF;
            for (int f = i; f < nslot; ++f) {
                if (s_rep[f] != i) continue;
                if (s_op[f] == kOpPb3) pb = py_log_add3(pb, s_a[f], s_b[f]);
                else if (s_op[f] == kOpPnb2) pnb = py_log_add2(pnb, s_a[f]);
                else pnb = py_log_add3(pnb, s_a[f], s_b[f]);
            }
            r_pb[i] = pb; r_pnb[i] = pnb;
            r_score[i] = py_log_add2(pb, pnb);
            atomicAdd(&s_nrep, 1);
        }
        __syncthreads();
        // ---- D: stable descending rank among representatives (sorted(..., reverse=True) keeps insertion order on ties) ----
        for (int i = tid >> 2; i < nslot; i += 64) {
            if (s_rep[i] != i) continue;
            const double me = r_score[i];
            int cnt = 0;
            for (int f = tid & 3; f < nslot; f += 4) {
                if (s_rep[f] != f) continue;
                const double ot = r_score[f];
                cnt += (ot > me || (ot == me && f < i)) ? 1 : 0;
            }
            if (cnt) atomicAdd(&s_rank[i], cnt);
        }
        __syncthreads();
        const int keep = s_nrep < beam ? s_nrep : beam;
        for (int i = tid; i < nslot; i += 256)
            if (s_rep[i] == i && s_rank[i] < beam) order[s_rank[i]] = i;
        __syncthreads();
        // ---- E: the next beam ----
        if (tid < keep) {
            const int f = order[tid];
            t_base[tid] = s_base[f]; t_tok[tid] = s_tok[f];
            t_len[tid] = s_len[f]; t_hash[tid] = s_hash[f]; t_pb[tid] = r_pb[f]; t_pnb[tid] = r_pnb[f];
            t_last[tid] = s_tok[f] >= 0 ? s_tok[f] : c_last[s_base[f]];
        }
        __syncthreads();
        {
            int maxlb = 0;
            for (int e = 0; e < keep; ++e) maxlb = c_len[t_base[e]] > maxlb ? c_len[t_base[e]] : maxlb;
            constexpr int UN = 4;
            for (int i0 = 0; i0 < keep * maxlb; i0 += 256 * UN) {
                int val[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int i = i0 + u * 256 + tid;
                    const int e = i / maxlb, q = i - e * maxlb;
                    const bool in = e < keep && q < c_len[t_base[e < keep ? e : 0]];
                    val[u] = in ? cur[(size_t)t_base[e] * T + q] : 0;
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int i = i0 + u * 256 + tid;
                    const int e = i / maxlb, q = i - e * maxlb;
                    if (e < keep && q < c_len[t_base[e]]) nxt[(size_t)e * T + q] = val[u];
                }
            }
            if (tid < keep) {
                const int lb = c_len[t_base[tid]];
                if (t_tok[tid] >= 0 && lb < T) nxt[(size_t)tid * T + lb] = t_tok[tid];
            }
        }
        __syncthreads();
        if (tid < keep) {
            c_len[tid] = t_len[tid]; c_hash[tid] = t_hash[tid]; c_pb[tid] = t_pb[tid]; c_pnb[tid] = t_pnb[tid];
            c_last[tid] = t_last[tid];
        }
        ncur = keep;
        sel = 1 - sel;
        __syncthreads();
    }
    const int32_t *fin = sq + (size_t)sel * beam * T;
    const int n = ncur;
    for (int i = tid; i < beam * T; i += 256) {
        const int e = i / T, q = i % T;
        hyps[((size_t)b * beam + e) * T + q] = (e < n && q < c_len[e]) ? fin[(size_t)e * T + q] : -1;
    }
    if (tid < beam) {
        hyp_lens[(size_t)b * beam + tid] = tid < n ? c_len[tid] : 0;
        scores[(size_t)b * beam + tid] = tid < n ? py_log_add2(c_pb[tid], c_pnb[tid]) : NINF;
    }
    if (tid == 0) n_hyps[b] = n;
}

int ctc_dec_check(int B, int T, int V, int blank)
{
    WR_REQUIRE(B > 0 && T > 0 && V > 1, WR_EINVAL, "ctc decode: B, T must be positive and V > 1 (got %d,%d,%d)", B, T, V);
    WR_REQUIRE(blank >= 0 && blank < V, WR_EINVAL, "ctc decode: blank %d out of range", blank);
    WR_REQUIRE((size_t)V * sizeof(float) <= 64 * 1024 - 256, WR_EUNSUPPORTED, "ctc decode: V=%d exceeds 16320", V);
    return WR_OK;
}

}  // namespace
}  // namespace wr

using namespace wr;

extern "C" size_t wr_ctc_decode_workspace_bytes(int B, int T, int beam)
{
    if (B <= 0 || T <= 0 || beam <= 0) return 0;
    return ctc_dec_layout(B, T, beam).total;
}

extern "C" int wr_ctc_greedy_search(const float *logits_d, const int32_t *lens_d, int B, int T, int V, int blank, int eos,
                                    int32_t *hyps_d, int32_t *hyp_lens_d, float *scores_d, void *workspace_d,
                                    size_t workspace_bytes, void *stream)
{
    if (int rc = ctc_dec_check(B, T, V, blank)) return rc;
    WR_REQUIRE(logits_d && lens_d && hyps_d && hyp_lens_d && scores_d && workspace_d, WR_EINVAL,
               "ctc_greedy_search: null pointer argument");
    const CtcDecWs w = ctc_dec_layout(B, T, 1);
    WR_REQUIRE(workspace_bytes >= w.total, WR_EWORKSPACE, "ctc_greedy_search: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_d);
    const int rows = B * T;
    int blocks = (rows + 3) / 4;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(ctc_frame_top1_kernel, dim3(blocks), dim3(256), 0, st, logits_d, rows, V,
                       reinterpret_cast<int32_t *>(ws + w.best_off), reinterpret_cast<float *>(ws + w.top_off));
    WR_CHECK_LAUNCH("ctc_frame_top1_kernel");
    hipLaunchKernelGGL(ctc_greedy_collapse_kernel, dim3(B), dim3(256), 0, st, reinterpret_cast<const int32_t *>(ws + w.best_off),
                       reinterpret_cast<const float *>(ws + w.top_off), lens_d, T, blank, eos, hyps_d, hyp_lens_d, scores_d);
    WR_CHECK_LAUNCH("ctc_greedy_collapse_kernel");
    return WR_OK;
}

extern "C" int wr_ctc_prefix_beam_search(const float *logits_d, const int32_t *lens_d, int B, int T, int V, int beam,
                                         int blank, int32_t *hyps_d, int32_t *hyp_lens_d, double *scores_d,
                                         int32_t *n_hyps_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    if (int rc = ctc_dec_check(B, T, V, blank)) return rc;
    WR_REQUIRE(logits_d && lens_d && hyps_d && hyp_lens_d && scores_d && n_hyps_d && workspace_d, WR_EINVAL,
               "ctc_prefix_beam_search: null pointer argument");
    WR_REQUIRE(beam >= 1 && beam <= kMaxCtcBeam && beam <= V, WR_EUNSUPPORTED,
               "ctc_prefix_beam_search: beam=%d (1..%d, at most V)", beam, kMaxCtcBeam);
    const CtcDecWs w = ctc_dec_layout(B, T, beam);
    WR_REQUIRE(workspace_bytes >= w.total, WR_EWORKSPACE, "ctc_prefix_beam_search: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_d);
    hipLaunchKernelGGL(ctc_frame_topk_kernel, dim3(B * T), dim3(256), (size_t)V * sizeof(float), st, logits_d, V, beam,
                       reinterpret_cast<float *>(ws + w.tkv_off), reinterpret_cast<int32_t *>(ws + w.tki_off));
    WR_CHECK_LAUNCH("ctc_frame_topk_kernel");
    hipLaunchKernelGGL(ctc_prefix_beam_kernel, dim3(B), dim3(256), 0, st, reinterpret_cast<const float *>(ws + w.tkv_off),
                       reinterpret_cast<const int32_t *>(ws + w.tki_off), lens_d, T, beam, blank,
                       reinterpret_cast<int32_t *>(ws + w.seq_off), hyps_d, hyp_lens_d, scores_d, n_hyps_d);
    WR_CHECK_LAUNCH("ctc_prefix_beam_kernel");
    return WR_OK;
}
