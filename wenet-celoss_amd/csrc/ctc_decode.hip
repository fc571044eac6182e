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
//                             wenet/utils/common.py:268-276), insertion-ordered, stable prune.
//                             Prefixes are compared by (length, 64-bit rolling hash) and verified token by token.
#include "wr_common.hpp"

namespace wr {
namespace {

constexpr int kMaxCtcBeam = 16;
constexpr int kMaxNext = kMaxCtcBeam * (kMaxCtcBeam + 1);

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

__global__ __launch_bounds__(256) void ctc_prefix_beam_kernel(
    const float *__restrict__ tkv, const int32_t *__restrict__ tki, const int32_t *__restrict__ lens, int T, int beam,
    int blank, int32_t *__restrict__ seqs /* [B][2][beam][T] */, int32_t *__restrict__ hyps /* [B][beam][T] */,
    int32_t *__restrict__ hyp_lens, double *__restrict__ scores, int32_t *__restrict__ n_hyps)
{
    // current beam
    __shared__ int c_len[kMaxCtcBeam], c_last[kMaxCtcBeam];
    __shared__ unsigned long long c_hash[kMaxCtcBeam];
    __shared__ double c_pb[kMaxCtcBeam], c_pnb[kMaxCtcBeam];
    // next_hyps in insertion order: key = (base prefix, appended token or -1)
    __shared__ int n_base[kMaxNext], n_tok[kMaxNext], n_len[kMaxNext], order[kMaxNext];
    __shared__ unsigned long long n_hash[kMaxNext];
    __shared__ double n_pb[kMaxNext], n_pnb[kMaxNext], n_score[kMaxNext];
    __shared__ int s_ncur, s_nnext, s_sel;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = lens[b] < T ? lens[b] : T;
    const double NINF = -__builtin_huge_val();
    int32_t *sq = seqs + (size_t)b * 2 * beam * T;
    if (tid == 0) {
        s_ncur = 1; s_sel = 0;
        c_len[0] = 0; c_last[0] = -1; c_hash[0] = 1469598103934665603ULL; c_pb[0] = 0.0; c_pnb[0] = NINF;
    }
    __syncthreads();
    for (int t = 0; t < len; ++t) {
        const int32_t *cur = sq + (size_t)s_sel * beam * T;
        int32_t *nxt = sq + (size_t)(1 - s_sel) * beam * T;
        if (tid == 0) {
            int nn = 0;
            const int ncur = s_ncur;
            auto seq_elem = [&](int base, int tok, int q) -> int {      // q-th token of prefix (base [+ tok])
                return (q < c_len[base]) ? cur[(size_t)base * T + q] : tok;
            };
            auto find_or_insert = [&](int base, int tok) -> int {
                const int ln = c_len[base] + (tok >= 0 ? 1 : 0);
                const unsigned long long hs = (tok >= 0) ? c_hash[base] * 1099511628211ULL + (unsigned long long)(tok + 1)
                                                         : c_hash[base];
                for (int e = 0; e < nn; ++e) {
                    if (n_len[e] != ln || n_hash[e] != hs) continue;
                    bool same = true;
                    if (!(n_base[e] == base && n_tok[e] == tok))
                        for (int q = ln - 1; q >= 0 && same; --q) same = seq_elem(base, tok, q) == seq_elem(n_base[e], n_tok[e], q);
                    if (same) return e;
                }
                n_base[nn] = base; n_tok[nn] = tok; n_len[nn] = ln; n_hash[nn] = hs; n_pb[nn] = NINF; n_pnb[nn] = NINF;
                return nn++;
            };
            const size_t row = ((size_t)b * T + t) * beam;
            for (int k = 0; k < beam; ++k) {
                const int s = tki[row + k];
                const double ps = (double)tkv[row + k];               // logp[s].item()
                for (int i = 0; i < ncur; ++i) {
                    const double pb = c_pb[i], pnb = c_pnb[i];
                    if (s == blank) {
                        const int e = find_or_insert(i, -1);
                        n_pb[e] = py_log_add3(n_pb[e], pb + ps, pnb + ps);
                    } else if (s == c_last[i]) {
                        const int e = find_or_insert(i, -1);             // *ss -> *s
                        n_pnb[e] = py_log_add2(n_pnb[e], pnb + ps);
                        const int f = find_or_insert(i, s);              // *s-s -> *ss
                        n_pnb[f] = py_log_add2(n_pnb[f], pb + ps);
                    } else {
                        const int f = find_or_insert(i, s);
                        n_pnb[f] = py_log_add3(n_pnb[f], pb + ps, pnb + ps);
                    }
                }
            }
            // sorted(..., key=log_add([pb, pnb]), reverse=True): stable
            for (int e = 0; e < nn; ++e) { n_score[e] = py_log_add2(n_pb[e], n_pnb[e]); order[e] = e; }
            for (int i = 1; i < nn; ++i) {
                const int o = order[i];
                int p = i - 1;
                while (p >= 0 && n_score[order[p]] < n_score[o]) { order[p + 1] = order[p]; --p; }
                order[p + 1] = o;
            }
            s_nnext = nn < beam ? nn : beam;
        }
        __syncthreads();
        const int keep = s_nnext;
        for (int e = 0; e < keep; ++e) {                                  // materialise the kept prefixes
            const int f = order[e];
            const int base = n_base[f], lb = c_len[base];
            for (int q = tid; q < lb; q += 256) nxt[(size_t)e * T + q] = cur[(size_t)base * T + q];
            if (tid == 0 && n_tok[f] >= 0 && lb < T) nxt[(size_t)e * T + lb] = n_tok[f];
        }
        __syncthreads();
        if (tid == 0) {
            int t_len[kMaxCtcBeam], t_last[kMaxCtcBeam];
            unsigned long long t_hash[kMaxCtcBeam];
            double t_pb[kMaxCtcBeam], t_pnb[kMaxCtcBeam];
            for (int e = 0; e < keep; ++e) {
                const int f = order[e];
                t_len[e] = n_len[f]; t_hash[e] = n_hash[f]; t_pb[e] = n_pb[f]; t_pnb[e] = n_pnb[f];
                t_last[e] = (n_tok[f] >= 0) ? n_tok[f] : c_last[n_base[f]];
            }
            for (int e = 0; e < keep; ++e) {
                c_len[e] = t_len[e]; c_hash[e] = t_hash[e]; c_pb[e] = t_pb[e]; c_pnb[e] = t_pnb[e]; c_last[e] = t_last[e];
            }
            s_ncur = keep;
            s_sel = 1 - s_sel;
        }
        __syncthreads();
    }
    const int32_t *fin = sq + (size_t)s_sel * beam * T;
    const int n = s_ncur;
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
