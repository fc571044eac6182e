// Transducer decoding on MI355X (gfx950): batched greedy search and batched
// prefix beam search with every per-step operation on the device.
//
// Replaces (semantics: SURVEY.md App. A.3 / A.4)
//   greedy   wenet/transducer/search/greedy_search copy.py:6-63  (the upstream core loop that
//            Transducer.greedy_search dispatches to, transducer.py:515-598)
//   beam     wenet/transducer/search/prefix_beam_search.py:42-148 (PrefixBeamSearch)
//   step API wenet/transducer/predictor.py:160-200 (RNNPredictor.forward_step, LSTM cell x L +
//            projection) and wenet/transducer/joint.py:45-70 for step shapes.
//
// The reference runs ONE utterance at a time from Python with a host<->device
// sync on every step (.item()).  Here `lanes` (independent streams for greedy;
// utterances x beam hypotheses for beam search) advance together with all control
// state in device memory.  Every activation that feeds a contraction is kept
// K-MAJOR ([feature][lane], lanes padded to 32) so that the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32) fragments of both operands are plain coalesced 128-byte
// reads from L2 -- no LDS staging, no transposes between stages:
//
//   lane_gemm_kernel        C^T = A^T-segments x k-major weights (+bias), 32 output columns per
//                           workgroup, K split over the 4 waves, register ping-pong on the L2
//                           fragment loads, LDS only for the final 4-way reduction.  Five launches
//                           per micro-step, the elementwise stages fused as epilogues:
//                             LSTM layer l   gates = x W_ih^T + h W_hh^T -> cell update (c, h),
//                                            predicated per lane ("predictor steps only after a
//                                            non-blank"); gate columns permuted so one tile holds
//                                            i,f,g,o of 8 hidden units
//                             projection     k-major store
//                             pred_ffn       -> joiner activation tanh(enc_ffn(enc)[utt,t] + .)
//                             ffn_out        row-major logits
//                           (the embedding of a lane's next token is written by the kernel that
//                           decides the token)
//   greedy_update_kernel    log-softmax + argmax (first index on ties) + the
//                           frame/emission state machine + cache commit
//   beam_topk_kernel /      log-softmax, CTC mixture, top-k from registers; per-utterance
//   beam_update_kernel      expansion, prefix fusion (float64 log_add), stable prune --
//                           one candidate per thread
//
// Micro-steps are replayed from a hipGraph (captured once per decoder handle) so
// the host only checks a "lanes still active" word every 16 steps.
#include "wr_common.hpp"

#include <math.h>
#include <string.h>

#include <new>
#include <vector>

namespace wr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- diagnostic build only (-DWR_STAMPS, tools/decode_stamps.py): per-wave s_memtime / s_memrealtime stamps of the
// micro-step kernels into a device array, to attribute a kernel's microseconds to load latency / MFMA / reduction /
// epilogue and to measure start skew and inter-kernel gaps.  In the product library none of this exists.
#ifdef WR_STAMPS
constexpr int kStampSlots = 12, kStampWgs = 1024, kStampWaves = 8, kStampPts = 10;
__device__ unsigned long long g_stamps[kStampSlots * kStampWgs * kStampWaves * kStampPts];
#define WR_STAMP_DECL unsigned long long st_[::wr::kStampPts] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define WR_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); st_[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WR_STAMP_RT(i) do { __builtin_amdgcn_sched_barrier(0); st_[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WR_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WR_STAMP_FLUSH(slot)                                                                                              \
    do {                                                                                                                  \
        const int wg_ = blockIdx.y * gridDim.x + blockIdx.x, wv_ = threadIdx.x >> 6;                                      \
        if ((threadIdx.x & 63) == 0 && (slot) >= 0 && (slot) < ::wr::kStampSlots && wg_ < ::wr::kStampWgs && wv_ < ::wr::kStampWaves) \
            for (int i_ = 0; i_ < ::wr::kStampPts; ++i_)                                                                  \
                ::wr::g_stamps[((size_t)((slot) * ::wr::kStampWgs + wg_) * ::wr::kStampWaves + wv_) * ::wr::kStampPts + i_] = st_[i_]; \
    } while (0)
#else
#define WR_STAMP_DECL
#define WR_STAMP(i)
#define WR_STAMP_RT(i)
#define WR_STAMP_DRAIN()
#define WR_STAMP_FLUSH(slot)
#endif

constexpr int kMaxLayers = WR_MAX_LSTM_LAYERS;
enum PredictorType { kPredLstm = 0, kPredEmbedding = 1, kPredConv = 2 };
constexpr int kMaxLanes = 1024;      // streams (greedy) or utterances x beam decoded together
constexpr int kMaxBeam = 16;
constexpr int kMaxLook = 4;          // greedy look-ahead: encoder frames evaluated per micro-step against one predictor state
#ifndef WR_STEPS_PER_GRAPH
#define WR_STEPS_PER_GRAPH 16
#endif
constexpr int kStepsPerGraph = WR_STEPS_PER_GRAPH;
constexpr int kStageSlots = 4;

struct Dims {
    int V, E, P, D, H, L, J;      // vocab, encoder dim, predictor out dim, embed dim, hidden, layers, join dim
    int Ve;                       // rows of the embedding table (= V unless the caller says otherwise)
    int Dp, Hp, Pp, Jp;           // K dimensions padded to a multiple of 8 (zero rows)
    int G4p, Vp;                  // N dimensions: 4H padded to 32, V padded to 256
    int NL, NLp;                  // lanes, lanes padded to 32
    int act;                      // joiner activation (wr_activation)
    int ptype, ctx, heads, pact;  // predictor: 0 LSTM / 1 embedding / 2 conv; context size, heads, activation (types 1, 2)
    float ln_eps;
};

// Device-resident decoder state; all pointers are carved from the caller's workspace.
// "T" suffix = k-major ([feature][NLp]).
struct DevState {
    Dims d;
    // k-major (transposed, zero padded) weights
    const float *embed;
    float *wt_ih[kMaxLayers], *wt_hh[kMaxLayers], *bsum[kMaxLayers];
    float *proj_wt, *predffn_wt, *encffn_wt, *out_wt;
    const float *proj_b, *predffn_b, *encffn_b, *out_b;
    // stateless predictors (predictor.py:203-481): the LSTM slots above stay empty; the "LSTM state" cache_hT[l] holds
    // the embedding of history slot l (l = 0 oldest), L = ctx - 1, H = D
    const float *pos_w, *pffn_b, *pnorm_w, *pnorm_b, *conv_w, *conv_b;
    float *pffn_wt;               // [Dp][up(D,32)] k-major ffn weight (type 1)
    // LSTM predictor: projection followed by pred_ffn is ONE linear map (nothing non-linear sits between predictor.py:198
    // and joint.py:57): pred_ffn(projection(h)) = (Wf Wp) h + (Wf bp + bf), composed once per handle
    float *projffn_wt, *projffn_b;   // [Hp][up(J,32)], [up(J,32)]
    float *combT, *ffnT;          // [Dp][NLp] head-weighted context sum / ffn output (type 1)
    // per-call inputs
    const float *enc;             // [n_utt, T, E]
    const int32_t *enc_lens;      // [n_utt]
    const float *ctc_logp;        // [n_utt, T, V] (beam only)
    int n_utt, T, lanes_per_utt, n_lanes;
    // lane state
    float *ep_all;                // [n_utt, T, J]
    int32_t *token, *lane_t, *noblk, *need_pred, *lane_active;
    int32_t *new_is_cache;        // streaming quirk: the pending predictor state equals the committed one
    float *xT;                    // [Dp][NLp]     embedding of each lane's token
    float *cache_hT, *cache_cT;   // [L][Hp][NLp]  committed LSTM state
    float *new_hT, *new_cT;       // [L][Hp][NLp]  output of the last predictor step
    // LSTM predictor in its hoisted form (see "LSTM predictor step" below): state lives in a pool of 2 * NLp slots
    float *etab;                  // [V][G4p]          W_ih(layer 0) . embed[v], gate columns permuted like the weights
    float *pool_c;                // [L][2 NLp][Hp]    cell state of a slot
    float *pool_g;                // [L][2 NLp][G4p]   W_hh . h + b_ih + b_hh of the slot's hidden state
    int32_t *comm_slot, *new_slot;   // [NLp] slot of a lane's committed state / slot its next predictor step writes
    float *outT;                  // [Pp][NLp]     projected predictor output
    float *ht;                    // [Jp][kMaxLook * NLp]  joiner activation, column f * NLp + n
    float *logits;                // [kMaxLook * NLp, V]   row f * NLp + n: lane n, frame t_n + f
    int32_t *row_tok;             // [kMaxLook * NLp]      argmax of each logits row (look-ahead greedy)
    float4 *row_part;             // [NLp][Vp / 32]        per-block row statistics written by the ffn_out epilogue
    int n_cb;                     // Vp / 32
    int32_t *active_count;        // lanes still decoding
    // greedy outputs / params
    int32_t *hyps;                // [NL, max_hyp]
    int32_t *hyp_lens;
    int max_hyp, n_steps, blank;
    // beam state
    int beam;
    float ctc_weight, tr_weight;
    float *topv;                  // [NL, beam]
    int32_t *topi;
    int32_t *bhyps;               // [2, n_utt, beam, Lmax]
    int32_t *bhyp_lens;           // [2, n_utt, beam]
    double *bscores;              // [n_utt, beam]
    unsigned long long *bhash;    // [n_utt, beam]   running hash of each hypothesis' tokens
    int32_t *n_hyps;              // [n_utt]
    int32_t *frame;               // [n_utt]
    int32_t *hyp_sel;             // [n_utt]
    int Lmax;
    // hot-word greedy (wr_greedy_search_hotword): the state machine of greedy_search.py:297-430 on the device
    int hw_on, hw_filter;         // mode flag; context_filter_state == 'on'
    int hw_nctx[2];               // entries of the empty list (0) and of the hot-word list (1)
    const int32_t *gate_tab;      // [n_utt, T]   top-1 of the hot-word gate per frame (hw_gate_table_kernel)
    int32_t *cur_gate;            // [NLp] result[-1]: 1 = hot-word biasing, 0 = empty-list biasing
    int32_t *gb_flag, *gb_end, *last_t;   // [NLp] go_back_flag, go_back_end, last_t
    int32_t *trace, *trace_len;   // [NL, trace_cap] / [NL]: `result`, the gate trace
    int trace_cap;
    float *biasT;                 // [Pp][NLp] biased predictor output (input of pred_ffn in hot-word mode)
};

struct GemmArgs {
    const float *A0, *B0;         // segment 0: A^T [K0][lda], B [K0][ldb]   (K multiples of 8)
    const float *A1, *B1;         // optional segment 1
    int K0, K1, lda, ldb;
    const float *bias;            // [N] or null
    float *C;
    int ldc, N, n_lanes;
    // epilogues that need lane state (LSTM cell, joiner activation): workspace pointers by value, so that the
    // kernel reads them from its argument segment instead of chasing them through the device-side state
    const DevState *st;           // per-call scalars (T, lanes_per_utt)
    const int32_t *lane_active, *need_pred, *lane_t;
    float *new_hT;                // this layer's new hidden state [Hp][NLp] (lane-indexed: the next GEMM's A operand)
    const float *ep_all;          // [n_utt, T, J]
    int H, J;
    int act;                      // joiner activation (wr_activation)
    int look, lane_stride;        // joiner activation: frames per lane and the column stride between frames (NLp)
    float4 *row_part;             // kEpiRowStats: per (row, 32-column block) {max, sum exp(x - max), runner-up, first index of max}
    int row_part_ld;              // blocks per row
    const int32_t *ep_gate;       // hot-word mode: per-lane selector of the encoder stream (ep_all + gate * ep_gate_stride)
    size_t ep_gate_stride;
    // LSTM layers >= 1 (kEpiLstmCell) and the recurrent products (kEpiSlotRow): state pool of this layer, row-major per slot
    float *pool_g;                // [2 NLp][G4p]  W_hh . h + b of the slot's hidden state
    float *pool_c;                // [2 NLp][Hp]   cell state
    const int32_t *comm_slot, *new_slot;   // [NLp]
    const int32_t *slot_idx;      // kEpiSlotRow: the slot each lane's row goes to
    int Hp, G4p;
#ifdef WR_STAMPS
    int dbg_slot;
#endif
};

// Handle-constant pointers of the per-lane state, passed to the deciding kernels BY VALUE: the kernel-argument segment
// is in the scalar registers within 0.06 us of the wave's start, so the lane's flags, slots and row records are requested
// at once instead of behind the load of the state block (one memory round trip, ~0.5 us, off the critical path of every
// micro-step).  Everything that changes from call to call (buffers of the caller, T, n_steps ...) stays in DevState.
struct LaneArgs {
    int32_t *lane_active, *lane_t, *noblk, *need_pred, *new_is_cache, *comm_slot, *new_slot;
    const float4 *row_part;
    int n_cb, V;
};

enum GemmEpilogue { kEpiKMajor = 0, kEpiRowMajor = 1, kEpiLstmCell = 2, kEpiJointAct = 3, kEpiRowStats = 4, kEpiSlotRow = 5 };

// ----------------------------------------------------------------- setup --
// src [R][C] row-major -> dst [C][Rp] (zero padded columns R..Rp-1 and rows C..Cp-1)
__global__ void transpose_pad_kernel(const float *__restrict__ src, int R, int C, int Rp, int Cp, float *__restrict__ dst)
{
    __shared__ float tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cp && r < Rp) dst[(size_t)c * Rp + r] = tile[tx][i];
    }
}

__global__ void add_bias_kernel(const float *a, const float *b, int n, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// ep_all[row, :] = enc[row, :] @ enc_ffn^T + b     (rows = n_utt * T)
__global__ __launch_bounds__(256) void ep_all_kernel(DevState *s, const float *__restrict__ enc, float *__restrict__ ep_out)
{
    extern __shared__ float xs[];              // [8][E]
    const Dims &d = s->d;
    const long rows = (long)s->n_utt * s->T;
    const long r0 = (long)blockIdx.x * 8;
    for (int i = threadIdx.x; i < 8 * d.E; i += blockDim.x) {
        const long r = r0 + i / d.E;
        xs[i] = (r < rows) ? enc[(size_t)r * d.E + i % d.E] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < d.J; j += blockDim.x) {
        float acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = 0.f;
        for (int k = 0; k < d.E; ++k) {
            const float w = s->encffn_wt[(size_t)k * d.J + j];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = fmaf(w, xs[r * d.E + k], acc[r]);
        }
        const float b = s->encffn_b[j];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r0 + r < rows) ep_out[(size_t)(r0 + r) * d.J + j] = acc[r] + b;
    }
}

// --------------------------------------------------------- predictor step --
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// LSTM weights: w [4H][K] (nn.LSTM layout, gate-major rows i,f,g,o) -> k-major [Kp][4*Hp] with the gate
// columns permuted so that every 32-column tile holds i,f,g,o of 8 consecutive hidden units
// (column c' -> unit (c'/32)*8 + c'%8, gate (c'%32)/8); padded rows/columns are zero.
__global__ void lstm_weight_prep_kernel(const float *__restrict__ w, int H, int Hp, int K, int Kp, float *__restrict__ dst)
{
    const int G4p = 4 * Hp;
    const long total = (long)Kp * G4p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i / G4p), c = (int)(i % G4p);
        const int unit = (c >> 5) * 8 + (c & 7), gate = (c & 31) >> 3;
        dst[i] = (k < K && unit < H) ? w[(size_t)(gate * H + unit) * K + k] : 0.f;
    }
}
__global__ void lstm_bias_prep_kernel(const float *__restrict__ b_ih, const float *__restrict__ b_hh, int H, int Hp,
                                      float *__restrict__ dst)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 4 * Hp) return;
    const int unit = (c >> 5) * 8 + (c & 7), gate = (c & 31) >> 3;
    dst[c] = unit < H ? b_ih[gate * H + unit] + b_hh[gate * H + unit] : 0.f;
}

// C = sum over segments of A^T B (+ bias).  grid = (column tiles of 32, lane tiles of 32*MT); 512 threads = 8
// waves, each takes an eighth of every segment's K.  Both operands come straight from L2 / Infinity Cache
// (k-major, coalesced 128-byte fragments) through a register ping-pong of CH k-steps; with the shipped sizes
// (K <= 512 per segment) the whole K slice of a wave is in flight before its first MFMA, so a launch pays one
// memory latency.  Loads are unconditional (clamped rows, masked by a 0/1 factor) so the waits stay counted.
// Epilogues: k-major store, row-major store (logits), LSTM cell (gate columns are permuted at create time so a
// 32-column tile holds i,f,g,o of 8 hidden units), joiner activation tanh(enc_ffn(enc)[t] + pred_ffn(pred)).
// VW ("virtual waves" per wave): with VW = 2 a workgroup has 4 waves (256 threads) and every wave works through two of
// the eight K slices one after the other, each into accumulators of its own -- the same eight partial tiles, summed in the
// same order, so the results are BIT-IDENTICAL to the 8-wave form, but four workgroups fit a CU instead of two (the
// 640-workgroup joiner GEMM of a 128-lane beam search then runs in one round instead of two).
constexpr int kGemmWaves = 8;
template <int MT, int EPI, int VW = 1>
__device__ __forceinline__ void lane_gemm_body(const GemmArgs &g, const int bx, float (&red)[kGemmWaves][MT][32 * 32])
{
    WR_STAMP_DECL;
    WR_STAMP_RT(7);
    WR_STAMP(0);
    const int n0 = bx * 32;
    const int lane0 = blockIdx.y * (32 * MT);                          // first decode lane of this workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave_phys = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int CH = (VW == 2 || MT >= 4) ? 8 : 16;                   // k-pairs per chunk (NBUF chunks in flight)
    constexpr int NT = 64 * kGemmWaves / VW;
#ifdef WR_STAMPS
    asm volatile("" ::"s"(g.K0), "s"(g.ldb));      // the kernel-argument segment has arrived
    WR_STAMP(9);
#endif
    const int kq0 = g.K0 / kGemmWaves, kq1 = (g.A1 != nullptr) ? g.K1 / kGemmWaves : 0;
    const int nch0 = (kq0 + 2 * CH - 1) / (2 * CH), nch1 = (kq1 + 2 * CH - 1) / (2 * CH);
    const int nch = nch0 + nch1;
    int wave = wave_phys * VW;                                          // the K slice being worked on
    f32x16 acc[MT];

    auto load_chunk = [&](int ci, float (&bv)[CH], float (&av)[MT][CH], int slice_ahead = 0) {
        const bool s1 = ci >= nch0;
        const int cj = s1 ? ci - nch0 : ci;
        const int kq = s1 ? kq1 : kq0;
        const float *__restrict__ As = s1 ? g.A1 : g.A0;
        const float *__restrict__ Bs = s1 ? g.B1 : g.B0;
        const int kb = (wave + slice_ahead) * kq;
        const float *__restrict__ A = As + (size_t)(kb + half) * g.lda + lane0 + l31;
        const float *__restrict__ Bm = Bs + (size_t)(kb + half) * g.ldb + n0 + l31;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k = cj * 2 * CH + 2 * i;
            const int kc = k < kq ? k : kq - 2;                         // K >= 16: kq - 2 is a valid row pair
            bv[i] = Bm[(size_t)kc * g.ldb];                             // nothing consumes the value here: the load stays
#pragma unroll                                                          // in flight until its MFMA (round 2 multiplied by a
            for (int m = 0; m < MT; ++m) av[m][i] = A[(size_t)kc * g.lda + m * 32];   // 0/1 mask here and so WAITED here)
        }
    };
    // chunk `ci`: its k-pairs beyond the slice's end (only when the slice is not a multiple of the chunk) are masked
    // at use; the test is wave-uniform and false for the shipped sizes
    auto mfma_chunk = [&](int ci, const float (&bv)[CH], const float (&av)[MT][CH]) {
        const bool s1 = ci >= nch0;
        const int valid = (s1 ? kq1 : kq0) - (s1 ? ci - nch0 : ci) * 2 * CH;   // k rows of this chunk inside the slice
        if (valid >= 2 * CH) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][i], bv[i], acc[m], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const float b = 2 * i < valid ? bv[i] : 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][i], b, acc[m], 0, 0, 0);
            }
        }
    };
    // Loads in three groups.  (1) The few per-lane indices that later loads depend on (state slots, flags, the lane's
    // frame) go out FIRST, so that they come back with the operands; (2) the first two operand chunks; (3) behind them
    // the epilogue operands (bias, pool rows, enc_ffn rows), whose addresses need (1): by then (1) has long arrived, and
    // (3) is consumed only after the K loop.  (Round 2 issued (1) after (2) and paid a second memory round trip.)
    constexpr int kItems = MT * 1024 / NT;                                // outputs per thread of the 32 x 32*MT tile
    float bias_v[kItems > 4 ? kItems : 4];
    bool cell_on = false, cell_in = false;
    int cell_j = 0, cell_n = 0, cell_unit = 0, cell_sn = 0, cell_sc = 0, cell_la = 0, cell_np = 0;
    float cell_c = 0.f, cell_g[4];
    int slot_v[kItems];
    bool act_on[kItems], act_in[kItems];
    int act_la[kItems], act_tt[kItems], act_gate[kItems];
    float act_ep[kMaxLook][kItems];
    int act_T = 1, act_lpu = 1;
    if (EPI == kEpiLstmCell) {
        const int q = tid & 255;
        cell_j = q >> 5;
        cell_n = lane0 + ((tid >> 8) % MT) * 32 + l31;
        cell_unit = (n0 >> 5) * 8 + cell_j;
        cell_in = tid < MT * 256 && cell_n < g.n_lanes && cell_unit < g.H;
        const int nn = cell_in ? cell_n : 0;
        cell_la = g.lane_active[nn]; cell_np = g.need_pred[nn];
        cell_sc = g.comm_slot[nn];
        cell_sn = g.new_slot[nn];
    } else if (EPI == kEpiSlotRow) {
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int i = tid + it * NT;
            const int n = lane0 + (i >> 10) * 32 + ((i & 1023) >> 5);
            slot_v[it] = g.slot_idx[n < g.n_lanes ? n : 0];
        }
    } else if (EPI == kEpiJointAct) {
        act_T = g.st->T; act_lpu = g.st->lanes_per_utt;
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int i = tid + it * NT;
            const int v = n0 + ((i & 1023) >> 5), n = lane0 + (i >> 10) * 32 + (i & 31);
            act_in[it] = n < g.n_lanes && v < g.N;
            const int nn = act_in[it] ? n : 0;
            act_la[it] = g.lane_active[nn];
            act_tt[it] = g.lane_t[nn];
            act_gate[it] = g.ep_gate ? g.ep_gate[nn] : 0;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // The wave works through its VW slices one after the other as ONE sequence of chunks with NBUF chunk buffers in
    // flight (a ring: the MFMAs of chunk c are followed by the request for chunk c + NBUF, across slice boundaries too);
    // at the end of a slice the accumulators go to LDS and start again from zero.  What bounds these kernels is how
    // many loads a CU has in flight (~2 us of memory latency): two buffers for the 8- and 4-wave forms, four where a
    // wave is alone on its SIMD (VW >= 4).
    constexpr int NBUF = VW >= 4 ? 4 : 2;
    float bb[NBUF][CH], ba[NBUF][MT][CH];
    const int n_chunks = VW * nch;
    int ld_cc = 0, ld_slice = 0;                                        // next chunk to request: (slice, chunk in slice)
    auto request = [&](float (&bv)[CH], float (&av)[MT][CH]) {
        wave = wave_phys * VW + ld_slice;
        load_chunk(ld_cc, bv, av);
        const bool last = ld_slice == VW - 1 && ld_cc == nch - 1;       // past the end: the last chunk again (counted waits)
        if (!last) { if (++ld_cc == nch) { ld_cc = 0; ++ld_slice; } }
    };
#pragma unroll
    for (int j = 0; j < NBUF; ++j) request(bb[j], ba[j]);
    __builtin_amdgcn_sched_barrier(0);

    if (EPI == kEpiRowMajor || EPI == kEpiRowStats || EPI == kEpiSlotRow) {
        const int v = n0 + l31;                                          // the column of every item of this thread
        bias_v[0] = g.bias ? g.bias[v < g.N ? v : g.N - 1] : 0.f;
    } else if (EPI == kEpiKMajor || EPI == kEpiJointAct) {
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int v = n0 + (((tid + it * NT) & 1023) >> 5);
            bias_v[it] = g.bias ? g.bias[v < g.N ? v : g.N - 1] : 0.f;
        }
    }
    if (EPI == kEpiLstmCell) {
        // gates = (W_ih . h_below: this GEMM) + (W_hh . h + b: pool_g of the lane's committed slot); one (lane, unit) per thread
        const float *__restrict__ gp = g.pool_g + (size_t)cell_sc * g.G4p + n0 + cell_j;
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) cell_g[gt] = gp[gt * 8];
        cell_c = g.pool_c[(size_t)cell_sc * g.Hp + (cell_in ? cell_unit : 0)];
        cell_on = cell_in && cell_la && cell_np;
    } else if (EPI == kEpiJointAct) {
#pragma unroll
        for (int it = 0; it < kItems; ++it) act_on[it] = act_in[it] && act_la[it];
#pragma unroll
        for (int f = 0; f < kMaxLook; ++f)
#pragma unroll
            for (int it = 0; it < kItems; ++it) {
                const int i = tid + it * NT;
                const int v = n0 + ((i & 1023) >> 5), n = lane0 + (i >> 10) * 32 + (i & 31);
                const int t = act_tt[it] + f < act_T ? act_tt[it] + f : act_T - 1;
                const size_t sel = (size_t)(act_on[it] ? act_gate[it] : 0) * g.ep_gate_stride;
                act_ep[f][it] = (act_on[it] && f < g.look) ? g.ep_all[sel + ((size_t)(n / act_lpu) * act_T + t) * g.J + v] : 0.f;
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    WR_STAMP(1);                                   // every load of the launch issued
#if defined(WR_STAMPS) && WR_STAMPS >= 2
    WR_STAMP_DRAIN();                              // level 2: wait for them, so that point 2 is the pure load latency
    WR_STAMP(2);
#endif

#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x16){0};
    int mm_cc = 0, mm_slice = 0;
    for (int c0 = 0; c0 < n_chunks; c0 += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            if (c0 + j < n_chunks) {
                mfma_chunk(mm_cc, bb[j], ba[j]);
                __builtin_amdgcn_sched_barrier(0);
#if defined(WR_STAMPS) && WR_STAMPS < 2
                if (c0 + j == 0) WR_STAMP(2);      // first chunk's operands arrived and its MFMAs issued
#endif
                if (mm_cc == nch - 1) {            // slice complete: its partial tile to LDS
                    const int wslice = wave_phys * VW + mm_slice;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;       // C/D layout of the 32x32 MFMA
                            red[wslice][m][row * 32 + (l31 ^ row)] = acc[m][r];    // XOR swizzle: both epilogue orders are conflict-free
                        }
                        acc[m] = (f32x16){0};
                    }
                    mm_cc = 0; ++mm_slice;
                } else {
                    ++mm_cc;
                }
                if (c0 + j + NBUF < n_chunks) {
                    request(bb[j], ba[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    WR_STAMP(3);                                   // MFMAs retired (accumulators read), partial tiles on their way to LDS
    __syncthreads();
    WR_STAMP(4);
    auto total = [&](int m, int ln, int col) {
        const int rc = ln * 32 + (col ^ ln);
        return ((red[0][m][rc] + red[1][m][rc]) + (red[2][m][rc] + red[3][m][rc])) +
               ((red[4][m][rc] + red[5][m][rc]) + (red[6][m][rc] + red[7][m][rc]));
    };
    if (EPI == kEpiRowMajor) {
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int i = tid + it * NT;
            const int m = i >> 10, ln = (i & 1023) >> 5, col = i & 31;
            const int n = lane0 + m * 32 + ln, v = n0 + col;
            if (n < g.n_lanes && v < g.N) g.C[(size_t)n * g.ldc + v] = total(m, ln, col) + bias_v[0];
        }
    } else if (EPI == kEpiRowStats) {
        // the logits leave as above, and with them the statistics of this workgroup's 32 columns of every row: the
        // update kernel then resolves log-softmax and argmax of a row from V / 32 records instead of V logits.  The 32
        // columns of a row sit in the 32 lanes of a half-wave: butterflies inside the half.
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int i = tid + it * NT;
            const int m = i >> 10, ln = (i & 1023) >> 5, col = i & 31;
            const int n = lane0 + m * 32 + ln, v = n0 + col;
            const bool in = v < g.N;
            const float x = total(m, ln, col) + bias_v[0];
            if (n < g.n_lanes && in) g.C[(size_t)n * g.ldc + v] = x;
            const float xs = in ? x : -3.0e38f;
            auto fmx = [](float a, float b) { return fmaxf(a, b); };
            const float mx = half_allreduce_f(xs, fmx);
            // first column holding the maximum
            const int first = half_allreduce_i((xs == mx) ? col : 32, [](int a, int b) { return a < b ? a : b; });
            const float second = half_allreduce_f((col == first) ? -3.0e38f : xs, fmx);     // the largest of the others
            // the sum: every lane ends with the same value only if the additions pair up identically in both halves of
            // each butterfly -- they do (a + b == b + a), so the record does not depend on the lane that stores it
            const float se = half_allreduce_f(in ? expf(x - mx) : 0.f, [](float a, float b) { return a + b; });
            if (col == 0 && n < g.n_lanes)
                g.row_part[(size_t)n * g.row_part_ld + bx] = make_float4(mx, se, second, __builtin_bit_cast(float, n0 + first));
        }
    } else if (EPI == kEpiKMajor) {
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int i = tid + it * NT;
            const int m = i >> 10, col = (i & 1023) >> 5, ln = i & 31;   // consecutive threads -> consecutive lanes
            const int v = n0 + col, n = lane0 + m * 32 + ln;
            if (v < g.N) g.C[(size_t)v * g.ldc + n] = total(m, ln, col) + bias_v[it];
        }
    } else if (EPI == kEpiLstmCell) {
        // tile columns: [i x8 | f x8 | g x8 | o x8] of hidden units u0 .. u0+7; only lanes whose predictor steps are
        // written.  One (lane, unit) per thread.
        if (cell_on) {
            const int j = cell_j, ln = l31;
            const int m = (tid >> 8) % MT;
            const float ig = sigmoidf_(total(m, ln, j) + cell_g[0]);
            const float fg = sigmoidf_(total(m, ln, 8 + j) + cell_g[1]);
            const float gg = tanhf(total(m, ln, 16 + j) + cell_g[2]);
            const float og = sigmoidf_(total(m, ln, 24 + j) + cell_g[3]);
            const float c = fg * cell_c + ig * gg;
            g.pool_c[(size_t)cell_sn * g.Hp + cell_unit] = c;
            g.new_hT[(size_t)cell_unit * g.lda + cell_n] = og * tanhf(c);       // lda = NLp
        }
    } else if (EPI == kEpiSlotRow) {
        // row-major rows addressed through a per-lane slot: pool_g[slot(n)][v] = W_hh[v] . h_n + b[v]
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int i = tid + it * NT;
            const int m = i >> 10, ln = (i & 1023) >> 5, col = i & 31;
            const int n = lane0 + m * 32 + ln, v = n0 + col;
            if (n < g.n_lanes && v < g.N) g.C[(size_t)slot_v[it] * g.ldc + v] = total(m, ln, col) + bias_v[0];
        }
    } else {   // kEpiJointAct: ht[j][f * stride + lane] = tanh(ep_all[utt, t_lane + f, j] + pp[j][lane]); zero for idle lanes
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int i = tid + it * NT;
            const int m = i >> 10, col = (i & 1023) >> 5, ln = i & 31;
            const int v = n0 + col, n = lane0 + m * 32 + ln;
            if (v >= g.N) continue;
            const float pp = total(m, ln, col) + bias_v[it];
#pragma unroll
            for (int f = 0; f < kMaxLook; ++f)
                if (f < g.look) g.C[(size_t)v * g.ldc + f * g.lane_stride + n] = act_on[it] ? act_value(g.act, act_ep[f][it] + pp) : 0.f;
        }
    }
    WR_STAMP(5);                                   // epilogue stores issued
    WR_STAMP_DRAIN();
    WR_STAMP(6);
    WR_STAMP_RT(8);
    WR_STAMP_FLUSH(g.dbg_slot);
}

template <int MT, int EPI>
__global__ __launch_bounds__(64 * kGemmWaves) void lane_gemm_kernel(GemmArgs g)
{
    __shared__ float red[kGemmWaves][MT][32 * 32];
    lane_gemm_body<MT, EPI>(g, blockIdx.x, red);
}

// the 1-wave form (VW = 8): ONE wave works through all eight K slices of a 32 x 32 tile.  For the 640 tiles of a 128-lane
// beam search's joiner GEMM this is the fastest shape: 640 independent waves on the chip's 1024 SIMDs, no barrier, each
// MFMA-bound for K / 2 * 64 cycles (7 us at K = 512) with its loads hidden behind them, where the 8-wave form needs two
// rounds of workgroups (20 us).  Same eight partial tiles summed in the same order: bit-identical.
template <int EPI>
__global__ __launch_bounds__(64) void lane_gemm1_kernel(GemmArgs g)
{
    __shared__ float red[kGemmWaves][1][32 * 32];
    lane_gemm_body<1, EPI, 8>(g, blockIdx.x, red);
}

// the 4-wave form (VW = 2, 32-lane tiles): same results, four workgroups per CU
template <int EPI>
__global__ __launch_bounds__(64 * kGemmWaves / 2, 4) void lane_gemm4_kernel(GemmArgs g)
{
    __shared__ float red[kGemmWaves][1][32 * 32];
    lane_gemm_body<1, EPI, 2>(g, blockIdx.x, red);
}

// Two independent GEMMs over the same lanes in ONE launch (column tiles [0, ct0) belong to the first): the small
// predictor GEMMs leave most of the chip idle, so the recurrent product of the layer below rides along for free.
template <int MT, int EPI0, int EPI1>
__global__ __launch_bounds__(64 * kGemmWaves) void lane_gemm_pair_kernel(GemmArgs g0, GemmArgs g1, int ct0)
{
    __shared__ float red[kGemmWaves][MT][32 * 32];
    if ((int)blockIdx.x < ct0) lane_gemm_body<MT, EPI0>(g0, blockIdx.x, red);
    else lane_gemm_body<MT, EPI1>(g1, (int)blockIdx.x - ct0, red);
}

// ---- LSTM predictor step, hoisted form -------------------------------------------------------------------------
// gates_l = W_ih_l . in_l + W_hh_l . h_l + b_l.  Two of the three terms do not have to wait for the step:
//   * layer 0's input is an embedding row, so W_ih_0 . embed[v] is a table (etab, built once per handle);
//   * W_hh_l . h_l + b_l depends only on the state the step starts from, and that state was itself produced by an
//     earlier step: it is computed right behind that step (kEpiSlotRow jobs riding in the launches of the next stage,
//     which leave most of the chip idle) and kept with the state (pool_g).
// Layer 0 is then elementwise and runs in the tail of the kernel that decides the token (update / resolve / beam
// update / init): one launch per micro-step less, and the layer >= 1 GEMMs have half the depth.  State lives in a pool
// of 2 * NLp slots addressed through comm_slot / new_slot: committing a step swaps two integers (greedy) and the beam
// search's survivors inherit slot numbers instead of copied caches.  Lane-indexed are only the hidden outputs of the
// last step (new_hT), which the next GEMM reads as its k-major A operand.
__device__ __forceinline__ void lstm_layer0_cell(const DevState &S, int n, int tok, int sc, int sn)
{
    const Dims &d = S.d;
    const float *__restrict__ e = S.etab + (size_t)tok * d.G4p;
    const float *__restrict__ g = S.pool_g + (size_t)sc * d.G4p;
    const float *__restrict__ c0 = S.pool_c + (size_t)sc * d.Hp;
    float *__restrict__ c1 = S.pool_c + (size_t)sn * d.Hp;
    for (int u = threadIdx.x; u < d.H; u += blockDim.x) {
        const int col = (u >> 3) * 32 + (u & 7);              // tile of 32 columns = i,f,g,o of 8 units
        const float ig = sigmoidf_(e[col] + g[col]);
        const float fg = sigmoidf_(e[col + 8] + g[col + 8]);
        const float gg = tanhf(e[col + 16] + g[col + 16]);
        const float og = sigmoidf_(e[col + 24] + g[col + 24]);
        const float c = fg * c0[u] + ig * gg;
        c1[u] = c;
        S.new_hT[(size_t)u * d.NLp + n] = og * tanhf(c);
    }
}

// A lane's two slots hold the zero state (h = 0, so W_hh . h + b = b), comm_slot = n, new_slot = NLp + n.
__device__ __forceinline__ void lstm_reset_lane(const DevState &S, int n)
{
    const Dims &d = S.d;
    const size_t S2 = 2 * (size_t)d.NLp;
    for (int l = 0; l < d.L; ++l) {
        float *pc = S.pool_c + ((size_t)l * S2 + n) * d.Hp;
        float *pg = S.pool_g + ((size_t)l * S2 + n) * d.G4p;
        for (int i = threadIdx.x; i < d.Hp; i += blockDim.x) pc[i] = 0.f;
        for (int i = threadIdx.x; i < d.G4p; i += blockDim.x) pg[i] = S.bsum[l][i];
    }
    if (threadIdx.x == 0) { S.comm_slot[n] = n; S.new_slot[n] = d.NLp + n; }
}

// etab[v][c'] = sum_k w_ih0[gate * H + unit][k] * embed[v][k]  (c' <-> (unit, gate) as in lstm_weight_prep_kernel); float64
// sums rounded once (setup, once per handle)
__global__ void lstm_etab_kernel(const float *__restrict__ embed, const float *__restrict__ w_ih, int V, int D, int H, int Hp,
                                 float *__restrict__ etab)
{
    const int G4p = 4 * Hp;
    const long total = (long)V * G4p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int v = (int)(i / G4p), c = (int)(i % G4p);
        const int unit = (c >> 5) * 8 + (c & 7), gate = (c & 31) >> 3;
        double a = 0.0;
        if (unit < H) {
            const float *__restrict__ wr = w_ih + (size_t)(gate * H + unit) * D;
            const float *__restrict__ er = embed + (size_t)v * D;
            for (int k = 0; k < D; ++k) a += (double)wr[k] * (double)er[k];
        }
        etab[i] = (float)a;
    }
}

// step API: layer 0 for every lane as a kernel of its own
__global__ void lstm_layer0_kernel(DevState *sp)
{
    const DevState &S = *sp;
    const int n = blockIdx.x;
    if (!(S.lane_active[n] && S.need_pred[n])) return;
    lstm_layer0_cell(S, n, S.token[n], S.comm_slot[n], S.new_slot[n]);
}

// xT[:, n] = embed[tok] for one lane (called by the kernels that decide a lane's next token)
__device__ __forceinline__ void write_embedding_column(DevState *s, int n, int tok)
{
    const Dims &d = s->d;
    for (int k = threadIdx.x; k < d.D; k += blockDim.x) s->xT[(size_t)k * d.NLp + n] = s->embed[(size_t)tok * d.D + k];
}

// A lane starts from the zero state with the blank as its first predictor input
__device__ __forceinline__ void first_predictor_input(DevState *s, int n)
{
    if (s->d.ptype == kPredLstm) {
        lstm_reset_lane(*s, n);
        __syncthreads();                                      // the slot rows written above are read below
        lstm_layer0_cell(*s, n, s->blank, n, s->d.NLp + n);
    } else {
        write_embedding_column(s, n, s->blank);
    }
}

// wt[k][j] = sum_p wf[j][p] * wp[p][k]  (k-major, as lane_gemm reads it),  bias[j] = bf[j] + sum_p wf[j][p] * bp[p];
// products and sums in float64, rounded to fp32 once (setup, once per handle)
__global__ void compose_proj_ffn_kernel(const float *__restrict__ wf /* [J][P] */, const float *__restrict__ bf,
                                        const float *__restrict__ wp /* [P][H] */, const float *__restrict__ bp, int J, int P,
                                        int H, int Jn, float *__restrict__ wt /* [Hp][Jn] */, float *__restrict__ bias)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)H * J) {
        const int k = (int)(idx / J), j = (int)(idx - (long)k * J);
        double a = 0.0;
        for (int p = 0; p < P; ++p) a += (double)wf[(size_t)j * P + p] * (double)wp[(size_t)p * H + k];
        wt[(size_t)k * Jn + j] = (float)a;
    } else if (idx < (long)H * J + J) {
        const int j = (int)(idx - (long)H * J);
        double a = (double)bf[j];
        for (int p = 0; p < P; ++p) a += (double)wf[(size_t)j * P + p] * (double)bp[p];
        bias[j] = (float)a;
    }
}

// ---------------------------------------------------- stateless predictors --
// EmbeddingPredictor.forward_step (predictor.py:325-372) and ConvPredictor.forward_step (:455-481), one lane per
// workgroup.  context = [history slots 0 .. ctx-2 (cache_hT), embedding of the lane's token (xT)];
//   embedding:  weight[h][c] = sum_e context[c][e] * pos[h][e][c]        (pos = pos_embed.weight viewed [heads, D, ctx])
//               comb[e] = sum_h sum_c weight[h][c] * context[c][e] / (heads * ctx)   -> ffn -> LayerNorm -> activation
//   conv:       comb[e] = sum_c context[c][e] * conv_w[e][c] (+ conv_b[e])            -> LayerNorm -> activation
// The new history is context[1 ..] (new_hT): the search kernels commit it on an emission exactly like an LSTM state.
// STAGE 0: embedding predictor, context weighting -> combT (+ new history); the ffn runs as a lane GEMM in between;
// STAGE 1: LayerNorm + activation of ffnT -> outT;  STAGE 2: conv predictor, all of it.
constexpr int kCtxThreads = 256;
constexpr int kCtxPer = 4;                    // embedding width <= 1024
constexpr int kMaxCtx = WR_MAX_LSTM_LAYERS + 1;

__device__ __forceinline__ float ctx_block_sum(float v, float *red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// LayerNorm (biased variance, as torch.nn.LayerNorm) + activation of this thread's elements e = tid + 256 j
__device__ __forceinline__ void ctx_norm_act_store(const DevState &S, const float (&x)[kCtxPer], int n, float *red)
{
    const Dims &d = S.d;
    const int tid = threadIdx.x;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < kCtxPer; ++j) part += (tid + kCtxThreads * j < d.D) ? x[j] : 0.f;
    const float mean = ctx_block_sum(part, red) / (float)d.D;
    part = 0.f;
#pragma unroll
    for (int j = 0; j < kCtxPer; ++j) {
        const float t = x[j] - mean;
        part += (tid + kCtxThreads * j < d.D) ? t * t : 0.f;
    }
    const float rstd = 1.f / sqrtf(ctx_block_sum(part, red) / (float)d.D + d.ln_eps);
#pragma unroll
    for (int j = 0; j < kCtxPer; ++j) {
        const int e = tid + kCtxThreads * j;
        if (e < d.D) S.outT[(size_t)e * d.NLp + n] = act_value(d.pact, (x[j] - mean) * rstd * S.pnorm_w[e] + S.pnorm_b[e]);
    }
}

template <int STAGE>
__global__ __launch_bounds__(kCtxThreads) void ctx_predictor_kernel(DevState *sp)
{
    const DevState &S = *sp;
    const Dims &d = S.d;
    const int n = blockIdx.x, tid = threadIdx.x;
    if (!(S.lane_active[n] && S.need_pred[n])) return;      // predicated per lane, like the LSTM cell epilogue
    __shared__ float red[4];
    __shared__ float part[64 * 4];
    const int D = d.D, ctx = d.ctx;
    const size_t ls = (size_t)d.Hp * d.NLp;
    if (STAGE == 1) {
        float x[kCtxPer];
#pragma unroll
        for (int j = 0; j < kCtxPer; ++j) {
            const int e = tid + kCtxThreads * j;
            x[j] = e < D ? S.ffnT[(size_t)e * d.NLp + n] : 0.f;
        }
        ctx_norm_act_store(S, x, n, red);
        return;
    }
    float cin[kMaxCtx][kCtxPer];
#pragma unroll
    for (int c = 0; c < kMaxCtx; ++c)
#pragma unroll
        for (int j = 0; j < kCtxPer; ++j) {
            const int e = tid + kCtxThreads * j;
            float v = 0.f;
            if (c < ctx && e < D) v = c < ctx - 1 ? S.cache_hT[(size_t)c * ls + (size_t)e * d.NLp + n] : S.xT[(size_t)e * d.NLp + n];
            cin[c][j] = v;
        }
    // new history = context[1 ..]
#pragma unroll
    for (int c = 1; c < kMaxCtx; ++c)
#pragma unroll
        for (int j = 0; j < kCtxPer; ++j) {
            const int e = tid + kCtxThreads * j;
            if (c < ctx && e < D) S.new_hT[(size_t)(c - 1) * ls + (size_t)e * d.NLp + n] = cin[c][j];
        }
    float o[kCtxPer];
    if (STAGE == 0) {
        const int heads = d.heads;
#pragma unroll
        for (int c = 0; c < kMaxCtx; ++c) {
            if (c >= ctx) continue;
            for (int h = 0; h < heads; ++h) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < kCtxPer; ++j) {
                    const int e = tid + kCtxThreads * j;
                    if (e < D) p += cin[c][j] * S.pos_w[((size_t)h * D + e) * ctx + c];
                }
                p = wave_sum(p);
                if ((tid & 63) == 0) part[(h * ctx + c) * 4 + (tid >> 6)] = p;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kCtxPer; ++j) o[j] = 0.f;
        for (int h = 0; h < heads; ++h) {                     // per head: weight[h] @ context, then the sum over heads
            float t[kCtxPer];
#pragma unroll
            for (int j = 0; j < kCtxPer; ++j) t[j] = 0.f;
#pragma unroll
            for (int c = 0; c < kMaxCtx; ++c) {
                if (c >= ctx) continue;
                const float *q = part + (h * ctx + c) * 4;
                const float w = q[0] + q[1] + q[2] + q[3];
#pragma unroll
                for (int j = 0; j < kCtxPer; ++j) t[j] += w * cin[c][j];
            }
#pragma unroll
            for (int j = 0; j < kCtxPer; ++j) o[j] += t[j];
        }
        const float inv = 1.f / (float)(heads * ctx);
#pragma unroll
        for (int j = 0; j < kCtxPer; ++j) {
            const int e = tid + kCtxThreads * j;
            if (e < D) S.combT[(size_t)e * d.NLp + n] = o[j] * inv;
        }
    } else {
#pragma unroll
        for (int j = 0; j < kCtxPer; ++j) {
            const int e = tid + kCtxThreads * j;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < kMaxCtx; ++c)
                if (c < ctx && e < D) a += cin[c][j] * S.conv_w[(size_t)e * ctx + c];
            o[j] = (e < D && S.conv_b) ? a + S.conv_b[e] : a;
        }
        ctx_norm_act_store(S, o, n, red);
    }
}

// Tail of the kernels that decide a lane's token: the commit (greedy_search copy.py:52 `cache = new_cache`) and the
// next predictor input.  LSTM predictor: a commit swaps the lane's two slot numbers, and layer 0 of the predictor step
// the emission causes is evaluated here (lstm_layer0_cell) from the committed slot; stateless predictors copy their token
// history and leave the new token's embedding column for ctx_predictor_kernel.  `sc` / `sn`: the lane's slots as
// loaded at the top of the kernel.
__device__ __forceinline__ void commit_and_feed(const DevState &S, int n, bool feed, bool commit, int tok, int sc, int sn)
{
    const Dims &d = S.d;
    const int tid = threadIdx.x;
    if (d.ptype == kPredLstm) {
        if (commit) {
            const int t = sc; sc = sn; sn = t;
            if (tid == 0) { S.comm_slot[n] = sc; S.new_slot[n] = sn; }
        }
        if (feed) lstm_layer0_cell(S, n, tok, sc, sn);
        return;
    }
    if (feed) {
        for (int q = tid; q < d.D; q += blockDim.x) S.xT[(size_t)q * d.NLp + n] = S.embed[(size_t)tok * d.D + q];
    }
    if (commit) {
        for (int i = tid; i < d.L * d.Hp; i += blockDim.x) {
            const size_t o = (size_t)i * d.NLp + n;
            S.cache_hT[o] = S.new_hT[o];
            S.cache_cT[o] = S.new_cT[o];
        }
    }
}

// ---------------------------------------------------------------- greedy --
__global__ void greedy_init_kernel(DevState *s)
{
    const Dims &d = s->d;
    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < d.L * d.Hp; i += blockDim.x) {
        const size_t o = (size_t)i * d.NLp + n;
        s->cache_hT[o] = 0.f; s->cache_cT[o] = 0.f; s->new_hT[o] = 0.f; s->new_cT[o] = 0.f;
    }
    if (tid == 0) {
        const int T = s->enc_lens[n] < s->T ? s->enc_lens[n] : s->T;
        s->token[n] = s->blank;
        s->lane_t[n] = 0;
        s->noblk[n] = 0;
        s->need_pred[n] = 1;
        s->new_is_cache[n] = 0;
        s->hyp_lens[n] = 0;
        const int act = T > 0;
        s->lane_active[n] = act;
        if (act) atomicAdd(s->active_count, 1);
    }
    first_predictor_input(s, n);
}

// Streaming: start the next chunk of every stream with the state the previous chunk left behind
// (reset_cache / forward_greedy_search of "transducer ref.py":541-606).  `ref_new_cache` reproduces the
// reference's `new_cache = self.cache` at the top of each chunk (the not-yet-committed predictor state of the
// previous chunk is dropped); 0 keeps it, so that chunked decoding equals decoding the concatenation.
__global__ void greedy_chunk_init_kernel(DevState *s, int ref_new_cache)
{
    const Dims &d = s->d;
    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    (void)d;
    if (tid == 0) {
        const int T = s->enc_lens[n] < s->T ? s->enc_lens[n] : s->T;
        s->lane_t[n] = 0;
        s->hyp_lens[n] = 0;
        // `new_cache = self.cache` at the top of a chunk: until the predictor steps again, committing is a no-op.
        // (The pending state itself is left alone -- the predictor OUTPUT of the last step stays valid, as
        // self.pred_out_step does in the reference.)
        s->new_is_cache[n] = ref_new_cache ? 1 : 0;
        const int act = T > 0;
        s->lane_active[n] = act;
        if (act) atomicAdd(s->active_count, 1);
    }
}

// The lane's scalars and its row records are fetched up front and the state machine runs redundantly in every thread
// (thread 0 stores), so the kernel has one memory round trip before the reductions and one after.
// HW = true: the hot-word variant (greedy_search.py:297-430).  The gate decision of the NEXT predictor step is made
// here, right after the emission that causes that step -- the gate is a function of the frame alone (see
// hw_gate_table_kernel), so it does not have to wait for the predictor.  A "go-back" (:369-385) withdraws exactly
// the emission this kernel has just decided on (the reference pops the last token, predictor output, cache and
// input and steps the predictor again from the popped state), so it costs nothing to undo: the token is not
// appended, the cache not committed, the next input not replaced.
template <bool HW, int PP>
__global__ __launch_bounds__(256) void greedy_update_kernel(DevState *sp, LaneArgs a)
{
    WR_STAMP_DECL;
    WR_STAMP_RT(7);
    WR_STAMP(0);
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    // no branch before the loads: everything this lane needs is requested in one batch, the per-lane state through the
    // kernel-argument pointers (no dependence on the state block), idle lanes leave after the first reduction
    const int act = a.lane_active[n];
    int t = a.lane_t[n], nb = a.noblk[n];
    const int need = a.need_pred[n];
    int nic = a.new_is_cache[n];
    const int sc = a.comm_slot[n], sn = a.new_slot[n];     // LSTM predictor: state slots (unused otherwise)
    // The row is resolved from the per-block records the ffn_out epilogue left (kEpiRowStats): {block max, sum of
    // exp(x - block max), runner-up, first index of the block max} for every 32 columns -- V / 32 records instead of V
    // logits.  log_softmax as the reference evaluates it, lp = (x - max) - log(sum(exp(x - max))), and its argmax with
    // the first index on ties: lp is a monotone function of x, so the maximum of lp is (max - max) - ls and the
    // winner is the first column whose lp equals it.  A block takes part if its maximum does; if its runner-up does too
    // (two logits that lp cannot tell apart: they differ by less than half an ulp of ls), its 32 logits are re-read and
    // scanned in order -- otherwise the recorded first index is the answer.
    // Every WAVE resolves the row on its own from all the records (PP per lane; they are a few KB and come from L2 once
    // per CU): the three reductions are register butterflies, no LDS and no workgroup barrier on the way to the token.
    const float4 *__restrict__ rp = a.row_part + (size_t)n * a.n_cb;
    const int ncb = (a.V + 31) / 32;
    float4 rec[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        const int bq = lane + i * 64;
        rec[i] = rp[bq < ncb ? bq : ncb - 1];
    }
    const DevState S = *sp;                         // by value: no reloads of the pointers after the stores below
    const Dims &d = S.d;
    const int len = S.hyp_lens[n];
    const int enc_len = S.enc_lens[n];
    int gate_cur = 1, gb_flag = 0, gb_end = 0, last_t = 0, tlen = 0, gate_t0 = 1, gate_t1 = 1, last_gate = 1;
    if (HW) {
        gate_cur = S.cur_gate[n]; gb_flag = S.gb_flag[n]; gb_end = S.gb_end[n]; last_t = S.last_t[n];
        tlen = S.trace_len[n];
        const int tc = t < S.T ? t : S.T - 1, tn = t + 1 < S.T ? t + 1 : S.T - 1;
        gate_t0 = S.gate_tab[(size_t)n * S.T + tc];                  // the next predictor step sees frame t or t + 1
        gate_t1 = S.gate_tab[(size_t)n * S.T + tn];
        last_gate = S.trace[(size_t)n * S.trace_cap + (tlen > 0 ? (tlen <= S.trace_cap ? tlen - 1 : S.trace_cap - 1) : 0)];
    }
    float m = -3.0e38f;
#pragma unroll
    for (int i = 0; i < PP; ++i) m = fmaxf(m, (lane + i * 64 < ncb) ? rec[i].x : -3.0e38f);
    m = wave_allmax_dpp(m);
    WR_STAMP(1);                                   // state and records arrived, first reduction done
    if (!act) return;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PP; ++i) sum += (lane + i * 64 < ncb) ? rec[i].y * expf(rec[i].x - m) : 0.f;
    sum = wave_allsum_dpp(sum);
    const float ls = logf(sum);
    const float top = (m - m) - ls;                 // log-probability of the maximum, exactly as the elementwise formula gives it
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        const int bq = lane + i * 64;
        if (bq >= ncb) continue;
        if (((rec[i].x - m) - ls) != top) continue;
        int idx = __builtin_bit_cast(int, rec[i].w);
        if (((rec[i].z - m) - ls) == top) {          // runner-up indistinguishable from the block maximum: scan the block
            const float *__restrict__ x = S.logits + (size_t)n * d.V;
            idx = 0x7fffffff;
            for (int c = 31; c >= 0; --c) {
                const int v = bq * 32 + c;
                if (v < d.V && ((x[v] - m) - ls) == top) idx = v;
            }
        }
        if (idx < bi) bi = idx;
    }
    bi = wave_allmin_dpp(bi);
    WR_STAMP(2);                                   // token decided
    const int k = bi;
    const bool emit = (k != S.blank);
    if (need) nic = 0;                             // the predictor stepped in this micro-step
    bool commit = emit && !nic;
    int need_next = need;
    const int t_dec = t;                           // the frame this decision was made on
    if (emit) { need_next = 1; nb += 1; }
    if (!emit || nb >= S.n_steps) {
        if (!emit) need_next = 0;
        t += 1;
        nb = 0;
    }
    const int T = enc_len < S.T ? enc_len : S.T;
    // hot-word gate of the predictor step this emission leads to (greedy_search.py:357-392)
    bool withdraw = false;
    int push0 = -1, push1 = -1;                    // values appended to the trace (after an optional pop)
    bool pop = false;
    if (HW && emit && t < T) {
        const int gt = (t == t_dec) ? gate_t0 : gate_t1;
        if (S.hw_filter) {
            if (!gb_flag) {
                if (gt == 0) { push0 = 0; last_t = t; gate_cur = 0; }
                else if (tlen > 0 && last_gate == 0) {
                    // go-back: forget the gate-0 step and the token it produced, resume from its frame with biasing on
                    withdraw = true;
                    gb_end = t; t = last_t; gb_flag = 1;
                    pop = true;
                    nb -= 1;
                    push0 = 1;                     // the re-run step is recorded with the gate forced to 1 (:388-391)
                    if (t >= gb_end) gb_flag = 0;
                    gate_cur = 1;
                } else { push0 = 1; gate_cur = 1; }
            } else {
                push0 = 1;
                if (t >= gb_end) gb_flag = 0;
                gate_cur = 1;
            }
        } else { push0 = 1; gate_cur = 1; }
    }
    (void)push1;
    if (withdraw) commit = false;
    if (tid == 0) {
        atomicAdd(S.active_count + (emit ? 2 : 1), 1);               // decision statistics for the look-ahead policy
        if (need) S.new_is_cache[n] = 0;
        if (emit && !withdraw) {
            if (len < S.max_hyp) S.hyps[(size_t)n * S.max_hyp + len] = k;
            S.hyp_lens[n] = len + 1;
            S.token[n] = k;
        }
        S.need_pred[n] = need_next;
        S.lane_t[n] = t;
        S.noblk[n] = nb;
        if (HW) {
            int tl = tlen - (pop ? 1 : 0);
            if (push0 >= 0) {
                if (tl < S.trace_cap) S.trace[(size_t)n * S.trace_cap + tl] = push0;
                tl += 1;
            }
            S.trace_len[n] = tl;
            S.cur_gate[n] = gate_cur;
            S.gb_flag[n] = gb_flag;
            S.gb_end[n] = gb_end;
            S.last_t[n] = last_t;
        }
        if (t >= T) {
            S.lane_active[n] = 0;
            atomicSub(S.active_count, 1);
        }
    }
    commit_and_feed(S, n, emit && !withdraw, commit, k, sc, sn);
    WR_STAMP(5);
    WR_STAMP_DRAIN();
    WR_STAMP(6);
    WR_STAMP_RT(8);
    WR_STAMP_FLUSH(HW ? 5 : 4);
}

// ---- look-ahead greedy: `look` encoder frames per micro-step against one predictor state -------------------
// While a lane keeps answering blank its predictor output does not change, so the joiner rows of frames t, t+1, ...
// can be evaluated together: greedy_rows_kernel takes the argmax of every (frame, lane) row in parallel,
// greedy_resolve_kernel then walks a lane's rows in order through the same state machine as greedy_update_kernel
// and stops at the first emission (the predictor must step before the next decision).  Token sequences are
// identical to the one-frame-per-step loop; a run of blanks costs one micro-step instead of `look`.
template <int NV>
__global__ __launch_bounds__(256) void greedy_rows_kernel(DevState *sp)
{
    __shared__ float sv[4];
    __shared__ int si[4];
    const DevState S = *sp;
    const Dims &d = S.d;
    const int n = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const int act = S.lane_active[n];
    const int t = S.lane_t[n];
    const int enc_len = S.enc_lens[n];
    const float *__restrict__ x = S.logits + ((size_t)f * d.NLp + n) * d.V;
    float xv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256;
        xv[i] = x[v < d.V ? v : d.V - 1];
    }
    float m = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NV; ++i) m = fmaxf(m, (tid + i * 256 < d.V) ? xv[i] : -3.0e38f);
    m = block_max(m, sv);
    const int T = enc_len < S.T ? enc_len : S.T;
    if (!act || t + f >= T) return;                // frames past the end are never consulted
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sum += (tid + i * 256 < d.V) ? expf(xv[i] - m) : 0.f;
    sum = block_sum(sum, sv);
    const float ls = logf(sum);
    float best = -3.0e38f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256;
        const float lp = (xv[i] - m) - ls;
        if (v < d.V && lp > best) { best = lp; bi = v; }   // strided ascending scan keeps the lowest index per thread
    }
    block_argmax(best, bi, sv, si);
    if (tid == 0) S.row_tok[(size_t)f * d.NLp + n] = bi;
}

__global__ __launch_bounds__(256) void greedy_resolve_kernel(DevState *sp, int look)
{
    const DevState S = *sp;
    const Dims &d = S.d;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int act = S.lane_active[n];
    int t = S.lane_t[n], nb = S.noblk[n];
    const int need0 = S.need_pred[n];
    int nic = S.new_is_cache[n];
    int len = S.hyp_lens[n];
    const int enc_len = S.enc_lens[n];
    const int sc = S.comm_slot[n], sn = S.new_slot[n];
    int toks[kMaxLook];
#pragma unroll
    for (int f = 0; f < kMaxLook; ++f) toks[f] = S.row_tok[(size_t)(f < look ? f : 0) * d.NLp + n];
    if (!act) return;
    const int T = enc_len < S.T ? enc_len : S.T;
    if (need0) nic = 0;                            // the predictor stepped in this micro-step
    int need = need0, emitted = -1, n_blank = 0;
    bool commit = false, done = false;
#pragma unroll
    for (int f = 0; f < kMaxLook; ++f) {
        if (f >= look || done) continue;
        const int k = toks[f];
        const bool emit = (k != S.blank);
        n_blank += emit ? 0 : 1;
        if (emit) {
            if (tid == 0 && len < S.max_hyp) S.hyps[(size_t)n * S.max_hyp + len] = k;
            len += 1;
            need = 1;
            nb += 1;
            commit = !nic;
            emitted = k;
        }
        if (!emit || nb >= S.n_steps) {
            if (!emit) need = 0;
            t += 1;
            nb = 0;
        }
        done = emit || t >= T;                     // after an emission the remaining rows belong to a stale predictor state
    }
    if (tid == 0) {
        if (n_blank) atomicAdd(S.active_count + 1, n_blank);         // decision statistics for the look-ahead policy
        if (emitted >= 0) atomicAdd(S.active_count + 2, 1);
        if (need0) S.new_is_cache[n] = 0;
        S.hyp_lens[n] = len;
        if (emitted >= 0) S.token[n] = emitted;
        S.need_pred[n] = need;
        S.lane_t[n] = t;
        S.noblk[n] = nb;
        if (t >= T) {
            S.lane_active[n] = 0;
            atomicSub(S.active_count, 1);
        }
    }
    commit_and_feed(S, n, emitted >= 0, commit, emitted, sc, sn);
}

// ------------------------------------------------------- hot-word greedy --
// Device copy of the ContextBias weights the greedy loop uses (wenet/transformer/context_bias.py:375-394), re-laid
// k-major where a thread block streams them (see wr_decoder_attach_hotword).
struct HwDev {
    int D, heads, HW, NLAB, max_ctx;
    const float *q_wt, *q_b;          // [D][D] k-major
    const float *k_w, *k_b, *v_w, *v_b;   // row-major (nn.Linear), used once per call for the list projections
    const float *o_wt, *o_b;          // [D][D] k-major
    const float *bn_w, *bn_b;         // predictor_bias_bias_norm
    const float *c_wt, *c_b;          // [2D][D] k-major
    const float *on_w, *on_b;         // predictor_bias_out_norm
    const float *he_w, *he_b, *hv_w, *hv_b, *ho_w, *ho_b, *hn_w, *hn_b, *hl_w, *hl_b;   // gate chain, row-major
    float *kbuf[2], *vbuf[2];         // [max_ctx][D]: K / V of the empty list (0) and of the hot-word list (1)
    float *cold_c;                    // [D] one-entry empty list: combine(second half) of its constant bias feature + bias
    int32_t *gate_tab;                // [max_utt * Tmax]
    // attention folded over a short list (heads * entries <= D), built once per call by hw_fold_kernel:
    //   scores = mqT^T x + mq_c      mqT[k][r] = sum_{j in head h} Wq[j][k] K[c][j] / sqrt(d_k),  r = h * n_ctx + c
    //   out    = noT^T p + b_o       noT[r][j] = sum_{i in head h} Wo[j][i] V[c][i]
    float *mqT[2], *mq_c[2], *noT[2]; // [D][D], [D], [D][D] per list (used when 1 < entries and heads * entries <= D)
    // the predictor's projection (predictor.py:198) is evaluated here too: x = proj_wt^T h + proj_b, from the LSTM's last
    // layer output (k-major [Hp][NLp]) -- one launch per decision less than a projection GEMM of its own
    const float *proj_wt, *proj_b, *h_lastT;
    int Hp, P, proj_ld, NLp;
    // ... and where the hidden size allows (Hp <= D) it is composed into what consumes x: nothing non-linear sits between
    // the projection and the query / combine Linears, so  combine[:, :D] (Wp h + bp) = c1p_wt^T h + c1p_b  (per handle,
    // float64 sums) and  scores = mqpT^T h + mqp_c  (per call, short lists) -- one dependent matrix-vector stage less
    float *c1p_wt, *c1p_b;            // [Hp][D], [D]
    float *cold_cp;                   // [D] cold_c with the composed bias: the one-entry empty list's whole constant
    float *mqpT[2], *mqp_c[2];        // [Hp][D], [D] per list
    // handle-constant pointers of the lane state the bias kernel starts from (kernel arguments: no wait for the state block)
    const int32_t *lane_active, *need_pred, *cur_gate;
};

// K / V projections of an encoded context list: out[c][j] = b[j] + sum_k hidden[c][k] * W[j][k].
// grid (n_ctx, 2): y = 0 -> K, 1 -> V.  A wave per output, lanes over k (row-major weights: coalesced), wave_sum.
__global__ __launch_bounds__(256) void hw_kv_kernel(HwDev hw, const float *__restrict__ hidden, int list)
{
    const int c = blockIdx.x, which = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *__restrict__ W = which ? hw.v_w : hw.k_w;
    const float *__restrict__ bb = which ? hw.v_b : hw.k_b;
    float *__restrict__ out = (which ? hw.vbuf[list] : hw.kbuf[list]) + (size_t)c * hw.D;
    const float *__restrict__ x = hidden + (size_t)c * hw.D;
    for (int j = wave; j < hw.D; j += 4) {
        float acc = 0.f;
        for (int k = lane; k < hw.D; k += 64) acc = fmaf(W[(size_t)j * hw.D + k], x[k], acc);
        acc = wave_sum(acc);
        if (lane == 0) out[j] = acc + bb[j];
    }
}

// y[j] = b[j] + sum_k W[j][k] * x[k] for a small row-major Linear inside one workgroup (x, y in LDS): a wave per output
__device__ __forceinline__ void block_linear_rowmajor(const float *__restrict__ W, const float *__restrict__ b, const float *x,
                                                      int K, int N, float *y)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j = wave; j < N; j += nw) {
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) acc = fmaf(W[(size_t)j * K + k], x[k], acc);
        acc = wave_sum(acc);
        if (lane == 0) y[j] = acc + b[j];
    }
    __syncthreads();
}

// LayerNorm over v[0..N) (LDS) as torch.nn.LayerNorm: biased variance, eps = 1e-5.  Result through `emit(j, value)`.
template <typename F>
__device__ __forceinline__ void block_layer_norm(const float *v, int N, const float *__restrict__ w, const float *__restrict__ b,
                                                 float *sv, F emit)
{
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += blockDim.x) s += v[j];
    const float mean = block_sum(s, sv) / (float)N;
    float q = 0.f;
    for (int j = threadIdx.x; j < N; j += blockDim.x) { const float dlt = v[j] - mean; q += dlt * dlt; }
    const float var = block_sum(q, sv) / (float)N;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    for (int j = threadIdx.x; j < N; j += blockDim.x) emit(j, (v[j] - mean) * rstd * w[j] + b[j]);
    __syncthreads();
}

// The hot-word gate of every frame (ContextBias.forward_hw_pred_both, context_bias.py:388-394, then topk(1) as
// greedy_search.py:359-364).  hw_bias attends from one query (the predictor-side feature) to ONE key / value (the
// encoder-side feature of frame t): softmax over a single score is exactly 1.0, so the attention output is exactly
// linear_out(linear_v(hw_output_layer_enc(enc_feat[t]))) whatever the query is -- the gate depends on the frame
// alone and is evaluated for all frames before the loop.  One workgroup per frame.
__global__ __launch_bounds__(256) void hw_gate_table_kernel(HwDev hw, const float *__restrict__ enc_feat, long rows)
{
    extern __shared__ float sm[];                   // x[D] | a[HW] | b[HW]
    __shared__ float sv[4];
    float *x = sm, *a = sm + hw.D, *bq = a + hw.HW;
    const long r = blockIdx.x;
    if (r >= rows) return;
    for (int k = threadIdx.x; k < hw.D; k += blockDim.x) x[k] = enc_feat[(size_t)r * hw.D + k];
    __syncthreads();
    block_linear_rowmajor(hw.he_w, hw.he_b, x, hw.D, hw.HW, a);         // hw_output_layer_enc
    block_linear_rowmajor(hw.hv_w, hw.hv_b, a, hw.HW, hw.HW, bq);       // hw_bias.linear_v  (attention weight == 1)
    block_linear_rowmajor(hw.ho_w, hw.ho_b, bq, hw.HW, hw.HW, a);       // hw_bias.linear_out
    block_layer_norm(a, hw.HW, hw.hn_w, hw.hn_b, sv, [&](int j, float v) { bq[j] = v; });   // hw_bias_norm
    block_linear_rowmajor(hw.hl_w, hw.hl_b, bq, hw.HW, hw.NLAB, a);     // hw_output_layer
    if (threadIdx.x == 0) {
        int best = 0;
        for (int j = 1; j < hw.NLAB; ++j)
            if (a[j] > a[best]) best = j;            // topk(1): the largest, first index on ties
        hw.gate_tab[r] = best;
    }
}

// y[j] = b[j] + sum_k WT[k][j] * x[k] with the K range split over the 16 waves of the workgroup (every wave has its
// whole share of the k-major weight rows in flight at once), partial sums through LDS.  x, y, part in LDS.
constexpr int kHwThreads = 1024;
__device__ __forceinline__ void block_gemv_kmajor(const float *__restrict__ WT, int ldw, const float *__restrict__ b, const float *x,
                                                  int K, int N, float *part, float *y)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = kHwThreads / 64;
    const int kper = (K + NW - 1) / NW;
    const int k0 = wave * kper;
    int k1 = k0 + kper;
    k1 = k1 < K ? k1 : K;
    for (int j0 = 0; j0 < N; j0 += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = k0; k < k1; ++k) {
            const float xv = x[k];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = j0 + lane + 64 * i;
                acc[i] = fmaf(WT[(size_t)k * ldw + (j < N ? j : N - 1)], xv, acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = j0 + lane + 64 * i;
            if (j < N) part[wave * N + j] = acc[i];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += kHwThreads) {
        float sacc = b[j];
#pragma unroll
        for (int w = 0; w < NW; ++w) sacc += part[w * N + j];
        y[j] = sacc;
    }
    __syncthreads();
}

// c1p_wt[i][j] = sum_p combine_w[j][p] * Wp[p][i],  c1p_b[j] = combine_b[j] + sum_p combine_w[j][p] * bp[p]   (p < P = D: the
// first half of the combine Linear's input); float64 sums rounded once.  proj_wt is the handle's k-major projection weight.
__global__ void hw_compose_kernel(const float *__restrict__ cw /* [D][2D] */, const float *__restrict__ cb,
                                  const float *__restrict__ proj_wt /* [Hp][Pn] */, const float *__restrict__ proj_b, int D, int P,
                                  int H, int Hp, int Pn, float *__restrict__ wt /* [Hp][D] */, float *__restrict__ bias)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)Hp * D) {
        const int i = (int)(idx / D), j = (int)(idx - (long)i * D);
        double a = 0.0;
        if (i < H)
            for (int p = 0; p < P; ++p) a += (double)cw[(size_t)j * 2 * D + p] * (double)proj_wt[(size_t)i * Pn + p];
        wt[idx] = (float)a;
    } else if (idx < (long)Hp * D + D) {
        const int j = (int)(idx - (long)Hp * D);
        double a = (double)cb[j];
        for (int p = 0; p < P; ++p) a += (double)cw[(size_t)j * 2 * D + p] * (double)proj_b[p];
        bias[j] = (float)a;
    }
}

// Per call, per list with heads * entries <= D: the query projection folded into the keys and the output projection folded
// into the values (see HwDev).  One workgroup per (head, entry) row r.
__global__ __launch_bounds__(256) void hw_fold_kernel(HwDev hw, int list, int nctx)
{
    const int r = blockIdx.x, tid = threadIdx.x;
    const int D = hw.D, dk = D / hw.heads;
    const int hh = r / nctx, c = r - hh * nctx;
    const float *__restrict__ kr = hw.kbuf[list] + (size_t)c * D + hh * dk;
    const float *__restrict__ vr = hw.vbuf[list] + (size_t)c * D + hh * dk;
    const float inv = 1.f / sqrtf((float)dk);
    const int R = hw.heads * nctx;
    for (int k = tid; k < D; k += 256) {
        const float *__restrict__ wq = hw.q_wt + (size_t)k * D + hh * dk;      // q_wt is k-major: [in k][out j]
        float a = 0.f;
        for (int j = 0; j < dk; ++j) a = fmaf(wq[j], kr[j], a);
        hw.mqT[list][(size_t)k * R + r] = a * inv;
        float o = 0.f;
        for (int i = 0; i < dk; ++i) o = fmaf(hw.o_wt[(size_t)(hh * dk + i) * D + k], vr[i], o);   // o_wt: [in i][out j = k here]
        hw.noT[list][(size_t)r * D + k] = o;
    }
    if (tid == 0) {
        float a = 0.f;
        for (int j = 0; j < dk; ++j) a = fmaf(hw.q_b[hh * dk + j], kr[j], a);
        hw.mq_c[list][r] = a * inv;
    }
    if (hw.proj_wt == nullptr || hw.Hp > D) return;
    // ... and the predictor's projection composed in front: mqpT[i][r] = sum_k Wp[k][i] mqT[k][r]
    __syncthreads();                                          // this workgroup's column of mqT is complete
    __shared__ float mcol[1024];
    for (int k = tid; k < D; k += 256) mcol[k] = hw.mqT[list][(size_t)k * R + r];
    __syncthreads();
    for (int i = tid; i < hw.Hp; i += 256) {
        const float *__restrict__ wp = hw.proj_wt + (size_t)i * hw.proj_ld;
        float a = 0.f;
        for (int k = 0; k < D; ++k) a = fmaf(wp[k], mcol[k], a);
        hw.mqpT[list][(size_t)i * R + r] = a;
    }
    if (tid == 0) {
        float a = hw.mq_c[list][r];
        for (int k = 0; k < D; ++k) a = fmaf(hw.proj_b[k], mcol[k], a);
        hw.mqp_c[list][r] = a;
    }
}

// ContextBias.forward_predictor_bias (context_bias.py:375-381) for the lanes whose predictor has just stepped, with
// the list the gate selected (cur_gate: 1 = hot words, 0 = empty list -- greedy_search.py:357,394-395 evaluate the hot
// variant first and replace it when the gate says 0; only the one that reaches the joiner is computed here):
//   pb  = predictor_bias_bias_norm(MultiHeadedAttention(query = pred, key = value = bias_hidden))
//   out = predictor_bias_out_norm(predictor_bias_combine(cat(pred, pb)))
// One workgroup per lane; K / V of both lists were projected once per call (hw_kv_kernel).
__global__ __launch_bounds__(kHwThreads) void hw_bias_kernel(DevState *sp, HwDev hw)
{
    extern __shared__ float sm[];                   // x[D] q[D] ctx[D] o[D] pb[D] | part[16][D] | p[heads * max_ctx]
    __shared__ float sv[kHwThreads / 64];
    const int n = blockIdx.x, tid = threadIdx.x;
    // the lane's flags and (below) its LSTM output are requested through the kernel-argument pointers, beside the load of
    // the state block instead of behind it
    const int la = hw.lane_active[n], np = hw.need_pred[n], gate = hw.cur_gate[n];
    const float h_early = (hw.proj_wt != nullptr && tid < hw.Hp) ? hw.h_lastT[(size_t)tid * hw.NLp + n] : 0.f;   // Hp <= 1024 = threads
    const DevState S = *sp;
    const Dims &d = S.d;
    if (!la || !np) return;                                       // after a blank the predictor output is kept
    const int D = hw.D, g = gate ? 1 : 0;
    const int nctx = S.hw_nctx[g];
    const float *__restrict__ Kb = hw.kbuf[g];
    const float *__restrict__ Vb = hw.vbuf[g];
    float *x = sm, *q = sm + D, *ctx = sm + 2 * D, *o = sm + 3 * D, *pb = sm + 4 * D;
    float *part = sm + 5 * D, *p = part + (kHwThreads / 64) * D;
    // x = projection(h_last) (predictor.py:198): the lane's column of the k-major LSTM output through LDS
    // (stateless predictors have no projection: their kernels leave the output in outT)
    const bool short_list = nctx > 1 && hw.heads * nctx <= D;     // (a function of the call's list length: hw_fold_kernel ran)
    const bool composed = hw.proj_wt != nullptr && hw.Hp <= D && (short_list || (g == 0 && nctx == 1));
    float *hcol = pb;                                             // the lane's LSTM output (pb is free until the first LayerNorm)
    if (composed) {
        if (tid < hw.Hp) hcol[tid] = h_early;
        __syncthreads();
    } else if (hw.proj_wt != nullptr) {
        if (tid < hw.Hp) q[tid] = h_early;
        __syncthreads();
        block_gemv_kmajor(hw.proj_wt, hw.proj_ld, hw.proj_b, q, hw.Hp, hw.P, part, x);
    } else {
        for (int j = tid; j < D; j += kHwThreads) x[j] = S.outT[(size_t)j * d.NLp + n];
        __syncthreads();
    }
    if (g == 0 && nctx == 1) {
        // The empty list has ONE entry: softmax over one score is exactly 1, the context is exactly that entry's value
        // row, so the bias feature -- and the second half of the combine Linear applied to it -- is the same vector
        // for every step of the call (hw_cold_kernel computed it once).  Only the first half depends on the predictor.
        if (composed) block_gemv_kmajor(hw.c1p_wt, D, hw.cold_cp, hcol, hw.Hp, D, part, ctx);
        else block_gemv_kmajor(hw.c_wt, D, hw.cold_c, x, D, D, part, ctx);
        block_layer_norm(ctx, D, hw.on_w, hw.on_b, sv, [&](int j, float v) { S.biasT[(size_t)j * d.NLp + n] = v; });
        return;
    }
    if (short_list) {
        // short list: scores and the attention output as ONE matrix-vector product each (folded projections, HwDev)
        const int R = hw.heads * nctx;
        if (composed) {
            block_gemv_kmajor(hw.mqpT[g], R, hw.mqp_c[g], hcol, hw.Hp, R, part, p);
            block_gemv_kmajor(hw.c1p_wt, D, hw.c1p_b, hcol, hw.Hp, D, part, q);   // first half of the combine Linear
        } else {
            block_gemv_kmajor(hw.mqT[g], R, hw.mq_c[g], x, D, R, part, p);
            block_gemv_kmajor(hw.c_wt, D, hw.c_b, x, D, D, part, q);        // first half of the combine Linear (needs x only)
        }
        {   // softmax over the list, one wave per head (torch.softmax: exp(x - max) / sum)
            const int lane = tid & 63, wave = tid >> 6;
            for (int hh = wave; hh < hw.heads; hh += kHwThreads / 64) {
                float mx = -3.0e38f;
                for (int c = lane; c < nctx; c += 64) mx = fmaxf(mx, p[hh * nctx + c]);
                mx = wave_max(mx);
                float ssum = 0.f;
                for (int c = lane; c < nctx; c += 64) { const float e = expf(p[hh * nctx + c] - mx); p[hh * nctx + c] = e; ssum += e; }
                ssum = wave_sum(ssum);
                for (int c = lane; c < nctx; c += 64) p[hh * nctx + c] = p[hh * nctx + c] / ssum;
            }
        }
        __syncthreads();
        block_gemv_kmajor(hw.noT[g], D, hw.o_b, p, R, D, part, o);
        block_layer_norm(o, D, hw.bn_w, hw.bn_b, sv, [&](int j, float v) { pb[j] = v; });
        block_gemv_kmajor(hw.c_wt + (size_t)D * D, D, q, pb, D, D, part, ctx);
        block_layer_norm(ctx, D, hw.on_w, hw.on_b, sv, [&](int j, float v) { S.biasT[(size_t)j * d.NLp + n] = v; });
        return;
    }
    block_gemv_kmajor(hw.q_wt, D, hw.q_b, x, D, D, part, q);         // linear_q
    // scores[h][c] = q_h . k_{c,h} / sqrt(d_k)   (attention.py:185): four threads per (head, entry) pair, each a
    // quarter of the head's d_k elements with its loads in flight together, summed across the quad by DPP
    const int dk = D / hw.heads;
    const float sq = sqrtf((float)dk);
    {
        const int quarter = (dk + 3) / 4, sub = tid & 3;
        for (int pr0 = 0; pr0 < hw.heads * nctx; pr0 += kHwThreads / 4) {
            const int idx = pr0 + (tid >> 2);
            const bool on = idx < hw.heads * nctx;
            const int hh = on ? idx / nctx : 0, c = on ? idx - hh * nctx : 0;
            const float *__restrict__ kr = Kb + (size_t)c * D + hh * dk;
            const float *qh = q + hh * dk;
            float acc = 0.f;
#pragma unroll 16
            for (int i = 0; i < quarter; ++i) {
                const int e = sub * quarter + i;
                const float kv = kr[e < dk ? e : dk - 1];
                acc = fmaf(e < dk ? qh[e] : 0.f, kv, acc);
            }
            acc += __shfl_xor(acc, 1, kWave);
            acc += __shfl_xor(acc, 2, kWave);
            if (on && sub == 0) p[idx] = acc / sq;
        }
    }
    __syncthreads();
    {   // softmax over the list, one wave per head (torch.softmax: exp(x - max) / sum)
        const int lane = tid & 63, wave = tid >> 6;
        for (int hh = wave; hh < hw.heads; hh += kHwThreads / 64) {
            float mx = -3.0e38f;
            for (int c = lane; c < nctx; c += 64) mx = fmaxf(mx, p[hh * nctx + c]);
            mx = wave_max(mx);
            float ssum = 0.f;
            for (int c = lane; c < nctx; c += 64) { const float e = expf(p[hh * nctx + c] - mx); p[hh * nctx + c] = e; ssum += e; }
            ssum = wave_sum(ssum);
            for (int c = lane; c < nctx; c += 64) p[hh * nctx + c] = p[hh * nctx + c] / ssum;
        }
    }
    __syncthreads();
    {   // context = attn @ v: the list split four ways over the thread block, partial sums through LDS
        constexpr int SPL = 4;
        const int per = kHwThreads / SPL;                            // 256 columns per pass
        const int sgrp = tid / per, jj = tid - sgrp * per;
        for (int j0 = 0; j0 < D; j0 += per) {
            const int j = j0 + jj;
            const int jc = j < D ? j : D - 1, hh = jc / dk;
            float acc = 0.f;
#pragma unroll 8
            for (int c = sgrp; c < nctx; c += SPL) acc = fmaf(p[hh * nctx + c], Vb[(size_t)c * D + jc], acc);
            if (j < D) part[sgrp * D + j] = acc;
        }
        __syncthreads();
        for (int j = tid; j < D; j += kHwThreads) ctx[j] = (part[j] + part[D + j]) + (part[2 * D + j] + part[3 * D + j]);
    }
    __syncthreads();
    block_gemv_kmajor(hw.o_wt, D, hw.o_b, ctx, D, D, part, o);       // linear_out
    block_layer_norm(o, D, hw.bn_w, hw.bn_b, sv, [&](int j, float v) { pb[j] = v; });
    // combine over cat(pred, pb): two K segments of the k-major [2D][D] weight; q / ctx are free again
    block_gemv_kmajor(hw.c_wt, D, hw.c_b, x, D, D, part, q);
    block_gemv_kmajor(hw.c_wt + (size_t)D * D, D, q, pb, D, D, part, ctx);   // "bias" = the first segment's sums
    block_layer_norm(ctx, D, hw.on_w, hw.on_b, sv, [&](int j, float v) { S.biasT[(size_t)j * d.NLp + n] = v; });
}

// One-entry empty list (the reference's `context_list_empty`, greedy_search.py:328-333): its attention output does not
// depend on the query, so   cold_c = predictor_bias_combine[:, D:] . predictor_bias_bias_norm(linear_out(v_0)) + bias
// is computed once per call.
__global__ __launch_bounds__(kHwThreads) void hw_cold_kernel(HwDev hw)
{
    extern __shared__ float sm[];                   // v[D] o[D] pb[D] | part[16][D]
    __shared__ float sv[kHwThreads / 64];
    const int D = hw.D, tid = threadIdx.x;
    float *v = sm, *o = sm + D, *pb = sm + 2 * D, *part = sm + 3 * D;
    for (int j = tid; j < D; j += kHwThreads) v[j] = hw.vbuf[0][j];
    __syncthreads();
    block_gemv_kmajor(hw.o_wt, D, hw.o_b, v, D, D, part, o);
    block_layer_norm(o, D, hw.bn_w, hw.bn_b, sv, [&](int j, float val) { pb[j] = val; });
    block_gemv_kmajor(hw.c_wt + (size_t)D * D, D, hw.c_b, pb, D, D, part, o);
    for (int j = tid; j < D; j += kHwThreads) {
        hw.cold_c[j] = o[j];
        if (hw.proj_wt != nullptr && hw.Hp <= D) hw.cold_cp[j] = o[j] + (hw.c1p_b[j] - hw.c_b[j]);   // + combine[:, :D] . bp
    }
}

__global__ void greedy_hw_init_kernel(DevState *s)
{
    const Dims &d = s->d;
    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < d.L * d.Hp; i += blockDim.x) {
        const size_t o = (size_t)i * d.NLp + n;
        s->cache_hT[o] = 0.f; s->cache_cT[o] = 0.f; s->new_hT[o] = 0.f; s->new_cT[o] = 0.f;
    }
    if (tid == 0) {
        const int T = s->enc_lens[n] < s->T ? s->enc_lens[n] : s->T;
        s->token[n] = s->blank;
        s->lane_t[n] = 0;
        s->noblk[n] = 0;
        s->need_pred[n] = 1;
        s->new_is_cache[n] = 0;
        s->hyp_lens[n] = 0;
        const int act = T > 0;
        s->lane_active[n] = act;
        if (act) atomicAdd(s->active_count, 1);
        // gate of the first predictor step (frame 0): nothing to go back to yet (greedy_search.py:365-368,386)
        int gate = 1;
        if (act && s->hw_filter && s->gate_tab[(size_t)n * s->T] == 0) gate = 0;
        s->cur_gate[n] = gate;
        s->gb_flag[n] = 0; s->gb_end[n] = -1; s->last_t[n] = 0;
        if (act && s->trace_cap > 0) s->trace[(size_t)n * s->trace_cap] = gate;
        s->trace_len[n] = act ? 1 : 0;
    }
    first_predictor_input(s, n);
}

// ------------------------------------------------------------------ beam --
// 64-bit running hash of a hypothesis' token sequence (kept per hypothesis, extended by one token per emission)
__device__ __forceinline__ unsigned long long hyp_hash_push(unsigned long long h, int tok)
{
    return (h ^ (unsigned long long)(unsigned)(tok + 1)) * 0x100000001b3ull + 0x9e3779b97f4a7c15ull;
}

__global__ void beam_init_kernel(DevState *s)
{
    const Dims &d = s->d;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int b = n / s->beam, j = n % s->beam;
    for (int i = tid; i < d.L * d.Hp; i += blockDim.x) {
        const size_t o = (size_t)i * d.NLp + n;
        s->cache_hT[o] = 0.f; s->cache_cT[o] = 0.f;
        s->new_hT[o] = 0.f; s->new_cT[o] = 0.f;
    }
    if (tid == 0) {
        const int T = s->enc_lens[b] < s->T ? s->enc_lens[b] : s->T;
        s->token[n] = s->blank;
        s->lane_t[n] = 0;
        s->need_pred[n] = 1;
        s->lane_active[n] = (j == 0 && T > 0);
        s->bhyp_lens[(size_t)b * s->beam + j] = (j == 0) ? 1 : 0;
        s->bhyps[((size_t)b * s->beam + j) * s->Lmax] = s->blank;
        s->bhash[n] = hyp_hash_push(0ull, s->blank);
        if (j == 0) {
            s->n_hyps[b] = 1;
            s->frame[b] = 0;
            s->hyp_sel[b] = 0;
            s->bscores[(size_t)b * s->beam] = 0.0;
        }
    }
    first_predictor_input(s, n);
}

// per lane: log-softmax, mixture with the CTC posterior of this frame, top-`beam`.  One 512-thread workgroup per lane
// (round 2: 256 threads with 24 classes each; 1 024 threads were tried too: the transcendentals shrink with the classes
// per thread, the fixed cost of a selection round -- two DPP reductions per wave -- grows with the waves per SIMD).  The row lives in registers; every wave extracts the top-`beam` of its part with
// wave-local argmax rounds (DPP, no barriers), wave 0 merges the 16 x beam survivors.  Ties go to the lowest index
// throughout.
constexpr int kTopkThreads = 512;
template <int NV>
__global__ __launch_bounds__(kTopkThreads) void beam_topk_kernel(DevState *sp)
{
    constexpr int NW = kTopkThreads / 64;
    __shared__ float sv[NW];
    __shared__ float wv[NW * kMaxBeam];
    __shared__ int wi[NW * kMaxBeam];
    WR_STAMP_DECL;
    WR_STAMP_RT(7);
    WR_STAMP(0);
    const DevState S = *sp;
    const Dims &d = S.d;
    const int n = blockIdx.x, tid = threadIdx.x;
    // no branch before the loads: the lane's scalars and both rows are requested in one batch (idle lanes leave
    // after the first reduction)
    const int act = S.lane_active[n];
    const int b = n / S.beam;
    int fr = S.frame[b];
    fr = fr < S.T ? fr : S.T - 1;
    const float *__restrict__ x = S.logits + (size_t)n * d.V;
    const float *__restrict__ cp = S.ctc_logp + ((size_t)b * S.T + fr) * d.V;
    const float ninf = -__builtin_huge_valf();
    float xv[NV], cv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * kTopkThreads;
        xv[i] = x[v < d.V ? v : d.V - 1];
        cv[i] = cp[v < d.V ? v : d.V - 1];
    }
    float m = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NV; ++i) m = fmaxf(m, (tid + i * kTopkThreads < d.V) ? xv[i] : -3.0e38f);
    m = block_max(m, sv);
    WR_STAMP(1);                                   // rows arrived, first reduction done
    if (!act) return;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sum += (tid + i * kTopkThreads < d.V) ? expf(xv[i] - m) : 0.f;
    sum = block_sum(sum, sv);
    const float ls = logf(sum);
    WR_STAMP(2);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float l = (xv[i] - m) - ls;
        // prefix_beam_search.py:99-101: log(tw * exp(logp) + cw * exp(ctc[i]))
        xv[i] = (tid + i * kTopkThreads < d.V) ? logf(S.tr_weight * expf(l) + S.ctc_weight * expf(cv[i])) : ninf;
    }
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < S.beam; ++k) {
        float best = ninf;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (xv[i] > best) { best = xv[i]; bi = tid + i * kTopkThreads; }   // ascending scan: lowest index wins ties
        wave_argmax_dpp(best, bi);
        if (lane == 0) { wv[wave * S.beam + k] = best; wi[wave * S.beam + k] = bi; }
        if (bi != 0x7fffffff && (bi % kTopkThreads) == tid) {                     // taken
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (i == bi / kTopkThreads) xv[i] = ninf;
        }
    }
    // touch the next frame's CTC row (one word per 128-byte line) so that it is already in the Infinity Cache and
    // its translation cached when the next micro-step asks for it
    if (fr + 1 < S.T && tid * 32 < d.V) {
        const float touch = cp[(size_t)d.V + tid * 32];
        asm volatile("" ::"v"(touch));
    }
    WR_STAMP(3);                                   // per-wave selection rounds done
    __syncthreads();
    WR_STAMP(4);
    if (wave == 0) {
        constexpr int PM = NW * kMaxBeam / 64;      // survivors per lane of wave 0
        const int total = NW * S.beam;
        float c0[PM];
        int ci[PM];
#pragma unroll
        for (int q = 0; q < PM; ++q) {
            const int j = lane + q * 64;
            c0[q] = j < total ? wv[j] : ninf;
            ci[q] = j < total ? wi[j] : 0x7fffffff;
        }
        for (int k = 0; k < S.beam; ++k) {
            float best = ninf;
            int bi = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < PM; ++q)
                if (c0[q] > best || (c0[q] == best && ci[q] < bi)) { best = c0[q]; bi = ci[q]; }
            wave_argmax_dpp(best, bi);
            if (lane == 0) {
                S.topv[(size_t)n * S.beam + k] = best;
                S.topi[(size_t)n * S.beam + k] = (bi < d.V) ? bi : 0;
            }
#pragma unroll
            for (int q = 0; q < PM; ++q)
                if (ci[q] == bi) c0[q] = ninf;
        }
    }
    WR_STAMP(5);
    WR_STAMP_DRAIN();
    WR_STAMP(6);
    WR_STAMP_RT(8);
    WR_STAMP_FLUSH(6);
}

__device__ __forceinline__ double log_add2(double a, double b)
{
    // wenet/utils/common.py:268-276 for two arguments, float64
    const double ninf = -__builtin_huge_val();
    if (a == ninf && b == ninf) return ninf;
    // mx + log(exp(a - mx) + exp(b - mx)): the larger argument contributes exp(0.0) = 1.0 exactly, so one exp suffices
    // (the sum is the same two addends in either order)
    const double mx = a > b ? a : b, mn = a > b ? b : a;
    return mx + log(1.0 + exp(mn - mx));
}

// one workgroup per utterance: expansion, prefix fusion, stable prune (prefix_beam_search.py:107-146).
// One candidate (hypothesis j, rank t) per thread; the all-pairs steps (which candidates spell the same token
// sequence, where a class ranks) run pair-per-thread.  Sequences are compared through their running hashes and
// verified token by token only when the hashes agree, so fusion is exact.  Then the survivors' hypotheses,
// predictor caches and next input embeddings are moved into place (rows of the k-major caches are independent,
// so each batch of rows is read into registers, synchronised and written back in place).
constexpr int kBeamUpdThreads = 1024;   // one workgroup per utterance; 256 threads in round 2 (the all-pairs steps and the
                                        // layer-0 cells of the survivors are 4 x as parallel now)
__global__ __launch_bounds__(kBeamUpdThreads) void beam_update_kernel(DevState *sp)
{
    constexpr int NT = kBeamUpdThreads;
    constexpr int MC = kMaxBeam * kMaxBeam;
    __shared__ int c_base[MC], c_tok[MC], c_len[MC], c_last[MC], c_rep[MC], c_rank[MC], order[kMaxBeam];
    __shared__ unsigned long long c_hash[MC];
    __shared__ int e_src[kMaxBeam], e_blank[kMaxBeam], e_tok[kMaxBeam], e_last[kMaxBeam], e_lb[kMaxBeam];
    __shared__ double c_score[MC], f_score[MC];
    __shared__ int s_keep;
    WR_STAMP_DECL;
    WR_STAMP_RT(7);
    WR_STAMP(0);
    const DevState S = *sp;
    const Dims &d = S.d;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int beam = S.beam;
    const int T = S.enc_lens[b] < S.T ? S.enc_lens[b] : S.T;
    const int fr = S.frame[b];
    if (fr >= T) return;
    const int N = S.n_hyps[b];
    const int C = N * beam;
    const int sel = S.hyp_sel[b];
    const size_t hstride = (size_t)S.n_utt * beam * S.Lmax;
    const int32_t *__restrict__ hy = S.bhyps + sel * hstride + (size_t)b * beam * S.Lmax;
    int32_t *__restrict__ hy2 = S.bhyps + (1 - sel) * hstride + (size_t)b * beam * S.Lmax;
    const int32_t *__restrict__ hl = S.bhyp_lens + (size_t)sel * S.n_utt * beam + (size_t)b * beam;
    int32_t *__restrict__ hl2 = S.bhyp_lens + (size_t)(1 - sel) * S.n_utt * beam + (size_t)b * beam;
    // phase 1: candidates in the reference's order (hypothesis-major, then top-k rank; :109-127)
    if (tid < kMaxBeam) e_lb[tid] = tid < N ? hl[tid] : 0;
    if (tid == 0) s_keep = 0;
    if (tid < C) {
        const int j = tid / beam, t = tid - j * beam;
        const int n = b * beam + j;
        const int tok = S.topi[(size_t)n * beam + t];
        // scores tensor is fp32 built from Python floats (:86); add in fp32 (:105); .item() -> float64 (:116)
        const float sj = (float)S.bscores[(size_t)b * beam + j];
        const float tv = S.topv[(size_t)n * beam + t];
        const int lb = hl[j];
        const int last = S.token[n];               // last token of hypothesis j (this frame's predictor input)
        const unsigned long long hh = S.bhash[n];
        c_score[tid] = (double)(sj + tv);
        c_base[tid] = j;
        c_tok[tid] = tok;
        c_len[tid] = lb + (tok != S.blank ? 1 : 0);
        c_last[tid] = (tok != S.blank) ? tok : last;
        c_hash[tid] = (tok != S.blank) ? hyp_hash_push(hh, tok) : hh;
        c_rep[tid] = tid;
        c_rank[tid] = 0;
    }
    __syncthreads();
    WR_STAMP(1);                                   // candidates built
    // phase 2: class representative = the first candidate with the same token sequence (:130-142)
    for (int p = tid; p < C * C; p += NT) {
        const int i = p / C, f = p - i * C;
        if (f >= i || c_hash[f] != c_hash[i] || c_len[f] != c_len[i] || c_last[f] != c_last[i]) continue;
        // equal hashes, lengths and last tokens: the sequences are equal iff the first len-1 tokens of the two base
        // hypotheses are (both bases hold at least that many); compared in batches of independent loads
        const int32_t *__restrict__ pi = hy + (size_t)c_base[i] * S.Lmax;
        const int32_t *__restrict__ pf = hy + (size_t)c_base[f] * S.Lmax;
        const int ncmp = c_len[i] - 1;
        bool same = true;
        constexpr int UC = 16;
        for (int q0 = 0; q0 < ncmp && same; q0 += UC) {
            int xa[UC], ya[UC];
#pragma unroll
            for (int u = 0; u < UC; ++u) {
                const int q = q0 + u < ncmp ? q0 + u : ncmp - 1;
                xa[u] = pi[q];
                ya[u] = pf[q];
            }
#pragma unroll
            for (int u = 0; u < UC; ++u) same = same && (xa[u] == ya[u]);
        }
        if (same) atomicMin(&c_rep[i], f);
    }
    __syncthreads();
    WR_STAMP(2);                                   // classes known
    // phase 3: a representative accumulates its duplicates' scores in candidate order with float64 log_add
    // (a float64 exp + log is ~2 us of dependent instructions and a wave runs it for one lane at a time: consecutive
    // candidates sit in different waves, so that the few duplicates of a frame are summed side by side)
    // Most frames have no duplicate at all: then the fused scores are the candidates' own (one count over the workgroup
    // instead of every candidate scanning all later ones).
    const int n_dup = __syncthreads_count(tid < C && c_rep[tid] != tid);
    if (n_dup == 0) {
        if (tid < C) f_score[tid] = c_score[tid];
    } else {
        const int c = (tid & 63) * (NT / 64) + (tid >> 6);
        if (c < C) {
            double sc = c_score[c];
            if (c_rep[c] == c) {
#pragma unroll 8
                for (int f = c + 1; f < C; ++f)
                    if (c_rep[f] == c) sc = log_add2(sc, c_score[f]);
            }
            f_score[c] = sc;
        }
    }
    __syncthreads();
    WR_STAMP(9);                                   // duplicates' scores summed
    // phase 4: stable descending sort position among representatives (list.sort(reverse=True) keeps order on ties)
    for (int p = tid; p < C * C; p += NT) {
        const int i = p / C, f = p - i * C;
        if (c_rep[i] != i || c_rep[f] != f) continue;
        const double me = f_score[i], ot = f_score[f];
        if (ot > me || (ot == me && f < i)) atomicAdd(&c_rank[i], 1);
    }
    __syncthreads();
    if (tid < C && c_rep[tid] == tid) {
        if (c_rank[tid] < beam) order[c_rank[tid]] = tid;
        atomicAdd(&s_keep, 1);
    }
    __syncthreads();
    const int keep = s_keep < beam ? s_keep : beam;
    if (tid < keep) {
        const int f = order[tid];
        e_src[tid] = c_base[f];
        e_tok[tid] = c_tok[f];
        e_blank[tid] = (c_tok[f] == S.blank);
        e_last[tid] = c_last[f];
        hl2[tid] = c_len[f];
        S.bscores[(size_t)b * beam + tid] = f_score[f];
        S.bhash[b * beam + tid] = c_hash[f];
        S.token[b * beam + tid] = c_last[f];       // next frame's predictor input: last token of the hypothesis (:78-80)
    }
    __syncthreads();
    WR_STAMP(3);                                   // fused, ranked, pruned
    // phase 5: the pruned beam -- hypotheses into the other buffer, (survivor, position) pairs in flight together
    {
        int maxlb = 0;
        for (int e = 0; e < keep; ++e) maxlb = e_lb[e_src[e]] > maxlb ? e_lb[e_src[e]] : maxlb;
        constexpr int UN = 4;
        for (int i0 = 0; i0 < keep * maxlb; i0 += NT * UN) {
            int val[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = i0 + u * NT + tid;
                const int e = i / maxlb, q = i - e * maxlb;
                const bool in = e < keep && q < e_lb[e_src[e < keep ? e : 0]];
                val[u] = in ? hy[(size_t)e_src[e] * S.Lmax + q] : 0;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = i0 + u * NT + tid;
                const int e = i / maxlb, q = i - e * maxlb;
                if (e < keep && q < e_lb[e_src[e]]) hy2[(size_t)e * S.Lmax + q] = val[u];
            }
        }
        if (tid < keep) {
            const int lb = e_lb[e_src[tid]];
            if (!e_blank[tid] && lb < S.Lmax) hy2[(size_t)tid * S.Lmax + lb] = e_tok[tid];
        }
    }
    WR_STAMP(4);                                   // hypotheses moved
    if (d.ptype == kPredLstm) {
        // LSTM predictor: a survivor inherits a slot NUMBER -- its base hypothesis' committed slot for a blank extension,
        // the slot of the base's new state for a label (:111-124; several survivors may share one) -- and gets a free slot
        // of the utterance's 2 * beam for its own next step; then layer 0 of that step (lstm_layer0_cell).  Nothing is
        // copied.  Slots of utterance b: lanes' numbers b * beam + j and NLp + b * beam + j.
        __shared__ int s_comm[kMaxBeam], s_new[kMaxBeam], s_oc[kMaxBeam], s_on[kMaxBeam];
        if (tid < beam) { s_oc[tid] = S.comm_slot[b * beam + tid]; s_on[tid] = S.new_slot[b * beam + tid]; }
        __syncthreads();
        if (tid == 0) {
            unsigned used = 0;                                   // bit j: slot b*beam + j, bit beam + j: slot NLp + b*beam + j
            for (int e = 0; e < keep; ++e) {
                const int sl = e_blank[e] ? s_oc[e_src[e]] : s_on[e_src[e]];
                s_comm[e] = sl;
                used |= 1u << (sl >= d.NLp ? beam + (sl - d.NLp - b * beam) : sl - b * beam);
            }
            // every lane of the utterance, idle ones too, gets a free slot of its own: the recurrent-product jobs write
            // pool_g[new_slot[n]] for all lanes and must never land in a slot that is in use
            int bit = 0;
            for (int e = 0; e < beam; ++e) {
                while (used & (1u << bit)) ++bit;
                s_new[e] = bit < beam ? b * beam + bit : d.NLp + b * beam + (bit - beam);
                if (e >= keep) s_comm[e] = s_new[e];
                ++bit;
            }
        }
        __syncthreads();
        if (tid < beam) { S.comm_slot[b * beam + tid] = s_comm[tid]; S.new_slot[b * beam + tid] = s_new[tid]; }
        // layer 0 of every survivor's next step (lstm_layer0_cell), (survivor, unit) pairs spread over the workgroup with
        // the loads of two pairs per thread in flight together
        {
            constexpr int UN = 2;
            const int items = keep * d.H;
            for (int i0 = 0; i0 < items; i0 += NT * UN) {
                float ev[UN][4], gv[UN][4], cv[UN];
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const int i = i0 + q * NT + tid;
                    const int e = i < items ? i / d.H : 0, u = i < items ? i - e * d.H : 0;
                    const int col = (u >> 3) * 32 + (u & 7);
                    const float *__restrict__ er = S.etab + (size_t)e_last[e] * d.G4p + col;
                    const float *__restrict__ gr = S.pool_g + (size_t)s_comm[e] * d.G4p + col;
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt) { ev[q][gt] = er[gt * 8]; gv[q][gt] = gr[gt * 8]; }
                    cv[q] = S.pool_c[(size_t)s_comm[e] * d.Hp + u];
                }
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const int i = i0 + q * NT + tid;
                    if (i >= items) continue;
                    const int e = i / d.H, u = i - e * d.H;
                    const float ig = sigmoidf_(ev[q][0] + gv[q][0]);
                    const float fg = sigmoidf_(ev[q][1] + gv[q][1]);
                    const float gg = tanhf(ev[q][2] + gv[q][2]);
                    const float og = sigmoidf_(ev[q][3] + gv[q][3]);
                    const float c = fg * cv[q] + ig * gg;
                    S.pool_c[(size_t)s_new[e] * d.Hp + u] = c;
                    S.new_hT[(size_t)u * d.NLp + b * beam + e] = og * tanhf(c);
                }
            }
        }
    } else {
    // next predictor inputs
    {
        constexpr int UN = 4;
        const float *__restrict__ emb = S.embed;
        for (int i0 = 0; i0 < keep * d.D; i0 += NT * UN) {
            float val[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = i0 + u * NT + tid;
                const int e = i / d.D, k = i - e * d.D;
                val[u] = e < keep ? emb[(size_t)e_last[e] * d.D + k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = i0 + u * NT + tid;
                const int e = i / d.D, k = i - e * d.D;
                if (e < keep) S.xT[(size_t)k * d.NLp + b * beam + e] = val[u];
            }
        }
    }
    // predictor caches: a blank extension keeps the base hypothesis' cache, a label takes the predictor's new
    // cache (:111-124).  Rows i of [L*Hp][NLp]; a thread owns (row, survivor) pairs, `beam` survivors per row.
    {
        constexpr int UN = 8;
        const int rows = d.L * d.Hp, per_it = NT / beam;           // rows per pass of the workgroup
        const int e = tid % beam, r0 = tid / beam;
        const bool on = e < keep && r0 < per_it;
        const int src = b * beam + (on ? e_src[e] : 0), dst = b * beam + e;
        const bool bl = on ? e_blank[e] != 0 : true;
        const float *hsrc = bl ? S.cache_hT : S.new_hT;
        const float *csrc = bl ? S.cache_cT : S.new_cT;
        for (int rb = 0; rb < rows; rb += per_it * UN) {
            float hv[UN], cw[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int r = rb + u * per_it + r0;
                const size_t so = (size_t)(r < rows ? r : 0) * d.NLp + src;
                hv[u] = hsrc[so];
                cw[u] = csrc[so];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int r = rb + u * per_it + r0;
                if (on && r < rows) {
                    const size_t dst_o = (size_t)r * d.NLp + dst;
                    S.cache_hT[dst_o] = hv[u];
                    S.cache_cT[dst_o] = cw[u];
                }
            }
        }
    }
    }
    if (tid < beam) {
        const int nfr = fr + 1;
        const int n = b * beam + tid;
        S.lane_active[n] = (tid < keep) && (nfr < T);
        S.lane_t[n] = nfr;
        S.need_pred[n] = 1;
        if (tid == 0) {
            S.n_hyps[b] = keep;
            S.frame[b] = nfr;
            S.hyp_sel[b] = 1 - sel;
        }
    }
    WR_STAMP(5);
    WR_STAMP_DRAIN();
    WR_STAMP(6);
    WR_STAMP_RT(8);
    WR_STAMP_FLUSH(7);
}

__global__ void beam_export_kernel(DevState *s, int32_t *hyps_out, int32_t *lens_out, double *scores_out, int32_t *n_out)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int beam = s->beam;
    const int sel = s->hyp_sel[b];
    const size_t hstride = (size_t)s->n_utt * beam * s->Lmax;
    const int32_t *hy = s->bhyps + sel * hstride + (size_t)b * beam * s->Lmax;
    const int32_t *hl = s->bhyp_lens + (size_t)sel * s->n_utt * beam + (size_t)b * beam;
    const int N = s->n_hyps[b];
    for (int i = tid; i < beam * s->Lmax; i += blockDim.x) {
        const int e = i / s->Lmax, q = i % s->Lmax;
        hyps_out[((size_t)b * beam + e) * s->Lmax + q] = (e < N && q < hl[e]) ? hy[(size_t)e * s->Lmax + q] : -1;
    }
    if (tid < beam) {
        lens_out[(size_t)b * beam + tid] = (tid < N) ? hl[tid] : 0;
        scores_out[(size_t)b * beam + tid] = (tid < N) ? s->bscores[(size_t)b * beam + tid] : -__builtin_huge_val();
    }
    if (tid == 0) n_out[b] = N;
}

}  // namespace
}  // namespace wr

using namespace wr;

// ------------------------------------------------------------ host handle --
struct wr_decoder {
    Dims d;
    DevState host;                // host copy of the device state block
    DevState *dev;                // device copy (first bytes of the workspace)
    char *ws;
    size_t ws_bytes;
    int max_utt, Tmax, max_hyp, max_beam;
    size_t zero_range[2];
    int32_t *h_active;            // pinned host word
    DevState *h_stage;            // pinned staging slots for the state block (kStageSlots), so that the upload is a
    hipEvent_t stage_ev[4];       // true async copy: a slot is reused only after the copy that read it has completed
    int stage_next;
    hipStream_t work;             // decode work runs here (graph capture is illegal on the legacy default stream)
    hipEvent_t ev_in, ev_out;     // ordering against the caller's stream, no device-wide sync
    hipGraphExec_t greedy_graph[kMaxLook + 1];   // one per look-ahead setting
    int greedy_graph_key[kMaxLook + 1];
    hipGraphExec_t beam_graph;
    int beam_graph_lanes;
    int stream_lanes;             // lanes whose streaming state (cache, token, flags) is live; -1: none
    int look;                     // greedy look-ahead: frames per micro-step (1..kMaxLook), 0: chosen per replay
    // hot-word greedy (wr_decoder_attach_hotword)
    bool hw_attached;
    HwDev hw;
    float *hw_ep2;                // [2][max_utt * Tmax * J] enc_ffn of the empty-list (0) / hot-word (1) encoder stream
    int32_t *hw_state;            // cur_gate | gb_flag | gb_end | last_t, NLp each
    float *hw_biasT;              // [Pp][NLp]
    hipGraphExec_t hw_graph;
    int hw_graph_key;
    bool use_graph;               // greedy micro-steps replayed from a hipGraph (default on)
    bool use_graph_beam;          // beam frames: plain launches measured faster (no host polling to amortise), default off
};

namespace {

struct Carver {
    char *base;
    size_t off;
    template <typename T>
    T *take(size_t n)
    {
        off = align_up(off, 256);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

size_t carve(const wr_transducer_weights *w, int max_lanes, int max_utt, int Tmax, int max_hyp, int max_beam, char *base,
             DevState *st, DevState **dev, size_t *zero_range = nullptr)
{
    Carver c{base, 0};
    Dims d;
    d.V = w->vocab_size; d.E = w->enc_dim; d.P = w->pred_dim; d.D = w->embed_dim; d.H = w->hidden; d.L = w->n_layers;
    d.J = w->join_dim;
    d.Ve = w->embed_rows > 0 ? w->embed_rows : w->vocab_size;
    d.act = w->activation;
    d.ptype = w->predictor_type; d.ctx = w->context_size; d.heads = w->n_head; d.pact = w->pred_activation;
    d.ln_eps = w->ln_eps;
    auto up = [](int x, int m) { return (x + m - 1) / m * m; };
    d.Dp = up(d.D, 16); d.Hp = up(d.H, 16); d.Pp = up(d.P, 16); d.Jp = up(d.J, 16);   // K: 8 waves x k-pairs
    d.G4p = 4 * d.Hp;                                      // 32-column tiles = i,f,g,o of 8 hidden units
    d.Vp = up(d.V, 256);
    d.NL = max_lanes;
    d.NLp = max_lanes <= 32 ? 32 : up(max_lanes, 64);     // lane tiles of the GEMM: 32, or pairs of 32
    DevState s;
    memset(&s, 0, sizeof(s));
    s.d = d;
    DevState *devp = c.take<DevState>(1);
    const size_t zero_begin = align_up(c.off, 256);       // everything from here to zero_end is zero-filled at create
    for (int l = 0; l < d.L && d.ptype == kPredLstm; ++l) {
        if (l > 0) s.wt_ih[l] = c.take<float>((size_t)d.Hp * d.G4p);     // layer 0's input product is the table `etab`
        s.wt_hh[l] = c.take<float>((size_t)d.Hp * d.G4p);
        s.bsum[l] = c.take<float>((size_t)d.G4p);
    }
    const int Pn = up(d.P, 32), Jn = up(d.J, 32);
    if (d.ptype == kPredLstm) {
        s.proj_wt = c.take<float>((size_t)d.Hp * Pn);
        s.projffn_wt = c.take<float>((size_t)d.Hp * Jn);
        s.projffn_b = c.take<float>((size_t)Jn);
    }
    if (d.ptype == kPredEmbedding) {
        s.pffn_wt = c.take<float>((size_t)d.Dp * up(d.D, 32));
        s.combT = c.take<float>((size_t)d.Dp * d.NLp);
        s.ffnT = c.take<float>((size_t)d.Dp * d.NLp);
    }
    s.predffn_wt = c.take<float>((size_t)d.Pp * Jn);
    s.encffn_wt = c.take<float>((size_t)d.E * d.J);
    s.out_wt = c.take<float>((size_t)d.Jp * d.Vp);
    s.xT = c.take<float>((size_t)d.Dp * d.NLp);
    const size_t cs = (size_t)d.L * d.Hp * d.NLp;
    s.cache_hT = c.take<float>(cs); s.cache_cT = c.take<float>(cs);
    s.new_hT = c.take<float>(cs); s.new_cT = c.take<float>(cs);
    s.outT = c.take<float>((size_t)d.Pp * d.NLp);
    s.ht = c.take<float>((size_t)d.Jp * kMaxLook * d.NLp);
    s.comm_slot = c.take<int32_t>(d.NLp); s.new_slot = c.take<int32_t>(d.NLp);
    if (d.ptype == kPredLstm) {
        s.pool_c = c.take<float>((size_t)d.L * 2 * d.NLp * d.Hp);
        s.pool_g = c.take<float>((size_t)d.L * 2 * d.NLp * d.G4p);
    }
    const size_t zero_end = align_up(c.off, 256);
    if (d.ptype == kPredLstm) s.etab = c.take<float>((size_t)d.Ve * d.G4p);
    s.ep_all = c.take<float>((size_t)max_utt * Tmax * d.J);
    s.token = c.take<int32_t>(d.NLp); s.lane_t = c.take<int32_t>(d.NLp); s.noblk = c.take<int32_t>(d.NLp);
    s.need_pred = c.take<int32_t>(d.NLp); s.lane_active = c.take<int32_t>(d.NLp);
    s.new_is_cache = c.take<int32_t>(d.NLp);
    s.logits = c.take<float>((size_t)kMaxLook * d.NLp * d.V);
    s.row_tok = c.take<int32_t>((size_t)kMaxLook * d.NLp);
    s.n_cb = d.Vp / 32;
    s.row_part = c.take<float4>((size_t)d.NLp * s.n_cb);
    s.active_count = c.take<int32_t>(64);
    s.topv = c.take<float>((size_t)d.NL * kMaxBeam); s.topi = c.take<int32_t>((size_t)d.NL * kMaxBeam);
    s.n_hyps = c.take<int32_t>(max_utt); s.frame = c.take<int32_t>(max_utt);
    s.hyp_sel = c.take<int32_t>(max_utt);
    const int Lmax = Tmax + 1;
    s.bhyps = c.take<int32_t>((size_t)2 * max_utt * max_beam * Lmax);
    s.bhyp_lens = c.take<int32_t>((size_t)2 * max_utt * max_beam);
    s.bscores = c.take<double>((size_t)max_utt * max_beam);
    s.bhash = c.take<unsigned long long>((size_t)max_utt * max_beam);
    s.Lmax = Lmax;
    s.max_hyp = max_hyp;
    if (zero_range) { zero_range[0] = zero_begin; zero_range[1] = zero_end; }
    if (st) *st = s;
    if (dev) *dev = devp;
    return align_up(c.off, 256);
}

int check_weights(const wr_transducer_weights *w)
{
    WR_REQUIRE(w != nullptr, WR_EINVAL, "decoder: weights is null");
    WR_REQUIRE(w->vocab_size > 1 && w->enc_dim > 0 && w->pred_dim > 0 && w->embed_dim > 0 && w->hidden > 0 && w->join_dim > 0,
               WR_EINVAL, "decoder: non-positive dimension");
    WR_REQUIRE(w->n_layers >= 1 && w->n_layers <= kMaxLayers, WR_EUNSUPPORTED, "decoder: n_layers=%d (max %d)", w->n_layers,
               kMaxLayers);
    WR_REQUIRE(w->activation >= WR_ACT_TANH && w->activation <= WR_ACT_GELU, WR_EINVAL,
               "decoder: activation code %d is not a wr_activation", w->activation);
    WR_REQUIRE(w->join_dim <= 1024 && w->pred_dim <= 1024 && w->hidden <= 1024 && w->embed_dim <= 1024 && w->enc_dim <= 1024,
               WR_EUNSUPPORTED, "decoder: a layer dimension exceeds 1024");
    WR_REQUIRE(w->vocab_size * sizeof(float) <= 64 * 1024, WR_EUNSUPPORTED, "decoder: vocab_size=%d exceeds 16384",
               w->vocab_size);
    WR_REQUIRE(w->embed && w->enc_ffn_w && w->enc_ffn_b && w->pred_ffn_w && w->pred_ffn_b && w->out_w && w->out_b, WR_EINVAL,
               "decoder: null weight pointer");
    WR_REQUIRE(w->predictor_type >= kPredLstm && w->predictor_type <= kPredConv, WR_EINVAL, "decoder: predictor_type %d",
               w->predictor_type);
    if (w->predictor_type == kPredLstm) {
        WR_REQUIRE(w->proj_w && w->proj_b, WR_EINVAL, "decoder: null weight pointer");
        for (int l = 0; l < w->n_layers; ++l)
            WR_REQUIRE(w->w_ih[l] && w->w_hh[l] && w->b_ih[l] && w->b_hh[l], WR_EINVAL, "decoder: null LSTM weight (layer %d)", l);
        return WR_OK;
    }
    // stateless predictors: the token history rides in the LSTM-state slots
    WR_REQUIRE(w->context_size >= 2 && w->context_size <= kMaxCtx && w->n_layers == w->context_size - 1, WR_EUNSUPPORTED,
               "decoder: context_size=%d (history_size + 1) must be 2..%d with n_layers = context_size - 1 (got %d)",
               w->context_size, kMaxCtx, w->n_layers);
    WR_REQUIRE(w->hidden == w->embed_dim && w->pred_dim == w->embed_dim, WR_EINVAL,
               "decoder: a stateless predictor has hidden = pred_dim = embed_dim");
    WR_REQUIRE(w->pred_activation >= WR_ACT_TANH && w->pred_activation <= WR_ACT_GELU, WR_EINVAL,
               "decoder: pred_activation code %d is not a wr_activation", w->pred_activation);
    WR_REQUIRE(w->ln_eps > 0.f && w->norm_w && w->norm_b, WR_EINVAL, "decoder: LayerNorm weights / epsilon missing");
    if (w->predictor_type == kPredEmbedding) {
        WR_REQUIRE(w->n_head >= 1 && w->n_head * w->context_size <= 64, WR_EUNSUPPORTED,
                   "decoder: n_head * context_size = %d exceeds 64", w->n_head * w->context_size);
        WR_REQUIRE(w->pos_w && w->ffn_w && w->ffn_b, WR_EINVAL, "decoder: null EmbeddingPredictor weight");
    } else {
        WR_REQUIRE(w->conv_w, WR_EINVAL, "decoder: null ConvPredictor weight");
    }
    return WR_OK;
}

void launch_transpose(const float *src, int R, int C, int Rp, int Cp, float *dst, hipStream_t st)
{
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((Rp + 31) / 32, (Cp + 31) / 32), dim3(256), 0, st, src, R, C, Rp, Cp, dst);
}

// a 128-lane workgroup reads A columns lane0 .. lane0 + 127: the k-major arrays must be that wide
inline bool d_nlp_lt128(const GemmArgs &g) { return (g.lda % 128) != 0; }

// Exact-fp32 MFMA runs at 256 flop/clk/CU, so a 64-lane x 32-column x 512-deep tile already costs 3.4 us on its
// CU: lane tiles of 32 (more, smaller workgroups) whenever they all fit on the chip at once.
template <int EPI>
void launch_gemm(const GemmArgs &g, int n_cols_padded, int n_lanes, hipStream_t st)
{
    const int mt = (n_lanes + 31) / 32, ct = n_cols_padded / 32;
    const int policy = tune_get(kTuneLaneGemmTile);               // 0: by occupancy, 1: 32-lane tiles, 2: 64-lane tiles
    if (mt == 1 || policy == 1 || (policy == 0 && ct * mt <= 256)) {
        hipLaunchKernelGGL((lane_gemm_kernel<1, EPI>), dim3(ct, mt), dim3(64 * kGemmWaves), 0, st, g);
    } else if (policy == 2 || (policy == 0 && ct * ((mt + 1) / 2) <= 256) || !(EPI == kEpiRowMajor || EPI == kEpiRowStats) || d_nlp_lt128(g)) {
        hipLaunchKernelGGL((lane_gemm_kernel<2, EPI>), dim3(ct, (mt + 1) / 2), dim3(64 * kGemmWaves), 0, st, g);
    } else {
        // more tiles than one round of 64-lane workgroups (bit-identical results in every form): 128-lane workgroups
        // (knob 6 = 3: 4-wave workgroups of 32 lanes, four per CU; 4: one wave per tile)
        if constexpr (EPI == kEpiRowMajor || EPI == kEpiRowStats) {
            if (policy == 3)
                hipLaunchKernelGGL((lane_gemm4_kernel<EPI>), dim3(ct, mt), dim3(64 * kGemmWaves / 2), 0, st, g);
            else if (policy == 4 && EPI == kEpiRowMajor)
                hipLaunchKernelGGL((lane_gemm1_kernel<kEpiRowMajor>), dim3(ct, mt), dim3(64), 0, st, g);
            else
                hipLaunchKernelGGL((lane_gemm_kernel<4, EPI>), dim3(ct, (mt + 3) / 4), dim3(64 * kGemmWaves), 0, st, g);
        }
    }
}

// two GEMMs over the same lanes in one launch (lane_gemm_pair_kernel)
template <int EPI0, int EPI1>
void launch_gemm_pair(const GemmArgs &g0, int cols0, const GemmArgs &g1, int cols1, int n_lanes, hipStream_t st)
{
    const int mt = (n_lanes + 31) / 32, ct0 = cols0 / 32, ct = ct0 + cols1 / 32;
    const int policy = tune_get(kTuneLaneGemmTile);
    if (mt == 1 || policy == 1 || (policy == 0 && ct * mt <= 256)) {
        hipLaunchKernelGGL((lane_gemm_pair_kernel<1, EPI0, EPI1>), dim3(ct, mt), dim3(64 * kGemmWaves), 0, st, g0, g1, ct0);
    } else {
        hipLaunchKernelGGL((lane_gemm_pair_kernel<2, EPI0, EPI1>), dim3(ct, (mt + 1) / 2), dim3(64 * kGemmWaves), 0, st, g0, g1, ct0);
    }
}

// pool_g[l][new_slot[n]] = W_hh_l . new_h_l[n] + b_l: the recurrent product of the state a lane's step has just produced,
// for whichever later step starts from that state (see "LSTM predictor step").  Rides in the launch of the next stage.
GemmArgs recurrent_job(const wr_decoder *h, int l, int n_lanes, const int32_t *slot_idx)
{
    const Dims &d = h->d;
    const DevState &s = h->host;
    GemmArgs g{};
    g.A0 = s.new_hT + (size_t)l * d.Hp * d.NLp; g.B0 = s.wt_hh[l]; g.K0 = d.Hp;
    g.lda = d.NLp; g.ldb = d.G4p; g.bias = s.bsum[l];
    g.C = s.pool_g + (size_t)l * 2 * d.NLp * d.G4p; g.ldc = d.G4p; g.N = d.G4p; g.n_lanes = n_lanes;
    g.slot_idx = slot_idx;
#ifdef WR_STAMPS
    g.dbg_slot = -1;
#endif
    return g;
}

// predictor step (predicated per lane).  LSTM: layer 0 was evaluated by the kernel that decided the token (or by the init
// kernel); here layers 1 .. L-1, each one GEMM of depth H with the recurrent product of the layer below in the same
// launch, and -- unless the caller folds it into pred_ffn -- the projection (with the last layer's recurrent product).
// Returns true if the last layer's recurrent product is still to be launched (the caller pairs it with its next GEMM).
bool launch_predictor(wr_decoder *h, int n_lanes, hipStream_t st, bool with_projection = true)
{
    const Dims &d = h->d;
    const DevState &s = h->host;
    auto up = [](int x, int m) { return (x + m - 1) / m * m; };
    const size_t ls = (size_t)d.Hp * d.NLp;
    if (d.ptype == kPredEmbedding) {            // context weighting -> ffn (lane GEMM) -> LayerNorm + activation
        hipLaunchKernelGGL(ctx_predictor_kernel<0>, dim3(n_lanes), dim3(kCtxThreads), 0, st, h->dev);
        GemmArgs g{};
        g.A0 = s.combT; g.B0 = s.pffn_wt; g.K0 = d.Dp;
        g.lda = d.NLp; g.ldb = up(d.D, 32); g.bias = s.pffn_b; g.C = s.ffnT; g.ldc = d.NLp; g.N = d.D; g.n_lanes = n_lanes;
#ifdef WR_STAMPS
        g.dbg_slot = -1;
#endif
        launch_gemm<kEpiKMajor>(g, up(d.D, 32), n_lanes, st);
        hipLaunchKernelGGL(ctx_predictor_kernel<1>, dim3(n_lanes), dim3(kCtxThreads), 0, st, h->dev);
        return false;
    }
    if (d.ptype == kPredConv) {
        hipLaunchKernelGGL(ctx_predictor_kernel<2>, dim3(n_lanes), dim3(kCtxThreads), 0, st, h->dev);
        return false;
    }
    for (int l = 1; l < d.L; ++l) {
        GemmArgs g{};
        g.A0 = s.new_hT + (size_t)(l - 1) * ls; g.B0 = s.wt_ih[l]; g.K0 = d.Hp;
        g.lda = d.NLp; g.ldb = d.G4p; g.N = d.G4p; g.n_lanes = n_lanes;
        g.lane_active = s.lane_active; g.need_pred = s.need_pred; g.H = d.H; g.Hp = d.Hp; g.G4p = d.G4p;
        g.pool_g = s.pool_g + (size_t)l * 2 * d.NLp * d.G4p; g.pool_c = s.pool_c + (size_t)l * 2 * d.NLp * d.Hp;
        g.comm_slot = s.comm_slot; g.new_slot = s.new_slot; g.new_hT = s.new_hT + (size_t)l * ls;
#ifdef WR_STAMPS
        g.dbg_slot = l < 2 ? l : -1;
#endif
        launch_gemm_pair<kEpiLstmCell, kEpiSlotRow>(g, d.G4p, recurrent_job(h, l - 1, n_lanes, s.new_slot), d.G4p, n_lanes, st);
    }
    if (!with_projection) return true;            // the caller applies the composed projection + pred_ffn
    GemmArgs g{};
    g.A0 = s.new_hT + (size_t)(d.L - 1) * ls; g.B0 = s.proj_wt; g.K0 = d.Hp;
    g.lda = d.NLp; g.ldb = up(d.P, 32); g.bias = s.proj_b; g.C = s.outT; g.ldc = d.NLp; g.N = d.P; g.n_lanes = n_lanes;
#ifdef WR_STAMPS
    g.dbg_slot = 8;
#endif
    launch_gemm_pair<kEpiKMajor, kEpiSlotRow>(g, up(d.P, 32), recurrent_job(h, d.L - 1, n_lanes, s.new_slot), d.G4p, n_lanes, st);
    return false;
}

void launch_predictor_and_joint(wr_decoder *h, int n_lanes, hipStream_t st, int look = 1, bool row_stats = false)
{
    const Dims &d = h->d;
    const DevState &s = h->host;
    auto up = [](int x, int m) { return (x + m - 1) / m * m; };
    // LSTM predictor: the projection is folded into pred_ffn (one launch less per micro-step; wr_tune_set(11, 1) keeps
    // them apart).  The hot-word search and wr_predictor_step need the projected output itself and keep the two stages.
    const bool fold = d.ptype == kPredLstm && tune_get(kTuneFoldProj) != 1;
    const bool recurrent_pending = launch_predictor(h, n_lanes, st, !fold);
    {   // pred_ffn with the joiner activation as epilogue
        GemmArgs g{};
        if (fold) {
            g.A0 = s.new_hT + (size_t)(d.L - 1) * d.Hp * d.NLp; g.B0 = s.projffn_wt; g.K0 = d.Hp; g.bias = s.projffn_b;
        } else {
            g.A0 = s.outT; g.B0 = s.predffn_wt; g.K0 = d.Pp; g.bias = s.predffn_b;
        }
        g.lda = d.NLp; g.ldb = up(d.J, 32); g.C = s.ht; g.ldc = kMaxLook * d.NLp; g.N = d.J; g.n_lanes = n_lanes;
        g.st = h->dev; g.lane_active = s.lane_active; g.lane_t = s.lane_t; g.ep_all = s.ep_all; g.J = d.J;
        g.look = look; g.lane_stride = d.NLp; g.act = d.act;
#ifdef WR_STAMPS
        g.dbg_slot = 2;
#endif
        if (recurrent_pending)
            launch_gemm_pair<kEpiJointAct, kEpiSlotRow>(g, up(d.J, 32), recurrent_job(h, d.L - 1, n_lanes, s.new_slot), d.G4p, n_lanes, st);
        else
            launch_gemm<kEpiJointAct>(g, up(d.J, 32), n_lanes, st);
    }
    GemmArgs g{};
#ifdef WR_STAMPS
    g.dbg_slot = 3;
#endif
    // the joiner output for every (frame, lane) column of ht: rows f * NLp + n of the logits
    const int rows = (look - 1) * d.NLp + n_lanes;
    g.A0 = s.ht; g.B0 = s.out_wt; g.K0 = d.Jp;
    g.lda = kMaxLook * d.NLp; g.ldb = d.Vp; g.bias = s.out_b; g.C = s.logits; g.ldc = d.V; g.N = d.V; g.n_lanes = rows;
    if (row_stats) {                               // greedy, one frame per micro-step: the update kernel reads the records
        g.row_part = s.row_part; g.row_part_ld = s.n_cb;
        launch_gemm<kEpiRowStats>(g, d.Vp, rows, st);
    } else {
        launch_gemm<kEpiRowMajor>(g, d.Vp, rows, st);
    }
}

}  // namespace

extern "C" size_t wr_decoder_workspace_bytes(const wr_transducer_weights *w, int max_lanes, int max_utt, int Tmax,
                                             int max_hyp, int max_beam)
{
    if (!w || max_lanes <= 0 || max_utt <= 0 || Tmax <= 0) return 0;
    return carve(w, max_lanes, max_utt, Tmax, max_hyp, max_beam > 0 ? max_beam : 1, nullptr, nullptr, nullptr);
}

extern "C" int wr_decoder_create(const wr_transducer_weights *w, int max_lanes, int max_utt, int Tmax, int max_hyp,
                                 int max_beam, void *workspace_d, size_t workspace_bytes, void *stream, wr_decoder **out)
{
    if (int rc = check_weights(w)) return rc;
    WR_REQUIRE(out && workspace_d, WR_EINVAL, "decoder_create: null pointer argument");
    WR_REQUIRE(max_lanes > 0 && max_lanes <= kMaxLanes, WR_EUNSUPPORTED, "decoder_create: max_lanes=%d (1..%d)", max_lanes,
               kMaxLanes);
    WR_REQUIRE(max_utt > 0 && max_utt <= max_lanes && Tmax > 0 && max_hyp >= 0, WR_EINVAL, "decoder_create: bad sizes");
    if (max_beam <= 0) max_beam = 1;
    WR_REQUIRE(max_beam <= kMaxBeam, WR_EUNSUPPORTED, "decoder_create: beam %d exceeds %d", max_beam, kMaxBeam);
    const size_t need = carve(w, max_lanes, max_utt, Tmax, max_hyp, max_beam, nullptr, nullptr, nullptr);
    WR_REQUIRE(workspace_bytes >= need, WR_EWORKSPACE, "decoder_create: workspace %zu < required %zu", workspace_bytes, need);
    wr_decoder *h = new (std::nothrow) wr_decoder();
    WR_REQUIRE(h != nullptr, WR_EINVAL, "decoder_create: out of host memory");
    h->ws = static_cast<char *>(workspace_d);
    h->ws_bytes = workspace_bytes;
    carve(w, max_lanes, max_utt, Tmax, max_hyp, max_beam, h->ws, &h->host, &h->dev, h->zero_range);
    h->d = h->host.d;
    h->max_utt = max_utt; h->Tmax = Tmax; h->max_hyp = max_hyp; h->max_beam = max_beam;
    for (int i = 0; i <= kMaxLook; ++i) { h->greedy_graph[i] = nullptr; h->greedy_graph_key[i] = -1; }
    h->beam_graph = nullptr; h->beam_graph_lanes = -1;
    h->hw_attached = false; h->hw_graph = nullptr; h->hw_graph_key = -1;
    h->use_graph = true;
    h->use_graph_beam = false;
    h->look = 0;
    h->stream_lanes = -1;
    h->h_active = nullptr;
    h->h_stage = nullptr;
    h->stage_next = 0;
    if (hipHostMalloc(reinterpret_cast<void **>(&h->h_active), 64, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&h->h_stage), kStageSlots * sizeof(DevState), hipHostMallocDefault) != hipSuccess) {
        if (h->h_active) (void)hipHostFree(h->h_active);
        delete h;
        set_error("decoder_create: hipHostMalloc failed");
        return WR_ELAUNCH;
    }
    bool ev_ok = true;
    for (int i = 0; i < kStageSlots; ++i) ev_ok = ev_ok && hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ev_ok || hipStreamCreateWithFlags(&h->work, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess) {
        (void)hipHostFree(h->h_active);
        (void)hipHostFree(h->h_stage);
        delete h;
        set_error("decoder_create: stream/event creation failed");
        return WR_ELAUNCH;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Dims &d = h->d;
    DevState &s = h->host;
    s.embed = w->embed; s.proj_b = w->proj_b; s.predffn_b = w->pred_ffn_b; s.encffn_b = w->enc_ffn_b; s.out_b = w->out_b;
    (void)hipMemsetAsync(h->ws + h->zero_range[0], 0, h->zero_range[1] - h->zero_range[0], st);
    auto up = [](int x, int m) { return (x + m - 1) / m * m; };
    s.pos_w = w->pos_w; s.pffn_b = w->ffn_b; s.pnorm_w = w->norm_w; s.pnorm_b = w->norm_b; s.conv_w = w->conv_w;
    s.conv_b = w->conv_b;
    if (d.ptype == kPredEmbedding) launch_transpose(w->ffn_w, d.D, d.D, up(d.D, 32), d.Dp, s.pffn_wt, st);
    for (int l = 0; l < d.L && d.ptype == kPredLstm; ++l) {
        if (l > 0)
            hipLaunchKernelGGL(lstm_weight_prep_kernel, dim3(256), dim3(256), 0, st, w->w_ih[l], d.H, d.Hp, d.H, d.Hp, s.wt_ih[l]);
        else
            hipLaunchKernelGGL(lstm_etab_kernel, dim3(2048), dim3(256), 0, st, w->embed, w->w_ih[0], d.Ve, d.D, d.H, d.Hp, s.etab);
        hipLaunchKernelGGL(lstm_weight_prep_kernel, dim3(256), dim3(256), 0, st, w->w_hh[l], d.H, d.Hp, d.H, d.Hp, s.wt_hh[l]);
        hipLaunchKernelGGL(lstm_bias_prep_kernel, dim3((d.G4p + 255) / 256), dim3(256), 0, st, w->b_ih[l], w->b_hh[l], d.H, d.Hp,
                           s.bsum[l]);
    }
    if (d.ptype == kPredLstm) {
        launch_transpose(w->proj_w, d.P, d.H, up(d.P, 32), d.Hp, s.proj_wt, st);
        const long n = (long)d.H * d.J + d.J;
        hipLaunchKernelGGL(compose_proj_ffn_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w->pred_ffn_w, w->pred_ffn_b,
                           w->proj_w, w->proj_b, d.J, d.P, d.H, up(d.J, 32), s.projffn_wt, s.projffn_b);
    }
    launch_transpose(w->pred_ffn_w, d.J, d.P, up(d.J, 32), d.Pp, s.predffn_wt, st);
    launch_transpose(w->enc_ffn_w, d.J, d.E, d.J, d.E, s.encffn_wt, st);
    launch_transpose(w->out_w, d.V, d.J, d.Vp, d.Jp, s.out_wt, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        (void)wr_decoder_destroy(h);
        set_error("decoder_create: weight preparation failed: %s", hipGetErrorString(e));
        return WR_ELAUNCH;
    }
    *out = h;
    return WR_OK;
}

extern "C" int wr_decoder_destroy(wr_decoder *h)
{
    if (!h) return WR_OK;
    for (int i = 0; i <= kMaxLook; ++i)
        if (h->greedy_graph[i]) (void)hipGraphExecDestroy(h->greedy_graph[i]);
    if (h->beam_graph) (void)hipGraphExecDestroy(h->beam_graph);
    if (h->hw_graph) (void)hipGraphExecDestroy(h->hw_graph);
    (void)hipStreamSynchronize(h->work);
    if (h->h_active) (void)hipHostFree(h->h_active);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    for (int i = 0; i < kStageSlots; ++i) (void)hipEventDestroy(h->stage_ev[i]);
    (void)hipEventDestroy(h->ev_in);
    (void)hipEventDestroy(h->ev_out);
    (void)hipStreamDestroy(h->work);
    delete h;
    return WR_OK;
}

extern "C" int wr_decoder_set_graph(wr_decoder *h, int enable)
{
    WR_REQUIRE(h != nullptr, WR_EINVAL, "decoder_set_graph: null handle");
    h->use_graph = h->use_graph_beam = enable != 0;
    return WR_OK;
}

extern "C" int wr_decoder_set_lookahead(wr_decoder *h, int frames)
{
    WR_REQUIRE(h != nullptr, WR_EINVAL, "decoder_set_lookahead: null handle");
    WR_REQUIRE(frames >= 0 && frames <= kMaxLook, WR_EINVAL, "decoder_set_lookahead: frames=%d (0..%d)", frames, kMaxLook);
    h->look = frames;
    return WR_OK;
}

namespace {

int upload_state(wr_decoder *h, hipStream_t st)
{
    // through a pinned slot: a copy from pageable memory would block the host until the work stream has drained
    const int slot = h->stage_next;
    h->stage_next = (slot + 1) % kStageSlots;
    (void)hipEventSynchronize(h->stage_ev[slot]);                  // the copy that last read this slot (long done, normally)
    h->h_stage[slot] = h->host;
    hipError_t e = hipMemcpyAsync(h->dev, &h->h_stage[slot], sizeof(DevState), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(h->stage_ev[slot], st);
    if (e != hipSuccess) { set_error("decoder: state upload failed: %s", hipGetErrorString(e)); return WR_ELAUNCH; }
    return WR_OK;
}

LaneArgs lane_args(const wr_decoder *h)
{
    const DevState &s = h->host;
    LaneArgs a;
    a.lane_active = s.lane_active; a.lane_t = s.lane_t; a.noblk = s.noblk; a.need_pred = s.need_pred;
    a.new_is_cache = s.new_is_cache; a.comm_slot = s.comm_slot; a.new_slot = s.new_slot;
    a.row_part = s.row_part; a.n_cb = s.n_cb; a.V = h->d.V;
    return a;
}

template <bool HW>
void launch_greedy_update(wr_decoder *h, int n_lanes, hipStream_t st)
{
    const LaneArgs a = lane_args(h);
    if ((h->d.V + 31) / 32 <= 3 * 64)              // records per lane of a wave: V <= 6144 -> 3, else 8 (V <= 16384)
        hipLaunchKernelGGL((greedy_update_kernel<HW, 3>), dim3(n_lanes), dim3(256), 0, st, h->dev, a);
    else
        hipLaunchKernelGGL((greedy_update_kernel<HW, 8>), dim3(n_lanes), dim3(256), 0, st, h->dev, a);
}

void greedy_micro_step(wr_decoder *h, int n_lanes, hipStream_t st, int look)
{
    launch_predictor_and_joint(h, n_lanes, st, look, look == 1);
    const int V = h->d.V;                           // <= 16384 (check_weights)
    if (look == 1) {
        launch_greedy_update<false>(h, n_lanes, st);
        return;
    }
    const dim3 grid(n_lanes, look);
    if (V <= 256 * 8) hipLaunchKernelGGL(greedy_rows_kernel<8>, grid, dim3(256), 0, st, h->dev);
    else if (V <= 256 * 24) hipLaunchKernelGGL(greedy_rows_kernel<24>, grid, dim3(256), 0, st, h->dev);
    else hipLaunchKernelGGL(greedy_rows_kernel<64>, grid, dim3(256), 0, st, h->dev);
    hipLaunchKernelGGL(greedy_resolve_kernel, dim3(n_lanes), dim3(256), 0, st, h->dev, look);
}

void beam_frame(wr_decoder *h, int n_lanes, int n_utt, hipStream_t st)
{
    launch_predictor_and_joint(h, n_lanes, st);
    const int V = h->d.V;                           // <= 16384 (check_weights)
    if (V <= kTopkThreads * 4) hipLaunchKernelGGL(beam_topk_kernel<4>, dim3(n_lanes), dim3(kTopkThreads), 0, st, h->dev);
    else if (V <= kTopkThreads * 10) hipLaunchKernelGGL(beam_topk_kernel<10>, dim3(n_lanes), dim3(kTopkThreads), 0, st, h->dev);
    else hipLaunchKernelGGL(beam_topk_kernel<32>, dim3(n_lanes), dim3(kTopkThreads), 0, st, h->dev);
    hipLaunchKernelGGL(beam_update_kernel, dim3(n_utt), dim3(kBeamUpdThreads), 0, st, h->dev);
}

// Order the decoder's work stream after everything already enqueued on the caller's stream ...
hipStream_t enter(wr_decoder *h, hipStream_t caller)
{
    (void)hipEventRecord(h->ev_in, caller);
    (void)hipStreamWaitEvent(h->work, h->ev_in, 0);
    return h->work;
}
// ... and the caller's stream after the decoder's work.
void leave(wr_decoder *h, hipStream_t caller)
{
    (void)hipEventRecord(h->ev_out, h->work);
    (void)hipStreamWaitEvent(caller, h->ev_out, 0);
}

// Scope of one entry point on the work stream: leave() runs on EVERY exit path, so the caller's stream is always
// ordered after whatever was enqueued; unless ok() was called, the handle's streaming state and cached graphs are
// dropped as well (a failed call leaves the lanes in an undefined state).
struct WorkScope {
    wr_decoder *h;
    hipStream_t caller, st;
    bool good = false;
    WorkScope(wr_decoder *h_, hipStream_t caller_) : h(h_), caller(caller_), st(enter(h_, caller_)) {}
    void ok() { good = true; }
    ~WorkScope();
};

// Capture `reps` repetitions of `body` on `st` into an executable graph.
template <typename F>
int capture(hipStream_t st, int reps, F body, hipGraphExec_t *out)
{
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { set_error("decoder: hipStreamBeginCapture failed: %s", hipGetErrorString(e)); return WR_ELAUNCH; }
    for (int i = 0; i < reps; ++i) body();
    e = hipStreamEndCapture(st, &g);
    if (e != hipSuccess || !g) { set_error("decoder: hipStreamEndCapture failed: %s", hipGetErrorString(e)); return WR_ELAUNCH; }
    e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { set_error("decoder: hipGraphInstantiate failed: %s", hipGetErrorString(e)); return WR_ELAUNCH; }
    return WR_OK;
}

}  // namespace

namespace {
WorkScope::~WorkScope()
{
    if (!good) {
        h->stream_lanes = -1;
        for (int i = 0; i <= kMaxLook; ++i) h->greedy_graph_key[i] = -1;
        h->beam_graph_lanes = -1;
        h->hw_graph_key = -1;
    }
    leave(h, caller);
}

// mode 0: fresh utterances; 1: next chunk, keep the pending predictor state; 2: next chunk, reference quirk
int greedy_run(wr_decoder *h, const float *enc_out_d, const int32_t *enc_lens_d, int N, int T, int n_steps, int blank,
               int32_t *hyps_d, int32_t *hyp_lens_d, void *stream, int mode)
{
    WR_REQUIRE(h && enc_out_d && enc_lens_d && hyps_d && hyp_lens_d, WR_EINVAL, "greedy_search: null pointer argument");
    WR_REQUIRE(N > 0 && N <= h->d.NL && N <= h->max_utt, WR_EINVAL, "greedy_search: N=%d exceeds the decoder's capacity", N);
    WR_REQUIRE(T > 0 && T <= h->Tmax, WR_EINVAL, "greedy_search: T=%d exceeds the decoder's Tmax=%d", T, h->Tmax);
    WR_REQUIRE(n_steps >= 1 && blank >= 0 && blank < h->d.V, WR_EINVAL, "greedy_search: bad n_steps/blank");
    WR_REQUIRE(mode == 0 || h->stream_lanes == N, WR_EINVAL,
               "greedy_search_chunk: no stream state for %d lanes (call with reset first)", N);
    WorkScope scope(h, static_cast<hipStream_t>(stream));
    hipStream_t st = scope.st;
    DevState &s = h->host;
    s.enc = enc_out_d; s.enc_lens = enc_lens_d; s.ctc_logp = nullptr;
    s.n_utt = N; s.T = T; s.lanes_per_utt = 1; s.n_lanes = N;
    s.hyps = hyps_d; s.hyp_lens = hyp_lens_d; s.max_hyp = h->max_hyp; s.n_steps = n_steps; s.blank = blank; s.beam = 1;
    if (int rc = upload_state(h, st)) return rc;
    (void)hipMemsetAsync(s.active_count, 0, 3 * sizeof(int32_t), st);    // lanes active, blank decisions, emissions
    (void)hipMemsetAsync(s.lane_active, 0, sizeof(int32_t) * h->d.NLp, st);
    hipLaunchKernelGGL(ep_all_kernel, dim3((unsigned)(((long)N * T + 7) / 8)), dim3(256), (size_t)8 * h->d.E * sizeof(float), st,
                       h->dev, enc_out_d, s.ep_all);
    if (mode == 0) hipLaunchKernelGGL(greedy_init_kernel, dim3(N), dim3(128), 0, st, h->dev);
    else hipLaunchKernelGGL(greedy_chunk_init_kernel, dim3(N), dim3(128), 0, st, h->dev, mode == 2 ? 1 : 0);
    WR_CHECK_LAUNCH("greedy_init");
    h->stream_lanes = N;
    // look-ahead per replay: fixed, or (h->look == 0) chosen from the share of blank decisions in the previous
    // replay -- any value gives the same tokens, so switching between replays is safe
    int look = h->look > 0 ? h->look : 1;
    int prev_blank = 0, prev_emit = 0;
    const long max_micro = (long)T * ((long)n_steps + 1) + 1;
    for (long done = 0; done < max_micro; done += kStepsPerGraph) {
        if (h->use_graph) {
            if (h->greedy_graph_key[look] != N) {
                if (h->greedy_graph[look]) { (void)hipGraphExecDestroy(h->greedy_graph[look]); h->greedy_graph[look] = nullptr; }
                if (int rc = capture(st, kStepsPerGraph, [&] { greedy_micro_step(h, N, st, look); }, &h->greedy_graph[look]))
                    return rc;
                h->greedy_graph_key[look] = N;
            }
            hipError_t e = hipGraphLaunch(h->greedy_graph[look], st);
            if (e != hipSuccess) { set_error("greedy_search: hipGraphLaunch failed: %s", hipGetErrorString(e)); return WR_ELAUNCH; }
        } else {
            for (int i = 0; i < kStepsPerGraph; ++i) greedy_micro_step(h, N, st, look);
        }
        // the reference synchronises on every step (.item()); we do once per kStepsPerGraph micro-steps
        (void)hipMemcpyAsync(h->h_active, s.active_count, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("greedy_search: stream error: %s", hipGetErrorString(e)); return WR_ELAUNCH; }
        if (h->h_active[0] <= 0) break;
        if (h->look == 0) {
            const int nb = h->h_active[1] - prev_blank, ne = h->h_active[2] - prev_emit;
            prev_blank = h->h_active[1]; prev_emit = h->h_active[2];
            const float share = (nb + ne) > 0 ? (float)nb / (float)(nb + ne) : 0.f;
            look = share >= 0.75f ? 4 : share >= 0.62f ? 2 : 1;
            // beyond ~256 joiner rows a micro-step is matrix-core bound and extra frames are no longer free
            while (look > 1 && look * N > 256) look >>= 1;
        }
    }
    WR_CHECK_LAUNCH("greedy_search");
    // every micro-step emits (at most n_steps per frame) or advances the frame, so the budget cannot run out; if it ever
    // does, say so instead of returning truncated hypotheses as WR_OK
    WR_REQUIRE(h->h_active[0] <= 0, WR_ELAUNCH, "greedy_search: %d streams still active after %ld micro-steps",
               h->h_active[0], max_micro);
    scope.ok();
    return WR_OK;
}
}  // namespace

extern "C" int wr_greedy_search(wr_decoder *h, const float *enc_out_d, const int32_t *enc_lens_d, int N, int T, int n_steps,
                                int blank, int32_t *hyps_d, int32_t *hyp_lens_d, void *stream)
{
    return greedy_run(h, enc_out_d, enc_lens_d, N, T, n_steps, blank, hyps_d, hyp_lens_d, stream, 0);
}

extern "C" int wr_greedy_search_chunk(wr_decoder *h, const float *enc_chunk_d, const int32_t *chunk_lens_d, int N, int T,
                                      int n_steps, int blank, int reset, int reference_new_cache, int32_t *hyps_d,
                                      int32_t *hyp_lens_d, void *stream)
{
    return greedy_run(h, enc_chunk_d, chunk_lens_d, N, T, n_steps, blank, hyps_d, hyp_lens_d, stream,
                      reset ? 0 : (reference_new_cache ? 2 : 1));
}

extern "C" int wr_prefix_beam_search(wr_decoder *h, const float *enc_out_d, const int32_t *enc_lens_d,
                                     const float *ctc_logp_d, int B, int T, int beam, float ctc_weight,
                                     float transducer_weight, int blank, int32_t *hyps_d, int32_t *hyp_lens_d,
                                     double *scores_d, int32_t *n_hyps_d, void *stream)
{
    WR_REQUIRE(h && enc_out_d && enc_lens_d && ctc_logp_d && hyps_d && hyp_lens_d && scores_d && n_hyps_d, WR_EINVAL,
               "prefix_beam_search: null pointer argument");
    WR_REQUIRE(beam >= 1 && beam <= h->max_beam, WR_EINVAL, "prefix_beam_search: beam=%d exceeds the decoder's max %d", beam,
               h->max_beam);
    WR_REQUIRE(B > 0 && B <= h->max_utt && B * beam <= h->d.NL, WR_EINVAL,
               "prefix_beam_search: B*beam=%d exceeds the decoder's capacity", B * beam);
    WR_REQUIRE(T > 0 && T <= h->Tmax, WR_EINVAL, "prefix_beam_search: T=%d exceeds the decoder's Tmax=%d", T, h->Tmax);
    WR_REQUIRE(blank >= 0 && blank < h->d.V, WR_EINVAL, "prefix_beam_search: bad blank");
    h->stream_lanes = -1;                         // the lanes' streaming state (caches, tokens) is overwritten
    WorkScope scope(h, static_cast<hipStream_t>(stream));
    hipStream_t st = scope.st;
    DevState &s = h->host;
    s.enc = enc_out_d; s.enc_lens = enc_lens_d; s.ctc_logp = ctc_logp_d;
    s.n_utt = B; s.T = T; s.lanes_per_utt = beam; s.n_lanes = B * beam;
    s.beam = beam; s.ctc_weight = ctc_weight; s.tr_weight = transducer_weight; s.blank = blank;
    s.Lmax = h->Tmax + 1;
    if (int rc = upload_state(h, st)) return rc;
    (void)hipMemsetAsync(s.lane_active, 0, sizeof(int32_t) * h->d.NLp, st);
    const int NLn = B * beam;
    hipLaunchKernelGGL(ep_all_kernel, dim3((unsigned)(((long)B * T + 7) / 8)), dim3(256), (size_t)8 * h->d.E * sizeof(float), st,
                       h->dev, enc_out_d, s.ep_all);
    hipLaunchKernelGGL(beam_init_kernel, dim3(NLn), dim3(128), 0, st, h->dev);
    WR_CHECK_LAUNCH("beam_init");
    const int key = NLn * 1000 + beam;
    if (h->use_graph_beam && h->beam_graph_lanes != key) {
        if (h->beam_graph) { (void)hipGraphExecDestroy(h->beam_graph); h->beam_graph = nullptr; }
        if (int rc = capture(st, kStepsPerGraph, [&] { beam_frame(h, NLn, B, st); }, &h->beam_graph)) return rc;
        h->beam_graph_lanes = key;
    }
    for (int f = 0; f < T; f += kStepsPerGraph) {
        if (h->use_graph_beam) {
            hipError_t e = hipGraphLaunch(h->beam_graph, st);
            if (e != hipSuccess) { set_error("prefix_beam_search: hipGraphLaunch failed: %s", hipGetErrorString(e)); return WR_ELAUNCH; }
        } else {
            for (int i = 0; i < kStepsPerGraph; ++i) beam_frame(h, NLn, B, st);
        }
    }
    hipLaunchKernelGGL(beam_export_kernel, dim3(B), dim3(256), 0, st, h->dev, hyps_d, hyp_lens_d, scores_d, n_hyps_d);
    WR_CHECK_LAUNCH("prefix_beam_search");
    scope.ok();
    return WR_OK;
}

namespace {
// row-major [L][N][H]  <->  k-major [L][Hp][NLp]
__global__ void cache_to_kmajor_kernel(const float *__restrict__ src, int N, int L, int H, int Hp, int NLp, float *__restrict__ dst)
{
    const long total = (long)L * N * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % H), n = (int)((i / H) % N), l = (int)(i / ((long)H * N));
        dst[((size_t)l * Hp + k) * NLp + n] = src[i];
    }
}
__global__ void cache_from_kmajor_kernel(const float *__restrict__ src, int N, int L, int H, int Hp, int NLp, float *__restrict__ dst)
{
    const long total = (long)L * N * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % H), n = (int)((i / H) % N), l = (int)(i / ((long)H * N));
        dst[i] = src[((size_t)l * Hp + k) * NLp + n];
    }
}

__global__ void step_setup_kernel(DevState *s, const int32_t *tokens, int N)
{
    const int n = blockIdx.x;                      // one workgroup per lane slot
    const int on = n < N;
    const int tok = on ? tokens[n] : 0;
    if (threadIdx.x == 0) {
        s->lane_active[n] = on;
        s->need_pred[n] = on;
        s->lane_t[n] = 0;
        s->token[n] = tok;
        s->comm_slot[n] = n;
        s->new_slot[n] = s->d.NLp + n;
    }
    if (s->d.ptype != kPredLstm) write_embedding_column(s, n, tok);
}

// LSTM step API: caller's cell state [L][N][H] -> the committed slots of the pool; the new slots' cell state -> caller
__global__ void pool_c_import_kernel(DevState *s, const float *__restrict__ src, int N)
{
    const Dims &d = s->d;
    const long total = (long)d.L * N * d.H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % d.H), n = (int)((i / d.H) % N), l = (int)(i / ((long)d.H * N));
        s->pool_c[((size_t)l * 2 * d.NLp + n) * d.Hp + k] = src[i];
    }
}
__global__ void pool_c_export_kernel(DevState *s, float *__restrict__ dst, int N)
{
    const Dims &d = s->d;
    const long total = (long)d.L * N * d.H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % d.H), n = (int)((i / d.H) % N), l = (int)(i / ((long)d.H * N));
        dst[i] = s->pool_c[((size_t)l * 2 * d.NLp + s->new_slot[n]) * d.Hp + k];
    }
}
}  // namespace

// Single predictor step for `N` lanes through the same kernels the decoders use (step API of
// predictor.py:160-200): tokens + cache in, projected output [N,P] and new cache [L,N,H] out.
extern "C" int wr_predictor_step(wr_decoder *h, const int32_t *tokens_d, const float *cache_h_d, const float *cache_c_d, int N,
                                 float *out_d, float *new_h_d, float *new_c_d, void *stream)
{
    WR_REQUIRE(h && tokens_d && cache_h_d && cache_c_d && out_d && new_h_d && new_c_d, WR_EINVAL,
               "predictor_step: null pointer argument");
    WR_REQUIRE(N > 0 && N <= h->d.NL, WR_EINVAL, "predictor_step: N=%d exceeds the decoder's capacity %d", N, h->d.NL);
    h->stream_lanes = -1;                         // the lanes' streaming state (caches, tokens) is overwritten
    WorkScope scope(h, static_cast<hipStream_t>(stream));
    hipStream_t st = scope.st;
    const Dims &d = h->d;
    DevState &s = h->host;
    s.n_utt = 1; s.T = 1; s.lanes_per_utt = d.NL; s.n_lanes = N; s.enc = nullptr; s.enc_lens = nullptr;
    if (int rc = upload_state(h, st)) return rc;
    hipLaunchKernelGGL(step_setup_kernel, dim3(d.NLp), dim3(64), 0, st, h->dev, tokens_d, N);
    if (d.ptype == kPredLstm) {
        // the caller's state becomes the lanes' committed slots: cell states as they are, hidden states through their
        // recurrent products (pool_g = W_hh . h + b, one GEMM per layer); then layer 0, layers >= 1, projection
        hipLaunchKernelGGL(cache_to_kmajor_kernel, dim3(64), dim3(256), 0, st, cache_h_d, N, d.L, d.H, d.Hp, d.NLp, s.new_hT);
        hipLaunchKernelGGL(pool_c_import_kernel, dim3(64), dim3(256), 0, st, h->dev, cache_c_d, N);
        for (int l = 0; l < d.L; ++l) launch_gemm<kEpiSlotRow>(recurrent_job(h, l, N, s.comm_slot), d.G4p, N, st);
        hipLaunchKernelGGL(lstm_layer0_kernel, dim3(N), dim3(256), 0, st, h->dev);
    } else {
        hipLaunchKernelGGL(cache_to_kmajor_kernel, dim3(64), dim3(256), 0, st, cache_h_d, N, d.L, d.H, d.Hp, d.NLp, s.cache_hT);
        hipLaunchKernelGGL(cache_to_kmajor_kernel, dim3(64), dim3(256), 0, st, cache_c_d, N, d.L, d.H, d.Hp, d.NLp, s.cache_cT);
    }
    launch_predictor(h, N, st);
    // outT [Pp][NLp] -> out [N][P]: the same index map with L = 1
    hipLaunchKernelGGL(cache_from_kmajor_kernel, dim3(32), dim3(256), 0, st, s.outT, N, 1, d.P, d.Pp, d.NLp, out_d);
    hipLaunchKernelGGL(cache_from_kmajor_kernel, dim3(64), dim3(256), 0, st, s.new_hT, N, d.L, d.H, d.Hp, d.NLp, new_h_d);
    if (d.ptype == kPredLstm)
        hipLaunchKernelGGL(pool_c_export_kernel, dim3(64), dim3(256), 0, st, h->dev, new_c_d, N);
    else
        hipLaunchKernelGGL(cache_from_kmajor_kernel, dim3(64), dim3(256), 0, st, s.new_cT, N, d.L, d.H, d.Hp, d.NLp, new_c_d);
    WR_CHECK_LAUNCH("predictor_step");
    scope.ok();
    return WR_OK;
}

// ------------------------------------------------------------ hot-word greedy: host side --
namespace {

struct HwCarve {
    size_t q_wt, o_wt, c_wt, kbuf[2], vbuf[2], cold_c, mqT[2], mq_c[2], noT[2], c1p_wt, c1p_b, cold_cp, mqpT[2], mqp_c[2], gate_tab,
        ep2, state, biasT, total;
};

HwCarve hw_carve(const wr_decoder *h, int D, int max_ctx)
{
    HwCarve c;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = align_up(off, 256); off = o + bytes; return o; };
    const Dims &d = h->d;
    c.q_wt = take((size_t)D * D * sizeof(float));
    c.o_wt = take((size_t)D * D * sizeof(float));
    c.c_wt = take((size_t)2 * D * D * sizeof(float));
    for (int i = 0; i < 2; ++i) { c.kbuf[i] = take((size_t)max_ctx * D * sizeof(float)); c.vbuf[i] = take((size_t)max_ctx * D * sizeof(float)); }
    c.cold_c = take((size_t)D * sizeof(float));
    for (int i = 0; i < 2; ++i) {
        c.mqT[i] = take((size_t)D * D * sizeof(float)); c.mq_c[i] = take((size_t)D * sizeof(float));
        c.noT[i] = take((size_t)D * D * sizeof(float));
        c.mqpT[i] = take((size_t)d.Hp * D * sizeof(float)); c.mqp_c[i] = take((size_t)D * sizeof(float));
    }
    c.c1p_wt = take((size_t)d.Hp * D * sizeof(float)); c.c1p_b = take((size_t)D * sizeof(float));
    c.cold_cp = take((size_t)D * sizeof(float));
    c.gate_tab = take((size_t)h->max_utt * h->Tmax * sizeof(int32_t));
    c.ep2 = take((size_t)2 * h->max_utt * h->Tmax * d.J * sizeof(float));
    c.state = take((size_t)4 * d.NLp * sizeof(int32_t));
    c.biasT = take((size_t)d.Pp * d.NLp * sizeof(float));
    c.total = align_up(off, 256);
    return c;
}

int hw_check(const wr_decoder *h, const wr_hotword_weights *hw, int max_ctx)
{
    WR_REQUIRE(h && hw, WR_EINVAL, "hotword: null pointer argument");
    WR_REQUIRE(hw->dim > 0 && hw->heads > 0 && hw->hw_dim > 0 && hw->n_labels > 0 && max_ctx > 0, WR_EINVAL,
               "hotword: non-positive dimension");
    WR_REQUIRE(hw->dim == h->d.E && hw->dim == h->d.P, WR_EINVAL,
               "hotword: dim=%d must equal the encoder (%d) and predictor (%d) output sizes", hw->dim, h->d.E, h->d.P);
    WR_REQUIRE(hw->dim <= 512 && hw->hw_dim <= 256 && hw->n_labels <= 8 && hw->dim % hw->heads == 0 && hw->heads <= 16, WR_EUNSUPPORTED,
               "hotword: dim=%d (<= 512), hw_dim=%d (<= 256), n_labels=%d (<= 8), heads=%d (divides dim, <= 16)", hw->dim,
               hw->hw_dim, hw->n_labels, hw->heads);
    WR_REQUIRE(h->d.ptype != kPredLstm || h->d.Hp <= 4 * hw->dim, WR_EUNSUPPORTED,
               "hotword: LSTM hidden size %d exceeds 4 * dim (%d)", h->d.H, hw->dim);
    WR_REQUIRE((size_t)(21 * hw->dim + hw->heads * max_ctx) * sizeof(float) <= 60 * 1024, WR_EUNSUPPORTED,
               "hotword: max_ctx=%d does not fit the bias kernel's LDS", max_ctx);
    WR_REQUIRE(hw->q_w && hw->q_b && hw->k_w && hw->k_b && hw->v_w && hw->v_b && hw->o_w && hw->o_b && hw->bias_norm_w &&
                   hw->bias_norm_b && hw->combine_w && hw->combine_b && hw->out_norm_w && hw->out_norm_b && hw->hw_enc_w &&
                   hw->hw_enc_b && hw->hw_v_w && hw->hw_v_b && hw->hw_o_w && hw->hw_o_b && hw->hw_norm_w && hw->hw_norm_b &&
                   hw->hw_out_w && hw->hw_out_b, WR_EINVAL, "hotword: null weight pointer");
    return WR_OK;
}

void hw_micro_step(wr_decoder *h, int n_lanes, hipStream_t st)
{
    const Dims &d = h->d;
    const DevState &s = h->host;
    auto up = [](int x, int m) { return (x + m - 1) / m * m; };
    // LSTM predictor: the projection runs inside hw_bias_kernel; stateless predictors write outT themselves
    const bool recurrent_pending = launch_predictor(h, n_lanes, st, d.ptype != kPredLstm);
    const size_t lds = (size_t)(21 * h->hw.D + h->hw.heads * h->hw.max_ctx) * sizeof(float);
    hipLaunchKernelGGL(hw_bias_kernel, dim3(n_lanes), dim3(kHwThreads), lds, st, h->dev, h->hw);
    {   // pred_ffn of the biased predictor output; the joiner activation takes the encoder stream the gate selected
        GemmArgs g{};
        g.A0 = h->hw_biasT; g.B0 = s.predffn_wt; g.K0 = d.Pp;
        g.lda = d.NLp; g.ldb = up(d.J, 32); g.bias = s.predffn_b; g.C = s.ht; g.ldc = kMaxLook * d.NLp; g.N = d.J; g.n_lanes = n_lanes;
        g.st = h->dev; g.lane_active = s.lane_active; g.lane_t = s.lane_t; g.ep_all = h->hw_ep2; g.J = d.J;
        g.look = 1; g.lane_stride = d.NLp; g.act = d.act;
        g.ep_gate = h->hw_state; g.ep_gate_stride = (size_t)h->max_utt * h->Tmax * d.J;
#ifdef WR_STAMPS
        g.dbg_slot = 2;
#endif
        if (recurrent_pending)
            launch_gemm_pair<kEpiJointAct, kEpiSlotRow>(g, up(d.J, 32), recurrent_job(h, d.L - 1, n_lanes, s.new_slot), d.G4p, n_lanes, st);
        else
            launch_gemm<kEpiJointAct>(g, up(d.J, 32), n_lanes, st);
    }
    GemmArgs g{};
#ifdef WR_STAMPS
    g.dbg_slot = 3;
#endif
    g.A0 = s.ht; g.B0 = s.out_wt; g.K0 = d.Jp;
    g.lda = kMaxLook * d.NLp; g.ldb = d.Vp; g.bias = s.out_b; g.C = s.logits; g.ldc = d.V; g.N = d.V; g.n_lanes = n_lanes;
    g.row_part = s.row_part; g.row_part_ld = s.n_cb;
    launch_gemm<kEpiRowStats>(g, d.Vp, n_lanes, st);
    launch_greedy_update<true>(h, n_lanes, st);
}

}  // namespace

extern "C" size_t wr_hotword_workspace_bytes(const wr_decoder *h, const wr_hotword_weights *hw, int max_ctx)
{
    if (!h || !hw || hw->dim <= 0 || max_ctx <= 0) return 0;
    return hw_carve(h, hw->dim, max_ctx).total;
}

extern "C" int wr_decoder_attach_hotword(wr_decoder *h, const wr_hotword_weights *hw, int max_ctx, void *workspace_d,
                                         size_t workspace_bytes, void *stream)
{
    if (int rc = hw_check(h, hw, max_ctx)) return rc;
    WR_REQUIRE(workspace_d != nullptr, WR_EINVAL, "decoder_attach_hotword: null workspace");
    const HwCarve c = hw_carve(h, hw->dim, max_ctx);
    WR_REQUIRE(workspace_bytes >= c.total, WR_EWORKSPACE, "decoder_attach_hotword: workspace %zu < required %zu", workspace_bytes,
               c.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_d);
    const int D = hw->dim;
    HwDev &v = h->hw;
    v.D = D; v.heads = hw->heads; v.HW = hw->hw_dim; v.NLAB = hw->n_labels; v.max_ctx = max_ctx;
    float *q_wt = reinterpret_cast<float *>(ws + c.q_wt), *o_wt = reinterpret_cast<float *>(ws + c.o_wt);
    float *c_wt = reinterpret_cast<float *>(ws + c.c_wt);
    launch_transpose(hw->q_w, D, D, D, D, q_wt, st);                  // [out][in] -> [in][out]
    launch_transpose(hw->o_w, D, D, D, D, o_wt, st);
    launch_transpose(hw->combine_w, D, 2 * D, D, 2 * D, c_wt, st);    // [D][2D] -> [2D][D]
    WR_CHECK_LAUNCH("decoder_attach_hotword (weight transposes)");
    v.q_wt = q_wt; v.q_b = hw->q_b; v.k_w = hw->k_w; v.k_b = hw->k_b; v.v_w = hw->v_w; v.v_b = hw->v_b;
    v.o_wt = o_wt; v.o_b = hw->o_b; v.bn_w = hw->bias_norm_w; v.bn_b = hw->bias_norm_b;
    v.c_wt = c_wt; v.c_b = hw->combine_b; v.on_w = hw->out_norm_w; v.on_b = hw->out_norm_b;
    v.he_w = hw->hw_enc_w; v.he_b = hw->hw_enc_b; v.hv_w = hw->hw_v_w; v.hv_b = hw->hw_v_b; v.ho_w = hw->hw_o_w;
    v.ho_b = hw->hw_o_b; v.hn_w = hw->hw_norm_w; v.hn_b = hw->hw_norm_b; v.hl_w = hw->hw_out_w; v.hl_b = hw->hw_out_b;
    for (int i = 0; i < 2; ++i) {
        v.kbuf[i] = reinterpret_cast<float *>(ws + c.kbuf[i]);
        v.vbuf[i] = reinterpret_cast<float *>(ws + c.vbuf[i]);
    }
    v.cold_c = reinterpret_cast<float *>(ws + c.cold_c);
    for (int i = 0; i < 2; ++i) {
        v.mqT[i] = reinterpret_cast<float *>(ws + c.mqT[i]); v.mq_c[i] = reinterpret_cast<float *>(ws + c.mq_c[i]);
        v.noT[i] = reinterpret_cast<float *>(ws + c.noT[i]);
        v.mqpT[i] = reinterpret_cast<float *>(ws + c.mqpT[i]); v.mqp_c[i] = reinterpret_cast<float *>(ws + c.mqp_c[i]);
    }
    v.lane_active = h->host.lane_active; v.need_pred = h->host.need_pred;
    v.cur_gate = reinterpret_cast<const int32_t *>(ws + c.state);          // = h->hw_state (cur_gate is its first NLp words)
    v.c1p_wt = reinterpret_cast<float *>(ws + c.c1p_wt); v.c1p_b = reinterpret_cast<float *>(ws + c.c1p_b);
    v.cold_cp = reinterpret_cast<float *>(ws + c.cold_cp);
    {
        const Dims &dd = h->d;
        auto up32 = [](int x) { return (x + 31) / 32 * 32; };
        v.proj_wt = dd.ptype == kPredLstm ? h->host.proj_wt : nullptr; v.proj_b = h->host.proj_b;
        v.h_lastT = h->host.new_hT + (size_t)(dd.L - 1) * dd.Hp * dd.NLp;
        v.Hp = dd.Hp; v.P = dd.P; v.proj_ld = up32(dd.P); v.NLp = dd.NLp;
        if (v.proj_wt != nullptr && dd.Hp <= D) {
            const long nn = (long)dd.Hp * D + D;
            hipLaunchKernelGGL(hw_compose_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, hw->combine_w, hw->combine_b,
                               v.proj_wt, v.proj_b, D, dd.P, dd.H, dd.Hp, v.proj_ld, v.c1p_wt, v.c1p_b);
            WR_CHECK_LAUNCH("decoder_attach_hotword (composed projection)");
        }
    }
    v.gate_tab = reinterpret_cast<int32_t *>(ws + c.gate_tab);
    h->hw_ep2 = reinterpret_cast<float *>(ws + c.ep2);
    h->hw_state = reinterpret_cast<int32_t *>(ws + c.state);
    h->hw_biasT = reinterpret_cast<float *>(ws + c.biasT);
    (void)hipMemsetAsync(h->hw_biasT, 0, (size_t)h->d.Pp * h->d.NLp * sizeof(float), st);   // padded rows / lanes stay zero
    h->hw_attached = true;
    h->hw_graph_key = -1;
    return WR_OK;
}

extern "C" int wr_greedy_search_hotword(wr_decoder *h, const float *enc_hot_d, const float *enc_cold_d, const float *enc_feat_d,
                                        const int32_t *enc_lens_d, const float *hidden_hot_d, int n_ctx_hot,
                                        const float *hidden_cold_d, int n_ctx_cold, int N, int T, int n_steps, int blank,
                                        int filter_on, int32_t *hyps_d, int32_t *hyp_lens_d, int32_t *trace_d, int trace_cap,
                                        int32_t *trace_lens_d, void *stream)
{
    WR_REQUIRE(h && enc_hot_d && enc_cold_d && enc_feat_d && enc_lens_d && hidden_hot_d && hidden_cold_d && hyps_d &&
                   hyp_lens_d && trace_d && trace_lens_d, WR_EINVAL, "greedy_search_hotword: null pointer argument");
    WR_REQUIRE(h->hw_attached, WR_EINVAL, "greedy_search_hotword: call wr_decoder_attach_hotword first");
    WR_REQUIRE(N > 0 && N <= h->d.NL && N <= h->max_utt, WR_EINVAL, "greedy_search_hotword: N=%d exceeds the decoder's capacity", N);
    WR_REQUIRE(T > 0 && T <= h->Tmax, WR_EINVAL, "greedy_search_hotword: T=%d exceeds the decoder's Tmax=%d", T, h->Tmax);
    WR_REQUIRE(n_steps >= 1 && blank >= 0 && blank < h->d.V && trace_cap >= 1, WR_EINVAL, "greedy_search_hotword: bad n_steps/blank/trace_cap");
    WR_REQUIRE(n_ctx_hot >= 1 && n_ctx_hot <= h->hw.max_ctx && n_ctx_cold >= 1 && n_ctx_cold <= h->hw.max_ctx, WR_EINVAL,
               "greedy_search_hotword: context lists of %d / %d entries (1..%d)", n_ctx_hot, n_ctx_cold, h->hw.max_ctx);
    h->stream_lanes = -1;                         // the lanes' streaming state is overwritten
    WorkScope scope(h, static_cast<hipStream_t>(stream));
    hipStream_t st = scope.st;
    const Dims &d = h->d;
    DevState &s = h->host;
    s.enc = enc_hot_d; s.enc_lens = enc_lens_d; s.ctc_logp = nullptr;
    s.n_utt = N; s.T = T; s.lanes_per_utt = 1; s.n_lanes = N;
    s.hyps = hyps_d; s.hyp_lens = hyp_lens_d; s.max_hyp = h->max_hyp; s.n_steps = n_steps; s.blank = blank; s.beam = 1;
    s.hw_on = 1; s.hw_filter = filter_on ? 1 : 0; s.hw_nctx[0] = n_ctx_cold; s.hw_nctx[1] = n_ctx_hot;
    s.gate_tab = h->hw.gate_tab;
    s.cur_gate = h->hw_state; s.gb_flag = h->hw_state + d.NLp; s.gb_end = h->hw_state + 2 * d.NLp; s.last_t = h->hw_state + 3 * d.NLp;
    s.trace = trace_d; s.trace_len = trace_lens_d; s.trace_cap = trace_cap;
    s.biasT = h->hw_biasT;
    const int rc_up = upload_state(h, st);
    s.hw_on = 0;                                  // the host copy goes back to plain greedy for the other entry points
    if (rc_up) return rc_up;
    (void)hipMemsetAsync(s.active_count, 0, 3 * sizeof(int32_t), st);
    (void)hipMemsetAsync(s.lane_active, 0, sizeof(int32_t) * d.NLp, st);
    // loop-invariant work: list projections, the gate of every frame, enc_ffn of both encoder streams
    hipLaunchKernelGGL(hw_kv_kernel, dim3(n_ctx_cold, 2), dim3(256), 0, st, h->hw, hidden_cold_d, 0);
    hipLaunchKernelGGL(hw_kv_kernel, dim3(n_ctx_hot, 2), dim3(256), 0, st, h->hw, hidden_hot_d, 1);
    {   // short lists: fold the query / output projections into K / V once per call
        const int nc[2] = {n_ctx_cold, n_ctx_hot};
        for (int i = 0; i < 2; ++i)
            if (nc[i] > 1 && h->hw.heads * nc[i] <= h->hw.D)
                hipLaunchKernelGGL(hw_fold_kernel, dim3(h->hw.heads * nc[i]), dim3(256), 0, st, h->hw, i, nc[i]);
    }
    if (n_ctx_cold == 1)
        hipLaunchKernelGGL(hw_cold_kernel, dim3(1), dim3(kHwThreads), (size_t)(3 + kHwThreads / 64) * h->hw.D * sizeof(float), st, h->hw);
    hipLaunchKernelGGL(hw_gate_table_kernel, dim3((unsigned)((long)N * T)), dim3(256),
                       (size_t)(h->hw.D + 2 * h->hw.HW) * sizeof(float), st, h->hw, enc_feat_d, (long)N * T);
    const size_t ep_stride = (size_t)h->max_utt * h->Tmax * d.J;
    const dim3 ep_grid((unsigned)(((long)N * T + 7) / 8));
    hipLaunchKernelGGL(ep_all_kernel, ep_grid, dim3(256), (size_t)8 * d.E * sizeof(float), st, h->dev, enc_cold_d, h->hw_ep2);
    hipLaunchKernelGGL(ep_all_kernel, ep_grid, dim3(256), (size_t)8 * d.E * sizeof(float), st, h->dev, enc_hot_d, h->hw_ep2 + ep_stride);
    hipLaunchKernelGGL(greedy_hw_init_kernel, dim3(N), dim3(128), 0, st, h->dev);
    WR_CHECK_LAUNCH("greedy_search_hotword (setup)");
    // every go-back re-decodes frames, at most once per 0 -> 1 flip of the gate: twice the plain loop's bound
    const long max_micro = 2 * ((long)T * ((long)n_steps + 2) + 1);
    for (long done = 0; done < max_micro; done += kStepsPerGraph) {
        if (h->use_graph) {
            if (h->hw_graph_key != N) {
                if (h->hw_graph) { (void)hipGraphExecDestroy(h->hw_graph); h->hw_graph = nullptr; }
                if (int rc = capture(st, kStepsPerGraph, [&] { hw_micro_step(h, N, st); }, &h->hw_graph)) return rc;
                h->hw_graph_key = N;
            }
            hipError_t e = hipGraphLaunch(h->hw_graph, st);
            if (e != hipSuccess) { set_error("greedy_search_hotword: hipGraphLaunch failed: %s", hipGetErrorString(e)); return WR_ELAUNCH; }
        } else {
            for (int i = 0; i < kStepsPerGraph; ++i) hw_micro_step(h, N, st);
        }
        (void)hipMemcpyAsync(h->h_active, s.active_count, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("greedy_search_hotword: stream error: %s", hipGetErrorString(e)); return WR_ELAUNCH; }
        if (h->h_active[0] <= 0) break;
    }
    WR_CHECK_LAUNCH("greedy_search_hotword");
    WR_REQUIRE(h->h_active[0] <= 0, WR_ELAUNCH, "greedy_search_hotword: %d streams still active after %ld micro-steps "
               "(go-backs exceeded the budget); hypotheses are incomplete", h->h_active[0], max_micro);
    scope.ok();
    return WR_OK;
}

#ifdef WR_STAMPS
// diagnostic build only: copy the stamp array to the host and clear it
extern "C" int wr_debug_read_stamps(unsigned long long *host, size_t n_words)
{
    const size_t total = (size_t)wr::kStampSlots * wr::kStampWgs * wr::kStampWaves * wr::kStampPts;
    if (!host || n_words < total) return (int)total;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(wr::g_stamps), total * sizeof(unsigned long long)) != hipSuccess) return -1;
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(wr::g_stamps)) == hipSuccess) (void)hipMemset(p, 0, total * sizeof(unsigned long long));
    return 0;
}
#endif
