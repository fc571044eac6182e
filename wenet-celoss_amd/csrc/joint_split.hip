// Transducer joint network on the bf16 matrix cores of MI355X (gfx950), split precision.
//
// Same operator as joint.hip (TransducerJoint.forward, wenet/transducer/joint.py:60-69, from the point where
// ep = enc_ffn(enc) [B,T,J] and pp = pred_ffn(pred) [B,U1,J] exist):
//     out[m, v] = sum_k tanh(ep[bt(m),k] + pp[bu(m),k]) * W[v,k] + bias[v]
// but the 512 -> V contraction runs on v_mfma_f32_32x32x16_bf16 (16x the rate of the exact-fp32 MFMA) with
// every fp32 operand x split into two bf16 numbers  x = hi + lo + O(2^-17 |x|):
//     terms = 3:  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi      (dropped: a_lo*b_lo ~ 2^-16 |a*b|)
//                 fp32 accumulation; |error| of a logit ~ 1e-5 of its scale -- inside the 1e-4 parity bar
//                 of the fp32 path (tests/test_joint_gpu.py states the tolerance);
//     terms = 1:  a_hi*b_hi only -- the AMP path: the reference under --use_amp (executor.py:91 autocast)
//                 runs this Linear in fp16 with fp32 accumulation; bf16 operands, fp32 accumulate here.
// The exact-fp32 kernels of joint.hip stay the default; this file is opt-in (precision argument of
// wenet_celoss_amd.TransducerJoint / joint_logits, or WR_JOINT_PRECISION).
//
// Forward  joint_fwd_split_kernel: one workgroup (4 waves, one per SIMD, 512 registers each) owns 64 consecutive
//   lattice cells and all V columns.  H = tanh(ep + pp) (64 x J; tanh through v_exp/v_rcp) is computed once,
//   split, and kept in LDS as bf16 hi / lo images (row-major, 16-byte padded rows: conflict-free ds_read_b128
//   fragments, read one k-step ahead of their MFMAs); W is re-laid once per call into MFMA fragment order (hi
//   and lo images; each (32 columns x 16 k) fragment = 1 KB contiguous) and streamed from L2 / Infinity Cache
//   straight into three rotating register sets of 4 k-steps (two in flight, 32 KB per wave).  A wave owns 64
//   columns (2 tiles) per round: 2 row tiles x 2 column tiles x `terms` MFMAs per k-step; logits leave with
//   non-temporal stores.  Measured (B=8, T=1000, U1=151, J=512, V=5000): terms=3 20.8 ms = 297 TFLOP/s
//   fp32-equivalent (892 TFLOP/s of bf16 MFMA, 2.7x the exact-fp32 kernel), terms=1 13.8 ms; matrix cores 40 %
//   busy (SQ_VALU_MFMA_BUSY_CYCLES) -- with one wave per SIMD the tile build, the k-loop and the epilogue of a
//   workgroup do not overlap, and 138 KB of LDS per workgroup rules out a second one per CU.
#include <type_traits>

#include "wr_common.hpp"
#include "joint_lse.hpp"

namespace wr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSM = 64;          // lattice cells per workgroup
constexpr int kSCT = 2;          // 32-column tiles per wave per round
constexpr int kSWaves = 4;       // one per SIMD, 512 registers each: three fragment sets without register reuse stalls
constexpr int kSPF = 4;          // k-steps (of 16) per register set; three sets rotate, two are in flight

// ---- diagnostic build only (-DWR_JS_STAMPS, tools/amp_stamps.py): s_memtime / s_memrealtime stamps of the forward
// kernel's phases, one record per wave of every 16th workgroup; the values go to a buffer nothing else reads.
#ifdef WR_JS_STAMPS
constexpr int kJsWgs = 2048, kJsPts = 12;
__device__ unsigned long long g_js[kJsWgs * kSWaves * kJsPts];
#define WR_JS_NOW(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WR_JS_NOW_RT(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WR_JS_NOW(v)
#define WR_JS_NOW_RT(v)
#endif

inline int split_jpad(int J) { return (J + 16 * kSPF - 1) / (16 * kSPF) * (16 * kSPF); }
inline int split_vpad(int V) { return (V + 32 * kSCT - 1) / (32 * kSCT) * (32 * kSCT); }

// round-to-nearest-even bf16 of a finite float
__device__ __forceinline__ unsigned bf16_bits(float x)
{
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split_bf16(float x, unsigned &hi, unsigned &lo)
{
    hi = bf16_bits(x);
    lo = bf16_bits(x - __builtin_bit_cast(float, hi << 16));
}

// two floats -> packed bf16 pair of the hi parts and of the lo parts (v_cvt_pk_bf16_f32, round to nearest even):
// five instructions per pair
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &hi, unsigned &lo)
{
    const f32x2 x = (f32x2){x0, x1};
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    const bf16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// tanh through the hardware exp2 / rcp (|error| ~ 1e-7 absolute: below the split's own 2^-17 operand error)
__device__ __forceinline__ float tanh_fast(float x)
{
    const float t = __builtin_amdgcn_exp2f(fabsf(x) * 2.885390082f);      // e^{2|x|}
    const float r = 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
    return __builtin_copysignf(r, x);
}

// W [V, J] (nn.Linear weight) -> fragment-major hi / lo images:
//   frag[(ct * S + s) * 64 + l][j] = W[ct*32 + (l & 31)][s*16 + 8*(l >> 5) + j]      (zero outside V x J)
__global__ void split_w_kernel(const float *__restrict__ w, int V, int J, int Vp, int Jp, unsigned short *__restrict__ wh,
                               unsigned short *__restrict__ wl)
{
    const int S = Jp / 16;
    const long total = (long)(Vp / 32) * S * 64;
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
        const int l = (int)(f & 63);
        const long cs = f >> 6;
        const int s = (int)(cs % S), ct = (int)(cs / S);
        const int v = ct * 32 + (l & 31), k0 = s * 16 + 8 * (l >> 5);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (v < V && k0 + j < J) ? w[(size_t)v * J + k0 + j] : 0.f;
            unsigned hi, lo;
            split_bf16(x, hi, lo);
            wh[f * 8 + j] = (unsigned short)hi;
            wl[f * 8 + j] = (unsigned short)lo;
        }
    }
}

#ifdef WR_SPLIT_PLAIN_STORE
#define WR_SPLIT_STORE(v, p) (*(p) = (v))
#else
#define WR_SPLIT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif
template <typename OutT> __device__ __forceinline__ OutT to_out(float x);
template <> __device__ __forceinline__ float to_out<float>(float x) { return x; }
template <> __device__ __forceinline__ _Float16 to_out<_Float16>(float x) { return (_Float16)x; }
template <> __device__ __forceinline__ __bf16 to_out<__bf16>(float x) { return (__bf16)x; }

// LSE = true (float logits, npart = 1): the epilogue also produces the RNN-T loss's row statistics (joint_lse.hpp).
// RT = row tiles of 32 lattice cells per workgroup: 2 (64 cells; all modes) or 4 (128 cells; single-term / AMP mode only:
// one bf16 image of 128 x J fits LDS where the hi + lo images of the split modes do not).  A workgroup streams ALL of W
// for its cells, so 128 cells halve the W traffic per logit -- the stream out of the L2s is what bounds this kernel
// (round 2: 9.7 TB/s at 64 cells).  A wave then owns 4 x 2 accumulator tiles per round.
// OCC = workgroups per CU the kernel is built for: 1 (512 registers per wave) or, single-term mode with 64 cells only, 2
// (256 registers per wave, LDS <= 80 KB: the bias comes from memory a round ahead instead of from an LDS slab, the store
// stage takes 16 or 8 rows at a time) -- with one wave per SIMD nothing overlaps a workgroup's tile build, the issue of
// its W loads (the CU's 64 B/clk vector-memory path is busy for as long as the MFMAs of the set) and its epilogues;
// a second workgroup's MFMAs do (stamps: profiles/r03_amp_stamps_*.json).
// TRN (single-term mode): the MFMAs take the W fragment as the A operand and the activation fragment as the B operand
// (both have the same register layout), so a tile comes out transposed -- a lane holds ONE lattice cell and 4 x 4
// consecutive vocabulary entries of it, which it adds the bias to, packs and stores itself (8 or 16 bytes per lane and
// store): no LDS stage, no wave barriers, a quarter of the store instructions of the cell-major layout.
// NW = waves per workgroup: 4 in every shipped form.  8 with RT = 4, OCC = 2 is the 128-cell form as ONE workgroup of eight
// waves per CU (two per SIMD, 256 registers each; a 4 x 2 register tile needs half the W bytes per MFMA): measured 9.1-10.2 ms
// against 7.8 for two 64-cell workgroups -- eight accumulator tiles in 256 registers spill in the staged epilogue
// (26 k cycles per round), the transposed epilogue does not but stores 16-byte pieces -- so it is not instantiated.
template <int TERMS, typename OutT, bool LSE = false, int RT = 2, int OCC = 1, bool TRN = false, int NW = kSWaves>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : OCC) void joint_fwd_split_kernel(
    const float *__restrict__ ep, const float *__restrict__ pp, const u32x4 *__restrict__ wh, const u32x4 *__restrict__ wl,
    const float *__restrict__ bias, const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens, int B, int T,
    int U1, int J, int Jp, int V, int Vp, int npart, int act, OutT *__restrict__ out, JointLse lse = JointLse{},
    int lds_bias_bytes = 0, int lds_stage = 0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_s[];
    constexpr int SM = 32 * RT;                            // lattice cells of this workgroup
    constexpr int PF = RT == 4 ? 2 : kSPF;                 // k-steps per register set (eight accumulator tiles leave room for less)
    static_assert(RT == 2 || (RT == 4 && TERMS == 1 && !LSE), "128-cell tiles: single-term mode without row statistics");
    static_assert(OCC == 1 || (OCC == 2 && TERMS == 1 && !LSE && ((RT == 2 && NW == 4) || (RT == 4 && NW == 8))),
                  "two waves per SIMD: single-term mode; 64 cells x two workgroups or 128 cells x eight waves");
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    static_assert(!TRN || (TERMS == 1 && !LSE), "transposed tiles: single-term mode without row statistics");
    const int JS = Jp + 8;                                 // padded row stride (bf16 elements): 16-byte pad
    unsigned short *Ahi = lds_s;                            // [SM][JS]
    unsigned short *Alo = lds_s + (size_t)SM * JS;          // [SM][JS]   (TERMS == 3)
    float *bias_s = reinterpret_cast<float *>(lds_s + (size_t)(TERMS == 3 ? 2 : 1) * SM * JS);    // this part's bias
    // per-wave store stage of the single-term mode (see finish_round): behind the bias slab; rows of V elements must keep
    // 16-byte alignment for the vector stores
    char *stage = reinterpret_cast<char *>(bias_s) + lds_bias_bytes;
    const bool stage_ok = TERMS == 1 && !LSE && lds_stage && ((size_t)V * sizeof(OutT)) % 16 == 0 &&
                          (reinterpret_cast<size_t>(out) & 15) == 0;
    const long M = (long)B * T * U1;
    // consecutive workgroups land on consecutive XCDs: part = blockIdx % npart keeps each XCD on one column slab of W
    // (<= ~2.5 MB of fragments, resident in its 4 MB L2) for the whole launch
    const int part = blockIdx.x % npart;
    const long m0 = (long)(blockIdx.x / npart) * SM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
#ifdef WR_JS_STAMPS
    unsigned long long js_t0 = 0, js_r0 = 0, js_t1 = 0, js_t2 = 0, js_r2 = 0, js_a = 0, js_b = 0, js_c = 0, js_d = 0;
    unsigned long long js_ld = 0, js_mm = 0, js_ep = 0, js_ep_n = 0, js_mm_first = 0;
    WR_JS_NOW_RT(js_r0);
    WR_JS_NOW(js_t0);
#endif
    float rm[32], rs[32];                                   // LSE: this lane's (reference, partial sum) of its 32 rows
    if (LSE) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { rm[i] = -3.0e38f; rs[i] = 0.f; }
    }

    if (llens != nullptr && tlens != nullptr) {
        int valid = 0;
        const long m = m0 + tid;
        if (tid < SM && m < M) {
            const long bt = m / U1;
            const int u = (int)(m - bt * U1);
            const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
            valid = (t < llens[b]) && (u <= tlens[b]);
        }
        if (!__syncthreads_or(valid)) return;
    }
    const int S = Jp / 16;                                  // k-steps per column tile; Jp is a multiple of 16 * PF
    const int cpr = S / PF;                               // register sets ("chunks") per round
    const int n_ct = Vp / 32;                               // column tiles
    const int pairs = n_ct / kSCT;                          // a wave's unit of work: 64 columns
    const int ppp = (pairs + npart - 1) / npart;            // pairs per part
    const int pair0 = part * ppp;
    const int npairs = pairs - pair0 < ppp ? pairs - pair0 : ppp;
    if (npairs <= 0) return;
    const int rounds = (npairs + NW - 1) / NW;
    const int total = rounds * cpr;

    // fragment pointers of this lane
    const u32x4 *__restrict__ whl = wh + lane;
    const u32x4 *__restrict__ wll = wl + lane;

    // W fragments of set li (PF k-steps x kSCT column tiles of this wave's pair) -> registers; k-step i alone when i >= 0
    // (the main loop issues a set's loads between the MFMAs of the set two ahead of it, one k-step's loads per k-step)
    auto set_base = [&](int lr, int lc0) -> size_t {
        const int pr = lr * NW + wave;
        const int ct0 = (pair0 + (pr < npairs ? pr : npairs - 1)) * kSCT;   // waves past the last pair reload it (results dropped)
        return ((size_t)ct0 * S + (size_t)lc0 * PF) * 64;
    };
    auto load_step = [&](size_t base, int i, u32x4 (&bh)[PF][kSCT], u32x4 (&bl)[PF][kSCT]) {
#pragma unroll
        for (int c = 0; c < kSCT; ++c) {
            const size_t f = base + ((size_t)c * S + i) * 64;
            bh[i][c] = whl[f];
            if (TERMS == 3) bl[i][c] = wll[f];
        }
    };
    u32x4 pbh[PF][kSCT], pbl[PF][kSCT], qbh[PF][kSCT], qbl[PF][kSCT], rbh[PF][kSCT], rbl[PF][kSCT];

    // bias of this part's columns -> LDS (read back per round without touching the vector-memory counter)
    if (OCC == 1) {
        for (int i = tid; i < npairs * 32 * kSCT; i += 64 * NW) {
            const int col = pair0 * 32 * kSCT + i;
            bias_s[i] = col < V ? bias[col] : 0.f;
        }
    }
    float bvr[kSCT] = {0.f, 0.f};                           // OCC == 2: this lane's bias values of the current round
    f32x4 bvt[kSCT][4];                                     // ... of a transposed tile: entries 8 g + 4 half + (0..3) of tile c
    const bool bias_vec = (reinterpret_cast<size_t>(bias) & 15) == 0;
    auto next_bias = [&](int r) {
        if (OCC == 1) return;
        const int pr = r * NW + wave;
        if constexpr (TRN) {
#pragma unroll
            for (int c = 0; c < kSCT; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int v = ((pair0 + pr) * kSCT + c) * 32 + 8 * g + 4 * half;
                    if (pr < npairs && v + 4 <= V && bias_vec) {
                        bvt[c][g] = *reinterpret_cast<const f32x4 *>(bias + v);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) bvt[c][g][j] = (pr < npairs && v + j < V) ? bias[v + j] : 0.f;
                    }
                }
        } else {
#pragma unroll
            for (int c = 0; c < kSCT; ++c) {
                const int col = ((pair0 + pr) * kSCT + c) * 32 + l31;
                bvr[c] = (pr < npairs && col < V) ? bias[col] : 0.f;
            }
        }
    };
    next_bias(0);
    // activation tile H = act(ep[b,t,:] + pp[b,u,:]): a wave takes rows wave, wave+4, ...; a lane owns k = VW*lane + 64*VW*i.
    // The vector-memory path of the CU is what the two resident workgroups share, so the build keeps its instruction count
    // down: 16-byte loads (J % 4 == 0), an ep row fetched again only when (b, t) changes from one of the wave's cells to
    // the next (64 consecutive cells span one or two frames when U1 >= 64).  Small batches: requesting a second batch
    // before the first is evaluated, or 8 rows at once, was 12 % SLOWER -- the other workgroup's W stream waits behind
    // those loads.
    // (b, t, u) of the wave's cells by stepping, one division pair per wave: cell m0 + wave + 4 j
    long w_bt, w_b;
    int w_u, w_t;
    {
        const long m = m0 + wave < M ? m0 + wave : M - 1;
        w_bt = m / U1;
        w_u = (int)(m - w_bt * U1);
        w_b = w_bt / T;
        w_t = (int)(w_bt - w_b * T);
    }
    auto build_tile = [&](auto vw_tag) {
        constexpr int VW = decltype(vw_tag)::value;         // floats per lane and load: 4 or 2
        constexpr int KI = 512 / (64 * VW);                 // Jp <= 512
        constexpr int RB = 2;                               // rows per batch (8 or 16 loads per lane in flight; larger batches
                                                            // measured no faster -- at 8 rows of 16-byte loads 12 % slower)
        typedef float fvec __attribute__((ext_vector_type(VW)));
        static_assert(SM / NW % RB == 0, "rows per wave");
        fvec ec[KI];                                        // the ep row of the cell being evaluated
        long ec_bt = -1;
        for (int rb = 0; rb < SM / NW; rb += RB) {
            fvec ev[RB][KI], pv[RB][KI];
            bool fresh[RB];
            long bt_last = ec_bt;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int row = wave + NW * (rb + q);
                const float *__restrict__ e = ep + (size_t)w_bt * J;
                const float *__restrict__ p = pp + ((size_t)w_b * U1 + w_u) * J;
                fresh[q] = w_bt != bt_last;
                bt_last = w_bt;
                if (m0 + row + NW < M) {               // the wave's next cell (cells past M keep the last valid address)
                    w_u += NW;
                    while (w_u >= U1) {
                        w_u -= U1;
                        ++w_bt;
                        if (++w_t == T) { w_t = 0; ++w_b; }
                    }
                }
#pragma unroll
                for (int i = 0; i < KI; ++i) {
                    const int k = VW * lane + 64 * VW * i;
                    const int kc = k < J ? k : J - VW;      // J is a multiple of VW
                    if (fresh[q]) ev[q][i] = *reinterpret_cast<const fvec *>(e + kc);
                    pv[q][i] = *reinterpret_cast<const fvec *>(p + kc);
                }
            }
            ec_bt = bt_last;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int row = wave + NW * (rb + q);
                const bool in = m0 + row < M;
#pragma unroll
                for (int i = 0; i < KI; ++i) {
                    if (fresh[q]) ec[i] = ev[q][i];
                    const int k = VW * lane + 64 * VW * i;
                    if (k >= Jp) continue;
                    unsigned hp[VW / 2], lp[VW / 2];
#pragma unroll
                    for (int j = 0; j < VW; j += 2) {
                        const bool kin = in && k + j < J;   // J is even
                        const float z0 = ec[i][j] + pv[q][i][j], z1 = ec[i][j + 1] + pv[q][i][j + 1];
                        const float a0 = act == WR_ACT_TANH ? tanh_fast(z0) : act_value(act, z0);
                        const float a1 = act == WR_ACT_TANH ? tanh_fast(z1) : act_value(act, z1);
                        split_pair(kin ? a0 : 0.f, kin ? a1 : 0.f, hp[j / 2], lp[j / 2]);
                    }
                    if constexpr (VW == 4) {
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2 *>(Ahi + (size_t)row * JS + k) = (u32x2){hp[0], hp[1]};
                        if (TERMS == 3) *reinterpret_cast<u32x2 *>(Alo + (size_t)row * JS + k) = (u32x2){lp[0], lp[1]};
                    } else {
                        *reinterpret_cast<unsigned *>(Ahi + (size_t)row * JS + k) = hp[0];
                        if (TERMS == 3) *reinterpret_cast<unsigned *>(Alo + (size_t)row * JS + k) = lp[0];
                    }
                }
            }
        }
    };
    if ((J & 3) == 0 && ((reinterpret_cast<size_t>(ep) | reinterpret_cast<size_t>(pp)) & 15) == 0)
        build_tile(std::integral_constant<int, 4>{});
    else
        build_tile(std::integral_constant<int, 2>{});
    {
        const size_t b0 = set_base(0, 0), b1 = set_base(cpr > 1 ? 0 : 1, cpr > 1 ? 1 : 0);
#pragma unroll
        for (int i = 0; i < PF; ++i) load_step(b0, i, pbh, pbl);
#pragma unroll
        for (int i = 0; i < PF; ++i) load_step(b1, i, qbh, qbl);
    }
    __syncthreads();
    WR_JS_NOW(js_t1);

    const unsigned short *a_hi = Ahi + (size_t)l31 * JS + 8 * half;
    const unsigned short *a_lo = Alo + (size_t)l31 * JS + 8 * half;
    f32x16 acc[RT][kSCT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < kSCT; ++c) acc[r][c] = (f32x16){0};

    // A fragments (LDS) run one k-step ahead of the MFMAs that consume them
    bf16x8 ah[2][RT], al[2][RT];                            // [buffer][row tile]
    auto read_a = [&](int s, int buf) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            ah[buf][r] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(a_hi + (size_t)(32 * r) * JS + 16 * s));
            if (TERMS == 3)
                al[buf][r] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(a_lo + (size_t)(32 * r) * JS + 16 * s));
        }
    };
    read_a(0, 0);
    // set counters: (cr, cc0) = round and chunk of the set being multiplied, (lr, lc0) of the set being loaded (two ahead)
    int cr = 0, cc0 = 0, lr = 0, lc0 = 0;
    auto advance = [&](int &r_, int &c_) { if (++c_ == cpr) { c_ = 0; ++r_; } };
    advance(lr, lc0);
    advance(lr, lc0);
    auto mfma_set = [&](const u32x4 (&bh)[PF][kSCT], const u32x4 (&bl)[PF][kSCT], u32x4 (&nh)[PF][kSCT], u32x4 (&nl)[PF][kSCT]) {
        const int c0 = cc0;
        const int cn = (c0 + 1 == cpr) ? 0 : c0 + 1;        // next chunk's first step (wraps to the next round)
        const bool more = lr < rounds;                      // sets past the last one are not loaded
        const size_t lbase = set_base(more ? lr : rounds - 1, more ? lc0 : cpr - 1);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int buf = i & 1;                          // PF is even: every chunk starts on buffer 0
            read_a(i + 1 < PF ? c0 * PF + i + 1 : cn * PF, buf ^ 1);
#ifndef WR_X_NOLOAD
            load_step(lbase, i, nh, nl);
#endif
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < kSCT; ++c) {
                    const bf16x8 bhv = __builtin_bit_cast(bf16x8, bh[i][c]);
                    if (TERMS == 3) {
                        const bf16x8 blv = __builtin_bit_cast(bf16x8, bl[i][c]);
                        acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[buf][r], bhv, acc[r][c], 0, 0, 0);   // small terms first
                        acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[buf][r], blv, acc[r][c], 0, 0, 0);
                    }
                    if (TRN) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bhv, ah[buf][r], acc[r][c], 0, 0, 0);
                    else acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[buf][r], bhv, acc[r][c], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        advance(lr, lc0);
    };
    auto finish_round = [&]() {                             // after the last chunk of a round: bias, store, reset
        const int r = cr;
        const bool last = cc0 + 1 == cpr;
        advance(cr, cc0);
        if (!last) return;
        const int pr = r * NW + wave;
        const int ct0 = (pair0 + pr) * kSCT;
        // interior tiles (all 64 cells and all 64 columns valid: everything but the matrix edges) store without
        // per-element guards, so the stores issue back to back
        const bool full = m0 + SM <= M && pr < npairs && (ct0 + kSCT) * 32 <= V;
        if constexpr (TRN) {
            // acc[rt][c][q] = logit of cell m0 + 32 rt + l31, vocabulary entry (ct0 + c) * 32 + 8 (q >> 2) + 4 half + (q & 3)
            constexpr int VB = 4 * (int)sizeof(OutT);                       // bytes a lane stores at once
            const bool vec = full && ((size_t)V * sizeof(OutT)) % VB == 0 && (reinterpret_cast<size_t>(out) % VB) == 0;
#ifdef WR_X_NOEPI
            if (vec) {                                      // experiment: the k-loop without its epilogue
                float sum = 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int c = 0; c < kSCT; ++c) { sum += acc[rt][c][0] + acc[rt][c][7]; acc[rt][c] = (f32x16){0}; }
                if (sum == 1.2345e30f) out[0] = to_out<OutT>(sum);
                next_bias(r + 1);
                return;
            }
#endif
#pragma unroll
            for (int c = 0; c < kSCT; ++c) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 b4;
                    if constexpr (OCC == 2) b4 = bvt[c][g];
                    else b4 = *reinterpret_cast<const f32x4 *>(bias_s + ((pr < npairs ? pr : 0) * kSCT + c) * 32 + 8 * g + 4 * half);
                    const int v = (ct0 + c) * 32 + 8 * g + 4 * half;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const long m = m0 + 32 * rt + l31;
                        const float x0 = acc[rt][c][4 * g] + b4[0], x1 = acc[rt][c][4 * g + 1] + b4[1];
                        const float x2 = acc[rt][c][4 * g + 2] + b4[2], x3 = acc[rt][c][4 * g + 3] + b4[3];
                        OutT *__restrict__ o = out + (size_t)m * V + v;
                        if (vec) {
                            if constexpr (sizeof(OutT) == 4) {
                                *reinterpret_cast<f32x4 *>(o) = (f32x4){x0, x1, x2, x3};
                            } else {
                                typedef OutT o2 __attribute__((ext_vector_type(2)));
                                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                                const o2 p0 = __builtin_convertvector((f32x2){x0, x1}, o2);
                                const o2 p1 = __builtin_convertvector((f32x2){x2, x3}, o2);
                                const u32x2 pk = (u32x2){__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
#ifdef WR_X_NOSTORE
                                if (pk.x == 0x7fc12345u && pk.y == 0x7fc12345u)   // experiment: the epilogue without its stores
#endif
                                *reinterpret_cast<u32x2 *>(o) = pk;
                            }
                        } else if (m < M && pr < npairs) {
                            if (v < V) o[0] = to_out<OutT>(x0);
                            if (v + 1 < V) o[1] = to_out<OutT>(x1);
                            if (v + 2 < V) o[2] = to_out<OutT>(x2);
                            if (v + 3 < V) o[3] = to_out<OutT>(x3);
                        }
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int c = 0; c < kSCT; ++c) acc[rt][c] = (f32x16){0};
            next_bias(r + 1);
            return;
        }
        if constexpr (TERMS == 1 && !LSE && !TRN) {
            // Interior tiles of the single-term mode leave through a per-wave LDS stage as whole 128-byte lines: in the
            // C/D layout a lane holds ONE column, so a direct store moves 2 or 4 bytes per lane (64 store instructions per
            // round and wave, 64-byte pieces of lines); staged, a lane stores 16 bytes of 8 (bf16 / f16) or 4 (fp32)
            // consecutive columns and one instruction covers 8 or 4 full rows of the wave's 64 columns.  Measured at the
            // B = 8 BASELINE slice, bf16 logits: 12.1 -> see DESIGN.md section 6.
            constexpr int ROWB = kSCT * 32 * (int)sizeof(OutT);            // bytes of the wave's 64 columns in one row
            constexpr int SROW = ROWB + 16;                                 // padded stage row (bank spread)
            constexpr int SR = OCC == 2 ? (sizeof(OutT) == 4 ? 8 : 16) : 32; // rows staged per pass
            constexpr int GP = SR / 8;                                      // accumulator row groups (of 8 rows) per pass
            if (full && stage_ok) {
                char *stg = stage + (size_t)wave * SR * SROW;
#ifdef WR_X_NOEPI
                {                                           // experiment: the k-loop without its epilogue
                    float sum = 0.f;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int c = 0; c < kSCT; ++c) { sum += acc[rt][c][0] + acc[rt][c][7]; acc[rt][c] = (f32x16){0}; }
                    if (sum == 1.2345e30f) out[0] = to_out<OutT>(sum);
                    next_bias(r + 1);
                    return;
                }
#endif
                float bv[kSCT];
#pragma unroll
                for (int c = 0; c < kSCT; ++c) bv[c] = OCC == 2 ? bvr[c] : bias_s[(pr * kSCT + c) * 32 + l31];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int g0 = 0; g0 < 4; g0 += GP) {
#pragma unroll
                        for (int c = 0; c < kSCT; ++c) {
#pragma unroll
                            for (int q = 4 * g0; q < 4 * (g0 + GP); ++q) {
                                const int row = (q & 3) + 8 * ((q >> 2) - g0) + 4 * half;
                                *reinterpret_cast<OutT *>(stg + row * SROW + (c * 32 + l31) * (int)sizeof(OutT)) =
                                    to_out<OutT>(acc[rt][c][q] + bv[c]);
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        constexpr int LPR = ROWB / 16;                      // lanes per row: 8 (16-bit logits) or 16 (fp32)
                        constexpr int RPI = 64 / LPR;                       // rows per store instruction
                        const int rr = lane / LPR, seg = lane - rr * LPR;
#ifdef WR_X_STORE_LOCAL
                        // experiment: every store lands in the same 2.5 MB of the output (no HBM write traffic)
                        char *obase = reinterpret_cast<char *>(out + (size_t)((m0 + 32 * rt + 8 * g0) & 255) * V + (size_t)ct0 * 32) + seg * 16;
#else
                        char *obase = reinterpret_cast<char *>(out + (size_t)(m0 + 32 * rt + 8 * g0) * V + (size_t)ct0 * 32) + seg * 16;
#endif
#pragma unroll
                        for (int i = 0; i < SR / RPI; ++i) {
                            const int row = i * RPI + rr;
                            const u32x4 v = *reinterpret_cast<const u32x4 *>(stg + row * SROW + seg * 16);
#ifdef WR_X_NOSTORE
                            if (v.x == 0x7fc12345u && v.y == 0x7fc12345u)   // experiment: the epilogue without its global stores
#endif
#ifdef WR_X_NT
                            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(obase + (size_t)row * V * sizeof(OutT)));
#else
                            *reinterpret_cast<u32x4 *>(obase + (size_t)row * V * sizeof(OutT)) = v;
#endif
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
#pragma unroll
                    for (int c = 0; c < kSCT; ++c) acc[rt][c] = (f32x16){0};
                }
#if defined(WR_JS_STAMPS) && WR_JS_STAMPS >= 2
                {                                           // level 2: how long until the round's stores have retired
                    unsigned long long d0, d1;
                    WR_JS_NOW(d0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    WR_JS_NOW(d1);
                    js_ld += d1 - d0;
                }
#endif
                next_bias(r + 1);
                return;
            }
        }
#pragma unroll
        for (int c = 0; c < kSCT; ++c) {
            const int col = (ct0 + c) * 32 + l31;
            const bool colin = pr < npairs && col < V;
            const float bv = colin ? (OCC == 2 ? bvr[c] : bias_s[(pr * kSCT + c) * 32 + l31]) : 0.f;
            OutT *__restrict__ ocol = out + (size_t)m0 * V + (colin ? col : 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if (LSE) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        joint_lse_add(rm[rt * 16 + q], rs[rt * 16 + q], acc[rt][c][q] + bv, colin, r == 0 && c == 0);
                }
                if (full) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = 32 * rt + (q & 3) + 8 * (q >> 2) + 4 * half;   // C/D layout of the 32x32 MFMA
                        WR_SPLIT_STORE(to_out<OutT>(acc[rt][c][q] + bv), ocol + (size_t)row * V);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = 32 * rt + (q & 3) + 8 * (q >> 2) + 4 * half;
                        if (m0 + row < M && colin) WR_SPLIT_STORE(to_out<OutT>(acc[rt][c][q] + bv), ocol + (size_t)row * V);
                    }
                }
                acc[rt][c] = (f32x16){0};
            }
        }
        next_bias(r + 1);
    };

#ifdef WR_JS_STAMPS
#define WR_JS_PHASE(P, Q)                                                                                                \
    WR_JS_NOW(js_b);                                                                                                      \
    mfma_set(Q##bh, Q##bl, P##bh, P##bl);                                                                                 \
    WR_JS_NOW(js_c);                                                                                                      \
    {                                                                                                                     \
        const bool first_ = cc0 == 0, last_ = cc0 + 1 == cpr;                                                             \
        finish_round();                                                                                                   \
        WR_JS_NOW(js_d);                                                                                                  \
        js_mm += js_c - js_b;                                                                                             \
        if (first_) js_mm_first += js_c - js_b;                                                                           \
        if (last_) { js_ep += js_d - js_c; js_ep_n += 1; }                                                                \
    }
#else
#define WR_JS_PHASE(P, Q)                                                                                                \
    mfma_set(Q##bh, Q##bl, P##bh, P##bl);                                                                                 \
    finish_round();                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);
#endif
    for (int ci = 0; ci < total; ci += 3) {
        WR_JS_PHASE(r, p)
        if (ci + 1 >= total) break;
        WR_JS_PHASE(p, q)
        if (ci + 2 >= total) break;
        WR_JS_PHASE(q, r)
    }
#undef WR_JS_PHASE
#ifdef WR_JS_STAMPS
    WR_JS_NOW(js_t2);
    WR_JS_NOW_RT(js_r2);
    if (lane == 0 && blockIdx.x % 16 == 0 && blockIdx.x / 16 < kJsWgs * kSWaves / NW) {
        unsigned long long *o = g_js + ((size_t)(blockIdx.x / 16) * NW + wave) * kJsPts;
        o[0] = js_t0; o[1] = js_t1; o[2] = js_t2; o[3] = js_r0; o[4] = js_r2; o[5] = js_ld; o[6] = js_mm; o[7] = js_ep;
        o[8] = js_ep_n; o[9] = js_mm_first; o[10] = (unsigned long long)total; o[11] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
#endif
    if (LSE) {
        __syncthreads();                                    // every wave is done with the activation images: reuse them
        joint_lse_finish<NW>(lse, reinterpret_cast<float *>(lds_s), rm, rs, llens, tlens,
                                  reinterpret_cast<const float *>(out), m0, M, T, U1, V);
    }
}

// ------------------------------------------------------------- backward, dZ --
// dZ[m, j] = (sum_v dY[m, v] * W[v, j]) * (1 - H[m, j]^2),  H = tanh(ep + pp) recomputed (exact tanhf).
// The reduction runs over the vocabulary in double steps of 32: lane half h reads 16 consecutive floats
// dY[row][32 d + 16 h ...] (64 bytes; a row's two halves make one 128-byte line), splits them into bf16 hi / lo in
// registers and feeds two MFMAs (elements 0-7, then 8-15); W is re-laid once per call into fragments that follow the
// same k permutation (k = 32 d + 16 h + 8 t + e), hi and lo images.  Kernel: joint_bwd_dz_split128_kernel below
// (128 cells per workgroup, a wave owns 32 of them and all J columns, the W fragments of a step staged once per
// workgroup in LDS; the first, 64-cell tiling of round 1 was 6-15 % slower and has been removed).
constexpr int kZWaves = 4;
constexpr int kZCT = 4;          // column tiles per wave: 4 waves x 4 x 32 = 512 = the largest join_dim

// frag[((jt * D + d) * 2 + t) * 64 + l][e] = W[32 d + 16 (l >> 5) + 8 t + e][jt * 32 + (l & 31)]   (zero outside V x J)
__global__ void split_w_dz_kernel(const float *__restrict__ w, int V, int J, int D, int n_jt, unsigned short *__restrict__ wh,
                                  unsigned short *__restrict__ wl)
{
    const long total = (long)n_jt * D * 2 * 64;
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
        const int l = (int)(f & 63);
        const long r = f >> 6;
        const int t = (int)(r & 1);
        const long jd = r >> 1;
        const int d = (int)(jd % D), jt = (int)(jd / D);
        const int j = jt * 32 + (l & 31), v0 = 32 * d + 16 * (l >> 5) + 8 * t;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (j < J && v0 + e < V) ? w[(size_t)(v0 + e) * J + j] : 0.f;
            unsigned hi, lo;
            split_bf16(x, hi, lo);
            wh[f * 8 + e] = (unsigned short)hi;
            wl[f * 8 + e] = (unsigned short)lo;
        }
    }
}

// eight floats -> bf16x8 hi (and lo) fragments
__device__ __forceinline__ void split8(const f32x4 &a, const f32x4 &b, bf16x8 &hi, bf16x8 &lo, bool want_lo)
{
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split_pair(a[0], a[1], h0, l0);
    split_pair(a[2], a[3], h1, l1);
    split_pair(b[0], b[1], h2, l2);
    split_pair(b[2], b[3], h3, l3);
    hi = __builtin_bit_cast(bf16x8, (u32x4){h0, h1, h2, h3});
    if (want_lo) lo = __builtin_bit_cast(bf16x8, (u32x4){l0, l1, l2, l3});
}

// dZ, second tiling: the W fragments of a step are needed by every row tile, so here the waves split the ROWS (a
// workgroup owns 128 cells, a wave 32 of them and all J columns: 16 accumulator tiles = 256 registers) and the
// fragments of one 16-deep step (hi and lo of every column tile, 32 KB at J = 512) are staged once per workgroup
// in LDS (three stages, two register sets: loads run two steps ahead of their LDS write, as in the weight-gradient
// kernel).  Half the L2 traffic for W per cell of the first tiling.
constexpr int kZM2 = 128;        // cells per workgroup
constexpr int kZStages = 3;

// FULL: all 16 column tiles are in use (J = 512, the shipped join_dim): no per-tile guards in the k-loop
// GT: dtype of the logits gradient -- float, or __bf16 (the AMP step: the loss hands back a bf16 gradient; its values ARE
// their own hi parts, so a lane's 16 values of a double step are two ready-made MFMA operands: half the bytes, no
// conversion pass, no split arithmetic, and the lo term vanishes)
template <int TERMS, bool FULL, typename GT = float>
__global__ __launch_bounds__(256) void joint_bwd_dz_split128_kernel(
    const GT *__restrict__ gout /* [M, V] */, const float *__restrict__ ep, const float *__restrict__ pp,
    const u32x4 *__restrict__ wh, const u32x4 *__restrict__ wl, const int32_t *__restrict__ llens,
    const int32_t *__restrict__ tlens, int B, int T, int U1, int J, int V, int D, int n_jt, int act,
    float *__restrict__ dz /* [M, J] */, float *__restrict__ hout /* [M, J] or null */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dz_lds[];
    // stage layout: [kZStages][2 images][16 column tiles][64 lanes] x 16 bytes
    u32x4 *stage = reinterpret_cast<u32x4 *>(dz_lds);
    long *row_e = reinterpret_cast<long *>(dz_lds + (size_t)kZStages * 2 * 16 * 64 * 16);
    long *row_p = row_e + kZM2;
    int *row_ok = reinterpret_cast<int *>(row_p + kZM2);
    const long M = (long)B * T * U1;
    const long m0 = (long)blockIdx.x * kZM2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    int valid = 0;
    if (tid < kZM2) {
        const long m = m0 + tid < M ? m0 + tid : M - 1;
        const long bt = m / U1;
        const int u = (int)(m - bt * U1);
        const long b = bt / T;
        valid = m0 + tid < M;
        if (valid && llens != nullptr && tlens != nullptr) {
            const int t = (int)(bt - b * T);
            valid = (t < llens[b]) && (u <= tlens[b]);
        }
        row_e[tid] = bt * J;
        row_p[tid] = (b * U1 + u) * J;
        row_ok[tid] = valid;
    }
    const bool any = __syncthreads_or(valid);
    if (!any) {
        for (int i = tid; i < kZM2 * J; i += 256) {
            const long m = m0 + i / J;
            if (m < M) {
                dz[(size_t)m * J + i % J] = 0.f;
                if (hout) hout[(size_t)m * J + i % J] = 0.f;
            }
        }
        return;
    }

    f32x16 acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = (f32x16){0};

    const long mrow = m0 + 32 * wave + l31 < M ? m0 + 32 * wave + l31 : M - 1;
    constexpr bool G16 = !std::is_same<GT, float>::value;
    const GT *__restrict__ arow = gout + (size_t)mrow * V + 16 * half;
    const int Dfull = V / 32;
    const int steps = 2 * D;                                // 16-deep steps, two per double step

    // W staging: chunk c = tid + 256 q (q < 8) is lane (c & 63) of fragment (c >> 6): fragments 0..15 hi, 16..31 lo
    // (single-term mode stages the hi image only: chunks 0-3)
    constexpr int NQ = TERMS == 3 ? 8 : 4;
    struct WRegs { u32x4 v[8]; };
    auto wload = [&](int s, WRegs &z) {
        const int ss = s < steps ? s : steps - 1;
        const int d = ss >> 1, t = ss & 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = tid + 256 * q, fr = c >> 6, ln = c & 63;
            int jt = fr & 15;
            jt = jt < n_jt ? jt : n_jt - 1;
            const size_t f = (((size_t)jt * D + d) * 2 + t) * 64 + ln;
            z.v[q] = (fr < 16) ? wh[f] : (TERMS == 3 ? wl[f] : wh[f]);
        }
    };
    auto wwrite = [&](int s, const WRegs &z) {
        u32x4 *st = stage + (size_t)(s % kZStages) * 2 * 16 * 64;
#pragma unroll
        for (int q = 0; q < NQ; ++q) st[tid + 256 * q] = z.v[q];
    };
    // the same, one fragment chunk at a time and without clamps (every step < steps: the fragment index is linear in the
    // step, ((jt * D + d) * 2 + t) = jt * 2D + s), for the main loop, which spreads a step's 8 LDS writes and 8 global loads
    // over the 16 column tiles of the step being computed
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto wload_fast = [&](int s, WRegs &z, int q) {
        int jt = (wave_u + 4 * q) & 15;
        jt = (FULL || jt < n_jt) ? jt : n_jt - 1;
        const u32x4 *img = (q < 4 || TERMS != 3) ? wh : wl;
        z.v[q] = img[((size_t)jt * 2 * D + s) * 64 + lane];
    };
    auto wwrite_fast = [&](int s, const WRegs &z, int q) {
        stage[(size_t)(s % kZStages) * 2 * 16 * 64 + tid + 256 * q] = z.v[q];
    };
    // dY of this lane's row: 16 floats per double step (elements 0-7 feed the even step, 8-15 the odd one)
    struct ARegs32 { f32x4 a[4]; };
    struct ARegs16 { u32x4 a[2]; };                         // elements 0-7 | 8-15 as stored
    using ARegs = typename std::conditional<G16, ARegs16, ARegs32>::type;
    auto aload = [&](int d, ARegs &z) {
        const int dd = d < Dfull ? d : (Dfull > 0 ? Dfull - 1 : 0);
        if constexpr (G16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) z.a[i] = *reinterpret_cast<const u32x4 *>(arow + 32 * dd + 8 * i);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) z.a[i] = *reinterpret_cast<const f32x4 *>(arow + 32 * dd + 4 * i);
        }
    };
    auto atail = [&](ARegs &z) {                            // the row's tail (V % 32 values): guarded scalar reads
        if constexpr (G16) {
            const unsigned short *arow16 = reinterpret_cast<const unsigned short *>(arow);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int v = 32 * Dfull + 16 * half + 8 * i + 2 * e;
                    const unsigned x0 = v < V ? arow16[32 * Dfull + 8 * i + 2 * e] : 0u;
                    const unsigned x1 = v + 1 < V ? arow16[32 * Dfull + 8 * i + 2 * e + 1] : 0u;
                    z.a[i][e] = x0 | (x1 << 16);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int v = 32 * Dfull + 16 * half + 4 * i + e;
                    z.a[i][e] = v < V ? arow[32 * Dfull + 4 * i + e] : 0.f;
                }
        }
    };
    auto compute = [&](int s, const ARegs &z, WRegs &zw, auto fast_tag) {
        constexpr bool fast = decltype(fast_tag)::value;
        const int t = s & 1;
        if (!fast) {
            wwrite(s + 2, zw);
            wload(s + 4, zw);
            __builtin_amdgcn_sched_barrier(0);
        }
        bf16x8 ah, al;
        if constexpr (G16) ah = __builtin_bit_cast(bf16x8, z.a[t]);
        else split8(z.a[2 * t], z.a[2 * t + 1], ah, al, TERMS == 3);
        const u32x4 *st = stage + (size_t)(s % kZStages) * 2 * 16 * 64 + lane;
        // the fragments of column tile c + 2 are requested before the MFMAs of tile c (the compiler issued each tile's two
        // reads directly in front of its three MFMAs: an LDS round trip per 96 matrix-core cycles, the reason these kernels
        // ran the matrix cores 40 % busy); tiles past n_jt read a valid stage address and feed nothing
        u32x4 rh[3], rl[3];
        rh[0] = st[0];
        rh[1] = st[64];
        if (TERMS == 3) { rl[0] = st[16 * 64]; rl[1] = st[17 * 64]; }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c + 2 < 16) {
                rh[(c + 2) % 3] = st[(c + 2) * 64];
                if (TERMS == 3) rl[(c + 2) % 3] = st[(16 + c + 2) * 64];
            }
            if (fast) {                                      // staging of step s + 2 (tiles 0-7), loads of step s + 4 (tiles 8-15)
                if (c < NQ) wwrite_fast(s + 2, zw, c);
                else if (c >= 8 && c < 8 + NQ) wload_fast(s + 4, zw, c - 8);
            }
            if (FULL || c < n_jt) {
                const bf16x8 bhv = __builtin_bit_cast(bf16x8, rh[c % 3]);
                if (TERMS == 3) {
                    const bf16x8 blv = __builtin_bit_cast(bf16x8, rl[c % 3]);
                    if (!G16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhv, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blv, acc[c], 0, 0, 0);
                }
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhv, acc[c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    WRegs w0, w1;
    ARegs a0, a1;
    wload(0, w0); wwrite(0, w0);
    wload(1, w0); wwrite(1, w0);
    wload(2, w0);
    wload(3, w1);
    if (Dfull > 0) aload(0, a0); else atail(a0);
    if (1 < Dfull) aload(1, a1); else if (1 < D) atail(a1);
    // the loop handles two double steps (four 16-deep steps) per trip; D is padded to even below by repeating the
    // last double step with zero fragments (W is zero past V, and the A tail is zero there too)
    // main loop: all four steps of a trip and everything they stage / load (up to step 2d + 7) lie inside the step range
    // and inside V: staging rides between the MFMAs, nothing is clamped; then the general loop for the last trips (two loops
    // in sequence: a branch inside one loop would duplicate the MFMA body and unsettle the accumulator allocation)
    int d = 0;
    for (; d + 3 < Dfull && 2 * d + 7 < steps; d += 2) {
        __syncthreads();
        compute(2 * d, a0, w0, std::true_type{});
        __syncthreads();
        compute(2 * d + 1, a0, w1, std::true_type{});
        aload(d + 2, a0);
        __syncthreads();
        compute(2 * d + 2, a1, w0, std::true_type{});
        __syncthreads();
        compute(2 * d + 3, a1, w1, std::true_type{});
        aload(d + 3, a1);
    }
    for (; d < D; d += 2) {
        // ---- double step d (registers a0) ----
        __syncthreads();
        compute(2 * d, a0, w0, std::false_type{});
        __syncthreads();
        compute(2 * d + 1, a0, w1, std::false_type{});
        if (d + 2 < Dfull) aload(d + 2, a0); else if (d + 2 < D) atail(a0);
        // ---- double step d + 1 (registers a1) ----
        __syncthreads();
        if (d + 1 < D) compute(2 * d + 2, a1, w0, std::false_type{});
        else { wwrite(2 * d + 4, w0); wload(2 * d + 6, w0); }
        __syncthreads();
        if (d + 1 < D) compute(2 * d + 3, a1, w1, std::false_type{});
        else { wwrite(2 * d + 5, w1); wload(2 * d + 7, w1); }
        if (d + 3 < Dfull) aload(d + 3, a1); else if (d + 3 < D) atail(a1);
    }

    // epilogue: dZ = dH * (1 - H^2), H recomputed per element; padded cells give zeros
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * half;
        const long m = m0 + row;
        if (m >= M) continue;
        const bool ok = row_ok[row] != 0;
        const float *__restrict__ e = ep + row_e[row];
        const float *__restrict__ p = pp + row_p[row];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int k = c * 32 + l31;
            if (c >= n_jt || k >= J) continue;
            float h, dh;
            act_value_grad(act, e[k] + p[k], h, dh);
            dz[(size_t)m * J + k] = ok ? acc[c][q] * dh : 0.f;
            if (hout) hout[(size_t)m * J + k] = ok ? h : 0.f;
        }
    }
}

// ---------------------------------------------------- backward, weight gradient --
// dW[v, j] = sum_m dY[m, v] * H[m, j],  db[v] = sum_m dY[m, v]     (M = B*T*U1 lattice cells, padded cells excluded)
// A reduction over millions of cells into a V x J matrix.  One workgroup (4 waves = 2 v-halves x 2 j-halves, one
// per SIMD) owns a 256 x 256 block of dW and one of `parts` contiguous cell ranges; a wave holds 4 x 4 accumulator
// tiles (256 registers).  Both operands are cell-major as stored: a staging thread loads an 8-cell x 4-column patch
// of dY (waves 0-1) or H (waves 2-3) with coalesced 16-byte loads, masks it, splits it to bf16 hi / lo ONCE for
// the whole workgroup and writes four ready-made MFMA fragments per image into LDS (three stages; global loads
// run two steps ahead).  The four columns of a patch become rows of four interleaved MFMA tiles (tile t holds
// columns 4*i + t), so the contraction index stays along the fragment and nothing is transposed; in the k-loop a
// lane only reads its fragments (16 ds_read_b128 per 48 MFMAs).
// The `parts` partial blocks are summed by a second, deterministic kernel (no float atomics).
constexpr int kWB = 256;         // dW block edge (v and j) per workgroup
constexpr int kWStages = 3;

// one byte per lattice cell: 1 inside [0, T_b) x [0, U_b], 0 in the padded region
__global__ void cell_mask_kernel(const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens, int T, int U1, long M,
                                 unsigned char *__restrict__ mask)
{
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        const long bt = m / U1;
        const int u = (int)(m - bt * U1);
        const long b = bt / T;
        const int t = (int)(bt - b * T);
        mask[m] = (t < llens[b]) && (u <= tlens[b]);
    }
}

// GT: dtype of the logits gradient (float, or __bf16 in the AMP step: a dY patch is then 4 x 2 bytes per cell)
template <int TERMS, typename GT = float>
__global__ __launch_bounds__(512) void joint_bwd_dw_split_kernel(
    const GT *__restrict__ gout /* [M, V] */, const float *__restrict__ h /* [M, J] */,
    const unsigned char *__restrict__ mask /* [M] or null */, long M, int V, int J, int n_vs, int n_js, long rows_per_part,
    float *__restrict__ part_dw /* [parts][V][J] */, float *__restrict__ part_db /* [parts][V] */)
{
    // stage layout (16-byte units): [kWStages][operand 2][hi/lo 2][cell group 2][column-in-patch 4][64 patches] --
    // ready-made MFMA fragments: the staging thread converts its 8-cell x 4-column patch once for every wave
    extern __shared__ __attribute__((aligned(16))) u32x4 stage[];
    constexpr int kStageUnits = 2 * 2 * 2 * 4 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // eight waves, two per SIMD (waves w and w+4 share one): wave = (v-half, j-quarter); the first four also stage,
    // so every SIMD holds a staging wave and a compute-only wave whose MFMAs fill the staging gaps
    const int vh = wave >> 2, jq = wave & 3;
    const bool stager = tid < 256;
    const int tiles = n_vs * n_js;
    const int part = blockIdx.x / tiles, tile = blockIdx.x - part * tiles;
    const int vs = tile / n_js, js = tile - vs * n_js;
    const int v0 = vs * kWB, j0 = js * kWB;
    const long mb = (long)part * rows_per_part;
    const long me = mb + rows_per_part < M ? mb + rows_per_part : M;
    const int steps = me > mb ? (int)((me - mb + 15) / 16) : 0;

    // staging map: thread -> operand (waves 0-1: dY, 2-3: H), cell group mg (8 cells), float4 column c4
    // (mg and op are the same for a whole wave: scalar, so that the two operands' load paths are real branches -- as
    // per-lane selects the 8-byte and the 16-byte load of a row shared their destination registers and waited on each other)
    const int c4 = tid & 63, mg = __builtin_amdgcn_readfirstlane((tid >> 6) & 1), op = __builtin_amdgcn_readfirstlane((tid >> 7) & 1);
    const bool col_in = op == 0 ? (v0 + 4 * c4 < V) : (j0 + 4 * c4 < J);   // V, J multiples of 4: wholly in or out
    const int ld = op == 0 ? V : J;
    constexpr bool G16 = !std::is_same<GT, float>::value;
    const float *__restrict__ gsrc = (op == 0 ? (G16 ? h : reinterpret_cast<const float *>(gout) + v0) : h + j0) + (col_in ? 4 * c4 : 0);
    const GT *__restrict__ gsrc16 = gout + v0 + (col_in ? 4 * c4 : 0);          // G16: the dY patches
    struct Regs { f32x4 x[8]; };                            // G16 dY patches: four bf16 as loaded in x[e][0..1] (widening them
                                                            // at the load would wait for it two steps early)
    auto gload = [&](int s, Regs &z) {
        if (!stager) return;
        if (G16 && op == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                long m = mb + 16L * s + 8 * mg + e;
                m = m < me ? m : me - 1;
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 r = *reinterpret_cast<const u32x2 *>(gsrc16 + (size_t)m * ld);
                z.x[e] = __builtin_bit_cast(f32x4, __builtin_shufflevector(r, r, 0, 1, -1, -1));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                long m = mb + 16L * s + 8 * mg + e;
                m = m < me ? m : me - 1;
                z.x[e] = *reinterpret_cast<const f32x4 *>(gsrc + (size_t)m * ld);
            }
        }
    };
    f32x4 dbacc = (f32x4){0, 0, 0, 0};
    auto lwrite = [&](int s, Regs &z) {                              // masks, accumulates db, converts, stores the fragments
        if (!stager) return;
        const f32x4 zero = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (G16 && op == 0) {                                    // four bf16 -> four floats (exact: a shift)
                const u32x4 raw = __builtin_bit_cast(u32x4, z.x[e]);
                const unsigned rx = raw[0], ry = raw[1];
                z.x[e] = (f32x4){__builtin_bit_cast(float, rx << 16), __builtin_bit_cast(float, rx & 0xffff0000u),
                                 __builtin_bit_cast(float, ry << 16), __builtin_bit_cast(float, ry & 0xffff0000u)};
            }
            const long m = mb + 16L * s + 8 * mg + e;
            bool on = m < me && col_in;
            if (on && mask != nullptr) on = mask[m] != 0;
            z.x[e] = on ? z.x[e] : zero;
            if (op == 0) dbacc += z.x[e];
        }
        u32x4 *st = stage + (size_t)(s % kWStages) * kStageUnits + (size_t)(op * 2) * 2 * 4 * 64 + (size_t)mg * 4 * 64 + c4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bf16x8 hi, lo;
            const f32x4 lo4 = (f32x4){z.x[0][t], z.x[1][t], z.x[2][t], z.x[3][t]};
            const f32x4 hi4 = (f32x4){z.x[4][t], z.x[5][t], z.x[6][t], z.x[7][t]};
            split8(lo4, hi4, hi, lo, TERMS == 3);
            st[t * 64] = __builtin_bit_cast(u32x4, hi);
            if (TERMS == 3) st[2 * 4 * 64 + t * 64] = __builtin_bit_cast(u32x4, lo);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x16){0};

    // B side: a wave covers 64 columns = 16 patches x 4; tile u takes patch columns 2u and 2u+1: lane l31 reads
    // column-in-patch 2u + (l31 >> 4) of patch 16 jq + (l31 & 15)
    const int bpatch = 16 * jq + (l31 & 15), bsub = l31 >> 4;
    auto compute = [&](int s) {
        const u32x4 *sa = stage + (size_t)(s % kWStages) * kStageUnits + (size_t)half * 4 * 64 + 32 * vh + l31;
        const u32x4 *sb = stage + (size_t)(s % kWStages) * kStageUnits + (size_t)2 * 2 * 4 * 64 + (size_t)half * 4 * 64 + bpatch;
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ah[t] = __builtin_bit_cast(bf16x8, sa[t * 64]);
            if (TERMS == 3) al[t] = __builtin_bit_cast(bf16x8, sa[2 * 4 * 64 + t * 64]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, sb[(2 * u + bsub) * 64]);
            bf16x8 bl;
            if (TERMS == 3) bl = __builtin_bit_cast(bf16x8, sb[2 * 4 * 64 + (2 * u + bsub) * 64]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (TERMS == 3) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl, acc[t][u], 0, 0, 0);
                }
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh, acc[t][u], 0, 0, 0);
            }
        }
    };

    if (steps > 0) {
        // global loads run two steps ahead of their LDS write (two register sets), the LDS stages one more.  The
        // loop runs an even number of steps (a step past the range stages zeros) so that it has a single exit.
        const int steps2 = (steps + 1) & ~1;
        Regs r0, r1;
        gload(0, r0);
        lwrite(0, r0);
        gload(1, r0);
        lwrite(1, r0);
        gload(2, r0);
        gload(3, r1);
        for (int s = 0; s < steps2; s += 2) {
            __syncthreads();                                         // stage s visible; stage s+2's buffer is free
            lwrite(s + 2, r0);
            gload(s + 4, r0);
            __builtin_amdgcn_sched_barrier(0);
            compute(s);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            lwrite(s + 3, r1);
            gload(s + 5, r1);
            __builtin_amdgcn_sched_barrier(0);
            compute(s + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // partial block of dW: tile (t, u) element (row i, col c) is dW[v0 + 128 vh + 4 i + t][j0 + 4 (16 jq + (c & 15)) + 2 u + (c >> 4)]
    float *__restrict__ pw = part_dw + (size_t)part * V * J;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = (q & 3) + 8 * (q >> 2) + 4 * half;
            const int v = v0 + 128 * vh + 4 * i + t;
            if (v >= V) continue;
            const int j = j0 + 4 * bpatch + bsub;                    // + 2 u
            if (j - bsub < J) {                                      // J % 4 == 0: a patch is wholly in or out
                pw[(size_t)v * J + j] = acc[t][0][q];
                pw[(size_t)v * J + j + 2] = acc[t][1][q];
            }
        }
    // db: the two cell groups of every float4 column -> one sum (j-block 0 only; the dY patches sit in waves 0-1)
    if (js == 0) {
        __syncthreads();
        f32x4 *red = reinterpret_cast<f32x4 *>(stage);
        if (op == 0) red[tid] = dbacc;
        __syncthreads();
        if (tid < 64 && col_in) {
            const f32x4 t4 = red[tid] + red[tid + 64];
            *reinterpret_cast<f32x4 *>(part_db + (size_t)part * V + v0 + 4 * tid) = t4;
        }
    }
}

// The same 256 x 256 block tiling with the EXACT fp32 MFMA (v_mfma_f32_32x32x2_f32): the default weight gradient.
// Four waves (2 v-halves x 2 j-halves), 4 x 4 accumulator tiles each; a 16-cell step of both operands is staged as
// fp32 ([16][256] each, three LDS stages, global loads two steps ahead); per pair of cells a lane reads one float4
// of each operand (its 4 interleaved tiles) and issues 16 MFMAs -- 128 MFMAs of 64 cycles per step against 16
// ds_read_b128, so the k-loop is matrix-core bound.  Padded cells are excluded through the lengths.
// MASKED = false (no length arrays: the fused joiner + loss node, whose gradient is zero in padded cells): the main loop
// runs the branch-free staging path; MASKED = true keeps the per-row validity arithmetic of the general path throughout.
template <bool MASKED>
__global__ __launch_bounds__(256) void joint_bwd_dw_block_kernel(
    const float *__restrict__ gout /* [M, V] */, const float *__restrict__ h /* [M, J] */,
    const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens, int T, int U1, long M, int V, int J, int n_vs,
    int n_js, long rows_per_part, float *__restrict__ part_dw /* [parts][V][J] */, float *__restrict__ part_db /* [parts][V] */)
{
    extern __shared__ __attribute__((aligned(16))) float fstage[];    // [kWStages][2 operands][16][kWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int vh = wave >> 1, jh = wave & 1;
    const int tiles = n_vs * n_js;
    const int part = blockIdx.x / tiles, tile = blockIdx.x - part * tiles;
    const int vs = tile / n_js, js = tile - vs * n_js;
    const int v0 = vs * kWB, j0 = js * kWB;
    const long mb = (long)part * rows_per_part;
    const long me = mb + rows_per_part < M ? mb + rows_per_part : M;
    const int steps = me > mb ? (int)((me - mb + 15) / 16) : 0;

    // staging: float4 column c4, rows r4 + 4 i.  r4 is the wave index: made scalar, so that a row's (utterance, frame,
    // label position) and its validity are scalar arithmetic and scalar loads, off the vector pipe
    const int c4 = tid & 63, r4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool a_in = v0 + 4 * c4 < V, b_in = j0 + 4 * c4 < J;
    const float *__restrict__ ga = gout + (a_in ? v0 + 4 * c4 : 0);
    const float *__restrict__ gb = h + (b_in ? j0 + 4 * c4 : 0);
    struct Regs { f32x4 a[4], b[4]; int on; };
    // Fast path (no lengths, all 16 cells of the step inside the part's range): a wave-uniform base that advances with s
    // plus a per-thread offset fixed for the whole kernel, no masks -- VALU work of the same wave does not hide behind its
    // MFMAs (tools/micro/mfma_valu_mix.hip), so the general path's validity arithmetic and selects stay out of the main
    // loop.  Columns v >= V / j >= J read a clamped address and produce rows / columns of dW that are never stored.
    unsigned aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aoff[i] = (unsigned)(((long)(r4 + 4 * i) * V + (a_in ? v0 + 4 * c4 : 0)) * sizeof(float));
        boff[i] = (unsigned)(((long)(r4 + 4 * i) * J + (b_in ? j0 + 4 * c4 : 0)) * sizeof(float));
    }
    const char *__restrict__ abase = reinterpret_cast<const char *>(gout + (size_t)mb * V);
    const char *__restrict__ bbase = reinterpret_cast<const char *>(h + (size_t)mb * J);
    auto gload_fast = [&](int s, Regs &z, int i) {
        z.a[i] = *reinterpret_cast<const f32x4 *>(abase + (size_t)s * 16 * V * sizeof(float) + aoff[i]);
        z.b[i] = *reinterpret_cast<const f32x4 *>(bbase + (size_t)s * 16 * J * sizeof(float) + boff[i]);
    };
    auto gload = [&](int s, Regs &z) {
        z.on = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long m = mb + 16L * s + r4 + 4 * i;
            // validity of the row, evaluated here -- two steps ahead of its use -- so that the length loads travel with
            // the operand loads
            bool on = m < me;
            if (MASKED && on) {                                       // M < 2^31 (checked by the caller): 32-bit divisions
                const unsigned mu = (unsigned)m, bt = mu / (unsigned)U1, u = mu - bt * (unsigned)U1;
                const unsigned b = bt / (unsigned)T, t = bt - b * (unsigned)T;
                on = ((int)t < llens[b]) && ((int)u <= tlens[b]);
            }
            z.on |= (on ? 1 : 0) << i;
            m = m < me ? m : me - 1;
            z.a[i] = *reinterpret_cast<const f32x4 *>(ga + (size_t)m * V);
            z.b[i] = *reinterpret_cast<const f32x4 *>(gb + (size_t)m * J);
        }
    };
    f32x4 dbacc = (f32x4){0, 0, 0, 0};
    auto lwrite_fast = [&](int s, const Regs &z, int i) {
        float *sa = fstage + (size_t)(s % kWStages) * 2 * 16 * kWB;
        float *sb = sa + 16 * kWB;
        dbacc += z.a[i];
        *reinterpret_cast<f32x4 *>(sa + (r4 + 4 * i) * kWB + 4 * c4) = z.a[i];
        *reinterpret_cast<f32x4 *>(sb + (r4 + 4 * i) * kWB + 4 * c4) = z.b[i];
    };
    auto lwrite = [&](int s, const Regs &z) {
        float *sa = fstage + (size_t)(s % kWStages) * 2 * 16 * kWB;
        float *sb = sa + 16 * kWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool on = (z.on >> i) & 1;
            const f32x4 zero = (f32x4){0, 0, 0, 0};
            const f32x4 av = (on && a_in) ? z.a[i] : zero;
            const f32x4 bv = (on && b_in) ? z.b[i] : zero;
            dbacc += av;
            *reinterpret_cast<f32x4 *>(sa + (r4 + 4 * i) * kWB + 4 * c4) = av;
            *reinterpret_cast<f32x4 *>(sb + (r4 + 4 * i) * kWB + 4 * c4) = bv;
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x16){0};

    // Step s: 8 pairs of cells; per pair a lane reads one float4 of each operand and issues 16 MFMAs.  As in the dZ block
    // kernel below: the fragments of the next pair (and, at the end of a step, of the next step, which became visible at the
    // barrier in front of this one) are requested before the current pair's MFMAs, and in the fast path the staging of step
    // s + 2 (pairs 0-3) and the loads of step s + 4 (pairs 4-7) ride between the MFMA groups.
    const int frag_a = half * kWB + 128 * vh + 4 * l31, frag_b = 16 * kWB + half * kWB + 128 * jh + 4 * l31;
    auto compute = [&](int s, Regs &z, f32x4 &pa, f32x4 &pb, auto fast_tag) {
        constexpr bool fast = decltype(fast_tag)::value;
        const float *sa = fstage + (size_t)(s % kWStages) * 2 * 16 * kWB + frag_a;
        const float *sb = fstage + (size_t)(s % kWStages) * 2 * 16 * kWB + frag_b;
        const float *sn = fstage + (size_t)((s + 1) % kWStages) * 2 * 16 * kWB;
        if (!fast) {
            lwrite(s + 2, z);
            gload(s + 4, z);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {                              // cells 2 kp + half
            f32x4 na, nb;
            if (kp < 7) {
                na = *reinterpret_cast<const f32x4 *>(sa + 2 * (kp + 1) * kWB);
                nb = *reinterpret_cast<const f32x4 *>(sb + 2 * (kp + 1) * kWB);
            } else {
                na = *reinterpret_cast<const f32x4 *>(sn + frag_a);
                nb = *reinterpret_cast<const f32x4 *>(sn + frag_b);
            }
            if (fast) {
                if (kp < 4) lwrite_fast(s + 2, z, kp);
                else gload_fast(s + 4, z, kp - 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[t], pb[u], acc[t][u], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            pa = na;
            pb = nb;
        }
    };

    if (steps > 0) {
        const int steps2 = (steps + 1) & ~1;                          // single loop exit (a step past the range stages zeros)
        Regs r0, r1;
        gload(0, r0);
        lwrite(0, r0);
        gload(1, r0);
        lwrite(1, r0);
        gload(2, r0);
        gload(3, r1);
        __syncthreads();                                             // steps 0 and 1 visible
        f32x4 pa = *reinterpret_cast<const f32x4 *>(fstage + frag_a);
        f32x4 pb = *reinterpret_cast<const f32x4 *>(fstage + frag_b);
        // main loop: no lengths, and the steps staged / loaded during the two computes (s + 2 .. s + 5) lie wholly inside the
        // part's range; then the general loop for the tail (two loops in sequence: duplicated bodies inside one loop made the
        // allocator shuffle the 256 accumulators between the copies)
        int s = 0;
        if (!MASKED)
            for (; s < steps2 && mb + 16L * (s + 5) + 16 <= me; s += 2) {
                __syncthreads();
                compute(s, r0, pa, pb, std::true_type{});
                __syncthreads();
                compute(s + 1, r1, pa, pb, std::true_type{});
            }
        for (; s < steps2; s += 2) {
            __syncthreads();
            compute(s, r0, pa, pb, std::false_type{});
            __syncthreads();
            compute(s + 1, r1, pa, pb, std::false_type{});
        }
    }

    // tile (t, u) element (row i, col c) is dW[v0 + 128 vh + 4 i + t][j0 + 128 jh + 4 c + u]
    float *__restrict__ pw = part_dw + (size_t)part * V * J;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = (q & 3) + 8 * (q >> 2) + 4 * half;
            const int v = v0 + 128 * vh + 4 * i + t;
            if (v >= V) continue;
            const int j = j0 + 128 * jh + 4 * l31;
            if (j < J) {
                const f32x4 o = (f32x4){acc[t][0][q], acc[t][1][q], acc[t][2][q], acc[t][3][q]};
                *reinterpret_cast<f32x4 *>(pw + (size_t)v * J + j) = o;
            }
        }
    if (js == 0) {
        __syncthreads();
        f32x4 *red = reinterpret_cast<f32x4 *>(fstage);
        red[tid] = dbacc;
        __syncthreads();
        if (tid < 64 && a_in) {
            const f32x4 t4 = (red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192]);
            *reinterpret_cast<f32x4 *>(part_db + (size_t)part * V + v0 + 4 * tid) = t4;
        }
    }
}

// Exact-fp32 activation gradient with the same block tiling: dZ[m, j] = (sum_v dY[m, v] W[v, j]) (1 - H[m, j]^2).
// A workgroup owns 256 cells x 256 join columns (4 waves = 2 x 2, 4 x 4 accumulator tiles each) and walks the
// vocabulary in steps of 16: dY arrives row-major (v contiguous) and is transposed while it is staged (LDS rows of
// 260 floats: the scattered 4-byte writes of a float4 hit 64 distinct banks), W [v][j] is staged as it is; per pair
// of v a lane reads one float4 of each (4 cells, 4 columns) and issues 16 exact-fp32 MFMAs.  W is streamed once per
// 256 cells (the 64-cell kernel in joint.hip streams it once per 64).
constexpr int kZB = 256;         // block edge (cells and join columns)
constexpr int kZApad = kZB + 4;  // transposed dY stage row stride (floats)

// ANY_ACT = false: tanh (the shipped joiner) in the epilogue.  ANY_ACT = true: the epilogue stores the raw product dH and
// joint_dz_act_kernel applies the activation's derivative afterwards (one more pass over the 2.5 GB / 8 utterances of dZ):
// a generic activation switch inside this epilogue cost the 256-accumulator kernel its register allocation -- scratch
// spills in the k-loop, 9.5 instead of 132 TFLOP/s measured.
template <bool ANY_ACT>
__global__ __launch_bounds__(256) void joint_bwd_dz_block_kernel(
    const float *__restrict__ gout /* [M, V] */, const float *__restrict__ ep, const float *__restrict__ pp,
    const float *__restrict__ w /* [V, J] */, const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens, int B,
    int T, int U1, int J, int V, int n_js, int act, float *__restrict__ dz /* [M, J] */, float *__restrict__ hout /* [M,J] or null */)
{
    extern __shared__ __attribute__((aligned(16))) float zstage[];   // [kWStages]{ A^T [16][260], B [16][256] }, row tables
    constexpr int kStageFloats = 16 * kZApad + 16 * kZB;
    long *row_e = reinterpret_cast<long *>(zstage + (size_t)kWStages * kStageFloats);
    long *row_p = row_e + kZB;
    int *row_ok = reinterpret_cast<int *>(row_p + kZB);
    const long M = (long)B * T * U1;
    const int js = blockIdx.x % n_js;
    const long m0 = (long)(blockIdx.x / n_js) * kZB;
    const int j0 = js * kZB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int mh = wave >> 1, jh = wave & 1;

    int valid = 0;
    {
        const long m = m0 + tid < M ? m0 + tid : M - 1;
        const long bt = m / U1;
        const int u = (int)(m - bt * U1);
        const long b = bt / T;
        valid = m0 + tid < M;
        if (valid && llens != nullptr && tlens != nullptr) {
            const int t = (int)(bt - b * T);
            valid = (t < llens[b]) && (u <= tlens[b]);
        }
        row_e[tid] = bt * J;
        row_p[tid] = (b * U1 + u) * J;
        row_ok[tid] = valid;
    }
    const bool any = __syncthreads_or(valid);
    if (!any) {                                             // wholly padded block: zeros, no arithmetic
        for (int i = tid; i < kZB * kZB; i += 256) {
            const long m = m0 + i / kZB;
            const int j = j0 + i % kZB;
            if (m < M && j < J) {
                dz[(size_t)m * J + j] = 0.f;
                if (hout) hout[(size_t)m * J + j] = 0.f;
            }
        }
        return;
    }

    // staging maps.  A: thread -> row (tid >> 2) + 64 i, float4 along v at 4 (tid & 3).  B: thread -> float4 column
    // (tid & 63) of the block, rows (tid >> 6) + 4 i of the step.
    const int arow = tid >> 2, av4 = tid & 3;
    const int bc4 = tid & 63, br4 = tid >> 6;
    const bool b_in = j0 + 4 * bc4 < J;
    const float *__restrict__ gb = w + (b_in ? j0 + 4 * bc4 : 0);
    const int steps = (V + 15) / 16;
    struct Regs { f32x4 a[4], b[4]; };
    // staging in quarters (i = 0..3: one float4 of dY and one of W per thread), so that the main loop can spread a step's
    // 8 global loads and 20 LDS writes over the MFMA groups of the step being computed
    // `piece` 0..2 cuts a quarter once more (dY part / W part), so that a group of 4 MFMAs has at most three memory
    // instructions and their address arithmetic in front of it; piece < 0: all of it
    // Fast path (every step whose 16 v lie inside V): addresses are a wave-uniform base that advances with s plus a
    // per-thread offset fixed for the whole kernel, and nothing is clamped or zeroed -- rows m >= M and columns j >= J read
    // a clamped address and produce accumulator rows / columns the epilogue never stores.  VALU work of the same wave
    // does NOT hide behind its MFMAs (tools/micro/mfma_valu_mix.hip: one VALU group per MFMA costs 14 % with one wave per
    // SIMD), so the bounds logic below is kept out of the common path.
    unsigned aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long m = m0 + arow + 64 * i;
        m = m < M ? m : M - 1;
        aoff[i] = (unsigned)(((m - m0) * V + 4 * av4) * sizeof(float));          // < 256 * V * 4 bytes
        boff[i] = (unsigned)(((long)(br4 + 4 * i) * J + (b_in ? j0 + 4 * bc4 : 0)) * sizeof(float));
    }
    const char *__restrict__ abase = reinterpret_cast<const char *>(gout + (size_t)m0 * V);
    const char *__restrict__ bbase = reinterpret_cast<const char *>(w);
    auto gload_part = [&](int s, Regs &z, int i, int piece = -1, bool fast = false) {
        if (fast) {                                                              // compile-time constant at the call sites
            if (piece < 0 || piece == 0)
                z.a[i] = *reinterpret_cast<const f32x4 *>(abase + (size_t)s * 16 * sizeof(float) + aoff[i]);
            if (piece < 0 || piece == 1)
                z.b[i] = *reinterpret_cast<const f32x4 *>(bbase + (size_t)s * 16 * J * sizeof(float) + boff[i]);
            return;
        }
        const int ss = s < steps ? s : steps - 1;
        if (piece < 0 || piece == 0) {
            const int v = 16 * ss + 4 * av4;
            const int vc = v < V ? v : V - 4;               // V % 4 == 0
            long m = m0 + arow + 64 * i;
            m = m < M ? m : M - 1;
            z.a[i] = *reinterpret_cast<const f32x4 *>(gout + (size_t)m * V + vc);
        }
        if (piece < 0 || piece == 1) {
            int vr = 16 * ss + br4 + 4 * i;
            vr = vr < V ? vr : V - 1;
            z.b[i] = *reinterpret_cast<const f32x4 *>(gb + (size_t)vr * J);
        }
    };
    auto lwrite_part = [&](int s, const Regs &z, int i, int piece = -1, bool fast = false) {
        float *sa = zstage + (size_t)(s % kWStages) * kStageFloats;
        float *sb = sa + 16 * kZApad;
        const int r = arow + 64 * i;
        if (fast) {                                                              // plain stores
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (piece < 0 || (e >> 1) == piece) sa[(4 * av4 + e) * kZApad + r] = z.a[i][e];
            if (piece < 0 || piece == 2) *reinterpret_cast<f32x4 *>(sb + (br4 + 4 * i) * kZB + 4 * bc4) = z.b[i];
            return;
        }
        const f32x4 zero = (f32x4){0, 0, 0, 0};
        if (piece < 0 || piece == 0 || piece == 1) {
            const bool a_in = 16 * s + 4 * av4 < V;
            const f32x4 av = (a_in && m0 + r < M) ? z.a[i] : zero;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (piece < 0 || (e >> 1) == piece) sa[(4 * av4 + e) * kZApad + r] = av[e];
        }
        if (piece < 0 || piece == 2) {
            const bool bon = b_in && 16 * s + br4 + 4 * i < V;
            *reinterpret_cast<f32x4 *>(sb + (br4 + 4 * i) * kZB + 4 * bc4) = bon ? z.b[i] : zero;
        }
    };
    auto gload = [&](int s, Regs &z) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gload_part(s, z, i);
    };
    auto lwrite = [&](int s, const Regs &z) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lwrite_part(s, z, i);
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x16){0};

    // Step s: 8 pairs of v; per pair a lane reads one float4 of each operand and issues 16 MFMAs.  The fragments of pair
    // kp + 1 are requested before the MFMAs of pair kp (the compiler does not pipeline this itself: it left the LDS latency
    // open every second pair), and the staging of step s + 2 (LDS writes from z, pairs 0-3) and the global loads of step
    // s + 4 (into z, pairs 4-7) ride between the MFMA groups instead of standing in front of them after the barrier:
    // 119.8 -> 126.4 (fragment pipelining) -> see DESIGN.md for the final figure.  sched_barrier pins the pair boundaries.
    const int frag_a = half * kZApad + 128 * mh + 4 * l31, frag_b = 16 * kZApad + half * kZB + 128 * jh + 4 * l31;
    // pa / pb: the fragments of the step's first pair, already requested by the previous step (step s + 1 became visible at
    // the barrier in front of step s, so its first fragments need not wait for the next barrier and its LDS latency)
    auto compute = [&](int s, Regs &z, f32x4 &pa, f32x4 &pb, auto fast_tag) {
        constexpr bool fast = decltype(fast_tag)::value;
        const float *sa = zstage + (size_t)(s % kWStages) * kStageFloats + frag_a;
        const float *sb = zstage + (size_t)(s % kWStages) * kStageFloats + frag_b;
        const float *sn = zstage + (size_t)((s + 1) % kWStages) * kStageFloats;
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {                              // v = 2 kp + half
            f32x4 na, nb;
            if (kp < 7) {
                na = *reinterpret_cast<const f32x4 *>(sa + 2 * (kp + 1) * kZApad);
                nb = *reinterpret_cast<const f32x4 *>(sb + 2 * (kp + 1) * kZB);
            } else {
                na = *reinterpret_cast<const f32x4 *>(sn + frag_a);
                nb = *reinterpret_cast<const f32x4 *>(sn + frag_b);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < 3) {
                    if (kp < 4) lwrite_part(s + 2, z, kp, u, fast);
                    else if (u < 2) gload_part(s + 4, z, kp - 4, u, fast);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[t], pb[u], acc[t][u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            pa = na;
            pb = nb;
        }
    };

    {
        const int steps2 = (steps + 1) & ~1;                          // single loop exit (a step past V stages zeros)
        Regs r0, r1;
        gload(0, r0);
        lwrite(0, r0);
        gload(1, r0);
        lwrite(1, r0);
        gload(2, r0);
        gload(3, r1);
        __syncthreads();                                             // steps 0 and 1 visible
        f32x4 pa = *reinterpret_cast<const f32x4 *>(zstage + frag_a);
        f32x4 pb = *reinterpret_cast<const f32x4 *>(zstage + frag_b);
        // main loop: every step staged (s + 2, s + 3) or loaded (s + 4, s + 5) during the two computes lies wholly inside V
        // -> the branch-free staging path; the last few steps (V tail, the ghost step of an odd count) run the general one.
        // Two loops in sequence, not a branch inside one: duplicated bodies inside one loop made the allocator shuffle the
        // 256 accumulators between the copies.
        int s = 0;
        for (; s < steps2 && 16 * (s + 5) + 16 <= V; s += 2) {
            __syncthreads();                                         // step s + 1 visible; the buffer of step s + 2 is free
            compute(s, r0, pa, pb, std::true_type{});
            __syncthreads();
            compute(s + 1, r1, pa, pb, std::true_type{});
        }
        for (; s < steps2; s += 2) {
            __syncthreads();
            compute(s, r0, pa, pb, std::false_type{});
            __syncthreads();
            compute(s + 1, r1, pa, pb, std::false_type{});
        }
    }

    // epilogue: tile (t, u) element (row i, col c) is cell m0 + 128 mh + 4 i + t, column j0 + 128 jh + 4 c + u: the four
    // u's of a (t, i) are four consecutive columns -> float4 loads of ep / pp and float4 stores
    const int jcol = j0 + 128 * jh + 4 * l31;
    if (jcol < J) {                                                   // J % 4 == 0
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * half;
                const int row = 128 * mh + 4 * i + t;
                const long m = m0 + row;
                if (m >= M) continue;
                const bool ok = row_ok[row] != 0;
                const f32x4 e4 = *reinterpret_cast<const f32x4 *>(ep + row_e[row] + jcol);
                const f32x4 p4 = *reinterpret_cast<const f32x4 *>(pp + row_p[row] + jcol);
                f32x4 h4, g4;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float hh = 0.f, dh = 1.f;
                    if (!ANY_ACT) {
                        hh = tanhf(e4[u] + p4[u]);
                        dh = 1.f - hh * hh;
                    }
                    h4[u] = ok ? hh : 0.f;
                    g4[u] = ok ? acc[t][u][q] * dh : 0.f;
                }
                *reinterpret_cast<f32x4 *>(dz + (size_t)m * J + jcol) = g4;
                if (!ANY_ACT && hout) *reinterpret_cast<f32x4 *>(hout + (size_t)m * J + jcol) = h4;
            }
    }
}

__global__ void split_dw_reduce_kernel(const float *__restrict__ part_dw, const float *__restrict__ part_db, int parts,
                                       long nw, int V, float *__restrict__ dw, float *__restrict__ db)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw + V; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        if (i < nw) {
            for (int p = 0; p < parts; ++p) s += part_dw[(size_t)p * nw + i];
            dw[i] = s;
        } else if (db != nullptr) {
            const long v = i - nw;
            for (int p = 0; p < parts; ++p) s += part_db[(size_t)p * V + v];
            db[v] = s;
        }
    }
}

inline int split_dw_parts(int V, int J)
{
    const int tiles = ((V + kWB - 1) / kWB) * ((J + kWB - 1) / kWB);
    const int parts = 256 / tiles;                                   // one workgroup per CU
    return parts < 1 ? 1 : parts;
}

int split_check(int B, int T, int U1, int J, int V, int terms, int out_dtype, int activation = WR_ACT_TANH)
{
    WR_REQUIRE(activation >= WR_ACT_TANH && activation <= WR_ACT_GELU, WR_EINVAL,
               "joint_split: activation code %d is not a wr_activation", activation);
    WR_REQUIRE(B > 0 && T > 0 && U1 > 0 && J > 0 && V > 0, WR_EINVAL,
               "joint_split: B, T, U1, J, V must be positive (got %d,%d,%d,%d,%d)", B, T, U1, J, V);
    WR_REQUIRE(J % 4 == 0 && J <= 512, WR_EUNSUPPORTED,
               "joint_split: join_dim=%d not supported (must be a multiple of 4, at most 512)", J);
    WR_REQUIRE(terms == 1 || terms == 3, WR_EINVAL, "joint_split: terms must be 1 (bf16) or 3 (split fp32), got %d", terms);
    WR_REQUIRE(out_dtype >= 0 && out_dtype <= 2, WR_EINVAL, "joint_split: bad output dtype code %d", out_dtype);
    WR_REQUIRE(((long)B * T * U1 + kSM - 1) / kSM < (1L << 31), WR_EUNSUPPORTED, "joint_split: too many lattice cells");
    return WR_OK;
}

}  // namespace

// dz[m, j] *= act'(ep[bt, j] + pp[bu, j]),  h[m, j] = act(...)   (zero in padded cells: dz already is): the second pass
// of the generic-activation path of joint_bwd_dz_block_kernel<true>
// HT: dtype of the activation copy (float, or __bf16 with row stride h_ld >= J for the AMP step, whose weight gradient is
// a library GEMM over bf16 operands: column J of a row is then 1 in valid cells -- the bias gradient falls out of the
// same GEMM as one more column -- and columns J+1 .. h_ld-1 are zero)
template <typename HT>
__global__ void joint_dz_act_kernel(const float *__restrict__ ep, const float *__restrict__ pp, const int32_t *__restrict__ llens,
                                    const int32_t *__restrict__ tlens, int T, int U1, int J, long M, int act,
                                    float *__restrict__ dz, HT *__restrict__ hout, int h_ld)
{
    const long n4 = M * (J / 4);                                      // J % 4 == 0
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long m = i / (J / 4);
        const int j = (int)(i - m * (J / 4)) * 4;
        const long bt = m / U1;
        const int u = (int)(m - bt * U1);
        const long b = bt / T;
        bool ok = true;
        if (llens != nullptr && tlens != nullptr) ok = ((int)(bt - b * T) < llens[b]) && (u <= tlens[b]);
        const f32x4 e4 = *reinterpret_cast<const f32x4 *>(ep + bt * J + j);
        const f32x4 p4 = *reinterpret_cast<const f32x4 *>(pp + (b * U1 + u) * J + j);
        f32x4 g4 = *reinterpret_cast<const f32x4 *>(dz + m * J + j), h4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float hh, dh;
            act_value_grad(act, e4[q] + p4[q], hh, dh);
            g4[q] = ok ? g4[q] * dh : 0.f;
            h4[q] = ok ? hh : 0.f;
        }
        *reinterpret_cast<f32x4 *>(dz + m * J + j) = g4;
        if (hout) {
            if constexpr (std::is_same<HT, float>::value) {
                *reinterpret_cast<f32x4 *>(hout + m * h_ld + j) = h4;
            } else {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                typedef HT h2 __attribute__((ext_vector_type(2)));
                const h2 p0 = __builtin_convertvector((f32x2){h4[0], h4[1]}, h2);
                const h2 p1 = __builtin_convertvector((f32x2){h4[2], h4[3]}, h2);
                *reinterpret_cast<u32x2 *>(hout + m * h_ld + j) = (u32x2){__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
            }
            if (j == 0)
                for (int c = J; c < h_ld; ++c) hout[m * h_ld + c] = (HT)((c == J && ok) ? 1.f : 0.f);
        }
    }
}

// Exact-fp32 activation gradient with the block tiling (called by wr_joint_bwd_dz in joint.hip).
int joint_bwd_dz_block(const float *gout_d, const float *ep_d, const float *pp_d, const float *w_d, const int32_t *llens_d,
                       const int32_t *tlens_d, int B, int T, int U1, int J, int V, int act, float *dz_d, float *h_d, hipStream_t st)
{
    const long M = (long)B * T * U1;
    const int n_js = (J + kZB - 1) / kZB;
    const long blocks = (M + kZB - 1) / kZB * n_js;
    WR_REQUIRE(blocks < (1L << 31), WR_EUNSUPPORTED, "joint_bwd_dz: too many lattice cells");
    const size_t lds = (size_t)kWStages * (16 * kZApad + 16 * kZB) * sizeof(float) + (size_t)kZB * (2 * sizeof(long) + sizeof(int));
    if (act == WR_ACT_TANH) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_bwd_dz_block_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(joint_bwd_dz_block_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, st, gout_d, ep_d, pp_d, w_d,
                           llens_d, tlens_d, B, T, U1, J, V, n_js, act, dz_d, h_d);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_bwd_dz_block_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(joint_bwd_dz_block_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, st, gout_d, ep_d, pp_d, w_d,
                           llens_d, tlens_d, B, T, U1, J, V, n_js, act, dz_d, h_d);
        hipLaunchKernelGGL(joint_dz_act_kernel<float>, dim3(256 * 16), dim3(256), 0, st, ep_d, pp_d, llens_d, tlens_d, T, U1, J, M, act,
                           dz_d, h_d, J);
    }
    WR_CHECK_LAUNCH("joint_bwd_dz_block_kernel");
    return WR_OK;
}

// Exact-fp32 weight gradient with the block tiling (called by wr_joint_bwd_dw in joint.hip).  `max_parts` is what the
// caller's workspace was sized for.
int joint_bwd_dw_block(const float *gout_d, const float *h_d, const int32_t *llens_d, const int32_t *tlens_d, int B, int T,
                       int U1, int J, int V, int max_parts, float *dw_d, float *db_d, float *part_dw, hipStream_t st)
{
    const long M = (long)B * T * U1;
    WR_REQUIRE(M < (1L << 31), WR_EUNSUPPORTED, "joint_bwd_dw: too many lattice cells");
    int parts = split_dw_parts(V, J);
    parts = parts > max_parts ? max_parts : parts;
    float *part_db = part_dw + (size_t)parts * V * J;
    const int n_vs = (V + kWB - 1) / kWB, n_js = (J + kWB - 1) / kWB;
    long rows_per_part = (M + parts - 1) / parts;
    rows_per_part = (rows_per_part + 15) / 16 * 16;
    const size_t lds = (size_t)kWStages * 2 * 16 * kWB * sizeof(float);
    if (llens_d != nullptr && tlens_d != nullptr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_bwd_dw_block_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(joint_bwd_dw_block_kernel<true>, dim3(n_vs * n_js * parts), dim3(256), lds, st, gout_d, h_d, llens_d,
                           tlens_d, T, U1, M, V, J, n_vs, n_js, rows_per_part, part_dw, part_db);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_bwd_dw_block_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(joint_bwd_dw_block_kernel<false>, dim3(n_vs * n_js * parts), dim3(256), lds, st, gout_d, h_d, llens_d,
                           tlens_d, T, U1, M, V, J, n_vs, n_js, rows_per_part, part_dw, part_db);
    }
    WR_CHECK_LAUNCH("joint_bwd_dw_block_kernel");
    hipLaunchKernelGGL(split_dw_reduce_kernel, dim3(1024), dim3(256), 0, st, part_dw, part_db, parts, (long)V * J, V, dw_d, db_d);
    WR_CHECK_LAUNCH("split_dw_reduce_kernel");
    return WR_OK;
}
}  // namespace wr

using namespace wr;

#ifdef WR_JS_STAMPS
extern "C" int wr_debug_read_js_stamps(void *host, size_t n_u64)
{
    const size_t have = sizeof(wr::g_js) / sizeof(unsigned long long);
    if (hipDeviceSynchronize() != hipSuccess) return WR_ELAUNCH;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(wr::g_js), (n_u64 < have ? n_u64 : have) * sizeof(unsigned long long)) == hipSuccess
               ? WR_OK : WR_ELAUNCH;
}
#endif

extern "C" size_t wr_joint_split_workspace_bytes(int J, int V)
{
    if (J <= 0 || V <= 0) return 0;
    return 2 * align_up((size_t)split_jpad(J) * split_vpad(V) * sizeof(unsigned short), 256);
}

namespace {
int joint_fwd_split_launch(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                           const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T, int U1, int J,
                           int V, int act, int terms, void *out_d, int out_dtype, void *workspace_d, size_t workspace_bytes,
                           const JointLse *lse, hipStream_t st)
{
    const int Vp = split_vpad(V), Jp = split_jpad(J);
    const size_t img = align_up((size_t)Jp * Vp * sizeof(unsigned short), 256);
    WR_REQUIRE(workspace_bytes >= 2 * img, WR_EWORKSPACE, "joint_fwd_split: workspace too small");
    unsigned short *wh = static_cast<unsigned short *>(workspace_d);
    unsigned short *wl = reinterpret_cast<unsigned short *>(static_cast<char *>(workspace_d) + img);
    hipLaunchKernelGGL(split_w_kernel, dim3(512), dim3(256), 0, st, w_out_d, V, J, Vp, Jp, wh, wl);
    WR_CHECK_LAUNCH("split_w_kernel");
    const long M = (long)B * T * U1;
    // Column slabs per XCD (part = blockIdx % npart) would keep W in each XCD's L2; measured (B=8: 20.8 / 21.5 / 23.0 ms
    // for 1 / 2 / 4 parts) the Infinity Cache already feeds the fragments fast enough and the extra activation tiles
    // cost more, so one part is the default; wr_tune_set(7, n) overrides for experiments.
    int npart = 1;
    if (const int forced = tune_get(kTuneSplitParts)) npart = forced;
    if (lse) npart = 1;                                     // the row statistics need every column in one workgroup
    // single-term (AMP) mode, experiment: 128 cells per workgroup -- their one bf16 image and the bias
    // slab fit LDS (J <= 512 at V = 5000) -- halve the W traffic per logit.  Measured at the B = 8 BASELINE slice: 13.84 ms
    // against 12.21 ms for the 64-cell form (eight accumulator tiles per wave leave the allocator 196 bytes of scratch per
    // lane, and with half the W traffic the kernel is no faster: the stream out of the L2s is not what bounds the
    // single-term mode).  Kept behind wr_tune_set(12, 2); the 64-cell form stays the default.
    const bool wide = terms == 1 && !lse && tune_get(kTuneSplitFwdCells) == 2 && M >= 128 * 512 &&
                      (size_t)128 * (Jp + 8) * sizeof(unsigned short) + (size_t)Vp * sizeof(float) <= 160 * 1024;
    const int cells = wide ? 128 : kSM;
    // single-term mode, default since round 3: TWO workgroups per CU (256 registers per wave, <= 80 KB of LDS each: no
    // bias slab, a store stage of 16 / 8 rows) -- wr_tune_set(12, 1) keeps the one-per-CU form
    const size_t esz0 = out_dtype == 0 ? 4 : 2;
    const size_t stage2 = (size_t)kSWaves * (esz0 == 4 ? 8 : 16) * (kSCT * 32 * esz0 + 16);
    const bool trn = terms == 1 && !lse && tune_get(kTuneSplitFwdStore) == 1;      // transposed tiles, stores from registers (measured slower: 16-byte pieces)
    const bool two = terms == 1 && !lse && !wide && tune_get(kTuneSplitFwdCells) != 1 &&
                     (size_t)kSM * (Jp + 8) * sizeof(unsigned short) + (trn ? 0 : stage2) <= 80 * 1024 &&
                     (trn || (((size_t)V * esz0) % 16 == 0 && (reinterpret_cast<size_t>(out_d) & 15) == 0));
    // two per CU: W in two column slabs (a workgroup covers one; consecutive workgroups land on consecutive XCDs, so an
    // XCD keeps re-reading ONE slab) once the bf16 image exceeds an XCD's 4 MB L2 -- 8.20 -> 7.81 ms at V = 5000, J = 512
    // although every activation tile is then built twice; four slabs: 8.8 ms
    // (larger vocabularies: as many slabs as keep one at <= 3 MB -- the measured case's 2.6 MB -- up to eight)
    if (two && tune_get(kTuneSplitParts) == 0 && img > (size_t)4 << 20) {
        npart = 2;
        while (npart < 8 && img / npart > (size_t)3 << 20) npart *= 2;
    }
    size_t tile_lds = (size_t)(terms == 3 ? 2 : 1) * cells * (Jp + 8) * sizeof(unsigned short);
    if (lse && tile_lds < joint_lse_exchange_bytes(kSWaves)) tile_lds = joint_lse_exchange_bytes(kSWaves);
    const size_t extra = 0;
    if (lse) (void)hipMemsetAsync(lse->repair, 0, sizeof(int32_t), st);
    while (!lse && npart < 64 &&
           tile_lds + (size_t)((Vp / (32 * kSCT) + npart - 1) / npart) * 32 * kSCT * sizeof(float) > 160 * 1024)
        npart *= 2;                                         // large vocabularies: the bias slab must fit beside the tile
    WR_REQUIRE((M + cells - 1) / cells * npart < (1L << 31), WR_EUNSUPPORTED, "joint_fwd_split: too many lattice cells");
    const int pairs_per_part = (Vp / (32 * kSCT) + npart - 1) / npart;
    const size_t bias_lds = (size_t)pairs_per_part * 32 * kSCT * sizeof(float);
    // single-term mode: a store stage of 32 rows x (64 columns + 16 bytes) per wave behind the (16-byte rounded) bias slab
    const size_t esz = out_dtype == 0 ? 4 : 2;
    const size_t bias_al = align_up(bias_lds, 16);
    const size_t stage_lds = (terms == 1 && !lse && !trn) ? (size_t)kSWaves * 32 * (kSCT * 32 * esz + 16) : 0;
    const bool stage_fits = stage_lds > 0 && tile_lds + bias_al + stage_lds <= 160 * 1024;
    const size_t lds = two ? tile_lds + (trn ? 0 : stage2) : tile_lds + (stage_fits ? bias_al + stage_lds : align_up(bias_lds, 16)) + extra;
    const int kb_bias = two ? 0 : (int)bias_al, kb_stage = (two || stage_fits) ? 1 : 0;
    WR_REQUIRE(lds <= 160 * 1024, WR_EUNSUPPORTED, "joint_fwd_split: V=%d needs %zu bytes of LDS", V, lds);
    const dim3 grid((unsigned)((M + cells - 1) / cells * npart));
#define WR_LAUNCH_SPLIT_WIDE(OutT, TRN)                                                                                  \
    do {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_fwd_split_kernel<1, OutT, false, 4, 1, TRN>),    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        hipLaunchKernelGGL((joint_fwd_split_kernel<1, OutT, false, 4, 1, TRN>), grid, dim3(64 * kSWaves), lds, st, ep_d, pp_d, \
                           reinterpret_cast<const u32x4 *>(wh), reinterpret_cast<const u32x4 *>(wl), b_out_d,          \
                           logit_lengths_d, target_lengths_d, B, T, U1, J, Jp, V, Vp, npart, act, static_cast<OutT *>(out_d), \
                           JointLse{}, kb_bias, kb_stage);                                                            \
    } while (0)
#define WR_LAUNCH_SPLIT_TWO(OutT, TRN)                                                                                   \
    do {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_fwd_split_kernel<1, OutT, false, 2, 2, TRN>),    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        hipLaunchKernelGGL((joint_fwd_split_kernel<1, OutT, false, 2, 2, TRN>), grid, dim3(64 * kSWaves), lds, st, ep_d, \
                           pp_d, reinterpret_cast<const u32x4 *>(wh), reinterpret_cast<const u32x4 *>(wl), b_out_d,    \
                           logit_lengths_d, target_lengths_d, B, T, U1, J, Jp, V, Vp, npart, act, static_cast<OutT *>(out_d), \
                           JointLse{}, kb_bias, kb_stage);                                                            \
    } while (0)
#define WR_LAUNCH_SPLIT(TERMS, OutT, TRN)                                                                             \
    do {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_fwd_split_kernel<TERMS, OutT, false, 2, 1, TRN>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        hipLaunchKernelGGL((joint_fwd_split_kernel<TERMS, OutT, false, 2, 1, TRN>), grid, dim3(64 * kSWaves), lds, st, ep_d, pp_d, \
                           reinterpret_cast<const u32x4 *>(wh), reinterpret_cast<const u32x4 *>(wl), b_out_d,          \
                           logit_lengths_d, target_lengths_d, B, T, U1, J, Jp, V, Vp, npart, act, static_cast<OutT *>(out_d), \
                           JointLse{}, kb_bias, kb_stage);                                                            \
    } while (0)
#define WR_LAUNCH_SPLIT_LSE(TERMS)                                                                                    \
    do {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_fwd_split_kernel<TERMS, float, true>),           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        hipLaunchKernelGGL((joint_fwd_split_kernel<TERMS, float, true>), grid, dim3(64 * kSWaves), lds, st, ep_d, pp_d, \
                           reinterpret_cast<const u32x4 *>(wh), reinterpret_cast<const u32x4 *>(wl), b_out_d,          \
                           logit_lengths_d, target_lengths_d, B, T, U1, J, Jp, V, Vp, npart, act, static_cast<float *>(out_d), \
                           *lse);                                                                                     \
    } while (0)
    if (lse) {
        if (terms == 3) WR_LAUNCH_SPLIT_LSE(3); else WR_LAUNCH_SPLIT_LSE(1);
    } else if (terms == 3) {
        if (out_dtype == 0) WR_LAUNCH_SPLIT(3, float, false);
        else if (out_dtype == 1) WR_LAUNCH_SPLIT(3, _Float16, false);
        else WR_LAUNCH_SPLIT(3, __bf16, false);
    } else if (two && trn) {
        if (out_dtype == 0) WR_LAUNCH_SPLIT_TWO(float, true);
        else if (out_dtype == 1) WR_LAUNCH_SPLIT_TWO(_Float16, true);
        else WR_LAUNCH_SPLIT_TWO(__bf16, true);
    } else if (two) {
        if (out_dtype == 0) WR_LAUNCH_SPLIT_TWO(float, false);
        else if (out_dtype == 1) WR_LAUNCH_SPLIT_TWO(_Float16, false);
        else WR_LAUNCH_SPLIT_TWO(__bf16, false);
    } else if (wide && trn) {
        if (out_dtype == 0) WR_LAUNCH_SPLIT_WIDE(float, true);
        else if (out_dtype == 1) WR_LAUNCH_SPLIT_WIDE(_Float16, true);
        else WR_LAUNCH_SPLIT_WIDE(__bf16, true);
    } else if (wide) {
        if (out_dtype == 0) WR_LAUNCH_SPLIT_WIDE(float, false);
        else if (out_dtype == 1) WR_LAUNCH_SPLIT_WIDE(_Float16, false);
        else WR_LAUNCH_SPLIT_WIDE(__bf16, false);
    } else if (trn) {
        if (out_dtype == 0) WR_LAUNCH_SPLIT(1, float, true);
        else if (out_dtype == 1) WR_LAUNCH_SPLIT(1, _Float16, true);
        else WR_LAUNCH_SPLIT(1, __bf16, true);
    } else {
        if (out_dtype == 0) WR_LAUNCH_SPLIT(1, float, false);
        else if (out_dtype == 1) WR_LAUNCH_SPLIT(1, _Float16, false);
        else WR_LAUNCH_SPLIT(1, __bf16, false);
    }
#undef WR_LAUNCH_SPLIT_WIDE
#undef WR_LAUNCH_SPLIT_TWO
#undef WR_LAUNCH_SPLIT
#undef WR_LAUNCH_SPLIT_LSE
    WR_CHECK_LAUNCH("joint_fwd_split_kernel");
    return WR_OK;
}
}  // namespace

extern "C" int wr_joint_fwd_split(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                                  const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T, int U1,
                                  int J, int V, int activation, int terms, void *out_d, int out_dtype, void *workspace_d,
                                  size_t workspace_bytes, void *stream)
{
    if (int rc = split_check(B, T, U1, J, V, terms, out_dtype, activation)) return rc;
    WR_REQUIRE(ep_d && pp_d && w_out_d && b_out_d && out_d && workspace_d, WR_EINVAL,
               "joint_fwd_split: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_fwd_split: pass both length arrays or neither");
    return joint_fwd_split_launch(ep_d, pp_d, w_out_d, b_out_d, logit_lengths_d, target_lengths_d, B, T, U1, J, V, activation,
                                  terms, out_d, out_dtype, workspace_d, workspace_bytes, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int wr_joint_fwd_split_lse(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                                      const int32_t *logit_lengths_d, const int32_t *target_lengths_d,
                                      const int32_t *targets_d, int B, int T, int U1, int J, int V, int activation, int blank,
                                      int terms, float *out_d, void *workspace_d, size_t workspace_bytes, void *rnnt_workspace_d,
                                      size_t rnnt_workspace_bytes, void *stream)
{
    if (int rc = split_check(B, T, U1, J, V, terms, WR_F32, activation)) return rc;
    WR_REQUIRE(ep_d && pp_d && w_out_d && b_out_d && out_d && workspace_d && rnnt_workspace_d, WR_EINVAL,
               "joint_fwd_split_lse: null pointer argument");
    WR_REQUIRE(logit_lengths_d && target_lengths_d, WR_EINVAL, "joint_fwd_split_lse: both length arrays are required");
    WR_REQUIRE(targets_d || U1 == 1, WR_EINVAL, "joint_fwd_split_lse: targets is null");
    WR_REQUIRE(blank >= 0 && blank < V, WR_EINVAL, "joint_fwd_split_lse: blank %d out of range [0,%d)", blank, V);
    WR_REQUIRE(U1 <= kRnntMaxCols, WR_EUNSUPPORTED, "joint_fwd_split_lse: U1=%d exceeds the loss's limit of %d", U1, kRnntMaxCols);
    const RnntWs w = rnnt_ws_layout(B, T, U1);
    WR_REQUIRE(rnnt_workspace_bytes >= w.total, WR_EWORKSPACE, "joint_fwd_split_lse: RNN-T workspace %zu < required %zu",
               rnnt_workspace_bytes, w.total);
    char *rws = static_cast<char *>(rnnt_workspace_d);
    JointLse lse{targets_d, blank, w.S, reinterpret_cast<float2 *>(rws + w.lp_off), reinterpret_cast<float *>(rws + w.denom_off),
                 reinterpret_cast<int32_t *>(rws + w.flag_off)};
    return joint_fwd_split_launch(ep_d, pp_d, w_out_d, b_out_d, logit_lengths_d, target_lengths_d, B, T, U1, J, V, activation,
                                  terms, out_d, WR_F32, workspace_d, workspace_bytes, &lse, static_cast<hipStream_t>(stream));
}

extern "C" size_t wr_joint_dz_split_workspace_bytes(int J, int V)
{
    if (J <= 0 || V <= 0) return 0;
    const size_t n_jt = (size_t)(J + 31) / 32, D = (size_t)(V + 31) / 32;
    return 2 * align_up(n_jt * D * 2 * 64 * 8 * sizeof(unsigned short), 256);
}

namespace {
int joint_bwd_dz_split_launch(const void *gout_d, bool g16, const float *ep_d, const float *pp_d, const float *w_out_d,
                              const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T, int U1,
                              int J, int V, int activation, int terms, float *dz_d, float *h_d, void *workspace_d,
                              size_t workspace_bytes, void *stream)
{
    if (int rc = split_check(B, T, U1, J, V, terms, 0, activation)) return rc;
    WR_REQUIRE(V % (g16 ? 8 : 4) == 0 && V >= 32, WR_EUNSUPPORTED,
               "joint_bwd_dz_split: V=%d not supported (16-byte aligned gradient rows: V a multiple of %d, at least 32)", V,
               g16 ? 8 : 4);
    WR_REQUIRE(gout_d && ep_d && pp_d && w_out_d && dz_d && workspace_d, WR_EINVAL, "joint_bwd_dz_split: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_bwd_dz_split: pass both length arrays or neither");
    const int n_jt = (J + 31) / 32, D = (V + 31) / 32;
    const size_t img = align_up((size_t)n_jt * D * 2 * 64 * 8 * sizeof(unsigned short), 256);
    WR_REQUIRE(workspace_bytes >= 2 * img, WR_EWORKSPACE, "joint_bwd_dz_split: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned short *wh = static_cast<unsigned short *>(workspace_d);
    unsigned short *wl = reinterpret_cast<unsigned short *>(static_cast<char *>(workspace_d) + img);
    hipLaunchKernelGGL(split_w_dz_kernel, dim3(512), dim3(256), 0, st, w_out_d, V, J, D, n_jt, wh, wl);
    WR_CHECK_LAUNCH("split_w_dz_kernel");
    const long M = (long)B * T * U1;
    const dim3 grid((unsigned)((M + kSM - 1) / kSM));
    {                                                       // 128-cell tiling, W fragments staged in LDS
        const size_t lds = (size_t)kZStages * 2 * 16 * 64 * 16 + (size_t)kZM2 * (2 * sizeof(long) + sizeof(int));
        const dim3 grid2((unsigned)((M + kZM2 - 1) / kZM2));
#define WR_LAUNCH_DZ2(TERMS, FULL_, GT)                                                                                \
        do {                                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_bwd_dz_split128_kernel<TERMS, FULL_, GT>),   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
            hipLaunchKernelGGL((joint_bwd_dz_split128_kernel<TERMS, FULL_, GT>), grid2, dim3(256), lds, st,             \
                               static_cast<const GT *>(gout_d), ep_d, pp_d,                                            \
                               reinterpret_cast<const u32x4 *>(wh), reinterpret_cast<const u32x4 *>(wl),               \
                               logit_lengths_d, target_lengths_d, B, T, U1, J, V, D, n_jt, activation, dz_d, h_d);     \
        } while (0)
#define WR_LAUNCH_DZ2_T(TERMS, FULL_) do { if (g16) WR_LAUNCH_DZ2(TERMS, FULL_, __bf16); else WR_LAUNCH_DZ2(TERMS, FULL_, float); } while (0)
        if (n_jt == 16) { if (terms == 3) WR_LAUNCH_DZ2_T(3, true); else WR_LAUNCH_DZ2_T(1, true); }
        else { if (terms == 3) WR_LAUNCH_DZ2_T(3, false); else WR_LAUNCH_DZ2_T(1, false); }
#undef WR_LAUNCH_DZ2_T
#undef WR_LAUNCH_DZ2
        WR_CHECK_LAUNCH("joint_bwd_dz_split128_kernel");
    }
    return WR_OK;
}
}  // namespace

extern "C" int wr_joint_bwd_dz_split(const float *gout_d, const float *ep_d, const float *pp_d, const float *w_out_d,
                                     const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T, int U1,
                                     int J, int V, int activation, int terms, float *dz_d, float *h_d, void *workspace_d,
                                     size_t workspace_bytes, void *stream)
{
    return joint_bwd_dz_split_launch(gout_d, false, ep_d, pp_d, w_out_d, logit_lengths_d, target_lengths_d, B, T, U1, J, V,
                                     activation, terms, dz_d, h_d, workspace_d, workspace_bytes, stream);
}

extern "C" int wr_joint_bwd_dz_split_bf16(const void *gout_bf16_d, const float *ep_d, const float *pp_d, const float *w_out_d,
                                          const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T,
                                          int U1, int J, int V, int activation, int terms, float *dz_d, float *h_d,
                                          void *workspace_d, size_t workspace_bytes, void *stream)
{
    return joint_bwd_dz_split_launch(gout_bf16_d, true, ep_d, pp_d, w_out_d, logit_lengths_d, target_lengths_d, B, T, U1, J,
                                     V, activation, terms, dz_d, h_d, workspace_d, workspace_bytes, stream);
}

// ---- bias gradient from a bf16 logits gradient: db[v] = sum over valid cells of gout[cell, v] ----
// (the AMP step's weight gradient is a library GEMM; the column sums are this memory-bound pass: a thread owns 8 columns,
// a workgroup of 640 threads whole rows of up to 5 120 columns -- sequential 10 KB reads at V = 5000 -- of one of `parts`
// row ranges; partial sums, then an ordered reduction: deterministic)
namespace wr {
namespace {
constexpr int kDbParts = 1024;
constexpr int kDbThreads = 640;
template <typename GT>
__global__ __launch_bounds__(kDbThreads) void joint_db_bf16_kernel(const GT *__restrict__ gout, const unsigned char *__restrict__ mask,
                                                                   long M, int V, long rows_per_part, float *__restrict__ part)
{
    const int v = (blockIdx.x * kDbThreads + threadIdx.x) * 8;
    const long r0 = (long)blockIdx.y * rows_per_part;
    const long r1 = r0 + rows_per_part < M ? r0 + rows_per_part : M;
    if (v >= V) return;                                     // V % 8 == 0: a thread's columns are wholly in or out
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const GT *__restrict__ p = gout + v;
    constexpr int R = 8;                                    // rows in flight per thread (rows past the range repeat the last
    for (long m = r0; m < r1; m += R) {                     // one with weight 0)
        u32x4 x[R];
        bool on[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const long mm = m + q < r1 ? m + q : r1 - 1;
            x[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + (size_t)mm * V));
            on[q] = m + q < r1 && (mask == nullptr || mask[mm] != 0);
        }
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo, hi;
                if constexpr (std::is_same<GT, __bf16>::value) {
                    lo = __builtin_bit_cast(float, x[q][i] << 16);
                    hi = __builtin_bit_cast(float, x[q][i] & 0xffff0000u);
                } else {
                    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                    const f16x8 hv = __builtin_bit_cast(f16x8, x[q]);
                    lo = (float)hv[2 * i];
                    hi = (float)hv[2 * i + 1];
                }
                acc[2 * i] += on[q] ? lo : 0.f;             // a select: padded cells may hold anything
                acc[2 * i + 1] += on[q] ? hi : 0.f;
            }
    }
    float *__restrict__ o = part + (size_t)blockIdx.y * V + v;
    *reinterpret_cast<f32x4 *>(o) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4 *>(o + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
}
__global__ void joint_db_reduce_kernel(const float *__restrict__ part, int parts, int V, float *__restrict__ db)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float s = 0.f;
    for (int p = 0; p < parts; ++p) s += part[(size_t)p * V + v];
    db[v] = s;
}
}  // namespace
}  // namespace wr

extern "C" size_t wr_joint_db_workspace_bytes(int B, int T, int U1, int V)
{
    if (B <= 0 || T <= 0 || U1 <= 0 || V <= 0) return 0;
    return wr::align_up((size_t)wr::kDbParts * V * sizeof(float), 256) + wr::align_up((size_t)B * T * U1, 256);
}

namespace {
int joint_db_16_launch(const void *gout_bf16_d, bool f16, const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B,
                       int T, int U1, int V, float *db_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    using namespace wr;
    WR_REQUIRE(B > 0 && T > 0 && U1 > 0 && V > 0 && V % 8 == 0, WR_EINVAL, "joint_db_bf16: sizes must be positive, V a multiple of 8");
    WR_REQUIRE(gout_bf16_d && db_d && workspace_d, WR_EINVAL, "joint_db_bf16: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_db_bf16: pass both length arrays or neither");
    WR_REQUIRE(workspace_bytes >= wr_joint_db_workspace_bytes(B, T, U1, V), WR_EWORKSPACE, "joint_db_bf16: workspace too small");
    const long M = (long)B * T * U1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *part = static_cast<float *>(workspace_d);
    unsigned char *mask = nullptr;
    if (logit_lengths_d != nullptr) {
        mask = reinterpret_cast<unsigned char *>(static_cast<char *>(workspace_d) + align_up((size_t)kDbParts * V * sizeof(float), 256));
        hipLaunchKernelGGL(cell_mask_kernel, dim3(1024), dim3(256), 0, st, logit_lengths_d, target_lengths_d, T, U1, M, mask);
        WR_CHECK_LAUNCH("cell_mask_kernel");
    }
    const int parts = M < kDbParts ? (int)M : kDbParts;
    const long rows_per_part = (M + parts - 1) / parts;
    if (f16)
        hipLaunchKernelGGL(joint_db_bf16_kernel<_Float16>, dim3((V / 8 + kDbThreads - 1) / kDbThreads, parts), dim3(kDbThreads), 0, st,
                           static_cast<const _Float16 *>(gout_bf16_d), mask, M, V, rows_per_part, part);
    else
        hipLaunchKernelGGL(joint_db_bf16_kernel<__bf16>, dim3((V / 8 + kDbThreads - 1) / kDbThreads, parts), dim3(kDbThreads), 0, st,
                           static_cast<const __bf16 *>(gout_bf16_d), mask, M, V, rows_per_part, part);
    WR_CHECK_LAUNCH("joint_db_bf16_kernel");
    hipLaunchKernelGGL(joint_db_reduce_kernel, dim3((V + 255) / 256), dim3(256), 0, st, part, parts, V, db_d);
    WR_CHECK_LAUNCH("joint_db_reduce_kernel");
    return WR_OK;
}
}  // namespace

extern "C" int wr_joint_db_bf16(const void *gout_bf16_d, const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B,
                                int T, int U1, int V, float *db_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    return joint_db_16_launch(gout_bf16_d, false, logit_lengths_d, target_lengths_d, B, T, U1, V, db_d, workspace_d, workspace_bytes,
                              stream);
}

extern "C" int wr_joint_db_f16(const void *gout_f16_d, const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B,
                               int T, int U1, int V, float *db_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    return joint_db_16_launch(gout_f16_d, true, logit_lengths_d, target_lengths_d, B, T, U1, V, db_d, workspace_d, workspace_bytes,
                              stream);
}

extern "C" int wr_joint_dz_act(float *dz_d, const float *ep_d, const float *pp_d, const int32_t *logit_lengths_d,
                               const int32_t *target_lengths_d, int B, int T, int U1, int J, int activation, void *h_d,
                               int h_dtype, int h_ld, void *stream)
{
    WR_REQUIRE(B > 0 && T > 0 && U1 > 0 && J > 0 && J % 4 == 0, WR_EINVAL, "joint_dz_act: sizes must be positive, J a multiple of 4");
    WR_REQUIRE(activation >= WR_ACT_TANH && activation <= WR_ACT_GELU, WR_EINVAL, "joint_dz_act: unknown activation %d", activation);
    WR_REQUIRE(dz_d && ep_d && pp_d, WR_EINVAL, "joint_dz_act: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_dz_act: pass both length arrays or neither");
    WR_REQUIRE(h_d == nullptr || ((h_dtype == WR_F32 || h_dtype == WR_BF16 || h_dtype == WR_F16) && h_ld >= J && h_ld % 4 == 0),
               WR_EINVAL, "joint_dz_act: h must be fp32, fp16 or bf16 with a row stride >= J that is a multiple of 4");
    const long M = (long)B * T * U1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (h_d != nullptr && h_dtype == WR_BF16)
        hipLaunchKernelGGL(joint_dz_act_kernel<__bf16>, dim3(256 * 16), dim3(256), 0, st, ep_d, pp_d, logit_lengths_d,
                           target_lengths_d, T, U1, J, M, activation, dz_d, static_cast<__bf16 *>(h_d), h_ld);
    else if (h_d != nullptr && h_dtype == WR_F16)
        hipLaunchKernelGGL(joint_dz_act_kernel<_Float16>, dim3(256 * 16), dim3(256), 0, st, ep_d, pp_d, logit_lengths_d,
                           target_lengths_d, T, U1, J, M, activation, dz_d, static_cast<_Float16 *>(h_d), h_ld);
    else
        hipLaunchKernelGGL(joint_dz_act_kernel<float>, dim3(256 * 16), dim3(256), 0, st, ep_d, pp_d, logit_lengths_d,
                           target_lengths_d, T, U1, J, M, activation, dz_d, static_cast<float *>(h_d), h_d ? h_ld : J);
    WR_CHECK_LAUNCH("joint_dz_act_kernel");
    return WR_OK;
}

extern "C" size_t wr_joint_dw_split_workspace_bytes(int B, int T, int U1, int J, int V)
{
    if (B <= 0 || T <= 0 || U1 <= 0 || J <= 0 || V <= 0) return 0;
    const size_t M = (size_t)B * T * U1;
    return align_up((size_t)split_dw_parts(V, J) * ((size_t)V * J + V) * sizeof(float), 256) + align_up(M, 256);
}

namespace {
int joint_bwd_dw_split_launch(const void *gout_d, bool g16, const float *h_d, const int32_t *logit_lengths_d,
                              const int32_t *target_lengths_d, int B, int T, int U1, int J, int V, int terms,
                              float *dw_d, float *db_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    if (int rc = split_check(B, T, U1, J, V, terms, 0)) return rc;
    WR_REQUIRE(V % 4 == 0 && J % 4 == 0, WR_EUNSUPPORTED,
               "joint_bwd_dw_split: V=%d, J=%d not supported (16-byte aligned rows: multiples of 4)", V, J);
    WR_REQUIRE(gout_d && h_d && dw_d && workspace_d, WR_EINVAL, "joint_bwd_dw_split: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_bwd_dw_split: pass both length arrays or neither");
    const long M = (long)B * T * U1;
    const int parts = split_dw_parts(V, J);
    const size_t pbytes = align_up((size_t)parts * ((size_t)V * J + V) * sizeof(float), 256);
    WR_REQUIRE(workspace_bytes >= pbytes + align_up((size_t)M, 256), WR_EWORKSPACE,
               "joint_bwd_dw_split: workspace %zu < required %zu", workspace_bytes, pbytes + align_up((size_t)M, 256));
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *part_dw = static_cast<float *>(workspace_d);
    float *part_db = part_dw + (size_t)parts * V * J;
    unsigned char *mask = nullptr;
    if (logit_lengths_d != nullptr) {
        mask = reinterpret_cast<unsigned char *>(static_cast<char *>(workspace_d) + pbytes);
        hipLaunchKernelGGL(cell_mask_kernel, dim3(1024), dim3(256), 0, st, logit_lengths_d, target_lengths_d, T, U1, M, mask);
        WR_CHECK_LAUNCH("cell_mask_kernel");
    }
    const int n_vs = (V + kWB - 1) / kWB, n_js = (J + kWB - 1) / kWB;
    long rows_per_part = (M + parts - 1) / parts;
    rows_per_part = (rows_per_part + 15) / 16 * 16;
    const size_t lds = (size_t)kWStages * 2 * 2 * 2 * 4 * 64 * 16;   // three stages of ready-made fragments
#define WR_LAUNCH_DW(TERMS, GT)                                                                                        \
    do {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_bwd_dw_split_kernel<TERMS, GT>),                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        hipLaunchKernelGGL((joint_bwd_dw_split_kernel<TERMS, GT>), dim3(n_vs * n_js * parts), dim3(512), lds, st,       \
                           static_cast<const GT *>(gout_d), h_d, mask, M, V, J, n_vs, n_js, rows_per_part, part_dw, part_db); \
    } while (0)
    if (g16) { if (terms == 3) WR_LAUNCH_DW(3, __bf16); else WR_LAUNCH_DW(1, __bf16); }
    else { if (terms == 3) WR_LAUNCH_DW(3, float); else WR_LAUNCH_DW(1, float); }
#undef WR_LAUNCH_DW
    WR_CHECK_LAUNCH("joint_bwd_dw_split_kernel");
    hipLaunchKernelGGL(split_dw_reduce_kernel, dim3(1024), dim3(256), 0, st, part_dw, part_db, parts, (long)V * J, V, dw_d,
                       db_d);
    WR_CHECK_LAUNCH("split_dw_reduce_kernel");
    return WR_OK;
}
}  // namespace

extern "C" int wr_joint_bwd_dw_split(const float *gout_d, const float *h_d, const int32_t *logit_lengths_d,
                                     const int32_t *target_lengths_d, int B, int T, int U1, int J, int V, int terms,
                                     float *dw_d, float *db_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    return joint_bwd_dw_split_launch(gout_d, false, h_d, logit_lengths_d, target_lengths_d, B, T, U1, J, V, terms, dw_d, db_d,
                                     workspace_d, workspace_bytes, stream);
}

extern "C" int wr_joint_bwd_dw_split_bf16(const void *gout_bf16_d, const float *h_d, const int32_t *logit_lengths_d,
                                          const int32_t *target_lengths_d, int B, int T, int U1, int J, int V, int terms,
                                          float *dw_d, float *db_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    return joint_bwd_dw_split_launch(gout_bf16_d, true, h_d, logit_lengths_d, target_lengths_d, B, T, U1, J, V, terms, dw_d,
                                     db_d, workspace_d, workspace_bytes, stream);
}
