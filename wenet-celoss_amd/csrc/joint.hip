// Transducer joint network for MI355X (gfx950): the one dense contraction on
// the path.  Replaces TransducerJoint.forward (wenet/transducer/joint.py:45-70)
//     out = ffn_out(tanh(enc_ffn(enc)[:, :, None, :] + pred_ffn(pred)[:, None, :, :]))
// from the point where the two small pre-join projections are available:
//     ep = enc_ffn(enc)  [B, T,  J]      pp = pred_ffn(pred)  [B, U1, J]
// (those are plain library GEMMs on tiny tensors and stay with rocBLAS).
//
// Forward  joint_fwd_direct_kernel   out[m, v] = sum_k tanh(ep[bt(m),k] + pp[bu(m),k]) * W[v,k] + bias[v]
//   M = B*T*U1 lattice cells (4.83 M at the BASELINE shape), K = J = 512, N = V.
//   One workgroup (8 waves) owns 64 consecutive cells and ALL V columns: the activation tile
//   H = tanh(ep + pp) (64 x J) is computed ONCE into LDS (k-major, +1 padded) and
//   never exists in HBM (the reference materialises it: 9.9 GB + its autograd copy);
//   each wave streams the W^T fragments of its own 32 columns straight from L2 into a register
//   ping-pong (no LDS, no barrier in the k-loop).  Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32).
//   MFMA-bound: 2*M*J*V flop = 24.74 TFLOP forward at the BASELINE shape vs 157.3 TFLOP/s.
//   With LSE = true the epilogue also writes the RNN-T loss's row statistics (joint_lse.hpp).
//
// Backward (general shapes; the shipped shapes take the 256 x 256 block tilings of joint_split.hip:
//   joint_bwd_dz_block_kernel, joint_bwd_dw_block_kernel)
//   joint_bwd_dz_kernel   dZ[m,k] = (sum_v dY[m,v] * W[v,k]) * (1 - H[m,k]^2): one workgroup owns 64 cells
//     and all J columns, streams dY (the RNN-T gradient) once in 16-deep v-slices; H is recomputed,
//     optionally written out for the weight gradient.  Any V (odd vocabularies, unaligned rows).
//   joint_bwd_dw_kernel   dW = dY^T H, db = sum dY: 128-row slabs of the vocabulary x all J columns, operands
//     straight into registers, partial slabs summed by a deterministic second kernel.  Any V.
//   d ep = sum_u dZ and d pp = sum_t dZ are library reductions on the host side.
#include "wr_common.hpp"
#include "joint_lse.hpp"

namespace wr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 64;        // lattice cells per workgroup
constexpr int kBN = 256;       // output columns per chunk (4 waves x 64)
constexpr int kBK = 8;         // k-slice depth of the streamed operand (forward)
constexpr int kHPad = kBM + 1; // k-major activation tile row stride (floats)

inline int joint_vpad(int V) { return (V + 2 * kBN - 1) / (2 * kBN) * (2 * kBN); }   // whole 512-column chunks (8 waves x 2 tiles)
inline int joint_jpad(int J) { return (J + 31) / 32 * 32; }   // forward k-depth, zero padded (multiple of 2*PFK)

// W [V, J] row-major (nn.Linear weight)  ->  Wt [J, Vp] (k-major, zero padded to a multiple of 256 columns)
__global__ void joint_transpose_w_kernel(const float *__restrict__ w, int V, int J, int Jp, int Vp, float *__restrict__ wt)
{
    __shared__ float tile[32][33];
    const int v0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int v = v0 + i, k = k0 + tx;
        tile[i][tx] = (v < V && k < J) ? w[(size_t)v * J + k] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, v = v0 + tx;
        if (k < Jp && v < Vp) wt[(size_t)k * Vp + v] = tile[tx][i];
    }
}

// Fill the k-major activation tile: Ht[k][row] = act(ep[bt,k] + pp[bu,k]) for the 64 cells m0..m0+63 (act: tanh unless
// the module was built with another activation, wr_common.hpp act_value).
__device__ __forceinline__ void fill_h_tile(float *__restrict__ Ht, const float *__restrict__ ep,
                                            const float *__restrict__ pp, long m0, long M, int T, int U1, int J,
                                            int Jp, int act)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int row = wave; row < kBM; row += nw) {
        const long m = m0 + row;
        if (m < M) {
            const long bt = m / U1;
            const int u = (int)(m - bt * U1);
            const long b = bt / T;
            const float *__restrict__ e = ep + (size_t)bt * J;
            const float *__restrict__ p = pp + ((size_t)b * U1 + u) * J;
            for (int k = lane; k < Jp; k += 64) Ht[k * kHPad + row] = (k < J) ? act_value(act, e[k] + p[k]) : 0.f;
        } else {
            for (int k = lane; k < Jp; k += 64) Ht[k * kHPad + row] = 0.f;
        }
    }
}

// ---------------------------------------------------------------- forward --
// The streamed operand never touches LDS.  Each wave owns its own 32 output columns per 256-column chunk, so
// its B fragments (k-major W^T rows, 128 contiguous bytes per 32 columns) are not shared with the other
// waves: they are loaded straight from L2 into registers, PFK k-steps ahead, and the k-loop has no barrier
// and no LDS write at all -- only the read-only activation tile lives in LDS.
// LSE = true: the epilogue also produces the RNN-T loss's row statistics (joint_lse.hpp) -- the workgroup owns
// every column of its 64 cells, so pass 1 of the loss never has to read the logits back.
template <int PFK, int CT /* 32-column tiles per wave */, bool LSE, int NW = (CT == 2 ? 4 : 8) /* waves */>
__global__ __launch_bounds__(64 * NW) void joint_fwd_direct_kernel(
    const float *__restrict__ ep, const float *__restrict__ pp, const float *__restrict__ wt /* [Jp, Vp] */,
    const float *__restrict__ bias, const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens,
    int B, int T, int U1, int J, int Jp, int V, int Vp, int act, float *__restrict__ out, JointLse lse)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Ht = lds;                                  // [Jp][65]
    const long M = (long)B * T * U1;
    const long m0 = (long)blockIdx.x * kBM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float rm[32], rs[32];                             // LSE: this lane's (reference, partial sum) of its 32 rows
    if (LSE) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { rm[i] = -3.0e38f; rs[i] = 0.f; }
    }

    if (llens != nullptr && tlens != nullptr) {
        int valid = 0;
        const long m = m0 + tid;
        if (tid < kBM && m < M) {
            const long bt = m / U1;
            const int u = (int)(m - bt * U1);
            const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
            valid = (t < llens[b]) && (u <= tlens[b]);
        }
        if (!__syncthreads_or(valid)) return;
    }
    fill_h_tile(Ht, ep, pp, m0, M, T, U1, J, Jp, act);
    __syncthreads();

    constexpr int kChunk = NW * 32 * CT;                // output columns per chunk
    const int nchunks = Vp / kChunk;
    const int half = lane >> 5, l31 = lane & 31;
    const int wcol = wave * (32 * CT);
    const float *__restrict__ Ah = Ht + half * kHPad + l31;       // + k * kHPad (+32 for the second row tile)

    for (int nc = 0; nc < nchunks; ++nc) {
        const int v0 = nc * kChunk;
        f32x16 acc[2][CT];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[r][c] = (f32x16){0};
        const float *__restrict__ Bp = wt + (size_t)half * Vp + v0 + wcol + l31;   // + k * Vp (+32 second column tile)
        // Register ping-pong in blocks of PFK k-steps: while the MFMAs of one block run, the B fragments (L2)
        // and A fragments (LDS) of the next block are in flight in the other register set.  Straight-line,
        // unconditional loads (the prefetch past the end wraps to block 0 of the same chunk); sched_barrier
        // keeps the load group ahead of the MFMA group.
        float pb[CT][PFK], pa[2][PFK];      // set P
        float qb[CT][PFK], qa[2][PFK];      // set Q
#define WR_LOAD_SET(b_, a_, kb_)                                                         \
        _Pragma("unroll") for (int c = 0; c < CT; ++c)                                   \
            _Pragma("unroll") for (int i = 0; i < PFK; ++i)                              \
                b_[c][i] = Bp[(size_t)((kb_) + 2 * i) * Vp + 32 * c];                    \
        _Pragma("unroll") for (int r = 0; r < 2; ++r)                                    \
            _Pragma("unroll") for (int i = 0; i < PFK; ++i)                              \
                a_[r][i] = Ah[((kb_) + 2 * i) * kHPad + 32 * r];
#define WR_MFMA_SET(b_, a_)                                                                              \
        _Pragma("unroll") for (int i = 0; i < PFK; ++i)                                                  \
            _Pragma("unroll") for (int r = 0; r < 2; ++r)                                                \
                _Pragma("unroll") for (int c = 0; c < CT; ++c)                                           \
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[r][i], b_[c][i], acc[r][c], 0, 0, 0);
        WR_LOAD_SET(pb, pa, 0)
        for (int kk = 0; kk < Jp; kk += 4 * PFK) {          // Jp is a multiple of 32 = 4 * PFK at PFK = 8
            const int k1 = kk + 2 * PFK;
            WR_LOAD_SET(qb, qa, k1)
            __builtin_amdgcn_sched_barrier(0);
            WR_MFMA_SET(pb, pa)
            __builtin_amdgcn_sched_barrier(0);
            const int k2 = (kk + 4 * PFK < Jp) ? kk + 4 * PFK : 0;
            WR_LOAD_SET(pb, pa, k2)
            __builtin_amdgcn_sched_barrier(0);
            WR_MFMA_SET(qb, qa)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WR_LOAD_SET
#undef WR_MFMA_SET
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = v0 + wcol + 32 * c + l31;
            const float bv = (col < V) ? bias[col] : 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half;   // C/D layout of the 32x32 MFMA
                    const long m = m0 + row;
                    const float x = acc[rt][c][r] + bv;
                    if (m < M && col < V) out[(size_t)m * V + col] = x;
                    if (LSE) joint_lse_add(rm[rt * 16 + r], rs[rt * 16 + r], x, col < V, nc == 0 && c == 0);
                }
        }
    }
    if (LSE) {
        __syncthreads();                               // every wave is done with the activation tile: reuse its storage
        joint_lse_finish<NW>(lse, lds, rm, rs, llens, tlens, out, m0, M, T, U1, V);
    }
}

// ---- forward, fragment layout ------------------------------------------------------------------------------------
// Same ownership (64 cells x all V columns per workgroup, 8 waves, one 32-column tile per wave and chunk), but both MFMA
// operands arrive 16 bytes per lane and instruction: VALU / VMEM instructions of a wave do not hide behind its own MFMAs
// (tools/micro/mfma_valu_mix.hip), and the kernel above issues 1.5 four-byte loads per MFMA.  Here
//   W is re-laid once per call as  wf[(ct * KG + kg) * 64 + lane] = float4{ W[32 ct + l31][8 kg + 2 i + half], i = 0..3 }
//     (ct: 32-column tile, kg: group of 8 k, lane = 32 half + l31): one global_load_dwordx4 feeds 4 MFMAs per row tile;
//   H lives in LDS as  Ht[row][half][k >> 1]  (row stride Jp + 4 floats: conflict-free ds_read_b128): one read = the A
//     operands of 4 MFMAs.
// 0.375 load instructions per MFMA instead of 1.5.
constexpr int kFPF = 2;                                // k-groups (of 8 k) in flight per register set

__global__ void joint_frag_w_kernel(const float *__restrict__ w, int V, int J, int Jp, int Vp, float4 *__restrict__ wf)
{
    const int KG = Jp / 8;
    const long n = (long)(Vp / 32) * KG * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long g = idx >> 6;
        const int kg = (int)(g % KG), ct = (int)(g / KG);
        const int v = 32 * ct + (lane & 31), half = lane >> 5;
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 8 * kg + 2 * i + half;
            x[i] = (v < V && k < J) ? w[(size_t)v * J + k] : 0.f;
        }
        wf[idx] = make_float4(x[0], x[1], x[2], x[3]);
    }
}

template <bool LSE, int CT>
__global__ __launch_bounds__(512) void joint_fwd_frag_kernel(
    const float *__restrict__ ep, const float *__restrict__ pp, const float4 *__restrict__ wf, const float *__restrict__ bias,
    const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens, int B, int T, int U1, int J, int Jp, int V, int Vp,
    int act, float *__restrict__ out, JointLse lse)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int AS = Jp + 4;                            // row stride of the activation tile (floats)
    float *Ht = lds;                                  // [64][AS]
    const long M = (long)B * T * U1;
    const long m0 = (long)blockIdx.x * kBM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    float rm[32], rs[32];
    if (LSE) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { rm[i] = -3.0e38f; rs[i] = 0.f; }
    }
    if (llens != nullptr && tlens != nullptr) {
        int valid = 0;
        const long m = m0 + tid;
        if (tid < kBM && m < M) {
            const long bt = m / U1;
            const int u = (int)(m - bt * U1);
            const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
            valid = (t < llens[b]) && (u <= tlens[b]);
        }
        if (!__syncthreads_or(valid)) return;
    }
    // activation tile: Ht[row][k & 1][k >> 1]
    for (int row = wave; row < kBM; row += 8) {
        const long m = m0 + row;
        float *hr = Ht + (size_t)row * AS;
        if (m < M) {
            const long bt = m / U1;
            const int u = (int)(m - bt * U1);
            const long b = bt / T;
            const float *__restrict__ e = ep + (size_t)bt * J;
            const float *__restrict__ p = pp + ((size_t)b * U1 + u) * J;
            for (int k = lane; k < Jp; k += 64) hr[(k & 1) * (Jp / 2) + (k >> 1)] = (k < J) ? act_value(act, e[k] + p[k]) : 0.f;
        } else {
            for (int k = lane; k < Jp; k += 64) hr[k] = 0.f;
        }
    }
    __syncthreads();

    const int KG = Jp / 8;                            // Jp is a multiple of 32: KG a multiple of kFPF
    constexpr int kChunk = 8 * 32 * CT;
    const int nchunks = Vp / kChunk;
    const f32x4 *__restrict__ A0 = reinterpret_cast<const f32x4 *>(Ht + (size_t)l31 * AS + half * (Jp / 2));
    const f32x4 *__restrict__ A1 = reinterpret_cast<const f32x4 *>(Ht + (size_t)(l31 + 32) * AS + half * (Jp / 2));
    // register set P is loaded one block ahead ACROSS chunk boundaries: the last block of a chunk requests the first
    // block of the next chunk's W tile, so a chunk does not start by waiting out an L2 round trip (one per 32 768 MFMA
    // cycles otherwise)
    f32x4 pb[CT][kFPF], pa0[kFPF], pa1[kFPF], qb[CT][kFPF], qa0[kFPF], qa1[kFPF];
#define WR_LOADF(B_, b_, a0_, a1_, g_)                                                 \
        _Pragma("unroll") for (int i = 0; i < kFPF; ++i) {                             \
            _Pragma("unroll") for (int c = 0; c < CT; ++c) b_[c][i] = B_[((size_t)c * KG + (g_) + i) * 64]; \
            a0_[i] = A0[(g_) + i];                                                     \
            a1_[i] = A1[(g_) + i];                                                     \
        }
    {
        const f32x4 *__restrict__ Bfirst = reinterpret_cast<const f32x4 *>(wf) + (size_t)(wave * CT) * KG * 64 + lane;
        WR_LOADF(Bfirst, pb, pa0, pa1, 0)
    }
    for (int nc = 0; nc < nchunks; ++nc) {
        const int v0 = nc * kChunk;
        const int ct = (v0 >> 5) + wave * CT;
        f32x16 acc[2][CT];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[r][c] = (f32x16){0};
        const f32x4 *__restrict__ Bf = reinterpret_cast<const f32x4 *>(wf) + (size_t)ct * KG * 64 + lane;
        const f32x4 *__restrict__ Bnext = nc + 1 < nchunks ? Bf + (size_t)8 * CT * KG * 64 : Bf;
#define WR_MFMAF(b_, a0_, a1_)                                                          \
        _Pragma("unroll") for (int i = 0; i < kFPF; ++i)                               \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                              \
                _Pragma("unroll") for (int c = 0; c < CT; ++c) {                       \
                    acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_[i][q], b_[c][i][q], acc[0][c], 0, 0, 0); \
                    acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_[i][q], b_[c][i][q], acc[1][c], 0, 0, 0); \
                }
        for (int g = 0; g < KG; g += 2 * kFPF) {
            WR_LOADF(Bf, qb, qa0, qa1, g + kFPF)
            __builtin_amdgcn_sched_barrier(0);
            WR_MFMAF(pb, pa0, pa1)
            __builtin_amdgcn_sched_barrier(0);
            const bool wrap = g + 2 * kFPF >= KG;
            const int g2 = wrap ? 0 : g + 2 * kFPF;
            const f32x4 *__restrict__ B2 = wrap ? Bnext : Bf;
            WR_LOADF(B2, pb, pa0, pa1, g2)
            __builtin_amdgcn_sched_barrier(0);
            WR_MFMAF(qb, qa0, qa1)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WR_MFMAF
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = v0 + 32 * (wave * CT + c) + l31;
            const float bv = (col < V) ? bias[col] : 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const long m = m0 + row;
                    const float x = acc[rt][c][r] + bv;
                    if (m < M && col < V) out[(size_t)m * V + col] = x;
                    if (LSE) joint_lse_add(rm[rt * 16 + r], rs[rt * 16 + r], x, col < V, nc == 0 && c == 0);
                }
        }
    }
#undef WR_LOADF
    if (LSE) {
        __syncthreads();
        joint_lse_finish<8>(lse, lds, rm, rs, llens, tlens, out, m0, M, T, U1, V);
    }
}

// --------------------------------------------------------------- backward --
// dZ tile (64 cells x J) = dY tile (64 x V) * W (V x J), then * (1 - H^2).
// 8 waves (2 per SIMD), each 64 rows x 64 J-columns.  The dY slice (shared by all waves, streamed once from
// HBM) is staged v-major in LDS, double buffered, one barrier per 16-deep slice; the W fragments belong to
// one wave only and are loaded straight from L2 into a register ping-pong (no LDS, same scheme as forward).
constexpr int kBKv = 16;       // v-slice depth of the streamed dY operand
constexpr int kBwdWaves = 8;

__global__ __launch_bounds__(512) void joint_bwd_dz_kernel(
    const float *__restrict__ gout /* [M, V] */, const float *__restrict__ ep, const float *__restrict__ pp,
    const float *__restrict__ w /* [V, J] */, const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens,
    int B, int T, int U1, int J, int V, int act, float *__restrict__ dz /* [M, J] */, float *__restrict__ hout /* [M,J] or null */)
{
    __shared__ float As[2][kBKv][kHPad];               // dY slice, v-major
    const long M = (long)B * T * U1;
    const long m0 = (long)blockIdx.x * kBM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wcol = wave * 64;
    const bool wave_on = wcol < J;                     // waves beyond join_dim only help staging

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

    // staging map for the dY slice: 64 rows x 16 v = 256 float4 -> threads 0..255
    const int arow = (tid & 255) >> 2, av = (tid & 3) * 4;
    const long am = m0 + arow;
    const bool stager = tid < 256;
    const int nslices = (V + kBKv - 1) / kBKv;

    auto load_a = [&](int s, f32x4 &r) {
        const int v = s * kBKv + av;
        r = (f32x4){0, 0, 0, 0};
        if (stager && am < M) {
            const float *src = gout + (size_t)am * V + v;
            if (v + 3 < V && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) r = *reinterpret_cast<const f32x4 *>(src);
            else {
                if (v < V) r.x = src[0];
                if (v + 1 < V) r.y = src[1];
                if (v + 2 < V) r.z = src[2];
                if (v + 3 < V) r.w = src[3];
            }
        }
    };
    auto store_a = [&](int buf, const f32x4 &r) {
        if (stager) {
            As[buf][av + 0][arow] = r.x;
            As[buf][av + 1][arow] = r.y;
            As[buf][av + 2][arow] = r.z;
            As[buf][av + 3][arow] = r.w;
        }
    };
    // W fragment addresses: lane reads W[v + half][wcol + 32*c + l31]; columns/rows past the end are clamped
    // (their products are multiplied by zero-padded dY or discarded in the epilogue)
    int colc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int col = wcol + 32 * c + l31;
        colc[c] = col < J ? col : J - 1;
    }
    auto load_b = [&](int s, float (&b)[2][kBKv / 2]) {
#pragma unroll
        for (int i = 0; i < kBKv / 2; ++i) {
            int v = s * kBKv + 2 * i + half;
            v = v < V ? v : V - 1;
#pragma unroll
            for (int c = 0; c < 2; ++c) b[c][i] = w[(size_t)v * J + colc[c]];
        }
    };
    auto mfma_slice = [&](int buf, const float (&b)[2][kBKv / 2]) {
#pragma unroll
        for (int i = 0; i < kBKv / 2; ++i) {
            const float a0 = As[buf][2 * i + half][l31];
            const float a1 = As[buf][2 * i + half][32 + l31];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[c][i], acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[c][i], acc[1][c], 0, 0, 0);
            }
        }
    };

    float pb[2][kBKv / 2], qb[2][kBKv / 2];
    {
        f32x4 ra;
        load_a(0, ra);
        load_b(0, pb);
        store_a(0, ra);
    }
    __syncthreads();
    for (int s = 0; s < nslices; s += 2) {
        // even slice: compute from (As[0], pb) while slice s+1 is loaded into (regs -> As[1], qb)
        {
            f32x4 ra;
            const int sn = (s + 1 < nslices) ? s + 1 : s;
            load_a(sn, ra);
            load_b(sn, qb);
            __builtin_amdgcn_sched_barrier(0);
            if (wave_on) mfma_slice(0, pb);
            __builtin_amdgcn_sched_barrier(0);
            store_a(1, ra);
        }
        __syncthreads();
        if (s + 1 >= nslices) break;
        {
            f32x4 ra;
            const int sn = (s + 2 < nslices) ? s + 2 : s + 1;
            load_a(sn, ra);
            load_b(sn, pb);
            __builtin_amdgcn_sched_barrier(0);
            if (wave_on) mfma_slice(1, qb);
            __builtin_amdgcn_sched_barrier(0);
            store_a(0, ra);
        }
        __syncthreads();
    }
    if (!wave_on) return;
    // epilogue: dZ = dH * act'(z) (tanh: 1 - H^2), H recomputed per element
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const long m = m0 + row;
            if (m >= M) continue;
            const long bt = m / U1;
            const int u = (int)(m - bt * U1);
            const long b = bt / T;
            bool valid = true;
            if (llens != nullptr && tlens != nullptr) {
                const int t = (int)(bt - b * T);
                valid = (t < llens[b]) && (u <= tlens[b]);
            }
            const float *__restrict__ e = ep + (size_t)bt * J;
            const float *__restrict__ p = pp + ((size_t)b * U1 + u) * J;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int k = wcol + c * 32 + l31;
                if (k >= J) continue;
                float h, dh;
                act_value_grad(act, e[k] + p[k], h, dh);
                const float g = valid ? acc[i][c][r] * dh : 0.f;
                dz[(size_t)m * J + k] = g;
                if (hout) hout[(size_t)m * J + k] = valid ? h : 0.f;
            }
        }
    }
}

// ------------------------------------------------------ weight gradient --
// dW[v][k] = sum_m dY[m][v] * H[m][k]   and   db[v] = sum_m dY[m][v]      (M = B*T*U1 lattice cells)
// A reduction over millions of cells: one workgroup owns a slab of 128 vocabulary rows x all J columns
// (8 waves = 2 V-halves x 4 J-quarters, 2x4 MFMA tiles = 128 accumulator registers per lane) and one of
// `parts` contiguous ranges of cells; both operands are k(=cell)-major as stored (dY [M][V], H [M][J]), so
// every fragment is a coalesced 128-byte read straight into a register ping-pong -- no LDS.  The `parts`
// partial slabs are summed by a second, deterministic kernel (no float atomics).  Workgroups of the same
// cell range are adjacent in the grid so that the H rows they all read are served from L2 / Infinity Cache.
constexpr int kDwSlab = 128;
constexpr int kDwPF = 4;       // k-steps (pairs of cells) per register set

__global__ __launch_bounds__(512) void joint_bwd_dw_kernel(
    const float *__restrict__ gout /* [M, V] */, const float *__restrict__ h /* [M, J] */,
    const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens, int B, int T, int U1, int J, int V,
    int nslabs, long rows_per_part, float *__restrict__ part_dw /* [parts][V][J] */,
    float *__restrict__ part_db /* [parts][V] */)
{
    const long M = (long)B * T * U1;
    const int slab = blockIdx.x % nslabs, part = blockIdx.x / nslabs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wv = wave >> 2, wk = wave & 3;                  // V-half, J-quarter of this wave
    const int v0 = slab * kDwSlab + wv * 64;
    const int k0 = wk * 128;
    const long mbeg = (long)part * rows_per_part;
    long mend = mbeg + rows_per_part;
    mend = mend < M ? mend : M;
    const bool masked = (llens != nullptr && tlens != nullptr);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x16){0};
    float bsum[2] = {0.f, 0.f};

    int vcol[2], kcol[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int v = v0 + 32 * i + l31; vcol[i] = v < V ? v : V - 1; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int k = k0 + 32 * j + l31; kcol[j] = k < J ? k : J - 1; }

    // a lattice cell contributes only if it is inside the utterance's valid region (padded cells of dY may hold
    // anything when the caller did not produce it with the RNN-T gradient kernel)
    auto row_scale = [&](long m) -> float {
        if (m >= mend) return 0.f;
        if (!masked) return 1.f;
        const long bt = m / U1;
        const int u = (int)(m - bt * U1);
        const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
        return (t < llens[b] && u <= tlens[b]) ? 1.f : 0.f;
    };
    auto load_set = [&](long m, float (&a)[2][kDwPF], float (&bb)[4][kDwPF], float (&sc)[kDwPF]) {
#pragma unroll
        for (int q = 0; q < kDwPF; ++q) {
            long r = m + 2 * q + half;
            sc[q] = row_scale(r);
            r = r < M ? r : M - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i][q] = gout[(size_t)r * V + vcol[i]];
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j][q] = h[(size_t)r * J + kcol[j]];
        }
    };
    auto mfma_set = [&](const float (&a)[2][kDwPF], const float (&bb)[4][kDwPF], const float (&sc)[kDwPF]) {
#pragma unroll
        for (int q = 0; q < kDwPF; ++q) {
            const float a0 = a[0][q] * sc[q], a1 = a[1][q] * sc[q];
            bsum[0] += a0;
            bsum[1] += a1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bb[j][q], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bb[j][q], acc[1][j], 0, 0, 0);
            }
        }
    };

    float pa[2][kDwPF], pb[4][kDwPF], ps[kDwPF];
    float qa[2][kDwPF], qb[4][kDwPF], qs[kDwPF];
    constexpr int STEP = 2 * kDwPF;                            // cells per register set
    load_set(mbeg, pa, pb, ps);
    for (long m = mbeg; m < mend; m += 2 * STEP) {
        load_set(m + STEP, qa, qb, qs);                        // rows past mend are scaled by 0
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(pa, pb, ps);
        __builtin_amdgcn_sched_barrier(0);
        load_set(m + 2 * STEP, pa, pb, ps);
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(qa, qb, qs);
        __builtin_amdgcn_sched_barrier(0);
    }
    // partial slab: D layout col = lane&31 (-> k), row = (r&3) + 8*(r>>2) + 4*half (-> v)
    float *__restrict__ pd = part_dw + (size_t)part * V * J;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = v0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (v >= V) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + 32 * j + l31;
                if (k < J) pd[(size_t)v * J + k] = acc[i][j][r];
            }
        }
    if (wk == 0) {                                             // column sums: add the two cell-parities
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32, kWave);
            const int v = v0 + 32 * i + l31;
            if (half == 0 && v < V) part_db[(size_t)part * V + v] = tot;
        }
    }
}

__global__ void joint_dw_reduce_kernel(const float *__restrict__ part_dw, const float *__restrict__ part_db, int parts,
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

inline int dw_parts(int V)
{
    const int nslabs = (V + kDwSlab - 1) / kDwSlab;
    int parts = 256 / nslabs;                                  // one workgroup per CU
    return parts < 1 ? 1 : parts;
}

int joint_check(int B, int T, int U1, int J, int V, int activation = WR_ACT_TANH)
{
    WR_REQUIRE(B > 0 && T > 0 && U1 > 0 && J > 0 && V > 0, WR_EINVAL,
               "joint: B, T, U1, J, V must be positive (got %d,%d,%d,%d,%d)", B, T, U1, J, V);
    WR_REQUIRE(activation >= WR_ACT_TANH && activation <= WR_ACT_GELU, WR_EINVAL,
               "joint: activation code %d is not a wr_activation", activation);
    WR_REQUIRE(J % 4 == 0 && J <= 512, WR_EUNSUPPORTED,
               "joint: join_dim=%d not supported (must be a multiple of 4, at most 512)", J);
    WR_REQUIRE(((long)B * T * U1 + kBM - 1) / kBM < (1L << 31), WR_EUNSUPPORTED, "joint: too many lattice cells");
    return WR_OK;
}

}  // namespace
}  // namespace wr

using namespace wr;

extern "C" size_t wr_joint_workspace_bytes(int J, int V)
{
    if (J <= 0 || V <= 0) return 0;
    return align_up((size_t)joint_jpad(J) * joint_vpad(V) * sizeof(float), 256);
}

namespace {
// shared body of wr_joint_fwd / wr_joint_fwd_lse
int joint_fwd_launch(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                     const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T, int U1, int J, int V,
                     int act, float *out_d, void *workspace_d, size_t workspace_bytes, const JointLse *lse, hipStream_t st)
{
    const int Vp = joint_vpad(V), Jp = joint_jpad(J);
    WR_REQUIRE(workspace_bytes >= (size_t)Jp * Vp * sizeof(float), WR_EWORKSPACE, "joint_fwd: workspace too small");
    float *wt = static_cast<float *>(workspace_d);
    const long M = (long)B * T * U1;
    const dim3 grid((unsigned)((M + kBM - 1) / kBM));
    const size_t tile = (size_t)Jp * kHPad * sizeof(float);
    // default: the fragment-layout kernel; wr_tune_set(5, 1) selects the first forward kernel (kept for the equivalence test)
    if (tune_get(kTuneJointFwdVariant) != 1) {
        float4 *wf = static_cast<float4 *>(workspace_d);
        hipLaunchKernelGGL(joint_frag_w_kernel, dim3(1024), dim3(256), 0, st, w_out_d, V, J, Jp, Vp, wf);
        const size_t tile2 = (size_t)kBM * (Jp + 4) * sizeof(float);
        if (lse == nullptr) {
#define WR_LAUNCH_FRAG(LSE_, CT_, lds_, lse_)                                                                          \
            do {                                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_fwd_frag_kernel<LSE_, CT_>),             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_));                    \
                hipLaunchKernelGGL((joint_fwd_frag_kernel<LSE_, CT_>), grid, dim3(512), lds_, st, ep_d, pp_d, wf, b_out_d, \
                                   logit_lengths_d, target_lengths_d, B, T, U1, J, Jp, V, Vp, act, out_d, lse_);       \
            } while (0)
            WR_LAUNCH_FRAG(false, 1, tile2, JointLse{});
        } else {
            const size_t lds2 = tile2 > joint_lse_exchange_bytes(8) ? tile2 : joint_lse_exchange_bytes(8);
            (void)hipMemsetAsync(lse->repair, 0, sizeof(int32_t), st);
            WR_LAUNCH_FRAG(true, 1, lds2, *lse);
#undef WR_LAUNCH_FRAG
        }
        WR_CHECK_LAUNCH("joint_fwd_frag_kernel");
        return WR_OK;
    }
    hipLaunchKernelGGL(joint_transpose_w_kernel, dim3(Vp / 32, (Jp + 31) / 32), dim3(256), 0, st, w_out_d, V, J, Jp, Vp,
                       wt);
    WR_CHECK_LAUNCH("joint_transpose_w_kernel");
    if (lse == nullptr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_fwd_direct_kernel<8, 1, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile);
        hipLaunchKernelGGL((joint_fwd_direct_kernel<8, 1, false>), grid, dim3(512), tile, st, ep_d, pp_d, wt, b_out_d,
                           logit_lengths_d, target_lengths_d, B, T, U1, J, Jp, V, Vp, act, out_d, JointLse{});
    } else {
        // the statistics exchange reuses the activation tile's storage
        const size_t lds = tile > joint_lse_exchange_bytes(8) ? tile : joint_lse_exchange_bytes(8);
        (void)hipMemsetAsync(lse->repair, 0, sizeof(int32_t), st);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_fwd_direct_kernel<8, 1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((joint_fwd_direct_kernel<8, 1, true>), grid, dim3(512), lds, st, ep_d, pp_d, wt, b_out_d,
                           logit_lengths_d, target_lengths_d, B, T, U1, J, Jp, V, Vp, act, out_d, *lse);
    }
    WR_CHECK_LAUNCH("joint_fwd_direct_kernel");
    return WR_OK;
}
}  // namespace

extern "C" int wr_joint_fwd(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                            const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T, int U1,
                            int J, int V, int activation, float *out_d, void *workspace_d, size_t workspace_bytes,
                            void *stream)
{
    if (int rc = joint_check(B, T, U1, J, V, activation)) return rc;
    WR_REQUIRE(ep_d && pp_d && w_out_d && b_out_d && out_d && workspace_d, WR_EINVAL, "joint_fwd: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_fwd: pass both length arrays or neither");
    return joint_fwd_launch(ep_d, pp_d, w_out_d, b_out_d, logit_lengths_d, target_lengths_d, B, T, U1, J, V, activation,
                            out_d, workspace_d, workspace_bytes, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int wr_joint_fwd_lse(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                                const int32_t *logit_lengths_d, const int32_t *target_lengths_d,
                                const int32_t *targets_d, int B, int T, int U1, int J, int V, int activation, int blank,
                                float *out_d, void *workspace_d, size_t workspace_bytes, void *rnnt_workspace_d,
                                size_t rnnt_workspace_bytes, void *stream)
{
    if (int rc = joint_check(B, T, U1, J, V, activation)) return rc;
    WR_REQUIRE(ep_d && pp_d && w_out_d && b_out_d && out_d && workspace_d && rnnt_workspace_d, WR_EINVAL,
               "joint_fwd_lse: null pointer argument");
    WR_REQUIRE(logit_lengths_d && target_lengths_d, WR_EINVAL, "joint_fwd_lse: both length arrays are required");
    WR_REQUIRE(targets_d || U1 == 1, WR_EINVAL, "joint_fwd_lse: targets is null");
    WR_REQUIRE(blank >= 0 && blank < V, WR_EINVAL, "joint_fwd_lse: blank %d out of range [0,%d)", blank, V);
    WR_REQUIRE(U1 <= kRnntMaxCols, WR_EUNSUPPORTED, "joint_fwd_lse: U1=%d exceeds the loss's limit of %d", U1, kRnntMaxCols);
    const RnntWs w = rnnt_ws_layout(B, T, U1);
    WR_REQUIRE(rnnt_workspace_bytes >= w.total, WR_EWORKSPACE, "joint_fwd_lse: RNN-T workspace %zu < required %zu",
               rnnt_workspace_bytes, w.total);
    char *rws = static_cast<char *>(rnnt_workspace_d);
    JointLse lse{targets_d, blank, w.S, reinterpret_cast<float2 *>(rws + w.lp_off), reinterpret_cast<float *>(rws + w.denom_off),
                 reinterpret_cast<int32_t *>(rws + w.flag_off)};
    return joint_fwd_launch(ep_d, pp_d, w_out_d, b_out_d, logit_lengths_d, target_lengths_d, B, T, U1, J, V, activation,
                            out_d, workspace_d, workspace_bytes, &lse, static_cast<hipStream_t>(stream));
}

extern "C" int wr_joint_bwd_dz(const float *gout_d, const float *ep_d, const float *pp_d, const float *w_out_d,
                               const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T, int U1,
                               int J, int V, int activation, float *dz_d, float *h_d, void *stream)
{
    if (int rc = joint_check(B, T, U1, J, V, activation)) return rc;
    WR_REQUIRE(gout_d && ep_d && pp_d && w_out_d && dz_d, WR_EINVAL, "joint_bwd_dz: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_bwd_dz: pass both length arrays or neither");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (V % 4 == 0 && V >= 16 && J % 4 == 0 && tune_get(kTuneDzExact) != 1)   // default: 256 x 256 block tiling (knob 10 = 1: 64-cell tiling)
        return joint_bwd_dz_block(gout_d, ep_d, pp_d, w_out_d, logit_lengths_d, target_lengths_d, B, T, U1, J, V, activation, dz_d, h_d, st);
    const long M = (long)B * T * U1;
    const dim3 grid((unsigned)((M + kBM - 1) / kBM));
    hipLaunchKernelGGL(joint_bwd_dz_kernel, grid, dim3(64 * kBwdWaves), 0, st, gout_d, ep_d, pp_d, w_out_d,
                       logit_lengths_d, target_lengths_d, B, T, U1, J, V, activation, dz_d, h_d);
    WR_CHECK_LAUNCH("joint_bwd_dz_kernel");
    return WR_OK;
}

extern "C" size_t wr_joint_dw_workspace_bytes(int J, int V)
{
    if (J <= 0 || V <= 0) return 0;
    return align_up((size_t)dw_parts(V) * ((size_t)V * J + V) * sizeof(float), 256);
}

extern "C" int wr_joint_bwd_dw(const float *gout_d, const float *h_d, const int32_t *logit_lengths_d,
                               const int32_t *target_lengths_d, int B, int T, int U1, int J, int V, float *dw_d,
                               float *db_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    if (int rc = joint_check(B, T, U1, J, V)) return rc;
    WR_REQUIRE(gout_d && h_d && dw_d && workspace_d, WR_EINVAL, "joint_bwd_dw: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_bwd_dw: pass both length arrays or neither");
    const int parts = dw_parts(V);
    const size_t need = (size_t)parts * ((size_t)V * J + V) * sizeof(float);
    WR_REQUIRE(workspace_bytes >= need, WR_EWORKSPACE, "joint_bwd_dw: workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (V % 4 == 0 && J % 4 == 0 && tune_get(kTuneDwExact) != 1)     // default: 256 x 256 block tiling (knob 9 = 1: slabs)
        return joint_bwd_dw_block(gout_d, h_d, logit_lengths_d, target_lengths_d, B, T, U1, J, V, parts, dw_d, db_d,
                                  static_cast<float *>(workspace_d), st);
    const long M = (long)B * T * U1;
    const int nslabs = (V + kDwSlab - 1) / kDwSlab;
    const long per = 4 * kDwPF;                                // the main loop advances in units of two register sets
    long rows_per_part = (M + parts - 1) / parts;
    rows_per_part = (rows_per_part + per - 1) / per * per;
    float *part_dw = static_cast<float *>(workspace_d);
    float *part_db = part_dw + (size_t)parts * V * J;
    hipLaunchKernelGGL(joint_bwd_dw_kernel, dim3(nslabs * parts), dim3(512), 0, st, gout_d, h_d, logit_lengths_d,
                       target_lengths_d, B, T, U1, J, V, nslabs, rows_per_part, part_dw, part_db);
    WR_CHECK_LAUNCH("joint_bwd_dw_kernel");
    hipLaunchKernelGGL(joint_dw_reduce_kernel, dim3(1024), dim3(256), 0, st, part_dw, part_db, parts, (long)V * J, V, dw_d,
                       db_d);
    WR_CHECK_LAUNCH("joint_dw_reduce_kernel");
    return WR_OK;
}
