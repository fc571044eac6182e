// Transducer joint network for MI355X (gfx950): the one dense contraction on
// the path.  Replaces TransducerJoint.forward (wenet/transducer/joint.py:45-70)
//     out = ffn_out(tanh(enc_ffn(enc)[:, :, None, :] + pred_ffn(pred)[:, None, :, :]))
// from the point where the two small pre-join projections are available:
//     ep = enc_ffn(enc)  [B, T,  J]      pp = pred_ffn(pred)  [B, U1, J]
// (those are plain library GEMMs on tiny tensors and stay with rocBLAS).
//
// Forward  joint_fwd_kernel      out[m, v] = sum_k tanh(ep[bt(m),k] + pp[bu(m),k]) * W[v,k] + bias[v]
//   M = B*T*U1 lattice cells (4.83 M at the BASELINE shape), K = J = 512, N = V.
//   One workgroup owns 64 consecutive cells and ALL V columns: the activation tile
//   H = tanh(ep + pp) (64 x J) is computed ONCE into LDS (k-major, +1 padded) and
//   never exists in HBM (the reference materialises it: 9.9 GB + its autograd copy);
//   W^T is streamed from L2 / Infinity Cache in 8-deep k-slices, double buffered.
//   Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32, bit-identical to an fmaf chain),
//   4 waves x (64 rows x 64 cols) per 256-column chunk.  MFMA-bound:
//   2*M*J*V flop = 24.74 TFLOP forward at the BASELINE shape vs 157.3 TFLOP/s.
//
// Backward joint_bwd_dz_kernel   dZ[m,k] = (sum_v dY[m,v] * W[v,k]) * (1 - H[m,k]^2)
//   same tiling transposed: one workgroup owns 64 cells and all J columns, streams
//   dY (the RNN-T gradient) once in 16-deep v-slices; H is recomputed, optionally
//   written out for the weight-gradient GEMM.  d ep = sum_u dZ, d pp = sum_t dZ and
//   dW = dY^T H are left to library reductions/GEMMs on the host side.
#include "wr_common.hpp"

namespace wr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 64;        // lattice cells per workgroup
constexpr int kBN = 256;       // output columns per chunk (4 waves x 64)
constexpr int kBK = 8;         // k-slice depth of the streamed operand (forward)
constexpr int kHPad = kBM + 1; // k-major activation tile row stride (floats)

inline int joint_vpad(int V) { return (V + kBN - 1) / kBN * kBN; }
inline int joint_jpad(int J) { return (J + kBK - 1) / kBK * kBK; }   // forward k-depth, zero padded

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

// Fill the k-major activation tile: Ht[k][row] = tanh(ep[bt,k] + pp[bu,k]) for the 64 cells m0..m0+63.
__device__ __forceinline__ void fill_h_tile(float *__restrict__ Ht, const float *__restrict__ ep,
                                            const float *__restrict__ pp, long m0, long M, int T, int U1, int J,
                                            int Jp)
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
            for (int k = lane; k < Jp; k += 64) Ht[k * kHPad + row] = (k < J) ? tanhf(e[k] + p[k]) : 0.f;
        } else {
            for (int k = lane; k < Jp; k += 64) Ht[k * kHPad + row] = 0.f;
        }
    }
}

// ---------------------------------------------------------------- forward --
__global__ __launch_bounds__(256) void joint_fwd_kernel(
    const float *__restrict__ ep, const float *__restrict__ pp, const float *__restrict__ wt /* [J, Vp] */,
    const float *__restrict__ bias, const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens,
    int B, int T, int U1, int J, int Jp, int V, int Vp, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Ht = lds;                                  // [Jp][65]
    float *Ws = lds + (size_t)Jp * kHPad;             // [2][kBK][kBN]
    const long M = (long)B * T * U1;
    const long m0 = (long)blockIdx.x * kBM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // Skip tiles that lie entirely in the padded region of the lattice (never read by the loss).
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

    fill_h_tile(Ht, ep, pp, m0, M, T, U1, J, Jp);

    const int nslices = Jp / kBK;
    const int nchunks = Vp / kBN;
    // staging map: 256 threads x 2 float4 cover one [8][256] slice
    const int srow = tid >> 6;            // 0..3 (+4 for the second half)
    const int scol = (tid & 63) * 4;
    const int half = lane >> 5, l31 = lane & 31;
    const int wcol = wave * 64;

    for (int nc = 0; nc < nchunks; ++nc) {
        const int v0 = nc * kBN;
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        // prologue: slice 0 -> buffer 0
        {
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(wt + (size_t)srow * Vp + v0 + scol);
            const f32x4 r1 = *reinterpret_cast<const f32x4 *>(wt + (size_t)(srow + 4) * Vp + v0 + scol);
            __syncthreads();              // previous chunk's readers are done with both buffers (and Ht is filled)
            *reinterpret_cast<f32x4 *>(Ws + srow * kBN + scol) = r0;
            *reinterpret_cast<f32x4 *>(Ws + (srow + 4) * kBN + scol) = r1;
        }
        __syncthreads();
        for (int s = 0; s < nslices; ++s) {
            const float *cur = Ws + (s & 1) * (kBK * kBN);
            float *nxt = Ws + ((s + 1) & 1) * (kBK * kBN);
            f32x4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0};
            const bool more = (s + 1 < nslices);
            if (more) {
                const int k = (s + 1) * kBK;
                r0 = *reinterpret_cast<const f32x4 *>(wt + (size_t)(k + srow) * Vp + v0 + scol);
                r1 = *reinterpret_cast<const f32x4 *>(wt + (size_t)(k + srow + 4) * Vp + v0 + scol);
            }
            const int kbase = s * kBK;
#pragma unroll
            for (int kk = 0; kk < kBK; kk += 2) {
                const float a0 = Ht[(kbase + kk + half) * kHPad + l31];
                const float a1 = Ht[(kbase + kk + half) * kHPad + 32 + l31];
                const float b0 = cur[(kk + half) * kBN + wcol + l31];
                const float b1 = cur[(kk + half) * kBN + wcol + 32 + l31];
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
            }
            if (more) {
                *reinterpret_cast<f32x4 *>(nxt + srow * kBN + scol) = r0;
                *reinterpret_cast<f32x4 *>(nxt + (srow + 4) * kBN + scol) = r1;
            }
            __syncthreads();
        }
        // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const int c0 = v0 + wcol + l31, c1 = c0 + 32;
        const float bias0 = (c0 < V) ? bias[c0] : 0.f;
        const float bias1 = (c1 < V) ? bias[c1] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const long ma = m0 + row, mb = m0 + 32 + row;
            if (ma < M) {
                if (c0 < V) out[(size_t)ma * V + c0] = acc00[r] + bias0;
                if (c1 < V) out[(size_t)ma * V + c1] = acc01[r] + bias1;
            }
            if (mb < M) {
                if (c0 < V) out[(size_t)mb * V + c0] = acc10[r] + bias0;
                if (c1 < V) out[(size_t)mb * V + c1] = acc11[r] + bias1;
            }
        }
    }
}

// --------------------------------------------------------------- backward --
// dZ tile (64 cells x J) = dY tile (64 x V) * W (V x J), then * (1 - H^2).
// 4 waves, each 64 rows x (J/4) columns; J <= 512 -> at most 2 x 4 MFMA tiles per wave.
constexpr int kBKv = 16;       // v-slice depth of the streamed dY / W operands

template <int NT /* 32-col tiles per wave = J/128 */>
__global__ __launch_bounds__(256) void joint_bwd_dz_kernel(
    const float *__restrict__ gout /* [M, V] */, const float *__restrict__ ep, const float *__restrict__ pp,
    const float *__restrict__ w /* [V, J] */, const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens,
    int B, int T, int U1, int J, int V, float *__restrict__ dz /* [M, J] */, float *__restrict__ hout /* [M,J] or null */)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                                   // [2][kBKv][65]   dY slice, v-major
    constexpr int JP = NT * 128;                       // W slice row stride (J zero padded)
    float *Bs = lds + 2 * kBKv * kHPad;                // [2][kBKv][JP]   W slice
    const long M = (long)B * T * U1;
    const long m0 = (long)blockIdx.x * kBM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wcol = wave * (NT * 32);

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x16){0};

    // staging maps
    //  A: 64 rows x 16 v  = 1024 floats -> 4 per thread: row = tid>>2, v4 = (tid&3)*4  (16-B pieces of a dY row)
    const int arow = tid >> 2, av = (tid & 3) * 4;
    const long am = m0 + arow;
    //  B: 16 v x J floats -> J/64 float4 per thread per ... generic loop below
    const int nslices = (V + kBKv - 1) / kBKv;
    const int b4_per_row = J / 4;                       // float4 per W row
    const int b4_total = kBKv * b4_per_row;             // float4 per slice

    auto load_a = [&](int s, f32x4 &r) {
        const int v = s * kBKv + av;
        r = (f32x4){0, 0, 0, 0};
        if (am < M) {
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
    auto store_a = [&](float *dst, const f32x4 &r) {
        dst[(av + 0) * kHPad + arow] = r.x;
        dst[(av + 1) * kHPad + arow] = r.y;
        dst[(av + 2) * kHPad + arow] = r.z;
        dst[(av + 3) * kHPad + arow] = r.w;
    };
    constexpr int kBMax = 8;                            // float4 per thread per slice at J=512: 16*128/256
    auto load_b = [&](int s, f32x4 (&r)[kBMax]) {
#pragma unroll
        for (int i = 0; i < kBMax; ++i) {
            const int idx = tid + i * 256;
            r[i] = (f32x4){0, 0, 0, 0};
            if (idx < b4_total) {
                const int vr = idx / b4_per_row, c4 = idx - vr * b4_per_row;
                const int v = s * kBKv + vr;
                if (v < V) r[i] = *reinterpret_cast<const f32x4 *>(w + (size_t)v * J + c4 * 4);
            }
        }
    };
    auto store_b = [&](float *dst, const f32x4 (&r)[kBMax]) {
#pragma unroll
        for (int i = 0; i < kBMax; ++i) {
            const int idx = tid + i * 256;
            if (idx < b4_total) {
                const int vr = idx / b4_per_row, c4 = idx - vr * b4_per_row;
                *reinterpret_cast<f32x4 *>(dst + (size_t)vr * JP + c4 * 4) = r[i];
            }
        }
    };
    if (J < JP) {                                       // zero the never-written pad columns of both buffers once
        for (int idx = tid; idx < 2 * kBKv * (JP - J); idx += 256) {
            const int vr = idx / (JP - J), c = J + idx - vr * (JP - J);
            Bs[(size_t)vr * JP + c] = 0.f;
        }
    }

    {
        f32x4 ra; f32x4 rb[kBMax];
        load_a(0, ra); load_b(0, rb);
        store_a(As, ra); store_b(Bs, rb);
    }
    __syncthreads();
    for (int s = 0; s < nslices; ++s) {
        const float *ca = As + (s & 1) * (kBKv * kHPad);
        const float *cb = Bs + (size_t)(s & 1) * (kBKv * JP);
        float *na = As + ((s + 1) & 1) * (kBKv * kHPad);
        float *nb = Bs + (size_t)((s + 1) & 1) * (kBKv * JP);
        f32x4 ra; f32x4 rb[kBMax];
        const bool more = (s + 1 < nslices);
        if (more) { load_a(s + 1, ra); load_b(s + 1, rb); }
#pragma unroll
        for (int kk = 0; kk < kBKv; kk += 2) {
            const float a0 = ca[(kk + half) * kHPad + l31];
            const float a1 = ca[(kk + half) * kHPad + 32 + l31];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float b = cb[(size_t)(kk + half) * JP + wcol + j * 32 + l31];
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][j], 0, 0, 0);
            }
        }
        if (more) { store_a(na, ra); store_b(nb, rb); }
        __syncthreads();
    }
    // epilogue: dZ = dH * (1 - H^2), H recomputed per element
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
            for (int j = 0; j < NT; ++j) {
                const int k = wcol + j * 32 + l31;
                if (k >= J) continue;
                const float h = tanhf(e[k] + p[k]);
                const float g = valid ? acc[i][j][r] * (1.f - h * h) : 0.f;
                dz[(size_t)m * J + k] = g;
                if (hout) hout[(size_t)m * J + k] = valid ? h : 0.f;
            }
        }
    }
}

int joint_check(int B, int T, int U1, int J, int V)
{
    WR_REQUIRE(B > 0 && T > 0 && U1 > 0 && J > 0 && V > 0, WR_EINVAL,
               "joint: B, T, U1, J, V must be positive (got %d,%d,%d,%d,%d)", B, T, U1, J, V);
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

extern "C" int wr_joint_fwd(const float *ep_d, const float *pp_d, const float *w_out_d, const float *b_out_d,
                            const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T, int U1,
                            int J, int V, float *out_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    if (int rc = joint_check(B, T, U1, J, V)) return rc;
    WR_REQUIRE(ep_d && pp_d && w_out_d && b_out_d && out_d && workspace_d, WR_EINVAL, "joint_fwd: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_fwd: pass both length arrays or neither");
    const int Vp = joint_vpad(V), Jp = joint_jpad(J);
    WR_REQUIRE(workspace_bytes >= (size_t)Jp * Vp * sizeof(float), WR_EWORKSPACE, "joint_fwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *wt = static_cast<float *>(workspace_d);
    hipLaunchKernelGGL(joint_transpose_w_kernel, dim3(Vp / 32, (Jp + 31) / 32), dim3(256), 0, st, w_out_d, V, J, Jp, Vp,
                       wt);
    WR_CHECK_LAUNCH("joint_transpose_w_kernel");
    const size_t lds = ((size_t)Jp * kHPad + 2 * kBK * kBN) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    const long M = (long)B * T * U1;
    hipLaunchKernelGGL(joint_fwd_kernel, dim3((unsigned)((M + kBM - 1) / kBM)), dim3(256), lds, st, ep_d, pp_d, wt,
                       b_out_d, logit_lengths_d, target_lengths_d, B, T, U1, J, Jp, V, Vp, out_d);
    WR_CHECK_LAUNCH("joint_fwd_kernel");
    return WR_OK;
}

extern "C" int wr_joint_bwd_dz(const float *gout_d, const float *ep_d, const float *pp_d, const float *w_out_d,
                               const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int T, int U1,
                               int J, int V, float *dz_d, float *h_d, void *stream)
{
    if (int rc = joint_check(B, T, U1, J, V)) return rc;
    WR_REQUIRE(gout_d && ep_d && pp_d && w_out_d && dz_d, WR_EINVAL, "joint_bwd_dz: null pointer argument");
    WR_REQUIRE((logit_lengths_d == nullptr) == (target_lengths_d == nullptr), WR_EINVAL,
               "joint_bwd_dz: pass both length arrays or neither");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long M = (long)B * T * U1;
    const dim3 grid((unsigned)((M + kBM - 1) / kBM));
    const int NTr = (J + 127) / 128;
    const size_t lds = (2 * kBKv * kHPad + 2 * (size_t)kBKv * NTr * 128) * sizeof(float);
#define WR_LAUNCH_DZ(NT)                                                                                         \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(joint_bwd_dz_kernel<NT>),                                  \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
    hipLaunchKernelGGL((joint_bwd_dz_kernel<NT>), grid, dim3(256), lds, st, gout_d, ep_d, pp_d, w_out_d,          \
                       logit_lengths_d, target_lengths_d, B, T, U1, J, V, dz_d, h_d)
    switch (NTr) {
        case 1: WR_LAUNCH_DZ(1); break;
        case 2: WR_LAUNCH_DZ(2); break;
        case 3: WR_LAUNCH_DZ(3); break;
        default: WR_LAUNCH_DZ(4); break;
    }
#undef WR_LAUNCH_DZ
    WR_CHECK_LAUNCH("joint_bwd_dz_kernel");
    return WR_OK;
}
