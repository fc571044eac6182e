// Shared device/host helpers for libwr_mi355x (gfx950 only; wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "wr_api.h"

namespace wr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNegInf = -__builtin_huge_valf();

// ---- host-side error reporting -------------------------------------------
void set_error(const char *fmt, ...);

#define WR_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ::wr::set_error(__VA_ARGS__);    \
            return (code);                   \
        }                                    \
    } while (0)

#define WR_CHECK_LAUNCH(what)                                                       \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            ::wr::set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e_)); \
            return WR_ELAUNCH;                                                      \
        }                                                                           \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- process-wide tuning knobs (wr_tune_set; defaults are the measured best) ----
enum TuneKey { kTuneLseBlocksPerCu = 0, kTuneGradBlocksPerCu = 1, kTuneNonTemporal = 2, kTuneGradUnroll = 3, kTuneLseUnroll = 4, kTuneJointFwdVariant = 5, kTuneLaneGemmTile = 6, kTuneSplitParts = 7, kTuneDzTile = 8, kTuneDwExact = 9, kTuneDzExact = 10, kTuneFoldProj = 11, kTuneSplitFwdCells = 12, kTuneSplitFwdStore = 13, kTuneCount = 14 };
int tune_get(int key);

// joint_split.hip: exact-fp32 activation gradient, 256 x 256 block tiling
int joint_bwd_dz_block(const float *gout_d, const float *ep_d, const float *pp_d, const float *w_d, const int32_t *llens_d,
                       const int32_t *tlens_d, int B, int T, int U1, int J, int V, int act, float *dz_d, float *h_d, hipStream_t st);

// joint_split.hip: exact-fp32 weight gradient, 256 x 256 block tiling (the partial blocks go to `part_dw`, sized by the
// caller for `max_parts` parts of V*J + V floats)
int joint_bwd_dw_block(const float *gout_d, const float *h_d, const int32_t *llens_d, const int32_t *tlens_d, int B, int T,
                       int U1, int J, int V, int max_parts, float *dw_d, float *db_d, float *part_dw, hipStream_t st);

// ---- RNN-T loss workspace (rnnt_loss.hip; the joiner's fused row-statistics epilogue writes into it too) ----
constexpr int kRnntMaxCols = 1024;      // lattice columns U+1: one lane per column in the sweep (16 waves)

struct RnntWs {
    int K;            // label columns per lane in the sweep (lane l owns u = l, l+64, ...)
    int S;            // number of anti-diagonals per utterance
    size_t lp_off, alpha_off, beta_off, denom_off, ll_off, cost_off, dump_off, flag_off, total;
};

__host__ __device__ inline int rnnt_cols_per_lane(int U1max) { return (U1max + kWave - 1) / kWave; }

inline RnntWs rnnt_ws_layout(int B, int Tmax, int U1max)
{
    RnntWs w;
    w.K = rnnt_cols_per_lane(U1max);
    w.S = Tmax + U1max - 1;          // anti-diagonals s = t + u
    const size_t diag = (size_t)B * w.S * U1max;
    size_t off = 0;
    w.lp_off = off;    off = align_up(off + diag * sizeof(float2), 256);
    w.alpha_off = off; off = align_up(off + diag * sizeof(double), 256);
    w.beta_off = off;  off = align_up(off + diag * sizeof(double), 256);
    w.denom_off = off; off = align_up(off + (size_t)B * Tmax * U1max * sizeof(float), 256);
    w.ll_off = off;    off = align_up(off + (size_t)B * sizeof(double), 256);
    w.cost_off = off;  off = align_up(off + (size_t)B * sizeof(double), 256);
    w.dump_off = off;  off = align_up(off + (size_t)B * 2 * kRnntMaxCols * sizeof(double), 256);
    w.flag_off = off;  off = align_up(off + 64, 256);   // "row statistics need the stand-alone pass" (joint_lse.hpp)
    w.total = off;
    return w;
}

// rnnt_loss.hip: lattice sweeps over a workspace whose row statistics (denom, skip/emit log-probs) are in place
void rnnt_launch_sweep(const RnntWs &w, char *ws, const int32_t *llens, const int32_t *tlens, int B, int Tmax, int U1max,
                       float *costs, hipStream_t st);

// ---- device helpers ---------------------------------------------------------
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// Joiner activations (wenet/utils/common.py:228-242 get_activation; codes WR_ACT_* of wr_api.h).  `act` is a kernel
// argument (wave-uniform: the switch is scalar control flow).  Value and derivative w.r.t. the pre-activation z as
// torch computes them: Hardtanh passes the gradient strictly inside (-1, 1), ReLU for z > 0, SELU / SiLU / GELU(erf)
// by their closed forms.
constexpr float kSeluAlpha = 1.6732632423543772848170429916717f;
constexpr float kSeluScale = 1.0507009873554804934193349852946f;

__device__ __forceinline__ float act_value(int act, float z)
{
    switch (act) {
    case WR_ACT_RELU: return fmaxf(z, 0.f);
    case WR_ACT_HARDTANH: return fminf(fmaxf(z, -1.f), 1.f);
    case WR_ACT_SELU: return kSeluScale * (z > 0.f ? z : kSeluAlpha * expm1f(z));
    case WR_ACT_SWISH: return z / (1.f + expf(-z));
    case WR_ACT_GELU: return 0.5f * z * (1.f + erff(z * 0.70710678118654752f));
    default: return tanhf(z);
    }
}

__device__ __forceinline__ void act_value_grad(int act, float z, float &h, float &d)
{
    switch (act) {
    case WR_ACT_RELU: h = fmaxf(z, 0.f); d = z > 0.f ? 1.f : 0.f; break;
    case WR_ACT_HARDTANH: h = fminf(fmaxf(z, -1.f), 1.f); d = (z > -1.f && z < 1.f) ? 1.f : 0.f; break;
    case WR_ACT_SELU: {
        const float e = kSeluAlpha * expf(z);
        h = kSeluScale * (z > 0.f ? z : kSeluAlpha * expm1f(z));
        d = kSeluScale * (z > 0.f ? 1.f : e);
        break;
    }
    case WR_ACT_SWISH: {
        const float sg = 1.f / (1.f + expf(-z));
        h = z * sg;
        d = sg * (1.f + z * (1.f - sg));
        break;
    }
    case WR_ACT_GELU: {
        const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752f));
        h = z * cdf;
        d = cdf + z * 0.39894228040143267794f * expf(-0.5f * z * z);
        break;
    }
    default: h = tanhf(z); d = 1.f - h * h; break;
    }
}

// log(exp(a)+exp(b)) in the natural-log domain; -inf safe.
__device__ __forceinline__ float log_add_exp(float a, float b)
{
    const float m = fmaxf(a, b);
    const float d = -fabsf(a - b);                 // NaN when both are -inf
    const float r = m + kLn2 * fast_log2(1.0f + fast_exp2(d * kLog2e));
    return (m == kNegInf) ? kNegInf : r;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Shift a value one lane up (lane l receives lane l-1; lane 0 receives `fill`).
// gfx9 DPP wave_shr:1 -- one VALU op, no LDS crossbar round trip.
__device__ __forceinline__ float lane_shift_up(float v, float fill)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                           0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

// Lane l receives lane l+1; lane 63 receives `fill`.  (DPP wave_shl:1)
__device__ __forceinline__ float lane_shift_down(float v, float fill)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                           0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// All-lanes max / min through DPP butterflies inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror: four VALU ops, no LDS crossbar) and four readlanes across the rows.
__device__ __forceinline__ float wave_allmax_dpp(float v)
{
#define WR_DPP_F(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
    v = fmaxf(v, WR_DPP_F(v, 0xB1));               // quad_perm:[1,0,3,2]
    v = fmaxf(v, WR_DPP_F(v, 0x4E));               // quad_perm:[2,3,0,1]
    v = fmaxf(v, WR_DPP_F(v, 0x141));              // row_half_mirror
    v = fmaxf(v, WR_DPP_F(v, 0x140));              // row_mirror
#undef WR_DPP_F
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// sum over the wave, the same value in every lane: butterflies inside the rows of 16 (both partners add the same two
// numbers, a + b == b + a), then the four row sums in a fixed order
__device__ __forceinline__ float wave_allsum_dpp(float v)
{
#define WR_DPP_F(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
    v += WR_DPP_F(v, 0xB1);
    v += WR_DPP_F(v, 0x4E);
    v += WR_DPP_F(v, 0x141);
    v += WR_DPP_F(v, 0x140);
#undef WR_DPP_F
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ int wave_allmin_dpp(int v)
{
#define WR_DPP_I(x, ctrl) __builtin_amdgcn_update_dpp(0, x, ctrl, 0xf, 0xf, false)
    v = min(v, WR_DPP_I(v, 0xB1));
    v = min(v, WR_DPP_I(v, 0x4E));
    v = min(v, WR_DPP_I(v, 0x141));
    v = min(v, WR_DPP_I(v, 0x140));
#undef WR_DPP_I
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return min(min(r0, r1), min(r2, r3));
}
// All-reduce over each aligned group of 32 lanes (a half-wave): four DPP butterflies inside the rows of 16 and one
// cross-row exchange (lane ^ 16).  OP is applied to (value, partner value).
template <typename OP>
__device__ __forceinline__ float half_allreduce_f(float v, OP op)
{
#define WR_DPP_F(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
    v = op(v, WR_DPP_F(v, 0xB1));                  // quad_perm:[1,0,3,2]
    v = op(v, WR_DPP_F(v, 0x4E));                  // quad_perm:[2,3,0,1]
    v = op(v, WR_DPP_F(v, 0x141));                 // row_half_mirror
    v = op(v, WR_DPP_F(v, 0x140));                 // row_mirror
#undef WR_DPP_F
    return op(v, __shfl_xor(v, 16, kWave));
}
template <typename OP>
__device__ __forceinline__ int half_allreduce_i(int v, OP op)
{
#define WR_DPP_I(x, ctrl) __builtin_amdgcn_update_dpp(0, x, ctrl, 0xf, 0xf, false)
    v = op(v, WR_DPP_I(v, 0xB1));
    v = op(v, WR_DPP_I(v, 0x4E));
    v = op(v, WR_DPP_I(v, 0x141));
    v = op(v, WR_DPP_I(v, 0x140));
#undef WR_DPP_I
    return op(v, __shfl_xor(v, 16, kWave));
}

// wave-wide argmax with "first index on ties"; result in every lane
__device__ __forceinline__ void wave_argmax_dpp(float &val, int &idx)
{
    const float m = wave_allmax_dpp(val);
    idx = wave_allmin_dpp(val == m ? idx : 0x7fffffff);
    val = m;
}

// block-wide argmax with "first index on ties"
__device__ __forceinline__ void block_argmax(float &val, int &idx, float *sv, int *si)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(val, o, kWave);
        const int oi = __shfl_xor(idx, o, kWave);
        if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
    }
    if (lane == 0) { sv[wave] = val; si[wave] = idx; }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    val = sv[0]; idx = si[0];
    for (int w = 1; w < nw; ++w)
        if (sv[w] > val || (sv[w] == val && si[w] < idx)) { val = sv[w]; idx = si[w]; }
    __syncthreads();
}

__device__ __forceinline__ float block_max(float v, float *sv)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    v = wave_max(v);
    if (lane == 0) sv[wave] = v;
    __syncthreads();
    float r = sv[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, sv[w]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ float block_sum(float v, float *sv)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    v = wave_sum(v);
    if (lane == 0) sv[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sv[w];
    __syncthreads();
    return r;
}


}  // namespace wr
