// RNN-T loss + gradient for MI355X (gfx950), replacing the
// torchaudio.functional.rnnt_loss call sites of the reference
// (wenet/transducer/transducer.py:142-147 and :296-301).  See include/wr_api.h.
//
// Three device passes over a batch of variable-length utterances packed into
// one grid (mathematics: SURVEY.md App. A.1):
//
//   pass 1  rnnt_lse_kernel    one wave per lattice cell (b,t,u): streams the
//           V logits of the cell once (16-B loads, online max/sum), writes
//           denom(t,u) and the two log-probs the lattice needs
//           (skip = blank, emit = label) into a *diagonal-skewed* array.
//           HBM-bound: 4*V bytes read per cell.
//   pass 2  rnnt_sweep_kernel  one workgroup per (utterance, direction), one lane per label column
//           (64 * ceil(U1max/64) threads).  At step s every lane works on the cell (t = s - u, u) of
//           anti-diagonal s; from diagonal s-1 it needs its own previous value and its neighbour
//           lane's: inside a wave one DPP wave shift, across a wave boundary an 8-byte LDS slot
//           (double buffered) and one s_barrier per step.  Because pass 1 stored the log-probs skewed
//           by the same rule, step s reads one contiguous row; PF rows are kept in flight under counted
//           waits.  State in fp64, the bounded log1p(exp(-|a-b|)) term in fp32.  Latency-bound
//           (T + U dependent steps, ~0.26 us each: 0.30 ms at the BASELINE shape).
//   pass 3  rnnt_grad_kernel   one wave per cell again: re-reads the V logits,
//           writes grad = g_b * (exp(logit + alpha + beta + cost - denom) with
//           the blank / label corrections), zero in the padded region.
//           HBM-bound: 4*V read + 4*V written per cell.
//
// Algorithmic traffic: 3 * 4 * V bytes per valid cell (+ 4*V per padded cell).
#include "row_stream.hpp"
#include "wr_common.hpp"

namespace wr {
namespace {

// ------------------------------------------------------------------ pass 1 --
// (row streaming helpers: row_stream.hpp)
template <typename T, bool NT, int UN>
__global__ __launch_bounds__(256) void rnnt_lse_kernel(
    const T *__restrict__ logits, const int32_t *__restrict__ targets,
    const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens,
    int B, int Tmax, int U1max, int V, int blank, int K, int S,
    float2 *__restrict__ lp_skew, float *__restrict__ denom, const int32_t *__restrict__ run_if)
{
    if (run_if != nullptr && *run_if == 0) return;     // repair pass behind the joiner's fused epilogue: usually nothing to do
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    const long nrows = (long)B * Tmax * U1max;
    const long stride = (long)gridDim.x * waves_per_block;
    const int cells = Tmax * U1max;
    for (long r = (long)blockIdx.x * waves_per_block + wid; r < nrows; r += stride) {
        const int b = (int)(r / cells);
        const int c = (int)(r - (long)b * cells);
        const int t = c / U1max;
        const int u = c - t * U1max;
        const int T_ = llens[b], U = tlens[b];
        if (t >= T_ || u > U) continue;
        const T *row = logits + (size_t)r * V;
        const float d = wave_row_lse<T, NT, UN>(row, V, lane);
        if (lane == 0) {
            const float xb = (float)row[blank];
            float em = 0.f;
            if (u < U) {
                const int lab = targets[(size_t)b * (U1max - 1) + u];
                em = (float)row[lab] - d;
            }
            denom[r] = d;
            const int s = t + u;
            lp_skew[((size_t)b * S + s) * U1max + u] = make_float2(xb - d, em);
        }
    }
}

// ------------------------------------------------------------------ pass 2 --
// One wave per (utterance, direction).  K label columns per lane.
//
// Precision: alpha/beta reach magnitudes ~ (T+U)*log(V) (1e4 at the BASELINE
// shape), where an fp32 ulp is 1e-3.  The lattice state is therefore carried in
// fp64 (adds, max) while the transcendental part log1p(exp(-|a-b|)) in [0, ln 2]
// is evaluated in fp32: absolute error ~1e-7 per step instead of ~5e-4.
__device__ __forceinline__ double log_add_exp_d(double a, double b)
{
    const double m = fmax(a, b);
    const float d = (float)(-fabs(a - b));          // NaN when both are -inf
    const float r = kLn2 * fast_log2(1.0f + fast_exp2(d * kLog2e));
    return (m == (double)kNegInf) ? (double)kNegInf : m + (double)r;
}

// Rotate a double by one lane: lane l receives lane l-1 (lane 0 receives lane 63).  gfx9 DPP wave_ror:1.
__device__ __forceinline__ double lane_rotate_up_d(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x13C, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x13C, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Lane l receives lane l+1 (lane 63 receives lane 0).  DPP wave_rol:1.
__device__ __forceinline__ double lane_rotate_down_d(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x134, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x134, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// One workgroup per (utterance, direction), one lane per label column u (NW = ceil(U1max/64) waves).
// At step s every lane works on the cell (t = s - u, u) of anti-diagonal s.  What a cell needs from
// diagonal s -/+ 1 is the lane's own previous value and its neighbouring lane's previous value: inside a
// wave that is one DPP wave rotate; across the wave boundary (lane 63 -> lane 0 of the next wave) it goes
// through an 8-byte LDS slot, double buffered, with one s_barrier per step.  The loop body is
// straight-line: loads are unconditional (clamped rows, values masked afterwards) and idle lanes store
// to a sink, so PF rows of log-probs stay in flight under counted s_waitcnt.
template <int PF>
__global__ __launch_bounds__(kRnntMaxCols) void rnnt_sweep_kernel(
    const float2 *__restrict__ lp_skew, const int32_t *__restrict__ llens,
    const int32_t *__restrict__ tlens, int Tmax, int U1max, int S,
    double *__restrict__ alpha_skew, double *__restrict__ beta_skew,
    double *__restrict__ ll_out, double *__restrict__ cost_ws, float *__restrict__ costs_out,
    double *__restrict__ dump /* [2*B*kRnntMaxCols] scratch that absorbs the stores of idle lanes */)
{
    constexpr double NEG = (double)kNegInf;
    __shared__ double xch[2][kRnntMaxCols / kWave];
    const int b = blockIdx.x;
    const bool backward = blockIdx.y != 0;
    const int u = threadIdx.x;
    const int lane = u & (kWave - 1), wave = u >> 6;
    const int nw = blockDim.x >> 6;
    int T = llens[b], U = tlens[b];
    T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
    U = U < 0 ? 0 : (U > U1max - 1 ? U1max - 1 : U);
    const int nsteps = (T > 0) ? T + U : 0;       // anti-diagonals that hold a valid cell

    if (nsteps == 0) {
        if (u == 0) {
            if (backward) { cost_ws[b] = 0.0; costs_out[b] = 0.f; } else ll_out[b] = 0.0;
        }
        return;
    }
    const bool in_row = u < U1max;
    const int col = in_row ? u : U1max - 1;
    const float2 *__restrict__ lp = lp_skew + (size_t)b * S * U1max + col;
    double *__restrict__ out = (backward ? beta_skew : alpha_skew) + (size_t)b * S * U1max + col;
    double *__restrict__ sink = dump + ((size_t)b * 2 + (backward ? 1 : 0)) * kRnntMaxCols + u;

    auto load_row = [&](int s) -> float2 {
        const int sc = s < 0 ? 0 : (s >= nsteps ? nsteps - 1 : s);
        return lp[(size_t)sc * U1max];
    };

    float2 ring[PF];
    double st = NEG;      // alpha(t-1, u) / beta(t+1, u): this lane's value on the previous diagonal
    float skp = 0.f;      // forward only: skip(t-1, u)
    double send = NEG;    // what the neighbouring lane needs from this lane's previous diagonal
    double result = 0.0;

    if (!backward) {
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = load_row(i);
        for (int base = 0; base < nsteps; base += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int s = base + i;           // steps s >= nsteps have no active lane
                const float2 cur = ring[i];
                ring[i] = load_row(s + PF);
                // alpha(t, u-1) + emit(t, u-1) from the lane to the left
                double recv = lane_rotate_up_d(send);
                if (lane == 0) recv = (wave == 0) ? ((s == 0) ? 0.0 : NEG) : xch[(s + 1) & 1][wave - 1];
                const int t = s - u;
                const bool active = (t >= 0) & (t < T) & (u <= U);
                const float sk = active ? cur.x : 0.f;
                const float em = active ? cur.y : 0.f;
                const double top = (t >= 1) ? st + (double)skp : NEG;
                double v = log_add_exp_d(top, recv);     // the origin sees recv = 0, top = -inf  ->  0
                v = active ? v : NEG;
                double *dst = active ? out + (size_t)s * U1max : sink;
                *dst = v;
                result = (active && t == T - 1 && u == U) ? v + (double)sk : result;
                send = v + (double)em;
                if (lane == kWave - 1) xch[s & 1][wave] = send;
                st = v;
                skp = sk;
                if (nw > 1) __syncthreads();
            }
        }
        if (u == U) ll_out[b] = result;          // exactly one lane saw the terminal cell (T-1, U)
    } else {
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = load_row(nsteps - 1 - i);
        for (int base = 0; base < nsteps; base += PF) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int s = nsteps - 1 - (base + i);   // steps s < 0 have no active lane
                const float2 cur = ring[i];
                ring[i] = load_row(s - PF);
                // beta(t, u+1) from the lane to the right
                double recv = lane_rotate_down_d(send);
                if (lane == kWave - 1) recv = (wave == nw - 1) ? NEG : xch[(s + 1) & 1][wave + 1];
                const int t = s - u;
                const bool active = (t >= 0) & (t < T) & (u <= U);
                const float sk = active ? cur.x : 0.f;
                const float em = active ? cur.y : 0.f;
                const double down = (t < T - 1) ? st + (double)sk : NEG;
                const double right = (u < U) ? recv + (double)em : NEG;
                double v = log_add_exp_d(down, right);
                v = (t == T - 1 && u == U) ? (double)sk : v;
                v = active ? v : NEG;
                double *dst = active ? out + (size_t)s * U1max : sink;
                *dst = v;
                result = (active && t == 0 && u == 0) ? v : result;
                send = v;
                if (lane == 0) xch[s & 1][wave] = send;
                st = v;
                if (nw > 1) __syncthreads();
            }
        }
        if (u == 0) { cost_ws[b] = -result; costs_out[b] = (float)(-result); }
    }
}

// ------------------------------------------------------------------ pass 3 --
template <typename T, bool NT /* loads */, bool NTS /* stores */, int UN>
__global__ __launch_bounds__(256) void rnnt_grad_kernel(
    const T *logits, const int32_t *__restrict__ targets,
    const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens,
    int B, int Tmax, int U1max, int V, int blank, float clamp, int K, int S,
    const double *__restrict__ alpha_skew, const double *__restrict__ beta_skew,
    const float *__restrict__ denom, const double *__restrict__ cost_ws,
    const float *__restrict__ grad_costs, T *grads)
{
    typedef typename VecOf<T>::type vec_t;
    constexpr int N = VecOf<T>::N;
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    const long nrows = (long)B * Tmax * U1max;
    const long stride = (long)gridDim.x * waves_per_block;
    const int cells = Tmax * U1max;
    for (long r = (long)blockIdx.x * waves_per_block + wid; r < nrows; r += stride) {
        const int b = (int)(r / cells);
        const int c = (int)(r - (long)b * cells);
        const int t = c / U1max;
        const int u = c - t * U1max;
        const int T_ = llens[b], U = tlens[b];
        const T *row = logits + (size_t)r * V;
        T *grow = grads + (size_t)r * V;

        const RowSplit<T> sp(row, V);
        const int h = sp.h, nv = sp.nv, tail = sp.tail;
        const vec_t *body = reinterpret_cast<const vec_t *>(row + h);
        vec_t *gbody = reinterpret_cast<vec_t *>(grow + h);

        if (t >= T_ || u > U) {          // padded cell: gradient is exactly zero
            vec_t z;
#pragma unroll
            for (int q = 0; q < N; ++q) z[q] = (T)0.f;
            if (lane < h) grow[lane] = (T)0.f;
            if (lane < tail) grow[h + N * nv + lane] = (T)0.f;
            for (int i = lane; i < nv; i += kWave) stv<NTS>(z, gbody + i);
            continue;
        }

        const size_t dbase = (size_t)b * S * U1max;
        const int s = t + u;
        // Lattice state is fp64; the combinations below are small in magnitude
        // (log-occupancies), so they are formed in fp64 and only then rounded.
        const double al = alpha_skew[dbase + (size_t)s * U1max + u];
        const double be = beta_skew[dbase + (size_t)s * U1max + u];
        const double cost = cost_ws[b];
        const float go = grad_costs ? grad_costs[b] : 1.f;
        const double cmd = al + cost - (double)denom[r];      // g = logit + cm
        const float c2 = (float)(cmd + be) * kLog2e;

        // special entries (SURVEY.md App. A.1 case chain; first match wins)
        bool blank_special = false;
        float blank_sub = 0.f;                      // exponent (natural log) of the subtracted term, minus logit
        if (t == T_ - 1 && u == U) { blank_special = true; blank_sub = (float)cmd; }
        else if (t < T_ - 1) {
            blank_special = true;
            blank_sub = (float)(cmd + beta_skew[dbase + (size_t)(s + 1) * U1max + u]);
        }
        int lab = -1;
        float lab_sub = 0.f;
        if (u < U) {
            lab = targets[(size_t)b * (U1max - 1) + u];
            if (lab == blank && blank_special) lab = -1;
            else lab_sub = (float)(cmd + beta_skew[dbase + (size_t)(s + 1) * U1max + (u + 1)]);
        }
        const int blk = blank_special ? blank : -1;

        auto fix = [&](float val, float x, int v) -> float {
            // val = exp(x + cm + be) already; subtract the case-chain term.
            if (v == blk) val -= fast_exp2((x + blank_sub) * kLog2e);
            else if (v == lab) val -= fast_exp2((x + lab_sub) * kLog2e);
            return val;
        };
        auto finish = [&](float val) -> float {
            if (clamp > 0.f) val = fminf(fmaxf(val, -clamp), clamp);
            return val * go;
        };

        if (lane < h) {
            const float x = (float)row[lane];
            grow[lane] = (T)finish(fix(fast_exp2(fmaf(x, kLog2e, c2)), x, lane));
        }
        if (lane < tail) {
            const int v = h + N * nv + lane;
            const float x = (float)row[v];
            grow[v] = (T)finish(fix(fast_exp2(fmaf(x, kLog2e, c2)), x, v));
        }
        // body vector i covers elements v = h + N*i .. h + N*i + N-1
        const int iblk = (blk >= h) ? ((blk - h) / N) : -1;
        const int ilab = (lab >= h) ? ((lab - h) / N) : -1;
        auto dov = [&](int i, const vec_t x) {
            float g[N];
#pragma unroll
            for (int q = 0; q < N; ++q) g[q] = fast_exp2(fmaf((float)x[q], kLog2e, c2));
            if (i == iblk || i == ilab) {
                const int v0 = h + N * i;
#pragma unroll
                for (int q = 0; q < N; ++q) g[q] = fix(g[q], (float)x[q], v0 + q);
            }
            vec_t o;
#pragma unroll
            for (int q = 0; q < N; ++q) o[q] = (T)finish(g[q]);
            stv<NTS>(o, gbody + i);
        };
        int i = lane;
        for (; i + (UN - 1) * kWave < nv; i += UN * kWave) {
            vec_t x[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) x[q] = ldv<NT>(body + i + q * kWave);
#pragma unroll
            for (int q = 0; q < UN; ++q) dov(i + q * kWave, x[q]);
        }
        if (i < nv) {                                    // remainder (fewer than UN vectors per lane): one guarded batch
            vec_t x[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q)
                if (i + q * kWave < nv) x[q] = ldv<NT>(body + i + q * kWave);
#pragma unroll
            for (int q = 0; q < UN; ++q)
                if (i + q * kWave < nv) dov(i + q * kWave, x[q]);
        }
    }
}

// ----------------------------------------------------- diagnostics export --
__global__ void rnnt_export_kernel(const double *__restrict__ alpha_skew, const double *__restrict__ beta_skew,
                                   const int32_t *__restrict__ llens, const int32_t *__restrict__ tlens,
                                   int B, int Tmax, int U1max, int K, int S,
                                   float *__restrict__ alpha, float *__restrict__ beta)
{
    const long n = (long)B * Tmax * U1max;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long)gridDim.x * blockDim.x) {
        const int cells = Tmax * U1max;
        const int b = (int)(r / cells);
        const int c = (int)(r - (long)b * cells);
        const int t = c / U1max, u = c - t * U1max;
        float a = 0.f, be = 0.f;
        if (t < llens[b] && u <= tlens[b]) {
            const size_t k = ((size_t)b * S + t + u) * U1max + u;
            a = (float)alpha_skew[k];
            be = (float)beta_skew[k];
        }
        alpha[r] = a;
        beta[r] = be;
    }
}

int check_shape(int B, int Tmax, int U1max, int V, int blank)
{
    WR_REQUIRE(B > 0 && Tmax > 0 && U1max > 0 && V > 0, WR_EINVAL,
               "rnnt: B, Tmax, U1max, V must be positive (got %d,%d,%d,%d)", B, Tmax, U1max, V);
    WR_REQUIRE(blank >= 0 && blank < V, WR_EINVAL, "rnnt: blank %d out of range [0,%d)", blank, V);
    WR_REQUIRE(U1max <= kRnntMaxCols, WR_EUNSUPPORTED,
               "rnnt: U1max=%d exceeds the sweep kernel's limit of %d label columns", U1max, kRnntMaxCols);
    WR_REQUIRE((long)B * Tmax * U1max < (1L << 31), WR_EUNSUPPORTED, "rnnt: more than 2^31 lattice cells");
    return WR_OK;
}

// Grid of a streaming pass: 4-wave workgroups, each wave visits rows r, r + G, r + 2G, ... (G = waves of the grid).
// Round 1 ran persistent grids (12 workgroups per CU, ~390 rows per wave at the BASELINE shape).  Measured in round 2
// (tools/tune_rnnt.py SWEEP=grid, sustained fwd + bwd sequence): many short-lived workgroups are faster -- the hardware
// dispatches them in index order as slots free up, so what is in flight at any moment is a tight window of adjacent rows
// (a few short-lived windows for k rows per wave) instead of 12 288 persistent waves that drift apart:
//     read-only row pass:         15.71 ms at 12 per CU -> 14.64 ms with one row per wave (monotonic in between);
//     read + write gradient pass: 34.49 ms at 16 per CU -> 33.65 (128) -> 33.35 (512) -> 33.18 (1 024) -> 32.96 ms at
//                                 1 400 per CU (3.4 rows per wave); one or two rows per wave lose again (2 048 per CU =
//                                 2.3 rows per wave: +2 %).
// `knob` > 0 forces that many workgroups per CU (the round-1 meaning of tuning keys 0 and 1); 0 = automatic: every wave
// gets about `bytes_per_wave` of logits.
int stream_grid(long nrows, int knob, size_t row_bytes, size_t bytes_per_wave)
{
    long blocks = (nrows + 3) / 4;
    if (knob > 0) {
        const long cap = 256L * knob;
        if (blocks > cap) blocks = cap;
    } else {
        const long all = blocks;                                  // one row per wave
        blocks = (long)(((double)nrows * (double)row_bytes) / (4.0 * (double)bytes_per_wave)) + 1;
        if (blocks > all) blocks = all;
        const long floor_blocks = 256L * 12 < (nrows + 3) / 4 ? 256L * 12 : (nrows + 3) / 4;   // never fewer than a persistent grid
        if (blocks < floor_blocks) blocks = floor_blocks;
    }
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

constexpr size_t kLseBytesPerWave = 16 * 1024;     // one row of 5 000 fp32 logits
constexpr size_t kGradBytesPerWave = 68 * 1024;    // 3.4 such rows on average

}  // namespace

void rnnt_launch_sweep(const RnntWs &w, char *ws, const int32_t *llens, const int32_t *tlens, int B, int Tmax,
                       int U1max, float *costs, hipStream_t st)
{
    hipLaunchKernelGGL((rnnt_sweep_kernel<8>), dim3(B, 2), dim3(64 * w.K), 0, st,
                       reinterpret_cast<const float2 *>(ws + w.lp_off), llens, tlens, Tmax, U1max, w.S,
                       reinterpret_cast<double *>(ws + w.alpha_off), reinterpret_cast<double *>(ws + w.beta_off),
                       reinterpret_cast<double *>(ws + w.ll_off), reinterpret_cast<double *>(ws + w.cost_off), costs,
                       reinterpret_cast<double *>(ws + w.dump_off));
}

}  // namespace wr

using namespace wr;

extern "C" size_t wr_rnnt_workspace_bytes(int B, int Tmax, int U1max)
{
    if (B <= 0 || Tmax <= 0 || U1max <= 0) return 0;
    return rnnt_ws_layout(B, Tmax, U1max).total;
}

extern "C" int wr_rnnt_loss_fwd(const void *logits_d, int dtype, const int32_t *targets_d,
                                const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int Tmax,
                                int U1max, int V, int blank, float *costs_d, void *workspace_d,
                                size_t workspace_bytes, void *stream)
{
    if (int rc = check_shape(B, Tmax, U1max, V, blank)) return rc;
    WR_REQUIRE(logits_d && logit_lengths_d && target_lengths_d && costs_d && workspace_d, WR_EINVAL,
               "rnnt_loss_fwd: null pointer argument");
    WR_REQUIRE(targets_d || U1max == 1, WR_EINVAL, "rnnt_loss_fwd: targets is null");
    WR_REQUIRE(dtype == WR_F32 || dtype == WR_F16 || dtype == WR_BF16, WR_EINVAL, "rnnt_loss_fwd: unknown dtype %d", dtype);
    const RnntWs w = rnnt_ws_layout(B, Tmax, U1max);
    WR_REQUIRE(workspace_bytes >= w.total, WR_EWORKSPACE, "rnnt_loss_fwd: workspace %zu < required %zu",
               workspace_bytes, w.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_d);
    const long nrows = (long)B * Tmax * U1max;
    const size_t row_bytes = (size_t)V * (dtype == WR_F32 ? 4 : 2);
    const dim3 grid1(stream_grid(nrows, tune_get(kTuneLseBlocksPerCu), row_bytes, kLseBytesPerWave));
#define WR_LAUNCH_LSE(T, NT)                                                                                        \
    if (tune_get(kTuneLseUnroll) >= 16) WR_LAUNCH_LSE_U(T, NT, 16); else if (tune_get(kTuneLseUnroll) >= 8) WR_LAUNCH_LSE_U(T, NT, 8); else WR_LAUNCH_LSE_U(T, NT, 4)
#define WR_LAUNCH_LSE_U(T, NT, UN)                                                                                  \
    hipLaunchKernelGGL((rnnt_lse_kernel<T, NT, UN>), grid1, dim3(256), 0, st, static_cast<const T *>(logits_d), targets_d, \
                       logit_lengths_d, target_lengths_d, B, Tmax, U1max, V, blank, w.K, w.S,                        \
                       reinterpret_cast<float2 *>(ws + w.lp_off), reinterpret_cast<float *>(ws + w.denom_off), nullptr)
    const bool nt = (tune_get(kTuneNonTemporal) & 4) != 0;
    if (dtype == WR_F32) { if (nt) { WR_LAUNCH_LSE(float, true); } else { WR_LAUNCH_LSE(float, false); } }
    else if (dtype == WR_F16) { if (nt) { WR_LAUNCH_LSE(_Float16, true); } else { WR_LAUNCH_LSE(_Float16, false); } }
    else { if (nt) { WR_LAUNCH_LSE(__bf16, true); } else { WR_LAUNCH_LSE(__bf16, false); } }
#undef WR_LAUNCH_LSE
#undef WR_LAUNCH_LSE_U
    WR_CHECK_LAUNCH("rnnt_lse_kernel");
    rnnt_launch_sweep(w, ws, logit_lengths_d, target_lengths_d, B, Tmax, U1max, costs_d, st);
    WR_CHECK_LAUNCH("rnnt_sweep_kernel");
    return WR_OK;
}

extern "C" int wr_rnnt_loss_fwd_from_lse(const float *logits_d, const int32_t *targets_d, const int32_t *logit_lengths_d,
                                         const int32_t *target_lengths_d, int B, int Tmax, int U1max, int V, int blank,
                                         float *costs_d, void *workspace_d, size_t workspace_bytes, void *stream)
{
    if (int rc = check_shape(B, Tmax, U1max, V, blank)) return rc;
    WR_REQUIRE(logits_d && logit_lengths_d && target_lengths_d && costs_d && workspace_d, WR_EINVAL,
               "rnnt_loss_fwd_from_lse: null pointer argument");
    WR_REQUIRE(targets_d || U1max == 1, WR_EINVAL, "rnnt_loss_fwd_from_lse: targets is null");
    const RnntWs w = rnnt_ws_layout(B, Tmax, U1max);
    WR_REQUIRE(workspace_bytes >= w.total, WR_EWORKSPACE, "rnnt_loss_fwd_from_lse: workspace %zu < required %zu",
               workspace_bytes, w.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace_d);
    // repair pass: the stand-alone row statistics, executed only if the joiner's epilogue raised the flag (a partial
    // sum overflowed: more than 88 nats of spread inside one row); otherwise every workgroup leaves at once
    const long nrows = (long)B * Tmax * U1max;
    hipLaunchKernelGGL((rnnt_lse_kernel<float, false, 8>), dim3(stream_grid(nrows, tune_get(kTuneLseBlocksPerCu), (size_t)V * 4, kLseBytesPerWave)), dim3(256), 0, st,
                       logits_d, targets_d, logit_lengths_d, target_lengths_d, B, Tmax, U1max, V, blank, w.K, w.S,
                       reinterpret_cast<float2 *>(ws + w.lp_off), reinterpret_cast<float *>(ws + w.denom_off),
                       reinterpret_cast<const int32_t *>(ws + w.flag_off));
    WR_CHECK_LAUNCH("rnnt_lse_kernel (repair)");
    rnnt_launch_sweep(w, ws, logit_lengths_d, target_lengths_d, B, Tmax, U1max, costs_d, st);
    WR_CHECK_LAUNCH("rnnt_sweep_kernel");
    return WR_OK;
}

extern "C" int wr_rnnt_loss_bwd(const void *logits_d, int dtype, const int32_t *targets_d,
                                const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B, int Tmax,
                                int U1max, int V, int blank, float clamp, const float *grad_costs_d, void *grads_d,
                                const void *workspace_d, size_t workspace_bytes, void *stream)
{
    if (int rc = check_shape(B, Tmax, U1max, V, blank)) return rc;
    WR_REQUIRE(logits_d && logit_lengths_d && target_lengths_d && grads_d && workspace_d, WR_EINVAL,
               "rnnt_loss_bwd: null pointer argument");
    WR_REQUIRE(targets_d || U1max == 1, WR_EINVAL, "rnnt_loss_bwd: targets is null");
    WR_REQUIRE(dtype == WR_F32 || dtype == WR_F16 || dtype == WR_BF16, WR_EINVAL, "rnnt_loss_bwd: unknown dtype %d", dtype);
    const RnntWs w = rnnt_ws_layout(B, Tmax, U1max);
    WR_REQUIRE(workspace_bytes >= w.total, WR_EWORKSPACE, "rnnt_loss_bwd: workspace %zu < required %zu",
               workspace_bytes, w.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *ws = static_cast<const char *>(workspace_d);
    const long nrows = (long)B * Tmax * U1max;
    const size_t row_bytes = (size_t)V * (dtype == WR_F32 ? 4 : 2);
    const dim3 grid3(stream_grid(nrows, tune_get(kTuneGradBlocksPerCu), row_bytes, kGradBytesPerWave));
#define WR_LAUNCH_GRAD_U(T, NT, UN)                                                                                 \
    if (nts) WR_LAUNCH_GRAD_US(T, NT, true, UN); else WR_LAUNCH_GRAD_US(T, NT, false, UN)
#define WR_LAUNCH_GRAD_US(T, NT, NTS, UN)                                                                           \
    hipLaunchKernelGGL((rnnt_grad_kernel<T, NT, NTS, UN>), grid3, dim3(256), 0, st, static_cast<const T *>(logits_d), targets_d, \
                       logit_lengths_d, target_lengths_d, B, Tmax, U1max, V, blank, clamp, w.K, w.S,                 \
                       reinterpret_cast<const double *>(ws + w.alpha_off),                                           \
                       reinterpret_cast<const double *>(ws + w.beta_off),                                            \
                       reinterpret_cast<const float *>(ws + w.denom_off),                                            \
                       reinterpret_cast<const double *>(ws + w.cost_off), grad_costs_d, static_cast<T *>(grads_d))
#define WR_LAUNCH_GRAD(T, NT)                                             \
    do {                                                                   \
        if (tune_get(kTuneGradUnroll) >= 16) { WR_LAUNCH_GRAD_U(T, NT, 16); } \
        else if (tune_get(kTuneGradUnroll) >= 8) { WR_LAUNCH_GRAD_U(T, NT, 8); } \
        else { WR_LAUNCH_GRAD_U(T, NT, 4); }                               \
    } while (0)
    const bool nt = (tune_get(kTuneNonTemporal) & 1) != 0;
    const bool nts = (tune_get(kTuneNonTemporal) & 2) != 0;
    if (dtype == WR_F32) { if (nt) WR_LAUNCH_GRAD(float, true); else WR_LAUNCH_GRAD(float, false); }
    else if (dtype == WR_F16) { if (nt) WR_LAUNCH_GRAD(_Float16, true); else WR_LAUNCH_GRAD(_Float16, false); }
    else { if (nt) WR_LAUNCH_GRAD(__bf16, true); else WR_LAUNCH_GRAD(__bf16, false); }
#undef WR_LAUNCH_GRAD
#undef WR_LAUNCH_GRAD_U
#undef WR_LAUNCH_GRAD_US
    WR_CHECK_LAUNCH("rnnt_grad_kernel");
    return WR_OK;
}

extern "C" int wr_rnnt_export_lattice(const void *workspace_d, size_t workspace_bytes,
                                      const int32_t *logit_lengths_d, const int32_t *target_lengths_d, int B,
                                      int Tmax, int U1max, float *alpha_d, float *beta_d, void *stream)
{
    if (int rc = check_shape(B, Tmax, U1max, 1, 0)) return rc;
    WR_REQUIRE(workspace_d && alpha_d && beta_d && logit_lengths_d && target_lengths_d, WR_EINVAL,
               "rnnt_export_lattice: null pointer argument");
    const RnntWs w = rnnt_ws_layout(B, Tmax, U1max);
    WR_REQUIRE(workspace_bytes >= w.total, WR_EWORKSPACE, "rnnt_export_lattice: workspace too small");
    const char *ws = static_cast<const char *>(workspace_d);
    hipLaunchKernelGGL(rnnt_export_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const double *>(ws + w.alpha_off),
                       reinterpret_cast<const double *>(ws + w.beta_off), logit_lengths_d, target_lengths_d, B, Tmax,
                       U1max, w.K, w.S, alpha_d, beta_d);
    WR_CHECK_LAUNCH("rnnt_export_kernel");
    return WR_OK;
}
