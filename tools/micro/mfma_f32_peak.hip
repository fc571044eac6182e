// Micro-benchmark: what does this device sustain on back-to-back fp32 MFMAs (no memory traffic), per instruction shape?
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_peak.hip -o /tmp/mfma && /tmp/mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

// KIND 0: v_mfma_f32_32x32x2_f32 (16 regs, 4096 flop... per wave 2*32*32*2), 1: 16x16x4 (4 regs), 2: 16x16x1 four blocks (16 regs),
// 3: 32x32x1 two blocks (32 regs), 4: 4x4x1 sixteen blocks (4 regs)
template <int KIND> struct Acc;
template <> struct Acc<0> { typedef f32x16 t; static constexpr double flop = 2.0 * 32 * 32 * 2; };
template <> struct Acc<1> { typedef f32x4 t;  static constexpr double flop = 2.0 * 16 * 16 * 4; };
template <> struct Acc<2> { typedef f32x16 t; static constexpr double flop = 2.0 * 16 * 16 * 1 * 4; };
template <> struct Acc<3> { typedef f32x32 t; static constexpr double flop = 2.0 * 32 * 32 * 1 * 2; };
template <> struct Acc<4> { typedef f32x4 t;  static constexpr double flop = 2.0 * 4 * 4 * 1 * 16; };

template <int KIND> __device__ __forceinline__ typename Acc<KIND>::t mm(float a, float b, typename Acc<KIND>::t c);
template <> __device__ __forceinline__ f32x16 mm<0>(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
template <> __device__ __forceinline__ f32x4 mm<1>(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <> __device__ __forceinline__ f32x16 mm<2>(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0); }
template <> __device__ __forceinline__ f32x32 mm<3>(float a, float b, f32x32 c) { return __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, c, 0, 0, 0); }
template <> __device__ __forceinline__ f32x4 mm<4>(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed)
{
    typedef typename Acc<KIND>::t acc_t;
    acc_t acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < (int)(sizeof(acc_t) / 4); ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f - threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = mm<KIND>(a, b, acc[i]);
        a += 1e-7f;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < (int)(sizeof(acc_t) / 4); ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NACC>
void run(int waves_per_simd, const char *name)
{
    float *d; hipMalloc(&d, 1 << 24);
    const int blocks = 256 * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<KIND, NACC><<<blocks, 256>>>(d, 1000, 1.f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop<KIND, NACC><<<blocks, 256>>>(d, iters, 1.f + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * NACC * Acc<KIND>::flop;
    printf("{\"what\": \"%s\", \"acc\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"frac_of_157.3\": %.3f}\n", name, NACC,
           waves_per_simd, best, flops / best / 1e9, flops / best / 1e9 / 157.3);
    hipFree(d);
}

int main()
{
    run<0, 2>(1, "32x32x2"); run<0, 2>(2, "32x32x2"); run<0, 4>(1, "32x32x2"); run<0, 8>(1, "32x32x2"); run<0, 16>(1, "32x32x2"); run<0, 4>(2, "32x32x2");
    run<1, 8>(1, "16x16x4"); run<1, 16>(1, "16x16x4"); run<1, 32>(1, "16x16x4"); run<1, 16>(2, "16x16x4");
    run<2, 8>(1, "16x16x1 x4 blocks"); run<2, 16>(1, "16x16x1 x4 blocks");
    run<3, 4>(1, "32x32x1 x2 blocks"); run<3, 8>(1, "32x32x1 x2 blocks");
    run<4, 16>(1, "4x4x1 x16 blocks");
    return 0;
}
