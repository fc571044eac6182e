// Micro-benchmark: what does this device sustain on back-to-back v_mfma_f32_32x32x2_f32 (no memory traffic)?
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_peak.hip -o /tmp/mfma && /tmp/mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x16){0};
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f - threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-7f;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd)
{
    float *d; hipMalloc(&d, 1 << 24);
    const int blocks = 256 * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, 256>>>(d, 1000, 1.f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop<NACC><<<blocks, 256>>>(d, iters, 1.f + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * NACC * 4096.0;
    printf("{\"what\": \"mfma_f32_32x32x2 peak\", \"acc\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", NACC,
           waves_per_simd, best, flops / best / 1e9);
    hipFree(d);
}

int main()
{
    run<4>(1); run<4>(2); run<1>(1); run<2>(2);
    return 0;
}
