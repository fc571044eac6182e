// Micro-benchmark: do VALU / LDS instructions of the SAME wave issue in the shadow of its fp32 MFMAs (1 wave per SIMD)?
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_mix.hip -o /tmp/mix && /tmp/mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL>
__global__ __launch_bounds__(256) void mix(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x16){0};
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f - threadIdx.x * 2e-3f;
    int x[8];
    for (int q = 0; q < 8; ++q) x[q] = threadIdx.x + q;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    f32x4 l = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) x[(i + v) & 7] = x[(i + v) & 7] * 3 + (x[(i + v + 1) & 7] ^ it);   // 2-3 VALU ops each
            if (NL > 0 && i % (16 / NL) == 0) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(lds + ((x[0] & 1023) & ~3));
                l += t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        a += 1e-7f;
    }
    float s = l[0] + l[1] + l[2] + l[3];
    for (int q = 0; q < 8; ++q) s += x[q];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NL>
void run()
{
    float *d; hipMalloc(&d, 1 << 24);
    const int blocks = 256, iters = 5000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<NV, NL><<<blocks, 256>>>(d, 500, 1.f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mix<NV, NL><<<blocks, 256>>>(d, iters, 1.f + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("{\"valu_groups_per_mfma\": %d, \"lds_reads_per_16_mfma\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"frac\": %.3f}\n", NV, NL, best,
           flops / best / 1e9, flops / best / 1e9 / 157.3);
    hipFree(d);
}

int main()
{
    run<0, 0>(); run<1, 0>(); run<2, 0>(); run<4, 0>(); run<6, 0>(); run<0, 2>(); run<0, 4>(); run<2, 2>(); run<2, 4>();
    return 0;
}
