// Where does the time of a tiny decode GEMM go?  Stand-alone probe: the lane GEMM's structure (K split over
// the waves of one workgroup per 32-column tile, register-staged fragments, LDS reduction) with pieces switched
// off, timed over back-to-back launches on one stream.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lane_gemm_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES, int MT, bool LOAD, bool MFMA, bool RED, int PAD = 32, bool ROWMAJOR = false>
__global__ __launch_bounds__(64 * WAVES) void probe(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                    int K, int lda, int ldb, int ldc)
{
    __shared__ float red[RED ? WAVES : 1][MT][32 * PAD];
    const int n0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    f32x16 acc[MT];
    for (int m = 0; m < MT; ++m) acc[m] = (f32x16){0};
    const int kq = K / WAVES;
    const float *a = A + (size_t)(wave * kq + half) * lda + l31;
    const float *b = B + (size_t)(wave * kq + half) * ldb + n0 + l31;
    constexpr int CH = 16;
    for (int k0 = 0; k0 < kq; k0 += 2 * CH) {
        float bv[CH], av[MT][CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k = k0 + 2 * i;
            const int kc = k < kq ? k : kq - 2;
            if (LOAD) {
                bv[i] = b[(size_t)kc * ldb] * (k < kq ? 1.f : 0.f);
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m][i] = a[(size_t)kc * lda + m * 32];
            } else {
                bv[i] = (float)(lane + i);
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m][i] = (float)(kc + m);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (MFMA) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][i], bv[i], acc[m], 0, 0, 0);
                else acc[m][i & 15] += av[m][i] * bv[i];
            }
    }
    if (RED) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave][m][((r & 3) + 8 * (r >> 2) + 4 * half) * PAD + l31] = acc[m][r];
        __syncthreads();
        for (int i = tid; i < MT * 1024; i += 64 * WAVES) {
            const int m = i >> 10, q = i & 1023;
            const int col = ROWMAJOR ? (q & 31) : (q >> 5), ln = ROWMAJOR ? (q >> 5) : (q & 31);
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += red[w][m][ln * PAD + col];
            if (ROWMAJOR) C[(size_t)(m * 32 + ln) * ldb + n0 + col] = t;
            else C[(size_t)(n0 + col) * ldc + m * 32 + ln] = t;
        }
    } else {
        float t = 0.f;
        for (int m = 0; m < MT; ++m)
            for (int r = 0; r < 16; ++r) t += acc[m][r];
        if (t == 12345.678f) C[tid] = t;
    }
}

__global__ void empty_kernel(float *C) { if (C == nullptr) C[0] = 0.f; }
template <int BYTES>
__global__ __launch_bounds__(512) void empty_lds_kernel(float *C)
{
    __shared__ float x[BYTES / 4];
    x[threadIdx.x] = 1.f;
    __syncthreads();
    if (x[(threadIdx.x + 1) & 511] == 2.f) C[0] = 0.f;
}

template <typename F>
static float time_us(F launch, int n = 2000)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / n;
}

int main()
{
    const int K = 512, N = 5120, NL = 64;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)K * NL * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)N * NL * 4);
    hipMemset(A, 0, (size_t)K * NL * 4); hipMemset(B, 0, (size_t)K * N * 4);
#define RUN(name, W, MT, L, M, R, k, cols)                                                                               \
    printf("%-44s K=%4d wgs=%4d  %7.2f us\n", name, k, (cols) / 32,                                                      \
           time_us([&] { hipLaunchKernelGGL((probe<W, MT, L, M, R>), dim3((cols) / 32), dim3(64 * W), 0, 0, A, B, C, k, NL, N, NL); }))
    printf("%-44s %7.2f us\n", "empty kernel, 1 wg x 64", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, C); }));
    printf("%-44s %7.2f us\n", "empty kernel, 160 wg x 512", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(160), dim3(512), 0, 0, C); }));
    printf("%-44s %7.2f us\n", "2KB-LDS kernel, 160 wg x 512", time_us([&] { hipLaunchKernelGGL(empty_lds_kernel<2048>, dim3(160), dim3(512), 0, 0, C); }));
    printf("%-44s %7.2f us\n", "64KB-LDS kernel, 160 wg x 512", time_us([&] { hipLaunchKernelGGL(empty_lds_kernel<65536>, dim3(160), dim3(512), 0, 0, C); }));
    for (int pass = 0; pass < 2; ++pass) {
        const int k = pass ? 256 : 512, cols = pass ? 256 : 5120;
        RUN("8 waves MT2 full", 8, 2, true, true, true, k, cols);
        printf("%-44s K=%4d wgs=%4d  %7.2f us\n", "8 waves MT2 full, pad 33 k-major", k, cols / 32, time_us([&] { hipLaunchKernelGGL((probe<8, 2, true, true, true, 33, false>), dim3(cols / 32), dim3(512), 0, 0, A, B, C, k, NL, N, NL); }));
        printf("%-44s K=%4d wgs=%4d  %7.2f us\n", "8 waves MT2 full, pad 32 row-major", k, cols / 32, time_us([&] { hipLaunchKernelGGL((probe<8, 2, true, true, true, 32, true>), dim3(cols / 32), dim3(512), 0, 0, A, B, C, k, NL, N, NL); }));
        printf("%-44s K=%4d wgs=%4d  %7.2f us\n", "4 waves MT2 full, pad 33 k-major", k, cols / 32, time_us([&] { hipLaunchKernelGGL((probe<4, 2, true, true, true, 33, false>), dim3(cols / 32), dim3(256), 0, 0, A, B, C, k, NL, N, NL); }));
        printf("%-44s K=%4d wgs=%4d  %7.2f us\n", "16 waves MT1 full, pad 33 k-major", k, cols / 32, time_us([&] { hipLaunchKernelGGL((probe<16, 1, true, true, true, 33, false>), dim3(cols / 32), dim3(1024), 0, 0, A, B, C, k, NL, N, NL); }));
        RUN("8 waves MT2 no global loads", 8, 2, false, true, true, k, cols);
        RUN("8 waves MT2 no mfma", 8, 2, true, false, true, k, cols);
        RUN("8 waves MT2 no LDS reduce", 8, 2, true, true, false, k, cols);
        RUN("8 waves MT2 nothing (regs only)", 8, 2, false, false, false, k, cols);
        RUN("4 waves MT2 full", 4, 2, true, true, true, k, cols);
        RUN("4 waves MT2 no LDS reduce", 4, 2, true, true, false, k, cols);
        RUN("16 waves MT1 full", 16, 1, true, true, true, k, cols);
        RUN("4 waves MT1 full", 4, 1, true, true, true, k, cols);
    }
    // same kernel, but the weight matrix rotates through 6 copies (60 MB > the 8 x 4 MB L2s): is the per-launch
    // cost a cold-L2 cost?
    {
        float *Bs[6];
        for (auto &p : Bs) { hipMalloc(&p, (size_t)K * N * 4); hipMemset(p, 0, (size_t)K * N * 4); }
        int r = 0;
        printf("%-44s K=%4d wgs=%4d  %7.2f us\n", "8 waves MT2 pad33, rotating B (6 x 10 MB)", 512, 160,
               time_us([&] { hipLaunchKernelGGL((probe<8, 2, true, true, true, 33, false>), dim3(160), dim3(512), 0, 0, A, Bs[r], C, 512, NL, N, NL); r = (r + 1) % 6; }));
        printf("%-44s K=%4d wgs=%4d  %7.2f us\n", "8 waves MT2 pad33, rotating B, K=256 8 wgs", 256, 8,
               time_us([&] { hipLaunchKernelGGL((probe<8, 2, true, true, true, 33, false>), dim3(8), dim3(512), 0, 0, A, Bs[r], C, 256, NL, N, NL); r = (r + 1) % 6; }));
        printf("%-44s K=%4d wgs=%4d  %7.2f us\n", "8 waves MT2 pad33, rotating B, K=512 32 wgs", 512, 32,
               time_us([&] { hipLaunchKernelGGL((probe<8, 2, true, true, true, 33, false>), dim3(32), dim3(512), 0, 0, A, Bs[r], C, 512, NL, N, NL); r = (r + 1) % 6; }));
        printf("%-44s K=%4d wgs=%4d  %7.2f us\n", "8 waves MT2 pad33, same B, K=512 32 wgs", 512, 32,
               time_us([&] { hipLaunchKernelGGL((probe<8, 2, true, true, true, 33, false>), dim3(32), dim3(512), 0, 0, A, B, C, 512, NL, N, NL); }));
    }
    return 0;
}
