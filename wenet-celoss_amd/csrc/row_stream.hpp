// Row streaming helpers shared by the RNN-T and CTC loss kernels: a row of V logits is read by one wave in 16-byte
// vectors, several in flight per lane, and reduced to its log-sum-exp with a per-lane online (max, sum).
#pragma once
#include "wr_common.hpp"

namespace wr {

// Element types: float (the parity bar), _Float16 and __bf16 (AMP logits; arithmetic stays fp32).
// A row is streamed in 16-byte vectors of VecOf<T>::N elements.
template <typename T> struct VecOf;
template <> struct VecOf<float>    { static constexpr int N = 4; typedef float    type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<_Float16> { static constexpr int N = 8; typedef _Float16 type __attribute__((ext_vector_type(8))); };
template <> struct VecOf<__bf16>   { static constexpr int N = 8; typedef __bf16   type __attribute__((ext_vector_type(8))); };

template <bool NT, typename V>
__device__ __forceinline__ V ldv(const V *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename V>
__device__ __forceinline__ void stv(V v, V *p)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// Per-lane online (max, sum) in the log2 domain over one row of V logits.
struct RowStat {
    float m;   // running max of x*log2e
    float s;   // running sum of 2^(x*log2e - m)
};

template <typename T>
__device__ __forceinline__ void stat_addv(RowStat &st, const typename VecOf<T>::type v)
{
    constexpr int N = VecOf<T>::N;
    float y[N];
    float cm = -3.0e38f;
#pragma unroll
    for (int i = 0; i < N; ++i) { y[i] = (float)v[i] * kLog2e; cm = fmaxf(cm, y[i]); }
    const float nm = fmaxf(st.m, cm);
    float acc = st.s * fast_exp2(st.m - nm);
#pragma unroll
    for (int i = 0; i < N; ++i) acc += fast_exp2(y[i] - nm);
    st.s = acc;
    st.m = nm;
}

__device__ __forceinline__ void stat_add1(RowStat &st, const float x)
{
    const float y = x * kLog2e;
    const float nm = fmaxf(st.m, y);
    st.s = st.s * fast_exp2(st.m - nm) + fast_exp2(y - nm);
    st.m = nm;
}

// Split row[0..V) into a scalar head, a 16-byte aligned vector body and a scalar tail (any V, any base).
template <typename T>
struct RowSplit {
    int h, nv, tail;
    __device__ __forceinline__ RowSplit(const T *row, int V)
    {
        constexpr int N = VecOf<T>::N;
        const int mis = (int)((reinterpret_cast<uintptr_t>(row) / sizeof(T)) & (N - 1));
        const int head = (N - mis) & (N - 1);
        h = head < V ? head : V;
        nv = (V - h) / N;
        tail = V - h - nv * N;
    }
};

// Natural-log log-sum-exp of row[0..V) computed by one wave.
template <typename T, bool NT, int UN>
__device__ __forceinline__ float wave_row_lse(const T *__restrict__ row, int V, int lane)
{
    typedef typename VecOf<T>::type vec_t;
    constexpr int N = VecOf<T>::N;
    RowStat st{-3.0e38f, 0.f};
    const RowSplit<T> sp(row, V);
    if (lane < sp.h) stat_add1(st, (float)row[lane]);
    if (lane < sp.tail) stat_add1(st, (float)row[sp.h + N * sp.nv + lane]);
    const vec_t *__restrict__ body = reinterpret_cast<const vec_t *>(row + sp.h);
    int i = lane;
    for (; i + (UN - 1) * kWave < sp.nv; i += UN * kWave) {
        vec_t x[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) x[q] = ldv<NT>(body + i + q * kWave);
#pragma unroll
        for (int q = 0; q < UN; ++q) stat_addv<T>(st, x[q]);
    }
    if (i < sp.nv) {                                        // remainder (fewer than UN vectors per lane): ONE guarded batch, so a
        vec_t x[UN];                                        // short row (16-bit logits: 625 vectors at V = 5000) has all its
#pragma unroll                                              // loads in flight at once instead of 8 + a straggler
        for (int q = 0; q < UN; ++q)
            if (i + q * kWave < sp.nv) x[q] = ldv<NT>(body + i + q * kWave);
#pragma unroll
        for (int q = 0; q < UN; ++q)
            if (i + q * kWave < sp.nv) stat_addv<T>(st, x[q]);
    }
    const float M = wave_max(st.m);
    const float s = wave_sum(st.s * fast_exp2(st.m - M));
    return (M + fast_log2(s)) * kLn2;
}

}  // namespace wr
