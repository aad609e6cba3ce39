// st_common.h -- shared helpers for the gfx950 kernels of libsemitts_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/semitts.h"

// ---------------------------------------------------------------- host-side error plumbing
void st_set_error(const char* fmt, ...);

#define ST_CHECK_ARG(cond, ...)                     \
    do {                                            \
        if (!(cond)) {                              \
            st_set_error(__VA_ARGS__);              \
            return -22; /* EINVAL */                \
        }                                           \
    } while (0)

#define ST_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            st_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -5; /* EIO */                                                       \
        }                                                                              \
    } while (0)

#define ST_LAUNCH_CHECK()                                                              \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            st_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return -5;                                                                 \
        }                                                                              \
    } while (0)

static __host__ __device__ inline bool st_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------- device helpers (wave = 64)
typedef __attribute__((ext_vector_type(4))) float f32x4;

// row chunks of the split column reductions (st_bn_stats, st_colsum, st_bn_bwd): enough blocks to fill 256 CUs
// even for 80-column tensors, chunks of >= 64 rows
static inline int st_colreduce_chunks(int M) { int c = M / 64; if (c < 1) c = 1; if (c > 64) c = 64; return c; }

#define ST_WAVE 64

__device__ __forceinline__ float st_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float st_act(float v, int act) {
    switch (act) {
        case ST_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case ST_ACT_TANH: return tanhf(v);
        case ST_ACT_SIGMOID: return st_sigmoid(v);
        default: return v;
    }
}

__device__ __forceinline__ float st_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float st_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// 16-byte global load of 4 consecutive floats (caller guarantees alignment)
__device__ __forceinline__ f32x4 st_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// guarded variant: elements with index >= n read as 0 (p may be unaligned)
__device__ __forceinline__ f32x4 st_ld4_guard(const float* p, int n) {
    f32x4 v;
    v[0] = n > 0 ? p[0] : 0.0f;
    v[1] = n > 1 ? p[1] : 0.0f;
    v[2] = n > 2 ? p[2] : 0.0f;
    v[3] = n > 3 ? p[3] : 0.0f;
    return v;
}
