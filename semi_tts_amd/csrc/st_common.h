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

// compute units of the current device (cached; 256 on MI355X)
static inline int st_device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 1;
    }
    return n;
}

static __host__ __device__ inline bool st_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------- device helpers (wave = 64)
typedef __attribute__((ext_vector_type(4))) float f32x4;

// row chunks of the split column reductions (st_bn_stats, st_colsum, st_bn_bwd): enough blocks to fill 256 CUs
// even for 80-column tensors, chunks of >= 64 rows
// row chunks of the column reductions (BatchNorm statistics / backward sums, bias gradients): ~32 rows per chunk, at most 128 chunks
// (two column blocks x 128 chunks = one workgroup per compute unit for the 128-channel layers of the CBHG)
static inline __host__ __device__ int st_colreduce_chunks(int M) { int c = M / 32; if (c < 1) c = 1; if (c > 128) c = 128; return c; }

#define ST_WAVE 64

__device__ __forceinline__ float st_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// Hardware-transcendental forms for the per-step LSTM epilogues (one v_exp_f32 + one v_rcp_f32 each, ~1e-7 absolute error;
// the libm forms cost 5-10x the instructions on a path where every instruction is latency)
__device__ __forceinline__ float st_sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float st_tanh_fast(float x) {
    const float t = __builtin_amdgcn_exp2f(fabsf(x) * -2.885390081777927f);      // exp(-2|x|): (1 - t) is exact-ish near 0
    return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), x);
}

// Highway combine y = H * T + x * (1 - T) in ONE spelled-out operation order (the compiler is free to contract `a*b + c*d` either
// way; every kernel that forms it -- GEMM epilogue, fused stack, training forward -- must round alike)
__device__ __forceinline__ float st_highway(float h, float t, float x) { return fmaf(h, t, x * (1.0f - t)); }

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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: it also waits
// for every global load still in flight, i.e. it exposes the full latency of anything prefetched into registers before it.
// Use this one where the threads only exchange data through LDS (never to publish global-memory writes to other threads).
__device__ __forceinline__ void st_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DPP forms of the wave reductions (all 64 lanes must be active): the permutations ride on the VALU instruction itself, where
// __shfl_xor is a ds_bpermute_b32 through the LDS crossbar (~100 cycles of latency per step in a dependent chain)
constexpr int ST_DPP_QUAD_XOR1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int ST_DPP_QUAD_XOR2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int ST_DPP_ROW_HALF_MIRROR = 0x141;
constexpr int ST_DPP_ROW_MIRROR = 0x140;
constexpr int ST_DPP_ROW_ROR8 = 0x128;       // lane i of a row of 16 reads lane (i + 8) % 16 = i ^ 8
template <int CTRL>
__device__ __forceinline__ float st_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float st_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// sum over the 8 lanes sharing lane >> 3 (every one of them gets it)
__device__ __forceinline__ float st_oct_sum_dpp(float v) {
    v += st_dpp<ST_DPP_QUAD_XOR1>(v);
    v += st_dpp<ST_DPP_QUAD_XOR2>(v);
    v += st_dpp<ST_DPP_ROW_HALF_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float st_wave_sum_dpp(float v) {
    v = st_oct_sum_dpp(v);
    v += st_dpp<ST_DPP_ROW_MIRROR>(v);
    return (st_lane(v, 0) + st_lane(v, 16)) + (st_lane(v, 32) + st_lane(v, 48));
}
__device__ __forceinline__ float st_wave_max_dpp(float v) {
    v = fmaxf(v, st_dpp<ST_DPP_QUAD_XOR1>(v));
    v = fmaxf(v, st_dpp<ST_DPP_QUAD_XOR2>(v));
    v = fmaxf(v, st_dpp<ST_DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, st_dpp<ST_DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(st_lane(v, 0), st_lane(v, 16)), fmaxf(st_lane(v, 32), st_lane(v, 48)));
}

// 16-byte load of 4 consecutive floats from an address that is only 4-byte aligned (rows of 1025 floats): gfx950 under HSA serves
// dword-aligned global_load_dwordx4 (the compiler emits exactly that for this copy)
__device__ __forceinline__ f32x4 st_ld4_u(const float* p) { f32x4 v; __builtin_memcpy(&v, p, 16); return v; }
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
