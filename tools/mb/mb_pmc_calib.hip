// Micro-benchmark (development tool): what FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for KNOWN byte counts in the access
// patterns of the attention launches of the decode step -- the guide calibrates FETCH_SIZE x 2 only for wide 16-byte-per-lane streaming
// reads.  Every kernel touches each byte of its buffer exactly once per launch; the buffers are far larger than the L2s and the
// Infinity Cache is flushed by a 1 GiB streaming write between the measured launches.
//   k_wide16   grid-stride 16 B / lane streaming read (the calibrated pattern)
//   k_row4     per-workgroup (L x A) tile, 4 B / lane, 256 contiguous bytes per wave load      (S / pm tile of pk_attnfin, at_kernel)
//   k_row16    per-workgroup (L x E) tile, 16 B / lane, 1 KiB contiguous per wave load          (encoder-memory tile)
//   k_row4s    the (L x A) tile with only HALF of each 128-byte line used per workgroup (two workgroups share the lines)
//   k_poll     one lane per wave re-reads ONE 8-byte word with an agent-scope atomic load, n times (the granule polls)
//   k_write16 / k_write4   16 B / 4 B per lane streaming stores
// hipcc --offload-arch=gfx950 -O3 -o mb_pmc_calib mb_pmc_calib.hip ;  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./mb_pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_wide16(const f32x4* p, size_t n4, float* out) {
    f32x4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const f32x4 v = p[i]; a += v; }
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) out[blockIdx.x] = 1.0f;
}
__global__ __launch_bounds__(512) void k_row4(const float* p, int L, int A, float* out) {
    const float* t = p + (size_t)blockIdx.x * L * A;
    const int a0 = threadIdx.x % A, grp = threadIdx.x / A, ngrp = 512 / A;
    float a = 0;
    for (int l = grp; l < L; l += ngrp) a += t[(size_t)l * A + a0];
    if (a == 12345.678f) out[blockIdx.x] = 1.0f;
}
__global__ __launch_bounds__(512) void k_row4s(const float* p, int L, int A, float* out) {       // workgroup pair shares the tile: each reads 16 floats of every 32
    const float* t = p + (size_t)(blockIdx.x >> 1) * L * A;
    const int half = blockIdx.x & 1;
    const int a0 = (threadIdx.x % (A / 2)), grp = threadIdx.x / (A / 2), ngrp = 512 / (A / 2);
    const int col = (a0 / 16) * 32 + half * 16 + (a0 % 16);
    float a = 0;
    for (int l = grp; l < L; l += ngrp) a += t[(size_t)l * A + col];
    if (a == 12345.678f) out[blockIdx.x] = 1.0f;
}
// two workgroups read the SAME (L x A) tile at the same time: `same` != 0 -> the partners are nb workgroups apart in launch order (nb a multiple
// of 8: the same XCD under round-robin placement), else adjacent (different XCDs)
__global__ __launch_bounds__(512) void k_pair(const float* p, int L, int A, int nb, int same, float* out) {
    const int tile = same ? (int)blockIdx.x % nb : (int)blockIdx.x >> 1;
    const float* t = p + (size_t)tile * L * A;
    const int a0 = threadIdx.x % A, grp = threadIdx.x / A, ngrp = 512 / A;
    float a = 0;
    for (int l = grp; l < L; l += ngrp) a += t[(size_t)l * A + a0];
    if (a == 12345.678f) out[blockIdx.x] = 1.0f;
}
__global__ __launch_bounds__(512) void k_row16(const float* p, int L, int E, float* out) {
    const float* t = p + (size_t)blockIdx.x * L * E;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 a = {0, 0, 0, 0};
    for (int l = wave; l < L; l += 8)
        for (int e = lane * 4; e < E; e += 256) a += *reinterpret_cast<const f32x4*>(t + (size_t)l * E + e);
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) out[blockIdx.x] = 1.0f;
}
__global__ __launch_bounds__(512) void k_poll(unsigned long long* w, int n, float* out) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    unsigned long long s = 0;
    if ((threadIdx.x & 63) == 0) {
        gu64* cp = (gu64*)(w + blockIdx.x * 8 + (threadIdx.x >> 6));
        for (int i = 0; i < n; ++i) s += __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (s == 12345ull) out[blockIdx.x] = 1.0f;
}
__global__ __launch_bounds__(256) void k_write16(f32x4* p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = f32x4{1, 2, 3, 4};
}
__global__ __launch_bounds__(256) void k_write4(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1.0f;
}

int main() {
    const int L = 43, A = 256, E = 512, NB = 2048;                 // 2048 "utterances": 90 MB of (L, A) tiles, 180 MB of (L, E) tiles
    const size_t nA = (size_t)NB * L * A, nE = (size_t)NB * L * E, nW = (size_t)64 << 20;      // floats
    float *pa, *pe, *pw, *out, *flush;
    unsigned long long* words;
    CK(hipMalloc(&pa, nA * 4)); CK(hipMalloc(&pe, nE * 4)); CK(hipMalloc(&pw, nW * 4)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&flush, (size_t)1 << 30));
    CK(hipMalloc(&words, 256 * 8 * 8));
    CK(hipMemset(pa, 0, nA * 4)); CK(hipMemset(pe, 0, nE * 4)); CK(hipMemset(pw, 0, nW * 4)); CK(hipMemset(words, 0, 256 * 8 * 8));
    auto fl = [&] { hipLaunchKernelGGL(k_write16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<f32x4*>(flush), ((size_t)1 << 30) / 16); };
    const int npoll = 20000;
    printf("known bytes per launch: k_wide16 %zu  k_row4 %zu  k_row4s %zu  k_row16 %zu  k_poll %zu (requested)  k_write16 %zu  k_write4 %zu\n",
           nW * 4, nA * 4, nA * 4, nE * 4, (size_t)256 * 8 * npoll * 8, nW * 4, nW * 4);
    for (int rep = 0; rep < 3; ++rep) {
        fl(); hipLaunchKernelGGL(k_wide16, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const f32x4*>(pw), nW / 4, out);
        fl(); hipLaunchKernelGGL(k_row4, dim3(NB), dim3(512), 0, 0, pa, L, A, out);
        fl(); hipLaunchKernelGGL(k_row4s, dim3(2 * NB), dim3(512), 0, 0, pa, L, A, out);
        fl(); hipLaunchKernelGGL(k_row16, dim3(NB), dim3(512), 0, 0, pe, L, E, out);
        fl(); hipLaunchKernelGGL(k_pair, dim3(256), dim3(512), 0, 0, pa, L, A, 128, 1, out);      // 128 tiles, partners on one XCD
        fl(); hipLaunchKernelGGL(k_pair, dim3(256), dim3(512), 0, 0, pa + (size_t)256 * L * A, L, A, 128, 0, out);      // ... on two XCDs
        fl(); hipLaunchKernelGGL(k_poll, dim3(256), dim3(512), 0, 0, words, npoll, out);
        fl(); hipLaunchKernelGGL(k_write16, dim3(2048), dim3(256), 0, 0, reinterpret_cast<f32x4*>(pw), nW / 4);
        fl(); hipLaunchKernelGGL(k_write4, dim3(2048), dim3(256), 0, 0, pw, nW);
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
