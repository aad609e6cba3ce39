// Micro-benchmark (development tool): per-CU ingest rate of an L2-resident activation block
// that EVERY workgroup reads in full (the X operand of the skinny kernels), by access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// PAT 0: lane (i = l&15, kq = l>>4): row i, 64 B per row per instruction (MFMA 16x16x4 operand pattern)
// PAT 1: lane (i = l>>3, j = l&7): 8 rows x 128 B per instruction
// PAT 2: fully contiguous: 1 KiB of one row per instruction
// every wave reads its K-slice of all ROWS rows; UNR loads in flight per lane
template <int PAT, int KW, int UNR>
__global__ __launch_bounds__(KW * 64) void k_load(const float* __restrict__ X, float* out, int ROWS, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 a = {0, 0, 0, 0};
    const int kper = K / KW;               // K-slice of this wave
    const int kbeg = wave * kper;
    if (PAT == 0) {
        for (int r0 = 0; r0 < ROWS; r0 += 16) {
            const float* p = X + (size_t)(r0 + (lane & 15)) * K + kbeg + (lane >> 4) * 4;
            for (int k = 0; k < kper; k += 16 * UNR) {
                f32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) v[u] = *(const f32x4*)(p + k + 16 * u);
#pragma unroll
                for (int u = 0; u < UNR; ++u) { a[0] += v[u][0]; a[1] += v[u][1]; a[2] += v[u][2]; a[3] += v[u][3]; }
            }
        }
    } else if (PAT == 1) {
        for (int r0 = 0; r0 < ROWS; r0 += 8) {
            const float* p = X + (size_t)(r0 + (lane >> 3)) * K + kbeg + (lane & 7) * 4;
            for (int k = 0; k < kper; k += 32 * UNR) {
                f32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) v[u] = *(const f32x4*)(p + k + 32 * u);
#pragma unroll
                for (int u = 0; u < UNR; ++u) { a[0] += v[u][0]; a[1] += v[u][1]; a[2] += v[u][2]; a[3] += v[u][3]; }
            }
        }
    } else if (PAT == 3) {   // 16 rows x 64 B per instruction, but the 4 lanes of a row are ADJACENT lanes
        for (int r0 = 0; r0 < ROWS; r0 += 16) {
            const float* p = X + (size_t)(r0 + (lane >> 2)) * K + kbeg + (lane & 3) * 4;
            for (int k = 0; k < kper; k += 16 * UNR) {
                f32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) v[u] = *(const f32x4*)(p + k + 16 * u);
#pragma unroll
                for (int u = 0; u < UNR; ++u) { a[0] += v[u][0]; a[1] += v[u][1]; a[2] += v[u][2]; a[3] += v[u][3]; }
            }
        }
    } else {
        for (int r = 0; r < ROWS; ++r) {
            const float* p = X + (size_t)r * K + kbeg + lane * 4;
            for (int k = 0; k < kper; k += 256 * UNR) {
                f32x4 v[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) { int kk = k + 256 * u; v[u] = kk + lane * 4 < kper ? *(const f32x4*)(p + kk) : f32x4{0, 0, 0, 0}; }
#pragma unroll
                for (int u = 0; u < UNR; ++u) { a[0] += v[u][0]; a[1] += v[u][1]; a[2] += v[u][2]; a[3] += v[u][3]; }
            }
        }
    }
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) out[0] = 1;
}

template <typename F>
float time_us(F f, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    const int ROWS = 32, K = 2048;
    float *X, *out;
    std::vector<float> h((size_t)ROWS * K, 1.0f);
    CK(hipMalloc(&X, h.size() * 4)); CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 4));
    const double bytes_per_wg = (double)ROWS * K * 4;
#define RUN(PAT, KW, UNR, GRID) { \
    float t = time_us([&] { hipLaunchKernelGGL((k_load<PAT, KW, UNR>), dim3(GRID), dim3(KW * 64), 0, 0, X, out, ROWS, K); }, 200); \
    printf("pat=%d KW=%2d unr=%d grid=%4d: %6.2f us  per-WG %.1f B/clk@2.4GHz  aggregate %.1f TB/s\n", PAT, KW, UNR, GRID, t, \
           bytes_per_wg / (t - 2.3) / 2400.0, bytes_per_wg * GRID / t / 1e6); }
    RUN(3, 8, 4, 256) RUN(3, 8, 8, 256) RUN(3, 8, 2, 256) RUN(3, 16, 4, 256) RUN(3, 8, 4, 1)
    RUN(0, 8, 4, 256) RUN(0, 8, 8, 256) RUN(1, 8, 4, 256) RUN(1, 8, 8, 256) RUN(2, 8, 1, 256) RUN(2, 8, 2, 256)
    RUN(0, 16, 4, 256) RUN(1, 16, 4, 256) RUN(0, 4, 8, 256) RUN(1, 4, 8, 256)
    RUN(0, 8, 4, 512) RUN(1, 8, 4, 512) RUN(0, 8, 4, 128) RUN(1, 8, 4, 128) RUN(0, 8, 4, 64) RUN(0, 8, 4, 1)
    RUN(1, 8, 4, 1) RUN(0, 16, 8, 1) RUN(0, 16, 8, 128)
    return 0;
}
