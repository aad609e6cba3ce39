// Micro-benchmark (development tool): skinny LSTM gate kernel on PRE-PACKED operands.
// W packed as [tile][kb][lane][4] (1 KiB per 16-row x 16-k block, MFMA A-operand lane order),
// X tiled as [batch tile][kb][lane][4].  Every wave instruction reads 1 KiB contiguous.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// RT = row tiles (of 16 rows) per workgroup; NB = batch tiles; KW waves split the k-blocks; TRIP k-blocks in flight
template <int RT, int NB, int KW, int TRIP, int FLAGS>
__global__ __launch_bounds__(KW * 64) void k_packed(const f32x4* __restrict__ W, const f32x4* __restrict__ X,
                                                    float* __restrict__ out, int H, int KB) {
    __shared__ f32x4 red[KW * RT * NB * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[RT][NB];
    for (int r = 0; r < RT; ++r) for (int bt = 0; bt < NB; ++bt) acc[r][bt] = f32x4{0, 0, 0, 0};
    const f32x4* wp[RT];
    for (int r = 0; r < RT; ++r) wp[r] = W + ((size_t)(blockIdx.x * RT + r) * KB) * 64 + lane;
    const f32x4* xp = X + lane;
    for (int kb = wave; kb < KB; kb += TRIP * KW) {
        f32x4 w[TRIP][RT], x[TRIP][NB];
#pragma unroll
        for (int t = 0; t < TRIP; ++t) {
            int k = kb + t * KW; if (k >= KB) k = KB - 1;
#pragma unroll
            for (int r = 0; r < RT; ++r) w[t][r] = (FLAGS & 4) ? f32x4{1, 2, 3, 4} : wp[r][(size_t)k * 64];
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) x[t][bt] = (FLAGS & 2) ? f32x4{1, 1, 1, 1} : xp[((size_t)bt * KB + k) * 64];
        }
#pragma unroll
        for (int t = 0; t < TRIP; ++t)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt)
                        acc[r][bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][r][cc], x[t][bt][cc], acc[r][bt], 0, 0, 0);
    }
    for (int r = 0; r < RT; ++r) for (int bt = 0; bt < NB; ++bt) red[((wave * RT + r) * NB + bt) * 64 + lane] = acc[r][bt];
    __syncthreads();
    if (tid >= RT * NB * 64) return;
    const int q = tid >> 6;   // (r, bt)
    f32x4 s = red[q * 64 + lane];
    for (int w = 1; w < KW; ++w) { f32x4 t = red[(w * RT * NB + q) * 64 + lane]; s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3]; }
    const int r = q / NB, bt = q % NB;
    const int b = bt * 16 + (lane & 15), u = (blockIdx.x * RT + r) * 4 + (lane >> 4);
    out[(size_t)b * H + u] = tanhf(s[0]) + s[1] * s[2] + s[3];
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
    const int H = 1024;
    const int Ks[2] = {1792, 2560};
    float *W[2], *X[2], *out;
    for (int j = 0; j < 2; ++j) {
        size_t nw = (size_t)4 * H * Ks[j];
        std::vector<float> hw(nw);
        for (size_t i = 0; i < nw; ++i) hw[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
        CK(hipMalloc(&W[j], nw * 4)); CK(hipMemcpy(W[j], hw.data(), nw * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&X[j], (size_t)64 * Ks[j] * 4)); CK(hipMemcpy(X[j], hw.data(), (size_t)64 * Ks[j] * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&out, (size_t)64 * H * 4));
#define RUN(RT, NB, KW, TRIP, FLAGS) { \
    float t = time_us([&] { for (int j = 0; j < 2; ++j) hipLaunchKernelGGL((k_packed<RT, NB, KW, TRIP, FLAGS>), dim3(H / 4 / RT), dim3(KW * 64), 0, 0, (const f32x4*)W[j], (const f32x4*)X[j], out, H, Ks[j] / 16); }, 100) / 2; \
    printf("packed RT=%d NB=%d KW=%2d trip=%d flags=%d: %6.2f us/launch  %.0f GB/s (weights)\n", RT, NB, KW, TRIP, FLAGS, t, 0.5 * 16.0 * H * (Ks[0] + Ks[1]) / t / 1e3); }
    RUN(1, 2, 8, 2, 0) RUN(1, 2, 8, 4, 0) RUN(1, 2, 8, 8, 0) RUN(1, 2, 16, 2, 0) RUN(1, 2, 16, 4, 0) RUN(1, 2, 4, 4, 0) RUN(1, 2, 4, 8, 0)
    RUN(2, 2, 8, 2, 0) RUN(2, 2, 8, 4, 0) RUN(2, 2, 16, 2, 0) RUN(2, 2, 16, 4, 0)
    RUN(1, 2, 8, 4, 2) RUN(1, 2, 8, 4, 4) RUN(1, 2, 8, 4, 6)
    RUN(1, 1, 8, 4, 0) RUN(1, 4, 8, 2, 0) RUN(1, 4, 8, 4, 0) RUN(1, 4, 16, 2, 0) RUN(2, 4, 16, 2, 0) RUN(2, 4, 8, 2, 0)
    return 0;
}
