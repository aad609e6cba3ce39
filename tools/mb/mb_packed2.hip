// Micro-benchmark (development tool): packed skinny LSTM kernel with explicit software pipelining
// (loads of k-block group g+1 are issued before the MFMAs of group g).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int NB, int TRIP>
struct Regs { f32x4 w[TRIP]; f32x4 x[TRIP][NB]; };

template <int NB, int KW, int TRIP>
__device__ __forceinline__ void load_group(Regs<NB, TRIP>& r, const f32x4* wp, const f32x4* xp, int kb, int KB) {
#pragma unroll
    for (int t = 0; t < TRIP; ++t) {
        int k = kb + t * KW; if (k >= KB) k = KB - 1;   // clamped duplicates are neutralised by the caller (zero weight)
        r.w[t] = wp[(size_t)k * 64];
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) r.x[t][bt] = xp[((size_t)bt * KB + k) * 64];
    }
}
template <int NB, int TRIP>
__device__ __forceinline__ void mma_group(const Regs<NB, TRIP>& r, f32x4 (&acc)[NB]) {
#pragma unroll
    for (int t = 0; t < TRIP; ++t)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
                acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.w[t][cc], r.x[t][bt][cc], acc[bt], 0, 0, 0);
}

template <int NB, int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void k_pipe(const f32x4* __restrict__ W, const f32x4* __restrict__ X,
                                                  float* __restrict__ out, int H, int KB) {
    __shared__ f32x4 red[KW * NB * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[NB];
    for (int bt = 0; bt < NB; ++bt) acc[bt] = f32x4{0, 0, 0, 0};
    const f32x4* wp = W + ((size_t)blockIdx.x * KB) * 64 + lane;
    const f32x4* xp = X + lane;
    const int step = TRIP * KW;
    Regs<NB, TRIP> ra, rb;
    int kb = wave;
    if (kb < KB) load_group<NB, KW, TRIP>(ra, wp, xp, kb, KB);
    while (kb < KB) {
        const int kn = kb + step;
        if (kn < KB) load_group<NB, KW, TRIP>(rb, wp, xp, kn, KB);
        mma_group<NB, TRIP>(ra, acc);
        kb = kn;
        if (kb >= KB) break;
        const int kn2 = kb + step;
        if (kn2 < KB) load_group<NB, KW, TRIP>(ra, wp, xp, kn2, KB);
        mma_group<NB, TRIP>(rb, acc);
        kb = kn2;
    }
    for (int bt = 0; bt < NB; ++bt) red[(wave * NB + bt) * 64 + lane] = acc[bt];
    __syncthreads();
    if (tid >= NB * 64) return;
    const int bt = tid >> 6;
    f32x4 s = red[bt * 64 + lane];
    for (int w = 1; w < KW; ++w) { f32x4 t = red[(w * NB + bt) * 64 + lane]; s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3]; }
    const int b = bt * 16 + (lane & 15), u = blockIdx.x * 4 + (lane >> 4);
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
#define RUN(NB, KW, TRIP) { \
    float t = time_us([&] { for (int j = 0; j < 2; ++j) hipLaunchKernelGGL((k_pipe<NB, KW, TRIP>), dim3(H / 4), dim3(KW * 64), 0, 0, (const f32x4*)W[j], (const f32x4*)X[j], out, H, Ks[j] / 16); }, 100) / 2; \
    printf("pipe NB=%d KW=%2d trip=%d: %6.2f us/launch  %.0f GB/s (weights)\n", NB, KW, TRIP, t, 0.5 * 16.0 * H * (Ks[0] + Ks[1]) / t / 1e3); }
    RUN(2, 8, 1) RUN(2, 8, 2) RUN(2, 8, 4) RUN(2, 16, 1) RUN(2, 16, 2) RUN(2, 16, 4) RUN(2, 4, 2) RUN(2, 4, 4) RUN(2, 4, 8)
    RUN(1, 8, 2) RUN(1, 16, 2) RUN(4, 8, 1) RUN(4, 8, 2) RUN(4, 16, 1) RUN(4, 16, 2)
    // graph-like measurement: the two launches inside a captured graph, replayed
    {
        hipStream_t st; CK(hipStreamCreate(&st));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int it = 0; it < 50; ++it)
            for (int j = 0; j < 2; ++j)
                hipLaunchKernelGGL((k_pipe<2, 16, 2>), dim3(H / 4), dim3(16 * 64), 0, st, (const f32x4*)W[j], (const f32x4*)X[j], out, H, Ks[j] / 16);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph replay of 100 launches (NB=2 KW=16 trip=2): %.2f us/launch\n", ms * 1e3 / 500);
    }
    return 0;
}
