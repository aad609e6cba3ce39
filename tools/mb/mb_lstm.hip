// Micro-benchmark (development tool, not part of the product): where does the time of the
// weight-streaming LSTM gate kernel go?  Ablation variants of the skinny kernel on one
// (4H x K) weight matrix, B = 32.   hipcc --offload-arch=gfx950 -O3 -o mb_lstm mb_lstm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// flags: bit0 ROT (rotate chunk order per tile), bit1 NOX, bit2 NOW, bit3 NOMFMA
template <int NB, int KW, int FLAGS, int TRIP>
__global__ __launch_bounds__(KW * 64) void k_lstm(const float* __restrict__ W, const float* __restrict__ X,
                                                  float* __restrict__ out, int H, int K, int B) {
    __shared__ f32x4 red[KW * NB * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool PERM = (FLAGS & 16) != 0;
    // load roles: PERM -> the 4 lanes of a row are adjacent lanes (row = lane>>2, piece = lane&3)
    const int i = PERM ? (lane >> 2) : (lane & 15), kq = PERM ? (lane & 3) : (lane >> 4), tile = blockIdx.x;
    const int paddr = (4 * (lane & 15) + (lane >> 4)) * 4;   // source lane of MFMA lane (i=l&15,kq=l>>4)
    const int wrow = (i & 3) * H + tile * 4 + (i >> 2);
    const float* wp = W + (size_t)wrow * K;
    const float* xp[NB];
    for (int bt = 0; bt < NB; ++bt) xp[bt] = X + (size_t)(bt * 16 + i) * K;
    f32x4 acc[NB];
    for (int bt = 0; bt < NB; ++bt) acc[bt] = f32x4{0, 0, 0, 0};
    const int nchunk = K / 32;
    const int rot = (FLAGS & 1) ? (tile * 7) % nchunk : 0;
    for (int c = wave; c < nchunk; c += TRIP * KW) {
        f32x4 w[TRIP][2], x[TRIP][2][NB];
#pragma unroll
        for (int t = 0; t < TRIP; ++t) {
            int cc = c + t * KW;
            if (cc >= nchunk) cc = nchunk - 1;
            cc = cc + rot; if (cc >= nchunk) cc -= nchunk;
            const int k0 = cc * 32 + kq * 4;
            if (!(FLAGS & 4)) { w[t][0] = *(const f32x4*)(wp + k0); w[t][1] = *(const f32x4*)(wp + k0 + 16); }
            else { w[t][0] = f32x4{1, 2, 3, 4}; w[t][1] = f32x4{1, 2, 3, 4}; }
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                if (!(FLAGS & 2)) { x[t][0][bt] = *(const f32x4*)(xp[bt] + k0); x[t][1][bt] = *(const f32x4*)(xp[bt] + k0 + 16); }
                else { x[t][0][bt] = f32x4{1, 1, 1, 1}; x[t][1][bt] = f32x4{1, 1, 1, 1}; }
            }
        }
        if (PERM) {
#pragma unroll
            for (int t = 0; t < TRIP; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        w[t][h][cc] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(paddr, __builtin_bit_cast(int, w[t][h][cc])));
#pragma unroll
                        for (int bt = 0; bt < NB; ++bt)
                            x[t][h][bt][cc] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(paddr, __builtin_bit_cast(int, x[t][h][bt][cc])));
                    }
        }
#pragma unroll
        for (int t = 0; t < TRIP; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt) {
                        if (FLAGS & 8) acc[bt][cc] += w[t][h][cc] * x[t][h][bt][cc];
                        else acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][h][cc], x[t][h][bt][cc], acc[bt], 0, 0, 0);
                    }
    }
    for (int bt = 0; bt < NB; ++bt) red[(wave * NB + bt) * 64 + lane] = acc[bt];
    __syncthreads();
    if (tid >= NB * 64) return;
    const int bt = tid >> 6;
    f32x4 s = red[bt * 64 + lane];
    for (int w = 1; w < KW; ++w) { f32x4 t = red[(w * NB + bt) * 64 + lane]; s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3]; }
    const int b = bt * 16 + (lane & 15), u = tile * 4 + (lane >> 4);
    if (b < B) out[(size_t)b * H + u] = tanhf(s[0]) + s[1] * s[2] + s[3];
}

// plain streaming read of the weights (sum reduction) -- what the memory system gives a simple kernel
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ W, size_t n4, float* out) {
    f32x4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = W[i]; a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) out[0] = 1;
}
__global__ void k_empty(float* out) { if (out == nullptr) out[0] = 0; }

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
    const int H = 1024, B = 32;
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
    printf("empty kernel: %.2f us\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, out); }, 500));
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        float t = time_us([&] { for (int j = 0; j < 2; ++j) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const f32x4*)W[j], (size_t)H * Ks[j], out); }, 100) / 2;
        printf("stream read grid=%4d: %.2f us/launch  %.0f GB/s\n", grid, t, 0.5 * 16.0 * H * (Ks[0] + Ks[1]) / t / 1e3);
    }
#define RUN(NB, KW, FLAGS, TRIP) { \
    float t = time_us([&] { for (int j = 0; j < 2; ++j) hipLaunchKernelGGL((k_lstm<NB, KW, FLAGS, TRIP>), dim3(H / 4), dim3(KW * 64), 0, 0, W[j], X[j], out, H, Ks[j], B); }, 100) / 2; \
    printf("lstm NB=%d KW=%2d flags=%2d trip=%d: %.2f us/launch  %.0f GB/s (weights)\n", NB, KW, FLAGS, TRIP, t, 0.5 * 16.0 * H * (Ks[0] + Ks[1]) / t / 1e3); }
    RUN(2, 8, 16, 2) RUN(2, 8, 16, 1) RUN(2, 8, 16, 4) RUN(2, 16, 16, 2) RUN(2, 16, 16, 1) RUN(2, 4, 16, 2) RUN(2, 4, 16, 4) RUN(1, 8, 16, 2) RUN(4, 8, 16, 2) RUN(4, 8, 16, 1) RUN(4, 16, 16, 1) RUN(2, 8, 18, 2) RUN(2, 8, 20, 2) RUN(2, 8, 24, 2)
    RUN(2, 8, 0, 2) RUN(2, 8, 1, 2) RUN(2, 8, 2, 2) RUN(2, 8, 4, 2) RUN(2, 8, 6, 2) RUN(2, 8, 8, 2) RUN(2, 8, 10, 2)
    RUN(2, 16, 0, 2) RUN(2, 16, 1, 2) RUN(2, 4, 0, 2) RUN(2, 4, 0, 4) RUN(2, 8, 0, 4) RUN(2, 8, 1, 4) RUN(2, 16, 0, 1) RUN(2, 8, 0, 1)
    RUN(1, 8, 0, 2) RUN(4, 8, 0, 2) RUN(1, 8, 2, 2) RUN(2, 16, 2, 2) RUN(2, 16, 2, 4)
    return 0;
}
