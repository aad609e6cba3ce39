// gemm_lab.hip -- standalone experiment bench for the fp32 MFMA GEMM core (C = A W^T, A (M,K), W (N,K) row-major).
// Not part of the library: it exists to measure inner-loop structures on the GPU box before they go into csrc/gemm.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o tools/gemm_lab && tools/gemm_lab
// "direct" variants: every wave loads its own MFMA fragments straight from global memory (16-byte loads, lane (r, q) takes
// row r, floats 4q..4q+3 of a 16-float k-block -- the same k assignment for A and W, so the product is unchanged), keeps
// DEPTH k-blocks in flight in registers, and never touches LDS or a barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MT, int NT, int DEPTH, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void direct_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                                float* __restrict__ C, int ldc, int M, int N, int K) {
    const int kper = ((K / 16 + gridDim.z - 1) / gridDim.z) * 16;
    const int kz0 = blockIdx.z * kper;
    A += kz0; W += kz0; C += (size_t)blockIdx.z * M * ldc;
    K = max(0, min(kper, K - kz0));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = (blockIdx.x * WM + wm) * 16 * MT, n0 = (blockIdx.y * WN + wn) * 16 * NT;
    const int r = lane & 15, q = lane >> 4;
    const float* pa[MT];
    const float* pw[NT];
#pragma unroll
    for (int t = 0; t < MT; ++t) pa[t] = A + (size_t)min(m0 + 16 * t + r, M - 1) * lda + 4 * q;
#pragma unroll
    for (int t = 0; t < NT; ++t) pw[t] = W + (size_t)min(n0 + 16 * t + r, N - 1) * ldw + 4 * q;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[DEPTH][MT], rb[DEPTH][NT];
    const int nkb = K / 16;
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
        const int kb = min(s, nkb - 1);
#pragma unroll
        for (int t = 0; t < MT; ++t) ra[s][t] = *reinterpret_cast<const f32x4*>(pa[t] + kb * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) rb[s][t] = *reinterpret_cast<const f32x4*>(pw[t] + kb * 16);
    }
    for (int kb0 = 0; kb0 < nkb; kb0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            if (kb0 + s < nkb) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[s][i][c], rb[s][j][c], acc[i][j], 0, 0, 0);
            }
            const int kn = min(kb0 + s + DEPTH, nkb - 1);
#pragma unroll
            for (int t = 0; t < MT; ++t) ra[s][t] = *reinterpret_cast<const f32x4*>(pa[t] + kn * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) rb[s][t] = *reinterpret_cast<const f32x4*>(pw[t] + kn * 16);
        }
    }
    // C fragment: lane holds rows 4q..4q+3, column r of each 16x16 tile
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + 16 * j + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + 16 * i + 4 * q + e;
                if (m < M && n < N) C[(size_t)m * ldc + n] = acc[i][j][e];
            }
        }
}

// LDS-staged reference structure (what csrc/gemm.hip does, reduced to the plain GEMM): 64 x 64 tile, 4 waves, one 16-float
// k-block per barrier, register prefetch of the next block.
__global__ __launch_bounds__(256) void lds_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                  float* __restrict__ C, int ldc, int M, int N, int K) {
    const int kper = ((K / 16 + gridDim.z - 1) / gridDim.z) * 16;
    const int kz0 = blockIdx.z * kper;
    A += kz0; W += kz0; C += (size_t)blockIdx.z * M * ldc;
    K = max(0, min(kper, K - kz0));
    constexpr int LD = 24;
    __shared__ __attribute__((aligned(16))) float As[2][64 * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int srow = tid >> 2, skq = tid & 3;
    const float* pa = A + (size_t)min(m0 + srow, M - 1) * lda + skq * 4;
    const float* pw = W + (size_t)min(n0 + srow, N - 1) * ldw + skq * 4;
    const int r = lane & 15, fk = (lane >> 4) * 4;
    f32x4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = K / 16;
    f32x4 na = *reinterpret_cast<const f32x4*>(pa), nw = *reinterpret_cast<const f32x4*>(pw);
    *reinterpret_cast<f32x4*>(As[0] + srow * LD + skq * 4) = na;
    *reinterpret_cast<f32x4*>(Bs[0] + srow * LD + skq * 4) = nw;
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        const int kn = min(kb + 1, nkb - 1);
        na = *reinterpret_cast<const f32x4*>(pa + kn * 16); nw = *reinterpret_cast<const f32x4*>(pw + kn * 16);
        f32x4 a4[2], b4[2];
        for (int t = 0; t < 2; ++t) {
            a4[t] = *reinterpret_cast<const f32x4*>(As[buf] + (wm * 32 + t * 16 + r) * LD + fk);
            b4[t] = *reinterpret_cast<const f32x4*>(Bs[buf] + (wn * 32 + t * 16 + r) * LD + fk);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][c], b4[j][c], acc[i][j], 0, 0, 0);
        *reinterpret_cast<f32x4*>(As[buf ^ 1] + srow * LD + skq * 4) = na;
        *reinterpret_cast<f32x4*>(Bs[buf ^ 1] + srow * LD + skq * 4) = nw;
        __syncthreads();
    }
    const int q = lane >> 4;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + 16 * j + r;
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * 32 + 16 * i + 4 * q + e;
                if (m < M && n < N) C[(size_t)m * ldc + n] = acc[i][j][e];
            }
        }
}

__global__ void reduce_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n4, int S) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
    for (int z = 1; z < S; ++z) { const f32x4 v = reinterpret_cast<const f32x4*>(part)[i + (size_t)z * n4]; s += v; }
    reinterpret_cast<f32x4*>(out)[i] = s;
}

__global__ void naive_kernel(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(A[(size_t)m * lda + k], W[(size_t)n * ldw + k], s);
    C[(size_t)m * ldc + n] = s;
}

struct Shape { const char* name; int M, N, K; };

template <typename F>
static float time_us(F launch, int iters = 30) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1000.f / iters;
}

static float* dA; static float* dW; static float* dC; static float* dR; static float* dP;
static std::vector<float> hC, hR;

static double check(int M, int N) {
    CK(hipMemcpy(hC.data(), dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t i = 0; i < (size_t)M * N; i += 97) worst = fmax(worst, fabs((double)hC[i] - hR[i]) / (1.0 + fabs((double)hR[i])));
    return worst;
}

template <int MT, int NT, int DEPTH, int WM, int WN>
static void run_direct(const Shape& s, int S = 1) {
    dim3 grid((s.M + 16 * MT * WM - 1) / (16 * MT * WM), (s.N + 16 * NT * WN - 1) / (16 * NT * WN), S);
    CK(hipMemset(dC, 0, (size_t)s.M * s.N * 4));
    const size_t n4 = (size_t)s.M * s.N / 4;
    auto f = [&] {
        hipLaunchKernelGGL((direct_kernel<MT, NT, DEPTH, WM, WN>), grid, dim3(64 * WM * WN), 0, 0, dA, s.K, dW, s.K, S > 1 ? dP : dC, s.N, s.M, s.N, s.K);
        if (S > 1) hipLaunchKernelGGL(reduce_kernel, dim3((n4 + 255) / 256), dim3(256), 0, 0, dP, dC, n4, S);
    };
    const float us = time_us(f);
    const double tf = 2.0 * s.M * s.N * s.K / us / 1e6;
    printf("  direct MT=%d NT=%d depth=%d waves=%dx%d splitK=%d wgs=%5d  %8.2f us  %6.1f TF  frac %.3f  err %.1e\n", MT, NT, DEPTH, WM, WN, S,
           grid.x * grid.y * S, us, tf, tf / 157.3, check(s.M, s.N));
}

static void run_lds(const Shape& s, int S) {
    dim3 grid((s.M + 63) / 64, (s.N + 63) / 64, S);
    const size_t n4 = (size_t)s.M * s.N / 4;
    auto f = [&] {
        hipLaunchKernelGGL(lds_kernel, grid, dim3(256), 0, 0, dA, s.K, dW, s.K, S > 1 ? dP : dC, s.N, s.M, s.N, s.K);
        if (S > 1) hipLaunchKernelGGL(reduce_kernel, dim3((n4 + 255) / 256), dim3(256), 0, 0, dP, dC, n4, S);
    };
    const float us = time_us(f);
    const double tf = 2.0 * s.M * s.N * s.K / us / 1e6;
    printf("  lds 64x64 splitK=%d                   wgs=%5d  %8.2f us  %6.1f TF  frac %.3f  err %.1e\n", S, grid.x * grid.y * S, us, tf, tf / 157.3, check(s.M, s.N));
}

int main() {
    const Shape shapes[] = {{"enc conv k5 512->512", 1376, 512, 2560}, {"enc lstm in-proj", 1376, 1024, 512}, {"memory layer", 1376, 256, 512},
                            {"bank conv k8 80->80", 8288, 80, 640}, {"proj conv 640->128", 8256, 128, 1920}, {"highway 80->80", 8256, 80, 80},
                            {"linear 160->1024", 8256, 1024, 160}, {"lstm wgrad-like", 4096, 1408, 2752}, {"big", 8192, 2048, 2048}};
    size_t maxA = 0, maxW = 0, maxC = 0;
    for (const Shape& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N); }
    CK(hipMalloc(&dA, maxA * 4)); CK(hipMalloc(&dW, maxW * 4)); CK(hipMalloc(&dC, maxC * 4)); CK(hipMalloc(&dR, maxC * 4)); CK(hipMalloc(&dP, maxC * 4 * 8));
    std::vector<float> h(std::max(maxA, maxW));
    srand(1);
    for (float& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(dA, h.data(), maxA * 4, hipMemcpyHostToDevice));
    for (float& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(dW, h.data(), maxW * 4, hipMemcpyHostToDevice));
    hC.resize(maxC); hR.resize(maxC);
    for (const Shape& s : shapes) {
        printf("%s  M=%d N=%d K=%d\n", s.name, s.M, s.N, s.K);
        hipLaunchKernelGGL(naive_kernel, dim3((s.N + 255) / 256, s.M), dim3(256), 0, 0, dA, s.K, dW, s.K, dR, s.N, s.M, s.N, s.K);
        CK(hipMemcpy(hR.data(), dR, (size_t)s.M * s.N * 4, hipMemcpyDeviceToHost));
        const bool huge = (size_t)s.M * s.N > 4u << 20;
        for (int S : {1, 2, 4, 8}) if (!(huge && S > 2)) run_lds(s, S);
        for (int S : {1, 2, 4, 8}) if (!(huge && S > 2)) run_direct<2, 2, 4, 1, 1>(s, S);
        for (int S : {1, 2, 4, 8}) if (!(huge && S > 2)) run_direct<2, 4, 3, 1, 1>(s, S);
        for (int S : {1, 2, 4}) if (!(huge && S > 2)) run_direct<1, 5, 6, 1, 1>(s, S);
    }
    return 0;
}
