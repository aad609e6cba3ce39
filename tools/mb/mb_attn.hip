// Micro-benchmark (development tool): phase timing of the fused attention step kernel.
// Includes the product source with AT_PROF defined so each wave stamps s_memtime at phase
// boundaries.  hipcc --offload-arch=gfx950 -O3 -o mb_attn mb_attn.hip ../../semi_tts_amd/csrc/runtime.hip
#include <hip/hip_runtime.h>
__device__ unsigned long long g_prof[64 * 8 * 16];
#define AT_PROF(n) do { if ((threadIdx.x & 63) == 0) g_prof[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (n)] = __builtin_readcyclecounter(); } while (0)
#include "../../semi_tts_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main() {
    const int B = 32, L = 43, A = 256, E = 512, F = 32, K = 31;
    auto dalloc = [&](size_t n, float v) { float* p; std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = v * (float)((i * 2654435761u) % 1000) / 1000.0f;
        CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p; };
    float* pq = dalloc((size_t)B * A, 1), *pm = dalloc((size_t)B * L * A, 1), *mem = dalloc((size_t)B * L * E, 1);
    float* wprev = dalloc((size_t)B * L, 0.02f), *wcum = dalloc((size_t)B * L, 0.1f);
    float* wout = dalloc((size_t)B * L, 0), *wcum2 = dalloc((size_t)B * L, 0);
    float* wc = dalloc((size_t)F * 2 * K, 0.1f), *wl = dalloc((size_t)A * F, 0.1f), *v = dalloc(A, 0.1f);
    float* ctx = dalloc((size_t)B * E, 0), *ctxt = dalloc((size_t)2 * 32 * 256, 0);
    st_t16_view cv = {ctxt, 32, 0};
    auto run = [&] { int rc = st_attn_step_t16_fwd(pq, pm, mem, wprev, L, wcum, wout, L, wcum2, wc, wl, v, &cv, 1, ctx, E, B, L, A, E, F, K, nullptr);
        if (rc) { printf("rc=%d %s\n", rc, st_last_error()); exit(1); } };
    for (int i = 0; i < 5; ++i) run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) run();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("attention step (with stamps): %.2f us/launch\n", ms * 1e3 / 200);
    std::vector<unsigned long long> h(64 * 8 * 16);
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_prof), h.size() * 8));
    const char* names[13] = {"start", "mem prefetch issued", "staging done", "sync", "conv done", "sync", "energy done", "sync",
                             "softmax done", "sync", "ctx partial done", "sync", "end"};
    for (int blk : {0, 17}) {
        printf("block %d (cycles since wave 0 start; s_memtime ticks):\n", blk);
        unsigned long long t0 = h[(blk * 8 + 0) * 16 + 0];
        for (int n = 0; n < 13; ++n) {
            printf("  %-22s", names[n]);
            for (int w = 0; w < 8; ++w) printf(" %7lld", (long long)(h[(blk * 8 + w) * 16 + n] - t0));
            printf("\n");
        }
    }
    return 0;
}
