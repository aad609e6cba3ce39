// Micro-benchmark (development tool): phase timing of the attention step kernels (at_kernel<.,PART>) at the decode shape.
// hipcc --offload-arch=gfx950 -O3 -I../../include -o mb_attn mb_attn.hip ../../semi_tts_amd/csrc/runtime.hip
#include <hip/hip_runtime.h>
__device__ unsigned long long g_prof[512 * 16];
#define AT_PROF(n) do { if (threadIdx.x == 0) g_prof[blockIdx.x * 16 + (n)] = __builtin_readcyclecounter(); } while (0)
#include "../../semi_tts_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main() {
    const int B = 32, L = 43, A = 256, E = 512, F = 32, K = 31;
    float *pq, *s, *mem, *cum, *wout, *cumo, *v, *ctx, *wc, *wl, *wprev;
    CK(hipMalloc(&pq, B * A * 4)); CK(hipMalloc(&s, B * L * A * 4)); CK(hipMalloc(&mem, B * L * E * 4)); CK(hipMalloc(&cum, B * L * 4));
    CK(hipMalloc(&wout, B * L * 4)); CK(hipMalloc(&cumo, B * L * 4)); CK(hipMalloc(&v, A * 4)); CK(hipMalloc(&ctx, B * E * 4));
    CK(hipMalloc(&wc, F * 2 * K * 4)); CK(hipMalloc(&wl, A * F * 4)); CK(hipMalloc(&wprev, B * L * 4));
    CK(hipMemset(pq, 0, B * A * 4)); CK(hipMemset(s, 0, B * L * A * 4)); CK(hipMemset(mem, 0, B * L * E * 4)); CK(hipMemset(cum, 0, B * L * 4));
    CK(hipMemset(v, 0, A * 4)); CK(hipMemset(wc, 0, F * 2 * K * 4)); CK(hipMemset(wl, 0, A * F * 4)); CK(hipMemset(wprev, 0, B * L * 4));
    float* junk; CK(hipMalloc(&junk, 64 << 20));
    const char* nm[13] = {"start", "mem rows requested", "-", "staged (pre-sync)", "synced", "W_l^T stored", "synced", "energies done", "synced",
                          "softmax done", "synced", "ctx partials done", "synced / stored"};
    const bool warm = getenv("MB_WARM") != nullptr;
    for (int mode = 0; mode < 5; ++mode) {     // 0..2: fin with 1/2/4 parts; 3..4: pre with 1/2 parts
        const int parts = mode < 3 ? (1 << mode) : (mode - 2);
        auto run = [&] {
            int rc = mode < 3 ? st_attn_fin_t16_fwd(pq, s, mem, cum, wout, L, cumo, v, nullptr, 0, ctx, E, parts, B, L, A, E, F, K, nullptr)
                              : st_attn_pre_fwd(s, wprev, L, cum, wc, wl, s, parts, B, L, A, F, K, nullptr);
            if (rc) { printf("rc=%d %s\n", rc, st_last_error()); exit(1); } };
        for (int i = 0; i < 3; ++i) run();
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float tot = 0;
        for (int it = 0; it < 50; ++it) {
            if (!warm) CK(hipMemsetAsync(junk, it, 64 << 20, nullptr));      // operands leave the L2s, as they do between decode steps
            CK(hipEventRecord(e0)); run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        std::vector<unsigned long long> hp(512 * 16);
        CK(hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_prof), hp.size() * 8));
        const int nwg = B * parts;
        printf("%s parts=%d: %.2f us/launch (event to event, cold L2); cycles since workgroup start, median / max over %d workgroups\n",
               mode < 3 ? "fin" : "pre", parts, tot * 1e3 / 50, nwg);
        for (int n = 1; n < 13; ++n) {
            if (mode >= 3 && n > 7) break;
            std::vector<long long> d;
            for (int b = 0; b < nwg; ++b) d.push_back((long long)(hp[b * 16 + n] - hp[b * 16]));
            std::sort(d.begin(), d.end());
            printf("  %2d %-22s %7lld %7lld\n", n, nm[n], d[nwg / 2], d[nwg - 1]);
        }
    }
    return 0;
}
