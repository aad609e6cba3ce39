// Micro-benchmark (development tool): phase timing of the attention-step backward kernel.
// hipcc --offload-arch=gfx950 -O3 -o mb_attn_bwd mb_attn_bwd.hip ../../semi_tts_amd/csrc/runtime.hip
#include <hip/hip_runtime.h>
__device__ unsigned long long g_prof[128 * 8 * 16];
#define AB_PROF(n) do { if ((threadIdx.x & 63) == 0) g_prof[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (n)] = __builtin_readcyclecounter(); } while (0)
#include "../../semi_tts_amd/csrc/attention_bwd.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// MB_NS=2|4: the split form (ab_body<true, 48, NS>: NS workgroups per utterance) on its own
template <int NS>
__global__ __launch_bounds__(AB_THREADS) void ab_split_kernel(const AbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ab_split_lds[];
    const int part = blockIdx.x / a.B;
    ab_body<true, AB_LBLK_MAX, NS>(a, blockIdx.x - part * a.B, ab_split_lds, part);
}

__global__ __launch_bounds__(AB_THREADS) void ab_hist_kernel(const AbHistArgs h) {
    extern __shared__ __attribute__((aligned(16))) float ab_hist_lds[];
    ab_hist_body(h, blockIdx.x, ab_hist_lds);
}

int main() {
    const int B = 32, L = 43, A = 256, E = 512, F = 32, K = 31;
    auto dalloc = [&](size_t n, float v) { float* p; std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = v * (float)((i * 2654435761u) % 1000) / 1000.0f;
        CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p; };
    float* pq = dalloc((size_t)B * A, 1), *pm = dalloc((size_t)B * L * A, 1), *mem = dalloc((size_t)B * L * E, 1);
    float* wprev = dalloc((size_t)B * L, 0.02f), *wcum = dalloc((size_t)B * L, 0.1f), *w = dalloc((size_t)B * L, 0.02f);
    float* wc = dalloc((size_t)F * 2 * K, 0.1f), *wl = dalloc((size_t)A * F, 0.1f), *v = dalloc(A, 0.1f);
    float* d0 = dalloc((size_t)B * E, 1), *d1 = dalloc((size_t)B * E, 1), *d2 = dalloc((size_t)B * E, 1);
    float* dwa = dalloc((size_t)B * 2 * L, 1), *dcum = dalloc((size_t)B * L, 0);
    float* dpq = dalloc((size_t)B * A, 0), *dhist = dalloc((size_t)B * 2 * L, 0);
    float* ds = dalloc((size_t)B * L * A, 0), *loc = dalloc((size_t)B * L * F, 0), *dloc = dalloc((size_t)B * L * F, 0);
    float* hist = dalloc((size_t)B * L * 2, 0), *dctx = dalloc((size_t)B * E, 0), *dv = dalloc((size_t)B * A, 0);
    const float* dctxs[3] = {d0, d1, d2}; const int ldc[3] = {E, E, E};
    const float* dws[1] = {dwa}; const int ldw[1] = {2 * L};
    const bool has_s = getenv("MB_S") != nullptr;      // S of the step given (training keeps it): no conv / W_l recompute
    auto run = [&] { int rc = st_attn_step_bwd_s(pq, pm, mem, wprev, L, wcum, w, L, wc, wl, v, dctxs, ldc, 3, dws, ldw, 1, dcum, dwa + L, 2 * L,
                                                dpq, dhist, ds, loc, dloc, hist, dctx, dv, has_s ? pm : nullptr, B, L, A, E, F, K, nullptr);
        if (rc) { printf("rc=%d %s\n", rc, st_last_error()); exit(1); } };
    const int ns = getenv("MB_NS") ? atoi(getenv("MB_NS")) : 1;
    float* dloc_part = dalloc((size_t)4 * B * L * F, 0);
    auto run_split = [&] {
        AbArgs a;
        if (ab_fill(a, nullptr, pq, pm, mem, wprev, L, wcum, w, L, wc, wl, v, dctxs, ldc, 3, dws, ldw, 1, dcum, nullptr, 0, dpq, dhist, ds, loc, dloc, hist, dctx, dv,
                    pm, B, L, A, E, F, K)) { printf("%s\n", st_last_error()); exit(1); }
        a.dloc_part = dloc_part;
        const size_t lds = ab_lds_bytes(a, true, ns);
        if (ns == 2) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ab_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                       hipLaunchKernelGGL(ab_split_kernel<2>, dim3(2 * B), dim3(AB_THREADS), lds, 0, a); }
        else { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ab_split_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
               hipLaunchKernelGGL(ab_split_kernel<4>, dim3(4 * B), dim3(AB_THREADS), lds, 0, a); }
    };
    auto run1 = run;
    const bool do_hist = getenv("MB_HIST") != nullptr;
    auto run_hist = [&] {
        AbHistArgs h;
        h.dloc_part = dloc_part; h.parts = 2; h.loc_conv_w = wc; h.w_prev = wprev; h.ld_wprev = L; h.w_cum_prev = wcum;
        h.dloc_t = dloc; h.hist_t = hist; h.dhist = dhist; h.dcum = dcum; h.B = B; h.L = L; h.F = F; h.K = K;
        hipLaunchKernelGGL(ab_hist_kernel, dim3(B), dim3(AB_THREADS), ab_hist_lds_floats(L, F, K) * 4, 0, h);
    };
    auto runx = [&] { if (do_hist) run_hist(); else if (ns > 1) run_split(); else run1(); };
#define run runx
    for (int i = 0; i < 5; ++i) run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) run();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("attention backward step (with stamps): %.2f us/launch\n", ms * 1e3 / 200);
    std::vector<unsigned long long> h(128 * 8 * 16);
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_prof), h.size() * 8));
    const char* names[8] = {"start", "P0 staged", "P1 conv done", "P2 dw done", "softmax bwd done", "P3 energy grad done", "P4 fold done", "end"};
    for (int blk : {0, 17}) {
        printf("block %d (cycles since wave 0 start):\n", blk);
        unsigned long long t0 = h[(blk * 8 + 0) * 16 + 0];
        const char* sub[7] = {"  p0: loads issued", "  p0: hist done", "  p0: pads done", "  p0: scatter done", "  p3: ds computed", "  p3: past barrier", "  p3: dloc MFMAs done"};
        for (int n = 8; n < 15; ++n) {
            printf("  %-22s", sub[n - 8]);
            for (int wv = 0; wv < 8; ++wv) printf(" %8lld", (long long)(h[(blk * 8 + wv) * 16 + n] - t0));
            printf("\n");
        }
        for (int n = 0; n < 8; ++n) {
            printf("  %-22s", names[n]);
            for (int wv = 0; wv < 8; ++wv) printf(" %8lld", (long long)(h[(blk * 8 + wv) * 16 + n] - t0));
            printf("\n");
        }
    }
    return 0;
}
