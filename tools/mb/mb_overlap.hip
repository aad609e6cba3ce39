// Micro-benchmark (development tool): can the attention-step backward (32 workgroups, bound by one compute unit each) run BESIDE the
// decoder cell's dgates . W^T product (320 workgroups streaming 42 MB)?  Both kernels alone, back to back on one stream, and
// flooding two streams at once (an upper bound for hosting them in one launch).
// hipcc --offload-arch=gfx950 -O3 -o mb_overlap mb_overlap.hip ../../semi_tts_amd/csrc/runtime.hip
#include <hip/hip_runtime.h>
#define AB_PROF(n)
#include "../../semi_tts_amd/csrc/attention_bwd.hip"
#include "../../semi_tts_amd/csrc/skinny_packed.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

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
    // the product: (32 x 4096) . (4096 x 2560)
    const int KD = 4096, ND = 2560;
    int ks[1] = {KD};
    const size_t wf = st_packed_weight_floats(ks, 1, ND, 0), xf = st_t16_floats(B, KD);
    float *pw, *px, *py;
    CK(hipMalloc(&pw, wf * 4)); CK(hipMalloc(&px, xf * 4)); CK(hipMalloc(&py, (size_t)B * ND * 4));
    CK(hipMemset(pw, 0, wf * 4)); CK(hipMemset(px, 0, xf * 4));
    st_t16_view xv = {px, (KD + 15) / 16, 0};
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    auto ab = [&](hipStream_t st) { int rc = st_attn_step_bwd_s(pq, pm, mem, wprev, L, wcum, w, L, wc, wl, v, dctxs, ldc, 3, dws, ldw, 1, dcum, dwa + L, 2 * L,
                                                                 dpq, dhist, ds, loc, dloc, hist, dctx, dv, pm, B, L, A, E, F, K, st);
        if (rc) { printf("rc=%d %s\n", rc, st_last_error()); exit(1); } };
    auto pk = [&](hipStream_t st) { int rc = st_skinny_linear_packed_fwd(pw, &xv, KD, nullptr, ST_ACT_NONE, nullptr, 0, py, ND, nullptr, 0, nullptr, 0, 0, 0, 0,
                                                                         nullptr, 0, nullptr, B, ND, st);
        if (rc) { printf("rc=%d %s\n", rc, st_last_error()); exit(1); } };
    for (int i = 0; i < 5; ++i) { ab(s1); pk(s1); }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* what, auto fn) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1));
        fn();
        CK(hipStreamSynchronize(s2));
        CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-60s %.2f us per iteration\n", what, ms * 1e3 / 200);
    };
    timed("attention backward alone", [&] { for (int i = 0; i < 200; ++i) ab(s1); });
    timed("dgates . W^T product alone", [&] { for (int i = 0; i < 200; ++i) pk(s1); });
    timed("both, back to back on one stream", [&] { for (int i = 0; i < 200; ++i) { ab(s1); pk(s1); } });
    timed("both, flooding two streams", [&] { for (int i = 0; i < 200; ++i) { ab(s1); pk(s2); } });
    return 0;
}
