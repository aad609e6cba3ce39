// Micro-benchmark (development tool): does an LSTM-cell launch get shorter when its weights are already in the L2 of the XCD that reads them?
// Sequence A: the two cells of a decode step alternate (71 MB of weights cycle through 8 x 4 MB of L2: every launch streams from the Infinity
// Cache).  Sequence B: in front of each cell a "warm" launch with the SAME grid reads exactly the weight tiles the cell's workgroup of the
// same index will read (same flat workgroup id -> same XCD under round-robin placement).  Compare the cells' durations in a kernel trace:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mb_l2warm mb_l2warm.hip -L../../semi_tts_amd/lib -lsemitts_hip -Wl,-rpath,'$ORIGIN/../../semi_tts_amd/lib'
//   rocprofv3 --kernel-trace --stats -- ./mb_l2warm
#include <hip/hip_runtime.h>
#include "../../include/semitts.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(512) void warm_kernel(const f32x4* wp, const int w_kbs, const int KB, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile0 = blockIdx.x * 2;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int rt = 0; rt < 2; ++rt) {
        const f32x4* p = wp + (size_t)(tile0 + rt) * w_kbs * 64 + lane;
        for (int kb = wave; kb < KB; kb += 8) { const f32x4 v = p[(size_t)kb * 64]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
    }
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = 1.0f;      // (never true: keeps the loads)
}
// the same amount of traffic from a buffer nobody else reads (a "warm" launch that does not help anybody: the control of sequence B)
__global__ __launch_bounds__(512) void cold_kernel(const f32x4* wp, const int w_kbs, const int KB, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile0 = blockIdx.x * 2;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int rt = 0; rt < 2; ++rt) {
        const f32x4* p = wp + (size_t)(tile0 + rt) * w_kbs * 64 + lane;
        for (int kb = wave; kb < KB; kb += 8) { const f32x4 v = p[(size_t)kb * 64]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
    }
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = 1.0f;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;       // 0: A then B then C
    const int B = 32, H = 1024;
    const int Ks[2] = {1792, 2560};
    float *w[2], *wc[2], *x[2], *h0, *ha, *c0, *c1, *bi, *bh, *as, *am, *sink;
    size_t hf = st_t16_floats(B, H);
    for (int i = 0; i < 2; ++i) {
        int ks[1] = {Ks[i]};
        size_t wf = st_packed_weight_floats(ks, 1, 4 * H, H), xf = st_t16_floats(B, Ks[i]);
        CK(hipMalloc(&w[i], wf * 4)); CK(hipMalloc(&wc[i], wf * 4)); CK(hipMalloc(&x[i], xf * 4));
        CK(hipMemset(w[i], 0, wf * 4)); CK(hipMemset(wc[i], 0, wf * 4)); CK(hipMemset(x[i], 0, xf * 4));
    }
    CK(hipMalloc(&h0, hf * 4)); CK(hipMalloc(&ha, hf * 4)); CK(hipMalloc(&c0, B * H * 4)); CK(hipMalloc(&c1, B * H * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&bi, 16 * H)); CK(hipMalloc(&bh, 16 * H)); CK(hipMalloc(&as, B * H * 4)); CK(hipMalloc(&am, B * H * 4));
    CK(hipMemset(h0, 0, hf * 4)); CK(hipMemset(ha, 0, hf * 4)); CK(hipMemset(c0, 0, B * H * 4)); CK(hipMemset(bi, 0, 16 * H));
    CK(hipMemset(bh, 0, 16 * H)); CK(hipMemset(as, 0, B * H * 4)); CK(hipMemset(am, 0, B * H * 4));
    auto cell = [&](int i) {
        st_t16_view xv = {x[i], (Ks[i] + 15) / 16, 0}, d0 = {h0, H / 16, 0}, da = {ha, H / 16, 0};
        int rc = st_lstm_cell_packed_fwd(w[i], &xv, Ks[i], bi, bh, c0, H, nullptr, &d0, nullptr, c1, H, nullptr, as, am, &da, B, H, nullptr);
        if (rc) { printf("rc=%d %s\n", rc, st_last_error()); exit(1); } };
    auto warm = [&](int i, bool real) {
        const int kbs = (Ks[i] + 15) / 16;
        if (real) hipLaunchKernelGGL(warm_kernel, dim3(128, 2), dim3(512), 0, 0, reinterpret_cast<const f32x4*>(w[i]), kbs, kbs, sink);
        else hipLaunchKernelGGL(cold_kernel, dim3(128, 2), dim3(512), 0, 0, reinterpret_cast<const f32x4*>(wc[i]), kbs, kbs, sink); };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, auto body) {
        for (int i = 0; i < 20; ++i) body(i & 1);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 400; ++i) body(i & 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-70s %.2f us per iteration\n", name, ms * 1e3 / 400);
    };
    if (mode == 0 || mode == 1) timed("A: cell alone (alternating K = 1792 / 2560)", [&](int i) { cell(i); });
    if (mode == 0 || mode == 2) timed("B: warm launch over the cell's own weights, then the cell", [&](int i) { warm(i, true); cell(i); });
    if (mode == 0 || mode == 3) timed("C: the same traffic from another buffer, then the cell (control)", [&](int i) { warm(i, false); cell(i); });
    return 0;
}
