// Micro-benchmark (development tool): the two LSTM cells of a teacher-forced step (query cell of step t+1, decoder cell of step t) are
// independent -- what would one launch for both buy?  Each alone, back to back on one stream, flooding two streams.
// hipcc --offload-arch=gfx950 -O3 -o mb_overlap2 mb_overlap2.hip ../../semi_tts_amd/csrc/runtime.hip
#include <hip/hip_runtime.h>
#include "../../semi_tts_amd/csrc/skinny_packed.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

struct Cell { float *w, *x, *h0, *h1, *ha, *c0, *c1, *bi, *bh, *as, *am; int K; };

int main() {
    const int B = 32, H = 1024;
    Cell cells[2];
    int Ks[2] = {1792, 2560};
    for (int i = 0; i < 2; ++i) {
        Cell& c = cells[i]; c.K = Ks[i];
        int ks[1] = {c.K};
        size_t wf = st_packed_weight_floats(ks, 1, 4 * H, H), xf = st_t16_floats(B, c.K), hf = st_t16_floats(B, H);
        CK(hipMalloc(&c.w, wf * 4)); CK(hipMalloc(&c.x, xf * 4)); CK(hipMalloc(&c.h0, hf * 4)); CK(hipMalloc(&c.h1, hf * 4)); CK(hipMalloc(&c.ha, hf * 4));
        CK(hipMalloc(&c.c0, B * H * 4)); CK(hipMalloc(&c.c1, B * H * 4)); CK(hipMalloc(&c.bi, 4 * H * 4)); CK(hipMalloc(&c.bh, 4 * H * 4));
        CK(hipMalloc(&c.as, B * H * 4)); CK(hipMalloc(&c.am, B * H * 4));
        CK(hipMemset(c.w, 0, wf * 4)); CK(hipMemset(c.x, 0, xf * 4)); CK(hipMemset(c.c0, 0, B * H * 4)); CK(hipMemset(c.bi, 0, 16 * H)); CK(hipMemset(c.bh, 0, 16 * H));
        CK(hipMemset(c.as, 0, B * H * 4)); CK(hipMemset(c.am, 0, B * H * 4)); CK(hipMemset(c.h0, 0, hf * 4)); CK(hipMemset(c.h1, 0, hf * 4)); CK(hipMemset(c.ha, 0, hf * 4));
    }
    auto run = [&](int i, hipStream_t st) {
        Cell& c = cells[i];
        st_t16_view xv = {c.x, (c.K + 15) / 16, 0}, d0 = {c.h0, H / 16, 0}, d1 = {c.h1, H / 16, 0}, da = {c.ha, H / 16, 0};
        int rc = st_lstm_cell_packed_fwd(c.w, &xv, c.K, c.bi, c.bh, c.c0, H, nullptr, &d0, &d1, c.c1, H, nullptr, c.as, c.am, &da, B, H, st);
        if (rc) { printf("rc=%d %s\n", rc, st_last_error()); exit(1); } };
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int i = 0; i < 5; ++i) { run(0, s1); run(1, s1); }
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
    timed("query cell (K = 1792) alone", [&] { for (int i = 0; i < 200; ++i) run(0, s1); });
    timed("decoder cell (K = 2560) alone", [&] { for (int i = 0; i < 200; ++i) run(1, s1); });
    timed("both, back to back on one stream", [&] { for (int i = 0; i < 200; ++i) { run(0, s1); run(1, s1); } });
    timed("both, flooding two streams", [&] { for (int i = 0; i < 200; ++i) { run(0, s1); run(1, s2); } });
    return 0;
}
