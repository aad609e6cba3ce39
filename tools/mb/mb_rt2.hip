// Micro-benchmark (development tool): phase stamps of the 2-D tiled LSTM cell (pk_lstm_rt2_kernel) at the C2 decode shapes, the two
// cells alternating as in the loop.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -o mb_rt2 mb_rt2.hip ../../semi_tts_amd/csrc/runtime.hip
#include <hip/hip_runtime.h>
__device__ unsigned long long g_rt2[2 * 128 * 8 * 8];
#define RT2_PROF(n) do { if ((threadIdx.x & 63) == 0) g_rt2[(((blockIdx.y * gridDim.x + blockIdx.x) * 8) + (threadIdx.x >> 6)) * 8 + (n)] = __builtin_readcyclecounter(); } while (0)
#include "../../semi_tts_amd/csrc/skinny_packed.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main() {
    const int B = 32, H = 1024;
    const int Ks[2] = {1792, 2560};
    float *w[2], *x[2], *h0, *ha, *c0, *c1, *bi, *bh, *as, *am;
    size_t hf = st_t16_floats(B, H);
    for (int i = 0; i < 2; ++i) {
        int ks[1] = {Ks[i]};
        size_t wf = st_packed_weight_floats(ks, 1, 4 * H, H), xf = st_t16_floats(B, Ks[i]);
        CK(hipMalloc(&w[i], wf * 4)); CK(hipMalloc(&x[i], xf * 4));
        CK(hipMemset(w[i], 0, wf * 4)); CK(hipMemset(x[i], 0, xf * 4));
    }
    CK(hipMalloc(&h0, hf * 4)); CK(hipMalloc(&ha, hf * 4)); CK(hipMalloc(&c0, B * H * 4)); CK(hipMalloc(&c1, B * H * 4));
    CK(hipMalloc(&bi, 16 * H)); CK(hipMalloc(&bh, 16 * H)); CK(hipMalloc(&as, B * H * 4)); CK(hipMalloc(&am, B * H * 4));
    CK(hipMemset(h0, 0, hf * 4)); CK(hipMemset(ha, 0, hf * 4)); CK(hipMemset(c0, 0, B * H * 4)); CK(hipMemset(bi, 0, 16 * H));
    CK(hipMemset(bh, 0, 16 * H)); CK(hipMemset(as, 0, B * H * 4)); CK(hipMemset(am, 0, B * H * 4));
    auto run = [&](int i) {
        st_t16_view xv = {x[i], (Ks[i] + 15) / 16, 0}, d0 = {h0, H / 16, 0}, da = {ha, H / 16, 0};
        int rc = st_lstm_cell_packed_fwd(w[i], &xv, Ks[i], bi, bh, c0, H, nullptr, &d0, nullptr, c1, H, nullptr, as, am, &da, B, H, nullptr);
        if (rc) { printf("rc=%d %s\n", rc, st_last_error()); exit(1); } };
    for (int i = 0; i < 20; ++i) run(i & 1);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 400; ++i) run(i & 1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("alternating cells, back to back (eager): %.2f us per launch\n", ms * 1e3 / 400);
    const char* nm[6] = {"kernel entry", "first two groups requested", "first group multiplied", "K loop done", "partials in LDS, barrier passed", "epilogue done"};
    for (int which = 0; which < 2; ++which) {
        run(1 - which); run(which);                      // the stamps of the LAST launch stay
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hp(2 * 128 * 8 * 8);
        CK(hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_rt2), hp.size() * 8));
        printf("K=%d: cycles since the workgroup's wave-0 entry (clocks of different XCDs are not comparable), median / max over 256 workgroups\n", Ks[which]);
        for (int n = 1; n < 6; ++n) {
            std::vector<long long> v0, vmax;
            for (int b = 0; b < 256; ++b) {
                const unsigned long long t0 = hp[(b * 8 + 0) * 8 + 0];
                long long mx = 0;
                const int nw = n == 5 ? 2 : 8;          // the epilogue runs on waves 0 and 1 only
                for (int wv = 0; wv < nw; ++wv) mx = std::max(mx, (long long)(hp[(b * 8 + wv) * 8 + n] - t0));
                v0.push_back((long long)(hp[(b * 8 + 0) * 8 + n] - t0));
                vmax.push_back(mx);
            }
            std::sort(v0.begin(), v0.end()); std::sort(vmax.begin(), vmax.end());
            printf("  %-34s wave 0: median %6lld max %6lld | slowest wave of the workgroup: median %6lld max %6lld\n", nm[n], v0[128], v0[255], vmax[128], vmax[255]);
        }
        // entry skew of the waves of a workgroup
        std::vector<long long> sk;
        for (int b = 0; b < 256; ++b) {
            long long mx = 0;
            for (int wv = 0; wv < 8; ++wv) mx = std::max(mx, (long long)(hp[(b * 8 + wv) * 8 + 0] - hp[(b * 8 + 0) * 8 + 0]));
            sk.push_back(mx);
        }
        std::sort(sk.begin(), sk.end());
        printf("  last wave of a workgroup enters %lld (median) / %lld (max) cycles after wave 0\n", sk[128], sk[255]);
    }
    return 0;
}
