// mb_wstat.hip -- micro-benchmark (development tool): a WEIGHT-STATIONARY decode step, cells only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/mb_wstat.hip -o tools/mb/mb_wstat && tools/mb/mb_wstat
// The question (round-3 review, item 5): the two LSTM cells of the C2 decode step stream 72.7 MB of weights per step (7.7 + 9.4 us
// in situ).  The 75.5 MB of fp32 weights fit the chip's registers (256 CUs x 512 KB), so ONE persistent launch could keep a
// 1/256 row slice of both cells' weights resident and exchange only activations.  What does such a step cost?
// Model: 256 workgroups (one per compute unit) x 8 waves.  A workgroup owns 16 gate rows = 4 hidden units of each cell
// (one MFMA row tile); wave w holds columns [w K/8, (w+1) K/8) of those rows in registers (56 + 80 VGPRs of dummy weights).
// Per step and cell:
//   1. ALL-GATHER of the cell's input vector x (32 batch x K) published by the 256 workgroups of the previous phase as 8-byte
//      {value, tag} granules (the data is the flag: one relaxed agent-scope store / load each, guide recipe R2), laid out in MFMA
//      operand order so that a wave's slice is a contiguous run: 229 KB (K = 1792) / 328 KB (K = 2560) of payload = 458 / 656 KB of
//      granules per compute unit, read with 16-byte L1-bypassing loads and re-read while a tag is stale;
//   2. 2 batch tiles x K/32 exact-fp32 MFMAs per wave against the resident weights;
//   3. the 8 waves' partial sums meet in LDS (16 KB), a pointwise stand-in (tanh) makes the 4 x 32 new state values;
//   4. PUBLISH: the workgroup writes ITS K/256 columns x 32 granules of the NEXT cell's input (write-through stores).
// No attention, no projection, no prenet: a lower bound for the cells' share of such a step.  Variants: 'nowait' reads the
// granules once without looking at the tags (ingest + arithmetic only), 'nogather' skips the reads (arithmetic + publish only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(2))) u64 u64x2;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NWG = 256, NWAVE = 8, BATCH = 32;
constexpr int KQ = 1792, KD = 2560;          // reduction lengths of the query / decoder cell (C2)

// granule buffer of one cell input: [batch tile 2][k block K/16][lane 64][4 granules]; element (batch b, column k) lives at
// tile b / 16, block k / 16, lane (b % 16) + 16 * ((k % 16) / 4) ... any fixed bijection serves the timing; what matters is that a
// wave's slice (its K/8 columns, both batch tiles) is contiguous per tile and read as whole 16-byte pieces
__device__ __forceinline__ size_t gran_index(int K, int tile, int kblock, int lane, int j) { return ((((size_t)tile * (K / 16) + kblock) * 64 + lane) * 4 + j); }

template <int K, int MODE>      // MODE 0: wait for fresh tags, 1: read once, 2: no reads
__device__ __forceinline__ void cell(const u64* __restrict__ xin, u64* __restrict__ xout, int Kout, const float (&wreg)[K / 32], unsigned tag_in, unsigned tag_out,
                                     float* __restrict__ lds, unsigned* __restrict__ status, float& sink) {
    constexpr int KB = K / 16 / NWAVE;           // k blocks of this wave's slice
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xin, 0, BATCH * K * 8, 0x00020000);
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        f32x4 xv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (MODE == 2) { xv[t] = f32x4{1.f, 2.f, 3.f, 4.f}; continue; }
            // (buffer loads: one lane-offset register for every address of the kernel; sc1 = L1-bypassing, 16 bytes = two granules)
            const int soff = (int)(gran_index(K, t, wave * KB + kb, 0, 0) * 8);
            u64 g0, g1, g2, g3;
            int spins = 0;
            while (true) {
                asm volatile("" ::: "memory");           // (a plain buffer load is loop-invariant to the compiler: without this the poll is hoisted)
                const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 32, soff, 16));
                const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 32 + 16, soff, 16));
                g0 = ((u64)__float_as_uint(lo[1]) << 32) | __float_as_uint(lo[0]); g1 = ((u64)__float_as_uint(lo[3]) << 32) | __float_as_uint(lo[2]);
                g2 = ((u64)__float_as_uint(hi[1]) << 32) | __float_as_uint(hi[0]); g3 = ((u64)__float_as_uint(hi[3]) << 32) | __float_as_uint(hi[2]);
                const bool fresh = (unsigned)(g0 >> 32) == tag_in && (unsigned)(g1 >> 32) == tag_in && (unsigned)(g2 >> 32) == tag_in && (unsigned)(g3 >> 32) == tag_in;
                if (MODE == 1 || __all(fresh)) break;
                if (++spins > (1 << 16)) { if (lane == 0) atomicOr(status, 1u); break; }
            }
            xv[t] = f32x4{__uint_as_float((unsigned)g0), __uint_as_float((unsigned)g1), __uint_as_float((unsigned)g2), __uint_as_float((unsigned)g3)};
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[(kb * 4 + c) % (K / 32)], xv[t][c], acc[t], 0, 0, 0);
        asm volatile("" ::: "memory");
    }
    // the 8 waves' partial sums: [wave][tile][lane][4] in LDS, then 512 threads finish 16 rows x 32 batch
#pragma unroll
    for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4*>(lds + ((wave * 2 + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) s += lds[w * 512 + tid];
    s = tanhf(s * 1e-3f);
    sink += s;
    __syncthreads();
    // publish this workgroup's share of the next cell's input: Kout / 256 columns x 32 batch granules (the first 128 of them are
    // the 4 x 32 state values just made, the rest stands in for the other producers' columns), write-through 8-byte stores
    const int per = Kout / NWG * BATCH;          // granules this workgroup publishes
    for (int i = tid; i < per; i += blockDim.x) {
        const int col = blockIdx.x * (Kout / NWG) + i / BATCH, b = i % BATCH;
        const size_t gi = gran_index(Kout, b / 16, col / 16, (b % 16) + 16 * ((col % 16) / 4), col % 4);
        __hip_atomic_store(xout + gi, ((u64)tag_out << 32) | (u64)__float_as_uint(s + (float)i * 1e-6f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MODE>
__global__ __launch_bounds__(NWAVE * 64) void wstat_kernel(u64* xq, u64* xd, int steps, unsigned tag0, unsigned* status, float* out, unsigned long long* cycles) {
    __shared__ __attribute__((aligned(16))) float lds[NWAVE * 512];
    float wq[KQ / 32], wd[KD / 32];              // 56 + 80 registers of resident (dummy) weights: one MFMA A operand per 4 columns of the slice
#pragma unroll
    for (int i = 0; i < KQ / 32; ++i) wq[i] = 1e-3f * (float)((threadIdx.x * 7 + i) % 13);
#pragma unroll
    for (int i = 0; i < KD / 32; ++i) wd[i] = 1e-3f * (float)((threadIdx.x * 5 + i) % 11);
    float sink = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        // query cell: reads xq (published with tag 2s by the previous decoder phase / the host), publishes the decoder cell's input
        cell<KQ, MODE>(xq, xd, KD, wq, tag0 + 2 * s, tag0 + 2 * s + 1, lds, status, sink);
        // decoder cell: reads xd, publishes the next step's query-cell input
        cell<KD, MODE>(xd, xq, KQ, wd, tag0 + 2 * s + 1, tag0 + 2 * s + 2, lds, status, sink);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = sink; cycles[blockIdx.x] = t1 - t0; }
}

template <int MODE>
static void run(const char* name, u64* xq, u64* xd, unsigned* status, float* out, unsigned long long* cyc, int steps) {
    // the host publishes step 0's query-cell input with tag `tag0`; every run uses fresh tags
    static unsigned tag0 = 16;
    std::vector<u64> h((size_t)BATCH * KQ);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((u64)tag0 << 32) | 0x3f800000ull;
    CK(hipMemcpy(xq, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(status, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((wstat_kernel<MODE>), dim3(NWG), dim3(NWAVE * 64), 0, 0, xq, xd, steps, tag0, status, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned st = 0;
    CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> hc(NWG);
    CK(hipMemcpy(hc.data(), cyc, NWG * 8, hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (auto c : hc) mx = c > mx ? c : mx;
    printf("%-52s %3d steps: %8.2f us per step (two cells), %7.0f cycles per step in the slowest workgroup%s\n", name, steps, ms * 1e3 / steps, (double)mx / steps,
           st ? "  [a wait timed out!]" : "");
    tag0 += 2 * steps + 16;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
    if (prop.multiProcessorCount < NWG) { printf("needs %d compute units\n", NWG); return 0; }
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, wstat_kernel<0>, NWAVE * 64, 0));
    printf("resident workgroups per CU (occupancy API): %d\n", occ);
    u64 *xq, *xd; unsigned* status; float* out; unsigned long long* cyc;
    CK(hipMalloc(&xq, (size_t)BATCH * KQ * 8)); CK(hipMalloc(&xd, (size_t)BATCH * KD * 8));
    CK(hipMemset(xd, 0, (size_t)BATCH * KD * 8));
    CK(hipMalloc(&status, 4)); CK(hipMalloc(&out, NWG * 4)); CK(hipMalloc(&cyc, NWG * 8));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("granule all-gather + MFMA + reduce + publish", xq, xd, status, out, cyc, 100);
        run<1>("same, granules read once (tags ignored)", xq, xd, status, out, cyc, 100);
        run<2>("MFMA + reduce + publish only (no gather)", xq, xd, status, out, cyc, 100);
    }
    return 0;
}
