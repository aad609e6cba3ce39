// Micro-benchmark (development tool): cost of a grid-wide barrier between N co-resident workgroups (one per CU),
// the building block of a persistent decode kernel with weights resident in registers.
// hipcc --offload-arch=gfx950 -O3 -o mb_gridsync mb_gridsync.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);           // agent scope by default for global atomics
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// flag-array barrier: no contended read-modify-write; every workgroup publishes its own flag, wave 0 polls all of them
__device__ __forceinline__ void flag_barrier(unsigned* flags, unsigned value, int n) {
    __syncthreads();
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __atomic_store_n(flags + blockIdx.x, value, __ATOMIC_RELEASE);
        bool done;
        do {
            done = true;
            for (int i = threadIdx.x; i < n; i += 64) done = done && (__atomic_load_n(flags + i, __ATOMIC_ACQUIRE) >= value);
            done = __all(done);
        } while (!done);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void flag_kernel(unsigned* flags, int iters, float* sink) {
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        acc += (float)i;
        flag_barrier(flags, (unsigned)(i + 1), gridDim.x);
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void bar_kernel(unsigned* counter, int iters, float* sink, long long spin_guard) {
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        acc += (float)i;
        grid_barrier(counter, (unsigned)(i + 1) * gridDim.x);
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    printf("%s: %d CUs, cooperative launch %d\n", prop.name, prop.multiProcessorCount, prop.cooperativeLaunch);
    unsigned* counter; float* sink;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&sink, 4096 * 4));
    for (int nwg : {32, 128, 256}) {
        if (nwg > prop.multiProcessorCount) continue;
        int iters = 2000;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(counter, 0, 4));
            long long guard = 0;
            void* args[] = {&counter, &iters, &sink, &guard};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((const void*)bar_kernel, dim3(nwg), dim3(256), args, 0, nullptr));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("grid barrier over %3d workgroups: %.3f us per barrier\n", nwg, ms * 1e3 / iters);
        }
        unsigned* flags; CK(hipMalloc(&flags, 4096));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(flags, 0, 4096));
            void* args[] = {&flags, &iters, &sink};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((const void*)flag_kernel, dim3(nwg), dim3(256), args, 0, nullptr));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("flag  barrier over %3d workgroups: %.3f us per barrier\n", nwg, ms * 1e3 / iters);
        }
    }
    return 0;
}
