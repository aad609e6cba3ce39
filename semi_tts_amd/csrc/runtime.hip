// runtime.hip -- error reporting, HIP stream / graph / event plumbing and small utility
// kernels of libsemitts_hip.so.
#include <stdarg.h>
#include "st_common.h"

static thread_local char g_err[512] = "";

void st_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* st_last_error(void) { return g_err; }
extern "C" int st_abi_version(void) { return ST_ABI_VERSION; }

extern "C" int st_device_info(int* n_cu, int* lds_bytes, char* name, int name_len) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    int dev = 0;
    ST_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    ST_HIP(hipGetDeviceProperties(&p, dev));
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    return 0;
}

// ---------------------------------------------------------------- streams / graphs / events
extern "C" int st_stream_create(void** stream_out) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(stream_out, "st_stream_create: null out pointer");
    hipStream_t s;
    ST_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = (void*)s;
    return 0;
}
extern "C" int st_stream_destroy(void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
extern "C" int st_stream_sync(void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
extern "C" int st_graph_begin(void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}
extern "C" int st_graph_end(void* stream, void** graph_exec_out) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(graph_exec_out, "st_graph_end: null out pointer");
    hipGraph_t graph = nullptr;
    ST_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        st_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return -5;
    }
    *graph_exec_out = (void*)exec;
    return 0;
}
extern "C" int st_graph_launch(void* graph_exec, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}
extern "C" int st_graph_destroy(void* graph_exec) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}
extern "C" int st_event_create(void** ev_out) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(ev_out, "st_event_create: null out pointer");
    hipEvent_t e;
    ST_HIP(hipEventCreate(&e));
    *ev_out = (void*)e;
    return 0;
}
extern "C" int st_event_record(void* ev, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
extern "C" int st_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(ms_out, "st_event_elapsed_ms: null out pointer");
    ST_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    ST_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return 0;
}
extern "C" int st_event_destroy(void* ev) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}

// ---------------------------------------------------------------- utility kernels
namespace {

__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ __launch_bounds__(256) void copy2d_kernel(float* dst, int ldd, const float* src, int lds, int rows, int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ldd + c] = src[r * lds + c];
    }
}

__global__ __launch_bounds__(256) void mean_rows_kernel(const float* src, float* dst, int B, int T, int D) {
    const int total = B * D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / D, d = i - b * D;
        float s = 0.0f;
        for (int t = 0; t < T; ++t) s += src[((size_t)b * T + t) * D + d];
        dst[i] = s / (float)T;
    }
}

inline int grid_for(size_t n, int cap = 2048) {
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > (size_t)cap ? cap : b);
}

}  // namespace

extern "C" int st_fill(float* p, float v, size_t n, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(p || n == 0, "st_fill: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, v, n);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_copy2d(float* dst, int ldd, const float* src, int lds, int rows, int cols, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(dst && src && rows > 0 && cols > 0 && ldd >= cols && lds >= cols, "st_copy2d: bad arguments");
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((size_t)rows * cols)), dim3(256), 0, (hipStream_t)stream,
                       dst, ldd, src, lds, rows, cols);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_mean_rows(const float* src, float* dst, int B, int T, int D, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(src && dst && B > 0 && T > 0 && D > 0, "st_mean_rows: bad arguments");
    hipLaunchKernelGGL(mean_rows_kernel, dim3(grid_for((size_t)B * D)), dim3(256), 0, (hipStream_t)stream, src, dst, B, T, D);
    ST_LAUNCH_CHECK();
    return 0;
}
