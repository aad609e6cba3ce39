// loss.hip -- the spectrogram loss of the trainer (ref: src/util.py:80-126, called bin/train_vqvae.py:219-223):
//   loss = w_all * crit(pred, label) + w_low * crit(pred[..., :n_low], label[..., :n_low])
//        + w_diff * crit(pred[:, 1:] - pred[:, :-1], label[:, 1:] - label[:, :-1])          crit = mean squared / absolute error
// One pass produces the loss value AND d loss / d pred (the loss is a leaf of the training graph, so its
// backward is this gradient times the incoming scalar).  The scalar is reduced in two deterministic stages.
#include "st_common.h"

namespace {

constexpr int FL_BLOCKS = 256;

__device__ __forceinline__ float fl_val(float e, int l1) { return l1 ? fabsf(e) : e * e; }
__device__ __forceinline__ float fl_grad(float e, int l1) { return l1 ? (e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f)) : 2.0f * e; }

__global__ __launch_bounds__(256) void freq_loss_kernel(const float* pred, const float* label, float* dpred, float* partial,
                                                        int B, int T, int D, int n_low, float c_all, float c_low, float c_diff, int l1) {
    __shared__ float red[4];
    const size_t total = (size_t)B * T * D;
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int t = (int)((i / D) % T);
        const float e = pred[i] - label[i];
        float v = c_all * fl_val(e, l1), g = c_all * fl_grad(e, l1);
        if (c < n_low) { v += c_low * fl_val(e, l1); g += c_low * fl_grad(e, l1); }
        if (c_diff != 0.0f) {
            // delta[t] = e[t+1] - e[t] contributes for t = 0..T-2; e[t] appears in delta[t] (-) and delta[t-1] (+)
            if (t + 1 < T) {
                const float dl = (pred[i + D] - label[i + D]) - e;
                v += c_diff * fl_val(dl, l1);
                g -= c_diff * fl_grad(dl, l1);
            }
            if (t > 0) {
                const float dl = e - (pred[i - D] - label[i - D]);
                g += c_diff * fl_grad(dl, l1);
            }
        }
        dpred[i] = g;
        acc += v;
    }
    acc = st_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void freq_loss_final_kernel(const float* partial, int n, float* loss) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += partial[i];
    *loss = s;
}

// y[i] = x[i] * (*s)
__global__ __launch_bounds__(256) void scale_by_kernel(const float* x, const float* s, float* y, size_t n) {
    const float f = *s;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] * f;
}

}  // namespace

extern "C" int st_freq_loss(const float* pred, const float* label, float* loss, float* dpred, float* ws,
                            int B, int T, int D, int n_low, float w_all, float w_low, float w_diff, int l1, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(pred && label && loss && dpred && ws && B > 0 && T > 0 && D > 0 && n_low >= 0 && n_low <= D,
                 "st_freq_loss: bad arguments");
    ST_CHECK_ARG(w_diff == 0.0f || T > 1, "st_freq_loss: differential term needs T > 1");
    const float c_all = w_all / ((float)B * T * D);
    const float c_low = n_low > 0 ? w_low / ((float)B * T * n_low) : 0.0f;
    const float c_diff = w_diff != 0.0f ? w_diff / ((float)B * (T - 1) * D) : 0.0f;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(freq_loss_kernel, dim3(FL_BLOCKS), dim3(256), 0, st, pred, label, dpred, ws, B, T, D,
                       c_low != 0.0f ? n_low : 0, c_all, c_low, c_diff, l1);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(freq_loss_final_kernel, dim3(1), dim3(1), 0, st, ws, FL_BLOCKS, loss);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_scale_by(const float* x, const float* scalar, float* y, size_t n, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(x && scalar && y && n > 0, "st_scale_by: bad arguments");
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_by_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, scalar, y, n);
    ST_LAUNCH_CHECK();
    return 0;
}
