// norm.hip -- row-wise normalisations of the reference's optional layers (gfx950):
//   * nn.LayerNorm over the last dimension: the CTC speech encoder's `layer_norm=True` (src/asr.py:38-39,58) and the normalised
//     prenet Linear (`prenet_norm_type='LayerNorm'`, src/module.py:508-521)
//   * log_softmax over the last dimension: ASRPostnet's output (src/asr.py:80)
//   * the per-step prenet normalisation of the decode loop (LayerNorm, or BatchNorm1d over the B rows of a step) fused with the
//     ReLU, the dropout mask and the T16 tiling of the layer's output (decoder.hip, prenet_own)
// One wavefront per row; a row of up to 1024 columns stays in registers between the passes (mean, then the centred second moment,
// as ATen does: no E[x^2] - E[x]^2 cancellation).
#include "st_common.h"

namespace {

constexpr int NR_WAVES = 4;           // rows per workgroup
constexpr int NR_REG = 16;            // columns per lane kept in registers (N <= 1024); longer rows are re-read

__device__ __forceinline__ void nr_row_stats(const float* xr, int N, int lane, float (&v)[NR_REG], float& mean, float& rstd, float eps) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NR_REG; ++j) {
        const int n = lane + j * 64;
        v[j] = n < N ? xr[n] : 0.0f;
        s += v[j];
    }
    for (int n = lane + NR_REG * 64; n < N; n += 64) s += xr[n];
    mean = st_wave_sum(s) / (float)N;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < NR_REG; ++j) {
        const int n = lane + j * 64;
        const float d = n < N ? v[j] - mean : 0.0f;
        q = fmaf(d, d, q);
    }
    for (int n = lane + NR_REG * 64; n < N; n += 64) { const float d = xr[n] - mean; q = fmaf(d, d, q); }
    rstd = 1.0f / sqrtf(st_wave_sum(q) / (float)N + eps);
}

__global__ __launch_bounds__(NR_WAVES * 64) void layer_norm_fwd_kernel(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                                                                       float* y, int ldy, float* mean_out, float* rstd_out, int M, int N) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * NR_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    float v[NR_REG], mean, rstd;
    nr_row_stats(xr, N, lane, v, mean, rstd, eps);
    float* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int j = 0; j < NR_REG; ++j) {
        const int n = lane + j * 64;
        if (n < N) yr[n] = fmaf((v[j] - mean) * rstd, gamma ? gamma[n] : 1.0f, beta ? beta[n] : 0.0f);
    }
    for (int n = lane + NR_REG * 64; n < N; n += 64) yr[n] = fmaf((xr[n] - mean) * rstd, gamma ? gamma[n] : 1.0f, beta ? beta[n] : 0.0f);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// dx = rstd * (g - mean_n(g) - xhat * mean_n(g * xhat)),  g = dy * gamma;  dyxhat = dy * xhat (its column sum is d gamma)
__global__ __launch_bounds__(NR_WAVES * 64) void layer_norm_bwd_kernel(const float* dy, int lddy, const float* x, int ldx, const float* gamma,
                                                                       const float* mean, const float* rstd, float* dx, int lddx,
                                                                       float* dyxhat, int M, int N) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * NR_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    const float* dr = dy + (size_t)row * lddy;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.0f, s2 = 0.0f;
    for (int n = lane; n < N; n += 64) {
        const float g = dr[n] * (gamma ? gamma[n] : 1.0f), xh = (xr[n] - mu) * rs;
        s1 += g;
        s2 = fmaf(g, xh, s2);
    }
    s1 = st_wave_sum(s1) / (float)N;
    s2 = st_wave_sum(s2) / (float)N;
    for (int n = lane; n < N; n += 64) {
        const float d = dr[n], g = d * (gamma ? gamma[n] : 1.0f), xh = (xr[n] - mu) * rs;
        dx[(size_t)row * lddx + n] = rs * (g - s1 - xh * s2);
        if (dyxhat) dyxhat[(size_t)row * N + n] = d * xh;
    }
}

__global__ __launch_bounds__(NR_WAVES * 64) void log_softmax_fwd_kernel(const float* x, float* y, int M, int N) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * NR_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * N;
    float m = -INFINITY;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, xr[n]);
    m = st_wave_max(m);
    float s = 0.0f;
    for (int n = lane; n < N; n += 64) s += expf(xr[n] - m);
    const float lse = m + logf(st_wave_sum(s));
    for (int n = lane; n < N; n += 64) y[(size_t)row * N + n] = xr[n] - lse;
}

// dx = dy - exp(y) * sum_n dy
__global__ __launch_bounds__(NR_WAVES * 64) void log_softmax_bwd_kernel(const float* dy, const float* y, float* dx, int M, int N) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * NR_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    float s = 0.0f;
    for (int n = lane; n < N; n += 64) s += dy[(size_t)row * N + n];
    s = st_wave_sum(s);
    for (int n = lane; n < N; n += 64) dx[(size_t)row * N + n] = dy[(size_t)row * N + n] - expf(y[(size_t)row * N + n]) * s;
}

// One prenet layer of a decode step after its Linear: y (B, P) natural  ->  relu(norm(y)) * mask  in T16 order.
//   mode 1 LayerNorm over the P columns of a row; mode 2 BatchNorm1d with the running statistics (eval);
//   mode 3 BatchNorm1d with the statistics of the step's B rows (training: biased variance to normalise, running statistics updated
//   with the unbiased one and momentum, num_batches_tracked += 1 -- what nn.BatchNorm1d does on a (B, P) input)
struct PnArgs {
    const float* y; int ldy; int mode;
    const float* gamma; const float* beta; float* run_mean; float* run_var; long long* nbt; float eps, momentum;
    const float* mask; int ldmask;
    float* dst; int kb_stride, kb0;
    int B, P, b0;                 // rows b0 .. B-1 (the batch of a BatchNorm1d step = the rows that feed their own output back)
};

__device__ __forceinline__ size_t pn_t16_off(int b, int k, int KB) {
    return (((size_t)(b >> 4) * KB + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (b & 15)) * 4 + (k & 3);
}

__global__ __launch_bounds__(256) void prenet_norm_kernel(const PnArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, P = a.P;
    if (a.mode == 1) {
        for (int b = a.b0 + blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
            const float* yr = a.y + (size_t)b * a.ldy;
            float v[NR_REG], mean, rstd;
            nr_row_stats(yr, P, lane, v, mean, rstd, a.eps);
            for (int n = lane; n < P; n += 64) {
                float o = fmaf((yr[n] - mean) * rstd, a.gamma[n], a.beta[n]);
                o = o > 0.0f ? o : 0.0f;
                if (a.mask) o *= a.mask[(size_t)b * a.ldmask + n];
                a.dst[pn_t16_off(b, a.kb0 * 16 + n, a.kb_stride)] = o;
            }
        }
        return;
    }
    for (int n = blockIdx.x * 256 + tid; n < P; n += gridDim.x * 256) {     // one thread per column (B <= a few dozen rows)
        float mean, var;
        if (a.mode == 2) { mean = a.run_mean[n]; var = a.run_var[n]; }
        else {
            const int nb = B - a.b0;
            float s = 0.0f;
            for (int b = a.b0; b < B; ++b) s += a.y[(size_t)b * a.ldy + n];
            mean = s / (float)nb;
            float q = 0.0f;
            for (int b = a.b0; b < B; ++b) { const float d = a.y[(size_t)b * a.ldy + n] - mean; q = fmaf(d, d, q); }
            var = q / (float)nb;
            a.run_mean[n] = (1.0f - a.momentum) * a.run_mean[n] + a.momentum * mean;
            a.run_var[n] = (1.0f - a.momentum) * a.run_var[n] + a.momentum * (q / (float)max(nb - 1, 1));
            if (n == 0 && a.nbt) a.nbt[0] += 1;
        }
        const float rs = 1.0f / sqrtf(var + a.eps), g = a.gamma[n], bt = a.beta[n];
        for (int b = a.b0; b < B; ++b) {
            float o = fmaf((a.y[(size_t)b * a.ldy + n] - mean) * rs, g, bt);
            o = o > 0.0f ? o : 0.0f;
            if (a.mask) o *= a.mask[(size_t)b * a.ldmask + n];
            a.dst[pn_t16_off(b, a.kb0 * 16 + n, a.kb_stride)] = o;
        }
    }
}

// Backward of one normalised prenet layer of a decode step, in place: dn (rows, P) = gradient at the norm's output (the ReLU /
// mask backward already applied)  ->  gradient at the Linear's output y; dgamma / dbeta get this step's sums ADDED (the steps of a
// backward run one after the other on one stream, one thread owns a column: fixed order, no atomics).  Statistics are recomputed
// from y.  One thread per column; LayerNorm first leaves the four per-row scalars in LDS (every block for all rows).
struct PnbArgs {
    float* dn; int ld; const float* y; int ldy; int mode;
    const float* gamma; const float* run_mean; const float* run_var; float eps;
    float* dgamma; float* dbeta; int rows, P;
};

__global__ __launch_bounds__(256) void prenet_norm_bwd_kernel(const PnbArgs a) {
    extern __shared__ float pnb_lds[];                    // mode 1: (rows, 4) = mean, rstd, mean(g), mean(g * xhat)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = a.rows, P = a.P;
    const int n = blockIdx.x * 256 + tid;
    if (a.mode == 1) {
        for (int r = wave; r < R; r += 4) {
            const float* yr = a.y + (size_t)r * a.ldy;
            const float* dr = a.dn + (size_t)r * a.ld;
            float s = 0.0f;
            for (int k = lane; k < P; k += 64) s += yr[k];
            const float mean = st_wave_sum(s) / (float)P;
            float q = 0.0f;
            for (int k = lane; k < P; k += 64) { const float d = yr[k] - mean; q = fmaf(d, d, q); }
            const float rstd = 1.0f / sqrtf(st_wave_sum(q) / (float)P + a.eps);
            float c1 = 0.0f, c2 = 0.0f;
            for (int k = lane; k < P; k += 64) {
                const float g = dr[k] * a.gamma[k];
                c1 += g;
                c2 = fmaf(g, (yr[k] - mean) * rstd, c2);
            }
            c1 = st_wave_sum(c1) / (float)P;
            c2 = st_wave_sum(c2) / (float)P;
            if (lane == 0) { pnb_lds[r * 4 + 0] = mean; pnb_lds[r * 4 + 1] = rstd; pnb_lds[r * 4 + 2] = c1; pnb_lds[r * 4 + 3] = c2; }
        }
        __syncthreads();
        if (n >= P) return;
        const float g = a.gamma[n];
        float dg = 0.0f, db = 0.0f;
        for (int r = 0; r < R; ++r) {
            const float mean = pnb_lds[r * 4 + 0], rstd = pnb_lds[r * 4 + 1], c1 = pnb_lds[r * 4 + 2], c2 = pnb_lds[r * 4 + 3];
            const float xh = (a.y[(size_t)r * a.ldy + n] - mean) * rstd;
            const float d = a.dn[(size_t)r * a.ld + n];
            dg = fmaf(d, xh, dg);
            db += d;
            a.dn[(size_t)r * a.ld + n] = rstd * (d * g - c1 - xh * c2);
        }
        a.dgamma[n] += dg;
        a.dbeta[n] += db;
        return;
    }
    if (n >= P) return;
    float mean, var;
    if (a.mode == 2) { mean = a.run_mean[n]; var = a.run_var[n]; }
    else {
        float s = 0.0f;
        for (int r = 0; r < R; ++r) s += a.y[(size_t)r * a.ldy + n];
        mean = s / (float)R;
        float q = 0.0f;
        for (int r = 0; r < R; ++r) { const float d = a.y[(size_t)r * a.ldy + n] - mean; q = fmaf(d, d, q); }
        var = q / (float)R;
    }
    const float rs = 1.0f / sqrtf(var + a.eps), g = a.gamma[n];
    float s1 = 0.0f, s2 = 0.0f;
    for (int r = 0; r < R; ++r) {
        const float d = a.dn[(size_t)r * a.ld + n];
        s1 += d;
        s2 = fmaf(d, (a.y[(size_t)r * a.ldy + n] - mean) * rs, s2);
    }
    const float m1 = a.mode == 3 ? s1 / (float)R : 0.0f, m2 = a.mode == 3 ? s2 / (float)R : 0.0f;
    for (int r = 0; r < R; ++r) {
        const float xh = (a.y[(size_t)r * a.ldy + n] - mean) * rs;
        a.dn[(size_t)r * a.ld + n] = g * rs * (a.dn[(size_t)r * a.ld + n] - m1 - xh * m2);
    }
    a.dgamma[n] += s2;
    a.dbeta[n] += s1;
}

}  // namespace

extern "C" int st_prenet_norm_bwd(float* dn, int ld, const float* y, int ldy, int mode, const float* gamma, const float* run_mean,
                                  const float* run_var, float eps, float* dgamma, float* dbeta, int rows, int P, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dn && y && gamma && dgamma && dbeta && rows > 0 && P > 0 && ld >= P && ldy >= P && mode >= 1 && mode <= 3,
                 "st_prenet_norm_bwd: bad arguments");
    ST_CHECK_ARG(mode != 2 || (run_mean && run_var), "st_prenet_norm_bwd: eval-mode BatchNorm needs the running statistics");
    ST_CHECK_ARG(rows <= 2048, "st_prenet_norm_bwd: more rows than one decode step has");
    PnbArgs a;
    a.dn = dn; a.ld = ld; a.y = y; a.ldy = ldy; a.mode = mode; a.gamma = gamma; a.run_mean = run_mean; a.run_var = run_var; a.eps = eps;
    a.dgamma = dgamma; a.dbeta = dbeta; a.rows = rows; a.P = P;
    hipLaunchKernelGGL(prenet_norm_bwd_kernel, dim3((P + 255) / 256), dim3(256), mode == 1 ? (size_t)rows * 4 * sizeof(float) : 0,
                       (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_layer_norm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* y, int ldy,
                                 float* mean_out, float* rstd_out, int M, int N, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(x && y && M > 0 && N > 0 && ldx >= N && ldy >= N, "st_layer_norm_fwd: bad arguments");
    hipLaunchKernelGGL(layer_norm_fwd_kernel, dim3((M + NR_WAVES - 1) / NR_WAVES), dim3(NR_WAVES * 64), 0, (hipStream_t)stream,
                       x, ldx, gamma, beta, eps, y, ldy, mean_out, rstd_out, M, N);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_layer_norm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* mean,
                                 const float* rstd, float* dx, int lddx, float* dyxhat, int M, int N, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dy && x && mean && rstd && dx && M > 0 && N > 0 && lddy >= N && ldx >= N && lddx >= N, "st_layer_norm_bwd: bad arguments");
    hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3((M + NR_WAVES - 1) / NR_WAVES), dim3(NR_WAVES * 64), 0, (hipStream_t)stream,
                       dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dyxhat, M, N);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_log_softmax_fwd(const float* x, float* y, int M, int N, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(x && y && M > 0 && N > 0, "st_log_softmax_fwd: bad arguments");
    hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3((M + NR_WAVES - 1) / NR_WAVES), dim3(NR_WAVES * 64), 0, (hipStream_t)stream, x, y, M, N);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_log_softmax_bwd(const float* dy, const float* y, float* dx, int M, int N, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dy && y && dx && M > 0 && N > 0, "st_log_softmax_bwd: bad arguments");
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((M + NR_WAVES - 1) / NR_WAVES), dim3(NR_WAVES * 64), 0, (hipStream_t)stream, dy, y, dx, M, N);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_prenet_norm_fwd(const float* y, int ldy, int mode, const float* gamma, const float* beta, float* run_mean,
                                  float* run_var, long long* batches_tracked, float eps, float momentum, const float* mask, int ldmask,
                                  const st_t16_view* dst, int b0, int B, int P, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(y && gamma && beta && dst && dst->base && b0 >= 0 && B > b0 && P > 0 && ldy >= P && mode >= 1 && mode <= 3,
                 "st_prenet_norm_fwd: bad arguments");
    ST_CHECK_ARG(mode == 1 || (run_mean && run_var), "st_prenet_norm_fwd: BatchNorm needs the running statistics");
    ST_CHECK_ARG(dst->kb0 >= 0 && dst->kb0 + ((P + 15) >> 4) <= dst->kb_stride, "st_prenet_norm_fwd: k-block range outside the T16 buffer");
    PnArgs a;
    a.y = y; a.ldy = ldy; a.mode = mode; a.gamma = gamma; a.beta = beta; a.run_mean = run_mean; a.run_var = run_var;
    a.nbt = batches_tracked; a.eps = eps; a.momentum = momentum; a.mask = mask; a.ldmask = ldmask;
    a.dst = dst->base; a.kb_stride = dst->kb_stride; a.kb0 = dst->kb0; a.B = B; a.P = P; a.b0 = b0;
    const int blocks = mode == 1 ? (B - b0 + 3) / 4 : (P + 255) / 256;
    hipLaunchKernelGGL(prenet_norm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}
