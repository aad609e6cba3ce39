// loss.hip -- the spectrogram loss of the trainer (ref: src/util.py:80-126, called bin/train_vqvae.py:219-223):
//   loss = w_all * crit(pred, label) + w_low * crit(pred[..., :n_low], label[..., :n_low])
//        + w_diff * crit(pred[:, 1:] - pred[:, :-1], label[:, 1:] - label[:, :-1])          crit = mean squared / absolute error
// One pass produces the loss value AND d loss / d pred (the loss is a leaf of the training graph, so its
// backward is this gradient times the incoming scalar).  The scalar is reduced in two deterministic stages.
#include "st_common.h"

namespace {

constexpr int FL_BLOCKS = 1024;

__device__ __forceinline__ float fl_val(float e, int l1) { return l1 ? fabsf(e) : e * e; }
__device__ __forceinline__ float fl_grad(float e, int l1) { return l1 ? (e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f)) : 2.0f * e; }

__global__ __launch_bounds__(256) void freq_loss_kernel(const float* pred, const float* label, float* dpred, float* partial,
                                                        int B, int T, int D, int n_low, float c_all, float c_low, float c_diff, int l1) {
    __shared__ float red[4];
    const size_t total = (size_t)B * T * D;
    float acc = 0.0f;
    // (row, channel) of the first element and of the stride once, then carried along: no 64-bit division per element
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    const unsigned rows_step = (unsigned)(step / D), c_step = (unsigned)(step % D);
    size_t row = i0 / D;
    unsigned c = (unsigned)(i0 - row * D);
    unsigned t = (unsigned)(row % T);
    const unsigned t_step = rows_step % (unsigned)T;
    for (size_t i = i0; i < total; i += step) {
        const float e = pred[i] - label[i];
        float v = c_all * fl_val(e, l1), g = c_all * fl_grad(e, l1);
        if ((int)c < n_low) { v += c_low * fl_val(e, l1); g += c_low * fl_grad(e, l1); }
        if (c_diff != 0.0f) {
            // delta[t] = e[t+1] - e[t] contributes for t = 0..T-2; e[t] appears in delta[t] (-) and delta[t-1] (+)
            if ((int)t + 1 < T) {
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
        c += c_step; t += t_step;
        if (c >= (unsigned)D) { c -= D; t += 1; }
        if (t >= (unsigned)T) t -= T;
        if (t >= (unsigned)T) t -= T;
    }
    acc = st_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// fixed-order sum of the block partials by one wave: lane j adds partials j, j + 64, ..., then the lanes' sums in a fixed tree
__global__ __launch_bounds__(64) void freq_loss_final_kernel(const float* partial, int n, float* loss) {
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = st_wave_sum(s);
    if (threadIdx.x == 0) *loss = s;
}

// y[i] = x[i] * (*s)
__global__ __launch_bounds__(256) void scale_by_kernel(const float* x, const float* s, float* y, size_t n) {
    const float f = *s;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] * f;
}

// ---- CTC loss (ref: torch.nn.CTCLoss() as called by compute_ctcloss, bin/train_vqvae.py:430-444) ----------------------------
//   lp(t, c) = log(prob(b, t, c) + eps);  targets = the non-zero tokens of text(b, :) in order;  every frame counts;  blank = 0
//   loss = mean_b [ nll_b / max(S_b, 1) ],  nll_b = -log sum over alignments  (the alpha recursion in log space)
// Per utterance one workgroup walks alpha forward over the T frames and another walks beta backward (both stored to a workspace); a third launch emits
// d loss / d prob with the formula ATen's CPU kernel uses (grad wrt lp = exp(lp) - exp(log sum_{s in c}(alpha beta) + nll - lp),
// chained through the log): one thread per state of the blank-extended target (2 S + 1 <= 256), lp values gathered per chunk of
// CTC_TC frames into LDS, ONE LDS barrier per frame (double-buffered state), every reduction in a fixed order.
constexpr int CTC_NT = 256, CTC_TC = 16;

__device__ __forceinline__ float ctc_lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// blank-extended target of utterance b and, per label, the chain of the states that carry it (thread 0; the others wait at the barrier)
__device__ __forceinline__ void ctc_setup(const int64_t* text, int b, int L, int V, int* lab, int* nxt, int* cls_first, int* S_out, int* bad_out) {
    const int tid = threadIdx.x;
    for (int c = tid; c < V; c += CTC_NT) cls_first[c] = -1;
    __syncthreads();
    if (tid == 0) {
        int S = 0, bad = 0;
        for (int i = 0; i < L; ++i) {
            int64_t tk = text[(size_t)b * L + i];
            if (tk < 0 || tk >= V) { bad = 1; tk = 0; }            // never index cls_first / prob out of range
            if (tk != 0) { lab[2 * S + 1] = (int)tk; ++S; }
        }
        *bad_out = bad;
        for (int s2 = 0; s2 <= 2 * S; s2 += 2) lab[s2] = 0;
        for (int s2 = 0; s2 <= 2 * S; ++s2) nxt[s2] = -1;
        for (int s2 = 2 * S - 1; s2 >= 1; s2 -= 2) {   // odd states, descending: cls_first ends up as the FIRST occurrence
            const int c = lab[s2];
            nxt[s2] = cls_first[c];
            cls_first[c] = s2;
        }
        *S_out = S;
    }
    __syncthreads();
}

// log_input != 0: `prob` already holds log-probabilities (ASRPostnet's log_softmax, compute_ctcloss(apply_log=False)); the gradient is
// then taken with respect to them.  A token outside [0, V) poisons the utterance's loss and gradient with NaN (torch raises).
// grid (B, 2): workgroup (b, 0) walks alpha forward (-> log_alpha, nll), workgroup (b, 1) walks beta backward (-> log_beta) AT THE SAME TIME
// on another compute unit -- the two recursions only meet in the gradient (ctc_grad_kernel); one workgroup doing both in turn took
// 295 us at T = 128.  grid (B, 1) when no gradient is wanted.
__global__ __launch_bounds__(CTC_NT) void ctc_loss_kernel(const float* prob, const int64_t* text, float eps, float* nll_out,
                                                          float* log_alpha, float* log_beta, int B, int T, int V, int L, int log_input) {
    extern __shared__ int ctc_dyn[];                 // cls_first[V]
    __shared__ int lab[CTC_NT], nxt[CTC_NT];
    __shared__ float st_a[2][CTC_NT];
    __shared__ float lpS[CTC_TC][CTC_NT];
    __shared__ int S_s, bad_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pb = prob + (size_t)b * T * V;
    ctc_setup(text, b, L, V, lab, nxt, ctc_dyn, &S_s, &bad_s);
    const int S = S_s, SP = 2 * S + 1;
    const bool on = tid < SP;
    const int my = on ? lab[tid] : 0;
    const bool skip_f = on && tid >= 2 && my != 0 && my != lab[tid - 2];            // alpha may come from s - 2
    const bool skip_b = on && tid + 2 < SP && lab[tid + 2] != 0 && lab[tid + 2] != my;   // beta may come from s + 2
    auto gather = [&](int t0, int nt, int dir) {      // lpS[i][s] = lp(t0 + dir * i, lab[s]) for i < nt
        for (int i = 0; i < nt; ++i) {
            const float pv = on ? pb[(size_t)(t0 + dir * i) * V + my] : 0.0f;
            lpS[i][tid] = on ? (log_input ? pv : logf(pv + eps)) : -INFINITY;
        }
    };
    int cur = 0;
    if (blockIdx.y == 0) {
        // ---- alpha
        float* la = log_alpha + (size_t)b * T * CTC_NT;
        for (int t0 = 0; t0 < T; t0 += CTC_TC) {
            const int nt = min(CTC_TC, T - t0);
            __syncthreads();
            gather(t0, nt, 1);
            __syncthreads();
            for (int i = 0; i < nt; ++i) {
                const int t = t0 + i;
                float a;
                if (t == 0) a = (tid < 2 && on) ? lpS[0][tid] : -INFINITY;
                else {
                    const float* pa = st_a[cur ^ 1];
                    const float a0 = on ? pa[tid] : -INFINITY, a1 = (on && tid >= 1) ? pa[tid - 1] : -INFINITY;
                    const float a2 = skip_f ? pa[tid - 2] : -INFINITY;
                    a = ctc_lse3(a0, a1, a2) + lpS[i][tid];
                    if (!on) a = -INFINITY;
                }
                st_a[cur][tid] = a;
                la[(size_t)t * CTC_NT + tid] = a;
                st_lds_barrier();
                cur ^= 1;
            }
        }
        if (tid == 0) {
            const float* pa = st_a[cur ^ 1];
            const float l1 = pa[SP - 1], l2 = SP >= 2 ? pa[SP - 2] : -INFINITY;
            const float m = fmaxf(l1, l2);
            float v = m == -INFINITY ? INFINITY : -(m + logf(expf(l1 - m) + expf(l2 - m)));
            if (bad_s) v = __builtin_nanf("");
            nll_out[b] = v;
        }
        return;
    }
    // ---- beta
    float* lb = log_beta + (size_t)b * T * CTC_NT;
    for (int t1 = T - 1; t1 >= 0; t1 -= CTC_TC) {
        const int nt = min(CTC_TC, t1 + 1);
        __syncthreads();
        gather(t1, nt, -1);
        __syncthreads();
        for (int i = 0; i < nt; ++i) {
            const int t = t1 - i;
            const float lps = lpS[i][tid];
            float bt;
            if (t == T - 1) bt = (on && tid >= SP - 2) ? lps : -INFINITY;
            else {
                const float* pbt = st_a[cur ^ 1];
                const float b0 = on ? pbt[tid] : -INFINITY, b1 = (on && tid + 1 < SP) ? pbt[tid + 1] : -INFINITY;
                const float b2 = skip_b ? pbt[tid + 2] : -INFINITY;
                bt = ctc_lse3(b0, b1, b2) + lps;
                if (!on) bt = -INFINITY;
            }
            st_a[cur][tid] = bt;
            lb[(size_t)t * CTC_NT + tid] = bt;
            st_lds_barrier();
            cur ^= 1;
        }
    }
}

// occupancies and the gradient: grid (B, ceil(T / CTC_TC)), a workgroup takes CTC_TC frames of one utterance -- every frame is independent
// once alpha and beta are known.  Same expressions and summation orders as when this was the tail of the beta walk.
__global__ __launch_bounds__(CTC_NT) void ctc_grad_kernel(const float* prob, const int64_t* text, float eps, const float* nll_in,
                                                          float* dprob, const float* log_alpha, const float* log_beta,
                                                          int B, int T, int V, int L, int log_input) {
    extern __shared__ int ctc_dyn[];                 // cls_first[V]
    __shared__ int lab[CTC_NT], nxt[CTC_NT];
    __shared__ float gam[2][CTC_NT];
    __shared__ int S_s, bad_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* pb = prob + (size_t)b * T * V;
    int* cls_first = ctc_dyn;
    ctc_setup(text, b, L, V, lab, nxt, cls_first, &S_s, &bad_s);
    const int S = S_s, SP = 2 * S + 1;
    const bool on = tid < SP;
    const int my = on ? lab[tid] : 0;
    const float nll = nll_in[b];
    const float gr = bad_s ? __builtin_nanf("") : 1.0f / ((float)max(S, 1) * (float)B);
    const float* la = log_alpha + (size_t)b * T * CTC_NT;
    const float* lb = log_beta + (size_t)b * T * CTC_NT;
    const int t0 = blockIdx.y * CTC_TC, t1 = min(T, t0 + CTC_TC);
    int cur = 0;
    for (int t = t0; t < t1; ++t) {
        const float pv = on ? pb[(size_t)t * V + my] : 0.0f;
        const float lps = on ? (log_input ? pv : logf(pv + eps)) : -INFINITY;
        const float ab = on ? la[(size_t)t * CTC_NT + tid] + lb[(size_t)t * CTC_NT + tid] : -INFINITY;
        gam[cur][tid] = ab == -INFINITY ? 0.0f : expf(ab - lps + nll);     // posterior occupancy of state s at frame t
        st_lds_barrier();
        const float* gq = gam[cur];
        float blank = 0.0f;
        if (tid < 64) {                          // blank = the even states: fixed-order wave reduction
            for (int s2 = 2 * lane; s2 < SP; s2 += 128) blank += gq[s2];
            blank = st_wave_sum_dpp(blank);
        }
        for (int c = tid; c < V; c += CTC_NT) {
            float occ = 0.0f;
            if (c == 0) occ = blank;
            else for (int s2 = cls_first[c]; s2 >= 0; s2 = nxt[s2]) occ += gq[s2];
            // d loss / d lp = gr * (exp(lp) - occ);  lp = log(p + eps)  ->  d / d prob = that / (p + eps)
            const float pin = pb[(size_t)t * V + c];
            const float p = log_input ? expf(pin) : pin + eps;
            dprob[((size_t)b * T + t) * V + c] = log_input ? gr * (p - occ) : gr * (p - occ) / p;
        }
        cur ^= 1;
    }
}

// loss = mean_b nll_b / max(S_b, 1): one wave, lane b counts its utterance's tokens (the lanes' loads run side by side; one thread walking
// all B x L tokens took 88 us at B = 32, L = 43), the sum over b in ascending order as before (bit-identical)
__global__ __launch_bounds__(64) void ctc_loss_final_kernel(const float* nll, const int64_t* text, int B, int L, float* loss) {
    __shared__ float term[64];
    float s = 0.0f;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + threadIdx.x;
        int S = 0;
        if (b < B) for (int i = 0; i < L; ++i) S += text[(size_t)b * L + i] != 0;
        term[threadIdx.x] = b < B ? nll[b] / (float)max(S, 1) : 0.0f;
        __syncthreads();
        if (threadIdx.x == 0) for (int k = 0; k < min(64, B - b0); ++k) s += term[k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = s / (float)B;
}

// the trainer's scalar arithmetic on loss values (total = sum_i w_i loss_i, plus the sums the log prints) as ONE launch:
// out[j] = sum_i W[j][i] * x_i, i in ascending order
struct ScArgs { const float* x[ST_SCALAR_MAX]; float w[4][ST_SCALAR_MAX]; float* out[4]; int n, m; };
__global__ void scalar_combine_kernel(ScArgs a) {
    const int j = threadIdx.x;
    if (j >= a.m) return;
    float s = 0.0f;
    for (int i = 0; i < a.n; ++i) if (j == 0 || a.w[j][i] != 0.0f) s += a.w[j][i] * a.x[i][0];      // (rows > 0 are sums over SUBSETS: a zero weight there means "not a member")
    a.out[j][0] = s;
}
// its backward: out[i] = w_i * dout
struct SfArgs { float w[ST_SCALAR_MAX]; int n; };
__global__ void scalar_fanout_kernel(const float* dout, SfArgs a, float* out) {
    const int i = threadIdx.x;
    if (i < a.n) out[i] = a.w[i] * dout[0];
}

}  // namespace

extern "C" int st_scalar_combine(const float* const* xs, int n, const float* W, int m, float* const* outs, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(xs && W && outs && n > 0 && n <= ST_SCALAR_MAX && m > 0 && m <= 4, "st_scalar_combine: 1..%d terms, 1..4 outputs", ST_SCALAR_MAX);
    ScArgs a{};
    a.n = n; a.m = m;
    for (int i = 0; i < n; ++i) { ST_CHECK_ARG(xs[i], "st_scalar_combine: null term"); a.x[i] = xs[i]; }
    for (int j = 0; j < m; ++j) {
        ST_CHECK_ARG(outs[j], "st_scalar_combine: null output");
        a.out[j] = outs[j];
        for (int i = 0; i < n; ++i) a.w[j][i] = W[j * n + i];
    }
    hipLaunchKernelGGL(scalar_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_scalar_fanout(const float* dout, const float* w, int n, float* out, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dout && w && out && n > 0 && n <= ST_SCALAR_MAX, "st_scalar_fanout: 1..%d terms", ST_SCALAR_MAX);
    SfArgs a{};
    a.n = n;
    for (int i = 0; i < n; ++i) a.w[i] = w[i];
    hipLaunchKernelGGL(scalar_fanout_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dout, a, out);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t st_ctc_workspace_floats(int B, int T) { return 2 * (size_t)B * T * CTC_NT + (size_t)B; }

extern "C" int st_ctc_loss(const float* prob, const int64_t* text, float eps, float* loss, float* dprob, float* ws,
                           int B, int T, int V, int L, int log_input, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(prob && text && loss && ws && B > 0 && T > 0 && V > 1 && L > 0, "st_ctc_loss: bad arguments");
    ST_CHECK_ARG(2 * L + 1 <= CTC_NT, "st_ctc_loss: transcripts of up to %d tokens (L=%d)", (CTC_NT - 1) / 2, L);
    // dynamic LDS (one int per class) on top of ~23 KiB of static LDS, within the 64 KiB a launch gets without raising
    // hipFuncAttributeMaxDynamicSharedMemorySize: the argument check, not the launch, is what fails for a huge codebook
    ST_CHECK_ARG((size_t)V * sizeof(int) <= 40 * 1024, "st_ctc_loss: V=%d too large (at most %d classes)", V, 40 * 1024 / 4);
    hipStream_t st = (hipStream_t)stream;
    float* lbeta = ws + (size_t)B * T * CTC_NT;
    float* nll = ws + 2 * (size_t)B * T * CTC_NT;
    hipLaunchKernelGGL(ctc_loss_kernel, dim3(B, dprob ? 2 : 1), dim3(CTC_NT), (size_t)V * sizeof(int), st, prob, text, eps, nll, ws, lbeta, B, T, V, L, log_input);
    ST_LAUNCH_CHECK();
    if (dprob) {
        hipLaunchKernelGGL(ctc_grad_kernel, dim3(B, (T + CTC_TC - 1) / CTC_TC), dim3(CTC_NT), (size_t)V * sizeof(int), st, prob, text, eps, nll, dprob, ws, lbeta,
                           B, T, V, L, log_input);
        ST_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ctc_loss_final_kernel, dim3(1), dim3(64), 0, st, nll, text, B, L, loss);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t st_freq_loss_workspace_floats(void) { return FL_BLOCKS; }

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
    hipLaunchKernelGGL(freq_loss_final_kernel, dim3(1), dim3(64), 0, st, ws, FL_BLOCKS, loss);
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
