// attention_bwd.hip -- backward of one location-sensitive attention step for gfx950 (MI355X).
//
// Forward (ref: src/module.py:371-407, state update :262-264), per utterance:
//   loc[l][f]  = sum_{c,k} Wc[f][c][k] * hist[c][l + k - pad]          hist = [w_{t-1} ; cum_{t-1}]
//   s[l][a]    = pq[a] + pm[l][a] + sum_f Wl[a][f] * loc[l][f]
//   e[l]       = sum_a v[a] * tanh(s[l][a]);   w = softmax_l(e);   ctx = sum_l w[l] * mem[l][:]
//   cum_t      = cum_{t-1} + w
// One workgroup (512 threads) per utterance recomputes loc and tanh(s) from the saved attention weights
// (nothing of size L x A is kept from the forward) and produces dpq (A) and dhist (2, L) for step t-1, which the
// recurrence needs immediately.  Everything that is a SUM OVER STEPS (dpm, dmem, dv, dW_l, dW_c) is not
// accumulated here (read-modify-write chains on global memory serialise on latency: 178 us/step measured):
// the kernel writes per-step tape slices -- ds (L,A), loc / dloc (L,F), the channels-last history (L,2), the
// total dctx (E), dv_t (A) -- and the caller reduces them after the loop with a handful of large launches
// (column sums and TN GEMMs).
#include "st_common.h"

#ifndef AB_PROF
#define AB_PROF(n)   // phase timestamps, only defined by tools/mb/mb_attn_bwd.hip
#endif

namespace {

constexpr int AB_THREADS = 512;
constexpr int AB_LBLK = 16;     // positions per block of the energy-gradient phase
constexpr int AB_FMAX = 32;     // location filters held in registers per thread

struct AbArgs {
    const float* pq; const float* pm; const float* memory;
    const float* w_prev; int ld_wprev;     // w_{t-1} (B rows, stride ld_wprev); NULL = zeros (t = 0)
    const float* w_cum_prev;               // cum_{t-1} (B, L)
    const float* w; int ld_w;              // w_t
    const float* loc_conv_w; const float* loc_lin_w; const float* v;
    const float* dctx[3]; int ld_dctx[3];  // gradient w.r.t. ctx_t = sum of up to three addends (NULL = absent)
    const float* dw_direct[3]; int ld_dw[3];   // gradient w.r.t. w_t: up to three addends (B rows each)
    float* dcum; const float* dcum_add; int ld_dcum_add;   // dL/dcum_t = dcum (B,L, in/out) + dcum_add; also an addend of dw
    float* dpq;                            // (B, A) out
    float* dhist;                          // (B, 2, L) out: gradient w.r.t. [w_{t-1} ; cum_{t-1}] through the conv
    // tape slices of this step (all written, never read back here)
    float* ds_t;      // (B, L, A)  d loss / d s[l][a]              -> dpm = sum_t, dW_l = ds^T loc
    float* loc_t;     // (B, L, F)  location features               -> dW_l
    float* dloc_t;    // (B, L, F)  gradient of the location features -> dW_c (conv weight gradient)
    float* hist_t;    // (B, L, 2)  [w_{t-1}, cum_{t-1}] channels-last  -> dW_c
    float* dctx_t;    // (B, E)     total gradient w.r.t. ctx_t      -> dmem[b] = w^T dctx
    float* dv_t;      // (B, A)     sum_l de[l] * tanh(s[l][a])      -> dv = sum over (t, b)
    int B, L, A, E, F, K;
};

struct AbLds { int hist, hl, wc, wl, wl_ld, loc, dloc, w, dw, dctx, dsb, red, total; };

__host__ __device__ inline AbLds ab_layout(int L, int A, int E, int F, int K) {
    AbLds o;
    int p = 0;
    o.hl = L + K;                         // pad zeros both sides
    o.hist = p; p += 2 * o.hl;
    o.wc = p; p += F * 2 * K;
    o.wl_ld = F | 1;                      // odd row stride: lanes with different a hit different banks
    o.wl = p; p += A * o.wl_ld;
    o.loc = p; p += L * F;
    o.dloc = p; p += L * F;
    o.w = p; p += L;
    o.dw = p; p += L;
    o.dctx = p; p += E;
    o.dsb = p; p += (AB_LBLK * A > 2 * AB_THREADS ? AB_LBLK * A : 2 * AB_THREADS);   // also the fold buffer of P4
    o.red = p; p += 16;
    o.total = p;
    return o;
}

__global__ __launch_bounds__(AB_THREADS) void ab_kernel(const AbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int L = a.L, A = a.A, E = a.E, F = a.F, K = a.K;
    const AbLds o = ab_layout(L, A, E, F, K);
    float* hist = lds + o.hist; float* Wc = lds + o.wc; float* Wl = lds + o.wl; float* loc = lds + o.loc;
    float* dloc = lds + o.dloc; float* ws = lds + o.w; float* dws = lds + o.dw; float* dctx = lds + o.dctx;
    float* dsb = lds + o.dsb; float* red = lds + o.red;
    const int pad = (K - 1) / 2, HL = o.hl, WLD = o.wl_ld;

    AB_PROF(0);
    // ---- P0: stage operands
    for (int i = tid; i < 2 * HL; i += AB_THREADS) {
        const int c = i / HL, j = i - c * HL, l = j - pad;
        float v = 0.0f;
        if (l >= 0 && l < L) {
            if (c == 0) v = a.w_prev ? a.w_prev[(size_t)b * a.ld_wprev + l] : 0.0f;
            else v = a.w_cum_prev[(size_t)b * L + l];
            a.hist_t[((size_t)b * L + l) * 2 + c] = v;
        }
        hist[i] = v;
    }
    for (int i = tid; i < F * 2 * K; i += AB_THREADS) Wc[i] = a.loc_conv_w[i];
    for (int i = tid; i < A * F; i += AB_THREADS) { const int aa = i / F, f = i - aa * F; Wl[aa * WLD + f] = a.loc_lin_w[i]; }
    for (int l = tid; l < L; l += AB_THREADS) {
        ws[l] = a.w[(size_t)b * a.ld_w + l];
        float g = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) if (a.dw_direct[j]) g += a.dw_direct[j][(size_t)b * a.ld_dw[j] + l];
        if (a.dcum) {   // cum_t = cum_{t-1} + w_t: the total gradient w.r.t. cum_t reaches w_t and is carried to cum_{t-1}
            float gc = a.dcum[(size_t)b * L + l];
            if (a.dcum_add) gc += a.dcum_add[(size_t)b * a.ld_dcum_add + l];
            a.dcum[(size_t)b * L + l] = gc;
            g += gc;
        }
        dws[l] = g;
    }
    for (int e = tid; e < E; e += AB_THREADS) {
        float g = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) if (a.dctx[j]) g += a.dctx[j][(size_t)b * a.ld_dctx[j] + e];
        dctx[e] = g;
        a.dctx_t[(size_t)b * E + e] = g;
    }
    __syncthreads();

    AB_PROF(1);
    // ---- P1: location features  loc[l][f]
    for (int i = tid; i < L * F; i += AB_THREADS) {
        const int l = i / F, f = i - l * F;
        float acc = 0.0f;
        for (int c = 0; c < 2; ++c) {
            const float* wr = Wc + (f * 2 + c) * K;
            const float* hr = hist + c * HL + l;
            for (int k = 0; k < K; ++k) acc = fmaf(wr[k], hr[k], acc);
        }
        loc[i] = acc;
        a.loc_t[(size_t)b * L * F + i] = acc;
    }
    AB_PROF(2);
    // ---- P2: dw[l] += dctx . mem[l]     (one wave per position, lanes over E; independent coalesced loads)
    const float* __restrict__ memb = a.memory + (size_t)b * L * E;
    for (int l = wave; l < L; l += AB_THREADS / 64) {
        float acc = 0.0f;
        for (int e = lane; e < E; e += 64) acc = fmaf(dctx[e], memb[(size_t)l * E + e], acc);
        acc = st_wave_sum(acc);
        if (lane == 0) dws[l] += acc;
    }
    __syncthreads();
    AB_PROF(3);
    // softmax backward: de[l] = w[l] * (dw[l] - sum_j w[j] dw[j])
    if (wave == 0) {
        float acc = 0.0f;
        for (int l = lane; l < L; l += 64) acc = fmaf(ws[l], dws[l], acc);
        acc = st_wave_sum(acc);
        if (lane == 0) red[0] = acc;
    }
    __syncthreads();
    const float dot = red[0];
    __syncthreads();
    for (int l = tid; l < L; l += AB_THREADS) dws[l] = ws[l] * (dws[l] - dot);     // dws now holds de
    __syncthreads();

    AB_PROF(4);
    // ---- P3: energy gradient, blocks of AB_LBLK positions; thread = fixed attention dim a0, positions l0 + grp, ...
    const int a0 = tid % A, grp = tid / A, ngrp = AB_THREADS / A;    // host guarantees A <= 512 and 512 % A == 0
    const float pq_a = a.pq[(size_t)b * A + a0], v_a = a.v[a0];
    const float* __restrict__ pmb = a.pm + (size_t)b * L * A;
    float* __restrict__ dsg = a.ds_t + (size_t)b * L * A;
    float* __restrict__ dlocg = a.dloc_t + (size_t)b * L * F;
    float dv_acc = 0.0f, dpq_acc = 0.0f;
    float wl_r[AB_FMAX];
#pragma unroll
    for (int f = 0; f < AB_FMAX; ++f) wl_r[f] = f < F ? Wl[a0 * WLD + f] : 0.0f;

    for (int l0 = 0; l0 < L; l0 += AB_LBLK) {
        const int lend = min(L, l0 + AB_LBLK);
        for (int l = l0 + grp; l < lend; l += ngrp) {
            const float* lr = loc + l * F;
            float s = pq_a + pmb[(size_t)l * A + a0];
#pragma unroll
            for (int f = 0; f < AB_FMAX; ++f) if (f < F) s = fmaf(wl_r[f], lr[f], s);
            const float th = tanhf(s);
            const float g = dws[l];
            dv_acc = fmaf(g, th, dv_acc);
            const float ds = g * v_a * (1.0f - th * th);
            dpq_acc += ds;
            dsg[(size_t)l * A + a0] = ds;
            dsb[(l - l0) * A + a0] = ds;
        }
        __syncthreads();
        // dloc[l][f] = sum_a Wl[a][f] * ds[l][a]
        for (int i = tid; i < (lend - l0) * F; i += AB_THREADS) {
            const int ll = i / F, f = i - ll * F;
            const float* dr = dsb + ll * A;
            float acc = 0.0f;
            for (int aa = 0; aa < A; ++aa) acc = fmaf(Wl[aa * WLD + f], dr[aa], acc);
            dloc[(l0 + ll) * F + f] = acc;
            dlocg[(l0 + ll) * F + f] = acc;
        }
        __syncthreads();
    }
    AB_PROF(5);
    // ---- P4: fold the per-thread sums of the ngrp threads sharing an attention dim (LDS, fixed order)
    float* fold = dsb;                       // reuse: [2][AB_THREADS]   sized in ab_layout
    fold[tid] = dv_acc; fold[AB_THREADS + tid] = dpq_acc;
    __syncthreads();
    if (grp == 0) {
        float sv = 0.0f, sp = 0.0f;
        for (int gq = 0; gq < ngrp; ++gq) { sv += fold[gq * A + a0]; sp += fold[AB_THREADS + gq * A + a0]; }
        a.dv_t[(size_t)b * A + a0] = sv;
        a.dpq[(size_t)b * A + a0] = sp;
    }
    AB_PROF(6);
    // ---- P5: gradient w.r.t. the attention history through the location conv
    // dhist[c][j] = sum_{f,k} dloc[j - k + pad][f] * Wc[f][c][k]; 4 adjacent lanes split the filters
    float* dh = a.dhist + (size_t)b * 2 * L;
    for (int base = 0; base < 2 * L; base += AB_THREADS / 4) {
        const int cj = base + (tid >> 2), q = tid & 3;
        float acc = 0.0f;
        if (cj < 2 * L) {
            const int c = cj / L, j = cj - c * L;
            for (int f = q; f < F; f += 4) {
                const float* wr = Wc + (f * 2 + c) * K;
                for (int k = 0; k < K; ++k) {
                    const int l = j - k + pad;
                    if (l >= 0 && l < L) acc = fmaf(dloc[l * F + f], wr[k], acc);
                }
            }
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (cj < 2 * L && q == 0) dh[cj] = acc;
    }
    AB_PROF(7);
}

// dmem[b][l][e] = sum_t w_t[b][l] * dctx_t[b][e]      (the context is ctx_t = sum_l w_t[l] mem[l])
__global__ __launch_bounds__(256) void attn_dmem_kernel(const float* align, const float* dctx, float* dmem, int B, int steps, int L, int E) {
    const size_t total = (size_t)B * L * E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const size_t bl = i / E;
        const int l = (int)(bl % L), b = (int)(bl / L);
        const float* wp = align + (size_t)b * steps * L + l;
        const float* dp = dctx + (size_t)b * E + e;
        float acc = 0.0f;
        for (int t = 0; t < steps; ++t) acc = fmaf(wp[(size_t)t * L], dp[(size_t)t * B * E], acc);
        dmem[i] = acc;
    }
}

}  // namespace

extern "C" int st_attn_dmem(const float* align, const float* dctx_tape, float* dmem, int B, int steps, int L, int E, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(align && dctx_tape && dmem && B > 0 && steps > 0 && L > 0 && E > 0, "st_attn_dmem: bad arguments");
    size_t blocks = ((size_t)B * L * E + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(attn_dmem_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, align, dctx_tape, dmem, B, steps, L, E);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_attn_step_bwd(const float* pq, const float* pm, const float* memory,
                                const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                                const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                const float* const* dctx, const int* ld_dctx, int n_dctx,
                                const float* const* dw_direct, const int* ld_dw, int n_dw,
                                float* dcum, const float* dcum_add, int ld_dcum_add,
                                float* dpq, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                                float* dctx_t, float* dv_t,
                                int B, int L, int A, int E, int F, int K, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(pq && pm && memory && w_cum_prev && w && loc_conv_w && loc_lin_w && v, "st_attn_step_bwd: null input");
    ST_CHECK_ARG(dpq && dhist && ds_t && loc_t && dloc_t && hist_t && dctx_t && dv_t, "st_attn_step_bwd: null output");
    ST_CHECK_ARG(B > 0 && L > 0 && A > 0 && E > 0 && F > 0 && K > 0 && (K & 1), "st_attn_step_bwd: bad dims (K must be odd)");
    ST_CHECK_ARG(A <= AB_THREADS && AB_THREADS % A == 0, "st_attn_step_bwd: attn_dim=%d must divide %d", A, AB_THREADS);
    ST_CHECK_ARG(F <= AB_FMAX, "st_attn_step_bwd: n_location_filters=%d > %d", F, AB_FMAX);
    ST_CHECK_ARG(n_dctx >= 0 && n_dctx <= 3 && n_dw >= 0 && n_dw <= 3, "st_attn_step_bwd: at most 3 addends");
    AbArgs a;
    memset(&a, 0, sizeof(a));
    a.pq = pq; a.pm = pm; a.memory = memory; a.w_prev = w_prev; a.ld_wprev = ld_wprev; a.w_cum_prev = w_cum_prev;
    a.w = w; a.ld_w = ld_w; a.loc_conv_w = loc_conv_w; a.loc_lin_w = loc_lin_w; a.v = v;
    for (int j = 0; j < n_dctx; ++j) { a.dctx[j] = dctx[j]; a.ld_dctx[j] = ld_dctx[j]; }
    for (int j = 0; j < n_dw; ++j) { a.dw_direct[j] = dw_direct[j]; a.ld_dw[j] = ld_dw[j]; }
    a.dcum = dcum; a.dcum_add = dcum_add; a.ld_dcum_add = ld_dcum_add;
    a.dpq = dpq; a.dhist = dhist; a.ds_t = ds_t; a.loc_t = loc_t; a.dloc_t = dloc_t; a.hist_t = hist_t; a.dctx_t = dctx_t; a.dv_t = dv_t;
    a.B = B; a.L = L; a.A = A; a.E = E; a.F = F; a.K = K;
    const size_t lds = (size_t)ab_layout(L, A, E, F, K).total * sizeof(float);
    ST_CHECK_ARG(lds <= 160 * 1024, "st_attn_step_bwd: L=%d needs %zu bytes of LDS (> 160 KiB)", L, lds);
    static size_t lds_enabled = 0;
    if (lds > 64 * 1024 && lds > lds_enabled) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ab_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_enabled = lds;
    }
    hipLaunchKernelGGL(ab_kernel, dim3(B), dim3(AB_THREADS), lds, (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}
