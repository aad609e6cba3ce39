// attention_bwd.hip -- backward of one location-sensitive attention step for gfx950 (MI355X).
//
// Forward (ref: src/module.py:371-407, state update :262-264), per utterance:
//   loc[l][f]  = sum_{c,k} Wc[f][c][k] * hist[c][l + k - pad]          hist = [w_{t-1} ; cum_{t-1}]
//   s[l][a]    = pq[a] + pm[l][a] + sum_f Wl[a][f] * loc[l][f]
//   e[l]       = sum_a v[a] * tanh(s[l][a]);   w = softmax_l(e);   ctx = sum_l w[l] * mem[l][:]
//   cum_t      = cum_{t-1} + w
// One workgroup (512 threads) per utterance recomputes loc and tanh(s) from the saved attention
// weights (nothing of size L x A is kept from the forward) and produces
//   dpq (A), dhist (2, L) for step t-1, and accumulates over the steps, in slabs it owns (no atomics):
//   dpm[b] (L,A), dmem[b] (L,E), dv_part[b] (A), dWl_part[b] (A,F), dWc_part[b] (F,2,K).
// The per-utterance weight-gradient slabs are summed over b once after the loop (st_colsum).
#include "st_common.h"

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
    float* dpm; float* dmem;               // (B, L, A), (B, L, E) accumulated in place
    float* dv_part; float* dwl_part; float* dwc_part;   // (B, A), (B, A, F), (B, F, 2, K) accumulated in place
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
    o.dsb = p; p += AB_LBLK * A;
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

    // ---- P0: stage operands
    for (int i = tid; i < 2 * HL; i += AB_THREADS) {
        const int c = i / HL, j = i - c * HL, l = j - pad;
        float v = 0.0f;
        if (l >= 0 && l < L) {
            if (c == 0) v = a.w_prev ? a.w_prev[(size_t)b * a.ld_wprev + l] : 0.0f;
            else v = a.w_cum_prev[(size_t)b * L + l];
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
    }
    __syncthreads();

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
    }
    // ---- P2: dw[l] += dctx . mem[l];  dmem[l] += w[l] * dctx     (one wave per position, lanes over E)
    const float* memb = a.memory + (size_t)b * L * E;
    float* dmemb = a.dmem + (size_t)b * L * E;
    for (int l = wave; l < L; l += AB_THREADS / 64) {
        const float wl_ = ws[l];
        float acc = 0.0f;
        for (int e = lane; e < E; e += 64) {
            const float dc = dctx[e];
            acc = fmaf(dc, memb[(size_t)l * E + e], acc);
            dmemb[(size_t)l * E + e] += wl_ * dc;
        }
        acc = st_wave_sum(acc);
        if (lane == 0) dws[l] += acc;
    }
    __syncthreads();
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

    // ---- P3: energy gradient, blocks of AB_LBLK positions; thread = fixed attention dim a0, positions l0 + grp, ...
    const int a0 = tid % A, grp = tid / A, ngrp = AB_THREADS / A;    // host guarantees A <= 512 and 512 % A == 0
    const float pq_a = a.pq[(size_t)b * A + a0], v_a = a.v[a0];
    const float* pmb = a.pm + (size_t)b * L * A;
    float* dpmb = a.dpm + (size_t)b * L * A;
    float dv_acc = 0.0f, dpq_acc = 0.0f;
    float dwl_acc[AB_FMAX];
#pragma unroll
    for (int f = 0; f < AB_FMAX; ++f) dwl_acc[f] = 0.0f;
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
            dpmb[(size_t)l * A + a0] += ds;
            dsb[(l - l0) * A + a0] = ds;
#pragma unroll
            for (int f = 0; f < AB_FMAX; ++f) if (f < F) dwl_acc[f] = fmaf(ds, lr[f], dwl_acc[f]);
        }
        __syncthreads();
        // dloc[l][f] = sum_a Wl[a][f] * ds[l][a]
        for (int i = tid; i < (lend - l0) * F; i += AB_THREADS) {
            const int ll = i / F, f = i - ll * F;
            const float* dr = dsb + ll * A;
            float acc = 0.0f;
            for (int aa = 0; aa < A; ++aa) acc = fmaf(Wl[aa * WLD + f], dr[aa], acc);
            dloc[(l0 + ll) * F + f] = acc;
        }
        __syncthreads();
    }
    // ---- P4: fold the per-thread accumulators of the ngrp threads sharing an attention dim, group by group
    // (fixed order: deterministic), into the slabs this workgroup owns
    float* dvp = a.dv_part + (size_t)b * A;
    float* dwlp = a.dwl_part + ((size_t)b * A + a0) * F;
    float* dpqo = a.dpq + (size_t)b * A;
    for (int gturn = 0; gturn < ngrp; ++gturn) {
        if (grp == gturn) {
            dvp[a0] += dv_acc;
            dpqo[a0] = gturn == 0 ? dpq_acc : dpqo[a0] + dpq_acc;
#pragma unroll
            for (int f = 0; f < AB_FMAX; ++f) if (f < F) dwlp[f] += dwl_acc[f];
        }
        __syncthreads();
    }
    // ---- P5: conv backward
    float* dwcp = a.dwc_part + (size_t)b * F * 2 * K;
    for (int i = tid; i < F * 2 * K; i += AB_THREADS) {
        const int f = i / (2 * K), r = i - f * 2 * K, c = r / K, k = r - c * K;
        const float* hr = hist + c * HL + k;
        float acc = 0.0f;
        for (int l = 0; l < L; ++l) acc = fmaf(dloc[l * F + f], hr[l], acc);
        dwcp[i] += acc;
    }
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
}

}  // namespace

extern "C" int st_attn_step_bwd(const float* pq, const float* pm, const float* memory,
                                const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                                const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                const float* const* dctx, const int* ld_dctx, int n_dctx,
                                const float* const* dw_direct, const int* ld_dw, int n_dw,
                                float* dcum, const float* dcum_add, int ld_dcum_add,
                                float* dpq, float* dhist, float* dpm, float* dmem,
                                float* dv_part, float* dwl_part, float* dwc_part,
                                int B, int L, int A, int E, int F, int K, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(pq && pm && memory && w_cum_prev && w && loc_conv_w && loc_lin_w && v, "st_attn_step_bwd: null input");
    ST_CHECK_ARG(dpq && dhist && dpm && dmem && dv_part && dwl_part && dwc_part, "st_attn_step_bwd: null output");
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
    a.dpq = dpq; a.dhist = dhist; a.dpm = dpm; a.dmem = dmem; a.dv_part = dv_part; a.dwl_part = dwl_part; a.dwc_part = dwc_part;
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
