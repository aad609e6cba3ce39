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
#include "attention_bwd_body.h"

namespace {

template <bool HAS_S, int LBLK>
__global__ __launch_bounds__(AB_THREADS) void ab_kernel(const AbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ab_dyn_lds[];
    ab_body<HAS_S, LBLK>(a, blockIdx.x, ab_dyn_lds);
}

// dmem[b][l][e] = sum_t w_t[b][l] * dctx_t[b][e]      (the context is ctx_t = sum_l w_t[l] mem[l])
__global__ __launch_bounds__(256) void attn_dmem_kernel(const float* align, const float* dctx, float* dmem, int B, int steps, int L, int E) {
    // a thread takes FOUR positions of one context dim: each dctx value (the bulk of the reads: steps x B x E, re-read by every position)
    // is loaded once per four outputs.  Per output the sum runs over t ascending, as before.
    const int LG = (L + 3) >> 2;
    const size_t total = (size_t)B * LG * E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const size_t bg = i / E;
        const int lg = (int)(bg % LG), b = (int)(bg / LG);
        const int l0 = lg * 4;
        const float* wp = align + (size_t)b * steps * L;
        const float* dp = dctx + (size_t)b * E + e;
        const int l1 = min(l0 + 1, L - 1), l2 = min(l0 + 2, L - 1), l3 = min(l0 + 3, L - 1);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll 8
        for (int t = 0; t < steps; ++t) {      // (unrolled: eight steps' loads in flight -- as a rolled loop it is a chain of dependent round trips)
            const float d = dp[(size_t)t * B * E];
            const float* wr = wp + (size_t)t * L;
            a0 = fmaf(wr[l0], d, a0); a1 = fmaf(wr[l1], d, a1); a2 = fmaf(wr[l2], d, a2); a3 = fmaf(wr[l3], d, a3);
        }
        float* o = dmem + ((size_t)b * L + l0) * E + e;
        o[0] = a0;
        if (l0 + 1 < L) o[(size_t)E] = a1;
        if (l0 + 2 < L) o[(size_t)2 * E] = a2;
        if (l0 + 3 < L) o[(size_t)3 * E] = a3;
    }
}

}  // namespace

extern "C" int st_attn_dmem(const float* align, const float* dctx_tape, float* dmem, int B, int steps, int L, int E, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(align && dctx_tape && dmem && B > 0 && steps > 0 && L > 0 && E > 0, "st_attn_dmem: bad arguments");
    size_t blocks = ((size_t)B * ((L + 3) / 4) * E + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(attn_dmem_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, align, dctx_tape, dmem, B, steps, L, E);
    ST_LAUNCH_CHECK();
    return 0;
}

static int ab_step_impl(const st_t16_view* dpq_t16, const float* pq, const float* pm, const float* memory,
                                const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                                const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                const float* const* dctx, const int* ld_dctx, int n_dctx,
                                const float* const* dw_direct, const int* ld_dw, int n_dw,
                                float* dcum, const float* dcum_add, int ld_dcum_add,
                                float* dpq, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                                float* dctx_t, float* dv_t, const float* s_in,
                                int B, int L, int A, int E, int F, int K, void* stream) {
    (void)hipGetLastError();
    AbArgs a;
    if (ab_fill(a, dpq_t16, pq, pm, memory, w_prev, ld_wprev, w_cum_prev, w, ld_w, loc_conv_w, loc_lin_w, v, dctx, ld_dctx, n_dctx,
                dw_direct, ld_dw, n_dw, dcum, dcum_add, ld_dcum_add, dpq, dhist, ds_t, loc_t, dloc_t, hist_t, dctx_t, dv_t, s_in,
                B, L, A, E, F, K)) return -1;
    // the wide block when the forward kept S and its LDS image fits; two workgroups of it never share a compute unit anyway (B workgroups)
    const bool wide = ab_wide(a);
    const size_t lds = ab_lds_bytes(a, wide);
    ST_CHECK_ARG(lds <= 160 * 1024, "st_attn_step_bwd: L=%d needs %zu bytes of LDS (> 160 KiB)", L, lds);
    static size_t lds_enabled = 0;
    if (lds > 64 * 1024 && lds > lds_enabled) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ab_kernel<false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ab_kernel<true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ab_kernel<true, AB_LBLK_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_enabled = lds;
    }
    if (wide) hipLaunchKernelGGL((ab_kernel<true, AB_LBLK_MAX>), dim3(B), dim3(AB_THREADS), lds, (hipStream_t)stream, a);
    else if (s_in) hipLaunchKernelGGL((ab_kernel<true, 16>), dim3(B), dim3(AB_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((ab_kernel<false, 16>), dim3(B), dim3(AB_THREADS), lds, (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}

// does the wide (48-position) block of the attention backward fit the LDS next to a hosting product's 8 KB? (what the split form needs)
extern "C" int st_attn_bwd_wide_fits(int L, int A, int E, int F, int K) {
    if (L <= 0 || A <= 0 || E <= 0 || F <= 0 || K <= 0) return 0;
    return (size_t)ab_layout(L, A, E, F, K, AB_LBLK_MAX, true).total * sizeof(float) + 8 * 64 * sizeof(f32x4) <= 160 * 1024 ? 1 : 0;
}

extern "C" int st_attn_step_bwd_s(const float* pq, const float* pm, const float* memory,
                                  const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                                  const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                  const float* const* dctx, const int* ld_dctx, int n_dctx,
                                  const float* const* dw_direct, const int* ld_dw, int n_dw,
                                  float* dcum, const float* dcum_add, int ld_dcum_add,
                                  float* dpq, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                                  float* dctx_t, float* dv_t, const float* s_in,
                                  int B, int L, int A, int E, int F, int K, void* stream) {
    return ab_step_impl(nullptr, pq, pm, memory, w_prev, ld_wprev, w_cum_prev, w, ld_w, loc_conv_w, loc_lin_w, v, dctx, ld_dctx, n_dctx,
                        dw_direct, ld_dw, n_dw, dcum, dcum_add, ld_dcum_add, dpq, dhist, ds_t, loc_t, dloc_t, hist_t, dctx_t, dv_t, s_in,
                        B, L, A, E, F, K, stream);
}

// the same with a second copy of dpq in the T16 tile layout (the x operand of the packed W_q^T product that follows in the BPTT loop)
extern "C" int st_attn_step_bwd_t16(const float* pq, const float* pm, const float* memory,
                                    const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                                    const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                    const float* const* dctx, const int* ld_dctx, int n_dctx,
                                    const float* const* dw_direct, const int* ld_dw, int n_dw,
                                    float* dcum, const float* dcum_add, int ld_dcum_add,
                                    float* dpq, const st_t16_view* dpq_t16, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                                    float* dctx_t, float* dv_t, const float* s_in,
                                    int B, int L, int A, int E, int F, int K, void* stream) {
    return ab_step_impl(dpq_t16, pq, pm, memory, w_prev, ld_wprev, w_cum_prev, w, ld_w, loc_conv_w, loc_lin_w, v, dctx, ld_dctx, n_dctx,
                        dw_direct, ld_dw, n_dw, dcum, dcum_add, ld_dcum_add, dpq, dhist, ds_t, loc_t, dloc_t, hist_t, dctx_t, dv_t, s_in,
                        B, L, A, E, F, K, stream);
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
    return st_attn_step_bwd_s(pq, pm, memory, w_prev, ld_wprev, w_cum_prev, w, ld_w, loc_conv_w, loc_lin_w, v, dctx, ld_dctx, n_dctx,
                              dw_direct, ld_dw, n_dw, dcum, dcum_add, ld_dcum_add, dpq, dhist, ds_t, loc_t, dloc_t, hist_t, dctx_t, dv_t,
                              nullptr, B, L, A, E, F, K, stream);
}
