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
// A workgroup takes four positions l0 .. l0 + 3 of one utterance and 256 context dims: 64 lanes x 4 dims, the steps dealt round-robin to
// the four waves (a wave's dctx reads are 16-byte loads, each feeding 16 multiply-adds; the four weights of a step are wave-uniform).  The
// four waves' sums meet in LDS in wave order: per output, four interleaved sums over t ascending, combined ((0 + 1) + (2 + 3)).
__global__ __launch_bounds__(256) void attn_dmem_kernel(const float* align, const float* dctx, float* dmem, int B, int steps, int L, int E) {
    __shared__ f32x4 part[3][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int LG = (L + 3) >> 2, EB = (E + 255) >> 8;
    int i = blockIdx.x;
    const int eb = i % EB; i /= EB;
    const int lg = i % LG, b = i / LG;
    const int l0 = lg * 4, e = eb * 256 + lane * 4;
    const bool eok = e < E;                                   // (E % 4 == 0: whole pieces)
    const int lc[4] = {l0, min(l0 + 1, L - 1), min(l0 + 2, L - 1), min(l0 + 3, L - 1)};
    const float* wp = align + (size_t)b * steps * L;
    const float* dp = dctx + (size_t)b * E + (eok ? e : 0);
    f32x4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int t = wave; t < steps; t += 4) {
        const f32x4 d = st_ld4(dp + (size_t)t * B * E);
        const float* wr = wp + (size_t)t * L;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float w = wr[lc[j]];
            a[j][0] = fmaf(w, d[0], a[j][0]); a[j][1] = fmaf(w, d[1], a[j][1]); a[j][2] = fmaf(w, d[2], a[j][2]); a[j][3] = fmaf(w, d[3], a[j][3]);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) part[wave - 1][j][lane] = a[j];
    }
    __syncthreads();
    if (wave == 0 && eok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (l0 + j >= L) break;
            const f32x4 p1 = part[0][j][lane], p2 = part[1][j][lane], p3 = part[2][j][lane];
            f32x4 r;
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = (a[j][c] + p1[c]) + (p2[c] + p3[c]);
            *reinterpret_cast<f32x4*>(dmem + ((size_t)b * L + l0 + j) * E + e) = r;
        }
    }
}

// (context widths that are not whole 16-byte pieces: one thread per output)
__global__ __launch_bounds__(256) void attn_dmem_scalar_kernel(const float* align, const float* dctx, float* dmem, int B, int steps, int L, int E) {
    const size_t total = (size_t)B * L * E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const size_t bl = i / E;
        const int l = (int)(bl % L), b = (int)(bl / L);
        float acc = 0.0f;
        for (int t = 0; t < steps; ++t) acc = fmaf(align[((size_t)b * steps + t) * L + l], dctx[((size_t)t * B + b) * E + e], acc);
        dmem[i] = acc;
    }
}

}  // namespace

extern "C" int st_attn_dmem(const float* align, const float* dctx_tape, float* dmem, int B, int steps, int L, int E, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(align && dctx_tape && dmem && B > 0 && steps > 0 && L > 0 && E > 0, "st_attn_dmem: bad arguments");
    if (E % 4 != 0 || !st_aligned16(dctx_tape) || !st_aligned16(dmem)) {
        size_t nb = ((size_t)B * L * E + 255) / 256;
        hipLaunchKernelGGL(attn_dmem_scalar_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, align, dctx_tape, dmem, B, steps, L, E);
        ST_LAUNCH_CHECK();
        return 0;
    }
    const size_t blocks = (size_t)B * ((L + 3) / 4) * ((E + 255) / 256);
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
