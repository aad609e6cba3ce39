// attention.hip -- launches and C ABI of the fused attention step (device code: attention_body.h)
#include "attention_body.h"

namespace {

template <int PART>
int at_launch(const AtArgs& a, hipStream_t stream) {
    ST_CHECK_ARG(a.B > 0 && a.L > 0 && a.A > 0 && a.E > 0 && a.F > 0 && a.K > 0, "attention step: bad dims");
    ST_CHECK_ARG(a.K % 2 == 1, "attention step: location kernel size %d must be odd", a.K);
    ST_CHECK_ARG(a.E % 4 == 0 && a.E / 4 <= AT_THREADS, "attention step: E=%d must be a multiple of 4 and <= %d", a.E, 4 * AT_THREADS);
    ST_CHECK_ARG(PART == 1 || st_aligned16(a.memory), "attention step: memory must be 16-byte aligned");
    ST_CHECK_ARG((a.A % 4 != 0) || (st_aligned16(a.pm) && st_aligned16(a.pq) && st_aligned16(a.v)),
                 "attention step: pm/pq/v must be 16-byte aligned");
    ST_CHECK_ARG(PART == 1 || a.ctx || a.ctx_dst[0].base, "attention step: no context output");
    ST_CHECK_ARG(PART == 0 || (a.s_buf && ((a.A % 4 != 0) || st_aligned16(a.s_buf))), "attention step: S buffer missing / unaligned");
    ST_CHECK_ARG(!a.h_q || (a.ada_std && a.ada_mean && a.h_adapt), "attention step: AdaIN pointers");
    const bool vec = (a.A % 4 == 0) && (a.F % 4 == 0) && st_aligned16(a.pm) && (PART == 1 || (st_aligned16(a.pq) && st_aligned16(a.v))) &&
                     st_aligned16(a.loc_lin_w);
    const int pre_parts = PART == 1 && a.pre_parts > 1 ? a.pre_parts : 1;
    const AtLds o = at_layout(a.L, a.A, a.E, a.F, a.K, PART, at_pos_per(a.L, pre_parts), PART == 1 && vec && a.F == 32 && a.A % 16 == 0);
    const size_t lds_bytes = (size_t)o.total * sizeof(float);
    ST_CHECK_ARG(lds_bytes <= 160 * 1024, "attention step: L=%d needs %zu B of LDS (> 160 KiB)%s", a.L, lds_bytes,
                 PART == 0 ? "; the split form (st_attn_pre_fwd with more parts + st_attn_fin_t16_fwd) takes longer texts" : "");
    static bool configured = false;
    if (!configured) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(at_kernel<true, PART>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(at_kernel<false, PART>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    const int nwg = a.B * (PART == 1 && a.pre_parts > 1 ? a.pre_parts : PART == 2 && a.fin_parts > 1 ? a.fin_parts : 1);
    // (a 256-thread form of the fin part -- one wave per SIMD -- measured SLOWER: 11968 vs 10376 cycles; the serial stretches
    // are bound by dependent-issue latency, not by two waves sharing a SIMD)
    if (vec) hipLaunchKernelGGL((at_kernel<true, PART>), dim3(nwg), dim3(AT_THREADS), lds_bytes, stream, a.pq, a.pm, a.v, a.w_cum_prev, a.memory,
                                a.s_buf, a.L, a.A, a.E, a.fin_parts, a);
    else hipLaunchKernelGGL((at_kernel<false, PART>), dim3(nwg), dim3(AT_THREADS), lds_bytes, stream, a.pq, a.pm, a.v, a.w_cum_prev, a.memory,
                            a.s_buf, a.L, a.A, a.E, a.fin_parts, a);
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int st_attn_step_fwd(const float* pq, const float* pm, const float* memory,
                                const float* w_prev, int ld_wprev, const float* w_cum_prev,
                                float* w_out, int ld_wout, float* w_cum_out,
                                const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                float* ctx, int ld_ctx,
                                const float* h_q, int ld_hq, const float* ada_std, const float* ada_mean,
                                float* h_adapt, int Q,
                                int B, int L, int A, int E, int F, int K, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    AtArgs a;
    memset(&a, 0, sizeof(a));
    a.pq = pq; a.pm = pm; a.memory = memory; a.w_prev = w_prev; a.ld_wprev = ld_wprev; a.w_cum_prev = w_cum_prev;
    a.w_out = w_out; a.ld_wout = ld_wout; a.w_cum_out = w_cum_out;
    a.loc_conv_w = loc_conv_w; a.loc_lin_w = loc_lin_w; a.v = v; a.ctx = ctx; a.ld_ctx = ld_ctx;
    a.h_q = h_q; a.ld_hq = ld_hq; a.ada_std = ada_std; a.ada_mean = ada_mean; a.h_adapt = h_adapt; a.Q = Q;
    a.B = B; a.L = L; a.A = A; a.E = E; a.F = F; a.K = K;
    return at_launch<0>(a, (hipStream_t)stream);
}

// decode-loop variant: context written to up to 3 T16 destinations (and optionally natural)
extern "C" int st_attn_step_t16_fwd(const float* pq, const float* pm, const float* memory,
                                    const float* w_prev, int ld_wprev, const float* w_cum_prev,
                                    float* w_out, int ld_wout, float* w_cum_out,
                                    const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                    const st_t16_view* ctx_dst, int n_ctx_dst, float* ctx, int ld_ctx,
                                    int B, int L, int A, int E, int F, int K, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(n_ctx_dst >= 0 && n_ctx_dst <= 3 && (n_ctx_dst == 0 || ctx_dst), "st_attn_step_t16_fwd: n_ctx_dst=%d", n_ctx_dst);
    AtArgs a;
    memset(&a, 0, sizeof(a));
    a.pq = pq; a.pm = pm; a.memory = memory; a.w_prev = w_prev; a.ld_wprev = ld_wprev; a.w_cum_prev = w_cum_prev;
    a.w_out = w_out; a.ld_wout = ld_wout; a.w_cum_out = w_cum_out;
    a.loc_conv_w = loc_conv_w; a.loc_lin_w = loc_lin_w; a.v = v; a.ctx = ctx; a.ld_ctx = ld_ctx;
    for (int d = 0; d < n_ctx_dst; ++d) a.ctx_dst[d] = ctx_dst[d];
    a.B = B; a.L = L; a.A = A; a.E = E; a.F = F; a.K = K;
    return at_launch<0>(a, (hipStream_t)stream);
}

// The step in two parts (see attention_body.h): `pre` needs only the previous step's weights (w_prev, w_cum_prev) and writes
// S = pm + W_l conv(hist); `fin` needs the processed query and S.  pre(t+1) can run any time after fin(t).
extern "C" int st_attn_pre_fwd(const float* pm, const float* w_prev, int ld_wprev, const float* w_cum_prev,
                               const float* loc_conv_w, const float* loc_lin_w, float* s_buf, int parts,
                               int B, int L, int A, int F, int K, void* stream) {
    (void)hipGetLastError();
    AtArgs a;
    memset(&a, 0, sizeof(a));
    a.pm = pm; a.w_prev = w_prev; a.ld_wprev = ld_wprev; a.w_cum_prev = w_cum_prev;
    a.loc_conv_w = loc_conv_w; a.loc_lin_w = loc_lin_w; a.s_buf = s_buf;
    a.pre_parts = (parts >= 2 && parts <= 64 && (parts & (parts - 1)) == 0) ? parts : 1;
    a.B = B; a.L = L; a.A = A; a.E = 4; a.F = F; a.K = K;
    return at_launch<1>(a, (hipStream_t)stream);
}

extern "C" int st_attn_fin_t16_fwd(const float* pq, const float* s_buf, const float* memory, const float* w_cum_prev,
                                   float* w_out, int ld_wout, float* w_cum_out, const float* v,
                                   const st_t16_view* ctx_dst, int n_ctx_dst, float* ctx, int ld_ctx, int parts,
                                   int B, int L, int A, int E, int F, int K, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(n_ctx_dst >= 0 && n_ctx_dst <= 3 && (n_ctx_dst == 0 || ctx_dst), "st_attn_fin_t16_fwd: n_ctx_dst=%d", n_ctx_dst);
    ST_CHECK_ARG(parts == 1 || ((parts == 2 || parts == 4 || parts == 8) && E % (4 * parts) == 0),
                 "st_attn_fin_t16_fwd: parts=%d must be 1, 2, 4 or 8 with E=%d a multiple of 4*parts", parts, E);
    AtArgs a;
    memset(&a, 0, sizeof(a));
    a.pq = pq; a.pm = s_buf; a.s_buf = const_cast<float*>(s_buf); a.memory = memory; a.w_cum_prev = w_cum_prev;
    a.w_out = w_out; a.ld_wout = ld_wout; a.w_cum_out = w_cum_out; a.v = v; a.ctx = ctx; a.ld_ctx = ld_ctx;
    a.loc_lin_w = s_buf;     // (unused by this part; only its alignment is looked at)
    a.fin_parts = parts;
    for (int d = 0; d < n_ctx_dst; ++d) a.ctx_dst[d] = ctx_dst[d];
    a.B = B; a.L = L; a.A = A; a.E = E; a.F = F; a.K = K;
    return at_launch<2>(a, (hipStream_t)stream);
}
