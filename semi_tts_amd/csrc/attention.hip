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

// ---- fin part split over POSITION ranges (long texts).  The E-split fin part above re-reads all of S in every workgroup and a
// workgroup's 2 x L x A x 4 bytes have to come through ONE compute unit (~30 GB/s): at L = 171 that is 350 KB = 12 us.  Here
// workgroup (b, p) takes the positions [p Lp, (p+1) Lp) of utterance b: energies from its S rows, a LOCAL softmax (maximum m_p,
// w~_l = exp(e_l - m_p), s_p = sum w~) and the un-normalised partial context c_p = sum_l w~_l memory_l -- S and the memory are read
// once in total.  at_combine_kernel then scales the partials by exp(m_p - M) / D (M = max m_p, D = sum s_p exp(m_p - M)): the
// same softmax, rounded differently (~1e-7).  workspace per (b, p): [m_p, s_p, pad, pad, c_p (E floats)].
struct AsArgs {
    const float* pq; const float* s_buf; const float* memory; const float* v;
    float* w_tmp; int ld_w;          // un-normalised weights w~ (the w_out buffer of the step)
    float* ws;                       // (B, P, 4 + E)
    int B, L, A, E, P, Lp;
};

__global__ __launch_bounds__(AT_THREADS) void at_split_kernel(const AsArgs a) {
    __shared__ float es[512];                           // energies, then w~ of this range (Lp <= 512)
    __shared__ __attribute__((aligned(16))) float part[4 * AT_THREADS];
    __shared__ float stat[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.P, p = blockIdx.x - b * a.P;
    const int L = a.L, A = a.A, E = a.E;
    const int l0 = p * a.Lp, l1 = min(L, l0 + a.Lp), n = l1 - l0;
    const float* sb = a.s_buf + ((size_t)b * L + l0) * A;
    const float* memb = a.memory + ((size_t)b * L + l0) * E;
    // context: thread (e4, g) owns 4 context dims and every (AT_THREADS / ne4)-th row; its first rows are requested now
    const int ne4 = E >> 2;
    const int ng = AT_THREADS / ne4, e4 = tid % ne4, g = tid / ne4;
    constexpr int PFR = 11;                 // (44 rows with E = 512: a whole range of <= 44 positions is in flight from the start)
    f32x4 mpf[PFR];
#pragma unroll
    for (int j = 0; j < PFR; ++j) {
        const int l = g + j * ng;
        mpf[j] = (g < ng && l < n) ? st_ld4(memb + (size_t)l * E + e4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // energies: a wave takes positions wave, wave + 8, ...; a lane 4 consecutive attention dims (A <= 256)
    const int a0 = min(lane * 4, A - 4);
    const bool a_on = lane * 4 < A;
    const f32x4 pq4 = st_ld4(a.pq + (size_t)b * A + a0), v4 = st_ld4(a.v + a0);
    const float vsum = (v4[0] + v4[1]) + (v4[2] + v4[3]);
    // the S rows of the wave's first NPW positions are requested at once (a load per position inside the loop is one exposed round
    // trip per position: 8.6 instead of the ~5 us the bytes take)
    constexpr int NPW = 8;
    f32x4 spf[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) spf[i] = st_ld4(sb + (size_t)min(wave + i * AT_WAVES, max(n - 1, 0)) * A + a0);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int l = wave + i * AT_WAVES;
        if (l >= n) break;
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((pq4[c] + spf[i][c]) * 2.885390081777927f));
            acc = fmaf(v4[c], r, acc);
        }
        float e = a_on ? fmaf(-2.0f, acc, vsum) : 0.0f;
        e = st_wave_sum_dpp(e);
        if (lane == 0) es[l] = e;
    }
    for (int l = wave + NPW * AT_WAVES; l < n; l += AT_WAVES) {
        const f32x4 s4 = st_ld4(sb + (size_t)l * A + a0);
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {      // v . tanh(x), tanh(x) = 1 - 2 / (1 + exp(2x)) as in at_body
            const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((pq4[c] + s4[c]) * 2.885390081777927f));
            acc = fmaf(v4[c], r, acc);
        }
        float e = a_on ? fmaf(-2.0f, acc, vsum) : 0.0f;
        e = st_wave_sum_dpp(e);
        if (lane == 0) es[l] = e;
    }
    __syncthreads();
    if (wave == 0) {      // local softmax statistics
        float m = -INFINITY;
        for (int l = lane; l < n; l += 64) m = fmaxf(m, es[l]);
        m = st_wave_max_dpp(m);
        float ssum = 0.0f;
        for (int l = lane; l < n; l += 64) {
            const float w = __expf(es[l] - m);
            es[l] = w;
            ssum += w;
            a.w_tmp[(size_t)b * a.ld_w + l0 + l] = w;
        }
        ssum = st_wave_sum_dpp(ssum);
        if (lane == 0) { stat[0] = m; stat[1] = ssum; }
    }
    __syncthreads();
    if (g < ng) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < PFR; ++j) {
            const int l = g + j * ng;
            const float w = l < n ? es[min(l, n - 1)] : 0.0f;
            acc[0] = fmaf(w, mpf[j][0], acc[0]); acc[1] = fmaf(w, mpf[j][1], acc[1]);
            acc[2] = fmaf(w, mpf[j][2], acc[2]); acc[3] = fmaf(w, mpf[j][3], acc[3]);
        }
        for (int l = g + PFR * ng; l < n; l += ng) {
            const f32x4 m4 = st_ld4(memb + (size_t)l * E + e4 * 4);
            const float w = es[l];
            acc[0] = fmaf(w, m4[0], acc[0]); acc[1] = fmaf(w, m4[1], acc[1]);
            acc[2] = fmaf(w, m4[2], acc[2]); acc[3] = fmaf(w, m4[3], acc[3]);
        }
        *reinterpret_cast<f32x4*>(part + (size_t)(g * ne4 + e4) * 4) = acc;
    }
    __syncthreads();
    float* wsb = a.ws + ((size_t)b * a.P + p) * (4 + E);
    if (tid == 0) { wsb[0] = stat[0]; wsb[1] = stat[1]; }
    for (int e = tid; e < E; e += AT_THREADS) {
        float s = 0.0f;
        for (int gg = 0; gg < ng; ++gg) s += part[gg * E + e];
        wsb[4 + e] = s;
    }
}

struct AcArgs {
    const float* ws; float* w; int ld_w; const float* w_cum_prev; float* w_cum_out;
    st_t16_view ctx_dst[3]; float* ctx; int ld_ctx;
    int B, L, E, P, Lp;
};

__global__ __launch_bounds__(AT_THREADS) void at_combine_kernel(const AcArgs a) {
    __shared__ float sc[64];            // per part: exp(m_p - M) / D
    const int tid = threadIdx.x, b = blockIdx.x;
    const int E = a.E, P = a.P;
    const float* wsb = a.ws + (size_t)b * P * (4 + E);
    // this thread's context dim of the first parts, requested before the statistics are looked at
    constexpr int CPF = 8;
    float cpf[CPF];
#pragma unroll
    for (int p = 0; p < CPF; ++p) cpf[p] = wsb[(size_t)min(p, P - 1) * (4 + E) + 4 + min(tid, E - 1)];
    if (tid < 64) {
        const float m = tid < P ? wsb[(size_t)tid * (4 + E)] : -INFINITY;
        const float s = tid < P ? wsb[(size_t)tid * (4 + E) + 1] : 0.0f;
        const float M = st_wave_max_dpp(m);
        const float t = tid < P ? s * __expf(m - M) : 0.0f;
        const float D = st_wave_sum_dpp(t);
        sc[tid] = tid < P ? __expf(m - M) / D : 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < E; e += AT_THREADS) {
        float s = 0.0f;
        if (e == tid) {
#pragma unroll
            for (int p = 0; p < CPF; ++p) s = p < P ? fmaf(sc[p], cpf[p], s) : s;
            for (int p = CPF; p < P; ++p) s = fmaf(sc[p], wsb[(size_t)p * (4 + E) + 4 + e], s);
        } else {
            for (int p = 0; p < P; ++p) s = fmaf(sc[p], wsb[(size_t)p * (4 + E) + 4 + e], s);
        }
        if (a.ctx) a.ctx[(size_t)b * a.ld_ctx + e] = s;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (a.ctx_dst[d].base) a.ctx_dst[d].base[at_t16_off(b, a.ctx_dst[d].kb0 * 16 + e, a.ctx_dst[d].kb_stride)] = s;
    }
    for (int l = tid; l < a.L; l += AT_THREADS) {
        const float w = a.w[(size_t)b * a.ld_w + l] * sc[l / a.Lp];
        a.w[(size_t)b * a.ld_w + l] = w;
        a.w_cum_out[(size_t)b * a.L + l] = w + a.w_cum_prev[(size_t)b * a.L + l];      // weights + attn_weights_sum, :264
    }
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

extern "C" size_t st_attn_fin_split_workspace_floats(int B, int E, int parts) { return (size_t)B * parts * (4 + E); }

// st_attn_fin_t16_fwd with the utterance split over `parts` POSITION ranges (2..64) + a combine launch: for long texts, where one
// compute unit cannot pull an utterance's S and memory rows fast enough (see at_split_kernel).  workspace:
// st_attn_fin_split_workspace_floats(B, E, parts) floats.
extern "C" int st_attn_fin_split_fwd(const float* pq, const float* s_buf, const float* memory, const float* w_cum_prev,
                                     float* w_out, int ld_wout, float* w_cum_out, const float* v,
                                     const st_t16_view* ctx_dst, int n_ctx_dst, float* ctx, int ld_ctx, float* workspace, int parts,
                                     int B, int L, int A, int E, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(pq && s_buf && memory && w_cum_prev && w_out && w_cum_out && v && workspace && B > 0 && L > 0, "st_attn_fin_split_fwd: bad arguments");
    ST_CHECK_ARG(n_ctx_dst >= 0 && n_ctx_dst <= 3 && (n_ctx_dst == 0 || ctx_dst) && (ctx || n_ctx_dst > 0), "st_attn_fin_split_fwd: context outputs");
    ST_CHECK_ARG(A % 4 == 0 && A >= 4 && A <= 256 && E % 4 == 0 && E / 4 <= AT_THREADS && AT_THREADS % (E / 4) == 0,
                 "st_attn_fin_split_fwd: A=%d must be a multiple of 4 up to 256, E=%d / 4 a divisor of %d", A, E, AT_THREADS);
    ST_CHECK_ARG(parts >= 2 && parts <= 64, "st_attn_fin_split_fwd: parts=%d (2..64)", parts);
    const int Lp = (L + parts - 1) / parts;
    ST_CHECK_ARG(Lp <= 512, "st_attn_fin_split_fwd: %d positions per part (> 512): use more parts", Lp);
    // ceil(L / parts) positions per part can leave trailing parts EMPTY (L = 9, parts = 4: 3 + 3 + 3 + 0); an empty part would
    // prefetch row L of its utterance -- the next utterance's, or one row past the end of s_buf for the last one.  Only the parts
    // that hold positions are launched and combined (the workspace is sized for the requested count, which is never smaller).
    parts = (L + Lp - 1) / Lp;
    ST_CHECK_ARG(st_aligned16(pq) && st_aligned16(s_buf) && st_aligned16(memory) && st_aligned16(v) && st_aligned16(workspace),
                 "st_attn_fin_split_fwd: operands must be 16-byte aligned");
    AsArgs s;
    memset(&s, 0, sizeof(s));
    s.pq = pq; s.s_buf = s_buf; s.memory = memory; s.v = v; s.w_tmp = w_out; s.ld_w = ld_wout; s.ws = workspace;
    s.B = B; s.L = L; s.A = A; s.E = E; s.P = parts; s.Lp = Lp;
    hipLaunchKernelGGL(at_split_kernel, dim3(B * parts), dim3(AT_THREADS), 0, (hipStream_t)stream, s);
    ST_LAUNCH_CHECK();
    AcArgs c;
    memset(&c, 0, sizeof(c));
    c.ws = workspace; c.w = w_out; c.ld_w = ld_wout; c.w_cum_prev = w_cum_prev; c.w_cum_out = w_cum_out; c.ctx = ctx; c.ld_ctx = ld_ctx;
    for (int d = 0; d < n_ctx_dst; ++d) c.ctx_dst[d] = ctx_dst[d];
    c.B = B; c.L = L; c.E = E; c.P = parts; c.Lp = Lp;
    hipLaunchKernelGGL(at_combine_kernel, dim3(B), dim3(AT_THREADS), 0, (hipStream_t)stream, c);
    ST_LAUNCH_CHECK();
    return 0;
}


// One wave runs the consumer side of the granule hand-off (at_wait_granules, the routine the fin workgroups of st_query_attn_fin_fwd
// wait in) on `granules` as they are: the A values, or NaN + status bit 0 after `max_spins` polls without every tag == epoch.
namespace {
__global__ __launch_bounds__(64) void handoff_wait_kernel(const unsigned long long* granules, unsigned epoch, int A, unsigned* status,
                                                          float* out, int max_spins) {
    const int lane = threadIdx.x;
    const f32x4 q4 = at_wait_granules(granules, epoch, lane, A, status, max_spins);
    if (lane * 4 < A) *reinterpret_cast<f32x4*>(out + lane * 4) = q4;
}
}  // namespace

extern "C" int st_handoff_wait_selftest(const unsigned long long* granules, unsigned epoch, int A, unsigned* status, float* out,
                                        int max_spins, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(granules && out && A > 0 && A % 4 == 0 && A <= 256 && max_spins > 0 && st_aligned16(out),
                 "st_handoff_wait_selftest: bad arguments");
    hipLaunchKernelGGL(handoff_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, granules, epoch, A, status, out, max_spins);
    ST_LAUNCH_CHECK();
    return 0;
}
