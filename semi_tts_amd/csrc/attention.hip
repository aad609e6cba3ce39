// attention.hip -- fused location-sensitive attention step for gfx950 (MI355X).
//
// One workgroup (8 wavefronts) per utterance does the whole of Attention.forward for one
// decode step (ref: src/module.py:371-407 + state update :262-264 + AdaIN :267-269):
//   P0  stage loc_linear^T (F x A), loc_conv (F x 2 x K) and the zero-padded attention
//       history (w_prev, w_cum) in LDS; issue the coalesced 16-byte loads of the encoder
//       memory rows this thread will need for the context (they land while P1-P3 run)
//   P1  location conv  cf[l][f]   (2 -> F channels, K taps)            -- LDS only
//   P2  energies       e[l] = v . tanh(pq + W_l cf[l] + pm[l])         -- one wave per l,
//       each lane owns 4 consecutive attention dims, processed-memory row read as one
//       coalesced 1 KiB wave load, wave shuffle reduction over the attention dim
//   P3  softmax over L by wave 0 (shuffle max / sum), alignment + cumulative weights out
//   P4  context = sum_l w[l] * memory[l][:]   (4 row-groups, LDS cross-group reduction)
// HBM/L2 traffic per step and utterance: pm (L*A*4) + memory (L*E*4) bytes, read exactly once.
#include "st_common.h"

namespace {

constexpr int AT_THREADS = 512;
constexpr int AT_PF = 12;  // encoder-memory rows prefetched into registers per thread

struct AtArgs {
    const float* pq; const float* pm; const float* memory;
    const float* w_prev; int ld_wprev; const float* w_cum_prev;
    float* w_out; int ld_wout; float* w_cum_out;
    const float* loc_conv_w; const float* loc_lin_w; const float* v;
    float* ctx; int ld_ctx;
    const float* h_q; int ld_hq; const float* ada_std; const float* ada_mean; float* h_adapt; int Q;
    int B, L, A, E, F, K;
};

struct AtLds {  // offsets in floats into dynamic LDS
    int wt, wc, hs, cf, e, part, total;
};

__host__ __device__ inline AtLds at_layout(int L, int A, int E, int F, int K) {
    AtLds o;
    int p = 0;
    const int A4 = (A + 3) & ~3;
    o.wt = p; p += F * A4;                 // loc_linear transposed: [f][a]
    o.wc = p; p += ((F * 2 * K + 3) & ~3);  // loc_conv [f][c][k]
    const int hl = L + K - 1;
    o.hs = p; p += ((2 * hl + 3) & ~3);     // padded history [c][l + k]
    o.cf = p; p += ((L * F + 3) & ~3);      // conv features [l][f]
    o.e = p; p += ((L + 3) & ~3);           // energies, then softmax weights
    o.part = p; p += 4 * AT_THREADS;        // context partials [group][E]
    o.total = p;
    return o;
}

__global__ __launch_bounds__(AT_THREADS) void at_kernel(const AtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int L = a.L, A = a.A, E = a.E, F = a.F, K = a.K;
    const int A4 = (A + 3) & ~3;
    const AtLds o = at_layout(L, A, E, F, K);
    float* Wt = lds + o.wt;
    float* Wc = lds + o.wc;
    float* hs = lds + o.hs;
    float* cf = lds + o.cf;
    float* es = lds + o.e;
    float* part = lds + o.part;
    const int pad = (K - 1) / 2;
    const int hl = L + K - 1;

    // ---- P0: context prefetch (memory rows l = g, g+ng, ...), then LDS staging
    const int ne4 = E >> 2;                    // E % 4 == 0 checked on the host
    const int ng = AT_THREADS / ne4;           // row groups (>= 1 checked on the host)
    const int e4 = tid % ne4, g = tid / ne4;
    const bool ctx_active = g < ng;
    const float* memb = a.memory + (size_t)b * L * E;
    f32x4 mpf[AT_PF];
#pragma unroll
    for (int j = 0; j < AT_PF; ++j) {
        const int l = g + j * ng;
        mpf[j] = (ctx_active && l < L) ? st_ld4(memb + (size_t)l * E + e4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int idx = tid; idx < A * F; idx += AT_THREADS) {  // global [a][f] -> LDS [f][a]
        const int aa = idx / F, f = idx - aa * F;
        Wt[f * A4 + aa] = a.loc_lin_w[idx];
    }
    for (int idx = tid; idx < F * 2 * K; idx += AT_THREADS) Wc[idx] = a.loc_conv_w[idx];
    for (int idx = tid; idx < 2 * hl; idx += AT_THREADS) {
        const int c = idx / hl, p = idx - c * hl;
        const int l = p - pad;
        float v = 0.0f;
        if (l >= 0 && l < L) v = c == 0 ? a.w_prev[(size_t)b * a.ld_wprev + l] : a.w_cum_prev[(size_t)b * L + l];
        hs[idx] = v;
    }
    if (a.h_q) {  // AdaIN: relu(W_s s + b) * (h_q - (W_m s + b)), the two Linears are hoisted
        for (int j = tid; j < a.Q; j += AT_THREADS) {
            const size_t q = (size_t)b * a.Q + j;
            a.h_adapt[q] = a.ada_std[q] * (a.h_q[(size_t)b * a.ld_hq + j] - a.ada_mean[q]);
        }
    }
    __syncthreads();

    // ---- P1: location conv, cf[l][f] = sum_c sum_k Wc[f][c][k] * hist[c][l + k - pad]
    for (int idx = tid; idx < L * F; idx += AT_THREADS) {
        const int l = idx / F, f = idx - l * F;
        const float* w0 = Wc + f * 2 * K;
        float acc = 0.0f;
        for (int c = 0; c < 2; ++c) {
            const float* h = hs + c * hl + l;
            const float* w = w0 + c * K;
            for (int k = 0; k < K; ++k) acc = fmaf(w[k], h[k], acc);
        }
        cf[idx] = acc;
    }
    __syncthreads();

    // ---- P2: energies, one wave per l, lane owns attention dims a0..a0+3 of each 256-chunk
    const float* pmb = a.pm + (size_t)b * L * A;
    const float* pqb = a.pq + (size_t)b * A;
    for (int l = wave; l < L; l += AT_THREADS / 64) {
        float esum = 0.0f;
        for (int a0 = lane * 4; a0 < A; a0 += 256) {
            const int rem = A - a0;
            f32x4 pm4, pq4, v4;
            if (rem >= 4 && (A & 3) == 0) {
                pm4 = st_ld4(pmb + (size_t)l * A + a0);
                pq4 = st_ld4(pqb + a0);
                v4 = st_ld4(a.v + a0);
            } else {
                pm4 = st_ld4_guard(pmb + (size_t)l * A + a0, rem);
                pq4 = st_ld4_guard(pqb + a0, rem);
                v4 = st_ld4_guard(a.v + a0, rem);
            }
            f32x4 loc = {0.f, 0.f, 0.f, 0.f};
            const float* cfl = cf + l * F;
            for (int f = 0; f < F; ++f) {
                const float cv = cfl[f];
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(Wt + f * A4 + a0);
                loc[0] = fmaf(w4[0], cv, loc[0]); loc[1] = fmaf(w4[1], cv, loc[1]);
                loc[2] = fmaf(w4[2], cv, loc[2]); loc[3] = fmaf(w4[3], cv, loc[3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (processed_query + processed_loc_feat) + processed_memory, module.py:389-390
                const float t = tanhf((pq4[j] + loc[j]) + pm4[j]);
                esum = fmaf(v4[j], j < rem ? t : 0.0f, esum);
            }
        }
        esum = st_wave_sum(esum);
        if (lane == 0) es[l] = esum;
    }
    __syncthreads();

    // ---- P3: softmax over L (wave 0), write alignment and cumulative weights
    if (wave == 0) {
        float m = -INFINITY;
        for (int l = lane; l < L; l += 64) m = fmaxf(m, es[l]);
        m = st_wave_max(m);
        float s = 0.0f;
        for (int l = lane; l < L; l += 64) {
            const float ex = expf(es[l] - m);
            es[l] = ex;
            s += ex;
        }
        s = st_wave_sum(s);
        for (int l = lane; l < L; l += 64) {
            const float w = es[l] / s;
            es[l] = w;
            a.w_out[(size_t)b * a.ld_wout + l] = w;
            a.w_cum_out[(size_t)b * L + l] = w + hs[hl + pad + l];   // weights + attn_weights_sum, :264
        }
    }
    __syncthreads();

    // ---- P4: context
    if (ctx_active) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < AT_PF; ++j) {
            const int l = g + j * ng;
            if (l < L) {
                const float w = es[l];
                acc[0] = fmaf(w, mpf[j][0], acc[0]); acc[1] = fmaf(w, mpf[j][1], acc[1]);
                acc[2] = fmaf(w, mpf[j][2], acc[2]); acc[3] = fmaf(w, mpf[j][3], acc[3]);
            }
        }
        for (int l = g + AT_PF * ng; l < L; l += ng) {
            const f32x4 m4 = st_ld4(memb + (size_t)l * E + e4 * 4);
            const float w = es[l];
            acc[0] = fmaf(w, m4[0], acc[0]); acc[1] = fmaf(w, m4[1], acc[1]);
            acc[2] = fmaf(w, m4[2], acc[2]); acc[3] = fmaf(w, m4[3], acc[3]);
        }
        *reinterpret_cast<f32x4*>(part + (size_t)(g * ne4 + e4) * 4) = acc;
    }
    __syncthreads();
    for (int e = tid; e < E; e += AT_THREADS) {
        float s = 0.0f;
        for (int gg = 0; gg < ng; ++gg) s += part[gg * E + e];
        a.ctx[(size_t)b * a.ld_ctx + e] = s;
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
    ST_CHECK_ARG(B > 0 && L > 0 && A > 0 && E > 0 && F > 0 && K > 0, "st_attn_step_fwd: bad dims");
    ST_CHECK_ARG(K % 2 == 1, "st_attn_step_fwd: location kernel size %d must be odd", K);
    ST_CHECK_ARG(E % 4 == 0 && E / 4 <= AT_THREADS, "st_attn_step_fwd: E=%d must be a multiple of 4 and <= %d", E, 4 * AT_THREADS);
    ST_CHECK_ARG(st_aligned16(memory), "st_attn_step_fwd: memory must be 16-byte aligned");
    ST_CHECK_ARG((A % 4 != 0) || (st_aligned16(pm) && st_aligned16(pq) && st_aligned16(v)),
                 "st_attn_step_fwd: pm/pq/v must be 16-byte aligned");
    ST_CHECK_ARG(!h_q || (ada_std && ada_mean && h_adapt), "st_attn_step_fwd: AdaIN pointers");
    AtArgs a;
    a.pq = pq; a.pm = pm; a.memory = memory; a.w_prev = w_prev; a.ld_wprev = ld_wprev; a.w_cum_prev = w_cum_prev;
    a.w_out = w_out; a.ld_wout = ld_wout; a.w_cum_out = w_cum_out;
    a.loc_conv_w = loc_conv_w; a.loc_lin_w = loc_lin_w; a.v = v; a.ctx = ctx; a.ld_ctx = ld_ctx;
    a.h_q = h_q; a.ld_hq = ld_hq; a.ada_std = ada_std; a.ada_mean = ada_mean; a.h_adapt = h_adapt; a.Q = Q;
    a.B = B; a.L = L; a.A = A; a.E = E; a.F = F; a.K = K;
    const AtLds o = at_layout(L, A, E, F, K);
    const size_t lds_bytes = (size_t)o.total * sizeof(float);
    ST_CHECK_ARG(lds_bytes <= 160 * 1024, "st_attn_step_fwd: L=%d needs %zu B of LDS (> 160 KiB)", L, lds_bytes);
    static size_t configured = 0;
    if (lds_bytes > configured) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(at_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = 160 * 1024;
    }
    hipLaunchKernelGGL(at_kernel, dim3(B), dim3(AT_THREADS), lds_bytes, (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}
