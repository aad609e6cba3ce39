// attention_body.h -- fused location-sensitive attention step for gfx950 (MI355X): device code, shared by attention.hip
// (the stand-alone launches) and skinny_packed.hip (the "pre" part as extra workgroups of the proj launch).
//
// One workgroup (8 wavefronts) per utterance does the whole of Attention.forward for one
// decode step (ref: src/module.py:371-407 + state update :262-264 [+ AdaIN :267-269]):
//   P0  issue the coalesced 16-byte loads of the encoder-memory rows this thread needs for
//       the context (they land while P1-P3 run); stage loc_linear (transposed to [f][a]),
//       loc_conv and the zero-padded attention history in LDS
//   P1  location conv  cf[f][l]  (2 -> F channels, K taps): one thread = one filter x 4
//       consecutive positions (sliding register window, 4 independent accumulators)
//   P2  energies  e[l] = v . tanh((pq + W_l cf[l]) + pm[l]): one wave = 8 consecutive
//       positions, each lane owns 4 attention dims; per filter one ds_read_b128 of W_l^T and two
//       broadcast ds_read_b128 of cf feed 32 FMAs; the processed-memory rows are 1 KiB
//       coalesced wave loads issued before the filter loop; wave shuffle reduction over dims
//   P3  softmax over L by wave 0 (shuffle max / sum), alignment + cumulative weights out
//   P4  context = sum_l w[l] * memory[l][:]   (row groups, LDS cross-group reduction)
// Everything between the loads and the stores stays in LDS/registers; per step and utterance
// the kernel reads pm (L*A*4 B) + memory (L*E*4 B) exactly once.
#pragma once
#include "st_common.h"

#ifndef AT_PROF
#define AT_PROF(n)   // phase timestamps, only defined by tools/mb/mb_attn.hip
#endif

namespace {

constexpr int AT_THREADS = 512;
constexpr int AT_WAVES = AT_THREADS / 64;
constexpr int AT_PF = 12;   // encoder-memory rows prefetched into registers per thread
constexpr int AT_LB = 8;    // slots of the folded energy reduction (values a wave reduces at once)
constexpr int AT_LP = 6;    // positions per wave and round in the energy phase: ceil(43 / 6) = 8 blocks = one per wave, so all
                            // four SIMDs carry two busy waves (with 8 positions per wave L = 43 gives 6 blocks: two SIMDs idle half the time)
constexpr int AT_CB = 4;    // positions per thread in the conv phase
constexpr int AT_WLPF = 4;  // float4 pieces of W_l each thread parks in registers during the conv

struct AtArgs {
    const float* pq; const float* pm; const float* memory;
    const float* w_prev; int ld_wprev; const float* w_cum_prev;
    float* w_out; int ld_wout; float* w_cum_out;
    const float* loc_conv_w; const float* loc_lin_w; const float* v;
    float* ctx; int ld_ctx; st_t16_view ctx_dst[3];
    const float* h_q; int ld_hq; const float* ada_std; const float* ada_mean; float* h_adapt; int Q;
    float* s_buf;      // (B, L, A): S = pm + W_l cf, written by the pre part, read by the fin part
    float* cf_out;     // pre part only, optional: location features (B, L, F) of this step, saved for the backward pass
    int pre_parts;     // pre part only: workgroups per utterance, each a contiguous range of positions (1, 2 or 4)
    int fin_parts;     // fin part only: workgroups per utterance, each a slice of the context dims E (1, 2, 4 or 8); every one
                       // repeats the energies + softmax (cheap), so the memory rows -- the bulk of the bytes -- are spread over more CUs
    int B, L, A, E, F, K;
    // fin part inside the query-projection launch (pk_attnfin_kernel): the processed query arrives as 8-byte {value, tag} granules
    // written by the linear's workgroups of the SAME launch; tag = epoch (decode step + 1)
    const unsigned long long* pq_gran; unsigned epoch;
    unsigned* status;  // optional: bit 0 set when the wait for the granules timed out
};

// Wait of one wave for the A {value, tag} granules of one utterance (lane l takes granules 4l .. 4l+3): re-read until every tag is
// `epoch`, at most `max_spins` times.  On a time-out the values are NaN AND bit 0 of *status is set (when status is given): the
// caller of the forward checks that word, so a starved launch is an error and not a NaN to be found in a .npy file later.
constexpr int AT_GRAN_SPINS = 1 << 18;
__device__ __forceinline__ f32x4 at_wait_granules(const unsigned long long* base, unsigned epoch, int lane, int A, unsigned* status,
                                                  int max_spins) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    gu64* gp = (gu64*)(base + min(lane * 4, A - 4));
    unsigned long long g0 = 0, g1 = 0, g2 = 0, g3 = 0;
    bool ok = false;
    for (int spins = 0; spins < max_spins; ++spins) {
        g0 = __hip_atomic_load(gp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g1 = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g2 = __hip_atomic_load(gp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g3 = __hip_atomic_load(gp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool mine = (unsigned)(g0 >> 32) == epoch && (unsigned)(g1 >> 32) == epoch &&
                          (unsigned)(g2 >> 32) == epoch && (unsigned)(g3 >> 32) == epoch;
        ok = __all(mine);
        if (ok) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (!ok && lane == 0 && status) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float nanv = __builtin_nanf("");
    return ok ? f32x4{__uint_as_float((unsigned)g0), __uint_as_float((unsigned)g1), __uint_as_float((unsigned)g2),
                      __uint_as_float((unsigned)g3)}
              : f32x4{nanv, nanv, nanv, nanv};
}

struct AtLds {  // offsets in floats into dynamic LDS
    int wt, wt_ld, wc, kp, hs, hl, cf, cf_ld, e, part, total;
};

// positions per workgroup of the pre part (multiples of 12 = lcm of the conv's 4-blocks and the energy phase's 6-blocks)
__host__ __device__ inline int at_pos_per(int L, int nparts) {
    return nparts > 1 ? (((L + nparts - 1) / nparts + 11) / 12) * 12 : L;
}

// part: 0 = the whole step, 1 = pre (history and conv features are held for the workgroup's OWN range of `Lr` positions only,
// and W_l^T is not staged when S runs on the matrix cores), 2 = fin (energies + context partials only) -- so the split step has
// no practical limit on the text length (the whole step keeps an utterance's conv features in LDS: L up to ~400)
__host__ __device__ inline AtLds at_layout(int L, int A, int E, int F, int K, int part = 0, int Lr = 0, bool pre_mfma = false) {
    AtLds o;
    int p = 0;
    const int Lh = part == 1 ? Lr : L;
    o.wt_ld = ((A + 3) & ~3) + 4;               // row stride of W_l^T: +4 spreads the transposing stores
    o.wt = p; if (part == 0 || (part == 1 && !pre_mfma)) p += F * o.wt_ld;
    o.kp = K <= 32 ? 32 : ((K + 3) & ~3);       // filter rows padded so they can be read as float4
    o.wc = p; if (part != 2) p += F * 2 * o.kp; // loc_conv [f][c][kp]
    o.hl = ((Lh + AT_CB + o.kp + 3) + 3) & ~3;  // padded history length per channel (window of kp + 4)
    o.hs = p; if (part != 2) p += 2 * o.hl;
    // conv features [f][l], l padded to the wave block, the conv's 4-wide stores and the 16-position MFMA tiles
    o.cf_ld = ((((Lh + AT_LP - 1) / AT_LP) * AT_LP + AT_CB + 3) & ~3) + (part == 1 ? 16 : 0);
    o.cf = p; if (part != 2) p += F * o.cf_ld;
    o.e = p; p += ((L + 3) & ~3);               // energies, then softmax weights
    o.part = p; p += 4 * AT_THREADS;            // context partials [group][E]
    o.total = p;
    return o;
}

// tanh from one v_exp_f32 and one fast reciprocal: |error| <= ~2e-7 absolute (the energies feed a
// softmax whose outputs are compared at 1e-5; the accurate tanhf costs ~10x more and was 50% of the kernel)
__device__ __forceinline__ float at_tanh(float x) {
    const float t = __expf(-2.0f * fabsf(x));
    return copysignf(__fdividef(1.0f - t, 1.0f + t), x);
}

__device__ __forceinline__ size_t at_t16_off(int b, int k, int KB) {
    return (((size_t)(b >> 4) * KB + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (b & 15)) * 4 + (k & 3);
}

__device__ __forceinline__ void at_put_granule(unsigned long long* p, float v, unsigned epoch) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    __hip_atomic_store((gu64*)p, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// ---- fin part over POSITION ranges with the combine INSIDE the launch (long texts; ref: Attention.forward src/module.py:371-407 +
// state update :262-264).  Workgroup (b, p) takes the positions [p Lp, (p+1) Lp) of utterance b exactly like at_split_kernel
// (attention.hip): energies from its S rows, a LOCAL softmax (m_p, w~_l = exp(e_l - m_p), s_p) and the un-normalised partial context
// c_p -- S and the memory rows are read once in total, 1/P of them per compute unit.  Instead of a second launch, the P workgroups
// of an utterance exchange (m_p, s_p, c_p) as 8-byte {value, tag = epoch} granules (the hand-off the processed query already uses)
// and every one of them finishes ITS share: the context dims [p E/P, (p+1) E/P) and the weights of its own positions, scaled by
// exp(m_p - M) / D.  All workgroups of the launch must be resident at once (the launcher asks the occupancy API).
struct ArArgs {
    const float* s_buf; const float* memory; const float* v; const float* w_cum_prev;
    float* w_out; int ld_wout; float* w_cum_out;
    st_t16_view ctx_dst[3];
    const unsigned long long* pq_gran;     // (B, A) granules of the processed query, written by the linear's workgroups of this launch
    unsigned long long* xchg;              // (B, P, E + 4) granules: c_p (E), m_p, s_p, two pads
    unsigned epoch; unsigned* status;
    int B, L, A, E, P, Lp;
};

template <int NT>
__device__ __forceinline__ void at_range_body(const ArArgs& a, const int wg) {
    __shared__ float es[512];                           // energies, then w~ of this range (Lp <= 512)
    __shared__ __attribute__((aligned(16))) float part[4 * NT];
    __shared__ __attribute__((aligned(16))) float pqs[256];
    __shared__ __attribute__((aligned(16))) float cb[8 * 256];   // the other parts' partial contexts, this workgroup's dims only
    __shared__ float stat[2], gst[16];
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, b = wg / P, p = wg - b * P;
    const int L = a.L, A = a.A, E = a.E;
    const int l0 = p * a.Lp, l1 = min(L, l0 + a.Lp), n = l1 - l0;
    const float* sb = a.s_buf + ((size_t)b * L + l0) * A;
    const float* memb = a.memory + ((size_t)b * L + l0) * E;
    const int ne4 = E >> 2;
    const int ng = NT / ne4, e4 = tid % ne4, g = tid / ne4;
    constexpr int PFR = 11;                 // (44 rows with E = 512: a whole range of <= 44 positions is in flight from the start)
    f32x4 mpf[PFR];
#pragma unroll
    for (int j = 0; j < PFR; ++j) {
        const int l = g + j * ng;
        mpf[j] = (g < ng && l < n) ? st_ld4(memb + (size_t)l * E + e4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int a0 = min(lane * 4, A - 4);
    const bool a_on = lane * 4 < A;
    const f32x4 v4 = st_ld4(a.v + a0);
    const float vsum = (v4[0] + v4[1]) + (v4[2] + v4[3]);
    constexpr int NPW = 8;
    f32x4 spf[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) spf[i] = st_ld4(sb + (size_t)min(wave + i * NW, max(n - 1, 0)) * A + a0);
    // the processed query of this step: one wave polls the granules, the others take it from LDS
    if (wave == 0) {
        const f32x4 q4 = at_wait_granules(a.pq_gran + (size_t)b * A, a.epoch, lane, A, a.status, AT_GRAN_SPINS);
        if (lane * 4 < A) *reinterpret_cast<f32x4*>(pqs + lane * 4) = q4;
    }
    __syncthreads();
    const f32x4 pq4 = *reinterpret_cast<const f32x4*>(pqs + a0);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int l = wave + i * NW;
        if (l >= n) break;
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {      // v . tanh(x), tanh(x) = 1 - 2 / (1 + exp(2x)) as in at_body
            const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((pq4[c] + spf[i][c]) * 2.885390081777927f));
            acc = fmaf(v4[c], r, acc);
        }
        float e = a_on ? fmaf(-2.0f, acc, vsum) : 0.0f;
        e = st_wave_sum_dpp(e);
        if (lane == 0) es[l] = e;
    }
    for (int l = wave + NPW * NW; l < n; l += NW) {
        const f32x4 s4 = st_ld4(sb + (size_t)l * A + a0);
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((pq4[c] + s4[c]) * 2.885390081777927f));
            acc = fmaf(v4[c], r, acc);
        }
        float e = a_on ? fmaf(-2.0f, acc, vsum) : 0.0f;
        e = st_wave_sum_dpp(e);
        if (lane == 0) es[l] = e;
    }
    __syncthreads();
    unsigned long long* xme = a.xchg + ((size_t)b * P + p) * (E + 4);
    if (wave == 0) {      // local softmax statistics, published at once
        float m = -INFINITY;
        for (int l = lane; l < n; l += 64) m = fmaxf(m, es[l]);
        m = st_wave_max_dpp(m);
        float ssum = 0.0f;
        for (int l = lane; l < n; l += 64) {
            const float w = __expf(es[l] - m);
            es[l] = w;
            ssum += w;
        }
        ssum = st_wave_sum_dpp(ssum);
        if (lane < 4) at_put_granule(xme + E + lane, lane == 0 ? m : (lane == 1 ? ssum : 0.0f), a.epoch);
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
    for (int e = tid; e < E; e += NT) {       // partial context of this range: one granule per context dim
        float s = 0.0f;
        for (int gg = 0; gg < ng; ++gg) s += part[gg * E + e];
        at_put_granule(xme + e, s, a.epoch);
    }
    // ---- combine: wave q collects part q's statistics and the slice of its partial context this workgroup finishes
    const int Es = E / P;
    if (wave < P) {
        const unsigned long long* xq = a.xchg + ((size_t)b * P + wave) * (E + 4);
        const f32x4 st4 = at_wait_granules(xq + E, a.epoch, lane, 4, a.status, AT_GRAN_SPINS);
        const f32x4 c4 = at_wait_granules(xq + (size_t)p * Es, a.epoch, lane, Es, a.status, AT_GRAN_SPINS);
        if (lane * 4 < Es) *reinterpret_cast<f32x4*>(cb + wave * 256 + lane * 4) = c4;
        if (lane == 0) { gst[2 * wave] = st4[0]; gst[2 * wave + 1] = st4[1]; }
    }
    __syncthreads();
    float M = -INFINITY;
    for (int q = 0; q < P; ++q) M = fmaxf(M, gst[2 * q]);
    float D = 0.0f;
    for (int q = 0; q < P; ++q) D += gst[2 * q + 1] * __expf(gst[2 * q] - M);
    if (tid < Es) {
        float s = 0.0f;
        for (int q = 0; q < P; ++q) s = fmaf(__expf(gst[2 * q] - M) / D, cb[q * 256 + tid], s);
        const int e = p * Es + tid;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (a.ctx_dst[d].base) a.ctx_dst[d].base[at_t16_off(b, a.ctx_dst[d].kb0 * 16 + e, a.ctx_dst[d].kb_stride)] = s;
    }
    const float scp = __expf(stat[0] - M) / D;
    for (int l = tid; l < n; l += NT) {
        const float w = es[l] * scp;
        a.w_out[(size_t)b * a.ld_wout + l0 + l] = w;
        a.w_cum_out[(size_t)b * L + l0 + l] = w + a.w_cum_prev[(size_t)b * L + l0 + l];      // weights + attn_weights_sum, :264
    }
}

// PART 0: the whole step.  PART 1 ("pre"): only what depends on the PREVIOUS step's attention weights -- location conv and
// S[l][a] = pm[l][a] + sum_f W_l[a][f] cf[f][l] -- written to a.s_buf; it can run while the rest of the decode step does
// (as extra workgroups of the proj launch, skinny_packed.hip).  PART 2 ("fin"): energies from S, softmax, context.
// NT = threads per workgroup (512 in every launch; a 256-thread fin part measured slower, see attention.hip)
template <bool VEC, int PART, int NT = AT_THREADS, bool GRAN = false>
__device__ __forceinline__ void at_body(const AtArgs& a, const int wg, float* lds) {
    static_assert(!GRAN || (PART == 2 && VEC), "granule hand-off: fin part, vector form only");
    constexpr int NWV = NT / 64;                      // waves
    constexpr int PFR = AT_PF * (AT_THREADS / NT);    // memory rows parked in registers per thread
    constexpr int NPB = PART == 2 ? AT_THREADS / NT : 0;   // energy blocks per wave whose operands are requested at kernel entry
    // the pre part may spread an utterance over several workgroups (ranges of positions): only the conv and the W_l product
    // scale with the range, the staging is repeated
    const int nparts = (PART == 1 && a.pre_parts > 1) ? a.pre_parts : (PART == 2 && a.fin_parts > 1) ? a.fin_parts : 1;
    // (parts are powers of two -- checked on the host: shifts instead of integer divisions; this prologue is issue bound)
    const int pshift = 31 - __builtin_clz((unsigned)nparts);
    const int b = wg >> pshift, ipart = wg & (nparts - 1);
    // kernel arguments of the first phases: fetched now, one wait (otherwise one scalar-cache round trip per first use)
#define AT_TOUCH(x) asm volatile("" :: "s"(x))
    AT_TOUCH(a.pq); AT_TOUCH(a.pm); AT_TOUCH(a.memory); AT_TOUCH(a.w_prev); AT_TOUCH(a.ld_wprev); AT_TOUCH(a.w_cum_prev);
    AT_TOUCH(a.s_buf); AT_TOUCH(a.w_out); AT_TOUCH(a.w_cum_out); AT_TOUCH(a.ld_wout);
    AT_TOUCH(a.loc_conv_w); AT_TOUCH(a.loc_lin_w); AT_TOUCH(a.v); AT_TOUCH(a.L); AT_TOUCH(a.A); AT_TOUCH(a.E); AT_TOUCH(a.F); AT_TOUCH(a.K);
#undef AT_TOUCH
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.L, A = a.A, E = a.E, F = a.F, K = a.K;
    // position range of this workgroup
    const int pos_per = PART == 1 ? at_pos_per(L, nparts) : L;
    const int pos_lo = PART == 1 ? min(L, ipart * pos_per) : 0, pos_hi = min(L, pos_lo + pos_per);
    const AtLds o = at_layout(L, A, E, F, K, PART, pos_per, PART == 1 && VEC && F == 32 && (A & 15) == 0);
    const int lb = PART == 1 ? pos_lo : 0;       // first position held in the LDS history / conv-feature arrays
    float* Wt = lds + o.wt;
    float* Wc = lds + o.wc;
    float* hs = lds + o.hs;
    float* cf = lds + o.cf;
    float* es = lds + o.e;
    float* part = lds + o.part;
    const int pad = (K - 1) / 2;

    AT_PROF(0);
    // ---- fin part: the operands of the energy phase are requested first (nothing is staged in LDS for this part, so the
    // wave's first block of S rows, pq and v can be in flight from the first instruction on), then cum_prev for the softmax
    const float* pmb = (PART == 2 ? a.s_buf : a.pm) + (size_t)b * L * A;
    const float* pqb = a.pq + (size_t)b * A;
    f32x4 pf_pm4[NPB > 0 ? NPB : 1][AT_LP], pf_pq4 = {0.f, 0.f, 0.f, 0.f}, pf_v4 = {0.f, 0.f, 0.f, 0.f};
    float pf_cum = 0.0f;
    const bool pf_ok = PART == 2 && VEC && lane * 4 < A;
    if (PART == 2 && VEC) {
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
#pragma unroll
            for (int j = 0; j < AT_LP; ++j) {
                const int l = min((wave + i * NWV) * AT_LP + j, L - 1);
                pf_pm4[i][j] = st_ld4(pmb + (size_t)l * A + min(lane * 4, A - 4));
            }
        }
        if (!GRAN) pf_pq4 = st_ld4(pqb + min(lane * 4, A - 4));
        pf_v4 = st_ld4(a.v + min(lane * 4, A - 4));
        if (wave == 0) pf_cum = a.w_cum_prev[(size_t)b * L + min(lane, L - 1)];
    }
    // ---- P0a: context prefetch (memory rows l = g, g+ng, ...)
    const int Es = PART == 2 ? E >> pshift : E; // context dims of this workgroup (E % (4 * parts) == 0 checked on the host)
    const int e_lo = PART == 2 ? ipart * Es : 0;
    const int ne4 = Es >> 2;
    int ng, e4, g;                     // row groups (>= 1 checked on the host), this thread's float4 column and group
    if ((ne4 & (ne4 - 1)) == 0) {      // power of two (the usual E = 512): no integer divisions
        const int sh = 31 - __builtin_clz((unsigned)ne4);
        ng = NT >> sh; e4 = tid & (ne4 - 1); g = tid >> sh;
    } else {
        ng = NT / ne4; e4 = tid % ne4; g = tid / ne4;
    }
    const bool ctx_active = g < ng;
    const float* memb = a.memory + (size_t)b * L * E;
    f32x4 mpf[PFR];
    if (PART != 1) {
        // buffer loads: one descriptor per utterance whose size ends at row L, so rows past L (and idle threads, sent past
        // the end) read zeros without a branch or a 64-bit address per row
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)memb, 0, L * E * 4, 0x00020000);
        const int v0 = ctx_active ? (g * E + e_lo + e4 * 4) * 4 : 0x7ffffff0;
        const int vstep = ng * E * 4;
#pragma unroll
        for (int j = 0; j < PFR; ++j)
            mpf[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, ctx_active ? v0 + j * vstep : v0, 0, 0));
    }

    if (GRAN) {
        // The processed query of THIS step is produced by the linear's workgroups of the same launch (pk_attnfin_kernel) and
        // handed over as 8-byte {value, tag} granules (one relaxed agent-scope store each, write-through; the data is the flag:
        // no fence on either side -- MI355X guide, hand-off recipe R2).  Everything else this workgroup needs (S rows, v, the
        // memory rows) was requested above and is in flight while the wave re-reads its four granules until every tag is
        // this step's epoch.  The spin is bounded: on a time-out the query is poisoned with NaN and bit 0 of the status word is set.
        // ONE wave per workgroup polls (eight polling waves per workgroup measured 0.35 us per step slower: the polls compete with
        // the producers' own loads); the others take the query from LDS -- the context-partial area is free at this point
        if (wave == 0) {
            const f32x4 q4 = at_wait_granules(a.pq_gran + (size_t)b * A, a.epoch, lane, A, a.status, AT_GRAN_SPINS);
            if (lane * 4 < A) *reinterpret_cast<f32x4*>(part + lane * 4) = q4;
        }
        __syncthreads();
        pf_pq4 = *reinterpret_cast<const f32x4*>(part + min(lane * 4, A - 4));
    }
    AT_PROF(1);
    const bool wl_vec = VEC || ((F & 3) == 0 && st_aligned16(a.loc_lin_w));
    const int f4n = F >> 2;
    const int KP = o.kp;
    f32x4 wl4[AT_WLPF];
    // pre part with F == 32, A % 16 == 0: S = pm + cf^T W_l^T runs on the matrix cores (exact-fp32 16x16x4 MFMA): a wave owns
    // 16-dim tiles of A (mt = wave, wave + 8, ...), a tile of 16 positions is one MFMA column block, the 32 filters are 8
    // k-steps (lane group g = lane >> 4 takes filters 8g..8g+7, so its A fragment is 32 contiguous bytes of a W_l row).  The
    // first two dim tiles' fragments and their processed-memory rows (three position tiles) are requested here, before the conv.
    constexpr int SM_MT = 2, SM_NT = 3;
    const bool s_mfma = PART == 1 && VEC && F == 32 && (A & 15) == 0;
    const int sm_g = lane >> 4, sm_n = lane & 15;
    const int sm_ntile = (pos_hi - pos_lo + 15) >> 4;
    float sm_aw[SM_MT][8];
    f32x4 sm_pm[SM_MT][SM_NT];
    const float* sm_pmb = a.pm + (size_t)b * L * A;
    if (PART != 2) {
    // ---- P0b: W_l (A,F) is fetched now (coalesced along f) and parked in registers; it is only
    // needed by P2, so its transposing LDS stores happen after the conv
    if (wl_vec && !s_mfma) {
#pragma unroll
        for (int j = 0; j < AT_WLPF; ++j) {
            const int idx = tid + j * NT;
            wl4[j] = idx < A * f4n ? st_ld4(a.loc_lin_w + (size_t)idx * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // conv filters [f][c][KP] (rows padded to KP so they can be read as float4) and padded history
    // (all global loads of this phase are issued before the first LDS store: a load -> store loop costs one memory round
    // trip per iteration)
    {
        const int nwc = F * 2 * KP, nhs = 2 * o.hl;
        float wcv[4], hv[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + j * NT;
            const int row = idx / KP, k = idx - row * KP;
            wcv[j] = (idx < nwc && k < K) ? a.loc_conv_w[row * K + k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + j * NT;
            const int c = idx / o.hl, p = idx - c * o.hl, l = lb + p - pad;
            float v = 0.0f;
            if (idx < nhs && l >= 0 && l < L) v = c == 0 ? a.w_prev[(size_t)b * a.ld_wprev + l] : a.w_cum_prev[(size_t)b * L + l];
            hv[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int idx = tid + j * NT; if (idx < nwc) Wc[idx] = wcv[j]; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int idx = tid + j * NT; if (idx < nhs) hs[idx] = hv[j]; }
        for (int idx = tid + 4 * NT; idx < nwc; idx += NT) {      // sizes beyond the register rounds
            const int row = idx / KP, k = idx - row * KP;
            Wc[idx] = k < K ? a.loc_conv_w[row * K + k] : 0.0f;
        }
        for (int idx = tid + 2 * NT; idx < nhs; idx += NT) {
            const int c = idx / o.hl, p = idx - c * o.hl, l = lb + p - pad;
            float v = 0.0f;
            if (l >= 0 && l < L) v = c == 0 ? a.w_prev[(size_t)b * a.ld_wprev + l] : a.w_cum_prev[(size_t)b * L + l];
            hs[idx] = v;
        }
    }
    }
    if (a.h_q && ipart == 0) {  // AdaIN: relu(W_s s + b) * (h_q - (W_m s + b)), the two Linears are hoisted
        for (int j = tid; j < a.Q; j += NT) {
            const size_t q = (size_t)b * a.Q + j;
            a.h_adapt[q] = a.ada_std[q] * (a.h_q[(size_t)b * a.ld_hq + j] - a.ada_mean[q]);
        }
    }
    AT_PROF(2);
    if (PART != 2) st_lds_barrier();
    AT_PROF(3);

    // Requested after the staging barrier (80 KB per workgroup in front of the other waves' staging loads delayed that
    // barrier by ~2000 cycles; the conv below covers their latency) and WITHOUT a run-time branch around them (a branch
    // makes the compiler wait for every outstanding load at the join): shapes that do not take the MFMA form read clamped,
    // valid addresses and ignore the values.
    if (PART == 1 && VEC) {
        const int wmax = A * F - 8, pmax = L * A - 4;
#pragma unroll
        for (int i = 0; i < SM_MT; ++i) {
            const int a0 = max(min(wave + i * NWV, (A >> 4) - 1), 0) * 16;        // a tile past A is loaded, not used
            const int wi = max(min((a0 + sm_n) * F + sm_g * 8, wmax), 0) & ~3;
            const f32x4 w0 = st_ld4(a.loc_lin_w + wi);
            const f32x4 w1 = st_ld4(a.loc_lin_w + wi + 4);
            sm_aw[i][0] = w0[0]; sm_aw[i][1] = w0[1]; sm_aw[i][2] = w0[2]; sm_aw[i][3] = w0[3];
            sm_aw[i][4] = w1[0]; sm_aw[i][5] = w1[1]; sm_aw[i][6] = w1[2]; sm_aw[i][7] = w1[3];
#pragma unroll
            for (int j = 0; j < SM_NT; ++j) {
                const int l = min(pos_lo + j * 16 + sm_n, L - 1);
                sm_pm[i][j] = st_ld4(sm_pmb + (min(l * A + a0 + 4 * sm_g, pmax) & ~3));
            }
        }
    }
    if (PART != 2) {
    // ---- P1: location conv, cf[f][l] = sum_c sum_k Wc[f][c][k] * hist[c][l + k - pad]
    {
        const int nlb = (pos_hi - pos_lo + AT_CB - 1) / AT_CB;
        for (int idx = tid; idx < F * nlb; idx += NT) {
            const int f = idx / nlb, l0 = pos_lo + (idx - f * nlb) * AT_CB;
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
            if (KP == 32) {
                // K <= 32: filters and history window are pulled with ds_read_b128, 8 taps at a time (2 + 3 reads feed 8 x 4 FMAs
                // out of 20 registers; the whole 32-tap window at once took 68 and set the register count of every kernel this
                // body is compiled into), taps in ascending order
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x4* hq = reinterpret_cast<const f32x4*>(hs + c * o.hl + (l0 - lb));
                    const f32x4* wq = reinterpret_cast<const f32x4*>(Wc + (f * 2 + c) * 32);
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) {
                        float h[12], w[8];
#pragma unroll
                        for (int j = 0; j < 3; ++j) { const f32x4 t = hq[2 * kc + j]; h[4 * j] = t[0]; h[4 * j + 1] = t[1]; h[4 * j + 2] = t[2]; h[4 * j + 3] = t[3]; }
#pragma unroll
                        for (int j = 0; j < 2; ++j) { const f32x4 t = wq[2 * kc + j]; w[4 * j] = t[0]; w[4 * j + 1] = t[1]; w[4 * j + 2] = t[2]; w[4 * j + 3] = t[3]; }
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            acc0 = fmaf(w[k], h[k], acc0); acc1 = fmaf(w[k], h[k + 1], acc1);
                            acc2 = fmaf(w[k], h[k + 2], acc2); acc3 = fmaf(w[k], h[k + 3], acc3);
                        }
                    }
                }
            } else {
                for (int c = 0; c < 2; ++c) {
                    const float* h = hs + c * o.hl + (l0 - lb);
                    const float* w = Wc + (f * 2 + c) * KP;
                    float h0 = h[0], h1 = h[1], h2 = h[2];
#pragma unroll 8
                    for (int k = 0; k < K; ++k) {
                        const float h3 = h[k + 3];
                        const float wk = w[k];
                        acc0 = fmaf(wk, h0, acc0); acc1 = fmaf(wk, h1, acc1);
                        acc2 = fmaf(wk, h2, acc2); acc3 = fmaf(wk, h3, acc3);
                        h0 = h1; h1 = h2; h2 = h3;
                    }
                }
            }
            float* dst = cf + f * o.cf_ld + (l0 - lb);   // cf_ld is padded, stores beyond the range are harmless
            dst[0] = acc0;
            if (l0 - lb + 1 < o.cf_ld) dst[1] = acc1;
            if (l0 - lb + 2 < o.cf_ld) dst[2] = acc2;
            if (l0 - lb + 3 < o.cf_ld) dst[3] = acc3;
            if (PART == 1 && a.cf_out) {             // training: the backward pass reuses them (dW_l = ds^T loc) instead of recomputing
                float* cg = a.cf_out + ((size_t)b * L + l0) * F + f;
                if (l0 < pos_hi) cg[0] = acc0;
                if (l0 + 1 < pos_hi) cg[F] = acc1;
                if (l0 + 2 < pos_hi) cg[2 * F] = acc2;
                if (l0 + 3 < pos_hi) cg[3 * F] = acc3;
            }
        }
    }
    // W_l^T into LDS: Wt[f][a] (row stride wt_ld = A4 + 4 spreads the transposing stores over banks)
    if (s_mfma) {
    } else if (wl_vec) {
#pragma unroll
        for (int j = 0; j < AT_WLPF; ++j) {
            const int idx = tid + j * NT;
            if (idx < A * f4n) {
                const int aa = idx / f4n, f0 = (idx - aa * f4n) * 4;
                Wt[(f0 + 0) * o.wt_ld + aa] = wl4[j][0]; Wt[(f0 + 1) * o.wt_ld + aa] = wl4[j][1];
                Wt[(f0 + 2) * o.wt_ld + aa] = wl4[j][2]; Wt[(f0 + 3) * o.wt_ld + aa] = wl4[j][3];
            }
        }
        for (int idx = tid + AT_WLPF * NT; idx < A * f4n; idx += NT) {
            const int aa = idx / f4n, f0 = (idx - aa * f4n) * 4;
            const f32x4 w4 = st_ld4(a.loc_lin_w + (size_t)idx * 4);
            Wt[(f0 + 0) * o.wt_ld + aa] = w4[0]; Wt[(f0 + 1) * o.wt_ld + aa] = w4[1];
            Wt[(f0 + 2) * o.wt_ld + aa] = w4[2]; Wt[(f0 + 3) * o.wt_ld + aa] = w4[3];
        }
    } else {
        for (int idx = tid; idx < A * F; idx += NT) {
            const int aa = idx / F, f = idx - aa * F;
            Wt[f * o.wt_ld + aa] = a.loc_lin_w[idx];
        }
    }
    }
    AT_PROF(4);
    if (PART != 2) st_lds_barrier();
    AT_PROF(5);

    if (PART == 1 && s_mfma) {
        // D[row = 4g + r][col = n] of a (dim tile, position tile): lane (g, n) ends up with S[l0 + n][a0 + 4g .. 4g + 3]
        auto sm_tile = [&](const int a0, const int nt, const float (&aw)[8], const f32x4 pm4) __attribute__((always_inline)) {
            const int l0 = pos_lo + nt * 16, l = l0 + sm_n;
            const float* cfp = cf + (sm_g * 8) * o.cf_ld + (l0 - lb) + sm_n;      // columns past the range read the row's padding: discarded
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[k], cfp[k * o.cf_ld], acc, 0, 0, 0);
            if (l < pos_hi) {
                const f32x4 sv = {acc[0] + pm4[0], acc[1] + pm4[1], acc[2] + pm4[2], acc[3] + pm4[3]};
                *reinterpret_cast<f32x4*>(a.s_buf + ((size_t)b * L + l) * A + a0 + 4 * sm_g) = sv;
            }
        };
#pragma unroll
        for (int i = 0; i < SM_MT; ++i) {
            const int mt = wave + i * NWV;
            if (mt >= (A >> 4)) break;
#pragma unroll
            for (int j = 0; j < SM_NT; ++j)
                if (j < sm_ntile) sm_tile(mt * 16, j, sm_aw[i], sm_pm[i][j]);
            for (int j = SM_NT; j < sm_ntile; ++j) {        // longer position ranges: rows fetched in place
                const int l = min(pos_lo + j * 16 + sm_n, L - 1);
                sm_tile(mt * 16, j, sm_aw[i], st_ld4(sm_pmb + (size_t)l * A + mt * 16 + 4 * sm_g));
            }
        }
        for (int mt = wave + SM_MT * NWV; mt < (A >> 4); mt += NWV) {      // A > 256: fragments fetched in place
            const int a0 = mt * 16;
            const f32x4 w0 = st_ld4(a.loc_lin_w + (size_t)(a0 + sm_n) * 32 + sm_g * 8);
            const f32x4 w1 = st_ld4(a.loc_lin_w + (size_t)(a0 + sm_n) * 32 + sm_g * 8 + 4);
            const float aw[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
            for (int j = 0; j < sm_ntile; ++j) {
                const int l = min(pos_lo + j * 16 + sm_n, L - 1);
                sm_tile(a0, j, aw, st_ld4(sm_pmb + (size_t)l * A + a0 + 4 * sm_g));
            }
        }
        return;
    }

    // ---- P2: energies; a wave owns AT_LB consecutive positions, a lane 4 consecutive dims
    // PART 2 reads S (= pm + W_l cf, written by the pre part) where the full kernel reads pm
    auto energy_block = [&](const int l0, const f32x4* pf) __attribute__((always_inline)) {
        float esum[AT_LB];
#pragma unroll
        for (int j = 0; j < AT_LB; ++j) esum[j] = 0.0f;        // slots AT_LP.. stay zero
        for (int a0 = lane * 4; a0 < A; a0 += 256) {
            const int rem = A - a0;
            f32x4 pm4[AT_LP], pq4, v4;
            if (pf && pf_ok && a0 == lane * 4) {      // requested at the top of the kernel
#pragma unroll
                for (int j = 0; j < AT_LP; ++j) pm4[j] = pf[j];
                pq4 = pf_pq4;
                v4 = pf_v4;
            } else if (VEC) {   // A % 4 == 0 and aligned operands: plain 16-byte loads, no per-lane branches
#pragma unroll
                for (int j = 0; j < AT_LP; ++j) {
                    const int l = l0 + j < L ? l0 + j : L - 1;
                    pm4[j] = st_ld4(pmb + (size_t)l * A + a0);
                }
                pq4 = GRAN ? pf_pq4 : st_ld4(pqb + a0);      // (granule form: A <= 256, so a0 == lane * 4)
                v4 = st_ld4(a.v + a0);
            } else {
#pragma unroll
                for (int j = 0; j < AT_LP; ++j) {
                    const int l = l0 + j < L ? l0 + j : L - 1;
                    pm4[j] = st_ld4_guard(pmb + (size_t)l * A + a0, rem);
                }
                pq4 = st_ld4_guard(pqb + a0, rem);
                v4 = st_ld4_guard(a.v + a0, rem);   // zero beyond A: those dims add nothing below
            }
            f32x4 loc[AT_LP];
#pragma unroll
            for (int j = 0; j < AT_LP; ++j) loc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef AT_ABLATE_FLOOP   // (tools/mb ablation switches; never defined in the product build)
#pragma unroll 4
            for (int f = 0; f < (PART == 2 ? 0 : F); ++f) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(Wt + f * o.wt_ld + a0);
                // l0 is a multiple of 6: 8-byte aligned, three ds_read_b64 cover the six positions
                typedef __attribute__((ext_vector_type(2))) float f32x2;
                const f32x2 c01 = *reinterpret_cast<const f32x2*>(cf + f * o.cf_ld + (l0 - lb));
                const f32x2 c23 = *reinterpret_cast<const f32x2*>(cf + f * o.cf_ld + (l0 - lb) + 2);
                const f32x2 c45 = *reinterpret_cast<const f32x2*>(cf + f * o.cf_ld + (l0 - lb) + 4);
                const float cvs[AT_LP] = {c01[0], c01[1], c23[0], c23[1], c45[0], c45[1]};
#pragma unroll
                for (int j = 0; j < AT_LP; ++j) {
                    const float cv = cvs[j];
                    loc[j][0] = fmaf(w4[0], cv, loc[j][0]); loc[j][1] = fmaf(w4[1], cv, loc[j][1]);
                    loc[j][2] = fmaf(w4[2], cv, loc[j][2]); loc[j][3] = fmaf(w4[3], cv, loc[j][3]);
                }
            }
#endif
            if (PART == 1) {       // S = pm + W_l cf: one 16-byte store per position (rows past L and dims past A are not written)
#pragma unroll
                for (int j = 0; j < AT_LP; ++j) {
                    if (l0 + j >= L) continue;
                    float* sp = a.s_buf + ((size_t)b * L + l0 + j) * A + a0;
                    const f32x4 sv = {loc[j][0] + pm4[j][0], loc[j][1] + pm4[j][1], loc[j][2] + pm4[j][2], loc[j][3] + pm4[j][3]};
                    if (VEC) *reinterpret_cast<f32x4*>(sp) = sv;
                    else for (int c = 0; c < 4 && c < rem; ++c) sp[c] = sv[c];
                }
                continue;
            }
            // v . tanh(x) with tanh(x) = 1 - 2 / (1 + exp(2x)):  sum_c v_c - 2 sum_c v_c / (1 + exp(2 x_c)) -- per element one
            // v_exp_f32, one v_rcp_f32 and three plain VALU ops (this phase is VALU-issue bound: 43 x 256 tanh per utterance on
            // one CU); exp -> inf gives 1, exp -> 0 gives -1, |error| ~1e-7 absolute like at_tanh
            const float vsum = (v4[0] + v4[1]) + (v4[2] + v4[3]);
#pragma unroll
            for (int j = 0; j < AT_LP; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // (processed_query + processed_loc_feat) + processed_memory, module.py:389-390 (fin part: loc is inside S)
                    const float x = PART == 2 ? pq4[c] + pm4[j][c] : (pq4[c] + loc[j][c]) + pm4[j][c];
                    float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.885390081777927f));
                    if (!VEC) r = c < rem ? r : 0.0f;   // W_l^T pad columns hold garbage (v4 is zero there)
                    acc = fmaf(v4[c], r, acc);
                }
                esum[j] += fmaf(-2.0f, acc, vsum);
            }
        }
        if (PART == 1) return;
        // 8 sums over 64 lanes with 10 shuffles: each butterfly step also halves the number of
        // values a lane carries (instead of 8 independent 6-step reductions)
        static_assert(AT_LB == 8 && AT_LP <= AT_LB && AT_LP % 2 == 0, "the folded reduction below reduces 8 slots per wave");
        {
            const bool hi32 = (lane & 32) != 0, hi16 = (lane & 16) != 0, hi8 = (lane & 8) != 0;
            float r4[4], r2[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float send = hi32 ? esum[j] : esum[j + 4];
                const float keep = hi32 ? esum[j + 4] : esum[j];
                r4[j] = keep + __shfl_xor(send, 32, 64);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float send = hi16 ? r4[j] : r4[j + 2];
                const float keep = hi16 ? r4[j + 2] : r4[j];
                r2[j] = keep + __shfl_xor(send, 16, 64);
            }
            // the last four steps stay inside a row of 16 lanes: DPP permutations instead of LDS-crossbar shuffles
            float r = (hi8 ? r2[1] : r2[0]) + st_dpp<ST_DPP_ROW_ROR8>(hi8 ? r2[0] : r2[1]);
            r = st_oct_sum_dpp(r);
            const int l = l0 + (lane >> 3);      // bits 5,4,3 of the lane select the position
            if ((lane & 7) == 0 && (lane >> 3) < AT_LP && l < L) es[l] = r;
        }
    };
#pragma unroll
    for (int i = 0; i < NPB; ++i) {      // (fin part: pos_lo = 0, pos_hi = L)
        const int l0 = (wave + i * NWV) * AT_LP;
        if (l0 < pos_hi) energy_block(l0, pf_pm4[i]);
    }
    for (int l0 = pos_lo + (wave + NPB * NWV) * AT_LP; l0 < pos_hi; l0 += NWV * AT_LP) energy_block(l0, nullptr);
    if (PART == 1) return;
    AT_PROF(6);
    st_lds_barrier();
    AT_PROF(7);

    // ---- P3: softmax over L (wave 0), write alignment and cumulative weights
    if (wave == 0) {
        float m = -INFINITY;
        for (int l = lane; l < L; l += 64) m = fmaxf(m, es[l]);
        m = st_wave_max_dpp(m);
        float s = 0.0f;
        for (int l = lane; l < L; l += 64) {
            const float ex = expf(es[l] - m);
            es[l] = ex;
            s += ex;
        }
        s = st_wave_sum_dpp(s);
        for (int l = lane; l < L; l += 64) {
            const float w = es[l] / s;
            es[l] = w;
            if (PART == 2 && ipart != 0) continue;      // the other context slices repeat the softmax, one of them writes it
            a.w_out[(size_t)b * a.ld_wout + l] = w;
            const float cum_prev = PART == 2 ? (VEC && l < 64 ? pf_cum : a.w_cum_prev[(size_t)b * L + l]) : hs[o.hl + pad + l];
            a.w_cum_out[(size_t)b * L + l] = w + cum_prev;              // weights + attn_weights_sum, :264
        }
    }
    AT_PROF(8);
    st_lds_barrier();
    AT_PROF(9);

    // ---- P4: context
    if (ctx_active) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < PFR; ++j) {        // rows past L were read as zeros: clamp the weight index, no branch
            const int l = g + j * ng;
            const float wv = es[min(l, L - 1)];
            const float w = l < L ? wv : 0.0f;
            acc[0] = fmaf(w, mpf[j][0], acc[0]); acc[1] = fmaf(w, mpf[j][1], acc[1]);
            acc[2] = fmaf(w, mpf[j][2], acc[2]); acc[3] = fmaf(w, mpf[j][3], acc[3]);
        }
        for (int l = g + PFR * ng; l < L; l += ng) {
            const f32x4 m4 = st_ld4(memb + (size_t)l * E + e_lo + e4 * 4);
            const float w = es[l];
            acc[0] = fmaf(w, m4[0], acc[0]); acc[1] = fmaf(w, m4[1], acc[1]);
            acc[2] = fmaf(w, m4[2], acc[2]); acc[3] = fmaf(w, m4[3], acc[3]);
        }
        *reinterpret_cast<f32x4*>(part + (size_t)(g * ne4 + e4) * 4) = acc;
    }
    AT_PROF(10);
    st_lds_barrier();
    AT_PROF(11);
    for (int e = tid; e < Es; e += NT) {
        float s = 0.0f;
        for (int gg = 0; gg < ng; ++gg) s += part[gg * Es + e];
        if (a.ctx) a.ctx[(size_t)b * a.ld_ctx + e_lo + e] = s;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (a.ctx_dst[d].base) a.ctx_dst[d].base[at_t16_off(b, a.ctx_dst[d].kb0 * 16 + e_lo + e, a.ctx_dst[d].kb_stride)] = s;
    }
    AT_PROF(12);
}


// (leading scalar arguments: handed over in user SGPRs at wave launch with -mllvm -amdgpu-kernarg-preload-count=16, so the
// operand requests at kernel entry do not wait for a scalar-cache round trip on the argument block)
template <bool VEC, int PART, int NT = AT_THREADS>
__global__ __launch_bounds__(NT) void at_kernel(const float* pq, const float* pm, const float* v, const float* w_cum_prev, const float* memory,
                                                float* s_buf, const int L, const int A, const int E, const int fin_parts, const AtArgs rest) {
    extern __shared__ __attribute__((aligned(16))) float at_lds[];
    AtArgs a = rest;
    a.pq = pq; a.pm = pm; a.v = v; a.w_cum_prev = w_cum_prev; a.memory = memory; a.s_buf = s_buf; a.L = L; a.A = A; a.E = E; a.fin_parts = fin_parts;
    at_body<VEC, PART, NT>(a, blockIdx.x, at_lds);
}

}  // namespace
