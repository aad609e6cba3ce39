// attention_bwd_body.h -- the attention-step backward as a device function: one workgroup (512 threads) per utterance.
// Used by ab_kernel (attention_bwd.hip) and, as extra workgroups of the decoder cell's dgates . W^T product, by pk_pw_ab_kernel
// (skinny_packed.hip): in the BPTT loop the decoder cell's recurrence does not depend on the attention / query-cell chain of the
// same step, so its product of step t-1 runs BESIDE the attention backward of step t (32 workgroups bound by one compute unit
// each next to 320 workgroups that stream 42 MB: 18.4 us per pair against 28.9 back to back, tools/mb/mb_overlap).
// See attention_bwd.hip for the math.
#pragma once
#include "st_common.h"

#ifndef AB_PROF
#define AB_PROF(n)   // phase timestamps, only defined by tools/mb/mb_attn_bwd.hip
#endif

namespace {


constexpr int AB_THREADS = 512;
// positions per block of the energy-gradient phase: 48 (three MFMA row tiles per pass: the usual utterance of <= 48 positions is ONE
// block -- three barriers instead of nine, 24 independent positions per thread in the tanh phase) when the forward kept S and the
// block fits the LDS, else 16
constexpr int AB_LBLK_MAX = 48;
constexpr int AB_FMAX = 32;     // location filters held in registers per thread
constexpr int AB_NDCTX = 6;     // addends of the context gradient: three (dxo, dxd, dxq) or, with dxq as K-split slabs, two + four

struct AbArgs {
    const float* pq; const float* pm; const float* memory;
    const float* w_prev; int ld_wprev;     // w_{t-1} (B rows, stride ld_wprev); NULL = zeros (t = 0)
    const float* w_cum_prev;               // cum_{t-1} (B, L)
    const float* w; int ld_w;              // w_t
    const float* loc_conv_w; const float* loc_lin_w; const float* v;
    const float* dctx[AB_NDCTX]; int ld_dctx[AB_NDCTX];  // gradient w.r.t. ctx_t = sum of up to six addends, added in index order (NULL = absent)
    const float* dw_direct[3]; int ld_dw[3];   // gradient w.r.t. w_t: up to three addends (B rows each)
    float* dcum; const float* dcum_add; int ld_dcum_add;   // dL/dcum_t = dcum (B,L, in/out) + dcum_add; also an addend of dw
    float* dpq;                            // (B, A) out
    float* dpq_t16; int dpq_kbs, dpq_kb0;  // optional second copy in the T16 tile layout (operand of the packed W_q^T product)
    float* dhist;                          // (B, 2, L) out: gradient w.r.t. [w_{t-1} ; cum_{t-1}] through the conv
    // tape slices of this step (all written, never read back here)
    float* ds_t;      // (B, L, A)  d loss / d s[l][a]              -> dpm = sum_t, dW_l = ds^T loc
    float* loc_t;     // (B, L, F)  location features               -> dW_l
    float* dloc_t;    // (B, L, F)  gradient of the location features -> dW_c (conv weight gradient)
    float* hist_t;    // (B, L, 2)  [w_{t-1}, cum_{t-1}] channels-last  -> dW_c
    float* dctx_t;    // (B, E)     total gradient w.r.t. ctx_t      -> dmem[b] = w^T dctx
    float* dv_t;      // (B, A)     sum_l de[l] * tanh(s[l][a])      -> dv = sum over (t, b)
    const float* s_in; // optional (B, L, A): pm + W_l loc of this step from the forward (then loc is neither recomputed nor written)
    int B, L, A, E, F, K;
    float* dloc_part;  // NS > 1 parts per utterance (ab_body<.., NS>): (NS, B, L, F) partial location-feature gradients, summed by ab_hist_body
};

// What is left of the step when NS workgroups share an utterance (each takes A / NS attention dims): the partial dloc of the parts summed
// in part order -> dloc_t tape slice, the history tape slice, the gradient w.r.t. the history through the location conv (P5) and the
// carried cum gradient dcum += dhist[1].  One workgroup per utterance, hosted by the NEXT launch of the BPTT step (the W_q^T dpq product).
struct AbHistArgs {
    const float* dloc_part; int parts;
    const float* loc_conv_w;
    const float* w_prev; int ld_wprev; const float* w_cum_prev;
    float* dloc_t; float* hist_t; float* dhist; float* dcum;
    int B, L, F, K;
};

// tanh from one v_exp_f32 and one fast reciprocal (same as the forward kernel): |error| <= ~2e-7 absolute
__device__ __forceinline__ float ab_tanh(float x) {
    const float t = __expf(-2.0f * fabsf(x));
    return copysignf(__fdividef(1.0f - t, 1.0f + t), x);
}

struct AbLds { int hist, hl, wct, f4, wl, wlt, ld, loc, dloc, w, dw, dctx, dsb, part, red, total; };

// lean (a part of a split step, ab_body<.., NS > 1>): no history, W_c, loc or dloc images -- two such workgroups (or one beside a
// product workgroup of the hosting launch) then fit one compute unit's LDS
__host__ __device__ inline AbLds ab_layout(int L, int A, int E, int F, int K, int lblk, bool has_s, bool lean = false) {
    AbLds o;
    int p = 0;
    o.f4 = (F + 3) & ~3;                  // rows of loc / dloc / W_l / transposed W_c padded to float4
    o.hl = (L + K + 4 + 3) & ~3;          // zero-padded history per channel
    o.hist = p; p += lean ? 0 : 2 * o.hl;
    o.wct = p; p += lean ? 0 : 2 * K * o.f4;   // W_c transposed to [c][k][f]
    o.wl = p; p += has_s ? 0 : A * o.f4;  // W_l [a][f] (only to recompute S)
    o.ld = ((A + 63) & ~63) + 4;          // row stride of the two MFMA operands: a padded to 64, +4 floats against bank conflicts
    o.wlt = p; p += 32 * o.ld;            // W_l^T [f][a] (rows f >= F and columns a >= A are zero)
    o.loc = p; p += lean ? 0 : L * AB_FMAX;    // rows padded to AB_FMAX zeros: the energy phase reads them unconditionally
    o.dloc = p; p += lean ? 0 : L * o.f4;
    o.w = p; p += (L + 3) & ~3;
    o.dw = p; p += (L + 3) & ~3;
    o.dctx = p; p += (E + 3) & ~3;
    o.dsb = p; p += (lblk * o.ld > 2 * AB_THREADS ? lblk * o.ld : 2 * AB_THREADS);   // ds block [l][a]; also the fold buffer of P4
    o.part = p; p += 4 * lblk * 32;       // partial dloc sums [a quarter][l][f lane]
    o.red = p; p += 16;
    o.total = p;
    return o;
}

// HAS_S: S = pm + W_l loc of the step comes from the forward pass (training keeps it: 1.4 MB per step against 288 GB) -- no
// location conv (P1) and no 32-filter product per (position, dim) in the energy gradient (P3)
// NS > 1 (the BPTT loop's hosted launch; needs S): NS workgroups per utterance, part `part` takes the attention dims [part A / NS,
// (part + 1) A / NS) of the energy gradient (P3: its tanh phase and its ds . W_l product shrink by NS; ds_t / dpq / dv_t are written per
// dim, so nothing needs a sum there), stages only its rows of W_l, writes its PARTIAL dloc to a.dloc_part and leaves everything behind
// the sum of the partials (dloc_t, hist_t, the conv-transpose P5, the carried dcum) to ab_hist_body in the next launch.  The staging of
// the addends, dw = dctx . mem (P2) and the softmax backward are replicated in every part.  a.dcum is then read-only (= the total
// gradient w.r.t. cum_t, maintained by ab_hist_body) and a.dcum_add must be NULL.
template <bool HAS_S, int LBLK, int NS = 1>
__device__ __forceinline__ void ab_body(const AbArgs& a, const int b, float* __restrict__ lds, const int prt = 0) {
    static_assert(NS == 1 || HAS_S, "parts need the forward's S");
    static_assert(NS == 1 || NS == 2 || NS == 4, "1, 2 or 4 parts");
    constexpr int AB_LBLK = LBLK, AB_LPT = LBLK / (2 * NS);   // positions per thread and block (at least 2 NS threads share an attention dim)
    constexpr int MTL = LBLK / 16;                      // MFMA row tiles per block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.L, A = a.A, E = a.E, F = a.F, K = a.K;
    const int As = A / NS, a_lo = prt * As;            // this part's attention dims (NS == 1: all of them)
    const AbLds o = ab_layout(L, As, E, F, K, LBLK, HAS_S, NS > 1);
    float* hist = lds + o.hist; float* WcT = lds + o.wct; float* Wl = lds + o.wl; float* WlT = lds + o.wlt; float* loc = lds + o.loc;
    float* dloc = lds + o.dloc; float* ws = lds + o.w; float* dws = lds + o.dw; float* dctx = lds + o.dctx;
    float* dsb = lds + o.dsb; float* part = lds + o.part; float* red = lds + o.red;
    const int pad = (K - 1) / 2, HL = o.hl, F4 = o.f4, LD = o.ld;

    AB_PROF(0);
    // energy-gradient role of this thread: fixed attention dim a0, positions l0 + grp, l0 + grp + ngrp, ...
    const int a_l = tid % As, a0 = a_lo + a_l, grp = tid / As, ngrp = AB_THREADS / As;    // host: A <= 256 and 512 % A == 0, so ngrp >= 2 NS
    const float* __restrict__ pmb = (HAS_S ? a.s_in : a.pm) + (size_t)b * L * A;
    // processed-memory values of the first block: issued before anything else, consumed in P3
    // (buffer loads: the descriptor ends at row L, rows past it and slots past the block read zeros without a branch)
    const __amdgpu_buffer_rsrc_t pm_rs = __builtin_amdgcn_make_buffer_rsrc((void*)pmb, 0, L * A * 4, 0x00020000);
    float pmr[AB_LPT];
#pragma unroll
    for (int i = 0; i < AB_LPT; ++i) {
        const int l = grp + i * ngrp;
        pmr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pm_rs, i * ngrp + grp < AB_LBLK ? (l * A + a0) * 4 : 0x7ffffff0, 0, 0));
    }
    // encoder-memory rows of the first round of P2 (wave w: positions w, w+8, w+16, w+24), issued now, consumed after P1
    constexpr int NW = AB_THREADS / 64;
    const float* __restrict__ memb = a.memory + (size_t)b * L * E;
    const bool mem_pf = E <= 512;
    const __amdgpu_buffer_rsrc_t mem_rs = __builtin_amdgcn_make_buffer_rsrc((void*)memb, 0, L * E * 4, 0x00020000);
    // (the split form has registers to spare: six rounds = 48 positions, the whole usual utterance -- a position past the prefetch
    // costs an exposed global round trip in P2)
    constexpr int MPF = NS > 1 ? 6 : 4;
    f32x4 mpf[MPF][2];
#pragma unroll
    for (int j = 0; j < MPF; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int l = wave + j * NW, e = lane * 4 + h * 256;
            mpf[j][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mem_rs, (mem_pf && e < E) ? (l * E + e) * 4 : 0x7ffffff0, 0, 0));
        }
    // addends of dw[l] / dctx[e], first round (l = tid, e = tid): requested with the other operands, consumed after the staging
    // (absent addends read a valid dummy address and are dropped by a select when consumed)
    const int pl_l = min(tid, L - 1), pl_e = min(tid, E - 1);
    const float* pl_dummy = a.w + (size_t)b * a.ld_w + pl_l;
    const float* pe_dummy = a.memory + (size_t)b * L * E + pl_e;
    const float pl_w = pl_dummy[0];
    const float pq_a = a.pq[(size_t)b * A + a0], v_a = a.v[a0];      // (consumed in P3: requested here, not behind the softmax)
    float pl_dl[3], pe_dl[AB_NDCTX];
#pragma unroll
    for (int j = 0; j < 3; ++j) pl_dl[j] = (a.dw_direct[j] ? a.dw_direct[j] + (size_t)b * a.ld_dw[j] + pl_l : pl_dummy)[0];
    const float pl_gc0 = (a.dcum ? a.dcum + (size_t)b * L + pl_l : pl_dummy)[0];
    const float pl_gc1 = (a.dcum && a.dcum_add ? a.dcum_add + (size_t)b * a.ld_dcum_add + pl_l : pl_dummy)[0];
#pragma unroll
    for (int j = 0; j < AB_NDCTX; ++j) pe_dl[j] = (a.dctx[j] ? a.dctx[j] + (size_t)b * a.ld_dctx[j] + pl_e : pe_dummy)[0];
    // ---- P0: stage operands.  The first round of the weight loads goes to registers before any LDS traffic so
    // that all global latencies of this phase overlap (each separate load -> store loop costs one round trip).
    const int nWc = NS == 1 ? F * 2 * K : 0, nWl = As * F;       // (parts: W_c is only needed behind the sum of the partials)
    const float* __restrict__ wl_src = a.loc_lin_w + (size_t)a_lo * F;      // rows [a_lo, a_lo + As) of W_l (A, F)
    constexpr int WLR = 4 / NS;                                  // 16-byte register rounds of the W_l rows (F == 32, A <= 256)
    float wc_v[4], wl_v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int i = tid + j * AB_THREADS; wc_v[j] = i < nWc ? a.loc_conv_w[i] : 0.0f; }
    // the usual shape (32 filters, 16-byte aligned W_l): four 16-byte loads per thread and shift / mask indexing below instead of
    // sixteen scalar loads and a division by F per element (this prologue is instruction-issue bound)
    const bool wl_fast = F == AB_FMAX && nWl <= 4 * WLR * AB_THREADS && st_aligned16(wl_src);
    if (wl_fast) {
#pragma unroll
        for (int j = 0; j < WLR; ++j) {
            const int i4 = tid + j * AB_THREADS;
            const f32x4 t = st_ld4(wl_src + (size_t)min(i4, (nWl >> 2) - 1) * 4);
            wl_v[4 * j] = t[0]; wl_v[4 * j + 1] = t[1]; wl_v[4 * j + 2] = t[2]; wl_v[4 * j + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int i = tid + j * AB_THREADS; wl_v[j] = i < nWl ? wl_src[i] : 0.0f; }
    }
    AB_PROF(8);
    if (NS == 1) for (int i = tid; i < 2 * HL; i += AB_THREADS) {
        const int c = i / HL, j = i - c * HL, l = j - pad;
        float v = 0.0f;
        if (l >= 0 && l < L) {
            if (c == 0) v = a.w_prev ? a.w_prev[(size_t)b * a.ld_wprev + l] : 0.0f;
            else v = a.w_cum_prev[(size_t)b * L + l];
            a.hist_t[((size_t)b * L + l) * 2 + c] = v;
        }
        hist[i] = v;
    }
    AB_PROF(9);
    // zero padding of the LDS operands: pad columns a >= A of the two MFMA operands always; everything else only
    // when F is not the full 32 filters (then whole arrays are cleared before the scattering stores)
    const bool ragged = F != AB_FMAX;
    if (ragged) {
        if (NS == 1) for (int i = tid; i < 2 * K * F4; i += AB_THREADS) WcT[i] = 0.0f;
        if (!HAS_S) for (int i = tid; i < A * F4; i += AB_THREADS) Wl[i] = 0.0f;      // (with S given W_l [a][f] has no LDS copy)
        for (int i = tid; i < 32 * LD; i += AB_THREADS) WlT[i] = 0.0f;
        for (int i = tid; i < AB_LBLK * LD; i += AB_THREADS) dsb[i] = 0.0f;
        if (NS == 1) for (int i = tid; i < L * AB_FMAX; i += AB_THREADS) loc[i] = 0.0f;
        st_lds_barrier();
    } else {
        const int padw = LD - As;
        for (int i = tid; i < (32 + AB_LBLK) * padw; i += AB_THREADS) {
            const int r = i / padw, c = As + (i - r * padw);
            if (r < 32) WlT[r * LD + c] = 0.0f; else dsb[(r - 32) * LD + c] = 0.0f;
        }
    }
    AB_PROF(10);
#pragma unroll
    for (int j = 0; j < 4; ++j) {                               // [c][k][f] <- [f][c][k]
        const int i = tid + j * AB_THREADS;
        if (i < nWc) { const int f = i / (2 * K), ck = i - f * 2 * K; WcT[ck * F4 + f] = wc_v[j]; }
    }
    if (wl_fast) {
#pragma unroll
        for (int j = 0; j < WLR; ++j) {
            const int i4 = tid + j * AB_THREADS;
            if (i4 < (nWl >> 2)) {
                const int aa = i4 >> 3, f0 = (i4 & 7) * 4;         // F == 32: eight float4 per W_l row
                if (!HAS_S) *reinterpret_cast<f32x4*>(Wl + aa * F4 + f0) = f32x4{wl_v[4 * j], wl_v[4 * j + 1], wl_v[4 * j + 2], wl_v[4 * j + 3]};
#pragma unroll
                for (int c = 0; c < 4; ++c) WlT[(f0 + c) * LD + aa] = wl_v[4 * j + c];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = tid + j * AB_THREADS;
            if (i < nWl) { const int aa = i / F, f = i - aa * F; if (!HAS_S) Wl[aa * F4 + f] = wl_v[j]; WlT[f * LD + aa] = wl_v[j]; }
        }
    }
    for (int i = tid + 4 * AB_THREADS; i < nWc; i += AB_THREADS) {      // sizes beyond the register rounds
        const int f = i / (2 * K), ck = i - f * 2 * K;
        WcT[ck * F4 + f] = a.loc_conv_w[i];
    }
    for (int i = tid + 16 * AB_THREADS; i < nWl; i += AB_THREADS) {
        const int aa = i / F, f = i - aa * F;
        const float wv = wl_src[i];
        if (!HAS_S) Wl[aa * F4 + f] = wv;
        WlT[f * LD + aa] = wv;
    }
    AB_PROF(11);
    // absent addends are read from a valid dummy address and dropped by a select afterwards: a chain of
    // `if (p) g += p[i]` makes the wave wait for every load in turn (one memory round trip per addend)
    for (int l = tid; l < L; l += AB_THREADS) {
        const bool first = l == tid;           // requested at the top of the kernel
        const float* dummy = a.w + (size_t)b * a.ld_w + l;
        const float wv = first ? pl_w : dummy[0];
        float dl[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) dl[j] = first ? pl_dl[j] : (a.dw_direct[j] ? a.dw_direct[j] + (size_t)b * a.ld_dw[j] + l : dummy)[0];
        const float gc0 = first ? pl_gc0 : (a.dcum ? a.dcum + (size_t)b * L + l : dummy)[0];
        const float gc1 = first ? pl_gc1 : (a.dcum && a.dcum_add ? a.dcum_add + (size_t)b * a.ld_dcum_add + l : dummy)[0];
        ws[l] = wv;
        float g = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) g += a.dw_direct[j] ? dl[j] : 0.0f;
        if (a.dcum) {   // cum_t = cum_{t-1} + w_t: the total gradient w.r.t. cum_t reaches w_t and is carried to cum_{t-1}
            const float gc = gc0 + (a.dcum_add ? gc1 : 0.0f);
            if (NS == 1) a.dcum[(size_t)b * L + l] = gc;      // (parts: dcum already is the total, kept by ab_hist_body)
            g += gc;
        }
        dws[l] = g;
    }
    for (int e = tid; e < E; e += AB_THREADS) {
        const bool first = e == tid;
        const float* dummy = a.memory + (size_t)b * L * E + e;
        float dl[AB_NDCTX];
#pragma unroll
        for (int j = 0; j < AB_NDCTX; ++j) dl[j] = first ? pe_dl[j] : (a.dctx[j] ? a.dctx[j] + (size_t)b * a.ld_dctx[j] + e : dummy)[0];
        float g = 0.0f;
#pragma unroll
        for (int j = 0; j < AB_NDCTX; ++j) g += a.dctx[j] ? dl[j] : 0.0f;
        dctx[e] = g;
        if (NS == 1 || prt == 0) a.dctx_t[(size_t)b * E + e] = g;
    }
    st_lds_barrier();
    // W_l row of this thread in registers (for s = pq + pm + W_l loc)
    float wl_r[AB_FMAX];
#pragma unroll
    for (int f = 0; f < AB_FMAX; ++f) wl_r[f] = 0.0f;
    if (HAS_S) {
    } else if (F4 == AB_FMAX) {
#pragma unroll
        for (int f = 0; f < AB_FMAX; f += 4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(Wl + a0 * F4 + f);
            wl_r[f] = w4[0]; wl_r[f + 1] = w4[1]; wl_r[f + 2] = w4[2]; wl_r[f + 3] = w4[3];
        }
    } else {
#pragma unroll
        for (int f = 0; f < AB_FMAX; ++f) if (f < F) wl_r[f] = Wl[a0 * F4 + f];
    }

    AB_PROF(1);
    // ---- P1: location features loc[l][f]: one thread = one filter x 4 consecutive positions, sliding window
    if (!HAS_S) {
        const int nlb = (L + 3) >> 2;
        for (int i = tid; i < nlb * F4; i += AB_THREADS) {
            const int f = i % F4, l0 = (i / F4) * 4;
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
            if (f < F) {
                for (int c = 0; c < 2; ++c) {
                    const float* hr = hist + c * HL + l0;
                    const float* wr = WcT + (size_t)c * K * F4 + f;
                    float h0 = hr[0], h1 = hr[1], h2 = hr[2];
#pragma unroll 4
                    for (int k = 0; k < K; ++k) {
                        const float h3 = hr[k + 3], wv = wr[k * F4];
                        acc0 = fmaf(wv, h0, acc0); acc1 = fmaf(wv, h1, acc1); acc2 = fmaf(wv, h2, acc2); acc3 = fmaf(wv, h3, acc3);
                        h0 = h1; h1 = h2; h2 = h3;
                    }
                }
            }
            const float r[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (l0 + j < L) {
                    loc[(l0 + j) * AB_FMAX + f] = r[j];
                    if (f < F) a.loc_t[((size_t)b * L + l0 + j) * F + f] = r[j];
                }
        }
    }
    AB_PROF(2);
    // ---- P2: dw[l] += dctx . mem[l]     (one wave per position, lanes over E)
    {
        int lstart = wave;
        if (mem_pf) {
            f32x4 d4[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) { const int e = lane * 4 + h * 256; d4[h] = e < E ? *reinterpret_cast<const f32x4*>(dctx + e) : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < MPF; ++j) {
                const int l = wave + j * NW;
                float acc = 0.0f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    acc = fmaf(d4[h][0], mpf[j][h][0], acc); acc = fmaf(d4[h][1], mpf[j][h][1], acc);
                    acc = fmaf(d4[h][2], mpf[j][h][2], acc); acc = fmaf(d4[h][3], mpf[j][h][3], acc);
                }
                acc = st_wave_sum_dpp(acc);
                if (lane == 0 && l < L) dws[l] += acc;
            }
            lstart = wave + MPF * NW;
        }
        for (int l = lstart; l < L; l += 2 * NW) {          // remaining positions: two rows' loads in flight per wave
            const int l2 = l + NW;
            float acc = 0.0f, acc2 = 0.0f;
            for (int e = lane * 4; e < E; e += 256) {      // E % 4 == 0 checked on the host
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dctx + e);
                const f32x4 m4 = st_ld4(memb + (size_t)l * E + e);
                f32x4 n4 = {0.f, 0.f, 0.f, 0.f};
                if (l2 < L) n4 = st_ld4(memb + (size_t)l2 * E + e);
                acc = fmaf(d4[0], m4[0], acc); acc = fmaf(d4[1], m4[1], acc); acc = fmaf(d4[2], m4[2], acc); acc = fmaf(d4[3], m4[3], acc);
                acc2 = fmaf(d4[0], n4[0], acc2); acc2 = fmaf(d4[1], n4[1], acc2); acc2 = fmaf(d4[2], n4[2], acc2); acc2 = fmaf(d4[3], n4[3], acc2);
            }
            acc = st_wave_sum_dpp(acc);
            acc2 = st_wave_sum_dpp(acc2);
            if (lane == 0) { dws[l] += acc; if (l2 < L) dws[l2] += acc2; }
        }
    }
    st_lds_barrier();
    AB_PROF(3);
    // softmax backward: de[l] = w[l] * (dw[l] - sum_j w[j] dw[j])
    if (wave == 0) {
        float acc = 0.0f;
        for (int l = lane; l < L; l += 64) acc = fmaf(ws[l], dws[l], acc);
        acc = st_wave_sum_dpp(acc);
        if (lane == 0) red[0] = acc;
    }
    st_lds_barrier();
    const float dot = red[0];
    st_lds_barrier();
    for (int l = tid; l < L; l += AB_THREADS) dws[l] = ws[l] * (dws[l] - dot);     // dws now holds de
    st_lds_barrier();

    AB_PROF(4);
    // ---- P3: energy gradient in blocks of AB_LBLK positions
    float* __restrict__ dsg = a.ds_t + (size_t)b * L * A;
    float* __restrict__ dlocg = a.dloc_t + (size_t)b * L * F;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 dv2 = {0.f, 0.f}, dpq2 = {0.f, 0.f};          // sums over this thread's even / odd positions
    const int fl = tid & 31;

    for (int l0 = 0; l0 < L; l0 += AB_LBLK) {
        const int lend = min(L, l0 + AB_LBLK);
        // prefetch the next block's processed-memory values while this block computes
        float pmn[AB_LPT];
#pragma unroll
        for (int i = 0; i < AB_LPT; ++i) {
            const int l = l0 + AB_LBLK + grp + i * ngrp;
            pmn[i] = (i * ngrp + grp < AB_LBLK && l < L) ? pmb[(size_t)l * A + a0] : 0.0f;
        }
        // branch-free: rows past the block / past L are computed on a clamped position with de = 0.  Three passes over the thread's
        // positions so that nothing serialises them: (1) all the energy gradients de[l] come out of LDS back to back (inside the loop each
        // read sat behind the previous position's LDS store -- the compiler cannot tell the two arrays apart -- and its latency, the
        // exp and the reciprocal formed one dependent chain per position: 350 cycles each); (2) all s = pq + S; (3) tanh and the outputs
        float gv[AB_LPT], sv[AB_LPT];
#pragma unroll
        for (int i = 0; i < AB_LPT; ++i) {
            const int row = i * ngrp + grp, l = l0 + row;
            gv[i] = dws[min(l, L - 1)];
            if (!(row < AB_LBLK && l < lend)) gv[i] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < AB_LPT; ++i) {
            const int l = min(l0 + grp + i * ngrp, L - 1);
            const float* lr = loc + l * AB_FMAX;
            float sa = pq_a + pmr[i], sb = 0.0f;
#pragma unroll
            for (int f = 0; f < (HAS_S ? 0 : AB_FMAX); f += 4) {          // pad columns are zero on both sides
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lr + f);
                sa = fmaf(wl_r[f], l4[0], sa); sb = fmaf(wl_r[f + 1], l4[1], sb);
                sa = fmaf(wl_r[f + 2], l4[2], sa); sb = fmaf(wl_r[f + 3], l4[3], sb);
            }
            sv[i] = sa + sb;
        }
        // two positions per instruction where the arithmetic allows (packed fp32: v_pk_mul / v_pk_add / v_pk_fma); the exponential and
        // the reciprocal stay one lane-value each.  This phase is bound by instruction issue on ONE compute unit (L x A elements).
        static_assert(AB_LPT % 2 == 0, "positions per thread come in pairs");
#pragma unroll
        for (int i = 0; i < AB_LPT; i += 2) {
            const f32x2 s2 = {sv[i], sv[i + 1]}, g2 = {gv[i], gv[i + 1]};
            const f32x2 e2 = f32x2{fabsf(s2[0]), fabsf(s2[1])} * -2.885390081777927f;          // exp(-2|x|) = exp2(-2 log2(e) |x|)
            const f32x2 t2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
            const f32x2 num = 1.0f - t2, den = 1.0f + t2;
            f32x2 th2 = num * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
            th2 = f32x2{copysignf(th2[0], s2[0]), copysignf(th2[1], s2[1])};
            dv2 = dv2 + g2 * th2;
            const f32x2 ds2 = (g2 * v_a) * (1.0f - th2 * th2);
            dpq2 = dpq2 + ds2;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int row = (i + ii) * ngrp + grp, l = l0 + row;
                const bool in_blk = row < AB_LBLK, valid = in_blk && l < lend;
#ifndef AB_ABLATE_DSG      // (tools/mb only: phase timing without the tape stores)
                if (valid) dsg[(size_t)l * A + a0] = ds2[ii];
#endif
                if (in_blk) dsb[row * LD + a_l] = ds2[ii];           // zero for rows past L
            }
        }
#pragma unroll
        for (int i = 0; i < AB_LPT; ++i) pmr[i] = pmn[i];
        AB_PROF(12);
        st_lds_barrier();
        AB_PROF(13);
        // dloc[l][f] = sum_a ds[l][a] * W_l[a][f] on the matrix cores: wave = (filter tile nt of 16, quarter kq of the
        // padded a range); exact-fp32 16x16x4 MFMAs, each lane's float4 along a feeds four of them (same k order
        // for both operands), 16 x 16 partial tiles to LDS, summed below
        {
            const int nt = wave & 1, kq = wave >> 1;
            const int APq = (LD - 4) >> 2;                      // a range of one quarter (multiple of 16)
            f32x4 acc[MTL];
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (nt * 16 < F4) {
                const float* ap = dsb + (lane & 15) * LD + kq * APq + 4 * (lane >> 4);
                const float* bp = WlT + (nt * 16 + (lane & 15)) * LD + kq * APq + 4 * (lane >> 4);
                for (int kc = 0; kc < APq; kc += 16) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + kc);       // one W_l fragment feeds every row tile
#pragma unroll
                    for (int mt = 0; mt < MTL; ++mt) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + mt * 16 * LD + kc);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c], b4[c], acc[mt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(kq * AB_LBLK + mt * 16 + 4 * (lane >> 4) + r) * 32 + nt * 16 + (lane & 15)] = acc[mt][r];
        }
        AB_PROF(14);
        st_lds_barrier();
#pragma unroll
        for (int rr = 0; rr < MTL; ++rr) {
            const int ll = rr * 16 + (tid >> 5);          // 16 positions x 32 filter lanes = 512 threads per pass
            const float sum = part[(0 * AB_LBLK + ll) * 32 + fl] + part[(1 * AB_LBLK + ll) * 32 + fl] +
                              part[(2 * AB_LBLK + ll) * 32 + fl] + part[(3 * AB_LBLK + ll) * 32 + fl];
            if (NS > 1) {            // this part's share of dloc: summed with the others' in ab_hist_body
                if (l0 + ll < L && fl < F) a.dloc_part[(((size_t)prt * a.B + b) * L + l0 + ll) * F + fl] = sum;
            } else if (l0 + ll < L && fl < F4) {
                dloc[(l0 + ll) * F4 + fl] = sum;
                if (fl < F) dlocg[(size_t)(l0 + ll) * F + fl] = sum;
            }
        }
        st_lds_barrier();
    }
    AB_PROF(5);
    // ---- P4: fold the per-thread sums of the ngrp threads sharing an attention dim (LDS, fixed order)
    float* fold = dsb;
    fold[tid] = dv2[0] + dv2[1]; fold[AB_THREADS + tid] = dpq2[0] + dpq2[1];
    st_lds_barrier();
    if (grp == 0) {
        float sv = 0.0f, sp = 0.0f;
        for (int gq = 0; gq < ngrp; ++gq) { sv += fold[gq * As + a_l]; sp += fold[AB_THREADS + gq * As + a_l]; }
        a.dv_t[(size_t)b * A + a0] = sv;
        a.dpq[(size_t)b * A + a0] = sp;
        if (a.dpq_t16) {
            const int k = a.dpq_kb0 * 16 + a0;
            a.dpq_t16[(((size_t)(b >> 4) * a.dpq_kbs + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (b & 15)) * 4 + (k & 3)] = sp;
        }
    }
    AB_PROF(6);
    // ---- P5: gradient w.r.t. the attention history through the location conv
    // dhist[c][j] = sum_{f,k} dloc[j - k + pad][f] * Wc[f][c][k]; 8 adjacent lanes split the filters in float4 groups
    if (NS > 1) { AB_PROF(7); return; }
    float* dh = a.dhist + (size_t)b * 2 * L;
    for (int base = 0; base < 2 * L; base += AB_THREADS / 8) {
        const int cj = base + (tid >> 3), q = tid & 7;
        float acc = 0.0f;
        if (cj < 2 * L) {
            const int c = cj / L, j = cj - c * L;
            const int klo = max(0, j + pad - (L - 1)), khi = min(K - 1, j + pad);     // 0 <= j - k + pad < L
            for (int f = q * 4; f < F4; f += 32) {
                const float* wr = WcT + (size_t)c * K * F4 + f;
#pragma unroll 4
                for (int k = klo; k <= khi; ++k) {
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(dloc + (j - k + pad) * F4 + f);
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + k * F4);
                    acc = fmaf(d4[0], w4[0], acc); acc = fmaf(d4[1], w4[1], acc);
                    acc = fmaf(d4[2], w4[2], acc); acc = fmaf(d4[3], w4[3], acc);
                }
            }
        }
        acc = st_oct_sum_dpp(acc);      // (uniform loop: all lanes active)
        if (cj < 2 * L && q == 0) dh[cj] = acc;
    }
    AB_PROF(7);
}

// LDS image of ab_hist_body: W_c as [c][k + 3][f] with three zero taps in front and behind, dloc as [l + AB_HZ][f] between zero rows
// (AB_HZ in front, AB_HZ + 4 behind) -- the sliding window of the conv-transpose then runs without a single bounds test.  Rows of dloc are
// F4 + 4 floats apart (consecutive rows on different bank groups).
constexpr int AB_HZ = 24;      // >= (K - 1) / 2 + 1 + AB_HU zero rows in front of dloc (host: K <= 31)
constexpr int AB_HU = 6;       // rows of the sliding window requested from LDS together
__host__ __device__ inline int ab_hist_lds_floats(int L, int F, int K) {
    const int F4 = (F + 3) & ~3;
    return 2 * (K + 6 + AB_HU) * F4 + (L + 2 * AB_HZ + 4) * (F4 + 4);
}

// See AbHistArgs.  dloc = part 0 + part 1 (+ ...) in part order; dhist[c][j] = sum_{f,k} dloc[j - k + pad][f] * Wc[f][c][k] as a sliding
// window in registers: a thread owns 4 filters and 4 consecutive positions j0 .. j0 + 3 of BOTH channels and walks the dloc rows
// j0 + 3 + pad down to j0 + pad - (K - 1): per row ONE dloc fragment and one new weight fragment per channel for 32 multiply-adds (the
// per-output form of ab_body reads two fragments per 4 multiply-adds: 680 KB through a 128 B/clk LDS, 6-9k cycles -- measured).  The 8
// lanes of a position block (filter groups) are then summed by DPP.
__device__ __forceinline__ void ab_hist_body(const AbHistArgs& h, const int b, float* __restrict__ lds) {
    const int tid = threadIdx.x;
    const int L = h.L, F = h.F, K = h.K, F4 = (F + 3) & ~3, pad = (K - 1) / 2;
    const int DLD = F4 + 4, WK = K + 6 + AB_HU;
    float* WcT = lds;                     // [c][k + 3][f]
    float* dlz = lds + 2 * WK * F4;       // [l + AB_HZ][f]
    float* dloc = dlz + AB_HZ * DLD;
    const int nWc = F * 2 * K, nD = L * F;
    AB_PROF(0);
    // every global operand of the first rounds is requested before anything is consumed (clamped addresses, selects afterwards)
    float wc_v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wc_v[j] = h.loc_conv_w[min(tid + j * AB_THREADS, nWc - 1)];
    const size_t pstride = (size_t)h.B * L * F;
    const float* p0 = h.dloc_part + (size_t)b * L * F;
    constexpr int DR = 3;                 // rounds of the partial sums held in registers (L F <= 1536: the usual 43 x 32)
    float pv[DR][4];
#pragma unroll
    for (int r = 0; r < DR; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[r][q] = p0[(size_t)(q < h.parts ? q : 0) * pstride + min(tid + r * AB_THREADS, nD - 1)];
    const int l_h = min(tid, L - 1);
    const float hw = (h.w_prev ? h.w_prev + (size_t)b * h.ld_wprev : h.w_cum_prev + (size_t)b * L)[l_h];
    const float hc = h.w_cum_prev[(size_t)b * L + l_h];
    // the carried cum gradient of this thread's output block (consumed at the very end)
    const int fg = tid & 7, ch = (tid >> 3) & 1, jb = tid >> 4, j0 = jb * 4;      // 4 filters x 1 channel x 4 positions per thread
    float dc_old[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dc_old[jj] = (h.dcum ? h.dcum : h.w_cum_prev)[(size_t)b * L + min(j0 + jj, L - 1)];
    AB_PROF(1);
    // zero frame: the six pad taps of W_c, the pad rows of dloc (and everything when F is ragged)
    // (the whole image: a linear 16-byte clear costs less than picking the pad rows out with divisions)
    for (int i = tid; i < ab_hist_lds_floats(L, F, K) / 4; i += AB_THREADS) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    st_lds_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {                               // [c][k + 3][f] <- [f][c][k]
        const int i = tid + j * AB_THREADS;
        if (i < nWc) { const int f = i / (2 * K), ck = i - f * 2 * K, c = ck / K, k = ck - c * K; WcT[(c * WK + k + 3) * F4 + f] = wc_v[j]; }
    }
    AB_PROF(2);
    for (int i = tid + 4 * AB_THREADS; i < nWc; i += AB_THREADS) {
        const int f = i / (2 * K), ck = i - f * 2 * K, c = ck / K, k = ck - c * K;
        WcT[(c * WK + k + 3) * F4 + f] = h.loc_conv_w[i];
    }
    float* dlt = h.dloc_t + (size_t)b * L * F;
#pragma unroll
    for (int r = 0; r < DR; ++r) {
        const int i = tid + r * AB_THREADS;
        float sum = pv[r][0];
#pragma unroll
        for (int q = 1; q < 4; ++q) sum += q < h.parts ? pv[r][q] : 0.0f;
        if (i < nD) { const int l = i / F, f = i - l * F; dloc[l * DLD + f] = sum; dlt[i] = sum; }
    }
    for (int i = tid + DR * AB_THREADS; i < nD; i += AB_THREADS) {
        float sum = p0[i];
        for (int q = 1; q < h.parts; ++q) sum += p0[(size_t)q * pstride + i];
        const int l = i / F, f = i - l * F;
        dloc[l * DLD + f] = sum; dlt[i] = sum;
    }
    // the history tape slice [w_{t-1}, cum_{t-1}] channels-last
    for (int l = tid; l < L; l += AB_THREADS) {
        const bool first = l == tid;
        const float v0 = h.w_prev ? (first ? hw : h.w_prev[(size_t)b * h.ld_wprev + l]) : 0.0f;
        const float v1 = first ? hc : h.w_cum_prev[(size_t)b * L + l];
        *reinterpret_cast<float2*>(h.hist_t + ((size_t)b * L + l) * 2) = float2{v0, v1};
    }
    AB_PROF(3);
    st_lds_barrier();
    AB_PROF(4);
    float* dh = h.dhist + (size_t)b * 2 * L;
    for (int base = 0; base < L; base += (AB_THREADS / 16) * 4) {       // 32 position blocks of 4 per pass: L <= 128 in one
        const int jq = base + j0;
        f32x4 acc[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (jq < L) {
            for (int f = fg * 4; f < F4; f += 32) {
                // row r = jq + 3 + pad - s (s = 0 .. K + 2) meets output jq + jj with tap k = jj + s - 3: the weight window w[jj] holds the taps
                // s - 3 .. s (zero taps outside [0, K)), shifted by one per row
                const float* dr = dloc + (jq + 3 + pad) * DLD + f;
                const float* w0 = WcT + ch * WK * F4 + f;
                f32x4 wa[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) wa[jj] = *reinterpret_cast<const f32x4*>(w0 + jj * F4);
                // AB_HU rows per batch, all their fragments requested before the first multiply-add (one LDS round trip per batch instead of
                // one per row: the three busy waves cannot hide it); the rows / taps a last batch reads past the window are zeros
                for (int s0 = 0; s0 < K + 3; s0 += AB_HU) {
                    f32x4 d4[AB_HU], na[AB_HU];
#pragma unroll
                    for (int u_ = 0; u_ < AB_HU; ++u_) {
                        d4[u_] = *reinterpret_cast<const f32x4*>(dr - (s0 + u_) * DLD);
                        na[u_] = *reinterpret_cast<const f32x4*>(w0 + (s0 + u_ + 4) * F4);
                    }
#pragma unroll
                    for (int u_ = 0; u_ < AB_HU; ++u_) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) acc[jj] += d4[u_] * wa[jj];
                        wa[0] = wa[1]; wa[1] = wa[2]; wa[2] = wa[3]; wa[3] = na[u_];
                    }
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            float v = (acc[jj][0] + acc[jj][1]) + (acc[jj][2] + acc[jj][3]);
            v = st_oct_sum_dpp(v);      // (uniform: all lanes active; fixed order)
            const int j = jq + jj;
            if (fg == 0 && j < L) {
                dh[ch * L + j] = v;
                // cum_t = cum_{t-1} + w_t: the gradient w.r.t. cum carries on
                if (ch == 1 && h.dcum) h.dcum[(size_t)b * L + j] = (base == 0 ? dc_old[jj] : h.dcum[(size_t)b * L + j]) + v;
            }
        }
    }
    AB_PROF(5);
}

// argument block of one step from the C-ABI arguments (shared by the plain and the hosted launch); returns 0 or -1 with the error set
inline int ab_fill(AbArgs& a, const st_t16_view* dpq_t16, const float* pq, const float* pm, const float* memory,
                   const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                   const float* loc_conv_w, const float* loc_lin_w, const float* v,
                   const float* const* dctx, const int* ld_dctx, int n_dctx,
                   const float* const* dw_direct, const int* ld_dw, int n_dw,
                   float* dcum, const float* dcum_add, int ld_dcum_add,
                   float* dpq, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                   float* dctx_t, float* dv_t, const float* s_in,
                   int B, int L, int A, int E, int F, int K) {
    ST_CHECK_ARG(pq && pm && memory && w_cum_prev && w && loc_conv_w && loc_lin_w && v, "st_attn_step_bwd: null input");
    ST_CHECK_ARG(dpq && dhist && ds_t && (loc_t || s_in) && dloc_t && hist_t && dctx_t && dv_t, "st_attn_step_bwd: null output");
    ST_CHECK_ARG(B > 0 && L > 0 && A > 0 && E > 0 && F > 0 && K > 0 && (K & 1), "st_attn_step_bwd: bad dims (K must be odd)");
    ST_CHECK_ARG(A <= AB_THREADS / 2 && AB_THREADS % A == 0, "st_attn_step_bwd: attn_dim=%d must divide %d", A, AB_THREADS / 2);
    ST_CHECK_ARG((E & 3) == 0 && st_aligned16(memory), "st_attn_step_bwd: E=%d must be a multiple of 4 (16-byte aligned rows)", E);
    ST_CHECK_ARG(F <= AB_FMAX, "st_attn_step_bwd: n_location_filters=%d > %d", F, AB_FMAX);
    ST_CHECK_ARG(n_dctx >= 0 && n_dctx <= AB_NDCTX && n_dw >= 0 && n_dw <= 3, "st_attn_step_bwd: at most %d context addends, 3 weight addends", AB_NDCTX);
    memset(&a, 0, sizeof(a));
    a.pq = pq; a.pm = pm; a.memory = memory; a.w_prev = w_prev; a.ld_wprev = ld_wprev; a.w_cum_prev = w_cum_prev;
    a.w = w; a.ld_w = ld_w; a.loc_conv_w = loc_conv_w; a.loc_lin_w = loc_lin_w; a.v = v;
    for (int j = 0; j < n_dctx; ++j) { a.dctx[j] = dctx[j]; a.ld_dctx[j] = ld_dctx[j]; }
    for (int j = 0; j < n_dw; ++j) { a.dw_direct[j] = dw_direct[j]; a.ld_dw[j] = ld_dw[j]; }
    a.dcum = dcum; a.dcum_add = dcum_add; a.ld_dcum_add = ld_dcum_add;
    a.dpq = dpq; a.dhist = dhist; a.ds_t = ds_t; a.loc_t = loc_t; a.dloc_t = dloc_t; a.hist_t = hist_t; a.dctx_t = dctx_t; a.dv_t = dv_t;
    a.s_in = s_in;
    if (dpq_t16 && dpq_t16->base) { a.dpq_t16 = dpq_t16->base; a.dpq_kbs = dpq_t16->kb_stride; a.dpq_kb0 = dpq_t16->kb0; }
    a.B = B; a.L = L; a.A = A; a.E = E; a.F = F; a.K = K;
    return 0;
}

// the wide energy-gradient block when the forward kept S and its LDS image (plus `extra` bytes of the host kernel) fits
inline bool ab_wide(const AbArgs& a, size_t extra = 0) {
    return a.s_in && (size_t)ab_layout(a.L, a.A, a.E, a.F, a.K, AB_LBLK_MAX, true).total * sizeof(float) + extra <= 160 * 1024;
}
inline size_t ab_lds_bytes(const AbArgs& a, bool wide, int parts = 1) {
    return (size_t)ab_layout(a.L, a.A / parts, a.E, a.F, a.K, wide ? AB_LBLK_MAX : 16, a.s_in != nullptr, parts > 1).total * sizeof(float);
}

}  // namespace
