// gemm.hip -- fp32 MFMA GEMM / implicit-GEMM conv1d with a fused epilogue, gfx950 (MI355X).
//
// Used for everything on the path that has many rows (once per utterance, not per step):
// encoder Conv1d+BN+ReLU stack, LSTM/GRU input projections, memory_layer, the whole-teacher
// prenet, the CBHG conv bank / projections / highways, and Linear(2*n_mels -> linear_dim).
// Activations are channels-last (rows = (utterance, frame)), so a k-tap conv is a sum over
// taps of row-shifted GEMMs and needs no im2col buffer; utterance boundaries are handled by
// zero-filling rows outside [0, Tin).
//   tile 64(M) x 64(N) x 16(K), 4 wavefronts as 2x2, each 32x32 = 2x2 v_mfma_f32_16x16x4_f32
//   (exact fp32).  A lane reads its MFMA operands as one ds_read_b128 per 16x16 sub-tile (4
//   consecutive k, fed to 4 successive MFMAs; A and B use the same k permutation); LDS rows
//   are padded to 20 floats so the 16 rows of a lane group fall on distinct 16-byte slots.
//   Global loads of tile t+1 are issued before the MFMAs of tile t.
#include "st_common.h"
#include <cstdlib>

namespace {

// LDS row stride 24 floats: the 16 lanes of each ds_read_b128 service group (rows 0-3 & 12-15 at one k offset, rows 4-11 at the next: see the
// microarch guide) then hit 16 distinct 16-byte slots of the 64 banks (slot = (6 row + k piece) mod 16); with 20 floats three pairs
// of a group collided (PMC bank-conflict ratio 0.33-0.55 in round 1)
constexpr int GM_BM = 64, GM_BN = 64, GM_BK = 16, GM_LD = 24, GM_THREADS = 256;

struct GmArgs {
    const float* A; int lda; const float* W; float* C; int ldc; int coff;
    int Bn, Tin, Tout, Cin, N, KT, pad, pool_prev, stride;
    int M;      // Bn * Tout
    int cpb;    // 16-float k-blocks per tap
    int kb_per_split;   // > 0: blockIdx.z takes k-blocks [z * kb_per_split, (z + 1) * kb_per_split) and writes its raw partial
    float* split_ws;    //      product to split_ws[z][m][n] (row stride N); gm_splitk_finish_kernel adds the slabs + epilogue
                        //      (gd_kernel: the same in 32-float chunks)
    st_gemm_epilogue ep;
};

// epilogue of one output element (shared by the tile epilogue and the split-K finish), in two halves: the element's optional
// operands (residual / highway H / dropout mask) are FETCHED first -- for all elements of a thread, before any is consumed -- and
// the arithmetic follows.  (As one function per element every operand was a load with a full s_waitcnt behind it: 16 elements x up
// to three serial round trips per thread at the end of a tile.)
__device__ __forceinline__ void gm_epilogue_fetch(const st_gemm_epilogue& ep, int m, int n, float& xr, float& xh, float& xm) {
    xr = ep.res ? ep.res[(size_t)m * ep.ldres + n] : 0.0f;
    xh = ep.highway_h ? ep.highway_h[(size_t)m * ep.ldhw + n] : 0.0f;
    xm = ep.mask ? ep.mask[(size_t)m * ep.ldmask + n] : 1.0f;
}
__device__ __forceinline__ float gm_epilogue_vals(const st_gemm_epilogue& ep, float v, float bias, float bn_m, float bn_s, float bn_w, float bn_b,
                                                  float xr, float xh, float xm) {
    v += bias;
    v = st_act(v, ep.act_pre);
    if (ep.bn_mean) v = (v - bn_m) * bn_s * bn_w + bn_b;
    v = st_act(v, ep.act_post);
    if (ep.highway_h) v = st_highway(xh, v, xr);        // Highway: y = H * T + x * (1 - T), v = T                (module.py:551-554)
    else if (ep.res) v += xr;
    if (ep.mask) v *= xm;
    return v;
}
__device__ __forceinline__ float gm_epilogue_one(const st_gemm_epilogue& ep, float v, int m, int n, float bias, float bn_m, float bn_s, float bn_w,
                                                 float bn_b) {
    float xr, xh, xm;
    gm_epilogue_fetch(ep, m, n, xr, xh, xm);
    return gm_epilogue_vals(ep, v, bias, bn_m, bn_s, bn_w, bn_b, xr, xh, xm);
}

// epilogue of the GEMM kernels: D[row = 4*(lane>>4) + r][col = lane&15] of the wave's MT x NT sub-tiles whose first element is (mb, nb)
template <int MT, int NT>
__device__ __forceinline__ void gm_epilogue_tiles(const GmArgs& g, const f32x4 (&acc)[MT][NT], int mb, int nb, int lane) {
    const st_gemm_epilogue& ep = g.ep;
    float* ws = g.kb_per_split > 0 ? g.split_ws + (size_t)blockIdx.z * g.M * g.N : nullptr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nb + nt * 16 + (lane & 15);
        if (n >= g.N) continue;
        const float bias = ep.bias ? ep.bias[n] : 0.0f;
        float bn_m = 0.f, bn_s = 1.f, bn_w = 1.f, bn_b = 0.f;
        if (ep.bn_mean) {
            bn_m = ep.bn_mean[n];
            bn_s = 1.0f / sqrtf(ep.bn_var[n] + ep.bn_eps);
            bn_w = ep.bn_w ? ep.bn_w[n] : 1.0f;
            bn_b = ep.bn_b ? ep.bn_b[n] : 0.0f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + mt * 16 + 4 * (lane >> 4) + r;
                if (m >= g.M) continue;
                if (ws) { ws[(size_t)m * g.N + n] = acc[mt][nt][r]; continue; }        // split-K: the raw partial product
                g.C[(size_t)m * g.ldc + g.coff + n] = gm_epilogue_one(ep, acc[mt][nt][r], m, n, bias, bn_m, bn_s, bn_w, bn_b);
            }
        }
    }
}

template <int MT = 2>
__device__ __forceinline__ void gm_epilogue(const GmArgs& g, const f32x4 (&acc)[MT][2], int m0, int n0, int wm, int wn, int lane) {
    gm_epilogue_tiles<MT, 2>(g, acc, m0 + wm * (16 * MT), n0 + wn * 32, lane);
}

// split-K finish: C = epilogue(sum_z slab_z), slabs added in a fixed order (deterministic); one thread per output element (VEC4: per
// four), consecutive threads = consecutive columns.  SS = compile-time slab count (0: run-time): all slab loads of a thread are
// requested before the first add (with a run-time loop the adds serialise the round trips: 8.0 us per call in the r04 forward trace
// for 4 slabs of 2.8 MB)
template <bool VEC4, int SS>
__global__ __launch_bounds__(256) void gm_splitk_finish_kernel(const GmArgs g, int Srt) {
    constexpr int W = VEC4 ? 4 : 1;                       // columns per thread (VEC4: N % 4 == 0, 16-byte slab loads)
    const int S = SS > 0 ? SS : Srt;
    const size_t total = (size_t)g.M * g.N, items = total / W;
    const st_gemm_epilogue& ep = g.ep;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        const size_t i = it * W;
        const int m = (int)(i / g.N), n0 = (int)(i - (size_t)m * g.N);
        float v[W];
        if (VEC4 && SS > 0) {
            f32x4 sl[SS > 0 ? SS : 1];
#pragma unroll
            for (int z = 0; z < SS; ++z) sl[z] = st_ld4(g.split_ws + (size_t)z * total + i);
            f32x4 s0 = sl[0];
#pragma unroll
            for (int z = 1; z < SS; ++z) s0 = s0 + sl[z];          // slab order 0, 1, 2, ...
#pragma unroll
            for (int c = 0; c < W; ++c) v[c] = s0[c];
        } else if (VEC4) {
            f32x4 s0 = st_ld4(g.split_ws + i);
            for (int z = 1; z < S; ++z) s0 = s0 + st_ld4(g.split_ws + (size_t)z * total + i);
#pragma unroll
            for (int c = 0; c < W; ++c) v[c] = s0[c];
        } else {
            float s0 = g.split_ws[i];
            for (int z = 1; z < S; ++z) s0 += g.split_ws[(size_t)z * total + i];
            v[0] = s0;
        }
        // every operand of the W elements is requested before any is consumed (one exposed latency instead of ~4 per column)
        float pb[W], pm[W], pv[W], pw[W], pbb[W], xr[W], xh[W], xm[W];
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const int n = n0 + c;
            pb[c] = ep.bias ? ep.bias[n] : 0.0f;
            pm[c] = ep.bn_mean ? ep.bn_mean[n] : 0.0f;
            pv[c] = ep.bn_mean ? ep.bn_var[n] : 1.0f;
            pw[c] = (ep.bn_mean && ep.bn_w) ? ep.bn_w[n] : 1.0f;
            pbb[c] = (ep.bn_mean && ep.bn_b) ? ep.bn_b[n] : 0.0f;
            gm_epilogue_fetch(ep, m, n, xr[c], xh[c], xm[c]);
        }
#pragma unroll
        for (int c = 0; c < W; ++c)
            v[c] = gm_epilogue_vals(ep, v[c], pb[c], pm[c], ep.bn_mean ? 1.0f / sqrtf(pv[c] + ep.bn_eps) : 1.0f, pw[c], pbb[c], xr[c], xh[c], xm[c]);
        float* cp = g.C + (size_t)m * g.ldc + g.coff + n0;
        if (VEC4 && st_aligned16(cp)) *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[VEC4 ? 1 : 0], v[VEC4 ? 2 : 0], v[VEC4 ? 3 : 0]};
        else {
#pragma unroll
            for (int c = 0; c < W; ++c) cp[c] = v[c];
        }
    }
}

template <bool VECA, bool VECW>
__global__ __launch_bounds__(GM_THREADS) void gm_kernel(const GmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[GM_BM * GM_LD];
    __shared__ __attribute__((aligned(16))) float Bs[GM_BN * GM_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * GM_BM, n0 = blockIdx.y * GM_BN;

    // staging role of this thread: one 4-float piece of row `srow` of each tile
    const int srow = tid >> 2, skq = tid & 3;
    const int am = m0 + srow;
    const bool a_row_ok = am < g.M;
    int ab = 0, ato = 0;
    if (a_row_ok) { ab = am / g.Tout; ato = am - ab * g.Tout; }
    const int wn_row = n0 + srow;
    const bool w_row_ok = wn_row < g.N;

    auto load_a = [&](int kb) -> f32x4 {
        const int tap = kb / g.cpb;
        const int ci = (kb - tap * g.cpb) * GM_BK + skq * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int ti = ato * g.stride + tap - g.pad;
        if (!a_row_ok || ti < 0 || ti >= g.Tin || ci >= g.Cin) return v;
        const float* p = g.A + ((size_t)ab * g.Tin + ti) * g.lda + ci;
        if (VECA) v = st_ld4(p); else v = st_ld4_guard(p, g.Cin - ci);
        if (g.pool_prev && ti > 0) {  // MaxPool1d(2, stride 1, padding 1)[:T] fused into the load
            f32x4 q;
            if (VECA) q = st_ld4(p - g.lda); else q = st_ld4_guard(p - g.lda, g.Cin - ci);
            v[0] = fmaxf(v[0], q[0]); v[1] = fmaxf(v[1], q[1]); v[2] = fmaxf(v[2], q[2]); v[3] = fmaxf(v[3], q[3]);
        }
        return v;
    };
    auto load_w = [&](int kb) -> f32x4 {
        const int tap = kb / g.cpb;
        const int ci = (kb - tap * g.cpb) * GM_BK + skq * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!w_row_ok || ci >= g.Cin) return v;
        if (VECW) {  // KT == 1 (torch Linear weight) or tap-major conv weight (N, KT, Cin): contiguous in ci
            v = st_ld4(g.W + ((size_t)wn_row * g.KT + tap) * g.Cin + ci);
        } else {     // torch Conv1d weight (N, Cin, KT): stride KT between consecutive ci
            const float* p = g.W + ((size_t)wn_row * g.Cin + ci) * g.KT + tap;
            const int rem = g.Cin - ci;
            v[0] = p[0];
            if (rem > 1) v[1] = p[g.KT];
            if (rem > 2) v[2] = p[2 * g.KT];
            if (rem > 3) v[3] = p[3 * g.KT];
        }
        return v;
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkb = g.KT * g.cpb;
    f32x4 ra = load_a(0), rw = load_w(0);
    const int fr = lane & 15, fk = (lane >> 4) * 4;
    for (int kb = 0; kb < nkb; ++kb) {
        *reinterpret_cast<f32x4*>(As + srow * GM_LD + skq * 4) = ra;
        *reinterpret_cast<f32x4*>(Bs + srow * GM_LD + skq * 4) = rw;
        __syncthreads();
        if (kb + 1 < nkb) { ra = load_a(kb + 1); rw = load_w(kb + 1); }
        f32x4 a4[2], b4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a4[t] = *reinterpret_cast<const f32x4*>(As + (wm * 32 + t * 16 + fr) * GM_LD + fk);
            b4[t] = *reinterpret_cast<const f32x4*>(Bs + (wn * 32 + t * 16 + fr) * GM_LD + fk);
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][cc], b4[nt][cc], acc[mt][nt], 0, 0, 0);
        __syncthreads();
    }

    gm_epilogue(g, acc, m0, n0, wm, wn, lane);
}

// Pipelined form for 16-byte addressable A (the usual case): two k-blocks of operands in flight in registers, requested with
// clamped (always valid) addresses and a validity flag applied when they are written to LDS -- no branch around a load, no
// integer division per block -- and barriers that only order the LDS traffic (__syncthreads would drain the prefetch).
// The one-block form below spent ~2100 cycles per k-block against 512 cycles of MFMAs.
// MT = 16-row MFMA tiles per wave along M: the workgroup tile is (32 MT) x 64.  MT = 1 halves the tile for grids that would not
// give every compute unit two workgroups with 64-row tiles: one wave per SIMD cannot overlap its own address arithmetic, LDS
// traffic and barrier with its MFMAs, a second workgroup on the SIMD can (measured: 258 workgroups of 64 x 64 ran at 0.29 of the
// fp32 matrix peak however the barriers were arranged).
// VECA = false: rows of A (and of W) that are not 16-byte addressable (Cin % 4 != 0: the input gradient of Linear(160, 1025) contracts
// over 1025 columns) -- the same pipeline with four scalar loads per piece, every column clamped and masked on its own (the
// one-block-per-barrier gm_kernel took 91 us for that product).
template <bool VECW, bool POOL, int MT, bool VECA = true>
__device__ __forceinline__ void gm_pipe_body(const GmArgs& g, float* __restrict__ As_, float* __restrict__ Bs_) {
    static_assert(VECA || !POOL, "the element-wise form has no fused max-pool");
    constexpr int BM = 32 * MT;
    // two LDS buffers of TWO 16-float k-blocks each: pair kp+1 is written while pair kp is multiplied -- one barrier per 32
    // MFMAs per wave.  (r02: with one k-block per barrier an iteration took ~1350 cycles for 512 cycles of MFMAs -- one wave per
    // SIMD cannot hide the LDS-write -> barrier -> LDS-read chain; the pair halves the barriers per MFMA.  A k-block stays the
    // addressing unit, so a pair may straddle two conv taps.)
    float (*As)[2][BM * GM_LD] = reinterpret_cast<float (*)[2][BM * GM_LD]>(As_);
    float (*Bs)[2][GM_BN * GM_LD] = reinterpret_cast<float (*)[2][GM_BN * GM_LD]>(Bs_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * GM_BN;
    const int srow = tid >> 2, skq = tid & 3;
    const int am = m0 + min(srow, BM - 1);          // (MT = 1: threads past the 32 A rows repeat the last row's loads, never stored)
    const bool a_stage = srow < BM;
    const bool a_row_ok = am < g.M;
    int ab = 0, ato = 0;
    if (a_row_ok) { ab = am / g.Tout; ato = am - ab * g.Tout; }
    const int wn_row = n0 + srow;
    const bool w_row_ok = wn_row < g.N;
    const float* __restrict__ abase = g.A + (size_t)ab * g.Tin * g.lda;
    const float* __restrict__ wbase = g.W + (size_t)min(wn_row, g.N - 1) * g.Cin * g.KT;
    const int t_base = ato * g.stride - g.pad;

    struct Blk { f32x4 a, q, w; bool va, vw; unsigned mk; };     // mk: per-column validity of the piece (element-wise form)
    // split-K: this workgroup's k-blocks are [kb_lo, kb_hi) of the KT * cpb blocks (the whole range without a split)
    const int nkb = g.KT * g.cpb;
    const int kb_lo = g.kb_per_split > 0 ? (int)blockIdx.z * g.kb_per_split : 0;
    const int kb_hi = g.kb_per_split > 0 ? min(nkb, kb_lo + g.kb_per_split) : nkb;
    int tap_n = kb_lo / g.cpb, cb_n = kb_lo - tap_n * g.cpb;      // (tap, 16-float block within the tap) of the next block to request
    int left_n = kb_hi - kb_lo;                                   // blocks of the range not requested yet
    // per-tap state of the 16-byte form (see issue())
    const float* pa_tap = abase; const float* pw_tap = wbase;
    bool oka_tap = false, okw_tap = false, pool_tap = false;
    auto set_tap = [&]() __attribute__((always_inline)) {
        const int tap = min(tap_n, g.KT - 1);
        const int ti = t_base + tap;
        const int tic = min(max(ti, 0), g.Tin - 1);
        pa_tap = abase + (size_t)tic * g.lda;
        pw_tap = VECW ? wbase + (size_t)tap * g.Cin : wbase + tap;
        oka_tap = tap_n < g.KT && a_row_ok && ti >= 0 && ti < g.Tin;
        okw_tap = tap_n < g.KT && w_row_ok;
        pool_tap = tic > 0;
    };
    if (VECA) set_tap();
    auto issue = [&](Blk& r) __attribute__((always_inline)) {
        const int ci = cb_n * GM_BK + skq * 4;
        const bool in_k = left_n > 0 && tap_n < g.KT && ci < g.Cin;
        --left_n;
        const int cic = min(ci, g.Cin - 4);
        if (!VECA) {
            const int tap = min(tap_n, g.KT - 1);
            const int ti = t_base + tap;
            const int tic = min(max(ti, 0), g.Tin - 1);
            const int c0 = min(ci, g.Cin - 1), c1 = min(ci + 1, g.Cin - 1), c2 = min(ci + 2, g.Cin - 1), c3 = min(ci + 3, g.Cin - 1);
            const float* pa = abase + (size_t)tic * g.lda;
            const bool piece = ci + 4 <= g.Cin;           // a whole piece: one 16-byte load from the (4-byte aligned) row; only the tail goes by element
            if (piece) r.a = st_ld4_u(pa + ci); else r.a = f32x4{pa[c0], pa[c1], pa[c2], pa[c3]};
            r.mk = (ci < g.Cin ? 1u : 0u) | (ci + 1 < g.Cin ? 2u : 0u) | (ci + 2 < g.Cin ? 4u : 0u) | (ci + 3 < g.Cin ? 8u : 0u);
            r.va = in_k && a_row_ok && ti >= 0 && ti < g.Tin;
            if (VECW) {      // Linear weight (KT == 1) or tap-major conv weight: consecutive ci are adjacent
                const float* pw = wbase + (size_t)tap * g.Cin;
                if (piece) r.w = st_ld4_u(pw + ci); else r.w = f32x4{pw[c0], pw[c1], pw[c2], pw[c3]};
            } else {         // torch Conv1d weight (N, Cin, KT)
                const float* pw = wbase + tap;
                r.w = f32x4{pw[(size_t)c0 * g.KT], pw[(size_t)c1 * g.KT], pw[(size_t)c2 * g.KT], pw[(size_t)c3 * g.KT]};
            }
            r.vw = in_k && w_row_ok;
            if (++cb_n == g.cpb) { cb_n = 0; ++tap_n; }
            return;
        }
        // (16-byte form: everything that only changes with the TAP -- the clamped row pointer of A, the weight pointer, the row
        //  validity -- is kept from block to block and refreshed by set_tap() when the tap advances: per block there is one clamp of the
        //  column, two pointer adds and the loads.  The per-block 64-bit address arithmetic cost as much issue time as the block's MFMAs.)
        r.a = st_ld4(pa_tap + cic);
        if (POOL) r.q = st_ld4((pool_tap ? pa_tap - g.lda : pa_tap) + cic);       // MaxPool1d(2, stride 1, padding 1)[:T] fused into the load
        r.va = in_k && oka_tap;
        if (VECW) {  // Linear weight (KT == 1) or tap-major conv weight (N, KT, Cin): a k-block is contiguous
            r.w = st_ld4(pw_tap + cic);
        } else {     // torch Conv1d weight (N, Cin, KT): stride KT between consecutive ci
            const float* pw = pw_tap + (size_t)cic * g.KT;
            r.w = f32x4{pw[0], pw[g.KT], pw[2 * g.KT], pw[3 * g.KT]};
        }
        r.vw = in_k && okw_tap;
        if (++cb_n == g.cpb) { cb_n = 0; ++tap_n; set_tap(); }
    };
    auto commit = [&](const Blk& r, const int buf, const int half) __attribute__((always_inline)) {
        f32x4 va = r.a;
        if (POOL) { va[0] = fmaxf(va[0], r.q[0]); va[1] = fmaxf(va[1], r.q[1]); va[2] = fmaxf(va[2], r.q[2]); va[3] = fmaxf(va[3], r.q[3]); }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 vw = r.w;
        if (!VECA) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (!(r.mk & (1u << j))) { va[j] = 0.0f; vw[j] = 0.0f; }
        }
        if (a_stage) *reinterpret_cast<f32x4*>(As[buf][half] + srow * GM_LD + skq * 4) = r.va ? va : z;
        *reinterpret_cast<f32x4*>(Bs[buf][half] + srow * GM_LD + skq * 4) = r.vw ? vw : z;
    };
    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = (lane >> 4) * 4;
    auto compute = [&](const int buf) __attribute__((always_inline)) {
        f32x4 a4[2][MT], b4[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int t = 0; t < MT; ++t) a4[h][t] = *reinterpret_cast<const f32x4*>(As[buf][h] + (wm * (16 * MT) + t * 16 + fr) * GM_LD + fk);
#pragma unroll
            for (int t = 0; t < 2; ++t) b4[h][t] = *reinterpret_cast<const f32x4*>(Bs[buf][h] + (wn * 32 + t * 16 + fr) * GM_LD + fk);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[h][mt][cc], b4[h][nt][cc], acc[mt][nt], 0, 0, 0);
    };

    const int npairs = (kb_hi - kb_lo + 1) >> 1;
    Blk r0[2], r1[2];
    issue(r0[0]); issue(r0[1]);
    issue(r1[0]); issue(r1[1]);
    commit(r0[0], 0, 0); commit(r0[1], 0, 1);
    issue(r0[0]); issue(r0[1]);     // pair 2 (past the end: clamped addresses, flags false)
    st_lds_barrier();
    for (int kp = 0; kp < npairs; kp += 2) {
        // buffer 0 holds pair kp; r1 = pair kp+1, r0 = pair kp+2
        commit(r1[0], 1, 0); commit(r1[1], 1, 1);   // (nobody reads buffer 1 now: its last readers passed the barrier below / above)
        issue(r1[0]); issue(r1[1]);                 // pair kp + 3
        compute(0);
        st_lds_barrier();                           // buffer 1 complete, buffer 0 free
        if (kp + 1 >= npairs) break;
        commit(r0[0], 0, 0); commit(r0[1], 0, 1);
        issue(r0[0]); issue(r0[1]);                 // pair kp + 4
        compute(1);
        st_lds_barrier();
    }
    gm_epilogue<MT>(g, acc, m0, n0, wm, wn, lane);
}

template <bool VECW, bool POOL, int MT = 2, bool VECA = true>
__global__ __launch_bounds__(GM_THREADS) void gm_pipe_kernel(const GmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[4 * (32 * MT) * GM_LD];
    __shared__ __attribute__((aligned(16))) float Bs[4 * GM_BN * GM_LD];
    gm_pipe_body<VECW, POOL, MT, VECA>(g, As, Bs);
}


// ---- gd_kernel: both operands by LDS-DMA (round 4) ------------------------------------------------------------------------
// global_load_lds_dwordx4 moves 8 rows x 128 bytes (one whole line per row) per wave instruction straight into LDS: no staging
// registers, no ds_write pass, no selects -- the register-staged pipeline above spent 30-40 % of its time on exactly those (lab
// ablations, tools/gemm_lab2.hip: 363 us with, 257 us without its loads on a 33 GFLOP conv; this form 314 us), and its 64-byte row
// pieces were half lines.
//   * LDS image per operand and 32-float chunk: [row][8 slots of 16 bytes], lane-linear as the DMA writes it; slot s of row L holds
//     the row's piece s ^ (L & 7) (the swizzle is applied to the SOURCE address), a fragment read of piece p goes to slot
//     p ^ (L & 7): the 16 lanes of every ds_read_b128 service group hit 16 distinct 16-byte slots;
//   * the reduction axis is the flat (tap, channel) index of tap-major weights: with channels-last activations of row stride Cin the
//     taps of one output row are ONE contiguous run of KT * Cin floats, so a piece's address is run start + k and its validity two
//     compares (frames outside the utterance, k past K, rows past M / N read 16 zero bytes from a constant instead);
//   * three buffers, ONE barrier per chunk: the requests of chunk c + 2 go out right after the barrier of iteration c (every wave
//     has left compute(c - 1), whose buffer they overwrite); the counted vmcnt in front of the barrier retires this wave's pieces
//     of chunk c and leaves chunk c + 1 in flight.  ALL LDS of the kernel is one array (a second __shared__ object makes the
//     compiler drain the DMA queue before every ds_read);
//   * the columns of a wave's MFMA tiles are INTERLEAVED (tile nt, column r <-> output column NT r + nt; the W rows are permuted
//     through the source address), so a lane holds NT consecutive output columns per row: 8-byte stores;
//   * per output element the k order is that of gm_pipe_kernel / highway_stack_kernel (16-float blocks ascending, the lane's four
//     k's in order): results are bit-identical to theirs.
// Shapes it takes: 16-byte addressable rows of A and W, weights contiguous in the flat k (Linear, or tap-major conv weights), row
// stride Cin for KT > 1, no fused max-pool.
__device__ const f32x4 gd_zero4 = {0.f, 0.f, 0.f, 0.f};
typedef const __attribute__((address_space(1))) void* gd_gptr_t;
typedef __attribute__((address_space(3))) void* gd_lptr_t;

template <int BM, int BN>
__device__ __forceinline__ void gd_body(const GmArgs& g, float* __restrict__ lds) {
    constexpr int NWM = 2, NWN = 2, NTH = 256;
    constexpr int MT = BM / NWM / 16, NT = BN / NWN / 16;
    constexpr int RPP = NTH / 8;                  // rows one pass of the workgroup covers (8 lanes per row)
    constexpr int LA = BM / RPP, LB = BN / RPP;   // DMA instructions per thread and chunk
    constexpr int BUF = (BM + BN) * 32;           // floats per buffer
    static_assert(BM % RPP == 0 && BN % RPP == 0 && MT >= 1 && NT >= 1, "geometry");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int lrow = lane >> 3, slot = lane & 7;
    const float* zp = reinterpret_cast<const float*>(&gd_zero4);
    const int K = g.KT * g.Cin;
    // per DMA: the row's run start (flat k = 0), its valid k range, the piece this lane fetches
    const float* abase[LA]; int aklo[LA], akhi[LA], apk[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int L = i * RPP + wave * 8 + lrow;              // LDS row = tile row
        const int m = m0 + L;
        const int mc = min(m, g.M - 1);
        const int b = mc / g.Tout, t = (mc - b * g.Tout) * g.stride - g.pad;      // first frame of the row's run (may lie in front of the utterance)
        abase[i] = g.A + ((ptrdiff_t)b * g.Tin + t) * g.lda;
        aklo[i] = m < g.M ? max(0, -t) * g.Cin : 0x7fffffff;
        akhi[i] = min(g.KT, g.Tin - t) * g.Cin;
        apk[i] = (slot ^ (L & 7)) * 4;
    }
    const float* wbase[LB]; int wkhi[LB], wpk[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int L = i * RPP + wave * 8 + lrow;              // LDS row; the W-tile row it holds: interleaved columns
        const int wb = L / (16 * NT), within = L % (16 * NT);
        const int r = wb * 16 * NT + NT * (within % 16) + within / 16;
        wbase[i] = g.W + (size_t)min(n0 + r, g.N - 1) * K;
        wkhi[i] = n0 + r < g.N ? K : 0;
        wpk[i] = (slot ^ (L & 7)) * 4;
    }
    const int nch = (K + 31) / 32;
    const int c_lo = g.kb_per_split > 0 ? (int)blockIdx.z * g.kb_per_split : 0;
    const int c_hi = g.kb_per_split > 0 ? min(nch, c_lo + g.kb_per_split) : nch;
    const int n_my = c_hi - c_lo;
    int kreq = c_lo * 32;                          // flat k of the next chunk to request
    auto issue = [&](int bufi) __attribute__((always_inline)) {
        float* ab = lds + bufi * BUF;
        float* bb = ab + BM * 32;
        const bool live = kreq < c_hi * 32;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int k = kreq + apk[i];
            const float* src = (live && k >= aklo[i] && k < akhi[i]) ? abase[i] + k : zp;
            __builtin_amdgcn_global_load_lds((gd_gptr_t)src, (gd_lptr_t)(ab + (i * RPP + wave * 8) * 32), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int k = kreq + wpk[i];
            const float* src = (live && k < wkhi[i]) ? wbase[i] + k : zp;
            __builtin_amdgcn_global_load_lds((gd_gptr_t)src, (gd_lptr_t)(bb + (i * RPP + wave * 8) * 32), 16, 0, 0);
        }
        kreq += 32;
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int bufi) __attribute__((always_inline)) {
        const float* ab = lds + bufi * BUF;
        const float* bb = ab + BM * 32;
        f32x4 a4[2][MT], b4[2][NT];      // the fragments of both 16-float halves first: one exposed LDS latency per chunk
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int t = 0; t < MT; ++t) { const int L = wm * 16 * MT + t * 16 + fr; a4[h][t] = *reinterpret_cast<const f32x4*>(ab + L * 32 + (((h * 4 + fq) ^ (L & 7)) * 4)); }
#pragma unroll
            for (int t = 0; t < NT; ++t) { const int L = wn * 16 * NT + t * 16 + fr; b4[h][t] = *reinterpret_cast<const f32x4*>(bb + L * 32 + (((h * 4 + fq) ^ (L & 7)) * 4)); }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[h][mt][cc], b4[h][nt][cc], acc[mt][nt], 0, 0, 0);
    };
    // the per-column epilogue parameters of this lane's NT columns are requested BEFORE the product (they are older than every DMA
    // request, so the counted waits below still hold) and consumed after it: in the epilogue each was a load with a full wait
    // behind it (bias, mean, var -> rsqrt, weight, bias: ~6 serial round trips per workgroup at the end of a one-tile launch)
    const st_gemm_epilogue& ep = g.ep;
    const int nb = n0 + wn * 16 * NT + NT * fr;
    float bias[NT], bn_m[NT], bn_v[NT], bn_w[NT], bn_b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = min(nb + nt, g.N - 1);
        bias[nt] = ep.bias ? ep.bias[n] : 0.0f;
        bn_m[nt] = ep.bn_mean ? ep.bn_mean[n] : 0.0f;
        bn_v[nt] = ep.bn_mean ? ep.bn_var[n] : 1.0f;
        bn_w[nt] = (ep.bn_mean && ep.bn_w) ? ep.bn_w[n] : 1.0f;
        bn_b[nt] = (ep.bn_mean && ep.bn_b) ? ep.bn_b[n] : 0.0f;
    }
    issue(0);
    issue(1);
    int bi = 0, bn = 2;
    for (int c = 0; c < n_my; ++c) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LA + LB) : "memory");      // chunk c has landed (this wave's pieces); chunk c + 1 may be in flight
        st_lds_barrier();                                                     // ... and every other wave's
        issue(bn);
        compute(bi);
        bi = bi == 2 ? 0 : bi + 1;
        bn = bn == 2 ? 0 : bn + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // (the two requests past the end: zeros into buffers nobody reads)
    // epilogue: lane (fr, fq) holds rows 4 fq + e of each row tile and the NT consecutive columns nb .. nb + NT - 1
    if (g.kb_per_split > 0) {                      // split-K: the raw partial product
        float* ws = g.split_ws + (size_t)blockIdx.z * g.M * g.N;
        const bool vec = NT == 2 && g.N % 2 == 0 && nb + 1 < g.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * 16 * MT + mt * 16 + 4 * fq + e;
                if (m >= g.M) continue;
                float* p = ws + (size_t)m * g.N + nb;
                if (vec) { typedef float f32x2 __attribute__((ext_vector_type(2))); *reinterpret_cast<f32x2*>(p) = f32x2{acc[mt][0][e], acc[mt][NT - 1][e]}; }
                else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) if (nb + nt < g.N) p[nt] = acc[mt][nt][e];
                }
            }
        return;
    }
    float bn_s[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bn_s[nt] = ep.bn_mean ? 1.0f / sqrtf(bn_v[nt] + ep.bn_eps) : 1.0f;
    const bool vec = NT == 2 && nb + 1 < g.N && ((g.ldc | g.coff) & 1) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 7) == 0;
    // the optional per-element operands of ALL the lane's elements first (clamped addresses: rows / columns past the edge re-read a valid one)
    float xr[MT][4][NT], xh[MT][4][NT], xm[MT][4][NT];
    if (ep.res || ep.highway_h || ep.mask) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    gm_epilogue_fetch(ep, min(m0 + wm * 16 * MT + mt * 16 + 4 * fq + e, g.M - 1), min(nb + nt, g.N - 1), xr[mt][e][nt], xh[mt][e][nt], xm[mt][e][nt]);
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { xr[mt][e][nt] = 0.0f; xh[mt][e][nt] = 0.0f; xm[mt][e][nt] = 1.0f; }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm * 16 * MT + mt * 16 + 4 * fq + e;
            if (m >= g.M) continue;
            float v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                v[nt] = nb + nt < g.N ? gm_epilogue_vals(ep, acc[mt][nt][e], bias[nt], bn_m[nt], bn_s[nt], bn_w[nt], bn_b[nt], xr[mt][e][nt], xh[mt][e][nt],
                                                         xm[mt][e][nt]) : 0.0f;
            float* p = g.C + (size_t)m * g.ldc + g.coff + nb;
            if (vec) { typedef float f32x2 __attribute__((ext_vector_type(2))); *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[NT - 1]}; }
            else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) if (nb + nt < g.N) p[nt] = v[nt];
            }
        }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gd_kernel(const GmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[3 * (BM + BN) * 32];
    gd_body<BM, BN>(g, lds);
}

// several jobs in one launch (blockIdx.z = job, longest reductions first): the conv bank's K convs over the same input.  A launch
// of its own gives each of them ~one workgroup per compute unit -- one wave per SIMD, nothing to overlap its LDS traffic and barriers
// with; together the K convs keep several workgroups resident per compute unit.
constexpr int GM_MAX_BATCH = 8;      // (kernel arguments: 8 jobs stay well under the 4 KB limit)
struct GmBatch { GmArgs g[GM_MAX_BATCH]; };

template <bool VECW, int MT>
__global__ __launch_bounds__(GM_THREADS) void gm_pipe_batch_kernel(const GmBatch b) {
    __shared__ __attribute__((aligned(16))) float As[4 * (32 * MT) * GM_LD];
    __shared__ __attribute__((aligned(16))) float Bs[4 * GM_BN * GM_LD];
    const GmArgs& g = b.g[blockIdx.z];
    if ((int)blockIdx.x * (32 * MT) >= g.M || (int)blockIdx.y * GM_BN >= g.N) return;      // (uniform: the whole workgroup leaves)
    gm_pipe_body<VECW, false, MT>(g, As, Bs);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gd_batch_kernel(const GmBatch b) {
    __shared__ __attribute__((aligned(16))) float lds[3 * (BM + BN) * 32];
    const GmArgs& g = b.g[blockIdx.z];
    if ((int)blockIdx.x * BM >= g.M || (int)blockIdx.y * BN >= g.N) return;      // (uniform: the whole workgroup leaves)
    gd_body<BM, BN>(g, lds);
}

// ---- the highway stack as ONE kernel (inference) ------------------------------------------------------------------------
// y = H(x) * T(x) + x * (1 - T(x)), H = relu(W_H x + b_H), T = sigmoid(W_T x + b_T), n_layers times      (src/module.py:541-555, :609-611)
// The layers are row-wise independent: a workgroup keeps its 32 rows in LDS through all the layers; the two weight matrices of a
// layer (2C x C) are staged in LDS while the previous layer is multiplied.  Per layer a wave takes 2-3 of the 10 (row tile, column
// tile) units and forms the H and the T tile of a unit side by side (same A fragment), so the combine happens in registers.  The k
// order per output element is that of gm_pipe_kernel (k-blocks of 16 ascending, the lane's four k's in order): bit-identical to the
// two-GEMMs-per-layer form it replaces (8 launches of ~10 us for 0.27 GFLOP each at the CBHG's 80 channels).
constexpr int HW_ROWS = 32, HW_MAXC = 80, HW_MAXL = 8;
struct HwArgs {
    const float* x; int ldx; float* y; int ldy; int M, C, NL;
    const float* wh[HW_MAXL]; const float* bh[HW_MAXL]; const float* wt[HW_MAXL]; const float* bt[HW_MAXL];
};

__global__ __launch_bounds__(256) void highway_stack_kernel(const HwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hw_lds[];
    const int C = a.C, LD = C + 4, C4 = C >> 2;          // LD: rows 16 apart in the b128 reads fall on distinct 16-byte slots (C % 16 == 0)
    float* xs[2] = {hw_lds, hw_lds + HW_ROWS * LD};
    // ONE weight buffer (round 4): with two, a workgroup took 129 KB of LDS -- one per compute unit, and the 258 workgroups of the C2
    // postnet (8256 rows / 32) needed a second round for their last two: 50 us in the forward trace.  75 KB lets two reside; the
    // next layer's weights wait in registers (requested before the layer's MFMAs) and are written after an extra barrier.
    float* wsb = hw_lds + 2 * HW_ROWS * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * HW_ROWS;
    const int nw4 = 2 * C * C4;                          // float4 pieces of one layer's [W_H ; W_T]
    constexpr int WPT = (2 * HW_MAXC * (HW_MAXC / 4) + 255) / 256;     // pieces per thread (13 at C = 80)
    f32x4 wreg[WPT];
    auto request_w = [&](int l) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int i = tid + j * 256;
            const int ic = i < nw4 ? i : 0;
            const int row = ic / C4, c4 = ic - row * C4;
            const float* src = row < C ? a.wh[l] + (size_t)row * C : a.wt[l] + (size_t)(row - C) * C;
            wreg[j] = st_ld4(src + 4 * c4);
        }
    };
    auto commit_w = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int i = tid + j * 256;
            if (i < nw4) { const int row = i / C4, c4 = i - row * C4; *reinterpret_cast<f32x4*>(dst + row * LD + 4 * c4) = wreg[j]; }
        }
    };
    request_w(0);
    for (int i = tid; i < HW_ROWS * C4; i += 256) {      // the rows of this workgroup (rows past M: zeros, never stored)
        const int r = i / C4, c4 = i - r * C4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + r < a.M) v = st_ld4(a.x + (size_t)(m0 + r) * a.ldx + 4 * c4);
        *reinterpret_cast<f32x4*>(xs[0] + r * LD + 4 * c4) = v;
    }
    commit_w(wsb);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    const int ntile = C >> 4, nunits = 2 * ntile;
    for (int l = 0; l < a.NL; ++l) {
        const float* xc = xs[l & 1];
        float* xn = xs[(l + 1) & 1];
        const float* wc = wsb;
        if (l + 1 < a.NL) request_w(l + 1);              // in flight while this layer is multiplied
        for (int u = wave; u < nunits; u += 4) {
            const int rt = u / ntile, j = u - rt * ntile;
            f32x4 acch = {0.f, 0.f, 0.f, 0.f}, acct = {0.f, 0.f, 0.f, 0.f};
            const float* ap = xc + (rt * 16 + fr) * LD + 4 * fq;
            const float* hp = wc + (j * 16 + fr) * LD + 4 * fq;
            const float* tp = wc + (C + j * 16 + fr) * LD + 4 * fq;
            for (int k0 = 0; k0 < C; k0 += 16) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + k0);
                const f32x4 h4 = *reinterpret_cast<const f32x4*>(hp + k0);
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(tp + k0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acch = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c], h4[c], acch, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c], t4[c], acct, 0, 0, 0);
                }
            }
            const int n = j * 16 + fr;
            const float bhv = a.bh[l] ? a.bh[l][n] : 0.0f, btv = a.bt[l] ? a.bt[l][n] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = rt * 16 + 4 * fq + e;
                const float hh = st_act(acch[e] + bhv, ST_ACT_RELU);
                const float tt = st_act(acct[e] + btv, ST_ACT_SIGMOID);
                const float xx = xc[m * LD + n];
                xn[m * LD + n] = st_highway(hh, tt, xx);
            }
        }
        if (l + 1 < a.NL) {
            st_lds_barrier();                            // every wave has read this layer's weights
            commit_w(wsb);
        }
        __syncthreads();
    }
    const float* xf = xs[a.NL & 1];
    for (int i = tid; i < HW_ROWS * C4; i += 256) {
        const int r = i / C4, c4 = i - r * C4;
        if (m0 + r < a.M) *reinterpret_cast<f32x4*>(a.y + (size_t)(m0 + r) * a.ldy + 4 * c4) = *reinterpret_cast<const f32x4*>(xf + r * LD + 4 * c4);
    }
}

// ---- training-mode BatchNorm helpers ---------------------------------------------------
// column statistics, two launches: (1) grid = (column blocks of 64, row chunks): every block reduces its chunk of
// rows to (mean_c, M2_c) with a chunk-local two-pass (the second pass re-reads <= 64 KB from cache);
// (2) one thread per column merges the chunks in a fixed order (Chan et al.: exact pairwise update of mean / M2).
__global__ __launch_bounds__(256) void bn_stats_chunk_kernel(const float* X, int ldx, int coff, int M, int N, int rows_per_chunk,
                                                             float* part) {   // part[chunk][2][N]
    __shared__ float red[4][64];
    __shared__ float smean[64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const bool ok = n < N;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    const float* __restrict__ xp = X + coff + min(n, N - 1);
    // rows m0 + rl, + 4, ...: four independent running sums (rows 16 apart) keep four loads in flight per thread -- with one sum the
    // loop is a chain of dependent round trips (12.9 us per 4 MB layer in round 2's form)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = m0 + rl;
    for (; m + 12 < m1; m += 16) {
        s0 += xp[(size_t)m * ldx]; s1 += xp[(size_t)(m + 4) * ldx]; s2 += xp[(size_t)(m + 8) * ldx]; s3 += xp[(size_t)(m + 12) * ldx];
    }
    for (; m < m1; m += 4) s0 += xp[(size_t)m * ldx];
    red[rl][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0) smean[c] = (red[0][c] + red[1][c] + red[2][c] + red[3][c]) / (float)max(m1 - m0, 1);
    __syncthreads();
    const float mean = smean[c];
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    m = m0 + rl;
    for (; m + 12 < m1; m += 16) {      // (second pass: the chunk's rows come from cache)
        const float d0 = xp[(size_t)m * ldx] - mean, d1 = xp[(size_t)(m + 4) * ldx] - mean;
        const float d2 = xp[(size_t)(m + 8) * ldx] - mean, d3 = xp[(size_t)(m + 12) * ldx] - mean;
        q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2); q3 = fmaf(d3, d3, q3);
    }
    for (; m < m1; m += 4) { const float d = xp[(size_t)m * ldx] - mean; q0 = fmaf(d, d, q0); }
    __syncthreads();
    red[rl][c] = (q0 + q1) + (q2 + q3);
    __syncthreads();
    if (rl == 0 && ok) {
        part[((size_t)blockIdx.y * 2 + 0) * N + n] = mean;
        part[((size_t)blockIdx.y * 2 + 1) * N + n] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}

// merge of the chunk records: 16 columns x 16 chunk lanes per workgroup; lane j holds chunks j, j + 16, ... (at most 8: chunks <= 128)
// in REGISTERS -- all its loads are issued at once, one memory round trip for both sweeps -- and the 16 lanes' sums are combined
// through LDS in a fixed order.
//   mean = sum_c n_c mean_c / M;   M2 = sum_c [M2_c + n_c (mean_c - mean)^2]      (exact decomposition, Chan et al.)
// (one thread per column walking all the chunks took 16 us per layer: 2 x 64 dependent-latency iterations)
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* part, int chunks, int rows_per_chunk, int M, int N,
                                                             float* mean_out, float* var_out, float* run_mean, float* run_var,
                                                             float momentum, long long* batches_tracked, int as_record) {
    constexpr int CL = 16, PER = 8;
    __shared__ float red[CL][16];
    __shared__ float smean[16];
    const int c = threadIdx.x & 15, cl = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + c;
    if (blockIdx.x == 0 && threadIdx.x == 0 && batches_tracked) batches_tracked[0] += 1;      // nn.BatchNorm1d.num_batches_tracked, without a launch of its own
    const bool ok = n < N;
    const float* pm = part + min(n, N - 1);
    const size_t cs = (size_t)2 * N;
    float mu[PER], m2[PER], cnt[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int ch = cl + j * CL;
        const bool in = ch < chunks;
        const int chc = in ? ch : 0;
        // (absent chunks re-read chunk 0 and are zeroed by a FACTOR: `if (!in) m2 = 0` makes the compiler branch around the load and wait)
        mu[j] = pm[(size_t)chc * cs]; m2[j] = pm[(size_t)chc * cs + N] * (in ? 1.0f : 0.0f);
        cnt[j] = in ? (float)max(min(M, (ch + 1) * rows_per_chunk) - ch * rows_per_chunk, 0) : 0.0f;
    }
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) a = fmaf(cnt[j], mu[j], a);
    red[cl][c] = a;
    __syncthreads();
    if (cl == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CL; ++k) t += red[k][c];
        smean[c] = t / (float)M;
    }
    __syncthreads();
    const float mean = smean[c];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) { const float d = mu[j] - mean; q += m2[j] + cnt[j] * d * d; }
    __syncthreads();
    red[cl][c] = q;
    __syncthreads();
    if (cl != 0 || !ok) return;
    float m2t = 0.f;
#pragma unroll
    for (int k = 0; k < CL; ++k) m2t += red[k][c];
    const float var_b = m2t / (float)M;
    mean_out[n] = mean;
    var_out[n] = as_record ? m2t : var_b;          // record form (SyncBN): (mean, M2, row count) of this rank's rows, one buffer
    if (as_record && n == 0) var_out[N] = (float)M;
    if (run_mean) {
        const float var_u = M > 1 ? m2t / (float)(M - 1) : var_b;
        run_mean[n] = (1.0f - momentum) * run_mean[n] + momentum * mean;
        run_var[n] = (1.0f - momentum) * run_var[n] + momentum * var_u;
    }
}

__global__ __launch_bounds__(256) void bn_sync_merge_kernel(const float* rec, int world, int N, float* mean_out, float* var_out,
                                                            float* run_mean, float* run_var, float momentum, float* inv_total_out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rs = (size_t)2 * N + 1;
    float total = 0.0f;
    for (int r = 0; r < world; ++r) total += rec[r * rs + 2 * N];
    if (n == 0) inv_total_out[0] = 1.0f / total;
    if (n >= N) return;
    float a = 0.0f;
    for (int r = 0; r < world; ++r) a = fmaf(rec[r * rs + 2 * N], rec[r * rs + n], a);
    const float mean = a / total;
    float m2 = 0.0f;
    for (int r = 0; r < world; ++r) { const float d = rec[r * rs + n] - mean; m2 += rec[r * rs + N + n] + rec[r * rs + 2 * N] * d * d; }
    mean_out[n] = mean;
    var_out[n] = m2 / total;
    if (run_mean) {
        run_mean[n] = (1.0f - momentum) * run_mean[n] + momentum * mean;
        run_var[n] = (1.0f - momentum) * run_var[n] + momentum * (m2 / fmaxf(total - 1.0f, 1.0f));
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(float* X, int ldx, int coff, int M, int N,
                                                       const float* mean, const float* var, const float* w,
                                                       const float* b, float eps, int act) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
        float* p = X + (size_t)m * ldx + coff + n;
        float v = (*p - mean[n]) / sqrtf(var[n] + eps);
        v = v * (w ? w[n] : 1.0f) + (b ? b[n] : 0.0f);
        *p = st_act(v, act);
    }
}

}  // namespace

// Tile of the LDS-DMA kernel for an (M, N) product (measured on the shapes of one C2 / C5 forward, tools/gemm_lab2.hip):
//   0 = 64 x 64: grids of >= 1024 such tiles, and the split-K shapes;  1 = 32 x 64: smaller grids (twice the workgroups);
//   2 = 64 x 32: narrow outputs whose last 64-column block would be at most half full (N = 80: 96 padded columns instead of 128)
static int gd_tile(long M, int N, int S) {
    const int rem = N % 64;
    if (N <= 192 && rem > 0 && rem <= 32) return 2;
    const long wgs64 = ((M + 63) / 64) * ((N + 63) / 64);
    return (S > 1 || wgs64 >= 1024) ? 0 : 1;
}

// Split-K pays where the 64 x 64 grid is small AND the reduction is long: the encoder convs (1376 x 512 x 2560: 176 tiles walking 80
// chunks of 32 floats each; 4 slabs incl. the finish pass: 76 -> 44 us), the 640 -> 128 projection conv.  Short reductions (K < 1536)
// lose or gain nothing.  Slabs: enough for ~700 workgroups, at least 12 chunks each.
extern "C" int st_gemm_splitk_slabs(int Bn, int Tout, int Cin, int N, int KT) {
    if (Bn <= 0 || Tout <= 0 || Cin < 4 || Cin % 4 != 0 || N <= 0 || KT <= 0) return 1;
    const long M = (long)Bn * Tout;
    const long wgs = ((M + GM_BM - 1) / GM_BM) * ((N + GM_BN - 1) / GM_BN);
    const int nch = (KT * Cin + 31) / 32;
    if (wgs >= 512 || nch < 48 || gd_tile(M, N, 1) == 2) return 1;
    int S = (int)((704 + wgs / 2) / wgs);
    if (S > nch / 12) S = nch / 12;
    if (S > 8) S = 8;
    return S < 2 ? 1 : S;
}

// argument checks of one GEMM job and its kernel arguments; veca / vecw = 16-byte addressable activations / weights
static int gm_prepare(const float* A, int lda, const float* W, float* C, int ldc, int coff,
                      int Bn, int Tin, int Tout, int Cin, int N, int KT, int pad, int stride, int pool_prev,
                      const st_gemm_epilogue* ep, GmArgs& g, bool& veca, bool& vecw) {
    ST_CHECK_ARG(A && W && C, "st_gemm_fwd: null pointer");
    ST_CHECK_ARG(Bn > 0 && Tin > 0 && Tout > 0 && Cin > 0 && N > 0 && KT > 0 && pad >= 0 && stride >= 1, "st_gemm_fwd: bad dims");
    ST_CHECK_ARG(lda >= Cin && ldc >= coff + N, "st_gemm_fwd: lda=%d < Cin=%d or ldc=%d < coff+N=%d", lda, Cin, ldc, coff + N);
    // rows of A outside [0, Tin) read as zero, so Tout may exceed the natural conv length (used by the
    // input-gradient pass, where the forward output was trimmed)
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.W = W; g.C = C; g.ldc = ldc; g.coff = coff;
    g.Bn = Bn; g.Tin = Tin; g.Tout = Tout; g.Cin = Cin; g.N = N; g.KT = KT; g.pad = pad; g.pool_prev = pool_prev; g.stride = stride;
    g.M = Bn * Tout;
    g.cpb = (Cin + GM_BK - 1) / GM_BK;
    if (ep) {
        g.ep = *ep;
        ST_CHECK_ARG(!ep->highway_h || ep->res, "st_gemm_fwd: highway epilogue needs res (the layer input)");
        ST_CHECK_ARG(!ep->bn_mean || ep->bn_var, "st_gemm_fwd: bn_mean without bn_var");
    }
    veca = st_aligned16(A) && (lda % 4 == 0) && (Cin % 4 == 0);
    const bool tap_major = ep && ep->w_tap_major && KT > 1;
    ST_CHECK_ARG(!tap_major || (st_aligned16(W) && Cin % 4 == 0), "st_gemm_fwd: tap-major weights need Cin %% 4 == 0 and 16-byte alignment");
    vecw = (KT == 1 || tap_major) && st_aligned16(W) && (Cin % 4 == 0);
    return 0;
}

// Several independent GEMM / conv jobs in ONE launch (the CBHG conv bank: K convolutions of the same input, src/module.py:590-598).
// Jobs the pipelined kernel can take without pooling or split-K go out in launches of up to GM_MAX_BATCH jobs, longest reduction
// first; a batch with any other job runs as separate st_gemm_fwd launches.
extern "C" int st_gemm_fwd_batch(const st_gemm_job* jobs, int n, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(jobs && n > 0 && n <= 64, "st_gemm_fwd_batch: bad arguments");
    GmArgs g[64];
    bool all_ok = true;
    for (int j = 0; j < n; ++j) {
        const st_gemm_job& q = jobs[j];
        bool veca, vecw;
        const int rc = gm_prepare(q.A, q.lda, q.W, q.C, q.ldc, q.coff, q.Bn, q.Tin, q.Tout, q.Cin, q.N, q.KT, q.pad, q.stride, q.pool_prev,
                                  &q.ep, g[j], veca, vecw);
        if (rc) return rc;
        all_ok = all_ok && veca && vecw && !q.pool_prev && !(q.ep.splitk_ws && q.ep.splitk_slabs > 1);
    }
    if (!all_ok || n == 1) {
        for (int j = 0; j < n; ++j) {
            const st_gemm_job& q = jobs[j];
            const int rc = st_gemm_fwd(q.A, q.lda, q.W, q.C, q.ldc, q.coff, q.Bn, q.Tin, q.Tout, q.Cin, q.N, q.KT, q.pad, q.stride, q.pool_prev,
                                       &q.ep, stream);
            if (rc) return rc;
        }
        return 0;
    }
    int order[64];
    for (int j = 0; j < n; ++j) order[j] = j;
    for (int i = 1; i < n; ++i)        // longest reductions first (insertion sort, stable)
        for (int j = i; j > 0 && g[order[j]].KT * g[order[j]].cpb > g[order[j - 1]].KT * g[order[j - 1]].cpb; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    for (int c0 = 0; c0 < n; c0 += GM_MAX_BATCH) {
        const int cnt = n - c0 < GM_MAX_BATCH ? n - c0 : GM_MAX_BATCH;
        GmBatch b;
        memset(&b, 0, sizeof(b));
        int maxM = 0, maxN = 0;
        for (int j = 0; j < cnt; ++j) {
            b.g[j] = g[order[c0 + j]];
            maxM = b.g[j].M > maxM ? b.g[j].M : maxM;
            maxN = b.g[j].N > maxN ? b.g[j].N : maxN;
        }
        // LDS-DMA kernel when every job can take it (see st_gemm_fwd); 64 x 32 tiles for the narrow outputs of the conv bank
        bool dma = true, narrow = true;
        for (int j = 0; j < cnt; ++j) {
            dma = dma && (b.g[j].KT == 1 || b.g[j].lda == b.g[j].Cin);
            narrow = narrow && gd_tile(b.g[j].M, b.g[j].N, 1) == 2;
        }
        if (dma && narrow) hipLaunchKernelGGL((gd_batch_kernel<64, 32>), dim3((maxM + 63) / 64, (maxN + 31) / 32, cnt), dim3(256), 0, (hipStream_t)stream, b);
        else if (dma) hipLaunchKernelGGL((gd_batch_kernel<64, 64>), dim3((maxM + 63) / 64, (maxN + 63) / 64, cnt), dim3(256), 0, (hipStream_t)stream, b);
        else hipLaunchKernelGGL((gm_pipe_batch_kernel<true, 2>), dim3((maxM + 63) / 64, (maxN + GM_BN - 1) / GM_BN, cnt), dim3(GM_THREADS), 0, (hipStream_t)stream, b);
        ST_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int st_gemm_fwd(const float* A, int lda, const float* W, float* C, int ldc, int coff,
                           int Bn, int Tin, int Tout, int Cin, int N, int KT, int pad, int stride, int pool_prev,
                           const st_gemm_epilogue* ep, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    GmArgs g;
    bool veca, vecw;
    { const int rc = gm_prepare(A, lda, W, C, ldc, coff, Bn, Tin, Tout, Cin, N, KT, pad, stride, pool_prev, ep, g, veca, vecw); if (rc) return rc; }
    dim3 grid((g.M + GM_BM - 1) / GM_BM, (N + GM_BN - 1) / GM_BN);
    hipStream_t st = (hipStream_t)stream;
    // split-K: the caller passes slabs = st_gemm_splitk_slabs(...) and a workspace of slabs * M * N floats in the epilogue struct
    const int S = (ep && ep->splitk_ws && veca) ? ep->splitk_slabs : 1;
    if (S > 1) {
        ST_CHECK_ARG(S == st_gemm_splitk_slabs(Bn, Tout, Cin, N, KT) && st_aligned16(ep->splitk_ws), "st_gemm_fwd: splitk_slabs=%d does not match st_gemm_splitk_slabs()", S);
        g.kb_per_split = (KT * g.cpb + S - 1) / S;
        g.split_ws = ep->splitk_ws;
        grid.z = S;
    }
    if (veca && vecw && !pool_prev && (KT == 1 || lda == Cin)) {      // the LDS-DMA kernel (gd_kernel)
        if (S > 1) g.kb_per_split = ((KT * Cin + 31) / 32 + S - 1) / S;          // (in 32-float chunks)
        const int tile = gd_tile(g.M, N, S);
        if (tile == 0) hipLaunchKernelGGL((gd_kernel<64, 64>), dim3((g.M + 63) / 64, (N + 63) / 64, S), dim3(256), 0, st, g);
        else if (tile == 1) hipLaunchKernelGGL((gd_kernel<32, 64>), dim3((g.M + 31) / 32, (N + 63) / 64, S), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gd_kernel<64, 32>), dim3((g.M + 63) / 64, (N + 31) / 32, S), dim3(256), 0, st, g);
    } else if (veca) {      // (16-byte addressable A: Cin % 4 == 0, so Cin >= 4) the pipelined kernel; otherwise the generic one-block form
        // 32-row tiles when 64-row tiles would leave compute units with fewer than two workgroups (st_device_info: 256 CUs)
        const bool small = S == 1 && (size_t)grid.x * grid.y < 512;
        const dim3 grid32((g.M + 31) / 32, grid.y);
#define GM_LAUNCH(VW, PL) do { if (small) hipLaunchKernelGGL((gm_pipe_kernel<VW, PL, 1>), grid32, dim3(GM_THREADS), 0, st, g); \
                               else hipLaunchKernelGGL((gm_pipe_kernel<VW, PL, 2>), grid, dim3(GM_THREADS), 0, st, g); } while (0)
        if (vecw && pool_prev) GM_LAUNCH(true, true);
        else if (vecw) GM_LAUNCH(true, false);
        else if (pool_prev) GM_LAUNCH(false, true);
        else GM_LAUNCH(false, false);
#undef GM_LAUNCH
    } else if (!pool_prev && Cin >= 4 && S == 1) {
        // rows that are not 16-byte addressable: the pipelined kernel in its element-wise form.  W counts as "adjacent in ci" when it is
        // a Linear weight or tap-major, whatever its alignment
        const bool adj = KT == 1 || (ep && ep->w_tap_major);
        // (64-row tiles whatever the grid: with 32-row tiles half the threads sit out the scalar A loads -- 124 vs 68 us measured)
        if (adj) hipLaunchKernelGGL((gm_pipe_kernel<true, false, 2, false>), grid, dim3(GM_THREADS), 0, st, g);
        else hipLaunchKernelGGL((gm_pipe_kernel<false, false, 2, false>), grid, dim3(GM_THREADS), 0, st, g);
    } else if (vecw) hipLaunchKernelGGL((gm_kernel<false, true>), grid, dim3(GM_THREADS), 0, st, g);
    else hipLaunchKernelGGL((gm_kernel<false, false>), grid, dim3(GM_THREADS), 0, st, g);
    ST_LAUNCH_CHECK();
    if (S > 1) {
        const bool vec4 = N % 4 == 0;
        const size_t items = (size_t)g.M * N / (vec4 ? 4 : 1);
        size_t blocks = (items + 255) / 256;
        if (blocks > 4096) blocks = 4096;
#define GM_FIN(SS) hipLaunchKernelGGL((gm_splitk_finish_kernel<true, SS>), dim3((unsigned)blocks), dim3(256), 0, st, g, S)
        if (vec4) switch (S) { case 2: GM_FIN(2); break; case 3: GM_FIN(3); break; case 4: GM_FIN(4); break; case 5: GM_FIN(5); break;
                               case 6: GM_FIN(6); break; case 7: GM_FIN(7); break; case 8: GM_FIN(8); break; default: GM_FIN(0); }
        else hipLaunchKernelGGL((gm_splitk_finish_kernel<false, 0>), dim3((unsigned)blocks), dim3(256), 0, st, g, S);
#undef GM_FIN
        ST_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int st_bn_stats(const float* X, int ldx, int coff, int M, int N, float* mean_out, float* var_out,
                           float* run_mean, float* run_var, float momentum, long long* batches_tracked, float* ws, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(X && mean_out && var_out && M > 0 && N > 0, "st_bn_stats: bad arguments");
    ST_CHECK_ARG((run_mean == nullptr) == (run_var == nullptr), "st_bn_stats: run_mean/run_var must both be given");
    ST_CHECK_ARG(ws, "st_bn_stats: null workspace (st_colreduce_workspace_floats)");
    const int chunks = st_colreduce_chunks(M);
    const int rpc = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(bn_stats_chunk_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream,
                       X, ldx, coff, M, N, rpc, ws);
    ST_LAUNCH_CHECK();
    ST_CHECK_ARG(chunks <= 128, "st_bn_stats: %d chunks (the merge kernel holds at most 128)", chunks);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                       ws, chunks, rpc, M, N, mean_out, var_out, run_mean, run_var, momentum, batches_tracked, 0);
    ST_LAUNCH_CHECK();
    return 0;
}

// SyncBN, local half: rec = (mean[N], M2[N], row count) of THIS rank's rows -- what the ranks exchange in one all-gather
extern "C" int st_bn_stats_record(const float* X, int ldx, int coff, int M, int N, float* rec, long long* batches_tracked, float* ws,
                                  void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(X && rec && ws && M > 0 && N > 0, "st_bn_stats_record: bad arguments");
    const int chunks = st_colreduce_chunks(M);
    const int rpc = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(bn_stats_chunk_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, X, ldx, coff, M, N, rpc, ws);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                       ws, chunks, rpc, M, N, rec, rec + N, nullptr, nullptr, 0.0f, batches_tracked, 1);
    ST_LAUNCH_CHECK();
    return 0;
}

// SyncBN, merged half: the gathered records of all ranks (world, 2N + 1) -> the statistics of the global batch, merged exactly and in
// rank order (Chan et al.), identically on every rank; running statistics updated as nn.BatchNorm1d does (unbiased variance over the
// GLOBAL row count); inv_total_out = 1 / (global row count), the factor the backward's two sums are divided by
extern "C" int st_bn_sync_merge(const float* rec, int world, int N, float* mean_out, float* var_out, float* run_mean, float* run_var,
                                float momentum, float* inv_total_out, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(rec && mean_out && var_out && inv_total_out && world > 0 && N > 0, "st_bn_sync_merge: bad arguments");
    ST_CHECK_ARG((run_mean == nullptr) == (run_var == nullptr), "st_bn_sync_merge: run_mean/run_var must both be given");
    hipLaunchKernelGGL(bn_sync_merge_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, rec, world, N, mean_out, var_out,
                       run_mean, run_var, momentum, inv_total_out);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_highway_stack_supported(int C, int n_layers) { return C > 0 && C % 16 == 0 && C <= HW_MAXC && n_layers >= 1 && n_layers <= HW_MAXL; }

extern "C" int st_highway_stack_fwd(const float* x, int ldx, const float* const* w_h, const float* const* b_h, const float* const* w_t,
                                    const float* const* b_t, int n_layers, float* y, int ldy, int M, int C, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(x && y && w_h && w_t && M > 0 && ldx >= C && ldy >= C, "st_highway_stack_fwd: bad arguments");
    ST_CHECK_ARG(st_highway_stack_supported(C, n_layers), "st_highway_stack_fwd: C=%d (a multiple of 16, <= %d), layers=%d (<= %d)", C, HW_MAXC, n_layers, HW_MAXL);
    ST_CHECK_ARG(st_aligned16(x) && st_aligned16(y) && ldx % 4 == 0 && ldy % 4 == 0, "st_highway_stack_fwd: rows must be 16-byte aligned");
    HwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.M = M; a.C = C; a.NL = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        ST_CHECK_ARG(w_h[l] && w_t[l] && st_aligned16(w_h[l]) && st_aligned16(w_t[l]), "st_highway_stack_fwd: layer %d weights", l);
        a.wh[l] = w_h[l]; a.wt[l] = w_t[l]; a.bh[l] = b_h ? b_h[l] : nullptr; a.bt[l] = b_t ? b_t[l] : nullptr;
    }
    const int LD = C + 4;
    const size_t lds = (size_t)(2 * HW_ROWS * LD + 2 * C * LD) * sizeof(float);
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(highway_stack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(highway_stack_kernel, dim3((M + HW_ROWS - 1) / HW_ROWS), dim3(256), lds, (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_bn_apply(float* X, int ldx, int coff, int M, int N, const float* mean, const float* var,
                           const float* w, const float* b, float eps, int act, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(X && mean && var && M > 0 && N > 0, "st_bn_apply: bad arguments");
    const size_t total = (size_t)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       X, ldx, coff, M, N, mean, var, w, b, eps, act);
    ST_LAUNCH_CHECK();
    return 0;
}
