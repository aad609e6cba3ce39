// grad.hip -- backward-pass building blocks for gfx950 (MI355X), training path.
//
//  * weight gradient of a conv1d / linear:  dW[n][ci][tap] = sum_rows dC[row][n] * A[row + tap - pad][ci]
//    ("TN" GEMM: the reduction runs over the (utterance, frame) rows).  fp32-MFMA tiles whose columns are
//    interleaved so that neither operand is ever transposed (see tn_kernel).  Rows are split over blockIdx.z
//    into partial slabs that a second kernel adds in a fixed order (deterministic, no atomics).
//  * the input gradient of a conv/linear is the FORWARD kernel (st_gemm_fwd) on dC with the
//    weight transposed and tap-flipped by the caller, so nothing is needed here for it.
//  * column sums (bias gradients), BatchNorm backward (training statistics), activation / dropout
//    backward, Highway backward, the fused max-pool backward, strided copies and row scatter-add.
#include "st_common.h"

namespace {

constexpr int TN_T = 64, TN_BK = 16, TN_THREADS = 256;

struct TnArgs {
    const float* dC; int lddc; int dcoff;       // (Bn*Tout, >= dcoff+N)
    const float* A; int lda;                    // (Bn*Tin, Cin)
    float* part;                                // [Z][N][Cin][KT]
    int Bn, Tin, Tout, Cin, N, KT, pad, pool_prev;
    int M, rows_per_z;
    int fold;      // few input channels: the taps are folded into the column axis (column = ci*KT + tap)
    int vecx, vecy;   // rows of dC / A are 16-byte aligned: one 16-byte load per thread instead of four scalar ones
    int lin;          // KT == 1, pad == 0, Tin == Tout, no pooling: row m of A pairs with row m of dC (no per-chunk integer division)
    float* db_part;   // optional [Z][N]: column sums of dC over this slab's rows (the bias gradient), formed by the workgroups of the FIRST
                      // column block from the dC chunks they stage anyway -- no colsum launches next to the product
};

// TM = tile edge (64 or 128): 4 waves as 2x2, each (TM/2) x (TM/2) = FR x FR MFMA 16x16 tiles, FR = TM/32.  The 128 tile
// halves the LDS reads per MFMA and is used when both N and the column count reach 128.
//
// No transposition anywhere: a 16-row chunk of dC / A is staged into LDS AS IT LIES IN MEMORY ([row][column], 16-byte
// loads and stores, 16 adjacent lanes = 256 contiguous bytes of one row), and the MFMA tiles take INTERLEAVED columns:
// tile t of a wave owns columns {FR r + t : r = 0..15} of the wave's 16 FR columns.  Lane (r, q) then reads, for k-step c,
// ONE FR-vector at [row 4q + c][FR r ..] that carries its operand of all FR tiles (the reduction index of MFMA slot
// (q, c) is row 4q + c for both operands), and in the result lane (r, q) holds, per output row, FR CONSECUTIVE columns
// (one from each tile) -- a 16-byte store.  (r01/r02 form: both operands transposed while staging with 16 scalar LDS
// stores per thread and chunk; the LSTM weight gradients ran at 52 TFLOP/s.)
template <int TM>
__global__ __launch_bounds__(TN_THREADS) void tn_kernel(const TnArgs g) {
    constexpr int FR = TM / 32;                                            // MFMA tiles per wave and dimension
    constexpr int LD = TM + (TM == 64 ? 8 : 0);                            // 64: rows 4 apart must not share banks (8-byte reads)
    typedef float frag_t __attribute__((ext_vector_type(FR)));
    __shared__ __attribute__((aligned(16))) float Xs[2][TN_BK * LD];   // [m][n]
    __shared__ __attribute__((aligned(16))) float Ys[2][TN_BK * LD];   // [m][ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * TM;
    const int cblocks = (g.Cin + TM - 1) / TM;
    const int tap = g.fold ? 0 : blockIdx.y / cblocks;
    const int c0 = g.fold ? blockIdx.y * TM : (blockIdx.y - tap * cblocks) * TM;
    const int ncols = g.fold ? g.Cin * g.KT : g.Cin;
    const int z = blockIdx.z;
    const int mbeg = z * g.rows_per_z, mend = min(g.M, mbeg + g.rows_per_z);

    // staging role: row tid/16 of the 16-row chunk, 4 consecutive columns (+64 for the second half of a 128 tile)
    const int sm = tid >> 4, sc = (tid & 15) * 4;
    f32x4 acc[FR][FR];
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_x = [&](int m, int cofs) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m >= mend) return v;
        const float* p = g.dC + (size_t)m * g.lddc + g.dcoff + n0 + sc + cofs;
        const int rem = g.N - (n0 + sc + cofs);
        if (g.vecx && rem >= 4) return st_ld4(p);
        if (rem > 0) v[0] = p[0];
        if (rem > 1) v[1] = p[1];
        if (rem > 2) v[2] = p[2];
        if (rem > 3) v[3] = p[3];
        return v;
    };
    auto load_y = [&](int m, int cofs) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m >= mend) return v;
        if (g.lin) {           // a Linear layer: plain rows
            const int ci = c0 + sc + cofs;
            if (ci >= g.Cin) return v;
            const float* p = g.A + (size_t)m * g.lda + ci;
            const int rem = g.Cin - ci;
            if (g.vecy && rem >= 4) return st_ld4(p);
            v[0] = p[0];
            if (rem > 1) v[1] = p[1];
            if (rem > 2) v[2] = p[2];
            if (rem > 3) v[3] = p[3];
            return v;
        }
        const int b = m / g.Tout, to = m - b * g.Tout;
        if (g.fold) {          // four consecutive (ci, tap) columns, each its own row shift
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = c0 + sc + cofs + j;
                if (cc >= ncols) continue;
                const int ci = cc / g.KT, tp = cc - ci * g.KT;
                const int ti = to + tp - g.pad;
                if (ti >= 0 && ti < g.Tin) v[j] = g.A[((size_t)b * g.Tin + ti) * g.lda + ci];
            }
            return v;
        }
        const int ti = to + tap - g.pad;
        const int ci = c0 + sc + cofs;
        if (ti < 0 || ti >= g.Tin || ci >= g.Cin) return v;
        const float* p = g.A + ((size_t)b * g.Tin + ti) * g.lda + ci;
        const int rem = g.Cin - ci;
        if (g.vecy && rem >= 4 && !g.pool_prev) return st_ld4(p);
        v[0] = p[0];
        if (rem > 1) v[1] = p[1];
        if (rem > 2) v[2] = p[2];
        if (rem > 3) v[3] = p[3];
        if (g.pool_prev && ti > 0) {
            const float* q = p - g.lda;
            v[0] = fmaxf(v[0], q[0]);
            if (rem > 1) v[1] = fmaxf(v[1], q[1]);
            if (rem > 2) v[2] = fmaxf(v[2], q[2]);
            if (rem > 3) v[3] = fmaxf(v[3], q[3]);
        }
        return v;
    };

    constexpr int NH = TM / 64;                                            // column halves staged per thread
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 rx[NH], ry[NH];
    // Linear layers with 16-byte addressable rows whose column pieces are whole (or wholly outside): running row pointers, one compare
    // and one 16-byte load per piece and chunk -- the general path's address arithmetic (a 64-bit multiply, the utterance / frame split of
    // the row index, the remainder logic) costs as many issue cycles per chunk as the 16 MFMAs of a 64-tile wave
    const float* px[NH]; const float* py[NH];
    bool okx[NH], oky[NH];
    bool whole = true;          // (16-byte loads also from rows that are only 4-byte aligned: st_ld4_u)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int colx = n0 + sc + h * 64, ci = c0 + sc + h * 64;
        okx[h] = colx + 4 <= g.N; oky[h] = ci + 4 <= g.Cin;
        whole = whole && (okx[h] || colx >= g.N) && (oky[h] || ci >= g.Cin);
        px[h] = g.dC + (size_t)(mbeg + sm) * g.lddc + g.dcoff + colx;
        py[h] = g.A + (size_t)(mbeg + sm) * g.lda + ci;
    }
    const bool fastp = whole && g.lin;
    // ... and convolutions (one tap per workgroup, no pooling): the (utterance, frame) of the thread's row is carried from chunk to chunk
    const bool fastc = whole && !g.lin && !g.fold;          // (a pooled input: two 16-byte loads and a max per piece)
    int mreq = mbeg + sm;                      // (fast paths: the row the next request takes; requests come in chunk order)
    int cb = 0, cto = 0;
    if (fastc) { cb = mreq / g.Tout; cto = mreq - cb * g.Tout; }
    auto request = [&](int m) __attribute__((always_inline)) {
        if (fastc) {
            const bool row = mreq < mend;
            const int ti = cto + tap - g.pad;
            const bool yrow = row && ti >= 0 && ti < g.Tin;
            const float* pa = g.A + ((size_t)cb * g.Tin + (yrow ? ti : 0)) * g.lda + c0 + sc;
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                rx[h] = (row && okx[h]) ? st_ld4_u(px[h]) : z4;
                ry[h] = (yrow && oky[h]) ? st_ld4_u(pa + h * 64) : z4;
                if (g.pool_prev && yrow && ti > 0 && oky[h]) {       // MaxPool1d(2, stride 1, padding 1)[:T] of the forward, fused into the load
                    const f32x4 q = st_ld4_u(pa - g.lda + h * 64);
                    ry[h] = f32x4{fmaxf(ry[h][0], q[0]), fmaxf(ry[h][1], q[1]), fmaxf(ry[h][2], q[2]), fmaxf(ry[h][3], q[3])};
                }
                px[h] += (size_t)TN_BK * g.lddc;
            }
            mreq += TN_BK; cto += TN_BK;
            while (cto >= g.Tout) { cto -= g.Tout; ++cb; }
            return;
        }
        if (fastp) {
            const bool row = mreq < mend;
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                rx[h] = (row && okx[h]) ? st_ld4_u(px[h]) : z4;
                ry[h] = (row && oky[h]) ? st_ld4_u(py[h]) : z4;
                px[h] += (size_t)TN_BK * g.lddc; py[h] += (size_t)TN_BK * g.lda;
            }
            mreq += TN_BK;
            return;
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) { rx[h] = load_x(m + sm, h * 64); ry[h] = load_y(m + sm, h * 64); }
    };
    const bool do_db = g.db_part != nullptr && blockIdx.y == 0;
    f32x4 dbacc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) dbacc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            *reinterpret_cast<f32x4*>(Xs[buf] + sm * LD + h * 64 + sc) = rx[h];
            *reinterpret_cast<f32x4*>(Ys[buf] + sm * LD + h * 64 + sc) = ry[h];
            if (do_db) dbacc[h] = dbacc[h] + rx[h];          // (rows past the slab were loaded as zeros)
        }
    };
    // two LDS buffers: chunk k+1 is written while chunk k is multiplied, one (LDS-only) barrier per chunk; the global loads of
    // chunk k+2 are in flight across it
    request(mbeg);
    commit(0);
    request(mbeg + TN_BK);
    st_lds_barrier();
    int buf = 0;
    for (int m = mbeg; m < mend; m += TN_BK, buf ^= 1) {
        commit(buf ^ 1);
        request(m + 2 * TN_BK);
        frag_t a4[4], b4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a4[c] = *reinterpret_cast<const frag_t*>(Xs[buf] + (4 * fq + c) * LD + wm * (TM / 2) + FR * fr);
            b4[c] = *reinterpret_cast<const frag_t*>(Ys[buf] + (4 * fq + c) * LD + wn * (TM / 2) + FR * fr);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mt = 0; mt < FR; ++mt)
#pragma unroll
                for (int nt = 0; nt < FR; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c][mt], b4[c][nt], acc[mt][nt], 0, 0, 0);
        st_lds_barrier();
    }
    if (do_db) {   // the 16 row groups' column sums meet in LDS (the staging buffer: the loop ended with a barrier), fixed order
#pragma unroll
        for (int h = 0; h < NH; ++h) *reinterpret_cast<f32x4*>(Xs[0] + sm * LD + h * 64 + sc) = dbacc[h];
        st_lds_barrier();
        if (tid < TM && n0 + tid < g.N) {
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < TN_BK; ++r) s += Xs[0][r * LD + tid];
            g.db_part[(size_t)z * g.N + n0 + tid] = s;
        }
    }
    // result: lane (r, q) holds D[i = 4q + e][j = r] of tile (mt, nt): row n = .. + FR i + mt, columns .. + FR r + nt
    float* out = g.part + (size_t)z * g.N * g.Cin * g.KT;
    const int cib = c0 + wn * (TM / 2) + FR * fr;
    const bool vec_out = FR == 4 && (g.fold || g.KT == 1) && (ncols % 4 == 0) && cib + 3 < ncols && st_aligned16(out);
#pragma unroll
    for (int mt = 0; mt < FR; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wm * (TM / 2) + FR * (4 * fq + e) + mt;
            if (n >= g.N) continue;
            if (vec_out) {
                f32x4 v;
#pragma unroll
                for (int nt = 0; nt < FR; ++nt) v[nt] = acc[mt][nt][e];
                *reinterpret_cast<f32x4*>(out + (size_t)n * ncols + cib) = v;
                continue;
            }
#pragma unroll
            for (int nt = 0; nt < FR; ++nt) {
                const int ci = cib + nt;
                if (ci >= ncols) continue;
                if (g.fold) out[(size_t)n * ncols + ci] = acc[mt][nt][e];
                else out[((size_t)n * g.Cin + ci) * g.KT + tap] = acc[mt][nt][e];
            }
        }
}

// ---- tn_dma_kernel: the same product with both operands moved by LDS-DMA (round 4) ---------------------------------------------
// A chunk of RC = 32 rows of dC / A goes straight into LDS AS IT LIES IN MEMORY (global_load_lds_dwordx4: a wave instruction moves
// 4 rows x 256 bytes at TM = 64, 2 rows x 512 bytes at TM = 128): no staging registers, no ds_write pass, and -- with the MFMA tiles'
// interleaved columns -- no swizzle either (a lane's fragment of row 4q + c is at [row][FR r ..]: 16 lanes read 16 FR consecutive
// floats of one row; rows 4 apart fall on the same banks at TM = 64, a 2-way conflict on 8 ds_read_b64 per 32 MFMAs).  Three buffers,
// ONE barrier per 32 rows (the register-staged form: one per 16).  Pieces that do not exist (rows past the slab, frames outside
// the utterance, columns past N / Cin) read 16 zero bytes from a constant.  Shapes: what tn_kernel's running-pointer forms take
// (16-byte addressable rows, whole 4-column pieces, no fold, no fused max-pool); the k order per output element is tn_kernel's
// (rows ascending), so the results are bit-identical.
__device__ const f32x4 tn_zero4 = {0.f, 0.f, 0.f, 0.f};
typedef const __attribute__((address_space(1))) void* tn_gptr_t;
typedef __attribute__((address_space(3))) void* tn_lptr_t;

template <int TM, int RC = 32>
__device__ __forceinline__ void tn_dma_body(const TnArgs& g, const int bx, const int by, const int bz) {
    constexpr int FR = TM / 32;
    constexpr int PPR = TM / 4;                         // 16-byte pieces per row
    constexpr int RPI = TN_THREADS / PPR;               // rows one DMA pass of the workgroup covers (16 at TM = 64, 8 at 128)
    constexpr int NI = RC / RPI;                        // DMA instructions per thread, operand and chunk
    constexpr int BUF = 2 * RC * TM;                    // floats per buffer: X then Y
    typedef float frag_t __attribute__((ext_vector_type(FR)));
    extern __shared__ __attribute__((aligned(16))) float tn_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = bx * TM;
    const int cblocks = (g.Cin + TM - 1) / TM;
    const int tap = by / cblocks;
    const int c0 = (by - tap * cblocks) * TM;
    const int z = bz;
    const int mbeg = z * g.rows_per_z, mend = min(g.M, mbeg + g.rows_per_z);
    const int srow = tid / PPR, sc = (tid % PPR) * 4;   // staging role: piece sc of rows srow, srow + RPI, ...
    const float* zp = reinterpret_cast<const float*>(&tn_zero4);
    const bool okx = n0 + sc + 4 <= g.N, oky = c0 + sc + 4 <= g.Cin;
    const float* px = g.dC + (size_t)(mbeg + srow) * g.lddc + g.dcoff + n0 + sc;
    const float* py = g.A + (size_t)(mbeg + srow) * g.lda + c0 + sc;          // (Linear layers)
    int mreq = mbeg + srow;                             // the row this thread's first piece of the next chunk belongs to
    int cb[NI], cto[NI];                                // (convolutions: utterance / frame of each of the thread's rows)
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int m = min(mreq + i * RPI, g.M - 1); cb[i] = m / g.Tout; cto[i] = m - cb[i] * g.Tout; }
    auto issue = [&](int bufi) __attribute__((always_inline)) {
        float* xs = tn_lds + bufi * BUF;
        float* ys = xs + RC * TM;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool row = mreq + i * RPI < mend;
            const float* sx = (row && okx) ? px + (size_t)i * RPI * g.lddc : zp;
            const float* sy;
            if (g.lin) sy = (row && oky) ? py + (size_t)i * RPI * g.lda : zp;
            else {
                const int ti = cto[i] + tap - g.pad;
                sy = (row && oky && ti >= 0 && ti < g.Tin) ? g.A + ((size_t)cb[i] * g.Tin + ti) * g.lda + c0 + sc : zp;
                cto[i] += RC;
                while (cto[i] >= g.Tout) { cto[i] -= g.Tout; ++cb[i]; }
            }
            // (a wave instruction covers 64 / PPR consecutive rows: wave w of pass i starts at row i * RPI + w * (64 / PPR))
            __builtin_amdgcn_global_load_lds((tn_gptr_t)sx, (tn_lptr_t)(xs + (i * RPI + wave * (64 / PPR)) * TM), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((tn_gptr_t)sy, (tn_lptr_t)(ys + (i * RPI + wave * (64 / PPR)) * TM), 16, 0, 0);
        }
        px += (size_t)RC * g.lddc; py += (size_t)RC * g.lda; mreq += RC;
    };
    f32x4 acc[FR][FR];
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const bool do_db = g.db_part != nullptr && by == 0;
    f32x4 dbacc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) dbacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nchunks = (mend - mbeg + RC - 1) / RC;
    issue(0);
    issue(1);
    int bi = 0, bn = 2;
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NI) : "memory");        // chunk c has landed (this wave's pieces); c + 1 may be in flight
        st_lds_barrier();
        issue(bn);
        const float* xs = tn_lds + bi * BUF;
        const float* ys = xs + RC * TM;
        if (do_db) {                                    // the bias gradient: column sums of the dC rows this thread's pieces cover
#pragma unroll
            for (int i = 0; i < NI; ++i) dbacc[i] = dbacc[i] + *reinterpret_cast<const f32x4*>(xs + (i * RPI + srow) * TM + sc);
        }
#pragma unroll
        for (int h = 0; h < RC / 16; ++h) {
            frag_t a4[4], b4[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                a4[cc] = *reinterpret_cast<const frag_t*>(xs + (h * 16 + 4 * fq + cc) * TM + wm * (TM / 2) + FR * fr);
                b4[cc] = *reinterpret_cast<const frag_t*>(ys + (h * 16 + 4 * fq + cc) * TM + wn * (TM / 2) + FR * fr);
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int mt = 0; mt < FR; ++mt)
#pragma unroll
                    for (int nt = 0; nt < FR; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[cc][mt], b4[cc][nt], acc[mt][nt], 0, 0, 0);
        }
        bi = bi == 2 ? 0 : bi + 1;
        bn = bn == 2 ? 0 : bn + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (do_db) {   // the row groups' column sums meet in LDS (buffer 0: everybody is past the loop after this barrier), fixed order
        st_lds_barrier();
        f32x4 t = dbacc[0];
#pragma unroll
        for (int i = 1; i < NI; ++i) t = t + dbacc[i];
        *reinterpret_cast<f32x4*>(tn_lds + srow * TM + sc) = t;
        st_lds_barrier();
        if (tid < TM && n0 + tid < g.N) {
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < RPI; ++r) s += tn_lds[r * TM + tid];
            g.db_part[(size_t)z * g.N + n0 + tid] = s;
        }
    }
    // result: lane (r, q) holds D[i = 4q + e][j = r] of tile (mt, nt): row n = .. + FR i + mt, columns .. + FR r + nt
    const int ncols = g.Cin;
    float* out = g.part + (size_t)z * g.N * g.Cin * g.KT;
    const int cib = c0 + wn * (TM / 2) + FR * fr;
    const bool vec_out = FR == 4 && g.KT == 1 && (ncols % 4 == 0) && cib + 3 < ncols && st_aligned16(out);
#pragma unroll
    for (int mt = 0; mt < FR; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wm * (TM / 2) + FR * (4 * fq + e) + mt;
            if (n >= g.N) continue;
            if (vec_out) {
                f32x4 v;
#pragma unroll
                for (int nt = 0; nt < FR; ++nt) v[nt] = acc[mt][nt][e];
                *reinterpret_cast<f32x4*>(out + (size_t)n * ncols + cib) = v;
                continue;
            }
#pragma unroll
            for (int nt = 0; nt < FR; ++nt) {
                const int ci = cib + nt;
                if (ci >= ncols) continue;
                out[((size_t)n * g.Cin + ci) * g.KT + tap] = acc[mt][nt][e];
            }
        }
}

template <int TM>
__global__ __launch_bounds__(TN_THREADS) void tn_dma_kernel(const TnArgs g) { tn_dma_body<TM>(g, blockIdx.x, blockIdx.y, blockIdx.z); }
// 16-row chunks: three buffers of 2 x 16 x TM floats (48 KB at TM = 128, 24 KB at 64) -- three workgroups of the 128-tile form share a
// compute unit instead of one (96 KB), and the waves of one cover the barrier and the DMA waits of another: the decoder LSTMs' weight
// gradients 932 -> 840 us.  Same k order per output element (rows ascending): bit-identical results.
template <int TM>
__global__ __launch_bounds__(TN_THREADS) void tn_dma_rc16_kernel(const TnArgs g) { tn_dma_body<TM, 16>(g, blockIdx.x, blockIdx.y, blockIdx.z); }

// ---- weight gradient of a convolution with very few input channels (the attention's location conv: 2 channels x 31 taps -> 32 filters,
// over steps x B sequences of L positions) --------------------------------------------------------------------------------------------------
// tn_kernel's folded form gathers its (channel, tap) columns one element at a time (an integer division and a dependent scalar load per
// element and 16-row chunk: 55 us for 0.46 GFLOP).  Here a workgroup stages a sequence's dC rows (64 at a time) and its input rows with
// their halo into LDS as they lie in memory and takes the im2col columns from LDS on the fly:
//   D[f][col = (ci, k)] += sum_t dC[t][f] * A[t + k - pad][ci]       wave w: column tile w, all filter tiles; reduction over t by 4.
// Workgroup z handles the (sequence, 64-row block) units z, z + Z, ... (two LDS buffers, one barrier per unit) and writes slab z; the
// slabs are summed in slab order (sum_partials_tall_kernel).  Shapes: N <= 64 filters (multiple of 4), Cin * KT <= 64 columns.
struct CsArgs { const float* dC; int lddc; const float* A; int lda; float* part; int Bn, Tin, Tout, Cin, N, KT, pad, Z; };
constexpr int CS_TB = 64;

__global__ __launch_bounds__(256) void convw_small_kernel(const CsArgs g) {
    __shared__ __attribute__((aligned(16))) float dcs[2][CS_TB * 68];          // [t][f], row stride 68: rows 4 apart on different banks
    __shared__ float as_[2][(CS_TB + 64) * 4];                                 // [t + halo][ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ar = lane & 15, q = lane >> 4;
    const int ncols = g.Cin * g.KT, NT = (g.N + 15) >> 4;
    const int col = wave * 16 + ar;
    const bool colok = col < ncols;
    const int ci = colok ? col / g.KT : 0, kk = colok ? col - ci * g.KT : 0;
    const int tblocks = (g.Tout + CS_TB - 1) / CS_TB;
    const int units = g.Bn * tblocks;
    const int halo = CS_TB + g.KT - 1;                                          // input rows a block of output rows reads
    const int pieces = g.N >> 2;                                               // 16-byte pieces per dC row
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging roles, fixed for the whole loop: up to four 16-byte pieces of dC and two input elements per thread and unit.  The next
    // unit's pieces are requested BEFORE the MFMAs of the current one and go to LDS after them (registers in between): a unit costs
    // its MFMAs, not a memory round trip.
    int pt[4], pp[4], hr[2], hc[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int i = tid + j * 256; pt[j] = i / pieces; pp[j] = (i - pt[j] * pieces) * 4; if (i >= CS_TB * pieces) pt[j] = -1; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int i = tid + j * 256; hr[j] = i / g.Cin; hc[j] = i - hr[j] * g.Cin; if (i >= halo * g.Cin) hr[j] = -1; }
    f32x4 rd[4];
    float ra[2];
    auto request = [&](int u) {
        const int b = u / tblocks, t0 = (u - b * tblocks) * CS_TB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pt[j] >= 0 && t0 + pt[j] < g.Tout) rd[j] = st_ld4(g.dC + ((size_t)b * g.Tout + t0 + pt[j]) * g.lddc + pp[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ti = t0 + hr[j] - g.pad;
            ra[j] = (hr[j] >= 0 && ti >= 0 && ti < g.Tin) ? g.A[((size_t)b * g.Tin + ti) * g.lda + hc[j]] : 0.0f;
        }
    };
    auto deposit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (pt[j] >= 0) *reinterpret_cast<f32x4*>(&dcs[buf][pt[j] * 68 + pp[j]]) = rd[j];
#pragma unroll
        for (int j = 0; j < 2; ++j) if (hr[j] >= 0) as_[buf][hr[j] * 4 + hc[j]] = ra[j];
    };
    int u = blockIdx.x, buf = 0;
    if (u < units) request(u);
    for (; u < units; u += g.Z) {
        deposit(buf);
        __syncthreads();                                   // unit u is in LDS; everybody is past the MFMAs of the unit before the last
        if (u + g.Z < units) request(u + g.Z);
        const float* dc = dcs[buf];
        const float* ap = as_[buf];
        const int t0 = (u % tblocks) * CS_TB;
        const int nsteps = (min(CS_TB, g.Tout - t0) + 3) >> 2;      // (rows past the sequence are zero: whole steps of them are skipped)
        for (int s = 0; s < nsteps; ++s) {
            const int t = 4 * s + q;
            const float bv = colok ? ap[(t + kk) * 4 + ci] : 0.0f;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                if (ft < NT) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(dc[t * 68 + ft * 16 + ar], bv, acc[ft], 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    // D[row = f = ft*16 + 4q + r][col]: slab z = blockIdx.x
    if (colok) {
        float* out = g.part + (size_t)blockIdx.x * g.N * ncols;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = ft * 16 + 4 * q + r;
                if (ft < NT && f < g.N) out[(size_t)f * ncols + col] = acc[ft][r];
            }
        }
    }
}

// Several weight-gradient products in ONE launch (st_gemm_wgrad_batch): the K convolutions of the CBHG bank (80 x 80 x k matrices: 4 k
// tiles each -- eight launches of 10 ... 36 us that each leave most of the chip idle) or the two directions of a recurrent layer.
// blockIdx.x runs through the jobs' own (x, y, z) grids one after the other.
constexpr int TN_MAXJ = 16;
struct TnBatch { TnArgs g[TN_MAXJ]; int blk0[TN_MAXJ + 1]; int gx[TN_MAXJ], gy[TN_MAXJ]; int n; };

template <int TM>
__global__ __launch_bounds__(TN_THREADS) void tn_dma_batch_kernel(const TnBatch b) {
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.blk0[j + 1]) ++j;
    const int r = (int)blockIdx.x - b.blk0[j];
    const int gx = b.gx[j], gxy = gx * b.gy[j];
    const int bz = r / gxy, rr = r - bz * gxy, by = rr / gx;
    tn_dma_body<TM>(b.g[j], rr - by * gx, by, bz);
}

// the slab sums of such a batch in one launch (each job's elements in its own block range; same per-element order as sum_partials*)
struct SumBatch { const float* part[TN_MAXJ]; float* out[TN_MAXJ]; const float* part_b[TN_MAXJ]; float* out_b[TN_MAXJ];
                  unsigned per[TN_MAXJ]; int nb[TN_MAXJ], Z[TN_MAXJ], acc[TN_MAXJ]; int blk0[TN_MAXJ + 1]; int n; };

__global__ __launch_bounds__(256) void sum_partials_batch_kernel(const SumBatch b) {
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.blk0[j + 1]) ++j;
    const size_t n = b.per[j], nb = (size_t)b.nb[j];
    const int Z = b.Z[j], nblk = b.blk0[j + 1] - b.blk0[j];
    const float* part = b.part[j]; const float* part_b = b.part_b[j];
    for (size_t i = (size_t)((int)blockIdx.x - b.blk0[j]) * blockDim.x + threadIdx.x; i < n + nb; i += (size_t)nblk * blockDim.x) {
        const bool isb = i >= n;
        const float* p = isb ? part_b + (i - n) : part + i;
        const size_t stride = isb ? nb : n;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = 0;
        for (; z + 4 <= Z; z += 4) {
            s0 += p[(size_t)z * stride]; s1 += p[(size_t)(z + 1) * stride]; s2 += p[(size_t)(z + 2) * stride]; s3 += p[(size_t)(z + 3) * stride];
        }
        for (; z < Z; ++z) s0 += p[(size_t)z * stride];
        const float sm = (s0 + s1) + (s2 + s3);
        if (isb) b.out_b[j][i - n] = sm;
        else b.out[j][i] = b.acc[j] ? b.out[j][i] + sm : sm;
    }
}

// out[i] (+)= sum_z part[z][i]   (fixed order)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* part, float* out, size_t n, int Z, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // fixed order: four interleaved running sums (loads of four slabs in flight), then a fixed combination
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = 0;
        for (; z + 4 <= Z; z += 4) {
            s0 += part[(size_t)z * n + i]; s1 += part[(size_t)(z + 1) * n + i];
            s2 += part[(size_t)(z + 2) * n + i]; s3 += part[(size_t)(z + 3) * n + i];
        }
        for (; z < Z; ++z) s0 += part[(size_t)z * n + i];
        const float s = (s0 + s1) + (s2 + s3);
        out[i] = accumulate ? out[i] + s : s;
    }
}

// very many slabs of a small matrix (the location conv's gradient: 256 slabs of 2k elements): with one thread per element the sum is a
// chain of Z dependent-latency loads on a handful of workgroups.  Here a workgroup takes 32 elements, its eight 32-lane groups every
// eighth slab each (four interleaved sums), and the eight partial sums meet in LDS in group order.
__global__ __launch_bounds__(256) void sum_partials_tall_kernel(const float* part, float* out, size_t n, int Z, int accumulate) {
    __shared__ float red[8][32];
    const int e = threadIdx.x & 31, zg = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + e;
    const bool ok = i < n;
    const float* p = part + (ok ? i : 0);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = zg;
    for (; z + 24 < Z; z += 32) { s0 += p[(size_t)z * n]; s1 += p[(size_t)(z + 8) * n]; s2 += p[(size_t)(z + 16) * n]; s3 += p[(size_t)(z + 24) * n]; }
    for (; z < Z; z += 8) s0 += p[(size_t)z * n];
    red[zg][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zg == 0 && ok) {
        const float s = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
        out[i] = accumulate ? out[i] + s : s;
    }
}

// the same for a product with its bias gradient: elements [0, n) go to out, [n, n + nb) to out_b (slabs part_b[z][nb]; never accumulated)
// SPLIT: the (rows, cols) result leaves as two matrices, columns [0, split) to out (rows, split) and [split, cols) to out1
// (rows, cols - split): the gradient of a product over concatenated inputs [W_ih | W_hh] goes straight to the two parameters'
// own gradient tensors (st_gemm_wgrad_split) instead of one tensor that is then cut up by two copies.
template <bool SPLIT>
__global__ __launch_bounds__(256) void sum_partials2_kernel(const float* part, float* out, size_t n, int Z, int accumulate,
                                                            const float* part_b, float* out_b, size_t nb,
                                                            float* out1, int cols, int split, float* out_b_dup) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + nb; i += (size_t)gridDim.x * blockDim.x) {
        const bool isb = i >= n;
        const float* p = isb ? part_b + (i - n) : part + i;
        const size_t stride = isb ? nb : n;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = 0;
        for (; z + 4 <= Z; z += 4) {
            s0 += p[(size_t)z * stride]; s1 += p[(size_t)(z + 1) * stride]; s2 += p[(size_t)(z + 2) * stride]; s3 += p[(size_t)(z + 3) * stride];
        }
        for (; z < Z; ++z) s0 += p[(size_t)z * stride];
        const float s = (s0 + s1) + (s2 + s3);
        if (isb) { out_b[i - n] = s; if (SPLIT && out_b_dup) out_b_dup[i - n] = s; continue; }
        float* o = out + i;
        if (SPLIT) {
            const size_t row = i / (size_t)cols;
            const int col = (int)(i - row * (size_t)cols);
            o = col < split ? out + row * (size_t)split + col : out1 + row * (size_t)(cols - split) + (col - split);
        }
        *o = accumulate ? *o + s : s;
    }
}

// column sums over M rows, optionally of X * Y:  out[n] (+)= sum_m X[m][n] (* Y[m][n]).  Rows are split into chunks
// over blockIdx.y (partials [chunk][N]); colsum_final adds the chunks in a fixed order.
__global__ __launch_bounds__(256) void colsum_kernel(const float* X, int ldx, int xoff, const float* Y, int ldy, int yoff,
                                                     int M, int N, int rows_per_chunk, float* part) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    const float* __restrict__ xp = X + xoff + min(n, N - 1);
    const float* __restrict__ yp = Y ? Y + yoff + min(n, N - 1) : nullptr;
    // four independent running sums (rows 16 apart): four (eight with Y) loads in flight per thread
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = m0 + rl;
    if (yp) {
        for (; m + 12 < m1; m += 16) {
            const float x0 = xp[(size_t)m * ldx], x1 = xp[(size_t)(m + 4) * ldx], x2 = xp[(size_t)(m + 8) * ldx], x3 = xp[(size_t)(m + 12) * ldx];
            const float y0 = yp[(size_t)m * ldy], y1 = yp[(size_t)(m + 4) * ldy], y2 = yp[(size_t)(m + 8) * ldy], y3 = yp[(size_t)(m + 12) * ldy];
            s0 = fmaf(x0, y0, s0); s1 = fmaf(x1, y1, s1); s2 = fmaf(x2, y2, s2); s3 = fmaf(x3, y3, s3);
        }
        for (; m < m1; m += 4) s0 = fmaf(xp[(size_t)m * ldx], yp[(size_t)m * ldy], s0);
    } else {
        for (; m + 12 < m1; m += 16) {
            s0 += xp[(size_t)m * ldx]; s1 += xp[(size_t)(m + 4) * ldx]; s2 += xp[(size_t)(m + 8) * ldx]; s3 += xp[(size_t)(m + 12) * ldx];
        }
        for (; m < m1; m += 4) s0 += xp[(size_t)m * ldx];
    }
    red[rl][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && n < N) part[(size_t)blockIdx.y * N + n] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// few rows, very many columns (the processed-memory gradient: 85 step tapes of B x L x A floats): one thread per four columns walks
// all rows with 16-byte loads -- no partials, no second launch.  Per column: four interleaved sums over the rows ascending.
__global__ __launch_bounds__(256) void colsum_wide_kernel(const float* X, int ldx, int M, int N4, float* out, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N4) return;
    const float* xp = X + (size_t)i * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int m = 0;
    for (; m + 4 <= M; m += 4) {
        s0 = s0 + st_ld4(xp + (size_t)m * ldx); s1 = s1 + st_ld4(xp + (size_t)(m + 1) * ldx);
        s2 = s2 + st_ld4(xp + (size_t)(m + 2) * ldx); s3 = s3 + st_ld4(xp + (size_t)(m + 3) * ldx);
    }
    for (; m < M; ++m) s0 = s0 + st_ld4(xp + (size_t)m * ldx);
    f32x4 r = (s0 + s1) + (s2 + s3);
    f32x4* o = reinterpret_cast<f32x4*>(out) + i;
    if (accumulate) r = r + *o;
    *o = r;
}

__device__ __forceinline__ float act_grad(float out, int act) {
    switch (act) {
        case ST_ACT_RELU: return out > 0.0f ? 1.0f : 0.0f;
        case ST_ACT_TANH: return 1.0f - out * out;
        case ST_ACT_SIGMOID: return out * (1.0f - out);
        default: return 1.0f;
    }
}

// dpre = dout * mask * act'(out)      (2-D strided views, M x N)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dout, int ldd, const float* out, int ldo, int act,
                                                      const float* mask, int ldm, float* dpre, int ldp, int M, int N) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / N;
        const int n = (int)(i - m * N);
        float g = dout[m * ldd + n];
        if (mask) g *= mask[m * ldm + n];
        if (act != ST_ACT_NONE) g *= act_grad(out[m * ldo + n], act);
        dpre[m * ldp + n] = g;
    }
}

// BatchNorm (batch statistics) backward.  y = act((x - mean) / sqrt(var + eps) * w + b).
//   pass 1 (bn_bwd_reduce): s1[n] = sum dyb, s2[n] = sum dyb * xhat     with dyb = dy * act'(y)
//   pass 2 (bn_bwd_apply):  dx = w / sigma * (dyb - s1/M - xhat * s2/M)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff,
                                                            int act, const float* x, int ldx, int xoff, const float* mean,
                                                            const float* var, float eps, int M, int N, int rows_per_chunk,
                                                            float* part, const float* mask = nullptr, int ldm = 0) {   // part[chunk][2][N]
    __shared__ float r1[4][64], r2[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int nc = min(n, N - 1);
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    const float mu = mean[nc], inv = 1.0f / sqrtf(var[nc] + eps);
    const float* __restrict__ dp = dy + doff + nc;
    const float* __restrict__ xp = x + xoff + nc;
    const float* __restrict__ yp = act != ST_ACT_NONE ? y + yoff + nc : nullptr;
    const float* __restrict__ kp = mask ? mask + nc : nullptr;        // dropout mask behind the activation: dyb = dy * mask * act'(y)
    // two rows per pass with independent sums: (up to) six loads in flight per thread instead of a chain of dependent round trips
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    int m = m0 + rl;
    for (; m + 4 < m1; m += 8) {
        float g0 = dp[(size_t)m * ldd], g1 = dp[(size_t)(m + 4) * ldd];
        const float x0 = xp[(size_t)m * ldx], x1 = xp[(size_t)(m + 4) * ldx];
        if (kp) { g0 *= kp[(size_t)m * ldm]; g1 *= kp[(size_t)(m + 4) * ldm]; }
        if (yp) { g0 *= act_grad(yp[(size_t)m * ldy], act); g1 *= act_grad(yp[(size_t)(m + 4) * ldy], act); }
        a0 += g0; a1 += g1;
        b0 = fmaf(g0, (x0 - mu) * inv, b0); b1 = fmaf(g1, (x1 - mu) * inv, b1);
    }
    for (; m < m1; m += 4) {
        float g = dp[(size_t)m * ldd];
        if (kp) g *= kp[(size_t)m * ldm];
        if (yp) g *= act_grad(yp[(size_t)m * ldy], act);
        a0 += g;
        b0 = fmaf(g, (xp[(size_t)m * ldx] - mu) * inv, b0);
    }
    r1[rl][c] = a0 + a1; r2[rl][c] = b0 + b1;
    __syncthreads();
    if (rl == 0 && n < N) {
        part[((size_t)blockIdx.y * 2 + 0) * N + n] = r1[0][c] + r1[1][c] + r1[2][c] + r1[3][c];
        part[((size_t)blockIdx.y * 2 + 1) * N + n] = r2[0][c] + r2[1][c] + r2[2][c] + r2[3][c];
    }
}

// out[q][n] (+)= sum_chunks part[chunk][q][n], q < Q.  16 (q, n) pairs x 16 chunk lanes per workgroup: lane j adds chunks j, j + 16, ...
// (at most 8, all loads issued at once), the lanes' sums are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void chunk_final_kernel(const float* part, int chunks, int Q, int N, float* out0, float* out1,
                                                          int accumulate) {
    constexpr int CL = 16, PER = 8;
    __shared__ float red[CL][16];
    const int c = threadIdx.x & 15, cl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + c;
    const bool ok = i < Q * N;
    const int ic = min(i, Q * N - 1);
    const int q = ic / N, n = ic - q * N;
    const float* p = part + (size_t)q * N + n;
    const size_t cs = (size_t)Q * N;
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) { const int ch = cl + j * CL; v[j] = p[(size_t)(ch < chunks ? ch : 0) * cs] * (ch < chunks ? 1.0f : 0.0f); }   // (a factor, not a select: no branch + wait per load)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) s += v[j];
    red[cl][c] = s;
    __syncthreads();
    if (cl != 0 || !ok) return;
    float* out = q == 0 ? out0 : out1;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < CL; ++k) t += red[k][c];
    out[n] = accumulate ? out[n] + t : t;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff,
                                                           int act, const float* x, int ldx, int xoff, const float* mean,
                                                           const float* var, const float* w, float eps, int M, int N,
                                                           const float* s1, const float* s2, float* dx, int lddx, int dxoff,
                                                           int Mstat, const float* inv_total, const float* mask = nullptr, int ldm = 0,
                                                           float* dres = nullptr, int lddr = 0) {
    const size_t total = (size_t)M * N;
    // rows the statistics (and s1, s2) were taken over: > M under SyncBN, where 1 / (global row count) comes as a device scalar
    const float invM = inv_total ? inv_total[0] : 1.0f / (float)Mstat;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / N;
        const int n = (int)(i - m * N);
        float g = dy[m * ldd + doff + n];
        if (mask) g *= mask[m * ldm + n];
        if (dres) dres[m * lddr + n] = g;          // the gradient of a residual input added behind the activation, in front of the mask
        if (act != ST_ACT_NONE) g *= act_grad(y[m * ldy + yoff + n], act);
        const float inv = 1.0f / sqrtf(var[n] + eps);
        const float xh = (x[m * ldx + xoff + n] - mean[n]) * inv;
        dx[m * lddx + dxoff + n] = (w ? w[n] : 1.0f) * inv * (g - s1[n] * invM - xh * s2[n] * invM);
    }
}

// Highway y = H*T + x*(1-T):  dH = dy*T, dT = dy*(H - x), dx_direct = dy*(1-T)      ref: src/module.py:551-554
// (the activation derivatives of H = relu(.) and T = sigmoid(.) are applied by the producers' backward)
__global__ __launch_bounds__(256) void highway_bwd_kernel(const float* dy, const float* H, const float* x, const float* T,
                                                          float* dH, float* dT, float* dxd, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float g = dy[i], t = T[i];
        dH[i] = g * t;
        dT[i] = g * (H[i] - x[i]);
        dxd[i] = g * (1.0f - t);
    }
}

// x'[b][t][:] = max(x[b][t-1][:], x[b][t][:]), t = 0 keeps x[b][0]: MaxPool1d(2, stride 1, padding 1)[:T] of the CBHG (src/module.py:600)
// as a tensor of its own -- the input of the LDS-DMA conv kernel, which cannot take a maximum on the way into LDS (21 MB each way at
// C2: ~10 us; the register-staged kernel with the pool fused into its loads took 84 us for the 640 -> 128 conv, the DMA kernel 52)
__global__ __launch_bounds__(256) void pool_prev_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int C4, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / C4;
        const int t = (int)(row % T);
        f32x4 v = st_ld4(x + 4 * i);
        if (t > 0) {
            const f32x4 q = st_ld4(x + 4 * (i - C4));
            v = f32x4{fmaxf(v[0], q[0]), fmaxf(v[1], q[1]), fmaxf(v[2], q[2]), fmaxf(v[3], q[3])};
        }
        *reinterpret_cast<f32x4*>(y + 4 * i) = v;
    }
}

// backward of x'[t] = max(x[t-1], x[t]) within an utterance (t = 0 keeps x[0]).  torch's MaxPool1d scans the
// window left to right and keeps the FIRST maximum, so on a tie x[t-1] receives the gradient of output t.
__global__ __launch_bounds__(256) void pool_prev_bwd_kernel(const float* dyp, const float* x, float* dx, int Bn, int T, int C) {
    const size_t total = (size_t)Bn * T * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t row = i / C;
        const int t = (int)(row % T);
        // all five operands requested at once from clamped addresses; the conditions select afterwards (as `if (cond) g += dyp[..]` every
        // load sat behind a branch and a full wait)
        const bool has_prev = t > 0, has_next = t + 1 < T;
        const float xv = x[i], xp = x[has_prev ? i - C : i], xn = x[has_next ? i + C : i];
        const float g0 = dyp[i], g1 = dyp[has_next ? i + C : i];
        float g = 0.0f;
        // contribution of output t: x[t] is selected when t == 0 or x[t] > x[t-1]
        g += ((!has_prev || xv > xp) ? 1.0f : 0.0f) * g0;          // (a factor: a select on the loaded value would pull the load behind the compare)
        // contribution of output t+1: x[t] is selected when x[t] >= x[t+1]
        g += ((has_next && xv >= xn) ? 1.0f : 0.0f) * g1;
        dx[i] = g;
        (void)c;
    }
}

// dst(b, t, c) (+)= src(b, t, c) over arbitrary (b, t) strides, contiguous c
__global__ __launch_bounds__(256) void copy3d_kernel(float* dst, long dsb, long dst_, const float* src, long ssb, long sst,
                                                     int Bn, int T, int C, int accumulate) {
    const size_t total = (size_t)Bn * T * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t row = i / C;
        const int t = (int)(row % T);
        const size_t b = row / T;
        const float v = src[b * ssb + (size_t)t * sst + c];
        float* p = dst + b * dsb + (size_t)t * dst_ + c;
        *p = accumulate ? *p + v : v;
    }
}

// dtable(v, :) = sum over the rows r with idx(r) == v of dout(r, :), added in row order: one workgroup column per table row
// scans the index vector (a few thousand entries, read through the scalar cache) -- deterministic, no atomics
// 64 columns x 8 row lanes per workgroup: lane q scans its contiguous eighth of the index vector in row order, the eight partial sums
// are added in lane order -- a fixed order whatever the data (one thread per column scanning all n rows took 56 us)
__global__ __launch_bounds__(512) void scatter_add_rows_kernel(const float* dout, const int64_t* idx, float* dtable,
                                                               int n, int D, int V) {
    __shared__ float red[8][64];
    const int v = blockIdx.x;
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int d = blockIdx.y * 64 + c;
    const int per = (n + 7) / 8, r0 = q * per, r1 = min(n, r0 + per);
    float acc = 0.0f;
    if (d < D) {
        // every row is loaded (a wave reads one 256-byte run) and kept or dropped by a 0 / 1 factor: `if (idx == v) acc += dout[..]` is a
        // branch around the load with a full wait behind it, `per` dependent round trips (finite gradients assumed: 0 * x = 0)
#pragma unroll 8
        for (int r = r0; r < r1; ++r) acc = fmaf(idx[r] == (int64_t)v ? 1.0f : 0.0f, dout[(size_t)r * D + d], acc);
    }
    red[q][c] = acc;
    __syncthreads();
    if (q != 0 || d >= D) return;
    float s = red[0][c];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][c];
    dtable[(size_t)v * D + d] += s;
}

// out-of-place BatchNorm normalisation: Y = act((X - mean) / sqrt(var + eps) * w + b); X is kept for the backward
__global__ __launch_bounds__(256) void bn_norm_fwd_kernel(const float* X, int ldx, int xoff, float* Y, int ldy, int yoff,
                                                          int M, int N, const float* mean, const float* var, const float* w,
                                                          const float* b, float eps, int act) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / N;
        const int n = (int)(i - m * N);
        float v = (X[m * ldx + xoff + n] - mean[n]) / sqrtf(var[n] + eps);
        v = v * (w ? w[n] : 1.0f) + (b ? b[n] : 0.0f);
        Y[m * ldy + yoff + n] = st_act(v, act);
    }
}

// ConvLayer's tail in training (src/module.py:641-646): T = act(BatchNorm(X)) kept for the backward, Y = (T + res) * mask the layer's output
// (res, mask optional); 16-byte pieces (N % 4 == 0, all rows 16-byte aligned)
__global__ __launch_bounds__(256) void bn_norm_res_mask_fwd_kernel(const float* __restrict__ X, float* __restrict__ Tact, float* __restrict__ Y,
                                                                   int N4, size_t total4, const float* __restrict__ mean,
                                                                   const float* __restrict__ var, const float* __restrict__ w,
                                                                   const float* __restrict__ b, float eps, int act,
                                                                   const float* __restrict__ res, const float* __restrict__ mask) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int n = 4 * (int)(i % N4);
        const f32x4 x = st_ld4(X + 4 * i), mu = st_ld4(mean + n), vr = st_ld4(var + n);
        f32x4 g = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f}, r = {0.f, 0.f, 0.f, 0.f}, k = {1.f, 1.f, 1.f, 1.f};
        if (w) g = st_ld4(w + n);
        if (b) be = st_ld4(b + n);
        if (res) r = st_ld4(res + 4 * i);
        if (mask) k = st_ld4(mask + 4 * i);
        f32x4 t, y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = (x[j] - mu[j]) / sqrtf(vr[j] + eps);           // (bn_norm_fwd_kernel's expression, element for element)
            v = v * g[j] + be[j];
            t[j] = st_act(v, act);
            y[j] = (t[j] + r[j]) * k[j];
        }
        if (Tact) *reinterpret_cast<f32x4*>(Tact + 4 * i) = t;
        *reinterpret_cast<f32x4*>(Y + 4 * i) = y;
    }
}

// Highway combine (training forward keeps H and T): y = H * T + x * (1 - T)      ref: src/module.py:554
__global__ __launch_bounds__(256) void highway_fwd_kernel(const float* H, const float* T, const float* x, float* y, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float t = T[i];
        y[i] = st_highway(H[i], t, x[i]);
    }
}

// ---- the K BatchNorm1d layers of a conv bank, one launch per phase (see st_bn_bank_fwd) ----------------------------------------------------
// The per-layer kernels above with a segment index in the grid: same chunking of the rows per segment, same merge order, so a bank of one
// segment gives what st_bn_stats / st_bn_norm_fwd / st_bn_bwd give.  part: [seg][chunk][2][N].
struct BnBank { st_bn_bank_seg s[ST_BN_BANK_MAX]; int nseg, Bn, N; };

__global__ __launch_bounds__(256) void bnb_stats_chunk_kernel(const BnBank a, float* part, int max_chunks) {
    __shared__ float red[4][64];
    __shared__ float smean[64];
    const st_bn_bank_seg& sg = a.s[blockIdx.z];
    const int N = a.N, M = a.Bn * sg.T;
    const int chunks = st_colreduce_chunks(M), rpc = (M + chunks - 1) / chunks;
    if ((int)blockIdx.y >= chunks) return;
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const bool ok = n < N;
    const int m0 = blockIdx.y * rpc, m1 = min(M, m0 + rpc);
    const int ldx = sg.ldx;
    const float* __restrict__ xp = sg.x + min(n, N - 1);
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
    for (; m + 12 < m1; m += 16) {
        const float d0 = xp[(size_t)m * ldx] - mean, d1 = xp[(size_t)(m + 4) * ldx] - mean;
        const float d2 = xp[(size_t)(m + 8) * ldx] - mean, d3 = xp[(size_t)(m + 12) * ldx] - mean;
        q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2); q3 = fmaf(d3, d3, q3);
    }
    for (; m < m1; m += 4) { const float d = xp[(size_t)m * ldx] - mean; q0 = fmaf(d, d, q0); }
    __syncthreads();
    red[rl][c] = (q0 + q1) + (q2 + q3);
    __syncthreads();
    if (rl == 0 && ok) {
        float* pp = part + ((size_t)blockIdx.z * max_chunks + blockIdx.y) * 2 * N;
        pp[n] = mean;
        pp[N + n] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}

// rec != NULL (SyncBN, local half): (mean, M2, row count) of segment k go to rec[k (2N + 1) ..]; nothing else is written but batches_tracked
__global__ __launch_bounds__(256) void bnb_stats_final_kernel(const BnBank a, const float* part, int max_chunks, float* rec) {
    constexpr int CL = 16, PER = 8;
    __shared__ float red[CL][16];
    __shared__ float smean[16];
    const st_bn_bank_seg& sg = a.s[blockIdx.y];
    const int N = a.N, M = a.Bn * sg.T;
    const int chunks = st_colreduce_chunks(M), rpc = (M + chunks - 1) / chunks;
    const int c = threadIdx.x & 15, cl = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + c;
    if (blockIdx.x == 0 && threadIdx.x == 0 && sg.batches_tracked) sg.batches_tracked[0] += 1;
    const bool ok = n < N;
    const float* pm = part + (size_t)blockIdx.y * max_chunks * 2 * N + min(n, N - 1);
    const size_t cs = (size_t)2 * N;
    float mu[PER], m2[PER], cnt[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int ch = cl + j * CL;
        const bool in = ch < chunks;
        const int chc = in ? ch : 0;
        mu[j] = pm[(size_t)chc * cs]; m2[j] = pm[(size_t)chc * cs + N] * (in ? 1.0f : 0.0f);
        cnt[j] = in ? (float)max(min(M, (ch + 1) * rpc) - ch * rpc, 0) : 0.0f;
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) acc = fmaf(cnt[j], mu[j], acc);
    red[cl][c] = acc;
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
    if (rec) {
        float* r = rec + (size_t)blockIdx.y * (2 * N + 1);
        r[n] = mean; r[N + n] = m2t;
        if (n == 0) r[2 * N] = (float)M;
        return;
    }
    sg.mean[n] = mean;
    sg.var[n] = var_b;
    if (sg.run_mean) {
        const float var_u = M > 1 ? m2t / (float)(M - 1) : var_b;
        sg.run_mean[n] = (1.0f - sg.momentum) * sg.run_mean[n] + sg.momentum * mean;
        sg.run_var[n] = (1.0f - sg.momentum) * sg.run_var[n] + sg.momentum * var_u;
    }
}

// SyncBN, merged half for all segments: allrec (world, nseg, 2N + 1) gathered records -> mean / var of the global batch of every segment
// (+ running statistics) and 1 / (global row count) per segment.  The per-layer bn_sync_merge_kernel with a segment index.
__global__ __launch_bounds__(256) void bnb_sync_merge_kernel(const BnBank a, const float* allrec, int world, float* inv_total_out) {
    const int k = blockIdx.y;
    const st_bn_bank_seg& sg = a.s[k];
    const int N = a.N;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rs = (size_t)2 * N + 1, ws_ = rs * a.nseg;
    const float* rec = allrec + (size_t)k * rs;
    float total = 0.0f;
    for (int r = 0; r < world; ++r) total += rec[r * ws_ + 2 * N];
    if (n == 0) inv_total_out[k] = 1.0f / total;
    if (n >= N) return;
    float acc = 0.0f;
    for (int r = 0; r < world; ++r) acc = fmaf(rec[r * ws_ + 2 * N], rec[r * ws_ + n], acc);
    const float mean = acc / total;
    float m2 = 0.0f;
    for (int r = 0; r < world; ++r) { const float d = rec[r * ws_ + n] - mean; m2 += rec[r * ws_ + N + n] + rec[r * ws_ + 2 * N] * d * d; }
    sg.mean[n] = mean;
    sg.var[n] = m2 / total;
    if (sg.run_mean) {
        sg.run_mean[n] = (1.0f - sg.momentum) * sg.run_mean[n] + sg.momentum * mean;
        sg.run_var[n] = (1.0f - sg.momentum) * sg.run_var[n] + sg.momentum * (m2 / fmaxf(total - 1.0f, 1.0f));
    }
}

// Y[(b Tout + t)][k N + n] = BN_k(x_k[b T_k + t][n]), t < Tout.  blockIdx.y = segment (its fields stay in scalar registers); four
// channels per thread when N % 4 == 0 and the rows are 16-byte addressable
__global__ __launch_bounds__(256) void bnb_norm_kernel(const BnBank a, float* __restrict__ Y, int ldy, int Tout, int vec) {
    const int k = blockIdx.y;
    const st_bn_bank_seg& sg = a.s[k];
    const int N = a.N, T = sg.T, ldx = sg.ldx;
    const float* __restrict__ x = sg.x;
    if (vec) {
        const int N4 = N >> 2;
        const size_t total = (size_t)a.Bn * Tout * N4;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const size_t r = i / N4;
            const int n = (int)(i - r * N4) * 4;
            const int b = (int)(r / Tout), t = (int)(r - (size_t)b * Tout);
            const f32x4 xv = st_ld4(x + ((size_t)b * T + t) * ldx + n);
            const f32x4 mu = st_ld4(sg.mean + n), va = st_ld4(sg.var + n);
            const f32x4 w4 = sg.w ? st_ld4(sg.w + n) : f32x4{1.f, 1.f, 1.f, 1.f}, b4 = sg.b ? st_ld4(sg.b + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (xv[c] - mu[c]) / sqrtf(va[c] + sg.eps) * w4[c] + b4[c];
            *reinterpret_cast<f32x4*>(Y + r * ldy + (size_t)k * N + n) = o;
        }
        return;
    }
    const size_t total = (size_t)a.Bn * Tout * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / N;
        const int n = (int)(i - r * N);
        const int b = (int)(r / Tout), t = (int)(r - (size_t)b * Tout);
        float v = (x[((size_t)b * T + t) * ldx + n] - sg.mean[n]) / sqrtf(sg.var[n] + sg.eps);
        v = v * (sg.w ? sg.w[n] : 1.0f) + (sg.b ? sg.b[n] : 0.0f);
        Y[r * ldy + (size_t)k * N + n] = v;
    }
}

__global__ __launch_bounds__(256) void bnb_bwd_reduce_kernel(const BnBank a, const float* __restrict__ dY, int lddy, int Tout, float* part, int max_chunks) {
    __shared__ float r1[4][64], r2[4][64];
    const int k = blockIdx.z;
    const st_bn_bank_seg& sg = a.s[k];
    const int N = a.N, T = sg.T, M = a.Bn * T;
    const int chunks = st_colreduce_chunks(M), rpc = (M + chunks - 1) / chunks;
    if ((int)blockIdx.y >= chunks) return;
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int nc = min(n, N - 1);
    const int m0 = blockIdx.y * rpc, m1 = min(M, m0 + rpc);
    const float mu = sg.mean[nc], inv = 1.0f / sqrtf(sg.var[nc] + sg.eps);
    const float* __restrict__ dp = dY + (size_t)k * N + nc;
    const float* __restrict__ xp = sg.x + nc;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    // rows whose frame lies past the bank (t >= Tout: the trimmed position of an even-k conv) have dy = 0 and add nothing
    auto dyrow = [&](int m) -> float {
        const int b = m / T, t = m - b * T;
        const float v = dp[((size_t)b * Tout + min(t, Tout - 1)) * lddy];
        return v * (t < Tout ? 1.0f : 0.0f);
    };
    int m = m0 + rl;
    for (; m + 4 < m1; m += 8) {
        const float g0 = dyrow(m), g1 = dyrow(m + 4);
        const float x0 = xp[(size_t)m * sg.ldx], x1 = xp[(size_t)(m + 4) * sg.ldx];
        a0 += g0; a1 += g1;
        b0 = fmaf(g0, (x0 - mu) * inv, b0); b1 = fmaf(g1, (x1 - mu) * inv, b1);
    }
    for (; m < m1; m += 4) {
        const float g = dyrow(m);
        a0 += g;
        b0 = fmaf(g, (xp[(size_t)m * sg.ldx] - mu) * inv, b0);
    }
    r1[rl][c] = a0 + a1; r2[rl][c] = b0 + b1;
    __syncthreads();
    if (rl == 0 && n < N) {
        float* pp = part + ((size_t)k * max_chunks + blockIdx.y) * 2 * N;
        pp[n] = r1[0][c] + r1[1][c] + r1[2][c] + r1[3][c];
        pp[N + n] = r2[0][c] + r2[1][c] + r2[2][c] + r2[3][c];
    }
}

// sums_k[q][n] = sum over the segment's chunks of part[k][chunk][q][n]   (chunk_final_kernel's merge order)
__global__ __launch_bounds__(256) void bnb_bwd_final_kernel(const BnBank a, const float* part, int max_chunks) {
    constexpr int CL = 16, PER = 8;
    __shared__ float red[CL][16];
    const st_bn_bank_seg& sg = a.s[blockIdx.y];
    const int N = a.N, M = a.Bn * sg.T;
    const int chunks = st_colreduce_chunks(M);
    const int c = threadIdx.x & 15, cl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + c;
    const bool ok = i < 2 * N;
    const int ic = min(i, 2 * N - 1);
    const float* p = part + (size_t)blockIdx.y * max_chunks * 2 * N + ic;
    const size_t cs = (size_t)2 * N;
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) { const int ch = cl + j * CL; v[j] = p[(size_t)(ch < chunks ? ch : 0) * cs] * (ch < chunks ? 1.0f : 0.0f); }
    float sacc = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) sacc += v[j];
    red[cl][c] = sacc;
    __syncthreads();
    if (cl != 0 || !ok) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < CL; ++k) t += red[k][c];
    sg.sums[ic] = t;
}

__global__ __launch_bounds__(256) void bnb_bwd_apply_kernel(const BnBank a, const float* __restrict__ dY, int lddy, int Tout, int relu_in, int vec,
                                                            const float* __restrict__ inv_total) {
    const int k = blockIdx.y;
    const st_bn_bank_seg& sg = a.s[k];
    const int N = a.N, T = sg.T, M = a.Bn * T;
    // SyncBN: the sums were taken over every rank's rows; 1 / (global row count) of the segment comes as a device scalar
    const float invM = inv_total ? inv_total[k] : 1.0f / (float)M;
    if (vec) {      // four channels per thread (N % 4 == 0, 16-byte addressable rows): one row / frame split per four elements
        const int N4 = N >> 2;
        const size_t total = (size_t)M * N4;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const int m = (int)(i / N4), n = (int)(i - (size_t)m * N4) * 4;
            const int b = m / T, t = m - b * T;
            const f32x4 g4 = st_ld4(dY + ((size_t)b * Tout + min(t, Tout - 1)) * lddy + (size_t)k * N + n);
            const f32x4 xv = st_ld4(sg.x + (size_t)m * sg.ldx + n);
            const f32x4 mu = st_ld4(sg.mean + n), va = st_ld4(sg.var + n), s1 = st_ld4(sg.sums + n), s2 = st_ld4(sg.sums + N + n);
            const f32x4 w4 = sg.w ? st_ld4(sg.w + n) : f32x4{1.f, 1.f, 1.f, 1.f};
            const float on = t < Tout ? 1.0f : 0.0f;
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float g = g4[c] * on;
                const float inv = 1.0f / sqrtf(va[c] + sg.eps);
                const float xh = (xv[c] - mu[c]) * inv;
                float d = w4[c] * inv * (g - s1[c] * invM - xh * s2[c] * invM);
                if (relu_in) d *= act_grad(xv[c], ST_ACT_RELU);
                o[c] = d;
            }
            *reinterpret_cast<f32x4*>(sg.dx + (size_t)m * sg.lddx + n) = o;
        }
        return;
    }
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
        const int b = m / T, t = m - b * T;
        const float g = dY[((size_t)b * Tout + min(t, Tout - 1)) * lddy + (size_t)k * N + n] * (t < Tout ? 1.0f : 0.0f);
        const float inv = 1.0f / sqrtf(sg.var[n] + sg.eps);
        const float xv = sg.x[(size_t)m * sg.ldx + n];
        const float xh = (xv - sg.mean[n]) * inv;
        float d = (sg.w ? sg.w[n] : 1.0f) * inv * (g - sg.sums[n] * invM - xh * sg.sums[N + n] * invM);
        if (relu_in) d *= act_grad(xv, ST_ACT_RELU);
        sg.dx[(size_t)m * sg.lddx + n] = d;
    }
}

// One Highway layer on the pre-activations of its two Linear layers side by side, ht (M, 2C) = [h | t]   (see st_highway_ht_fwd)
__global__ __launch_bounds__(256) void highway_ht_fwd_kernel(const float* __restrict__ ht, const float* __restrict__ x, float* __restrict__ y,
                                                             int C, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / C;
        const int c = (int)(i - m * C);
        const float h = st_act(ht[m * 2 * C + c], ST_ACT_RELU), t = st_act(ht[m * 2 * C + C + c], ST_ACT_SIGMOID);
        y[i] = st_highway(h, t, x[i]);
    }
}
__global__ __launch_bounds__(256) void highway_ht_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ht, const float* __restrict__ x,
                                                             float* __restrict__ dht, float* __restrict__ dxd, int C, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / C;
        const int c = (int)(i - m * C);
        const float h = st_act(ht[m * 2 * C + c], ST_ACT_RELU), t = st_act(ht[m * 2 * C + C + c], ST_ACT_SIGMOID);
        const float g = dy[i];
        dht[m * 2 * C + c] = (g * t) * act_grad(h, ST_ACT_RELU);                      // dH = dy T, through relu
        dht[m * 2 * C + C + c] = (g * (h - x[i])) * act_grad(t, ST_ACT_SIGMOID);      // dT = dy (H - x), through sigmoid
        dxd[i] = g * (1.0f - t);
    }
}

// Re-layouts of MANY parameters in one launch (training re-lays every conv / linear weight out once per step: one launch instead of
// one torch copy per weight and layout).  Workgroup b serves the descriptor d with blk0[d] <= b < blk0[d] + nblk[d]; a thread
// writes four consecutive destination floats (one 16-byte store when aligned) gathered from the source.
//   mode 0: (N, Cin, KT) -> (N, KT, Cin)                      tap-major conv weight (forward GEMM operand)
//   mode 1: (N, Cin, KT) -> (Cin, KT, N), taps reversed        weight of the input-gradient conv, tap-major (KT = 1: plain transpose)
__global__ __launch_bounds__(256) void relayout_batch_kernel(const st_relayout_desc* __restrict__ table, int n, const int* __restrict__ blk_desc) {
    int lo = 0, hi = n - 1;
    if (blk_desc) lo = blk_desc[blockIdx.x];      // (one load; the search below is log2(n) DEPENDENT round trips in front of every workgroup)
    else while (lo < hi) {                 // last descriptor whose first workgroup is <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const st_relayout_desc d = table[lo];
    const size_t total = (size_t)d.N * d.Cin * d.KT;
    const size_t i0 = ((size_t)((int)blockIdx.x - d.blk0) * 256 + threadIdx.x) * 4;
    if (i0 >= total) return;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // (elements past the end re-read the last one and are never stored: four independent loads instead of four behind a test each)
        const size_t i = i0 + c < total ? i0 + c : total - 1;
        if (d.mode == 0) {
            const int ci = (int)(i % d.Cin);
            const size_t r = i / d.Cin;
            const int kt = (int)(r % d.KT), nn = (int)(r / d.KT);
            v[c] = d.src[((size_t)nn * d.Cin + ci) * d.KT + kt];
        } else {
            const int nn = (int)(i % d.N);
            const size_t r = i / d.N;
            const int kt = (int)(r % d.KT), ci = (int)(r / d.KT);
            v[c] = d.src[((size_t)nn * d.Cin + ci) * d.KT + (d.KT - 1 - kt)];
        }
    }
    if (d.mode == 1 && d.ld_dst > d.N) {      // a column block of a wider matrix: row r = i / N of the layout starts at dst + r * ld_dst
        for (int c = 0; c < 4 && i0 + c < total; ++c) { const size_t i = i0 + c; d.dst[(i / d.N) * d.ld_dst + (i % d.N)] = v[c]; }
        return;
    }
    if (i0 + 4 <= total && st_aligned16(d.dst + i0)) *reinterpret_cast<f32x4*>(d.dst + i0) = f32x4{v[0], v[1], v[2], v[3]};
    else
        for (int c = 0; c < 4 && i0 + c < total; ++c) d.dst[i0 + c] = v[c];
}

inline int blocks_for(size_t n, int cap = 4096) {
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > (size_t)cap ? cap : b);
}

}  // namespace

static inline bool tn_fold(int Cin, int KT, int pool_prev) { return Cin < 16 && KT > 1 && !pool_prev; }

// 128 x 128 tiles only where they still give every compute unit a workgroup (>= 256 tiles: the two decoder-LSTM gradients, 4096 x 2560
// and 4096 x 1792); below that the 64 x 64 form wins through occupancy (re-measured with the 16-row-chunk 128 form, three workgroups per
// unit: the encoder conv's gradient 81.8 vs 59.3 us, the final Linear's 68.4 vs 52.5; training step 8.87 / 9.02+ vs 8.80 ms at thresholds 64 / 16).  Measured per training step (all weight-gradient products):
// 128-tiles wherever both dimensions reach 128: 2270 us; from 64 tiles on 2153; from 256 on 2112; from 512 on 2148
static inline int tn_tile(int Cin, int N, int KT) {
    if (!(N >= 128 && Cin >= 128) || tn_fold(Cin, KT, 0)) return TN_T;
    const long tiles128 = (long)((N + 127) / 128) * ((Cin + 127) / 128) * KT;
    return tiles128 >= 256 ? 128 : TN_T;
}

// shapes convw_small_kernel takes (next to: no bias gradient wanted, 16-byte addressable dC rows)
static inline bool cs_shape(int Cin, int N, int KT) { return Cin <= 4 && KT > 1 && N <= 64 && N % 4 == 0 && Cin * KT <= 64; }

extern "C" size_t st_gemm_wgrad_workspace_floats(int Bn, int Tout, int Cin, int N, int KT) {
    const int M = Bn * Tout;
    if (cs_shape(Cin, N, KT)) {
        // one slab per workgroup, several workgroups per compute unit (a unit = 64 rows of one sequence is a memory round trip plus a
        // few hundred cycles of MFMAs: occupancy hides the round trips); the many slabs are cheap to add (sum_partials_tall_kernel)
        const long units = (long)Bn * ((Tout + 63) / 64);
        long Z = units / 2;
        if (Z > 1024) Z = 1024;
        if (Z < 2) Z = 2;
        return (size_t)Z * N * Cin * KT + (size_t)Z * N;
    }
    const int TM = tn_tile(Cin, N, KT);
    const int colblocks = tn_fold(Cin, KT, 0) ? (Cin * KT + TM - 1) / TM : ((Cin + TM - 1) / TM) * KT;
    const int tiles = ((N + TM - 1) / TM) * colblocks;
    int Z = (512 + tiles - 1) / tiles;
    // more tiles than compute units and a badly filled last round (640 tiles = 2.5 rounds of 256: the decoder LSTM's 4096 x 2560
    // gradient): two slabs when they fill the rounds exactly -- the half-length products win more than the extra slab costs
    if (Z == 1 && tiles > 256) {
        const int r1 = (tiles + 255) / 256, r2 = (2 * tiles + 255) / 256;
        if (tiles * 100 < r1 * 256 * 85 && 2 * tiles * 100 >= r2 * 256 * 98) Z = 2;
    }
    const int maxz = (M + 255) / 256;
    if (Z > maxz) Z = maxz;
    // the slabs are added by one thread per output element: at most 96 of them -- 256 for a small matrix reduced over very many rows (the
    // location conv / W_l gradients: 118k rows of the attention tapes into 2k / 8k elements, where 96 workgroups walk 77 chunks each)
    const int zcap = (size_t)N * Cin * KT <= 16384 ? 256 : 96;
    if (Z > zcap) Z = zcap;
    if (Z < 1) Z = 1;
    return (size_t)Z * N * Cin * KT + (size_t)Z * N;      // (+ the bias-gradient slabs of st_gemm_wgrad_db)
}

static int tn_impl(const float* dC, int lddc, int dcoff, const float* A, int lda, float* dW, float* db, float* ws,
                   int Bn, int Tin, int Tout, int Cin, int N, int KT, int pad, int pool_prev, int accumulate,
                   void* stream, float* dW1 = nullptr, int split = 0, float* db_dup = nullptr) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dC && A && dW && ws && Bn > 0 && Tin > 0 && Tout > 0 && Cin > 0 && N > 0 && KT > 0, "st_gemm_wgrad: bad arguments");
    TnArgs g;
    memset(&g, 0, sizeof(g));
    g.dC = dC; g.lddc = lddc; g.dcoff = dcoff; g.A = A; g.lda = lda; g.part = ws;
    g.Bn = Bn; g.Tin = Tin; g.Tout = Tout; g.Cin = Cin; g.N = N; g.KT = KT; g.pad = pad; g.pool_prev = pool_prev;
    g.M = Bn * Tout;
    const size_t per = (size_t)N * Cin * KT;
    const int Z = (int)(st_gemm_wgrad_workspace_floats(Bn, Tout, Cin, N, KT) / (per + N));
    g.rows_per_z = (((g.M + Z - 1) / Z) + 31) / 32 * 32;          // (whole 32-row chunks: the DMA form's unit)
    // the workspace (hence Z) is sized for the folded layout whenever Cin < 16; a pooled input falls back to the per-tap grid
    g.fold = tn_fold(Cin, KT, pool_prev) ? 1 : 0;
    g.lin = (KT == 1 && pad == 0 && Tin == Tout && !pool_prev) ? 1 : 0;
    g.vecx = st_aligned16(dC) && (lddc % 4 == 0) && (dcoff % 4 == 0);
    g.vecy = st_aligned16(A) && (lda % 4 == 0);
    const int TM = tn_tile(Cin, N, KT);
    dim3 grid((N + TM - 1) / TM, g.fold ? (Cin * KT + TM - 1) / TM : ((Cin + TM - 1) / TM) * KT, Z);
    hipStream_t st = (hipStream_t)stream;
    if (cs_shape(Cin, N, KT) && !db && !dW1 && !pool_prev && g.vecx) {
        // few input channels: the im2col columns come out of LDS (convw_small_kernel), Z workgroups = Z slabs
        CsArgs c;
        c.dC = dC + dcoff; c.lddc = lddc; c.A = A; c.lda = lda; c.part = ws; c.Bn = Bn; c.Tin = Tin; c.Tout = Tout; c.Cin = Cin; c.N = N;
        c.KT = KT; c.pad = pad; c.Z = Z;
        hipLaunchKernelGGL(convw_small_kernel, dim3(Z), dim3(256), 0, st, c);
        ST_LAUNCH_CHECK();
        hipLaunchKernelGGL(sum_partials_tall_kernel, dim3((unsigned)((per + 31) / 32)), dim3(256), 0, st, ws, dW, per, Z, accumulate);
        ST_LAUNCH_CHECK();
        return 0;
    }
    // one slab and nothing to add to: the product goes straight to dW (the decoder LSTMs' weight gradients are 29 and 42 MB --
    // the "sum" of one slab was a 17 us copy)
    const bool direct = Z == 1 && !accumulate && !dW1;      // (a split result always leaves through the slab sum)
    if (direct) g.part = dW;
    if (db) g.db_part = direct ? db : ws + (size_t)Z * per;
    // the LDS-DMA form where every piece is whole (or wholly outside) and 16-byte addressable, no fold, no fused max-pool
    const bool whole = N % 4 == 0 && Cin % 4 == 0;
    const bool dma = !g.fold && !pool_prev && g.vecx && g.vecy && whole && g.rows_per_z % 32 == 0;
    if (dma) {
        const size_t lds = (size_t)3 * 2 * 32 * TM * sizeof(float);       // 48 KB at TM = 64, 96 KB at 128
        // (128 tiles: 16-row chunks, 48 KB -- three workgroups per compute unit; 64 tiles: 32-row chunks, 48 KB: 16-row chunks change nothing)
        if (TM == 128) hipLaunchKernelGGL((tn_dma_rc16_kernel<128>), grid, dim3(TN_THREADS), lds / 2, st, g);
        else hipLaunchKernelGGL((tn_dma_kernel<64>), grid, dim3(TN_THREADS), lds, st, g);
    } else if (TM == 128) hipLaunchKernelGGL((tn_kernel<128>), grid, dim3(TN_THREADS), 0, st, g);
    else hipLaunchKernelGGL((tn_kernel<64>), grid, dim3(TN_THREADS), 0, st, g);
    ST_LAUNCH_CHECK();
    if (direct) return 0;
    if (dW1) hipLaunchKernelGGL(sum_partials2_kernel<true>, dim3(blocks_for(per + (db ? N : 0))), dim3(256), 0, st, ws, dW, per, Z, accumulate,
                                db ? g.db_part : nullptr, db, (size_t)(db ? N : 0), dW1, Cin, split, db ? db_dup : nullptr);
    else if (db) hipLaunchKernelGGL(sum_partials2_kernel<false>, dim3(blocks_for(per + N)), dim3(256), 0, st, ws, dW, per, Z, accumulate, g.db_part, db, (size_t)N,
                                    nullptr, 0, 0, nullptr);
    else if (Z >= 64 && per <= 65536) hipLaunchKernelGGL(sum_partials_tall_kernel, dim3((unsigned)((per + 31) / 32)), dim3(256), 0, st, ws, dW, per, Z, accumulate);
    else hipLaunchKernelGGL(sum_partials_kernel, dim3(blocks_for(per)), dim3(256), 0, st, ws, dW, per, Z, accumulate);
    ST_LAUNCH_CHECK();
    return 0;
}

// Several st_gemm_wgrad[_db] (no pooling, no accumulate): the jobs that take the LDS-DMA form with 64-tiles (the small matrices this is
// for) leave as ONE product launch + ONE slab-sum launch per group of up to TN_MAXJ, the others one after the other.  Every job keeps the
// slab count and the summation order it has on its own: the results are bit for bit those of the separate calls.
static size_t tn_ws4(const st_wgrad_job& j) { return (st_gemm_wgrad_workspace_floats(j.Bn, j.Tout, j.Cin, j.N, j.KT) + 3) / 4 * 4; }

extern "C" size_t st_gemm_wgrad_batch_workspace_floats(const st_wgrad_job* jobs, int n) {
    size_t t = 0;
    for (int i = 0; i < n; ++i) t += tn_ws4(jobs[i]);
    return t;
}

extern "C" int st_gemm_wgrad_batch(const st_wgrad_job* jobs, int n, float* ws, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(jobs && n > 0 && ws, "st_gemm_wgrad_batch: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    TnBatch tb;
    SumBatch sb;
    int m = 0, blocks = 0, sblocks = 0, nsum = 0, gi[TN_MAXJ];
    size_t goff[TN_MAXJ];
    auto reset = [&] { memset(&tb, 0, sizeof(tb)); memset(&sb, 0, sizeof(sb)); m = blocks = sblocks = nsum = 0; };
    auto single = [&](int i, size_t off) {
        const st_wgrad_job& j = jobs[i];
        return tn_impl(j.dC, j.lddc, j.dcoff, j.A, j.lda, j.dW, j.db, ws + off, j.Bn, j.Tin, j.Tout, j.Cin, j.N, j.KT, j.pad, 0, 0, stream);
    };
    auto flush = [&]() -> int {
        if (m == 1) { const int rc = single(gi[0], goff[0]); reset(); return rc; }
        if (m == 0) return 0;
        tb.blk0[m] = blocks; tb.n = m;
        hipLaunchKernelGGL((tn_dma_batch_kernel<64>), dim3(blocks), dim3(TN_THREADS), (size_t)3 * 2 * 32 * 64 * sizeof(float), st, tb);
        ST_LAUNCH_CHECK();
        if (nsum) {
            sb.blk0[nsum] = sblocks; sb.n = nsum;
            hipLaunchKernelGGL(sum_partials_batch_kernel, dim3(sblocks), dim3(256), 0, st, sb);
            ST_LAUNCH_CHECK();
        }
        reset();
        return 0;
    };
    reset();
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        const st_wgrad_job& j = jobs[i];
        ST_CHECK_ARG(j.dC && j.A && j.dW && j.Bn > 0 && j.Tin > 0 && j.Tout > 0 && j.Cin > 0 && j.N > 0 && j.KT > 0, "st_gemm_wgrad_batch: bad job %d", i);
        TnArgs g;
        memset(&g, 0, sizeof(g));
        g.dC = j.dC; g.lddc = j.lddc; g.dcoff = j.dcoff; g.A = j.A; g.lda = j.lda;
        g.Bn = j.Bn; g.Tin = j.Tin; g.Tout = j.Tout; g.Cin = j.Cin; g.N = j.N; g.KT = j.KT; g.pad = j.pad; g.pool_prev = 0;
        g.M = j.Bn * j.Tout;
        const size_t per = (size_t)j.N * j.Cin * j.KT;
        const int Z = (int)(st_gemm_wgrad_workspace_floats(j.Bn, j.Tout, j.Cin, j.N, j.KT) / (per + j.N));
        g.rows_per_z = (((g.M + Z - 1) / Z) + 31) / 32 * 32;
        g.fold = tn_fold(j.Cin, j.KT, 0) ? 1 : 0;
        g.lin = (j.KT == 1 && j.pad == 0 && j.Tin == j.Tout) ? 1 : 0;
        g.vecx = st_aligned16(j.dC) && (j.lddc % 4 == 0) && (j.dcoff % 4 == 0);
        g.vecy = st_aligned16(j.A) && (j.lda % 4 == 0);
        const bool dma = !g.fold && g.vecx && g.vecy && j.N % 4 == 0 && j.Cin % 4 == 0 && g.rows_per_z % 32 == 0;
        if (!dma || tn_tile(j.Cin, j.N, j.KT) != 64 || per >= (1ull << 32)) {
            const int rc = single(i, off);
            if (rc) return rc;
        } else {
            float* wsj = ws + off;
            const bool direct = Z == 1;
            g.part = direct ? j.dW : wsj;
            if (j.db) g.db_part = direct ? j.db : wsj + (size_t)Z * per;
            tb.g[m] = g;
            tb.gx[m] = (j.N + 63) / 64; tb.gy[m] = ((j.Cin + 63) / 64) * j.KT;
            tb.blk0[m] = blocks;
            blocks += tb.gx[m] * tb.gy[m] * Z;
            if (!direct) {
                sb.part[nsum] = wsj; sb.out[nsum] = j.dW; sb.part_b[nsum] = j.db ? wsj + (size_t)Z * per : wsj; sb.out_b[nsum] = j.db;
                sb.per[nsum] = (unsigned)per; sb.nb[nsum] = j.db ? j.N : 0; sb.Z[nsum] = Z; sb.acc[nsum] = 0;
                sb.blk0[nsum] = sblocks;
                sblocks += blocks_for(per + (j.db ? j.N : 0));
                ++nsum;
            }
            gi[m] = i; goff[m] = off;
            if (++m == TN_MAXJ) { const int rc = flush(); if (rc) return rc; }
        }
        off += tn_ws4(j);
    }
    return flush();
}

extern "C" int st_gemm_wgrad(const float* dC, int lddc, int dcoff, const float* A, int lda, float* dW, float* ws,
                             int Bn, int Tin, int Tout, int Cin, int N, int KT, int pad, int pool_prev, int accumulate,
                             void* stream) {
    return tn_impl(dC, lddc, dcoff, A, lda, dW, nullptr, ws, Bn, Tin, Tout, Cin, N, KT, pad, pool_prev, accumulate, stream);
}

// st_gemm_wgrad that also writes db[n] = sum over all rows of dC[:, dcoff + n] (the bias gradient) from the chunks of dC the product
// stages anyway
extern "C" int st_gemm_wgrad_db(const float* dC, int lddc, int dcoff, const float* A, int lda, float* dW, float* db, float* ws,
                                int Bn, int Tin, int Tout, int Cin, int N, int KT, int pad, int pool_prev, int accumulate,
                                void* stream) {
    ST_CHECK_ARG(db, "st_gemm_wgrad_db: null db");
    return tn_impl(dC, lddc, dcoff, A, lda, dW, db, ws, Bn, Tin, Tout, Cin, N, KT, pad, pool_prev, accumulate, stream);
}

// st_gemm_wgrad[_db] of a Linear over concatenated inputs, the result cut at input column `split`: dW0 (N, split) and dW1 (N, Cin - split)
// are the gradients of the two weights whose columns the product saw side by side ([W_ih | W_hh] of an LSTM cell fed with [x | h]);
// db (optional) = the column sums of dC, db_dup (optional) a second copy of them (b_ih and b_hh receive the same gradient).
extern "C" int st_gemm_wgrad_split(const float* dC, int lddc, int dcoff, const float* A, int lda, float* dW0, int split, float* dW1,
                                   float* db, float* db_dup, float* ws, int M, int Cin, int N, int accumulate, void* stream) {
    ST_CHECK_ARG(dW0 && dW1 && split > 0 && split < Cin, "st_gemm_wgrad_split: needs two outputs and 0 < split (%d) < Cin (%d)", split, Cin);
    ST_CHECK_ARG(db || !db_dup, "st_gemm_wgrad_split: db_dup without db");
    return tn_impl(dC, lddc, dcoff, A, lda, dW0, db, ws, 1, M, M, Cin, N, 1, 0, 0, accumulate, stream, dW1, split, db_dup);
}

extern "C" size_t st_colreduce_workspace_floats(int M, int N) { return (size_t)2 * st_colreduce_chunks(M) * N + 2 * (size_t)N; }

extern "C" int st_colsum(const float* X, int ldx, int xoff, const float* Y, int ldy, int yoff, int M, int N,
                         float* out, int accumulate, float* ws, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(X && out && ws && M > 0 && N > 0, "st_colsum: bad arguments");
    const int chunks = st_colreduce_chunks(M);
    const int rpc = (M + chunks - 1) / chunks;
    hipStream_t st = (hipStream_t)stream;
    if (!Y && M <= 256 && N >= 65536 && N % 4 == 0 && ldx % 4 == 0 && xoff % 4 == 0 && st_aligned16(X) && st_aligned16(out)) {
        hipLaunchKernelGGL(colsum_wide_kernel, dim3((N / 4 + 255) / 256), dim3(256), 0, st, X + xoff, ldx, M, N / 4, out, accumulate);
        ST_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, st, X, ldx, xoff, Y, ldy, yoff, M, N, rpc, ws);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(chunk_final_kernel, dim3((N + 15) / 16), dim3(256), 0, st, ws, chunks, 1, N, out, out, accumulate);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_act_bwd(const float* dout, int ldd, const float* out, int ldo, int act, const float* mask, int ldm,
                          float* dpre, int ldp, int M, int N, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dout && dpre && M > 0 && N > 0 && (act == ST_ACT_NONE || out), "st_act_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for((size_t)M * N)), dim3(256), 0, (hipStream_t)stream,
                       dout, ldd, out, ldo, act, mask, ldm, dpre, ldp, M, N);
    ST_LAUNCH_CHECK();
    return 0;
}

// BatchNorm backward in two halves so that a data-parallel caller can all-reduce the two sums in between (SyncBN):
//   reduce: s[0:N] = sum_rows dyb, s[N:2N] = sum_rows dyb * xhat      (local rows)
//   apply:  dx = w/sigma * (dyb - s1/Mstat - xhat * s2/Mstat)          (Mstat = rows behind mean / var / s)
extern "C" int st_bn_bwd_reduce(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff, int act,
                                const float* x, int ldx, int xoff, const float* mean, const float* var, float eps,
                                int M, int N, float* s, float* ws, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dy && x && mean && var && s && ws && M > 0 && N > 0 && (act == ST_ACT_NONE || y), "st_bn_bwd_reduce: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    float* part = ws + 2 * (size_t)N;
    const int chunks = st_colreduce_chunks(M);
    const int rpc = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, st, dy, ldd, doff, y, ldy, yoff, act,
                       x, ldx, xoff, mean, var, eps, M, N, rpc, part);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(chunk_final_kernel, dim3((2 * N + 15) / 16), dim3(256), 0, st, part, chunks, 2, N, s, s + N, 0);
    ST_LAUNCH_CHECK();
    return 0;
}

static int bn_bwd_apply_impl(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff, int act,
                             const float* x, int ldx, int xoff, const float* mean, const float* var, const float* w, float eps,
                             int M, int N, const float* s, int Mstat, const float* inv_total, float* dx, int lddx, int dxoff, void* stream,
                             const float* mask = nullptr, int ldm = 0, float* dres = nullptr, int lddr = 0) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dy && x && mean && var && s && dx && M > 0 && N > 0 && Mstat >= M && (act == ST_ACT_NONE || y), "st_bn_bwd_apply: bad arguments");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for((size_t)M * N)), dim3(256), 0, (hipStream_t)stream, dy, ldd, doff, y, ldy, yoff,
                       act, x, ldx, xoff, mean, var, w, eps, M, N, s, s + N, dx, lddx, dxoff, Mstat, inv_total, mask, ldm, dres, lddr);
    ST_LAUNCH_CHECK();
    return 0;
}

// The two halves of the BatchNorm backward for a layer whose forward was Y = (act(BN(x)) + res) * mask (st_bn_norm_res_mask_fwd; ConvLayer,
// src/module.py:641-646): the incoming gradient is multiplied by the mask on the way in (no launch of its own), `y` is the kept act(BN(x)),
// and the apply half also hands out dres = dy * mask, the gradient of the residual input (NULL: no residual).  inv_total as in
// st_bn_bwd_apply_sync (NULL: the sums are divided by Mstat).
extern "C" int st_bn_bwd_reduce_masked(const float* dy, int ldd, const float* mask, int ldm, const float* y, int ldy, int act,
                                       const float* x, int ldx, const float* mean, const float* var, float eps,
                                       int M, int N, float* s, float* ws, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dy && x && mean && var && s && ws && M > 0 && N > 0 && (act == ST_ACT_NONE || y), "st_bn_bwd_reduce_masked: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    float* part = ws + 2 * (size_t)N;
    const int chunks = st_colreduce_chunks(M);
    const int rpc = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, st, dy, ldd, 0, y, ldy, 0, act,
                       x, ldx, 0, mean, var, eps, M, N, rpc, part, mask, ldm);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(chunk_final_kernel, dim3((2 * N + 15) / 16), dim3(256), 0, st, part, chunks, 2, N, s, s + N, 0);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_bn_bwd_apply_masked(const float* dy, int ldd, const float* mask, int ldm, const float* y, int ldy, int act,
                                      const float* x, int ldx, const float* mean, const float* var, const float* w, float eps,
                                      int M, int N, const float* s, int Mstat, const float* inv_total, float* dx, int lddx,
                                      float* dres, int lddr, void* stream) {
    return bn_bwd_apply_impl(dy, ldd, 0, y, ldy, 0, act, x, ldx, 0, mean, var, w, eps, M, N, s, inv_total ? M : Mstat, inv_total, dx, lddx, 0, stream,
                             mask, ldm, dres, lddr);
}

extern "C" int st_bn_bwd_apply(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff, int act,
                               const float* x, int ldx, int xoff, const float* mean, const float* var, const float* w, float eps,
                               int M, int N, const float* s, int Mstat, float* dx, int lddx, int dxoff, void* stream) {
    return bn_bwd_apply_impl(dy, ldd, doff, y, ldy, yoff, act, x, ldx, xoff, mean, var, w, eps, M, N, s, Mstat, nullptr, dx, lddx, dxoff, stream);
}

// st_bn_bwd_apply under SyncBN: s holds the sums over ALL ranks' rows and *inv_total (device scalar, st_bn_sync_merge) = 1 / their count
extern "C" int st_bn_bwd_apply_sync(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff, int act,
                                    const float* x, int ldx, int xoff, const float* mean, const float* var, const float* w, float eps,
                                    int M, int N, const float* s, const float* inv_total, float* dx, int lddx, int dxoff, void* stream) {
    ST_CHECK_ARG(inv_total, "st_bn_bwd_apply_sync: null inv_total");
    return bn_bwd_apply_impl(dy, ldd, doff, y, ldy, yoff, act, x, ldx, xoff, mean, var, w, eps, M, N, s, M, inv_total, dx, lddx, dxoff, stream);
}



extern "C" int st_bn_bwd(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff, int act,
                         const float* x, int ldx, int xoff, const float* mean, const float* var, const float* w, float eps,
                         int M, int N, float* dx, int lddx, int dxoff, float* dw, float* db, int accumulate, float* ws,
                         void* stream) {
    ST_CHECK_ARG(ws, "st_bn_bwd: null workspace");
    int rc = st_bn_bwd_reduce(dy, ldd, doff, y, ldy, yoff, act, x, ldx, xoff, mean, var, eps, M, N, ws, ws, stream);
    if (rc) return rc;
    rc = st_bn_bwd_apply(dy, ldd, doff, y, ldy, yoff, act, x, ldx, xoff, mean, var, w, eps, M, N, ws, M, dx, lddx, dxoff, stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (db) {   // d beta = s1, d gamma = s2
        hipLaunchKernelGGL(sum_partials_kernel, dim3(blocks_for(N)), dim3(256), 0, st, ws, db, (size_t)N, 1, accumulate);
        ST_LAUNCH_CHECK();
    }
    if (dw) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(blocks_for(N)), dim3(256), 0, st, ws + N, dw, (size_t)N, 1, accumulate);
        ST_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int st_highway_bwd(const float* dy, const float* H, const float* x, const float* Tgate,
                              float* dH, float* dT, float* dx_direct, size_t total, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dy && H && x && Tgate && dH && dT && dx_direct && total > 0, "st_highway_bwd: bad arguments");
    hipLaunchKernelGGL(highway_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       dy, H, x, Tgate, dH, dT, dx_direct, total);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t st_bn_bank_workspace_floats(int nseg, int max_rows, int N) {
    return (size_t)nseg * st_colreduce_chunks(max_rows) * 2 * N;
}

static int bnb_fill(BnBank& a, const st_bn_bank_seg* segs, int nseg, int Bn, int N, int& max_chunks, size_t& max_rows, const char* who) {
    ST_CHECK_ARG(segs && nseg > 0 && nseg <= ST_BN_BANK_MAX && Bn > 0 && N > 0, "%s: bad arguments (at most %d segments)", who, ST_BN_BANK_MAX);
    a.nseg = nseg; a.Bn = Bn; a.N = N;
    max_chunks = 1; max_rows = 0;
    for (int k = 0; k < nseg; ++k) {
        ST_CHECK_ARG(segs[k].x && segs[k].T > 0 && segs[k].ldx >= N && segs[k].mean && segs[k].var, "%s: segment %d incomplete", who, k);
        a.s[k] = segs[k];
        const int M = Bn * segs[k].T;
        max_chunks = max(max_chunks, st_colreduce_chunks(M));
        max_rows = max(max_rows, (size_t)M);
    }
    return 0;
}

static int bnb_norm_launch(const BnBank& a, const st_bn_bank_seg* segs, int nseg, int Bn, int N, float* Y, int ldy, int Tout, hipStream_t st);

extern "C" int st_bn_bank_fwd(const st_bn_bank_seg* segs, int nseg, int Bn, int N, float* Y, int ldy, int Tout, float* ws, void* stream) {
    (void)hipGetLastError();
    BnBank a; int mc; size_t mr;
    { const int rc = bnb_fill(a, segs, nseg, Bn, N, mc, mr, "st_bn_bank_fwd"); if (rc) return rc; }
    ST_CHECK_ARG(Y && ws && Tout > 0 && ldy >= nseg * N, "st_bn_bank_fwd: bad output / workspace");
    for (int k = 0; k < nseg; ++k) ST_CHECK_ARG(segs[k].T >= Tout, "st_bn_bank_fwd: segment %d has %d frames, the bank %d", k, segs[k].T, Tout);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bnb_stats_chunk_kernel, dim3((N + 63) / 64, mc, nseg), dim3(256), 0, st, a, ws, mc);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(bnb_stats_final_kernel, dim3((N + 15) / 16, nseg), dim3(256), 0, st, a, ws, mc, (float*)nullptr);
    ST_LAUNCH_CHECK();
    return bnb_norm_launch(a, segs, nseg, Bn, N, Y, ldy, Tout, st);
}

// ---- the same in its SyncBN stages (the collectives between them are the caller's): records -> [all-gather] -> merge -> normalise
extern "C" int st_bn_bank_stats_record(const st_bn_bank_seg* segs, int nseg, int Bn, int N, float* rec, float* ws, void* stream) {
    (void)hipGetLastError();
    BnBank a; int mc; size_t mr;
    { const int rc = bnb_fill(a, segs, nseg, Bn, N, mc, mr, "st_bn_bank_stats_record"); if (rc) return rc; }
    ST_CHECK_ARG(rec && ws, "st_bn_bank_stats_record: null record / workspace");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bnb_stats_chunk_kernel, dim3((N + 63) / 64, mc, nseg), dim3(256), 0, st, a, ws, mc);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(bnb_stats_final_kernel, dim3((N + 15) / 16, nseg), dim3(256), 0, st, a, ws, mc, rec);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_bn_bank_sync_merge(const st_bn_bank_seg* segs, int nseg, int Bn, int N, const float* allrec, int world, float* inv_total,
                                     void* stream) {
    (void)hipGetLastError();
    BnBank a; int mc; size_t mr;
    { const int rc = bnb_fill(a, segs, nseg, Bn, N, mc, mr, "st_bn_bank_sync_merge"); if (rc) return rc; }
    ST_CHECK_ARG(allrec && world > 0 && inv_total, "st_bn_bank_sync_merge: bad arguments");
    hipLaunchKernelGGL(bnb_sync_merge_kernel, dim3((N + 255) / 256, nseg), dim3(256), 0, (hipStream_t)stream, a, allrec, world, inv_total);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_bn_bank_norm(const st_bn_bank_seg* segs, int nseg, int Bn, int N, float* Y, int ldy, int Tout, void* stream) {
    (void)hipGetLastError();
    BnBank a; int mc; size_t mr;
    { const int rc = bnb_fill(a, segs, nseg, Bn, N, mc, mr, "st_bn_bank_norm"); if (rc) return rc; }
    ST_CHECK_ARG(Y && Tout > 0 && ldy >= nseg * N, "st_bn_bank_norm: bad output");
    for (int k = 0; k < nseg; ++k) ST_CHECK_ARG(segs[k].T >= Tout, "st_bn_bank_norm: segment %d has %d frames, the bank %d", k, segs[k].T, Tout);
    return bnb_norm_launch(a, segs, nseg, Bn, N, Y, ldy, Tout, (hipStream_t)stream);
}

static int bnb_norm_launch(const BnBank& a, const st_bn_bank_seg* segs, int nseg, int Bn, int N, float* Y, int ldy, int Tout, hipStream_t st) {
    int vec = N % 4 == 0 && ldy % 4 == 0 && st_aligned16(Y);
    for (int k = 0; k < nseg && vec; ++k)
        vec = segs[k].ldx % 4 == 0 && st_aligned16(segs[k].x) && st_aligned16(segs[k].mean) && st_aligned16(segs[k].var) &&
              (!segs[k].w || st_aligned16(segs[k].w)) && (!segs[k].b || st_aligned16(segs[k].b));
    hipLaunchKernelGGL(bnb_norm_kernel, dim3(blocks_for((size_t)Bn * Tout * N / (vec ? 4 : 1), 1024), nseg), dim3(256), 0, st, a, Y, ldy, Tout, vec);
    ST_LAUNCH_CHECK();
    return 0;
}

static int bnb_bwd_common(const st_bn_bank_seg* segs, int nseg, int Bn, int N, const float* dY, int lddy, int Tout, int relu_in, float* ws,
                          int stage, const float* inv_total, void* stream, const char* who) {
    BnBank a; int mc; size_t mr;
    { const int rc = bnb_fill(a, segs, nseg, Bn, N, mc, mr, who); if (rc) return rc; }
    ST_CHECK_ARG(dY && Tout > 0 && lddy >= nseg * N && (stage == 2 || ws), "%s: bad gradient / workspace", who);
    for (int k = 0; k < nseg; ++k)
        ST_CHECK_ARG(segs[k].sums && (stage == 1 || (segs[k].dx && segs[k].lddx >= N)) && segs[k].T >= Tout, "%s: segment %d incomplete", who, k);
    hipStream_t st = (hipStream_t)stream;
    if (stage != 2) {        // sums of this rank's rows
        hipLaunchKernelGGL(bnb_bwd_reduce_kernel, dim3((N + 63) / 64, mc, nseg), dim3(256), 0, st, a, dY, lddy, Tout, ws, mc);
        ST_LAUNCH_CHECK();
        hipLaunchKernelGGL(bnb_bwd_final_kernel, dim3((2 * N + 15) / 16, nseg), dim3(256), 0, st, a, ws, mc);
        ST_LAUNCH_CHECK();
    }
    if (stage != 1) {
        int vec = N % 4 == 0 && lddy % 4 == 0 && st_aligned16(dY);
        for (int k = 0; k < nseg && vec; ++k)
            vec = segs[k].ldx % 4 == 0 && segs[k].lddx % 4 == 0 && st_aligned16(segs[k].x) && st_aligned16(segs[k].dx) && st_aligned16(segs[k].mean) &&
                  st_aligned16(segs[k].var) && st_aligned16(segs[k].sums) && (!segs[k].w || st_aligned16(segs[k].w));
        hipLaunchKernelGGL(bnb_bwd_apply_kernel, dim3(blocks_for(mr * N / (vec ? 4 : 1), 1024), nseg), dim3(256), 0, st, a, dY, lddy, Tout, relu_in, vec,
                           inv_total);
        ST_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int st_bn_bank_bwd(const st_bn_bank_seg* segs, int nseg, int Bn, int N, const float* dY, int lddy, int Tout, int relu_in, float* ws,
                              void* stream) {
    (void)hipGetLastError();
    return bnb_bwd_common(segs, nseg, Bn, N, dY, lddy, Tout, relu_in, ws, 0, nullptr, stream, "st_bn_bank_bwd");
}

// SyncBN stages: local sums -> [all-reduce by the caller] -> apply with the global sums and 1 / (global row count) per segment
extern "C" int st_bn_bank_bwd_reduce(const st_bn_bank_seg* segs, int nseg, int Bn, int N, const float* dY, int lddy, int Tout, float* ws, void* stream) {
    (void)hipGetLastError();
    return bnb_bwd_common(segs, nseg, Bn, N, dY, lddy, Tout, 0, ws, 1, nullptr, stream, "st_bn_bank_bwd_reduce");
}

extern "C" int st_bn_bank_bwd_apply(const st_bn_bank_seg* segs, int nseg, int Bn, int N, const float* dY, int lddy, int Tout, int relu_in,
                                    const float* inv_total, void* stream) {
    (void)hipGetLastError();
    return bnb_bwd_common(segs, nseg, Bn, N, dY, lddy, Tout, relu_in, nullptr, 2, inv_total, stream, "st_bn_bank_bwd_apply");
}

extern "C" int st_highway_ht_fwd(const float* ht, const float* x, float* y, int M, int C, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(ht && x && y && M > 0 && C > 0, "st_highway_ht_fwd: bad arguments");
    const size_t total = (size_t)M * C;
    hipLaunchKernelGGL(highway_ht_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, ht, x, y, C, total);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_highway_ht_bwd(const float* dy, const float* ht, const float* x, float* dht, float* dx_direct, int M, int C, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dy && ht && x && dht && dx_direct && M > 0 && C > 0, "st_highway_ht_bwd: bad arguments");
    const size_t total = (size_t)M * C;
    hipLaunchKernelGGL(highway_ht_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, ht, x, dht, dx_direct, C, total);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_pool_prev_fwd(const float* x, float* y, int Bn, int T, int C, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(x && y && Bn > 0 && T > 0 && C > 0 && C % 4 == 0 && st_aligned16(x) && st_aligned16(y), "st_pool_prev_fwd: bad arguments (C %% 4 == 0, 16-byte aligned)");
    const size_t total4 = (size_t)Bn * T * C / 4;
    size_t blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pool_prev_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, T, C / 4, total4);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_pool_prev_bwd(const float* dy_pooled, const float* x, float* dx, int Bn, int T, int C, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dy_pooled && x && dx && Bn > 0 && T > 0 && C > 0, "st_pool_prev_bwd: bad arguments");
    hipLaunchKernelGGL(pool_prev_bwd_kernel, dim3(blocks_for((size_t)Bn * T * C)), dim3(256), 0, (hipStream_t)stream,
                       dy_pooled, x, dx, Bn, T, C);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_copy3d(float* dst, long dst_sb, long dst_st, const float* src, long src_sb, long src_st,
                         int Bn, int T, int C, int accumulate, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dst && src && Bn > 0 && T > 0 && C > 0, "st_copy3d: bad arguments");
    hipLaunchKernelGGL(copy3d_kernel, dim3(blocks_for((size_t)Bn * T * C)), dim3(256), 0, (hipStream_t)stream,
                       dst, dst_sb, dst_st, src, src_sb, src_st, Bn, T, C, accumulate);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_relayout_blocks(int N, int Cin, int KT) { return (int)(((size_t)N * Cin * KT + 1023) / 1024); }

extern "C" int st_relayout_batch(const st_relayout_desc* table_dev, int n, int total_blocks, const int* blk_desc_dev, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(table_dev && n > 0 && total_blocks > 0, "st_relayout_batch: bad arguments");
    hipLaunchKernelGGL(relayout_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n, blk_desc_dev);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_scatter_add_rows(const float* dout, const int64_t* idx, float* dtable, int n, int D, int V, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dout && idx && dtable && n > 0 && D > 0 && V > 0, "st_scatter_add_rows: bad arguments");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(V, (D + 63) / 64), dim3(512), 0, (hipStream_t)stream,
                       dout, idx, dtable, n, D, V);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_bn_norm_fwd(const float* X, int ldx, int xoff, float* Y, int ldy, int yoff, int M, int N,
                              const float* mean, const float* var, const float* w, const float* b, float eps, int act,
                              void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(X && Y && mean && var && M > 0 && N > 0, "st_bn_norm_fwd: bad arguments");
    hipLaunchKernelGGL(bn_norm_fwd_kernel, dim3(blocks_for((size_t)M * N)), dim3(256), 0, (hipStream_t)stream,
                       X, ldx, xoff, Y, ldy, yoff, M, N, mean, var, w, b, eps, act);
    ST_LAUNCH_CHECK();
    return 0;
}

// Tact (M, N) = act(BatchNorm(X)) (may be NULL when nobody needs it), Y (M, N) = (Tact + res) * mask; X, res, mask, Tact, Y contiguous rows
// of N floats, N % 4 == 0, 16-byte aligned.  ref: ConvLayer.forward, src/module.py:641-646
extern "C" int st_bn_norm_res_mask_fwd(const float* X, float* Tact, float* Y, int M, int N, const float* mean, const float* var,
                                       const float* w, const float* b, float eps, int act, const float* res, const float* mask,
                                       void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(X && Y && mean && var && M > 0 && N > 0 && N % 4 == 0, "st_bn_norm_res_mask_fwd: bad arguments (N %% 4 == 0)");
    ST_CHECK_ARG(st_aligned16(X) && st_aligned16(Y) && st_aligned16(Tact) && st_aligned16(res) && st_aligned16(mask) && st_aligned16(mean) && st_aligned16(var) && st_aligned16(w) && st_aligned16(b),
                 "st_bn_norm_res_mask_fwd: 16-byte aligned operands");
    const size_t total4 = (size_t)M * (N / 4);
    hipLaunchKernelGGL(bn_norm_res_mask_fwd_kernel, dim3(blocks_for(total4)), dim3(256), 0, (hipStream_t)stream,
                       X, Tact, Y, N / 4, total4, mean, var, w, b, eps, act, res, mask);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_highway_fwd(const float* H, const float* Tgate, const float* x, float* y, size_t total, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(H && Tgate && x && y && total > 0, "st_highway_fwd: bad arguments");
    hipLaunchKernelGGL(highway_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, H, Tgate, x, y, total);
    ST_LAUNCH_CHECK();
    return 0;
}
