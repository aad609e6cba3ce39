// vq.hip -- VQ codebook kernels for gfx950 (MI355X): table build, row gather, nearest-code
// search with softmax/argmax, and the row softmax+argmax of the 'seperate' codebook.
//
// Nearest-code search (ref: L2Embedding.forward src/embed.py:105-147, neg_batch_l2 :208-213):
//   one wavefront per input vector; the whole table is staged once per workgroup in LDS,
//   TRANSPOSED ([d][v], so the 64 lanes of a wave read 64 consecutive codes of one dimension
//   from 64 distinct banks), together with |e|^2 per code.  Each lane scores the codes
//   v = lane, lane+64, ...; the softmax max/sum and the argmax are wavefront shuffle
//   reductions.  argmax is taken over the softmax output p (as the reference does) and the
//   FIRST maximum wins (torch.argmax semantics).  The straight-through forward value is
//   fl(fl(x + e) - x), not e (embed.py:145).
// HBM traffic per vector: D*4 in, D*4 out, 8 idx, V*4 p_code; the table is read once per
// workgroup (grid is capped so each workgroup handles many vectors).
#include "st_common.h"
#include <cstdlib>

namespace {

constexpr int VQ_WAVES = 4;

__global__ __launch_bounds__(256) void vq_build_table_kernel(const float* learnable, int Dl, const float* attr,
                                                             int n_attr, const float* attr_w, const float* attr_b,
                                                             int Da, float* table, int V) {
    const int D = Dl + Da;
    const int total = V * D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int v = i / D, d = i - v * D;
        float val;
        if (d < Dl) {
            val = learnable[(size_t)v * Dl + d];
        } else {  // proj_attr(phn_attr.weight)[v][d - Dl]
            const int o = d - Dl;
            float acc = 0.0f;
            for (int k = 0; k < n_attr; ++k) acc = fmaf(attr[(size_t)v * n_attr + k], attr_w[(size_t)o * n_attr + k], acc);
            val = acc + attr_b[o];
        }
        table[i] = val;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* table, const int64_t* idx, float* out,
                                                          int n, int D, int V) {
    const size_t total = (size_t)n * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / D;
        const int d = (int)(i - r * D);
        int64_t v = idx[r];
        if (v < 0) v = 0;
        if (v >= V) v = V - 1;
        out[i] = table[(size_t)v * D + d];
    }
}

// wave-level argmax with "first maximum wins": larger value, or equal value and smaller index
__device__ __forceinline__ void wave_argmax(float& val, int& idx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(val, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
    }
}

// softmax + argmax over sims[0..V) held in wave-private LDS; writes p (global) and returns idx
__device__ __forceinline__ int wave_softmax_argmax(float* sims, int V, int lane, float* p_out) {
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, sims[v]);
    m = st_wave_max(m);
    float s = 0.0f;
    for (int v = lane; v < V; v += 64) { const float e = expf(sims[v] - m); sims[v] = e; s += e; }
    s = st_wave_sum(s);
    float best = -1.0f;
    int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
        const float p = sims[v] / s;
        p_out[v] = p;
        if (p > best) { best = p; bi = v; }   // ascending v per lane: strict > keeps the first
    }
    wave_argmax(best, bi);
    return bi;
}

__global__ __launch_bounds__(VQ_WAVES * 64) void vq_l2_kernel(const float* x, const float* table, const float* temp,
                                                              float* p_code, int64_t* idx_out, float* out,
                                                              int n, int D, int V) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Vp = V + 1;                       // padded row length of the transposed table
    float* et = lds;                            // [D][Vp]
    float* e2 = et + (size_t)D * Vp;            // [V]
    float* xs = e2 + V;                         // [VQ_WAVES][D]
    float* sims = xs + VQ_WAVES * D;            // [VQ_WAVES][V]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < V * D; i += blockDim.x) {     // coalesced read, transposed LDS write
        const int v = i / D, d = i - v * D;
        et[d * Vp + v] = table[i];
    }
    __syncthreads();
    for (int v = tid; v < V; v += blockDim.x) {         // sum(y.pow(2), -1)
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) { const float e = et[d * Vp + v]; acc = fmaf(e, e, acc); }
        e2[v] = acc;
    }
    __syncthreads();
    const float tscale = fmaxf(temp[0], 0.0f);          // F.relu(self.temp)
    float* myx = xs + wave * D;
    float* mysim = sims + wave * V;
    for (int r = blockIdx.x * VQ_WAVES + wave; r < n; r += gridDim.x * VQ_WAVES) {
        const float* xr = x + (size_t)r * D;
        float xx = 0.0f;
        for (int d = lane; d < D; d += 64) { const float xv = xr[d]; myx[d] = xv; xx = fmaf(xv, xv, xx); }
        xx = st_wave_sum(xx);                           // sum(flat_x.pow(2), -1)
        for (int v = lane; v < V; v += 64) {
            float dot = 0.0f;
            for (int d = 0; d < D; ++d) dot = fmaf(myx[d], et[d * Vp + v], dot);
            const float dist = (xx + e2[v]) - 2.0f * dot;     // embed.py:210-212 association order
            mysim[v] = tscale * (-dist);
        }
        const int bi = wave_softmax_argmax(mysim, V, lane, p_code + (size_t)r * V);
        if (lane == 0) idx_out[r] = bi;
        for (int d = lane; d < D; d += 64) {
            const float xv = myx[d];
            out[(size_t)r * D + d] = (xv + et[d * Vp + bi]) - xv;   // x + code - x.detach()
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Nearest-code search on the matrix cores (D <= 64, D % 4 == 0, V <= 1024): sims = X E^T is a GEMM with K = D, and the
// exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: an fmaf chain over k, bitwise) reproduces the dimension-ascending dot product of
// the scalar kernel above, so the indices stay what the reference's golden vectors pin.
//   * the table is re-packed once per table version into MFMA B-operand order (vq_pack_table_kernel, L2 resident):
//     [code tile of 16][group of 4 k-steps][lane][4 floats], plus |e|^2 per code -- a wave fetches a tile's operands with
//     four 16-byte loads per lane;
//   * PERSISTENT tile loop: a workgroup (4 waves, one per SIMD; two workgroups per compute unit when the registers allow) fetches
//     its waves' table operands ONCE, keeps them in registers and walks the 16-vector tiles blockIdx.x, + gridDim.x, ...; the next
//     tile's vectors are requested while the current one is scored.  (One tile per workgroup re-requested the whole packed table
//     -- 131 KB at V = 512 -- per 16 vectors: 3.2x the problem's bytes in L2 reads, each behind a launch-to-first-MFMA latency.)
//   * wave w scores TPW code tiles; the softmax max / sum and the argmax are DPP reductions over the 16 lanes that hold one
//     vector's codes, then over the waves through LDS;
//   * code <-> MFMA column: with TPW % 4 == 0 wave w owns the CONTIGUOUS codes [16 TPW w, 16 TPW (w + 1)) and column nn of its tile t
//     is code 16 TPW w + 64 (t >> 2) + 4 nn + (t & 3): a lane then holds four consecutive codes of a vector (tiles 4h .. 4h + 3)
//     and p_code leaves as 16-byte stores, 256 contiguous bytes per 16 lanes (it is 4V of the 520 + 4V bytes per vector).
//     Per lane the codes still ascend with t (first maximum wins); TPW < 4 keeps code = 16 (w + NW t) + nn and 4-byte stores;
//   * |x|^2 uses the same summation tree as the scalar kernel's wave butterfly (dims d, d^32 first ... d^1 last).
// Algorithmic bytes per vector: 4D in + 4D out + 8 idx + 4V p_code; the packed table is read once per workgroup.
constexpr int VQM_XLD = 68;
// waves per workgroup: 4 (one per SIMD) up to 32 code tiles (V <= 512), 8 beyond.  (Measured and rejected in round 4: 8 waves x 4
// tiles at V = 512 so that two workgroups = four waves per SIMD stay resident -- the 128-register budget spills 42 registers and the
// search takes 58 instead of 37.5 us.  The exact-fp32 MFMA runs on the vector ALUs: it does not overlap with another wave's VALU
// work on the same SIMD -- phase stamps: the partner's 4096-cycle MFMA burst stretches a 1.2k-cycle softmax phase to 5.3k -- so more
// resident waves can only hide latency, never add a second pipe.)
__host__ __device__ inline int vq_nw(int n_ct) { return n_ct <= 32 ? 4 : 8; }
// code tiles per wave for a table of n_ct code tiles (the kernel instantiations below)
__host__ __device__ inline int vq_tpw(int n_ct) { return n_ct <= 4 ? 1 : n_ct <= 8 ? 2 : n_ct <= 16 ? 4 : 8; }
// code held by column nn of packed tile pt = wave + NW * t
__host__ __device__ inline int vq_code_of(int nw, int tpw, int pt, int nn) {
    const int wave = pt % nw, t = pt / nw;
    return tpw % 4 == 0 ? wave * 16 * tpw + (t >> 2) * 64 + nn * 4 + (t & 3) : pt * 16 + nn;
}

__global__ __launch_bounds__(256) void vq_pack_table_kernel(const float* table, int V, int D, float* ws, int n_pt, int ks4n, int nw, int tpw) {
    const size_t total = (size_t)n_pt * ks4n * 256;
    float* e2 = ws + total;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total + (size_t)n_pt * 16; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < total) {
            const int c = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
            const int blk = (int)(idx >> 8), ks4 = blk % ks4n, pt = blk / ks4n;
            const int code = vq_code_of(nw, tpw, pt, lane & 15), d = (ks4 * 4 + c) * 4 + (lane >> 4);
            ws[idx] = (code < V && d < D) ? table[(size_t)code * D + d] : 0.0f;
        } else {                                        // sum(y.pow(2), -1): dimension-ascending fma chain (as the scalar kernel)
            const int slot = (int)(idx - total);
            const int code = vq_code_of(nw, tpw, slot >> 4, slot & 15);
            float acc = 0.0f;
            if (code < V) {
                f32x4 row[16];          // the whole row is requested up front (D <= 64, D % 4 == 0: checked on the host)
#pragma unroll
                for (int q = 0; q < 16; ++q) row[q] = 4 * q < D ? *reinterpret_cast<const f32x4*>(table + (size_t)code * D + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc = fmaf(row[q][c], row[q][c], acc);      // zeros past D leave the chain unchanged
            }
            e2[slot] = acc;
        }
    }
}

template <int CTRL>
__device__ __forceinline__ int vq_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

// all-reduce over the 16 lanes of a DPP row
__device__ __forceinline__ float vq_row_max(float v) {
    v = fmaxf(v, st_dpp<ST_DPP_QUAD_XOR1>(v)); v = fmaxf(v, st_dpp<ST_DPP_QUAD_XOR2>(v));
    v = fmaxf(v, st_dpp<ST_DPP_ROW_HALF_MIRROR>(v)); v = fmaxf(v, st_dpp<ST_DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float vq_row_sum(float v) {
    v += st_dpp<ST_DPP_QUAD_XOR1>(v); v += st_dpp<ST_DPP_QUAD_XOR2>(v);
    v += st_dpp<ST_DPP_ROW_HALF_MIRROR>(v); v += st_dpp<ST_DPP_ROW_MIRROR>(v);
    return v;
}
#ifdef VQ_STAMPS      // phase stamps of workgroup 0 / wave 0, second tile (experiment builds only: tools/exp_vq_stamps.py)
__device__ unsigned long long vq_stamps[32];
#define VQ_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && tile == (int)gridDim.x) vq_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define VQ_STAMP(i) do { } while (0)
#endif

// exp(x) for x <= 0 as the softmax needs it: one multiply + v_exp_f32 (libm's expf is 13 instructions, half of them quarter-rate:
// phase stamps showed 7.0k of a tile's 21k cycles in the 32 exponentials per lane).  Accurate to ~1 ulp where the argmax is decided
// (x ~ 0: exp2 of a correctly rounded product), relative error ~ |x| 2^-24 in the tail; weakly monotone like expf, and exp(0) = 1
// exactly.  The reference's own p_code comes from ATen's vectorised exp, so no form is bitwise "the" reference; the golden vectors
// (ties, near-ties, duplicates) pin the indices.
__device__ __forceinline__ float vq_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// e / s with ONE IEEE division per row (r = 1 / s) and two fused multiply-adds per element: q = e r, q' = q + r (e - q s) is the
// correctly rounded quotient whenever r is the correctly rounded reciprocal (Markstein), i.e. the value `e / s` itself -- at 3
// instructions instead of the 11 of the division sequence (32 of them per lane and tile)
__device__ __forceinline__ float vq_div(float e, float s, float r) {
    const float q = e * r;
    return fmaf(fmaf(-q, s, e), r, q);
}

// TPW code tiles per wave (V <= 64 TPW); WPS = workgroups the launcher places per compute unit (register budget 512 / WPS per lane);
// FULL: V == 64 TPW, every code of every tile exists (no masks in the loop)
template <int NW, int TPW, int WPS, bool FULL>
__global__ __launch_bounds__(NW * 64, WPS * NW / 4) void vq_l2_mfma_kernel(const float* x, const float* table, const float* ws, const float* temp,
                                                                         float* p_code, int64_t* idx_out, float* out, int n, int D, int V,
                                                                         int ks4n, int n_tiles, int p_vec4) {
    constexpr bool PERM = TPW % 4 == 0;
    __shared__ __attribute__((aligned(16))) float xts[2][16 * VQM_XLD];      // the tile's vectors, double-buffered: the output stage of
    __shared__ float xxs[NW][16];                                           // tile i reads them during tile i + 1
    __shared__ float red[NW][16];
    __shared__ int redi[NW][16];
    __shared__ int fidx[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nn = lane & 15;
    const f32x4* wsp = reinterpret_cast<const f32x4*>(ws);
    const float* e2 = ws + (size_t)NW * TPW * ks4n * 256;
    // the first tile's vectors (staging role of this thread: 16 bytes of vector tid >> 4; rows past n repeat the last one and are
    // zeroed when they are written to LDS -- no branch around the load)
    const int svec = (tid >> 4) & 15, sd4 = min((tid & 15) * 4, D - 4);
    const bool sd_ok = (tid & 15) * 4 < D && tid < 256;      // (the first four waves stage the 16 vectors and write their output rows)
    int tile = blockIdx.x;
    f32x4 xr = *reinterpret_cast<const f32x4*>(x + (size_t)min(tile * 16 + svec, n - 1) * D + sd4);
    // every table operand of this wave: requested once, kept in registers for all the tiles of the workgroup
    f32x4 bq[TPW][4];
    float e2v[TPW];
    float off[FULL ? 1 : TPW];            // 0 for an existing code, -inf past the table: added to the similarity (x + 0 == x)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int pt = wave + NW * t;
        e2v[t] = e2[pt * 16 + nn];
        if (!FULL) off[t] = vq_code_of(NW, TPW, pt, nn) < V ? 0.0f : -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[t][q] = q < ks4n ? wsp[((size_t)pt * ks4n + q) * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float tscale = fmaxf(temp[0], 0.0f);          // F.relu(self.temp)
    // every barrier of the loop orders LDS traffic only (st_lds_barrier): __syncthreads() also waits for the global stores of p_code
    // and for the prefetched vectors of the next tile -- a memory round trip per barrier, seven per tile
    f32x4 oe = {0.f, 0.f, 0.f, 0.f};                    // the previous tile's picked code row (in flight)
    int ov = -1;                                        // ... of vector ov
    int tog = 0;
    for (; tile < n_tiles; tile += gridDim.x, tog ^= 1) {
        const int v0 = tile * 16;
        float* xt = xts[tog];
        VQ_STAMP(0);
        if (tid < 256) *reinterpret_cast<f32x4*>(xt + svec * VQM_XLD + (tid & 15) * 4) = (sd_ok && v0 + svec < n) ? xr : f32x4{0.f, 0.f, 0.f, 0.f};
        st_lds_barrier();
        VQ_STAMP(1);
        {   // the next tile's vectors travel while this one is scored
            const int nt = tile + (int)gridDim.x;
            xr = *reinterpret_cast<const f32x4*>(x + (size_t)min(min(nt, n_tiles - 1) * 16 + svec, n - 1) * D + sd4);
        }
        {   // |x|^2 with the butterfly tree of the scalar kernel: lane (vec = lane >> 2, j = lane & 3) owns dims j, j + 4, ...
            const int vec = lane >> 2, j = lane & 3;
            float a[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { const float t = xt[vec * VQM_XLD + j + 4 * m]; a[m] = t * t; }
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] += a[m + 8];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] += a[m + 4];
            a[0] += a[2]; a[1] += a[3];
            float f = a[0] + a[1];
            f += st_dpp<ST_DPP_QUAD_XOR2>(f);
            f += st_dpp<ST_DPP_QUAD_XOR1>(f);
            if (j == 0) xxs[wave][vec] = f;
        }
        asm volatile("" ::: "memory");                  // (the 16 squares above die before the fragments are read: register pressure)
        float ax[16];                                   // A fragments: x[vec nn][4 ks + g]
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) ax[ks] = xt[nn * VQM_XLD + ks * 4 + g];
        VQ_STAMP(2);
        // the MFMAs run k-step by k-step ACROSS the tiles -- consecutive MFMAs write different accumulators (no dependent-issue
        // stalls), and every accumulator still sums its dimensions in ascending order
        f32x4 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[q * 4 + c], bq[t][q][c], acc[t], 0, 0, 0);
        if (ov >= 0 && ov < n && sd_ok) {               // the previous tile's output rows (the code rows have landed by now)
            const f32x4 ox = *reinterpret_cast<const f32x4*>(xts[tog ^ 1] + svec * VQM_XLD + (tid & 15) * 4);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (ox[c] + oe[c]) - ox[c];
            *reinterpret_cast<f32x4*>(out + (size_t)ov * D + (tid & 15) * 4) = o;
        }
        // lane (g, nn) now holds dot(x[vec 4g + r], e[code of (tile wave + NW t, column nn)]) in acc[t][r]
        float xxr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xxr[r] = xxs[wave][4 * g + r];
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dist = (xxr[r] + e2v[t]) - 2.0f * acc[t][r];     // embed.py:210-212 association order
                float sim = tscale * (-dist);
                if (!FULL) sim += off[t];
                acc[t][r] = sim;
                mx[r] = fmaxf(mx[r], sim);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { mx[r] = vq_row_max(mx[r]); if (nn == 0) red[wave][4 * g + r] = mx[r]; }
        st_lds_barrier();
        VQ_STAMP(3);
        float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = red[0][4 * g + r];
#pragma unroll
            for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w][4 * g + r]);
            mx[r] = m;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = vq_exp(acc[t][r] - mx[r]); acc[t][r] = e; sm[r] += e; }
        st_lds_barrier();                                     // everybody has read the maxima: the slots are reused for the sums
        VQ_STAMP(4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { sm[r] = vq_row_sum(sm[r]); if (nn == 0) red[wave][4 * g + r] = sm[r]; }
        st_lds_barrier();
        VQ_STAMP(5);
        // argmax over p with "first maximum wins": the largest e of a row is exp(0) = 1 exactly, so the row maximum of p is the value
        // p takes for e = 1 -- known without a reduction -- and the index is the SMALLEST code whose p equals it (an integer min:
        // one DPP min per step instead of a compare + two selects on (value, index) pairs: 2.7k of a tile's 21k cycles)
        int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
        float rs[4], pmax[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {          // fixed order: wave 0, 1, ...
            float sacc = red[0][4 * g + r];
#pragma unroll
            for (int w = 1; w < NW; ++w) sacc += red[w][4 * g + r];
            sm[r] = sacc;
            rs[r] = 1.0f / sacc;
            pmax[r] = vq_div(1.0f, sacc, rs[r]);
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int code = vq_code_of(NW, TPW, wave + NW * t, nn);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = vq_div(acc[t][r], sm[r], rs[r]);
                acc[t][r] = p;
                bi[r] = min(bi[r], p == pmax[r] ? code : 0x7fffffff);      // (codes past the table have p = 0 < pmax)
            }
        }
        const bool whole = FULL && v0 + 16 <= n && p_vec4;       // (uniform) every vector and every code of the tile exists
        VQ_STAMP(6);
        if (PERM && whole) {      // four consecutive codes per lane and vector: 16-byte stores, 256 contiguous bytes per 16 lanes
            float* pp = p_code + (size_t)(v0 + 4 * g) * V + vq_code_of(NW, TPW, wave, nn);
#pragma unroll
            for (int h = 0; h < (PERM ? TPW / 4 : 0); ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<f32x4*>(pp + (size_t)r * V + 64 * h) = f32x4{acc[4 * h][r], acc[4 * h + 1][r], acc[4 * h + 2][r], acc[4 * h + 3][r]};
        } else if (PERM) {
#pragma unroll
            for (int h = 0; h < (PERM ? TPW / 4 : 0); ++h) {
                const int c0 = vq_code_of(NW, TPW, wave + NW * 4 * h, nn);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int vec = v0 + 4 * g + r;
                    if (vec >= n || c0 >= V) continue;
                    float* pp = p_code + (size_t)vec * V + c0;
                    if (p_vec4 && c0 + 3 < V) *reinterpret_cast<f32x4*>(pp) = f32x4{acc[4 * h][r], acc[4 * h + 1][r], acc[4 * h + 2][r], acc[4 * h + 3][r]};
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (c0 + j < V) pp[j] = acc[4 * h + j][r];
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int code = vq_code_of(NW, TPW, wave + NW * t, nn);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int vec = v0 + 4 * g + r;
                    if (code < V && vec < n) p_code[(size_t)vec * V + code] = acc[t][r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bi[r] = min(bi[r], vq_dpp_i<ST_DPP_QUAD_XOR1>(bi[r])); bi[r] = min(bi[r], vq_dpp_i<ST_DPP_QUAD_XOR2>(bi[r]));
            bi[r] = min(bi[r], vq_dpp_i<ST_DPP_ROW_HALF_MIRROR>(bi[r])); bi[r] = min(bi[r], vq_dpp_i<ST_DPP_ROW_MIRROR>(bi[r]));
            if (nn == 0) redi[wave][4 * g + r] = bi[r];
        }
        VQ_STAMP(7);
        st_lds_barrier();
        VQ_STAMP(8);
        if (tid < 16) {
            int i = redi[0][tid];
#pragma unroll
            for (int w = 1; w < NW; ++w) i = min(i, redi[w][tid]);
            fidx[tid] = i;
            if (v0 + tid < n) idx_out[v0 + tid] = i;
        }
        st_lds_barrier();
        VQ_STAMP(9);
        // out = (x + code) - x.detach(), the straight-through forward value (embed.py:145): the picked code's row is REQUESTED here and
        // consumed one tile later (the gather's round trip, 2.1k cycles, sat on every tile's critical path)
        oe = *reinterpret_cast<const f32x4*>(table + (size_t)fidx[svec] * D + sd4);
        ov = v0 + svec;
        VQ_STAMP(10);
    }
    if (ov >= 0 && ov < n && sd_ok) {
        const f32x4 ox = *reinterpret_cast<const f32x4*>(xts[tog ^ 1] + svec * VQM_XLD + (tid & 15) * 4);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = (ox[c] + oe[c]) - ox[c];
        *reinterpret_cast<f32x4*>(out + (size_t)ov * D + (tid & 15) * 4) = o;
    }
}

__global__ __launch_bounds__(VQ_WAVES * 64) void softmax_argmax_kernel(const float* logits, float* p, int64_t* idx_out,
                                                                       int n, int V) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mysim = lds + wave * V;
    for (int r = blockIdx.x * VQ_WAVES + wave; r < n; r += gridDim.x * VQ_WAVES) {
        for (int v = lane; v < V; v += 64) mysim[v] = logits[(size_t)r * V + v];
        const int bi = wave_softmax_argmax(mysim, V, lane, p + (size_t)r * V);
        if (lane == 0) idx_out[r] = bi;
    }
}

// ---- backward pieces of the codebook lookup (ref: what autograd derives for src/embed.py:105-147 / :187-205) -------------------
// softmax backward of one row per wave: dz = scale * p * (dp - sum_v dp p); optionally the row sums of dz (they multiply the
// |x|^2 term of the L2 similarity: zero up to rounding, kept for fidelity)
__global__ __launch_bounds__(VQ_WAVES * 64) void softmax_bwd_kernel(const float* p, const float* dp, const float* scale_ptr, float scale,
                                                                    float* dz, float* rowsum, int n, int V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float sc = scale_ptr ? fmaxf(scale_ptr[0], 0.0f) * scale : scale;       // relu(temp) of the L2 codebook
    for (int r = blockIdx.x * VQ_WAVES + wave; r < n; r += gridDim.x * VQ_WAVES) {
        const float* pr = p + (size_t)r * V;
        const float* dr = dp + (size_t)r * V;
        float dot = 0.0f;
        for (int v = lane; v < V; v += 64) dot = fmaf(dr[v], pr[v], dot);
        dot = st_wave_sum_dpp(dot);
        float rs = 0.0f;
        for (int v = lane; v < V; v += 64) {
            const float g = sc * pr[v] * (dr[v] - dot);
            dz[(size_t)r * V + v] = g;
            rs += g;
        }
        if (rowsum) { rs = st_wave_sum_dpp(rs); if (lane == 0) rowsum[r] = rs; }
    }
}

// out[m][d] = alpha * a[m][d] + beta * x[m][d] * r[m] (+ c[m][d])     (the |x|^2 / |e|^2 terms and the straight-through addend)
__global__ __launch_bounds__(256) void rowscale_combine_kernel(const float* a, float alpha, const float* x, const float* r, float beta,
                                                               const float* c, float* out, int M, int D) {
    const size_t total = (size_t)M * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / D;
        float v = alpha * a[i];
        if (x) v = fmaf(beta * r[m], x[i], v);
        if (c) v += c[i];
        out[i] = v;
    }
}

// ---- run-length merge of the VQ codes (ref: VQVAE.mean_forward, src/vqvae.py:218-257) --------------------------
// One workgroup per utterance.  Phase 1: argmax of p_code per frame (first maximum wins).  Phase 2: thread 0 walks
// the frame indices exactly like the reference's Python loop (a new segment starts when the code changes or the
// current one already holds max_frames+1 frames; blank (0) segments are dropped) -- T integer steps in LDS instead
// of a D2H copy plus a host loop per utterance.  Phase 3: all threads average the frames of the kept segments.
__global__ __launch_bounds__(256) void vq_mean_fwd_kernel(const float* p_code, const float* latent, float* out, int* lens,
                                                          int* frame_seg, float* frame_w, int B, int T, int D, int V,
                                                          int max_frames) {
    extern __shared__ __attribute__((aligned(16))) int ilds[];
    int* idx = ilds;              // [T]
    int* seg_start = ilds + T;    // [T] first frame of kept segment n
    int* seg_len = ilds + 2 * T;  // [T]
    __shared__ int nseg_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int t = tid; t < T; t += blockDim.x) {
        const float* p = p_code + ((size_t)b * T + t) * V;
        float best = p[0];
        int bi = 0;
        for (int v = 1; v < V; ++v) { const float x = p[v]; if (x > best) { best = x; bi = v; } }
        idx[t] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0, last_idx = idx[0], last_pos = 0;
        for (int t = 0; t < T; ++t) {
            const int i = idx[t];
            if (last_idx != i || (t - last_pos) > max_frames) {                     // :231
                if (last_idx != 0) { seg_start[n] = last_pos; seg_len[n] = t - last_pos; ++n; }
                last_idx = i; last_pos = t;
            }
        }
        if (last_idx != 0) { seg_start[n] = last_pos; seg_len[n] = T - last_pos; ++n; }   // :239-245
        nseg_s = n;
        lens[b] = n;
    }
    __syncthreads();
    const int nseg = nseg_s;
    for (int t = tid; t < T; t += blockDim.x) { frame_seg[(size_t)b * T + t] = -1; frame_w[(size_t)b * T + t] = 0.0f; }
    __syncthreads();
    for (int i = tid; i < nseg * D; i += blockDim.x) {
        const int n = i / D, d = i - n * D;
        const int s0 = seg_start[n], len = seg_len[n];
        const float* lp = latent + ((size_t)b * T + s0) * D + d;
        float acc = 0.0f;
        for (int j = 0; j < len; ++j) acc += lp[(size_t)j * D];
        out[((size_t)b * T + n) * D + d] = len == 1 ? acc : acc / (float)len;
        if (d == 0) for (int j = 0; j < len; ++j) { frame_seg[(size_t)b * T + s0 + j] = n; frame_w[(size_t)b * T + s0 + j] = 1.0f / (float)len; }
    }
}

// dlatent(b, t, :) = dout(b, seg(t), :) / len(seg(t))   (zero for dropped frames)
__global__ __launch_bounds__(256) void vq_mean_bwd_kernel(const float* dout, int ld_seg, const int* frame_seg, const float* frame_w,
                                                          float* dlat, int B, int T, int D) {
    const size_t total = (size_t)B * T * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const size_t bt = i / D;
        const int b = (int)(bt / T);
        const int n = frame_seg[bt];
        dlat[i] = (n >= 0 && n < ld_seg) ? dout[((size_t)b * ld_seg + n) * D + d] * frame_w[bt] : 0.0f;
    }
}

}  // namespace

#ifdef VQ_STAMPS
extern "C" int st_vq_debug_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vq_stamps), sizeof(vq_stamps)); }
#endif

extern "C" int st_vq_build_table(const float* learnable, int Dl, const float* attr, int n_attr,
                                 const float* attr_w, const float* attr_b, int Da, float* table, int V, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(learnable && table && V > 0 && Dl > 0 && Da >= 0, "st_vq_build_table: bad arguments");
    ST_CHECK_ARG(Da == 0 || (attr && attr_w && attr_b && n_attr > 0), "st_vq_build_table: attribute pointers missing");
    const int total = V * (Dl + Da);
    hipLaunchKernelGGL(vq_build_table_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       learnable, Dl, attr, n_attr, attr_w, attr_b, Da, table, V);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_gather_rows(const float* table, const int64_t* idx, float* out, int n, int D, int V, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(table && idx && out && n > 0 && D > 0 && V > 0, "st_gather_rows: bad arguments");
    size_t total = (size_t)n * D;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, idx, out, n, D, V);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t st_vq_l2_workspace_floats(int D, int V) {
    if (D <= 0 || V <= 0) return 0;
    const size_t n_pt = (size_t)vq_nw((V + 15) / 16) * vq_tpw((V + 15) / 16), ks4n = ((size_t)D + 15) / 16;      // packed tiles incl. the padding ones
    return n_pt * ks4n * 256 + n_pt * 16;
}

// shapes the matrix-core search takes: D <= 64 in 16-byte pieces, at most 16 code tiles per wave
static inline bool vq_mfma_shape(int D, int V) { return D <= 64 && D % 4 == 0 && V <= 1024; }

extern "C" int st_vq_pack_table(const float* table, float* packed, int D, int V, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(table && packed && vq_mfma_shape(D, V) && D > 0 && V > 0 && st_aligned16(table) && st_aligned16(packed),
                 "st_vq_pack_table: D=%d V=%d not a matrix-core shape, or unaligned operands", D, V);
    const int nw = vq_nw((V + 15) / 16), tpw = vq_tpw((V + 15) / 16), n_pt = nw * tpw, ks4n = (D + 15) / 16;
    const size_t total = (size_t)n_pt * ks4n * 256 + (size_t)n_pt * 16;
    hipLaunchKernelGGL(vq_pack_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, V, D,
                       packed, n_pt, ks4n, nw, tpw);
    ST_LAUNCH_CHECK();
    return 0;
}

static int vq_l2_impl(const float* x, const float* table, const float* temp, float* p_code,
                      int64_t* idx, float* out, float* workspace, bool pack, int n, int D, int V, void* stream);

extern "C" int st_vq_l2_packed_fwd(const float* x, const float* table, const float* packed, const float* temp, float* p_code,
                                   int64_t* idx, float* out, int n, int D, int V, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(x && table && packed && temp && p_code && idx && out && n > 0 && D > 0 && V > 0, "st_vq_l2_packed_fwd: bad arguments");
    ST_CHECK_ARG(vq_mfma_shape(D, V) && st_aligned16(x) && st_aligned16(table) && st_aligned16(out) && st_aligned16(packed),
                 "st_vq_l2_packed_fwd: D=%d V=%d not a matrix-core shape, or unaligned operands", D, V);
    return vq_l2_impl(x, table, temp, p_code, idx, out, const_cast<float*>(packed), false, n, D, V, stream);
}

extern "C" int st_vq_l2_fwd(const float* x, const float* table, const float* temp, float* p_code,
                            int64_t* idx, float* out, float* workspace, int n, int D, int V, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(x && table && temp && p_code && idx && out && n > 0 && D > 0 && V > 0, "st_vq_l2_fwd: bad arguments");
    return vq_l2_impl(x, table, temp, p_code, idx, out, workspace, true, n, D, V, stream);
}

static int vq_l2_impl(const float* x, const float* table, const float* temp, float* p_code,
                      int64_t* idx, float* out, float* workspace, bool pack, int n, int D, int V, void* stream) {
    // matrix-core form: a caller-provided workspace for the packed table (packed here, or once per table version by st_vq_pack_table)
    if (workspace && vq_mfma_shape(D, V) && st_aligned16(x) && st_aligned16(table) && st_aligned16(out) &&
        st_aligned16(workspace)) {
        const int ks4n = (D + 15) / 16;
        if (pack) {
            int rc = st_vq_pack_table(table, workspace, D, V, stream);
            if (rc) return rc;
        }
        // persistent tile loop: at most WPS workgroups per compute unit (what the registers hold: 8 code tiles per wave = 128 operand
        // registers fit twice into a SIMD's file, 16 tiles once), each walking tiles blockIdx.x, + gridDim.x, ...
        const int n_tiles = (n + 15) / 16;
        const int p_vec4 = (V % 4 == 0 && st_aligned16(p_code)) ? 1 : 0;
#define VQ_LAUNCH_ONE(NW_, T, WPS, FULL_) do { static int occ = 0; \
        if (occ == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vq_l2_mfma_kernel<NW_, T, WPS, FULL_>, NW_ * 64, 0) != hipSuccess || occ < 1)) occ = WPS; \
        const int gmax = (getenv("ST_VQ_WPS") ? atoi(getenv("ST_VQ_WPS")) : occ) * st_device_cus();      /* (ST_VQ_WPS: experiments) */ \
        hipLaunchKernelGGL((vq_l2_mfma_kernel<NW_, T, WPS, FULL_>), dim3(n_tiles < gmax ? n_tiles : gmax), dim3(NW_ * 64), 0, (hipStream_t)stream, \
                           x, table, workspace, temp, p_code, idx, out, n, D, V, ks4n, n_tiles, p_vec4); } while (0)
        // (the grid: as many workgroups as the registers keep resident -- 6 per compute unit at one code tile per wave)
#define VQ_LAUNCH(NW_, T, WPS) do { if (V == 16 * NW_ * T) VQ_LAUNCH_ONE(NW_, T, WPS, true); else VQ_LAUNCH_ONE(NW_, T, WPS, false); } while (0)
        const int n_ct = (V + 15) / 16;
        if (n_ct <= 4) VQ_LAUNCH(4, 1, 2);
        else if (n_ct <= 8) VQ_LAUNCH(4, 2, 2);
        else if (n_ct <= 16) VQ_LAUNCH(4, 4, 2);
        else if (n_ct <= 32) VQ_LAUNCH(4, 8, 2);
        else VQ_LAUNCH(8, 8, 1);
#undef VQ_LAUNCH
#undef VQ_LAUNCH_ONE
        ST_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = ((size_t)D * (V + 1) + V + (size_t)VQ_WAVES * D + (size_t)VQ_WAVES * V) * sizeof(float);
    ST_CHECK_ARG(lds <= 160 * 1024, "st_vq_l2_fwd: V=%d x D=%d table needs %zu B of LDS (> 160 KiB)", V, D, lds);
    static bool configured = false;
    if (!configured) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vq_l2_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    int blocks = (n + VQ_WAVES - 1) / VQ_WAVES;
    const int cap = lds > 64 * 1024 ? 256 : 1024;   // the table is re-staged per workgroup: keep them fat
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(vq_l2_kernel, dim3(blocks), dim3(VQ_WAVES * 64), lds, (hipStream_t)stream,
                       x, table, temp, p_code, idx, out, n, D, V);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_softmax_bwd(const float* p, const float* dp, const float* relu_scale, float scale, float* dz, float* rowsum,
                              int n, int V, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(p && dp && dz && n > 0 && V > 0, "st_softmax_bwd: bad arguments");
    int blocks = (n + VQ_WAVES - 1) / VQ_WAVES;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(blocks), dim3(VQ_WAVES * 64), 0, (hipStream_t)stream, p, dp, relu_scale, scale, dz, rowsum, n, V);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_rowscale_combine(const float* a, float alpha, const float* x, const float* r, float beta, const float* c,
                                   float* out, int M, int D, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(a && out && M > 0 && D > 0 && ((x == nullptr) == (r == nullptr)), "st_rowscale_combine: bad arguments");
    size_t blocks = ((size_t)M * D + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rowscale_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, alpha, x, r, beta, c, out, M, D);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_softmax_argmax(const float* logits, float* p, int64_t* idx, int n, int V, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(logits && p && idx && n > 0 && V > 0, "st_softmax_argmax: bad arguments");
    const size_t lds = (size_t)VQ_WAVES * V * sizeof(float);
    ST_CHECK_ARG(lds <= 64 * 1024, "st_softmax_argmax: V=%d too large", V);
    int blocks = (n + VQ_WAVES - 1) / VQ_WAVES;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3(blocks), dim3(VQ_WAVES * 64), lds, (hipStream_t)stream,
                       logits, p, idx, n, V);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_vq_mean_fwd(const float* p_code, const float* latent, float* out, int* lens, int* frame_seg, float* frame_w,
                              int B, int T, int D, int V, int max_frames_per_phn, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(p_code && latent && out && lens && frame_seg && frame_w && B > 0 && T > 0 && D > 0 && V > 0 && max_frames_per_phn >= 0,
                 "st_vq_mean_fwd: bad arguments");
    const size_t lds = (size_t)3 * T * sizeof(int);
    ST_CHECK_ARG(lds <= 64 * 1024, "st_vq_mean_fwd: T=%d too long (3*T ints of LDS)", T);
    hipLaunchKernelGGL(vq_mean_fwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, p_code, latent, out, lens, frame_seg,
                       frame_w, B, T, D, V, max_frames_per_phn);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_vq_mean_bwd(const float* dout, int n_seg_rows, const int* frame_seg, const float* frame_w, float* dlatent,
                              int B, int T, int D, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dout && frame_seg && frame_w && dlatent && n_seg_rows > 0 && B > 0 && T > 0 && D > 0, "st_vq_mean_bwd: bad arguments");
    size_t blocks = ((size_t)B * T * D + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vq_mean_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, n_seg_rows, frame_seg,
                       frame_w, dlatent, B, T, D);
    ST_LAUNCH_CHECK();
    return 0;
}
