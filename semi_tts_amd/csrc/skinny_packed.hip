// skinny_packed.hip -- the weight-streaming kernels of the decode loop on PRE-PACKED operands.
//
// Measured on MI355X (tools/mb): a global load whose 16-lane quarter touches 16 different
// cache lines (the natural MFMA operand pattern: lane l holds row l&15) moves only ~18 B/clk
// per CU, while 1 KiB-contiguous wave loads move ~90-140 B/clk.  So the decode loop keeps its
// operands in HBM already in MFMA lane order:
//   * weights are packed ONCE per forward ("P16"): [row tile][k block][lane][4 floats], one
//     1 KiB block per (16 rows x 16 k); a workgroup's whole weight stream is one contiguous
//     run of 16*K*4 bytes, every wave instruction reads 1 KiB of it;
//   * activations live in the tiled layout "T16": [batch tile][k block][lane][4 floats]; the
//     kernels that PRODUCE activations (LSTM cell, attention, prenet) write them tiled, so the
//     consumers pay nothing.
// block layout (both): lane = 16*((k>>2)&3) + (r&15), component = k&3, i.e. exactly the
// v_mfma_f32_16x16x4_f32 operand registers A[i=lane&15][k=lane>>4] / B[k=lane>>4][n=lane&15]
// when a lane feeds component c of its float4 to the c-th of 4 successive MFMAs.
// The K loop is software pipelined (loads of group g+1 are in flight during the MFMAs of g).
#include "st_common.h"
#include "attention_body.h"
#include "attention_bwd_body.h"
#include <cstdlib>   // the "pre" part of the attention step runs as extra workgroups of a small linear

#ifndef PK_PROF
#define PK_PROF(n)   // phase timestamps, only defined by tools/mb/mb_pk.hip
#endif
#ifndef RT2_PROF
#define RT2_PROF(n)  // phase timestamps of the 2-D tiled LSTM cell, only defined by tools/mb/mb_rt2.hip
#endif

namespace {

constexpr int PK_MAXSEG = 3;

__host__ __device__ inline int pk_kb(int k) { return (k + 15) >> 4; }

// ------------------------------------------------------------------------------ layout kernels
struct PackArgs {
    const float* w[PK_MAXSEG]; int ldw[PK_MAXSEG]; int k[PK_MAXSEG]; int kb0[PK_MAXSEG];
    int nseg; int N; int H; int KB; int tiles; float* out;
};

// one thread per (tile, kb, lane): gathers 4 floats of one weight row
__device__ __forceinline__ void pack_weight_body(const PackArgs& a);
__global__ __launch_bounds__(256) void pack_weight_kernel(const PackArgs a) { pack_weight_body(a); }

// several matrices in one launch (blockIdx.y = matrix): the six P16 images of a decoder forward were six launches of ~10 us
constexpr int PACK_BATCH_MAX = 8;
struct PackBatch { PackArgs j[PACK_BATCH_MAX]; };
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const PackBatch b) { pack_weight_body(b.j[blockIdx.y]); }

__device__ __forceinline__ void pack_weight_body(const PackArgs& a) {
    const size_t total = (size_t)a.tiles * a.KB * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t blk = idx >> 6;
        const int kb = (int)(blk % a.KB), tile = (int)(blk / a.KB);
        const int i = lane & 15, kq = lane >> 4;
        int row;
        if (a.H > 0) row = (i & 3) * a.H + tile * 4 + (i >> 2);    // LSTM: (gate, unit) interleave
        else row = tile * 16 + i;
        int s = 0;
        while (s + 1 < a.nseg && kb >= a.kb0[s + 1]) ++s;
        const int k = (kb - a.kb0[s]) * 16 + kq * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < a.N) {
            const float* p = a.w[s] + (size_t)row * a.ldw[s] + k;
            const int rem = a.k[s] - k;
            if (rem >= 4 && st_aligned16(p)) v = st_ld4(p);          // whole pieces: one 16-byte load (four conditional scalar loads were four round trips)
            else {
                if (rem > 0) v[0] = p[0];
                if (rem > 1) v[1] = p[1];
                if (rem > 2) v[2] = p[2];
                if (rem > 3) v[3] = p[3];
            }
        }
        reinterpret_cast<f32x4*>(a.out)[idx] = v;
    }
}

// P16 image of the TRANSPOSE of a column concatenation: packed row r (segment s holds rows [r0_s, r0_s + k_s)) = column r - r0_s of
// w[s], packed k = source row.  One thread per 4 (rows) x 4 (k) micro-tile: four 16-byte loads along the source rows, a register
// transposition, four 16-byte stores to four adjacent lanes' slots -- the backward's W^T operands without a cat / transpose copy first.
struct PackTArgs {
    const float* w[PK_MAXSEG]; int ldw[PK_MAXSEG]; int r0[PK_MAXSEG + 1];
    int nseg; int N; int K; int KB; int tiles; int vec; float* out;
};

__global__ __launch_bounds__(256) void pack_weight_t_kernel(const PackTArgs a) {
    const size_t total = (size_t)a.tiles * a.KB * 16;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ig = (int)(idx & 3), kq = (int)((idx >> 2) & 3);
        const size_t blk = idx >> 4;
        const int kb = (int)(blk % a.KB), tile = (int)(blk / a.KB);
        const int row = tile * 16 + ig * 4, k = kb * 16 + kq * 4;
        f32x4 in[4];       // in[j] = source row k + j, columns row .. row + 3
#pragma unroll
        for (int j = 0; j < 4; ++j) in[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < a.N) {
            int s = 0;
            while (s + 1 < a.nseg && row >= a.r0[s + 1]) ++s;
            const bool whole = a.vec && row + 4 <= a.r0[s + 1];      // the four rows lie in one segment, 16-byte addressable
            if (whole && k + 4 <= a.K) {      // the usual piece: four independent 16-byte loads (with the per-row tests below each load sat behind a branch)
                const float* p0 = a.w[s] + (size_t)k * a.ldw[s] + (row - a.r0[s]);
#pragma unroll
                for (int j = 0; j < 4; ++j) in[j] = st_ld4(p0 + (size_t)j * a.ldw[s]);
            } else
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k + j >= a.K) continue;
                if (whole) {
                    in[j] = st_ld4(a.w[s] + (size_t)(k + j) * a.ldw[s] + (row - a.r0[s]));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = row + c;
                        if (r >= a.N) continue;
                        int sc = s;
                        while (sc + 1 < a.nseg && r >= a.r0[sc + 1]) ++sc;
                        in[j][c] = a.w[sc][(size_t)(k + j) * a.ldw[sc] + (r - a.r0[sc])];
                    }
                }
            }
        }
        f32x4* o = reinterpret_cast<f32x4*>(a.out) + blk * 64 + kq * 16 + ig * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = f32x4{in[0][c], in[1][c], in[2][c], in[3][c]};
    }
}

// natural (B, K) rows with stride ld  ->  tiled T16 (pads written as zero)
__global__ __launch_bounds__(256) void tile_rows_kernel(const float* src, int ld, float* dst, int kbs, int kb0, int B, int K) {
    const int KB = pk_kb(K), BT = (B + 15) >> 4;
    const size_t total = (size_t)BT * KB * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t blk = idx >> 6;
        const int kb = (int)(blk % KB), bt = (int)(blk / KB);
        const int b = bt * 16 + (lane & 15), k = kb * 16 + (lane >> 4) * 4;
        if (b >= B) continue;       // rows beyond B are left untouched (the destination is pre-zeroed)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* p = src + (size_t)b * ld + k;
        const int rem = K - k;
        if (rem > 0) v[0] = p[0];
        if (rem > 1) v[1] = p[1];
        if (rem > 2) v[2] = p[2];
        if (rem > 3) v[3] = p[3];
        reinterpret_cast<f32x4*>(dst)[((size_t)bt * kbs + kb0 + kb) * 64 + lane] = v;
    }
}

// the same by whole 16-byte pieces (rows of dst 16-byte addressable): a wave reads one tile as it lies (1 KB) and every lane stores its
// four columns of its row -- the element-wise form below moved 95 MB of step tapes per training step at 2 TB/s
__global__ __launch_bounds__(256) void untile_rows_vec_kernel(const float* src, int kbs, int kb0, float* dst, int ld, int B, int K) {
    const int KB = pk_kb(K), BT = (B + 15) >> 4;
    const size_t total = (size_t)BT * KB * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t blk = idx >> 6;
        const int kb = (int)(blk % KB), bt = (int)(blk / KB);
        const int b = bt * 16 + (lane & 15), k = kb * 16 + (lane >> 4) * 4;
        if (b >= B || k >= K) continue;
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[((size_t)bt * kbs + kb0 + kb) * 64 + lane];
        float* o = dst + (size_t)b * ld + k;
        if (k + 4 <= K) *reinterpret_cast<f32x4*>(o) = v;
        else { o[0] = v[0]; if (k + 1 < K) o[1] = v[1]; if (k + 2 < K) o[2] = v[2]; }
    }
}

__global__ __launch_bounds__(256) void untile_rows_kernel(const float* src, int kbs, int kb0, float* dst, int ld, int B, int K) {
    const size_t total = (size_t)B * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / K), k = (int)(idx - (size_t)b * K);
        const size_t off = (((size_t)(b >> 4) * kbs + kb0 + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (b & 15)) * 4 + (k & 3);
        dst[(size_t)b * ld + k] = src[off];
    }
}

// element offset (in floats) of (b, k) in a T16 buffer with KB k-blocks
__device__ __forceinline__ size_t t16_off(int b, int k, int KB) {
    return (((size_t)(b >> 4) * KB + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (b & 15)) * 4 + (k & 3);
}

// ------------------------------------------------------------------------------ the kernel
// One contiguous weight run (P16) times one contiguous activation run (a k-block range of a T16
// buffer): the K loop is two pointer increments and TRIP*(1+NB) 1-KiB wave loads per group.
struct PkOut { float* base; int kb_stride; int kb0; };   // a k-block range of a T16 buffer

struct PkArgs {
    const f32x4* w; int w_kbs;           // packed weights [tile][w_kbs][64], pre-offset to the first k-block used
    const f32x4* x; int x_kbs;           // activations: T16 buffer pre-offset to the first k-block; its k-block stride
    int KB;                              // k-blocks to reduce over
    int B, N, H;
    // LSTM epilogue
    const float* b_ih; const float* b_hh;
    const float* c_prev; int ldc_prev; const float* mask;
    float* c_out; int ldc; float* gates_out;
    PkOut h_dst[2];                                                        // tiled destinations of the new h
    const float* ada_std; const float* ada_mean; PkOut ha_dst;             // optional AdaIN of the new h
    // linear epilogue
    const float* bias; int act; const float* lmask; int ldmask;
    float* y; int ldy; PkOut y_dst; int n_split; float* y2; int ldy2; int rep;
    // optional third row range [n_split2, N): act2/mask2 applied, written to y3_dst (column n - n_split2)
    int n_split2; int act2; const float* mask2; int ldmask2; PkOut y3_dst;
    // optional: the plain range leaves as 8-byte {value, tag = epoch} granules (B, N) for consumer workgroups of the SAME launch
    unsigned long long* gran; unsigned epoch;
    // optional: the third range likewise, (B, N - n_split2) granules (the operand of pk_gran_linear_body workgroups of the same launch)
    unsigned long long* gran3;
    // optional (LSTM cell, 2-D tiled form): a slab (B, N) of a partial product of the SAME weight rows over the k-blocks this launch does
    // not reduce (pk_part_body of an earlier launch, st_query_attn_fin_part_fwd), added to the gates
    const float* part;
};

// MODE 2: a linear whose output columns [n0, n0 + H) are dh of an LSTM cell -- the pointwise half of the cell's backward step
// (st_lstm_cell_bwd_pointwise) runs in the epilogue of the product that makes dh, instead of as a launch of its own:
//   dh = (y + dh1 [+ dh2 * scale2]) * mask;  dc = dc_in + dh * o * (1 - tanh(c)^2);  dgates (i, f, g, o);  dc_out = dc * f
struct PkPw {
    int n0, H;
    const float* dh1; int ld1;                       // addend, (B rows, stride ld1), column u
    int dh1_slabs; long dh1_slab_stride;             // > 1: dh1 is the first of that many slabs (floats apart) of a K-split product, added in order
    const float* dh2; int ld2; const float* scale2;  // optional addend times scale2 (B, H)
    const float* mask;                               // optional (B, H)
    const float* gates;                              // (B, 4, H) activated gates of the step
    const float* c; int ldc; const float* c_prev; int ldcp;   // c_t, c_{t-1} (NULL = zeros)
    float* dc;                                       // (B, H) in / out
    float* dgates; int ldg;                          // (B, 4H) out, torch gate order
    PkOut dg_t16;                                    // out: the same in T16 (operand of the next packed product)
};

__device__ __forceinline__ void pk_store(const PkOut& d, int b, int k, float v) {
    if (d.base) d.base[t16_off(b, d.kb0 * 16 + k, d.kb_stride)] = v;
}

template <int NB, int TRIP>
struct PkRegs { f32x4 w[TRIP]; f32x4 x[TRIP][NB]; };

typedef __attribute__((ext_vector_type(4))) unsigned int pk_u32x4;

// Loads go through buffer descriptors (SRSRC): base in four scalar registers, the wave-uniform k-block offset in a scalar
// register, only the lane offset (lane * 16 B) in a vector register -- no 64-bit vector address arithmetic per load.
#ifndef PK_W_AUX
#define PK_W_AUX 0      // cache policy of the weight stream; 2 (nt) measured 12 % slower: the default policy keeps the 75 MB of weights in the 256 MB MALL across steps
#endif
struct PkSrc {
    __amdgpu_buffer_rsrc_t w, x;   // descriptors of the weight tile base / the activation base of this workgroup
    unsigned voff;                 // lane * 16
    int x_kbs;
};

// The weight descriptor ends at k-block KB (num_records = KB KiB past the tile base): a k-block past KB reads zeros by
// itself, so there is no clamp, no validity flag and no select in front of the MFMAs.
template <int NB, int KW, int TRIP>
__device__ __forceinline__ void pk_load(PkRegs<NB, TRIP>& r, const PkSrc& src, int kb, int KB) {
    (void)KB;
#pragma unroll
    for (int t = 0; t < TRIP; ++t) {
        const unsigned vo = src.voff + (unsigned)(kb + t * KW) * 1024u;
        r.w[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src.w, vo, 0, PK_W_AUX));
#pragma unroll
        for (int bt = 0; bt < NB; ++bt)
            r.x[t][bt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src.x, vo, bt * src.x_kbs * 1024, 0));
    }
}

template <int NB, int TRIP>
__device__ __forceinline__ void pk_mma(const PkRegs<NB, TRIP>& r, f32x4 (&acc)[NB]) {
#pragma unroll
    for (int t = 0; t < TRIP; ++t) {
        const f32x4 w = r.w[t];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int bt = 0; bt < NB; ++bt)
                acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cc], r.x[t][bt][cc], acc[bt], 0, 0, 0);
    }
}

// MODE 0: LSTM cell, MODE 1: linear
template <int MODE, int NB, int KW, int TRIP>
__device__ __forceinline__ void pk_body(const PkArgs& a, const int tile, const int by, f32x4* red, const PkPw* pw = nullptr, const int tid0 = 0) {
    // the wave index is wave-uniform: telling the compiler (readfirstlane) keeps every k-block address in scalar registers,
    // so a load is `global_load v, lane_offset, s[base]` instead of a 64-bit vector address computation per load
    // (tid0: first thread of this body's KW waves when two bodies share a workgroup, pk_pw_ab_dual_kernel)
    const int tid = (int)threadIdx.x - tid0, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bt0 = by * NB;
    const int BT = (a.B + 15) >> 4;

    PK_PROF(0);
    // All kernel arguments the prologue needs are pulled into scalar registers NOW, with one wait: left to itself the
    // compiler fetches the argument block piecemeal, one scalar-cache round trip (300-500 cycles) in front of each use.
#define PK_TOUCH(x) asm volatile("" :: "s"(x))
    PK_TOUCH(a.w); PK_TOUCH(a.x); PK_TOUCH(a.w_kbs); PK_TOUCH(a.x_kbs); PK_TOUCH(a.KB); PK_TOUCH(a.B); PK_TOUCH(a.N); PK_TOUCH(a.H);
    if (MODE == 0) {
        PK_TOUCH(a.b_ih); PK_TOUCH(a.b_hh); PK_TOUCH(a.c_prev); PK_TOUCH(a.ldc_prev);
        PK_TOUCH(a.mask); PK_TOUCH(a.ada_std); PK_TOUCH(a.ada_mean); PK_TOUCH(a.ha_dst.base);
    }
    if (MODE >= 1) {
        PK_TOUCH(a.bias); PK_TOUCH(a.lmask); PK_TOUCH(a.ldmask); PK_TOUCH(a.mask2); PK_TOUCH(a.ldmask2); PK_TOUCH(a.n_split2);
    }
#undef PK_TOUCH
    f32x4 acc[NB];
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) acc[bt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // batch tiles beyond BT alias the last valid ones (their results are discarded)
    const int bt_base = bt0 + NB <= BT ? bt0 : (BT >= NB ? BT - NB : 0);
    const int KB = a.KB;
    PkSrc src;
    src.w = __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (size_t)tile * a.w_kbs * 64), 0, KB * 1024, 0x00020000);
    // (measured on gfx950: the scalar offset IS part of the range check, so the activation descriptor spans all NB batch
    // tiles; a k-block past KB of an earlier tile then reads finite data of the same buffer against a zero weight block)
    src.x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)bt_base * a.x_kbs * 64), 0, ((NB - 1) * a.x_kbs + KB) * 1024, 0x00020000);
    src.voff = (unsigned)lane * 16u;
    src.x_kbs = a.x_kbs;
    constexpr int STEP = KW * TRIP;
    PK_PROF(6);
    // Double-buffered groups of TRIP k-blocks per wave (measured faster than one big group for every shape).  Every
    // workgroup walks the K axis from a different starting group (rotated by its tile index): at any moment the 256
    // workgroups then read different lines of the shared activation operand and different HBM channels, instead of all
    // requesting the same first k-blocks at kernel start.  The first two groups are requested before anything else.
    const int G = (KB + STEP - 1) / STEP;                  // groups per wave
#ifdef PK_NO_ROTATE
    const int rot = 0;
#else
    const int rot = (int)((((unsigned)tile & 255u) * (unsigned)G) >> 8);      // in [0, G), spread over the tiles, no division
#endif
    auto kb_of = [&](int gq) { int q = gq + rot; if (q >= G) q -= G; return q * STEP + wave; };
    PkRegs<NB, TRIP> ra, rb;
    if (G > 0) pk_load<NB, KW, TRIP>(ra, src, kb_of(0), KB);
    if (G > 1) pk_load<NB, KW, TRIP>(rb, src, kb_of(1), KB);
    // epilogue operands of the threads that will run the epilogue: requested now, consumed after the
    // K loop, so their latency is hidden behind the weight stream
    const int eb = (bt_base + (tid >> 6)) * 16 + (lane & 15);
    const bool e_on = tid < NB * 64 && eb < a.B && bt_base + (tid >> 6) >= bt0;
    // IMPORTANT: every load lands in its own register and nothing is combined before the K loop.  An accumulation
    // like `e += p[i]` inside `if (p)` forces the wave to WAIT for that load right there (the add sits in the branch),
    // which serialised ~12 memory round trips (2-4 us) in front of the weight stream of waves 0..NB-1.
    float e_bi[4] = {0.f, 0.f, 0.f, 0.f}, e_bh[4] = {0.f, 0.f, 0.f, 0.f};
    float e_c = 0.f, e_m = 1.f, e_s = 0.f, e_mu = 0.f;
    float l_bias[4] = {0.f, 0.f, 0.f, 0.f}, l_m1[4] = {1.f, 1.f, 1.f, 1.f}, l_m2[4] = {1.f, 1.f, 1.f, 1.f};
    // Branch-free: absent operands are read from a valid dummy address (the weight buffer) and replaced by their neutral
    // value when they are consumed, so the block is a straight line of independent loads.
    const float* dummy = reinterpret_cast<const float*>(a.w);
    auto epi_prefetch = [&]() __attribute__((always_inline)) {
    if (MODE == 0 && e_on) {
        const int u = tile * 4 + (lane >> 4);
        const float* pbi = a.b_ih ? a.b_ih + u : dummy;
        const float* pbh = a.b_hh ? a.b_hh + u : dummy;
        const int sH = a.H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e_bi[r] = pbi[a.b_ih ? r * sH : 0];
            e_bh[r] = pbh[a.b_hh ? r * sH : 0];
        }
        e_c = (a.c_prev ? a.c_prev + (size_t)eb * a.ldc_prev + u : dummy)[0];
        e_m = (a.mask ? a.mask + (size_t)eb * a.H + u : dummy)[0];
        e_s = (a.ha_dst.base ? a.ada_std + (size_t)eb * a.H + u : dummy)[0];
        e_mu = (a.ha_dst.base ? a.ada_mean + (size_t)eb * a.H + u : dummy)[0];
    }
    if (MODE >= 1 && e_on) {      // linear epilogue operands (bias, dropout masks) requested up front as well
        const int nb = tile * 16 + 4 * (lane >> 4);
        const bool has_m2 = a.mask2 && a.n_split2 > 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nb + r < a.N ? nb + r : a.N - 1;                      // clamped: rows past N are never stored
            l_bias[r] = (a.bias ? a.bias + n : dummy)[0];
            l_m1[r] = (a.lmask ? a.lmask + (size_t)eb * a.ldmask + n : dummy)[0];
            l_m2[r] = (has_m2 && n >= a.n_split2 ? a.mask2 + (size_t)eb * a.ldmask2 + (n - a.n_split2) : dummy)[0];
        }
    }
    };
    // long K loops: the address arithmetic of this request (~100 instructions on the two waves that run the epilogue) goes
    // behind the first group's MFMAs instead of in front of them -- those two waves set the pace of the K loop's barrier
    const bool epi_late = G >= 3;
    if (!epi_late) epi_prefetch();
    PK_PROF(7);
    {
        int g = 0;
        PK_PROF(1);
        bool first = true;
        while (g < G) {                                   // ra holds group g, rb group g+1
            pk_mma<NB, TRIP>(ra, acc);
            if (first) { PK_PROF(2); first = false; }
            if (g + 2 < G) pk_load<NB, KW, TRIP>(ra, src, kb_of(g + 2), KB);
            if (g == 0 && epi_late) epi_prefetch();
            ++g;
            if (g >= G) break;
            pk_mma<NB, TRIP>(rb, acc);
            if (g + 2 < G) pk_load<NB, KW, TRIP>(rb, src, kb_of(g + 2), KB);
            ++g;
        }
    }

    PK_PROF(3);
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) red[(wave * NB + bt) * 64 + lane] = acc[bt];
    __syncthreads();
    PK_PROF(4);
    if (tid >= NB * 64) return;
    const int btl = tid >> 6;
    f32x4 s = red[btl * 64 + lane];
#pragma unroll
    for (int w = 1; w < KW; ++w) {
        const f32x4 t = red[(w * NB + btl) * 64 + lane];
        s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    const int b = (bt_base + btl) * 16 + (lane & 15);
    if (b >= a.B || bt_base + btl < bt0) return;    // aliased tiles: somebody else owns these rows

    if (MODE == 0) {
        const int H = a.H;
        const int u = tile * 4 + (lane >> 4);
        float e_b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e_b[r] = (a.b_ih ? e_bi[r] : 0.0f) + (a.b_hh ? e_bh[r] : 0.0f);
        e_c = a.c_prev ? e_c : 0.0f;
        e_m = a.mask ? e_m : 1.0f;
        const float gi = st_sigmoid_fast(s[0] + e_b[0]), gf = st_sigmoid_fast(s[1] + e_b[1]);
        const float gg = st_tanh_fast(s[2] + e_b[2]), go = st_sigmoid_fast(s[3] + e_b[3]);
        const float c2 = gf * e_c + gi * gg;
        const float h2 = go * st_tanh_fast(c2) * e_m;
        a.c_out[(size_t)b * a.ldc + u] = c2;
        pk_store(a.h_dst[0], b, u, h2);
        pk_store(a.h_dst[1], b, u, h2);
        // AdaIN: relu(W_s s + b) * (h - (W_m s + b))                        ref: src/module.py:268-269
        if (a.ha_dst.base) pk_store(a.ha_dst, b, u, e_s * (h2 - e_mu));
        if (a.gates_out) {
            float* gp = a.gates_out + (size_t)b * 4 * H + u;
            gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
        }
        PK_PROF(5);
    } else {
        // a thread holds 4 consecutive outputs n0..n0+3 of one batch row: when they all belong to the plain range they leave as
        // ONE 16-byte store per destination (natural y and / or the T16 tile, where 4 consecutive k of a row are adjacent)
        const int n0 = tile * 16 + 4 * (lane >> 4);
        const int lim = a.n_split > 0 ? a.n_split : (a.n_split2 > 0 ? a.n_split2 : a.N);
        if (n0 + 3 < min(lim, a.N)) {
            f32x4 v4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = st_act(s[r] + (a.bias ? l_bias[r] : 0.0f), a.act);
                v4[r] = a.lmask ? v * l_m1[r] : v;
            }
            if (a.gran) {   // one relaxed agent-scope (write-through) 8-byte store per value: the tag travels with the data
                typedef __attribute__((address_space(1))) unsigned long long gu64;
                gu64* gp = (gu64*)(a.gran + (size_t)b * a.N + n0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __hip_atomic_store(gp + r, ((unsigned long long)a.epoch << 32) | (unsigned long long)__float_as_uint(v4[r]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (a.y) {
                float* py = a.y + (size_t)b * a.ldy + n0;
                if ((a.ldy & 3) == 0 && st_aligned16(a.y)) *reinterpret_cast<f32x4*>(py) = v4;
                else { py[0] = v4[0]; py[1] = v4[1]; py[2] = v4[2]; py[3] = v4[3]; }
            }
            if (a.y_dst.base) *reinterpret_cast<f32x4*>(a.y_dst.base + t16_off(b, a.y_dst.kb0 * 16 + n0, a.y_dst.kb_stride)) = v4;
            if (MODE == 2 && n0 >= pw->n0 && n0 + 3 < pw->n0 + pw->H) {
                // pointwise LSTM backward of the 4 hidden units u .. u+3 of batch row b (host: H, n0, every row stride multiples of 4,
                // 16-byte aligned bases); every operand is requested before the first one is used
                const PkPw& q = *pw;
                const int H = q.H, u = n0 - q.n0;
                const size_t bu = (size_t)b * H + u;
                const float* gp = q.gates + (size_t)b * 4 * H + u;
                const f32x4 gi = st_ld4(gp), gf = st_ld4(gp + H), gg = st_ld4(gp + 2 * H), go = st_ld4(gp + 3 * H);
                const f32x4 cr = st_ld4(q.c + (size_t)b * q.ldc + u);
                const f32x4 dci = st_ld4(q.dc + bu);
                const float* dummy = q.gates;
                const f32x4 l1 = st_ld4(q.dh1 ? q.dh1 + (size_t)b * q.ld1 + u : dummy);
                f32x4 l1s[3];                          // further slabs of a K-split addend (absent: the dummy address, dropped below)
#pragma unroll
                for (int sl = 0; sl < 3; ++sl) l1s[sl] = st_ld4(q.dh1 && sl + 1 < q.dh1_slabs ? q.dh1 + (size_t)(sl + 1) * q.dh1_slab_stride + (size_t)b * q.ld1 + u : dummy);
                const f32x4 l2 = st_ld4(q.dh2 ? q.dh2 + (size_t)b * q.ld2 + u : dummy);
                const f32x4 sc = st_ld4(q.scale2 ? q.scale2 + bu : dummy);
                const f32x4 mk = st_ld4(q.mask ? q.mask + bu : dummy);
                const f32x4 cp = st_ld4(q.c_prev ? q.c_prev + (size_t)b * q.ldcp + u : dummy);
                f32x4 d0, d1, d2, d3, dco;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float dh = v4[r];
                    if (q.dh1) {
                        float a1 = l1[r];
#pragma unroll
                        for (int sl = 0; sl < 3; ++sl) a1 += sl + 1 < q.dh1_slabs ? l1s[sl][r] : 0.0f;       // (slab order: fixed)
                        dh += a1;
                    }
                    if (q.dh2) dh += l2[r] * (q.scale2 ? sc[r] : 1.0f);
                    if (q.mask) dh *= mk[r];
                    const float tc = tanhf(cr[r]);
                    const float cpv = q.c_prev ? cp[r] : 0.0f;
                    const float dc = dci[r] + dh * go[r] * (1.0f - tc * tc);
                    d0[r] = dc * gg[r] * gi[r] * (1.0f - gi[r]); d1[r] = dc * cpv * gf[r] * (1.0f - gf[r]);
                    d2[r] = dc * gi[r] * (1.0f - gg[r] * gg[r]); d3[r] = dh * tc * go[r] * (1.0f - go[r]);
                    dco[r] = dc * gf[r];
                }
                float* dg = q.dgates + (size_t)b * q.ldg + u;
                *reinterpret_cast<f32x4*>(dg) = d0; *reinterpret_cast<f32x4*>(dg + H) = d1;
                *reinterpret_cast<f32x4*>(dg + 2 * H) = d2; *reinterpret_cast<f32x4*>(dg + 3 * H) = d3;
                if (q.dg_t16.base) {
                    const int k0 = q.dg_t16.kb0 * 16 + u;
                    *reinterpret_cast<f32x4*>(q.dg_t16.base + t16_off(b, k0, q.dg_t16.kb_stride)) = d0;
                    *reinterpret_cast<f32x4*>(q.dg_t16.base + t16_off(b, k0 + H, q.dg_t16.kb_stride)) = d1;
                    *reinterpret_cast<f32x4*>(q.dg_t16.base + t16_off(b, k0 + 2 * H, q.dg_t16.kb_stride)) = d2;
                    *reinterpret_cast<f32x4*>(q.dg_t16.base + t16_off(b, k0 + 3 * H, q.dg_t16.kb_stride)) = d3;
                }
                *reinterpret_cast<f32x4*>(q.dc + bu) = dco;
            }
            PK_PROF(5);
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + r;
            if (n >= a.N) continue;
            float v = s[r] + (a.bias ? l_bias[r] : 0.0f);
            v = st_act(v, a.act);
            if (a.lmask) v *= l_m1[r];
            if (a.n_split2 > 0 && n >= a.n_split2) {
                v = st_act(v, a.act2);
                if (a.mask2) v *= l_m2[r];
                pk_store(a.y3_dst, b, n - a.n_split2, v);
                if (a.gran3) {
                    typedef __attribute__((address_space(1))) unsigned long long gu64;
                    __hip_atomic_store((gu64*)(a.gran3 + (size_t)b * (a.N - a.n_split2) + (n - a.n_split2)),
                                       ((unsigned long long)a.epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (a.n_split > 0 && n >= a.n_split) {
                float* p = a.y2 + (size_t)b * a.ldy2 + (size_t)(n - a.n_split) * a.rep;
                for (int j = 0; j < a.rep; ++j) p[j] = v;
            } else {
                if (a.y) a.y[(size_t)b * a.ldy + n] = v;
                pk_store(a.y_dst, b, n, v);
            }
        }
        PK_PROF(5);
    }
}

// LSTM cell with 2-D tiling for an even number of batch tiles: a workgroup takes TWO adjacent row tiles (32 gate rows = 8 hidden
// units) and HALF of the batch tiles (grid.y picks the half), instead of one row tile and all batch tiles.  Per MFMA the
// activation traffic halves (every workgroup of the one-row-tile form re-reads ALL of x: 229 KB at K = 1792, B = 32; 655 KB at
// K = 2560, B = 64); each weight tile is read by the two batch halves, whose workgroups are gridDim.x (a multiple of 8) apart in
// launch order (= the same XCD under round-robin placement), so the second read is an L2 hit.
// AUX0 / AUX1: cache policy of the weight loads of the workgroup's first / second row tile (0 default, 2 = nt)
template <int KW, int TRIP, int NB, int AUX0 = 0, int AUX1 = 0>
__device__ __forceinline__ void pk_lstm_rt2_body(const f32x4* wp, const f32x4* xp, const int w_kbs, const int x_kbs, const int KB,
                                                 const int B, const int H, const PkArgs& a, const int bx, const int by, f32x4* red) {
    constexpr int RT = 2;
    RT2_PROF(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile0 = bx * RT, bt_base = by * NB;          // host: B in 49..64, so both batch tiles of a half exist
    __amdgpu_buffer_rsrc_t rw[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
        rw[rt] = __builtin_amdgcn_make_buffer_rsrc((void*)(wp + (size_t)(tile0 + rt) * w_kbs * 64), 0, KB * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xp + (size_t)bt_base * x_kbs * 64), 0, ((NB - 1) * x_kbs + KB) * 1024, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    constexpr int STEP = KW * TRIP;
    const int G = (KB + STEP - 1) / STEP;
    const int rot = (int)((((unsigned)bx & 255u) * (unsigned)G) >> 8);
    auto kb_of = [&](int gq) { int q = gq + rot; if (q >= G) q -= G; return q * STEP + wave; };
    struct Regs { f32x4 w[TRIP][RT]; f32x4 x[TRIP][NB]; };
    auto load = [&](Regs& r, int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TRIP; ++t) {
            const unsigned vo = voff + (unsigned)(kb + t * KW) * 1024u;         // past KB: zeros from the range check
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                r.w[t][rt] = rt == 0 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw[0], vo, 0, AUX0))
                                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw[rt], vo, 0, AUX1));
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) r.x[t][bt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, bt * x_kbs * 1024, 0));
        }
    };
    f32x4 acc[RT][NB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) acc[rt][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const Regs& r) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TRIP; ++t)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt)
                        acc[rt][bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.w[t][rt][cc], r.x[t][bt][cc], acc[rt][bt], 0, 0, 0);
    };
    Regs ra, rb;
    if (G > 0) load(ra, kb_of(0));
    if (G > 1) load(rb, kb_of(1));
    RT2_PROF(1);
    // epilogue operands of the 4 waves that run the epilogue (wave = rt * NB + bt)
    const int e_rt = (tid >> 6) / NB, e_bt = (tid >> 6) % NB;
    const int eb = (bt_base + e_bt) * 16 + (lane & 15);
    const int u = (tile0 + e_rt) * 4 + (lane >> 4);
    const bool e_on = tid < RT * NB * 64 && eb < B;
    float e_bi[4] = {0.f, 0.f, 0.f, 0.f}, e_bh[4] = {0.f, 0.f, 0.f, 0.f};
    float e_c = 0.f, e_m = 1.f, e_s = 0.f, e_mu = 0.f;
    f32x4 e_p = {0.f, 0.f, 0.f, 0.f};
    const float* dummy = reinterpret_cast<const float*>(wp);
    auto epi_prefetch = [&]() __attribute__((always_inline)) {
        if (e_on) {
            // the partial gates of the k-blocks an earlier launch reduced: (batch row, the 4 gate rows of hidden unit u) -- pk_part_body's layout
            if (a.part) e_p = st_ld4(a.part + (size_t)eb * (4 * H) + (size_t)(tile0 + e_rt) * 16 + 4 * (lane >> 4));
            const float* pbi = a.b_ih ? a.b_ih + u : dummy;
            const float* pbh = a.b_hh ? a.b_hh + u : dummy;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e_bi[r] = pbi[a.b_ih ? r * H : 0];
                e_bh[r] = pbh[a.b_hh ? r * H : 0];
            }
            e_c = (a.c_prev ? a.c_prev + (size_t)eb * a.ldc_prev + u : dummy)[0];
            e_m = (a.mask ? a.mask + (size_t)eb * H + u : dummy)[0];
            e_s = (a.ha_dst.base ? a.ada_std + (size_t)eb * H + u : dummy)[0];
            e_mu = (a.ha_dst.base ? a.ada_mean + (size_t)eb * H + u : dummy)[0];
        }
    };
    {
        int g = 0;
        while (g < G) {
            mma(ra);
            if (g == 0) RT2_PROF(2);
            if (g + 2 < G) load(ra, kb_of(g + 2));
            if (g == 0) epi_prefetch();
            ++g;
            if (g >= G) break;
            mma(rb);
            if (g + 2 < G) load(rb, kb_of(g + 2));
            ++g;
        }
    }
    RT2_PROF(3);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) red[((wave * RT + rt) * NB + bt) * 64 + lane] = acc[rt][bt];
    __syncthreads();
    RT2_PROF(4);
    if (tid >= RT * NB * 64) return;
    f32x4 s = red[(e_rt * NB + e_bt) * 64 + lane];
#pragma unroll
    for (int w = 1; w < KW; ++w) {
        const f32x4 t = red[((w * RT + e_rt) * NB + e_bt) * 64 + lane];
        s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    if (eb >= B) return;
    if (a.part) { s[0] += e_p[0]; s[1] += e_p[1]; s[2] += e_p[2]; s[3] += e_p[3]; }
    float e_b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) e_b[r] = (a.b_ih ? e_bi[r] : 0.0f) + (a.b_hh ? e_bh[r] : 0.0f);
    e_c = a.c_prev ? e_c : 0.0f;
    e_m = a.mask ? e_m : 1.0f;
    const float gi = st_sigmoid_fast(s[0] + e_b[0]), gf = st_sigmoid_fast(s[1] + e_b[1]);
    const float gg = st_tanh_fast(s[2] + e_b[2]), go = st_sigmoid_fast(s[3] + e_b[3]);
    const float c2 = gf * e_c + gi * gg;
    const float h2 = go * st_tanh_fast(c2) * e_m;
    a.c_out[(size_t)eb * a.ldc + u] = c2;
    pk_store(a.h_dst[0], eb, u, h2);
    pk_store(a.h_dst[1], eb, u, h2);
    if (a.ha_dst.base) pk_store(a.ha_dst, eb, u, e_s * (h2 - e_mu));
    if (a.gates_out) {
        float* gp = a.gates_out + (size_t)eb * 4 * H + u;
        gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
    }
    RT2_PROF(5);
}

template <int KW, int TRIP, int NB, int AUX0 = 0, int AUX1 = 0>
__global__ __launch_bounds__(KW * 64) void pk_lstm_rt2_kernel(const f32x4* wp, const f32x4* xp, const int w_kbs, const int x_kbs, const int KB,
                                                          const int B, const int H, const PkArgs rest) {
    __shared__ f32x4 red[KW * 2 * NB * 64];
    pk_lstm_rt2_body<KW, TRIP, NB, AUX0, AUX1>(wp, xp, w_kbs, x_kbs, KB, B, H, rest, blockIdx.x, blockIdx.y, red);
}

// Two independent cells in one launch (teacher-forced training: the decoder cell of step t and the query cell of step t+1 both
// only wait for the attention of step t): the first n0 workgroups run cell 0, the others cell 1; within a cell the two batch halves
// of a row-tile pair are half a cell apart in launch order, as in the single launch (same XCD: the second weight read is an L2 hit).
template <int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void pk_lstm_rt2_pair_kernel(const int n0, const PkArgs a0, const PkArgs a1) {
    __shared__ f32x4 red[KW * 2 * 1 * 64];
    const bool second = (int)blockIdx.x >= n0;
    const PkArgs& a = second ? a1 : a0;
    const int i = second ? (int)blockIdx.x - n0 : (int)blockIdx.x;
    const int half = a.H >> 3;                  // row-tile pairs of the cell (H / 4 tiles, two per workgroup)
    const int by = i >= half ? 1 : 0;
    pk_lstm_rt2_body<KW, TRIP, 1>(a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.H, a, i - by * half, by, red);
}

// The arguments the prologue needs come first and as plain scalars: with -mllvm -amdgpu-kernarg-preload-count=16 (build.py) the
// command processor hands them over in user SGPRs at wave launch, so the first loads do not wait for a scalar-cache round
// trip on the argument block (a struct passed by value is never preloaded).
template <int MODE, int NB, int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void pk_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                     const int B, const int N, const int H, const PkArgs rest) {
    __shared__ f32x4 red[KW * NB * 64];
    PkArgs a = rest;
    a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N; a.H = H;
    pk_body<MODE, NB, KW, TRIP>(a, blockIdx.x, blockIdx.y, red);
}

// linear + pointwise LSTM backward in the epilogue of the columns that are a cell's dh (pk_body MODE 2)
template <int NB, int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void pk_pw_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                        const int B, const int N, const PkArgs rest, const PkPw pw) {
    __shared__ f32x4 red[KW * NB * 64];
    PkArgs a = rest;
    a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N; a.H = 0;
    pk_body<2, NB, KW, TRIP>(a, blockIdx.x, blockIdx.y, red, &pw);
}

// two products with a pointwise LSTM backward in their epilogues in one launch: the two directions of a bidirectional layer's BPTT
// step (blockIdx.x < n0: job 0)
template <int NB, int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void pk_pw_pair_kernel(const int n0, const int tiles, const PkArgs a0, const PkPw pw0, const PkArgs a1,
                                                             const PkPw pw1) {
    __shared__ f32x4 red[KW * NB * 64];
    const bool second = (int)blockIdx.x >= n0;
    const PkArgs& a = second ? a1 : a0;
    const PkPw& pw = second ? pw1 : pw0;
    const int i = second ? (int)blockIdx.x - n0 : (int)blockIdx.x;
    const int by = i / tiles;
    pk_body<2, NB, KW, TRIP>(a, i - by * tiles, by, red, &pw);
}

// The decoder cell's backward product of step t-1 (pk_body MODE 2) with the attention-step backward of step t BESIDE it: in the BPTT
// loop dgates_d(t-1) only needs the decoder cell's own recurrence, so the two do not depend on each other.  The attention
// workgroups come FIRST (dispatched first: each holds a compute unit for ~17 us), the product's workgroups after them.
// NS > 1: NS attention workgroups per utterance (ab_body<.., NS>: each takes A / NS attention dims; what needs the sum of their partial
// results runs as ab_hist_body in the next launch, pk_pw_hist_kernel).  Part p of utterance b is workgroup p B + b: the parts of an
// utterance are a multiple of 8 apart in launch order -- the same XCD, so the memory / S tiles they both read meet in one L2.
// NS > 1 is built for TWO workgroups per compute unit (at most 128 VGPRs, a lean LDS image): the 2 B attention workgroups and the product's
// 2 N / 16 all reside at once -- with one workgroup per unit the product alone needs two rounds.
template <int NB, int KW, int TRIP, int LBLK, int NS>
__global__ __launch_bounds__(KW * 64) __attribute__((amdgpu_waves_per_eu((NS > 1 && NB == 1) ? 4 : 1, (NS > 1 && NB == 1) ? 4 : 8))) void pk_pw_ab_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                           const int B, const int N, const int tiles_a, const int n_ab,
                                                           const PkArgs rest, const PkPw pw, const AbArgs ab) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[KW * NB * 64];
    static_assert(KW * 64 == AB_THREADS, "both parts use 512-thread workgroups");
    const int i = blockIdx.x;
    if (i < n_ab) {
        if (NS == 1) ab_body<true, LBLK, 1>(ab, i, pk_dyn_lds);
        else { const int part = i / ab.B; ab_body<true, LBLK, NS>(ab, i - part * ab.B, pk_dyn_lds, part); }
        return;
    }
    PkArgs a = rest;
    a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N; a.H = 0;
    const int j = i - n_ab;
    const int by = j / tiles_a;
    pk_body<2, NB, KW, TRIP>(a, j - by * tiles_a, by, red, &pw);
}

// the argument block of an attention-backward job: its context addends dctx[0 .. n_dctx) followed by dctx_more[0 .. n_dctx_more)
static int ab_fill_job(AbArgs& t, const st_attn_bwd_job* ab) {
    const float* dctx[AB_NDCTX]; int ld[AB_NDCTX]; int n = 0;
    ST_CHECK_ARG(ab->n_dctx >= 0 && ab->n_dctx <= 3 && ab->n_dctx_more >= 0 && ab->n_dctx_more <= 3, "attention backward job: at most 3 + 3 context addends");
    for (int j = 0; j < ab->n_dctx; ++j) { dctx[n] = ab->dctx[j]; ld[n++] = ab->ld_dctx[j]; }
    for (int j = 0; j < ab->n_dctx_more; ++j) { dctx[n] = ab->dctx_more[j]; ld[n++] = ab->ld_dctx_more[j]; }
    return ab_fill(t, &ab->dpq_t16, ab->pq, ab->pm, ab->memory, ab->w_prev, ab->ld_wprev, ab->w_cum_prev, ab->w, ab->ld_w, ab->loc_conv_w,
                   ab->loc_lin_w, ab->v, dctx, ld, n, ab->dw_direct, ab->ld_dw, ab->n_dw, ab->dcum, ab->dcum_add,
                   ab->ld_dcum_add, ab->dpq, ab->dhist, ab->ds_t, ab->loc_t, ab->dloc_t, ab->hist_t, ab->dctx_t, ab->dv_t, ab->s_in,
                   ab->B, ab->L, ab->A, ab->E, ab->F, ab->K);
}

// (timing ablations only, in the SEPARATE library tools/gpu_ablate.sh builds with -DST_ABLATE: ST_EXP makes parts of the hosted BPTT
// launches return at once -- results are then garbage.  The product library has no such switch: the flag is the constant 0.)
#ifdef ST_ABLATE
static int st_exp_flag() { static int v = -1; if (v < 0) { const char* e = getenv("ST_EXP"); v = e ? atoi(e) : 0; } return v; }
#else
static inline int st_exp_flag() { return 0; }
#endif

// ---- K-split partial products ------------------------------------------------------------------------------------------------------
// y = x W^T for B <= 32 rows (two batch tiles) with TWO row tiles and BOTH batch tiles per workgroup -- every weight fragment and every
// activation fragment feeds two MFMAs, so a workgroup takes in half the bytes per output of pk_body<.., NB = 1> (whose 2 N / 16 workgroups
// each re-read a whole batch tile of x: 168 MB through the compute units for the 42 MB decoder-cell product, 15 us) -- and the reduction
// axis cut into S ranges over workgroups so that N / 32 * S of them still fill the chip.  Split s of a tile pair writes its partial tile to
// part[s] (S, B, N); a pk_sum_body job of the NEXT launch adds the S slabs in split order (fixed: bitwise reproducible) and runs the
// epilogue (the pointwise LSTM backward of the columns that are a cell's dh).
struct PkPartArgs {
    const f32x4* w; int w_kbs;      // packed weights [tile][w_kbs][64]
    const f32x4* x; int x_kbs;      // T16 activations (two batch tiles)
    int KB, S;                      // k-blocks in total, splits (KB % S == 0)
    int B, N;                       // N % 32 == 0
    float* part;                    // (S, B, N)
};

template <int KW, int TRIP>
__device__ __forceinline__ void pk_part_body(const PkPartArgs& p, const int j, f32x4* red) {
    constexpr int RT = 2, NB = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tp = p.N >> 5;
    const int sp = j / n_tp, tp = j - sp * n_tp;           // the splits of a tile pair are n_tp workgroups apart
    const int KBs = p.KB / p.S, kb_lo = sp * KBs;
    __amdgpu_buffer_rsrc_t rw[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
        rw[rt] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + ((size_t)(tp * RT + rt) * p.w_kbs + kb_lo) * 64), 0, KBs * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)kb_lo * 64), 0, ((NB - 1) * p.x_kbs + KBs) * 1024, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    constexpr int STEP = KW * TRIP;
    const int G = (KBs + STEP - 1) / STEP;
    const int rot = (int)((((unsigned)tp & 255u) * (unsigned)G) >> 8);
    auto kb_of = [&](int gq) { int q = gq + rot; if (q >= G) q -= G; return q * STEP + wave; };
    struct Regs { f32x4 w[TRIP][RT]; f32x4 x[TRIP][NB]; };
    auto load = [&](Regs& r, int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TRIP; ++t) {
            const unsigned vo = voff + (unsigned)(kb + t * KW) * 1024u;         // past the split's range: zeros from the range check
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) r.w[t][rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw[rt], vo, 0, 0));
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) r.x[t][bt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, bt * p.x_kbs * 1024, 0));
        }
    };
    f32x4 acc[RT][NB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) acc[rt][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const Regs& r) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TRIP; ++t)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int bt = 0; bt < NB; ++bt)
                        acc[rt][bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.w[t][rt][cc], r.x[t][bt][cc], acc[rt][bt], 0, 0, 0);
    };
    Regs ra, rb;
    if (G > 0) load(ra, kb_of(0));
    if (G > 1) load(rb, kb_of(1));
    {
        int g = 0;
        while (g < G) {
            mma(ra);
            if (g + 2 < G) load(ra, kb_of(g + 2));
            ++g;
            if (g >= G) break;
            mma(rb);
            if (g + 2 < G) load(rb, kb_of(g + 2));
            ++g;
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) red[((wave * RT + rt) * NB + bt) * 64 + lane] = acc[rt][bt];
    __syncthreads();
    if (tid >= RT * NB * 64) return;
    const int e_rt = (tid >> 6) / NB, e_bt = (tid >> 6) % NB;       // waves 0..3 add the eight waves' partial tiles (wave order: fixed)
    f32x4 sm = red[(e_rt * NB + e_bt) * 64 + lane];
#pragma unroll
    for (int w = 1; w < KW; ++w) {
        const f32x4 t = red[((w * RT + e_rt) * NB + e_bt) * 64 + lane];
        sm[0] += t[0]; sm[1] += t[1]; sm[2] += t[2]; sm[3] += t[3];
    }
    const int b = e_bt * 16 + (lane & 15);
    const int n0 = (tp * RT + e_rt) * 16 + 4 * (lane >> 4);       // the lane's 4 consecutive outputs of batch row b
    if (b < p.B) *reinterpret_cast<f32x4*>(p.part + ((size_t)sp * p.B + b) * p.N + n0) = sm;
}

// y(b, n0 .. n0+3) = part[0] + part[1] + ... (split order), then -- for the columns [pw.n0, pw.n0 + pw.H) -- the pointwise LSTM backward
// of pk_body's MODE 2 epilogue on them.  One thread per (batch row, 4 columns); blocks of KW * 64 threads; job block jb.
struct PkSumArgs {
    const float* part; int S;
    int B, N;
    float* y; int ldy;
    PkPw pw;                        // pw.H == 0: no cell
};

__device__ __forceinline__ void pk_sum_body(const PkSumArgs& a, const int jb, const int nthreads) {
    const int idx = jb * nthreads + (int)threadIdx.x;
    const int n4 = a.N >> 2;
    if (idx >= a.B * n4) return;
    const int b = idx / n4, n0 = (idx - b * n4) * 4;
    const PkPw& q = a.pw;
    const bool cell = q.H > 0 && n0 >= q.n0 && n0 + 3 < q.n0 + q.H;
    // every operand is requested before the first one is used (clamped / dummy addresses, selects afterwards)
    const float* pp = a.part + (size_t)b * a.N + n0;
    const size_t sstride = (size_t)a.B * a.N;
    f32x4 pv[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) pv[s_] = st_ld4(pp + (size_t)(s_ < a.S ? s_ : 0) * sstride);
    const int H = cell ? q.H : 4, u = cell ? n0 - q.n0 : 0;
    const size_t bu = cell ? (size_t)b * H + u : 0;
    const float* dummy = a.part;
    const float* gp = cell ? q.gates + (size_t)b * 4 * H + u : dummy;
    const f32x4 gi = st_ld4(gp), gf = st_ld4(cell ? gp + H : dummy), gg = st_ld4(cell ? gp + 2 * H : dummy), go = st_ld4(cell ? gp + 3 * H : dummy);
    const f32x4 cr = st_ld4(cell ? q.c + (size_t)b * q.ldc + u : dummy);
    const f32x4 dci = st_ld4(cell ? q.dc + bu : dummy);
    const f32x4 l1 = st_ld4(cell && q.dh1 ? q.dh1 + (size_t)b * q.ld1 + u : dummy);
    f32x4 l1s[3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) l1s[sl] = st_ld4(cell && q.dh1 && sl + 1 < q.dh1_slabs ? q.dh1 + (size_t)(sl + 1) * q.dh1_slab_stride + (size_t)b * q.ld1 + u : dummy);
    const f32x4 l2 = st_ld4(cell && q.dh2 ? q.dh2 + (size_t)b * q.ld2 + u : dummy);
    const f32x4 sc = st_ld4(cell && q.scale2 ? q.scale2 + bu : dummy);
    const f32x4 mk = st_ld4(cell && q.mask ? q.mask + bu : dummy);
    const f32x4 cp = st_ld4(cell && q.c_prev ? q.c_prev + (size_t)b * q.ldcp + u : dummy);
    f32x4 v4 = pv[0];
#pragma unroll
    for (int s_ = 1; s_ < 4; ++s_)
        if (s_ < a.S) { v4[0] += pv[s_][0]; v4[1] += pv[s_][1]; v4[2] += pv[s_][2]; v4[3] += pv[s_][3]; }
    for (int s_ = 4; s_ < a.S; ++s_) { const f32x4 t = st_ld4(pp + (size_t)s_ * sstride); v4[0] += t[0]; v4[1] += t[1]; v4[2] += t[2]; v4[3] += t[3]; }
    *reinterpret_cast<f32x4*>(a.y + (size_t)b * a.ldy + n0) = v4;
    if (!cell) return;
    f32x4 d0, d1, d2, d3, dco;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float dh = v4[r];
        if (q.dh1) {
            float a1 = l1[r];
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) a1 += sl + 1 < q.dh1_slabs ? l1s[sl][r] : 0.0f;
            dh += a1;
        }
        if (q.dh2) dh += l2[r] * (q.scale2 ? sc[r] : 1.0f);
        if (q.mask) dh *= mk[r];
        const float tc = tanhf(cr[r]);
        const float cpv = q.c_prev ? cp[r] : 0.0f;
        const float dc = dci[r] + dh * go[r] * (1.0f - tc * tc);
        d0[r] = dc * gg[r] * gi[r] * (1.0f - gi[r]); d1[r] = dc * cpv * gf[r] * (1.0f - gf[r]);
        d2[r] = dc * gi[r] * (1.0f - gg[r] * gg[r]); d3[r] = dh * tc * go[r] * (1.0f - go[r]);
        dco[r] = dc * gf[r];
    }
    float* dg = q.dgates + (size_t)b * q.ldg + u;
    *reinterpret_cast<f32x4*>(dg) = d0; *reinterpret_cast<f32x4*>(dg + H) = d1;
    *reinterpret_cast<f32x4*>(dg + 2 * H) = d2; *reinterpret_cast<f32x4*>(dg + 3 * H) = d3;
    if (q.dg_t16.base) {
        const int k0 = q.dg_t16.kb0 * 16 + u;
        *reinterpret_cast<f32x4*>(q.dg_t16.base + t16_off(b, k0, q.dg_t16.kb_stride)) = d0;
        *reinterpret_cast<f32x4*>(q.dg_t16.base + t16_off(b, k0 + H, q.dg_t16.kb_stride)) = d1;
        *reinterpret_cast<f32x4*>(q.dg_t16.base + t16_off(b, k0 + 2 * H, q.dg_t16.kb_stride)) = d2;
        *reinterpret_cast<f32x4*>(q.dg_t16.base + t16_off(b, k0 + 3 * H, q.dg_t16.kb_stride)) = d3;
    }
    *reinterpret_cast<f32x4*>(q.dc + bu) = dco;
}

// BPTT launch 1 in the partial form: the NS B workgroups of the split attention backward of step t, then the K-split partial product
// dgates_d(t-1) . [W_ih | W_hh] -- N / 32 * S workgroups; with S = 2 that is 64 + 160 at C2: one round, a compute unit each
template <int KW, int TRIP, int LBLK, int NS>
__global__ __launch_bounds__(KW * 64) void pk_part_ab_kernel(const PkPartArgs p, const int n_ab, const AbArgs ab, const int exp_) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[KW * 4 * 64];
    static_assert(KW * 64 >= AB_THREADS, "the attention part runs on the first 512 threads");
    const int i = blockIdx.x;
    if (exp_ == 1 && i < n_ab) return;
    if (exp_ == 2 && i >= n_ab) return;
    if (i < n_ab) {
        if (KW * 64 > AB_THREADS && (int)threadIdx.x >= AB_THREADS) return;
        const int part = i / ab.B;
        ab_body<true, LBLK, NS>(ab, i - part * ab.B, pk_dyn_lds, part);
        return;
    }
    pk_part_body<KW, TRIP>(p, i - n_ab, red);
}

// the partial product with the history part of a split attention backward beside it (BPTT launch 3 in the partial form)
template <int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void pk_part_hist_kernel(const PkPartArgs p, const int n_h, const AbHistArgs hist) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[KW * 4 * 64];
    const int i = blockIdx.x;
    if (i < n_h) { ab_hist_body(hist, i, pk_dyn_lds); return; }
    pk_part_body<KW, TRIP>(p, i - n_h, red);
}

// the partial product on its own (no attention backward beside it)
template <int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void pk_part_kernel(const PkPartArgs p) {
    __shared__ f32x4 red[KW * 4 * 64];
    pk_part_body<KW, TRIP>(p, blockIdx.x, red);
}

// The same launch with the product's TWO batch tiles of a row tile as the two halves of ONE 16-wave workgroup (B <= 32): N / 16 product
// workgroups + NS B attention workgroups (the upper half of those exits at once) <= 256 -- every workgroup has a compute unit of its own,
// in one round.  (320 + 64 eight-wave workgroups put up to three on a unit: 17.1 us per launch against 12.7 for the product alone; the two
// halves of a pair also stream the same weight tile at the same time.)
template <int KW, int TRIP, int LBLK, int NS>
__global__ __launch_bounds__(2 * KW * 64) void pk_pw_ab_dual_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                                    const int B, const int N, const int n_ab,
                                                                    const PkArgs rest, const PkPw pw, const AbArgs ab, const int exp_) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[2 * KW * 64];
    static_assert(KW * 64 == AB_THREADS, "both parts use 512-thread bodies");
    const int i = blockIdx.x;
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / (KW * 64));
    if (exp_ == 1 && i < n_ab) return;
    if (exp_ == 2 && i >= n_ab) return;
    if (i < n_ab) {
        if (half) return;
        const int part = i / ab.B;
        ab_body<true, LBLK, NS>(ab, i - part * ab.B, pk_dyn_lds, part);
        return;
    }
    PkArgs a = rest;
    a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N; a.H = 0;
    pk_body<2, 1, KW, TRIP>(a, i - n_ab, half, red + half * KW * 64, &pw, half * KW * 64);
}

// The W_q^T dpq product (+ the query cell's pointwise backward in its epilogue) of BPTT step t with what the split attention backward of
// the same step left behind (ab_hist_body: one workgroup per utterance, dispatched first) BESIDE it: the product occupies half of the
// compute units (128 workgroups), the history part a few microseconds on 32 of the others.
// n_sum > 0: n_sum more workgroups in front of the product add the slabs of the previous launch's K-split partial product (pk_sum_body)
// MODE 1: the plain product (the BPTT step's dgates_q . W launch: 224 of 256 compute units busy for 9 us -- the history part hides there)
template <int MODE, int NB, int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void pk_pw_hist_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                             const int B, const int N, const int tiles_a, const int n_h, const int n_sum,
                                                             const PkArgs rest, const PkPw pw, const AbHistArgs hist, const PkSumArgs sum, const int exp_) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[KW * NB * 64];
    const int i = blockIdx.x;
    if (exp_ == 3 && i < n_h) return;
    if (exp_ == 4 && i >= n_h && i < n_h + n_sum) return;
    if (exp_ == 5 && i >= n_h + n_sum) return;
    if (i < n_h) { ab_hist_body(hist, i, pk_dyn_lds); return; }
    if (i < n_h + n_sum) { pk_sum_body(sum, i - n_h, KW * 64); return; }
    PkArgs a = rest;
    a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N; a.H = 0;
    const int j = i - n_h - n_sum;
    const int by = j / tiles_a;
    pk_body<MODE, NB, KW, TRIP>(a, j - by * tiles_a, by, red, MODE == 2 ? &pw : nullptr);
}

// A packed linear whose x operand is PRODUCED BY OTHER WORKGROUPS OF THE SAME LAUNCH and arrives as {value, tag} granules, (B, K) row-major
// (PkArgs.gran3 of the producer): prenet layer 2 behind the proj (+) gate (+) prenet-layer-1 product of a free-running decode step.  One
// batch tile and one row tile per workgroup, KW waves; wave w takes the k-blocks w, w + KW, ...: its weight fragments are requested
// BEFORE the wait (they do not depend on the producer), then each lane polls the 4 granules of its (batch row, 4 k's) slot per k-block
// until every tag of the wave is `epoch` (at most PK_GRAN_SPINS rounds: then NaN and bit 0 of *status, like at_wait_granules), MFMAs,
// the waves' partial sums through LDS, and the plain-range epilogue of pk_body (bias, activation, mask -> y / T16 destination).
// K <= 16 KW MAXG: the weight fragments of a wave stay in registers.
constexpr int PK_GRAN_SPINS = 1 << 18;
template <int KW, int MAXG>
__device__ __forceinline__ void pk_gran_linear_body(const PkArgs& a, const unsigned long long* xg, const int ldg, const unsigned epoch,
                                                    unsigned* status, const int tile, const int bt, f32x4* red) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KB = a.KB;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (size_t)tile * a.w_kbs * 64), 0, KB * 1024, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    f32x4 w[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) w[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff + (unsigned)(wave + g * KW) * 1024u, 0, 0));   // past KB: zeros
    // epilogue operands (bias, mask) requested before the wait as well
    const int eb = bt * 16 + (lane & 15);
    const int ebc = min(eb, a.B - 1);
    const int n0 = tile * 16 + 4 * (lane >> 4);
    const float* dummy = reinterpret_cast<const float*>(a.w);
    float l_bias[4], l_m1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = min(n0 + r, a.N - 1);
        l_bias[r] = (a.bias ? a.bias + n : dummy)[0];
        l_m1[r] = (a.lmask ? a.lmask + (size_t)ebc * a.ldmask + n : dummy)[0];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bool dead = false;
    {   // ONE lane per wave watches ONE granule of the wave's first slot until it is fresh (the producers finish within a fraction of a
        // microsecond of each other): 512 lanes polling four words each for the whole product would take fabric bandwidth from it
        gu64* cp = (gu64*)(xg + (size_t)min(bt * 16, a.B - 1) * ldg + min(wave * 16, ldg - 4));
        for (int sp = 0; sp < PK_GRAN_SPINS; ++sp) {
            const unsigned long long c0 = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(__builtin_amdgcn_readfirstlane((unsigned)(c0 >> 32))) == epoch) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int kb = wave + g * KW;
        if (kb >= KB) break;                                  // (wave-uniform)
        // the lane's slot: batch row ebc, k = 16 kb + 4 (lane >> 4) .. + 3 (clamped to the last whole slot: columns past K meet zero weights)
        gu64* gp = (gu64*)(xg + (size_t)ebc * ldg + min(kb * 16 + 4 * (lane >> 4), ldg - 4));
        unsigned long long g0 = 0, g1 = 0, g2 = 0, g3 = 0;
        bool ok = false;
        const int spins = dead ? 1 : PK_GRAN_SPINS;
        for (int sp = 0; sp < spins; ++sp) {
            g0 = __hip_atomic_load(gp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g1 = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g2 = __hip_atomic_load(gp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g3 = __hip_atomic_load(gp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool mine = (unsigned)(g0 >> 32) == epoch && (unsigned)(g1 >> 32) == epoch && (unsigned)(g2 >> 32) == epoch &&
                              (unsigned)(g3 >> 32) == epoch;
            ok = __all(mine);
            if (ok) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) {
            dead = true;
            if (lane == 0 && status) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float nanv = __builtin_nanf("");
        const f32x4 x = ok ? f32x4{__uint_as_float((unsigned)g0), __uint_as_float((unsigned)g1), __uint_as_float((unsigned)g2), __uint_as_float((unsigned)g3)}
                           : f32x4{nanv, nanv, nanv, nanv};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g][cc], x[cc], acc, 0, 0, 0);
    }
    red[wave * 64 + lane] = acc;
    __syncthreads();
    if (tid >= 64) return;
    f32x4 s = red[lane];
#pragma unroll
    for (int ww = 1; ww < KW; ++ww) { const f32x4 t = red[ww * 64 + lane]; s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3]; }
    if (eb >= a.B) return;
    if (n0 + 3 < a.N) {
        f32x4 v4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = st_act(s[r] + (a.bias ? l_bias[r] : 0.0f), a.act);
            v4[r] = a.lmask ? v * l_m1[r] : v;
        }
        if (a.y) { float* py = a.y + (size_t)eb * a.ldy + n0; py[0] = v4[0]; py[1] = v4[1]; py[2] = v4[2]; py[3] = v4[3]; }
        if (a.y_dst.base) *reinterpret_cast<f32x4*>(a.y_dst.base + t16_off(eb, a.y_dst.kb0 * 16 + n0, a.y_dst.kb_stride)) = v4;
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + r;
        if (n >= a.N) continue;
        float v = st_act(s[r] + (a.bias ? l_bias[r] : 0.0f), a.act);
        if (a.lmask) v *= l_m1[r];
        if (a.y) a.y[(size_t)eb * a.ldy + n] = v;
        pk_store(a.y_dst, eb, n, v);
    }
}

// The proj (+) gate launch of decode step t with, on the compute units it leaves idle, the part of the attention of step t+1
// that only needs the attention weights of step t (location conv + W_l + processed memory -> S): one workgroup per
// utterance after the linear's workgroups.  The attention launch of step t+1 then starts from S.
constexpr int PK_P2_MAXG = 4;      // k-blocks per wave of the in-launch second linear: K <= 16 * 8 * 4 = 512

// ... and, P2: behind those, the workgroups of ONE MORE packed linear that consumes the product's third range as granules
// (pk_gran_linear_body: prenet layer 2 of the next decoder input; n_at = attention workgroups, p2_tiles row tiles per batch tile)
template <int NB, int KW, int TRIP, bool VEC, bool P2>
__global__ __launch_bounds__(KW * 64) void pk_attnpre_p2_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                                const int B, const int N, const int tiles_a, const int n_lin,
                                                                const float* at_pm, const float* at_wprev, const int at_L,
                                                                const PkArgs a_rest, const AtArgs t_rest, const int n_at, const int p2_tiles,
                                                                const PkArgs p2, unsigned* p2_status) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[KW * NB * 64];
    static_assert(KW * 64 == AT_THREADS, "both parts use 512-thread workgroups");
    const int i = blockIdx.x;
    if (i < n_lin) {
        PkArgs a = a_rest;
        a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N;
        const int by = n_lin <= 2 * tiles_a ? (i >= tiles_a ? 1 : 0) : i / tiles_a;
        pk_body<1, NB, KW, TRIP>(a, i - by * tiles_a, by, red);
    } else if (i < n_lin + n_at) {
        AtArgs t = t_rest;
        t.pm = at_pm; t.w_prev = at_wprev; t.L = at_L;
        at_body<VEC, 1>(t, i - n_lin, pk_dyn_lds);
    } else if (P2) {
        const int j = i - n_lin - n_at;
        const int bt = j / p2_tiles;
        pk_gran_linear_body<KW, PK_P2_MAXG>(p2, a_rest.gran3, p2.KB * 16, a_rest.epoch, p2_status, j - bt * p2_tiles, bt, red);
    }
}

template <int NB, int KW, int TRIP, bool VEC>
__global__ __launch_bounds__(KW * 64) void pk_attnpre_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                             const int B, const int N, const int tiles_a, const int n_lin,
                                                             const float* at_pm, const float* at_wprev, const int at_L,
                                                             const PkArgs a_rest, const AtArgs t_rest) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[KW * NB * 64];
    // (leading scalar arguments: preloaded into SGPRs at wave launch, see pk_kernel)
    PkArgs a = a_rest;
    a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N;
    AtArgs t = t_rest;
    t.pm = at_pm; t.w_prev = at_wprev; t.L = at_L;
    static_assert(KW * 64 == AT_THREADS, "both parts use 512-thread workgroups");
    const int i = blockIdx.x;
    if (i < n_lin) {
        // (tile, batch tile group) of this workgroup without an integer division when there are at most two groups
        const int by = n_lin <= 2 * tiles_a ? (i >= tiles_a ? 1 : 0) : i / tiles_a;
        pk_body<1, NB, KW, TRIP>(a, i - by * tiles_a, by, red);
    }
    else at_body<VEC, 1>(t, i - n_lin, pk_dyn_lds);
}

// ... or, behind the attention-pre workgroups, the workgroups of a K-split partial product (pk_part_body): teacher-forced training hosts the
// tail of the decoder cell's gate reduction here (its operands come from the launch before), the paired cell launch adds the slab.
template <int NB, int KW, int TRIP, bool VEC>
__global__ __launch_bounds__(KW * 64) void pk_attnpre_part_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                                  const int B, const int N, const int tiles_a, const int n_lin,
                                                                  const float* at_pm, const float* at_wprev, const int at_L,
                                                                  const PkArgs a_rest, const AtArgs t_rest, const int n_at, const PkPartArgs p) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[KW * NB * 64];
    static_assert(KW * 64 == AT_THREADS, "both parts use 512-thread workgroups");
    const int i = blockIdx.x;
    if (i < n_lin) {
        PkArgs a = a_rest;
        a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N;
        const int by = n_lin <= 2 * tiles_a ? (i >= tiles_a ? 1 : 0) : i / tiles_a;
        pk_body<1, NB, KW, TRIP>(a, i - by * tiles_a, by, red);
    } else if (i < n_lin + n_at) {
        AtArgs t = t_rest;
        t.pm = at_pm; t.w_prev = at_wprev; t.L = at_L;
        at_body<VEC, 1>(t, i - n_lin, pk_dyn_lds);
    } else pk_part_body<KW, 2>(p, i - n_lin - n_at, reinterpret_cast<f32x4*>(pk_dyn_lds));
}

// A small linear whose CONSUMER rides in the same launch: the query projection pq = W_q h_q (first n_lin workgroups) and the fin
// part of the attention step (energies, softmax, context; the workgroups after them).  The fin workgroups request everything
// that does not depend on pq at once and then wait for pq as {value, tag} granules (at_body<.., GRAN = true>): one kernel
// boundary and the fin part's exposed first-load latency less per decode step.  All workgroups must be co-resident (the
// launcher checks n_lin + B * parts against the compute units).

template <int NB, int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void pk_attnfin_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                             const int B, const int N, const int tiles_a, const int n_lin,
                                                             const PkArgs a_rest, const AtArgs t) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[KW * NB * 64];
    static_assert(KW * 64 == AT_THREADS, "both parts use 512-thread workgroups");
    const int i = blockIdx.x;
    if (i < n_lin) {
        PkArgs a = a_rest;
        a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N;
        const int by = n_lin <= 2 * tiles_a ? (i >= tiles_a ? 1 : 0) : i / tiles_a;
        pk_body<1, NB, KW, TRIP>(a, i - by * tiles_a, by, red);
    }
    else at_body<true, 2, AT_THREADS, true>(t, i - n_lin, pk_dyn_lds);
}

// ... and, beside them, a partial product of the DECODER cell's gates over the operands that are known before the attention runs
// (W_hh_d h_d(t-1) and W_ih_d[:, E:] AdaIN(h_q(t)): 2048 of the cell's 2560 reduction columns at C2) on the compute units the pq / fin
// launch leaves idle (32 + 64 of 256 workgroups at C2); the cell launch that follows reduces the context columns only and adds the slab
// (PkArgs.part).  The part workgroups use the dynamic LDS region as their reduction buffer.  Round 6, DESIGN.md section 3.1.
template <int NB, int KW, int TRIP>
__global__ __launch_bounds__(KW * 64) void pk_attnfin_part_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                                  const int B, const int N, const int tiles_a, const int n_lin,
                                                                  const PkArgs a_rest, const AtArgs t, const int n_att, const PkPartArgs p) {
    extern __shared__ __attribute__((aligned(16))) float pk_dyn_lds[];
    __shared__ f32x4 red[KW * NB * 64];
    static_assert(KW * 64 == AT_THREADS, "both parts use 512-thread workgroups");
    const int i = blockIdx.x;
    if (i < n_lin) {
        PkArgs a = a_rest;
        a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N;
        const int by = n_lin <= 2 * tiles_a ? (i >= tiles_a ? 1 : 0) : i / tiles_a;
        pk_body<1, NB, KW, TRIP>(a, i - by * tiles_a, by, red);
    }
    else if (i < n_att) at_body<true, 2, AT_THREADS, true>(t, i - n_lin, pk_dyn_lds);
    else pk_part_body<KW, 2>(p, i - n_att, reinterpret_cast<f32x4*>(pk_dyn_lds));
}

// Long texts: the query projection and the fin part over POSITION ranges (at_range_body) with the combine inside the launch: the
// linear's workgroups, then B * P range workgroups that exchange their partial results as granules.  At most 128 VGPRs
// (__launch_bounds__(512, 4)) so that two workgroups fit a compute unit: 64 + 64 * 4 = 320 workgroups at C5 must be resident at once.
template <int NB, int KW, int TRIP>
__global__ __launch_bounds__(KW * 64, 4) void pk_attnrng_kernel(const f32x4* w, const f32x4* x, const int w_kbs, const int x_kbs, const int KB,
                                                                const int B, const int N, const int tiles_a, const int n_lin,
                                                                const PkArgs a_rest, const ArArgs t) {
    __shared__ f32x4 red[KW * NB * 64];
    static_assert(KW * 64 == AT_THREADS, "both parts use 512-thread workgroups");
    const int i = blockIdx.x;
    if (i < n_lin) {
        PkArgs a = a_rest;
        a.w = w; a.x = x; a.w_kbs = w_kbs; a.x_kbs = x_kbs; a.KB = KB; a.B = B; a.N = N;
        const int by = n_lin <= 2 * tiles_a ? (i >= tiles_a ? 1 : 0) : i / tiles_a;
        pk_body<1, NB, KW, TRIP>(a, i - by * tiles_a, by, red);
    }
    else at_range_body<AT_THREADS>(t, i - n_lin);
}

#ifndef PK_TRIP_SMALL
#define PK_TRIP_SMALL 2     // k-blocks per wave and group of the one-batch-tile linears (double buffered)
#endif
int pk_fill(PkArgs& a, const float* packed_w, const st_t16_view* x, int K, const char* who);

template <int NB>
int pk_launch_attnpre(const PkArgs& a, int tiles, const AtArgs& t, hipStream_t st, const PkArgs* p2 = nullptr, unsigned* p2_status = nullptr,
                      const PkPartArgs* pp = nullptr) {
    constexpr int KW = 8, TRIP = NB == 1 ? PK_TRIP_SMALL : 2;
    const int BT = (a.B + 15) >> 4, gy = (BT + NB - 1) / NB;
    if constexpr (NB == 1) if (pp) {       // + the workgroups of a hosted partial product (their reduction buffer lives in the dynamic LDS region)
        const bool vec = (t.A % 4 == 0) && (t.F % 4 == 0) && st_aligned16(t.pm) && st_aligned16(t.loc_lin_w) && st_aligned16(t.s_buf);
        const AtLds o = at_layout(t.L, t.A, t.E, t.F, t.K, 1, at_pos_per(t.L, t.pre_parts > 1 ? t.pre_parts : 1), vec && t.F == 32 && t.A % 16 == 0);
        size_t lds = (size_t)o.total * sizeof(float);
        if (lds < sizeof(f32x4) * KW * 4 * 64) lds = sizeof(f32x4) * KW * 4 * 64;
        ST_CHECK_ARG(lds + sizeof(f32x4) * KW * NB * 64 <= 160 * 1024, "linear + attention-pre launch: L=%d needs too much LDS (use more parts)", t.L);
        auto kern = vec ? pk_attnpre_part_kernel<NB, KW, TRIP, true> : pk_attnpre_part_kernel<NB, KW, TRIP, false>;
        static size_t configured[2] = {0, 0};
        if (lds > 32 * 1024 && lds > configured[vec ? 1 : 0]) {
            ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured[vec ? 1 : 0] = lds;
        }
        const int n_at = t.B * (t.pre_parts > 1 ? t.pre_parts : 1);
        hipLaunchKernelGGL(kern, dim3(tiles * gy + n_at + (pp->N >> 5)), dim3(KW * 64), lds, st, a.w, a.x, a.w_kbs, a.x_kbs,
                           a.KB, a.B, a.N, tiles, tiles * gy, t.pm, t.w_prev, t.L, a, t, n_at, *pp);
        ST_LAUNCH_CHECK();
        return 0;
    }
    if constexpr (NB == 1) if (p2) {       // + the workgroups of the in-launch second linear (every workgroup of the launch resident at once: checked by the caller)
        const bool vec = (t.A % 4 == 0) && (t.F % 4 == 0) && st_aligned16(t.pm) && st_aligned16(t.loc_lin_w) && st_aligned16(t.s_buf);
        const AtLds o = at_layout(t.L, t.A, t.E, t.F, t.K, 1, at_pos_per(t.L, t.pre_parts > 1 ? t.pre_parts : 1), vec && t.F == 32 && t.A % 16 == 0);
        const size_t lds = (size_t)o.total * sizeof(float);
        ST_CHECK_ARG(lds + sizeof(f32x4) * KW * NB * 64 <= 160 * 1024, "linear + attention-pre launch: L=%d needs too much LDS (use more parts)", t.L);
        auto kern = vec ? pk_attnpre_p2_kernel<NB, KW, TRIP, true, true> : pk_attnpre_p2_kernel<NB, KW, TRIP, false, true>;
        static size_t configured[2] = {0, 0};
        if (lds > 48 * 1024 && lds > configured[vec ? 1 : 0]) {
            ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured[vec ? 1 : 0] = lds;
        }
        const int n_at = t.B * (t.pre_parts > 1 ? t.pre_parts : 1), p2_tiles = (p2->N + 15) / 16;
        hipLaunchKernelGGL(kern, dim3(tiles * gy + n_at + p2_tiles * BT), dim3(KW * 64), lds, st, a.w, a.x, a.w_kbs, a.x_kbs,
                           a.KB, a.B, a.N, tiles, tiles * gy, t.pm, t.w_prev, t.L, a, t, n_at, p2_tiles, *p2, p2_status);
        ST_LAUNCH_CHECK();
        return 0;
    }
    const bool vec = (t.A % 4 == 0) && (t.F % 4 == 0) && st_aligned16(t.pm) && st_aligned16(t.loc_lin_w) && st_aligned16(t.s_buf);
    const AtLds o = at_layout(t.L, t.A, t.E, t.F, t.K, 1, at_pos_per(t.L, t.pre_parts > 1 ? t.pre_parts : 1), vec && t.F == 32 && t.A % 16 == 0);
    const size_t lds = (size_t)o.total * sizeof(float);
    ST_CHECK_ARG(lds + sizeof(f32x4) * KW * NB * 64 <= 160 * 1024, "linear + attention-pre launch: L=%d needs too much LDS (use more parts)", t.L);
    auto kern = vec ? pk_attnpre_kernel<NB, KW, TRIP, true> : pk_attnpre_kernel<NB, KW, TRIP, false>;
    static size_t configured[2] = {0, 0};
    if (lds > 48 * 1024 && lds > configured[vec ? 1 : 0]) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured[vec ? 1 : 0] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(tiles * gy + t.B * (t.pre_parts > 1 ? t.pre_parts : 1)), dim3(KW * 64), lds, st, a.w, a.x, a.w_kbs, a.x_kbs,
                       a.KB, a.B, a.N, tiles, tiles * gy, t.pm, t.w_prev, t.L, a, t);
    ST_LAUNCH_CHECK();
    return 0;
}

template <int MODE, int NB>
int pk_launch(const PkArgs& a, int tiles, hipStream_t st) {
    // 8 waves x 2 k-blocks in flight, double buffered.  Measured alternatives on MI355X (us per launch in
    // the decode graph, pq / proj / prenet): 16 waves x 6 single-buffered 7.2 / 8.8 / 7.3; 8 waves x 6
    // single-buffered 8.6 / 10.6 / 5.7; this configuration 6.1 / 8.8 / 4.8.
    constexpr int KW = 8, TRIP = (MODE == 1 && NB == 1) ? PK_TRIP_SMALL : 2;
    const int BT = (a.B + 15) >> 4;
    dim3 grid(tiles, (BT + NB - 1) / NB);
    hipLaunchKernelGGL((pk_kernel<MODE, NB, KW, TRIP>), grid, dim3(KW * 64), 0, st, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.N, a.H, a);
    ST_LAUNCH_CHECK();
    return 0;
}

// waves per workgroup x k-blocks per wave and group of the 2-D tiled LSTM cell (tools/gpu_variants.sh sweeps them)
#ifndef PK_LSTM_KW
#define PK_LSTM_KW 8
#endif
#ifndef PK_LSTM_TRIP
#define PK_LSTM_TRIP 2
#endif
// shapes the 2-D tiled LSTM cell takes (two row tiles x half the batch tiles per workgroup)
inline bool pk_rt2_shape(int B, int tiles) {
    const int BT = (B + 15) >> 4;
    return (tiles & 1) == 0 && (BT == 2 || BT == 4) && B > 16 * (BT - 1);
}

template <int MODE>
int pk_dispatch(const PkArgs& a, int tiles, hipStream_t st) {
    const int BT = (a.B + 15) >> 4;
    // LSTM cell with an even number of batch tiles (B = 17..32 or 49..64): 2-D tiling, two row tiles x half the batch tiles per
    // workgroup (pk_lstm_rt2_kernel).  Measured: B = 32 9.54 -> 9.05 us per cell (3.16 -> 3.09 ms per C2 pass), B = 64 15.0 -> 13.7 us.
    if (MODE == 0 && pk_rt2_shape(a.B, tiles)) {
        constexpr int LKW = PK_LSTM_KW, LTR = PK_LSTM_TRIP;
#if defined(PK_AUX_Q0) || defined(PK_AUX_D0)      // cache-policy experiment (tools/gpu_lstm_variants.sh): long reductions (the decoder cell) vs short ones
        if (BT == 2 && a.KB > 128)
            hipLaunchKernelGGL((pk_lstm_rt2_kernel<LKW, LTR, 1, PK_AUX_D0, PK_AUX_D1>), dim3(tiles / 2, 2), dim3(LKW * 64), 0, st, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.H, a);
        else if (BT == 2)
            hipLaunchKernelGGL((pk_lstm_rt2_kernel<LKW, LTR, 1, PK_AUX_Q0, PK_AUX_Q1>), dim3(tiles / 2, 2), dim3(LKW * 64), 0, st, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.H, a);
        else
#endif
        if (BT == 2)
            hipLaunchKernelGGL((pk_lstm_rt2_kernel<LKW, LTR, 1>), dim3(tiles / 2, 2), dim3(LKW * 64), 0, st, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.H, a);
#ifdef PK_C5_NB4      // experiment (round 4, tools/gpu_c5_nb4.sh): two row tiles x ALL FOUR batch tiles per workgroup -- 0.75 operand loads per MFMA instead of 1.0, half the workgroups
        else if (getenv("ST_C5_NB4"))
            hipLaunchKernelGGL((pk_lstm_rt2_kernel<LKW, LTR, 4>), dim3(tiles / 2, 1), dim3(LKW * 64), 0, st, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.H, a);
#endif
        else
            hipLaunchKernelGGL((pk_lstm_rt2_kernel<LKW, LTR, 2>), dim3(tiles / 2, 2), dim3(LKW * 64), 0, st, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.H, a);
        ST_LAUNCH_CHECK();
        return 0;
    }
    // a small linear (few row tiles) is bound by what ONE compute unit can pull in (its weight tile + the whole activation
    // operand): one batch tile per workgroup halves the activation bytes per workgroup and doubles the workgroups
    // ... and a linear with MANY row tiles still wants every compute unit pulling (one CU takes in ~30 GB/s): the fewest batch
    // tiles per workgroup that keep the grid within 512 workgroups (the 2560-row dgates . W^T of the decoder-LSTM backward: 160
    // workgroups of two batch tiles -> 320 of one; the second reader of a weight tile is gridDim.x workgroups on: same XCD, an L2 hit)
    if (MODE == 1 && BT > 1) {
        int nb = 1;
        while (nb < 4 && tiles * ((BT + nb - 1) / nb) > 512) ++nb;
        if (nb == 1) return pk_launch<MODE, 1>(a, tiles, st);
        if (nb == 2) return pk_launch<MODE, 2>(a, tiles, st);
        if (nb == 3) return pk_launch<MODE, 3>(a, tiles, st);
        return pk_launch<MODE, 4>(a, tiles, st);
    }
    if (BT == 1) return pk_launch<MODE, 1>(a, tiles, st);
    if (BT == 2) return pk_launch<MODE, 2>(a, tiles, st);
    if (BT == 3) return pk_launch<MODE, 3>(a, tiles, st);
    return pk_launch<MODE, 4>(a, tiles, st);
}

PkOut pk_out(const st_t16_view* v) {
    PkOut o = {nullptr, 0, 0};
    if (v && v->base) { o.base = v->base; o.kb_stride = v->kb_stride; o.kb0 = v->kb0; }
    return o;
}

int pk_fill(PkArgs& a, const float* packed_w, const st_t16_view* x, int K, const char* who) {
    ST_CHECK_ARG(packed_w && x && x->base && K > 0, "%s: bad packed operands", who);
    ST_CHECK_ARG(st_aligned16(packed_w) && st_aligned16(x->base), "%s: operands must be 16-byte aligned", who);
    ST_CHECK_ARG(x->kb0 >= 0 && x->kb0 + pk_kb(K) <= x->kb_stride, "%s: k-block range [%d,%d) outside the T16 buffer (%d)",
                 who, x->kb0, x->kb0 + pk_kb(K), x->kb_stride);
    a.w = reinterpret_cast<const f32x4*>(packed_w);
    a.w_kbs = pk_kb(K);
    a.x = reinterpret_cast<const f32x4*>(x->base) + (size_t)x->kb0 * 64;
    a.x_kbs = x->kb_stride;
    a.KB = pk_kb(K);
    return 0;
}

}  // namespace


extern "C" size_t st_packed_weight_floats(const int* k, int nseg, int N, int lstm_H) {
    size_t KB = 0;
    for (int s = 0; s < nseg; ++s) KB += (size_t)pk_kb(k[s]);
    const size_t tiles = lstm_H > 0 ? (size_t)lstm_H / 4 : ((size_t)N + 15) / 16;
    return tiles * KB * 256;
}

extern "C" size_t st_t16_floats(int B, int K) { return (size_t)((B + 15) >> 4) * pk_kb(K) * 256; }

static int pack_fill(PackArgs& a, const float* const* w, const int* ldw, const int* k, int nseg, int N, int lstm_H, float* packed) {
    ST_CHECK_ARG(w && ldw && k && packed && nseg >= 1 && nseg <= PK_MAXSEG && N > 0, "st_pack_weight: bad arguments");
    ST_CHECK_ARG(lstm_H == 0 || (lstm_H % 4 == 0 && N == 4 * lstm_H), "st_pack_weight: lstm_H=%d N=%d", lstm_H, N);
    memset(&a, 0, sizeof(a));
    int kb = 0;
    for (int s = 0; s < nseg; ++s) {
        ST_CHECK_ARG(w[s] && k[s] > 0 && ldw[s] >= k[s], "st_pack_weight: segment %d invalid", s);
        a.w[s] = w[s]; a.ldw[s] = ldw[s]; a.k[s] = k[s]; a.kb0[s] = kb;
        kb += pk_kb(k[s]);
    }
    a.nseg = nseg; a.N = N; a.H = lstm_H; a.KB = kb;
    a.tiles = lstm_H > 0 ? lstm_H / 4 : (N + 15) / 16;
    a.out = packed;
    return 0;
}

// st_pack_weight for up to eight matrices in ONE launch (st_decoder_pack: the six matrices the decode loop streams)
extern "C" int st_pack_weight_batch(const st_pack_job* jobs, int n, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(jobs && n >= 1 && n <= PACK_BATCH_MAX, "st_pack_weight_batch: 1..%d jobs", PACK_BATCH_MAX);
    PackBatch b;
    size_t most = 0;
    for (int i = 0; i < n; ++i) {
        int rc = pack_fill(b.j[i], jobs[i].w, jobs[i].ldw, jobs[i].k, jobs[i].nseg, jobs[i].N, jobs[i].lstm_H, jobs[i].packed);
        if (rc) return rc;
        const size_t total = (size_t)b.j[i].tiles * b.j[i].KB * 64;
        most = total > most ? total : most;
    }
    int blocks = (int)((most + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, b);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_pack_weight(const float* const* w, const int* ldw, const int* k, int nseg, int N, int lstm_H,
                              float* packed, void* stream) {
    (void)hipGetLastError();
    PackArgs a;
    int rc = pack_fill(a, w, ldw, k, nseg, N, lstm_H, packed);
    if (rc) return rc;
    const size_t total = (size_t)a.tiles * a.KB * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_pack_weight_t(const float* const* w, const int* ldw, const int* cols, int nseg, int K, float* packed, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(w && ldw && cols && packed && nseg >= 1 && nseg <= PK_MAXSEG && K > 0, "st_pack_weight_t: bad arguments");
    PackTArgs a;
    memset(&a, 0, sizeof(a));
    int r = 0, vec = 1;
    for (int s = 0; s < nseg; ++s) {
        ST_CHECK_ARG(w[s] && cols[s] > 0 && ldw[s] >= cols[s], "st_pack_weight_t: segment %d invalid", s);
        a.w[s] = w[s]; a.ldw[s] = ldw[s]; a.r0[s] = r;
        vec = vec && st_aligned16(w[s]) && ldw[s] % 4 == 0 && r % 4 == 0;
        r += cols[s];
    }
    a.r0[nseg] = r;
    a.nseg = nseg; a.N = r; a.K = K; a.KB = pk_kb(K); a.tiles = (r + 15) / 16; a.vec = vec; a.out = packed;
    const size_t total = (size_t)a.tiles * a.KB * 16;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_weight_t_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_tile_rows(const float* src, int ld, const st_t16_view* dst, int B, int K, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(src && dst && dst->base && B > 0 && K > 0 && ld >= K, "st_tile_rows: bad arguments");
    ST_CHECK_ARG(dst->kb0 >= 0 && dst->kb0 + pk_kb(K) <= dst->kb_stride, "st_tile_rows: k-block range outside the buffer");
    const size_t total = (size_t)((B + 15) >> 4) * pk_kb(K) * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(tile_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld, dst->base, dst->kb_stride,
                       dst->kb0, B, K);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_untile_rows(const st_t16_view* src, float* dst, int ld, int B, int K, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(src && src->base && dst && B > 0 && K > 0 && ld >= K, "st_untile_rows: bad arguments");
    if (ld % 4 == 0 && st_aligned16(dst) && st_aligned16(src->base)) {
        const size_t pieces = (size_t)((B + 15) >> 4) * pk_kb(K) * 64;
        const size_t nb = (pieces + 255) / 256;
        hipLaunchKernelGGL(untile_rows_vec_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, (hipStream_t)stream, src->base,
                           src->kb_stride, src->kb0, dst, ld, B, K);
        ST_LAUNCH_CHECK();
        return 0;
    }
    const size_t total = (size_t)B * K;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(untile_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src->base, src->kb_stride,
                       src->kb0, dst, ld, B, K);
    ST_LAUNCH_CHECK();
    return 0;
}

static int lstm_cell_packed_impl(const float* packed_w, const st_t16_view* x, int K,
                                 const float* b_ih, const float* b_hh,
                                 const float* c_prev, int ldc_prev, const float* mask,
                                 const st_t16_view* h_dst0, const st_t16_view* h_dst1,
                                 float* c_out, int ldc, float* gates_out,
                                 const float* ada_std, const float* ada_mean, const st_t16_view* hadapt_dst,
                                 int B, int H, int w_kbs, const float* part, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(B > 0 && H > 0 && H % 4 == 0 && c_out && h_dst0 && h_dst0->base, "st_lstm_cell_packed_fwd: bad arguments");
    PkArgs a;
    memset(&a, 0, sizeof(a));
    int rc = pk_fill(a, packed_w, x, K, "st_lstm_cell_packed_fwd");
    if (rc) return rc;
    if (part) {      // K covers the leading k-blocks only; the products over the others arrive as a slab (2-D tiled form, B = 17..32)
        ST_CHECK_ARG(pk_rt2_shape(B, H / 4) && ((B + 15) >> 4) == 2 && w_kbs >= a.KB && st_aligned16(part),
                     "st_lstm_cell_packed_part_fwd: needs B = 17..32, an even number of row tiles, w_kbs >= K / 16 and a 16-byte aligned slab");
        a.part = part; a.w_kbs = w_kbs;
    }
    a.B = B; a.N = 4 * H; a.H = H;
    a.b_ih = b_ih; a.b_hh = b_hh;
    a.c_prev = c_prev; a.ldc_prev = ldc_prev; a.mask = mask;
    a.c_out = c_out; a.ldc = ldc; a.gates_out = gates_out;
    a.h_dst[0] = pk_out(h_dst0); a.h_dst[1] = pk_out(h_dst1);
    a.ada_std = ada_std; a.ada_mean = ada_mean; a.ha_dst = pk_out(hadapt_dst);
    return pk_dispatch<0>(a, H / 4, (hipStream_t)stream);
}

extern "C" int st_lstm_cell_packed_fwd(const float* packed_w, const st_t16_view* x, int K,
                                       const float* b_ih, const float* b_hh,
                                       const float* c_prev, int ldc_prev, const float* mask,
                                       const st_t16_view* h_dst0, const st_t16_view* h_dst1,
                                       float* c_out, int ldc, float* gates_out,
                                       const float* ada_std, const float* ada_mean, const st_t16_view* hadapt_dst,
                                       int B, int H, void* stream) {
    return lstm_cell_packed_impl(packed_w, x, K, b_ih, b_hh, c_prev, ldc_prev, mask, h_dst0, h_dst1, c_out, ldc, gates_out, ada_std, ada_mean,
                                 hadapt_dst, B, H, 0, nullptr, stream);
}

extern "C" int st_lstm_cell_packed_part_fwd(const float* packed_w, int w_kbs, const st_t16_view* x, int K, const float* part,
                                            const float* b_ih, const float* b_hh,
                                            const float* c_prev, int ldc_prev, const float* mask,
                                            const st_t16_view* h_dst0, const st_t16_view* h_dst1,
                                            float* c_out, int ldc, float* gates_out, int B, int H, void* stream) {
    ST_CHECK_ARG(part, "st_lstm_cell_packed_part_fwd: null slab");
    return lstm_cell_packed_impl(packed_w, x, K, b_ih, b_hh, c_prev, ldc_prev, mask, h_dst0, h_dst1, c_out, ldc, gates_out, nullptr, nullptr,
                                 nullptr, B, H, w_kbs, part, stream);
}

static int pk_lstm_fill(PkArgs& a, const st_lstm_cell_packed_job* j, const char* who) {
    ST_CHECK_ARG(j && j->B > 0 && j->H > 0 && j->H % 4 == 0 && j->c_out && j->h_dst0.base, "%s: bad arguments", who);
    memset(&a, 0, sizeof(a));
    int rc = pk_fill(a, j->packed_w, &j->x, j->K, who);
    if (rc) return rc;
    a.B = j->B; a.N = 4 * j->H; a.H = j->H;
    a.b_ih = j->b_ih; a.b_hh = j->b_hh;
    a.c_prev = j->c_prev; a.ldc_prev = j->ldc_prev; a.mask = j->mask;
    a.c_out = j->c_out; a.ldc = j->ldc; a.gates_out = j->gates_out;
    a.h_dst[0] = pk_out(&j->h_dst0); a.h_dst[1] = pk_out(&j->h_dst1);
    a.ada_std = j->ada_std; a.ada_mean = j->ada_mean; a.ha_dst = pk_out(&j->hadapt_dst);
    if (j->part) {       // (as lstm_cell_packed_impl: K covers the leading k-blocks, the others arrive as a slab)
        ST_CHECK_ARG(pk_rt2_shape(j->B, j->H / 4) && ((j->B + 15) >> 4) == 2 && j->w_kbs >= a.KB && st_aligned16(j->part),
                     "%s: a slab needs B = 17..32, an even number of row tiles, w_kbs >= K / 16 and 16-byte alignment", who);
        a.part = j->part; a.w_kbs = j->w_kbs;
    }
    return 0;
}

// two independent LSTM cells (the arguments of st_lstm_cell_packed_fwd as structs): one launch when both take the 2-D tiled kernel
// with one batch tile per workgroup (B = 17..32), otherwise one launch each
extern "C" int st_lstm_cell_packed_pair_fwd(const st_lstm_cell_packed_job* j0, const st_lstm_cell_packed_job* j1, void* stream) {
    (void)hipGetLastError();
    PkArgs a0, a1;
    int rc = pk_lstm_fill(a0, j0, "st_lstm_cell_packed_pair_fwd");
    if (rc) return rc;
    rc = pk_lstm_fill(a1, j1, "st_lstm_cell_packed_pair_fwd");
    if (rc) return rc;
    const int t0 = a0.H / 4, t1 = a1.H / 4;
    const bool pair = pk_rt2_shape(a0.B, t0) && pk_rt2_shape(a1.B, t1) && ((a0.B + 15) >> 4) == 2 && ((a1.B + 15) >> 4) == 2 &&
                      a0.H % 8 == 0 && a1.H % 8 == 0;
    if (!pair) {
        rc = pk_dispatch<0>(a0, t0, (hipStream_t)stream);
        if (rc) return rc;
        return pk_dispatch<0>(a1, t1, (hipStream_t)stream);
    }
    hipLaunchKernelGGL((pk_lstm_rt2_pair_kernel<PK_LSTM_KW, PK_LSTM_TRIP>), dim3(t0 + t1), dim3(PK_LSTM_KW * 64), 0, (hipStream_t)stream, t0, a0, a1);
    ST_LAUNCH_CHECK();
    return 0;
}

static int pk_linear_impl(const float* packed_w, const st_t16_view* x, int K,
                          const float* bias, int act, const float* mask, int ldmask,
                          float* y, int ldy, const st_t16_view* y_dst,
                          int n_split, float* y2, int ldy2, int rep,
                          int n_split2, int act2, const float* mask2, int ldmask2,
                          const st_t16_view* y3_dst,
                          int B, int N, const st_attn_pre_job* pre, void* stream) {
    ST_CHECK_ARG(n_split2 <= 0 || (y3_dst && y3_dst->base && n_split2 >= n_split), "st_skinny_linear_packed_fwd: third range");
    ST_CHECK_ARG(B > 0 && N > 0 && (y || (y_dst && y_dst->base)), "st_skinny_linear_packed_fwd: bad arguments");
    ST_CHECK_ARG(n_split <= 0 || (y2 && rep >= 1), "st_skinny_linear_packed_fwd: n_split without y2/rep");
    PkArgs a;
    memset(&a, 0, sizeof(a));
    int rc = pk_fill(a, packed_w, x, K, "st_skinny_linear_packed_fwd");
    if (rc) return rc;
    a.B = B; a.N = N; a.H = 0;
    a.bias = bias; a.act = act; a.lmask = mask; a.ldmask = ldmask;
    a.y = y; a.ldy = ldy; a.y_dst = pk_out(y_dst);
    a.n_split = n_split; a.y2 = y2; a.ldy2 = ldy2; a.rep = rep;
    a.n_split2 = n_split2; a.act2 = act2; a.mask2 = mask2; a.ldmask2 = ldmask2; a.y3_dst = pk_out(y3_dst);
    const int tiles = (N + 15) / 16;
    if (pre && pre->s_buf) {
        ST_CHECK_ARG(pre->pm && pre->w_prev && pre->w_cum_prev && pre->loc_conv_w && pre->loc_lin_w && pre->L > 0 && pre->A > 0 &&
                     pre->F > 0 && pre->K > 0 && (pre->K & 1), "st_skinny_linear_packed_attnpre_fwd: bad attention job");
        AtArgs t;
        memset(&t, 0, sizeof(t));
        t.pm = pre->pm; t.w_prev = pre->w_prev; t.ld_wprev = pre->ld_wprev; t.w_cum_prev = pre->w_cum_prev;
        t.loc_conv_w = pre->loc_conv_w; t.loc_lin_w = pre->loc_lin_w; t.s_buf = pre->s_buf; t.cf_out = pre->cf_out;
        t.B = B; t.L = pre->L; t.A = pre->A; t.E = 4; t.F = pre->F; t.K = pre->K;
        t.pre_parts = (pre->parts >= 2 && pre->parts <= 64 && (pre->parts & (pre->parts - 1)) == 0) ? pre->parts : 1;
        const int BT = (B + 15) >> 4;
        if (pre->p2_packed_w) {
            // prenet layer 2 (or any linear over the third range) inside this launch: its operand travels as granules
            ST_CHECK_ARG(!pre->part, "st_skinny_linear_packed_attnpre_fwd: p2 and a hosted partial product exclude each other");
            ST_CHECK_ARG(n_split2 > 0 && pre->p2_gran && pre->p2_epoch != 0 && pre->p2_K == N - n_split2 && pre->p2_K % 16 == 0 &&
                         pre->p2_K <= 16 * 8 * PK_P2_MAXG && pre->p2_N > 0 && pre->p2_dst.base && (reinterpret_cast<uintptr_t>(pre->p2_gran) & 7) == 0,
                         "st_skinny_linear_packed_attnpre_fwd: bad in-launch linear (K = third range, a multiple of 16, at most %d)", 16 * 8 * PK_P2_MAXG);
            PkArgs p2;
            memset(&p2, 0, sizeof(p2));
            p2.w = reinterpret_cast<const f32x4*>(pre->p2_packed_w);
            p2.KB = pre->p2_K / 16; p2.w_kbs = p2.KB;
            p2.B = B; p2.N = pre->p2_N; p2.act = pre->p2_act; p2.lmask = pre->p2_mask; p2.ldmask = pre->p2_ldmask;
            p2.y_dst = pk_out(&pre->p2_dst);
            a.gran3 = pre->p2_gran; a.epoch = pre->p2_epoch;
            const int gy1 = BT;       // (NB = 1 below)
            const int n_at = B * t.pre_parts, n_p2 = ((pre->p2_N + 15) / 16) * BT;
            ST_CHECK_ARG(tiles * gy1 + n_at + n_p2 <= st_device_cus(), "st_skinny_linear_packed_attnpre_fwd: %d + %d + %d workgroups do not fit the device at once",
                         tiles * gy1, n_at, n_p2);
            return pk_launch_attnpre<1>(a, tiles, t, (hipStream_t)stream, &p2, pre->p2_status);
        }
        if (pre->part) {
            const st_partial_product_job* pj = pre->part;
            ST_CHECK_ARG(pj->packed_w && pj->x.base && pj->part && pj->KB > 0 && pj->kb0 >= 0 && pj->kb0 + pj->KB <= pj->w_kbs &&
                         pj->x.kb0 + pj->KB <= pj->x.kb_stride && B > 16 && B <= 32 && pj->N > 0 && pj->N % 32 == 0 && st_aligned16(pj->packed_w) &&
                         st_aligned16(pj->x.base) && st_aligned16(pj->part) && tiles <= 128,
                         "st_skinny_linear_packed_attnpre_fwd: bad partial product (B = 17..32, N %% 32 == 0, a linear of at most 128 row tiles)");
            PkPartArgs pp;
            memset(&pp, 0, sizeof(pp));
            pp.w = reinterpret_cast<const f32x4*>(pj->packed_w) + (size_t)pj->kb0 * 64; pp.w_kbs = pj->w_kbs;
            pp.x = reinterpret_cast<const f32x4*>(pj->x.base) + (size_t)pj->x.kb0 * 64; pp.x_kbs = pj->x.kb_stride;
            pp.KB = pj->KB; pp.S = 1; pp.B = B; pp.N = pj->N; pp.part = pj->part;
            return pk_launch_attnpre<1>(a, tiles, t, (hipStream_t)stream, nullptr, nullptr, &pp);
        }
        if (BT == 1 || tiles <= 128) return pk_launch_attnpre<1>(a, tiles, t, (hipStream_t)stream);
        if (BT == 2) return pk_launch_attnpre<2>(a, tiles, t, (hipStream_t)stream);
        if (BT == 3) return pk_launch_attnpre<3>(a, tiles, t, (hipStream_t)stream);
        return pk_launch_attnpre<4>(a, tiles, t, (hipStream_t)stream);
    }
    return pk_dispatch<1>(a, tiles, (hipStream_t)stream);
}

static int query_attn_fin_impl(const float* packed_wq, const st_t16_view* h_q, int Q, unsigned long long* granules, unsigned epoch,
                               const st_attn_fin_job* job, int B, const st_partial_product_job* pj, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(packed_wq && h_q && h_q->base && granules && epoch != 0 && job && B > 0, "st_query_attn_fin_fwd: bad arguments");
    const int L = job->L, A = job->A, E = job->E, parts = job->parts;
    ST_CHECK_ARG(L > 0 && E > 0 && A > 0 && A % 16 == 0 && A <= 256, "st_query_attn_fin_fwd: attention dim %d must be a multiple of 16, at most 256", A);
    ST_CHECK_ARG(parts == 1 || ((parts == 2 || parts == 4 || parts == 8) && E % (4 * parts) == 0), "st_query_attn_fin_fwd: parts=%d, E=%d", parts, E);
    ST_CHECK_ARG(job->s_buf && job->memory && job->w_cum_prev && job->w_out && job->w_cum_out && job->v, "st_query_attn_fin_fwd: null attention operand");
    ST_CHECK_ARG(job->n_ctx_dst >= 0 && job->n_ctx_dst <= 3, "st_query_attn_fin_fwd: n_ctx_dst=%d", job->n_ctx_dst);
    ST_CHECK_ARG(st_aligned16(job->s_buf) && st_aligned16(job->memory) && st_aligned16(job->v) && (reinterpret_cast<uintptr_t>(granules) & 7) == 0 &&
                 E % 4 == 0 && E / 4 <= AT_THREADS, "st_query_attn_fin_fwd: operands must be 16-byte aligned, E a multiple of 4");
    PkArgs a;
    memset(&a, 0, sizeof(a));
    int rc = pk_fill(a, packed_wq, h_q, Q, "st_query_attn_fin_fwd");
    if (rc) return rc;
    a.B = B; a.N = A; a.H = 0; a.act = ST_ACT_NONE;
    a.gran = granules; a.epoch = epoch;
    AtArgs t;
    memset(&t, 0, sizeof(t));
    t.pm = job->s_buf; t.s_buf = const_cast<float*>(job->s_buf); t.memory = job->memory; t.w_cum_prev = job->w_cum_prev;
    t.w_out = job->w_out; t.ld_wout = job->ld_wout; t.w_cum_out = job->w_cum_out; t.v = job->v; t.loc_lin_w = job->s_buf;
    t.fin_parts = parts;
    for (int d = 0; d < job->n_ctx_dst; ++d) t.ctx_dst[d] = job->ctx_dst[d];
    t.B = B; t.L = L; t.A = A; t.E = E; t.F = job->F; t.K = job->K;
    t.pq_gran = granules; t.epoch = epoch; t.status = job->status;
    const int tiles = A / 16, BT = (B + 15) >> 4;
    const int n_lin = tiles * BT, n_fin = B * parts;
    // the waiting workgroups hold their compute units: everything must be resident at once
    ST_CHECK_ARG(n_lin + n_fin <= st_device_cus(), "st_query_attn_fin_fwd: %d + %d workgroups do not fit the device at once (use the two launches)",
                 n_lin, n_fin);
    const AtLds o = at_layout(L, A, E, job->F, job->K, 2, L, false);
    size_t lds = (size_t)o.total * sizeof(float);
    constexpr int KW = 8;
    ST_CHECK_ARG(lds + sizeof(f32x4) * KW * 64 <= 160 * 1024, "st_query_attn_fin_fwd: L=%d needs too much LDS", L);
    if (pj) {       // + the workgroups of the hosted partial product, one compute unit each like everything else in this launch
        ST_CHECK_ARG(pj->packed_w && pj->x.base && pj->part && pj->KB > 0 && pj->kb0 >= 0 && pj->kb0 + pj->KB <= pj->w_kbs &&
                     pj->x.kb0 + pj->KB <= pj->x.kb_stride && B > 16 && B <= 32 && pj->N > 0 && pj->N % 32 == 0 && st_aligned16(pj->packed_w) &&
                     st_aligned16(pj->x.base) && st_aligned16(pj->part), "st_query_attn_fin_part_fwd: bad partial product (B = 17..32, N %% 32 == 0)");
        const int n_part = pj->N >> 5;
        ST_CHECK_ARG(n_lin + n_fin + n_part <= st_device_cus(), "st_query_attn_fin_part_fwd: %d + %d + %d workgroups do not fit the device at once",
                     n_lin, n_fin, n_part);
        PkPartArgs pp;
        memset(&pp, 0, sizeof(pp));
        pp.w = reinterpret_cast<const f32x4*>(pj->packed_w) + (size_t)pj->kb0 * 64; pp.w_kbs = pj->w_kbs;
        pp.x = reinterpret_cast<const f32x4*>(pj->x.base) + (size_t)pj->x.kb0 * 64; pp.x_kbs = pj->x.kb_stride;
        pp.KB = pj->KB; pp.S = 1; pp.B = B; pp.N = pj->N; pp.part = pj->part;
        if (lds < sizeof(f32x4) * KW * 4 * 64) lds = sizeof(f32x4) * KW * 4 * 64;        // (the part workgroups' reduction buffer lives in the dynamic region)
        auto kernp = pk_attnfin_part_kernel<1, KW, PK_TRIP_SMALL>;
        static size_t configured_p = 0;
        if (lds > 48 * 1024 && lds > configured_p) {
            ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured_p = lds;
        }
        hipLaunchKernelGGL(kernp, dim3(n_lin + n_fin + n_part), dim3(KW * 64), lds, (hipStream_t)stream, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.N,
                           tiles, n_lin, a, t, n_lin + n_fin, pp);
        ST_LAUNCH_CHECK();
        return 0;
    }
    auto kern = pk_attnfin_kernel<1, KW, PK_TRIP_SMALL>;
    static size_t configured = 0;
    if (lds > 48 * 1024 && lds > configured) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    hipLaunchKernelGGL(kern, dim3(n_lin + n_fin), dim3(KW * 64), lds, (hipStream_t)stream, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.N,
                       tiles, n_lin, a, t);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_query_attn_fin_fwd(const float* packed_wq, const st_t16_view* h_q, int Q, unsigned long long* granules, unsigned epoch,
                                     const st_attn_fin_job* job, int B, void* stream) {
    return query_attn_fin_impl(packed_wq, h_q, Q, granules, epoch, job, B, nullptr, stream);
}

extern "C" int st_query_attn_fin_part_fwd(const float* packed_wq, const st_t16_view* h_q, int Q, unsigned long long* granules, unsigned epoch,
                                          const st_attn_fin_job* job, int B, const st_partial_product_job* part, void* stream) {
    ST_CHECK_ARG(part, "st_query_attn_fin_part_fwd: null partial product job");
    return query_attn_fin_impl(packed_wq, h_q, Q, granules, epoch, job, B, part, stream);
}

// the partial product as a launch of its own (the decode loop's two-launch pq / fin form: same arithmetic, one launch more)
extern "C" int st_partial_product_fwd(const st_partial_product_job* pj, int B, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(pj && pj->packed_w && pj->x.base && pj->part && pj->KB > 0 && pj->kb0 >= 0 && pj->kb0 + pj->KB <= pj->w_kbs &&
                 pj->x.kb0 + pj->KB <= pj->x.kb_stride && B > 16 && B <= 32 && pj->N > 0 && pj->N % 32 == 0 && st_aligned16(pj->packed_w) &&
                 st_aligned16(pj->x.base) && st_aligned16(pj->part), "st_partial_product_fwd: bad job (B = 17..32, N %% 32 == 0)");
    PkPartArgs pp;
    memset(&pp, 0, sizeof(pp));
    pp.w = reinterpret_cast<const f32x4*>(pj->packed_w) + (size_t)pj->kb0 * 64; pp.w_kbs = pj->w_kbs;
    pp.x = reinterpret_cast<const f32x4*>(pj->x.base) + (size_t)pj->x.kb0 * 64; pp.x_kbs = pj->x.kb_stride;
    pp.KB = pj->KB; pp.S = 1; pp.B = B; pp.N = pj->N; pp.part = pj->part;
    hipLaunchKernelGGL((pk_part_kernel<8, 2>), dim3(pj->N >> 5), dim3(8 * 64), 0, (hipStream_t)stream, pp);
    ST_LAUNCH_CHECK();
    return 0;
}

// workgroups of pk_attnrng_kernel the device holds at once (occupancy API x compute units; 0 on error)
static int pk_attnrng_capacity() {
    static int cap = -1;
    if (cap < 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(pk_attnrng_kernel<1, 8, PK_TRIP_SMALL>), 8 * 64, 0) != hipSuccess)
            per_cu = 0;
        (void)hipGetLastError();
        cap = per_cu * st_device_cus();
    }
    return cap;
}

extern "C" int st_query_attn_rng_fits(int B, int A, int parts) {
    if (B <= 0 || A <= 0 || A % 16 != 0 || parts < 2 || parts > 8) return 0;
    return (A / 16) * ((B + 15) / 16) + B * parts <= pk_attnrng_capacity() ? 1 : 0;
}

extern "C" size_t st_attn_rng_xchg_words(int B, int E, int parts) { return (size_t)B * parts * (E + 4); }

extern "C" int st_query_attn_rng_fwd(const float* packed_wq, const st_t16_view* h_q, int Q, unsigned long long* granules,
                                     unsigned long long* xchg, unsigned epoch, const st_attn_fin_job* job, int B, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(packed_wq && h_q && h_q->base && granules && xchg && epoch != 0 && job && B > 0, "st_query_attn_rng_fwd: bad arguments");
    const int L = job->L, A = job->A, E = job->E, parts = job->parts;
    ST_CHECK_ARG(L > 0 && A > 0 && A % 16 == 0 && A <= 256, "st_query_attn_rng_fwd: attention dim %d must be a multiple of 16, at most 256", A);
    ST_CHECK_ARG(parts >= 2 && parts <= 8 && E % (4 * parts) == 0 && E / parts <= 256 && E / 4 <= AT_THREADS && AT_THREADS % (E / 4) == 0,
                 "st_query_attn_rng_fwd: parts=%d (2..8), E=%d", parts, E);
    const int Lp = (L + parts - 1) / parts;
    ST_CHECK_ARG(Lp <= 512 && (parts - 1) * Lp < L, "st_query_attn_rng_fwd: L=%d over %d parts leaves an empty or oversized range", L, parts);
    ST_CHECK_ARG(job->s_buf && job->memory && job->w_cum_prev && job->w_out && job->w_cum_out && job->v, "st_query_attn_rng_fwd: null attention operand");
    ST_CHECK_ARG(job->n_ctx_dst >= 1 && job->n_ctx_dst <= 3, "st_query_attn_rng_fwd: n_ctx_dst=%d", job->n_ctx_dst);
    ST_CHECK_ARG(st_aligned16(job->s_buf) && st_aligned16(job->memory) && st_aligned16(job->v) && (reinterpret_cast<uintptr_t>(granules) & 7) == 0 &&
                 (reinterpret_cast<uintptr_t>(xchg) & 7) == 0, "st_query_attn_rng_fwd: operands must be 16-byte aligned");
    PkArgs a;
    memset(&a, 0, sizeof(a));
    int rc = pk_fill(a, packed_wq, h_q, Q, "st_query_attn_rng_fwd");
    if (rc) return rc;
    a.B = B; a.N = A; a.H = 0; a.act = ST_ACT_NONE;
    a.gran = granules; a.epoch = epoch;
    ArArgs t;
    memset(&t, 0, sizeof(t));
    t.s_buf = job->s_buf; t.memory = job->memory; t.v = job->v; t.w_cum_prev = job->w_cum_prev;
    t.w_out = job->w_out; t.ld_wout = job->ld_wout; t.w_cum_out = job->w_cum_out;
    for (int d = 0; d < job->n_ctx_dst; ++d) t.ctx_dst[d] = job->ctx_dst[d];
    t.pq_gran = granules; t.xchg = xchg; t.epoch = epoch; t.status = job->status;
    t.B = B; t.L = L; t.A = A; t.E = E; t.P = parts; t.Lp = Lp;
    const int tiles = A / 16, BT = (B + 15) >> 4;
    const int n_lin = tiles * BT, n_rng = B * parts;
    // the waiting workgroups hold their compute units: everything must be resident at once
    ST_CHECK_ARG(n_lin + n_rng <= pk_attnrng_capacity(), "st_query_attn_rng_fwd: %d + %d workgroups do not fit the device at once (%d)",
                 n_lin, n_rng, pk_attnrng_capacity());
    constexpr int KW = 8;
    hipLaunchKernelGGL((pk_attnrng_kernel<1, KW, PK_TRIP_SMALL>), dim3(n_lin + n_rng), dim3(KW * 64), 0, (hipStream_t)stream, a.w, a.x, a.w_kbs,
                       a.x_kbs, a.KB, a.B, a.N, tiles, n_lin, a, t);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_skinny_linear_packed_fwd(const float* packed_w, const st_t16_view* x, int K,
                                           const float* bias, int act, const float* mask, int ldmask,
                                           float* y, int ldy, const st_t16_view* y_dst,
                                           int n_split, float* y2, int ldy2, int rep,
                                           int n_split2, int act2, const float* mask2, int ldmask2,
                                           const st_t16_view* y3_dst,
                                           int B, int N, void* stream) {
    (void)hipGetLastError();
    return pk_linear_impl(packed_w, x, K, bias, act, mask, ldmask, y, ldy, y_dst, n_split, y2, ldy2, rep,
                          n_split2, act2, mask2, ldmask2, y3_dst, B, N, nullptr, stream);
}

extern "C" int st_skinny_linear_packed_lstm_bwd_fwd(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                                    const st_lstm_pw_job* job, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(y && job && B > 0 && N > 0 && ldy >= N, "st_skinny_linear_packed_lstm_bwd_fwd: bad arguments");
    const int H = job->H;
    ST_CHECK_ARG(H > 0 && H % 4 == 0 && job->n0 >= 0 && job->n0 % 16 == 0 && job->n0 + H <= N && (N % 16 == 0 || job->n0 + H <= (N & ~15)),
                 "st_skinny_linear_packed_lstm_bwd_fwd: the cell's columns [%d, %d) must be whole tiles of the %d outputs", job->n0, job->n0 + H, N);
    ST_CHECK_ARG(job->gates && job->c && job->dc && job->dgates && job->ldg >= 4 * H && job->ldg % 4 == 0 && job->ldc % 4 == 0 &&
                 (!job->c_prev || job->ldcp % 4 == 0) && (!job->dh1 || job->ld1 % 4 == 0) && (!job->dh2 || job->ld2 % 4 == 0) && ldy % 4 == 0,
                 "st_skinny_linear_packed_lstm_bwd_fwd: null operand or a row stride that is not a multiple of 4");
    ST_CHECK_ARG(st_aligned16(job->gates) && st_aligned16(job->c) && st_aligned16(job->dc) && st_aligned16(job->dgates) && st_aligned16(y) &&
                 (!job->c_prev || st_aligned16(job->c_prev)) && (!job->dh1 || st_aligned16(job->dh1)) && (!job->dh2 || st_aligned16(job->dh2)) &&
                 (!job->scale2 || st_aligned16(job->scale2)) && (!job->mask || st_aligned16(job->mask)) &&
                 (!job->dgates_t16.base || st_aligned16(job->dgates_t16.base)), "st_skinny_linear_packed_lstm_bwd_fwd: operands must be 16-byte aligned");
    PkArgs a;
    memset(&a, 0, sizeof(a));
    int rc = pk_fill(a, packed_w, x, K, "st_skinny_linear_packed_lstm_bwd_fwd");
    if (rc) return rc;
    a.B = B; a.N = N; a.H = 0; a.act = ST_ACT_NONE;
    a.y = y; a.ldy = ldy;
    PkPw q;
    memset(&q, 0, sizeof(q));
    q.n0 = job->n0; q.H = H; q.dh1 = job->dh1; q.ld1 = job->ld1; q.dh1_slabs = job->dh1_slabs; q.dh1_slab_stride = job->dh1_slab_stride; q.dh2 = job->dh2; q.ld2 = job->ld2; q.scale2 = job->scale2; q.mask = job->mask;
    q.gates = job->gates; q.c = job->c; q.ldc = job->ldc; q.c_prev = job->c_prev; q.ldcp = job->ldcp; q.dc = job->dc;
    q.dgates = job->dgates; q.ldg = job->ldg; q.dg_t16 = pk_out(&job->dgates_t16);
    const int tiles = (N + 15) / 16, BT = (B + 15) >> 4;
    // one batch tile per workgroup, as the plain linear of these shapes runs (pk_dispatch<1>)
    hipLaunchKernelGGL((pk_pw_kernel<1, 8, 2>), dim3(tiles, BT), dim3(8 * 64), 0, (hipStream_t)stream, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B,
                       a.N, a, q);
    ST_LAUNCH_CHECK();
    return 0;
}

static int pk_pw_fill(PkPw& q, const st_lstm_pw_job* job, int N, int ldy, const float* y, const char* who) {
    memset(&q, 0, sizeof(q));
    const int H = job->H;
    ST_CHECK_ARG(H > 0 && H % 4 == 0 && job->n0 >= 0 && job->n0 % 16 == 0 && job->n0 + H <= N && (N % 16 == 0 || job->n0 + H <= (N & ~15)),
                 "%s: the cell's columns [%d, %d) must be whole tiles of the %d outputs", who, job->n0, job->n0 + H, N);
    ST_CHECK_ARG(job->gates && job->c && job->dc && job->dgates && job->ldg >= 4 * H && job->ldg % 4 == 0 && job->ldc % 4 == 0 &&
                 (!job->c_prev || job->ldcp % 4 == 0) && (!job->dh1 || job->ld1 % 4 == 0) && (!job->dh2 || job->ld2 % 4 == 0) && ldy % 4 == 0,
                 "%s: null operand or a row stride that is not a multiple of 4", who);
    ST_CHECK_ARG(st_aligned16(job->gates) && st_aligned16(job->c) && st_aligned16(job->dc) && st_aligned16(job->dgates) && (!y || st_aligned16(y)) &&
                 (!job->c_prev || st_aligned16(job->c_prev)) && (!job->dh1 || st_aligned16(job->dh1)) && (!job->dh2 || st_aligned16(job->dh2)) &&
                 (!job->scale2 || st_aligned16(job->scale2)) && (!job->mask || st_aligned16(job->mask)) &&
                 (!job->dgates_t16.base || st_aligned16(job->dgates_t16.base)), "%s: operands must be 16-byte aligned", who);
    q.n0 = job->n0; q.H = H; q.dh1 = job->dh1; q.ld1 = job->ld1; q.dh1_slabs = job->dh1_slabs; q.dh1_slab_stride = job->dh1_slab_stride; q.dh2 = job->dh2; q.ld2 = job->ld2; q.scale2 = job->scale2; q.mask = job->mask;
    q.gates = job->gates; q.c = job->c; q.ldc = job->ldc; q.c_prev = job->c_prev; q.ldcp = job->ldcp; q.dc = job->dc;
    q.dgates = job->dgates; q.ldg = job->ldg; q.dg_t16 = pk_out(&job->dgates_t16);
    return 0;
}

// two st_skinny_linear_packed_lstm_bwd_fwd of the same shape in one launch (arrays of two; y2 entries may be NULL: the product is
// only consumed by its epilogue): the two directions of a bidirectional LSTM layer's BPTT step
extern "C" int st_skinny_linear_packed_lstm_bwd_pair_fwd(const float* const* packed_w2, const st_t16_view* x2, int K, float* const* y2, int ldy,
                                                         int B, int N, const st_lstm_pw_job* job2, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(packed_w2 && x2 && job2 && B > 0 && N > 0 && ldy >= N, "st_skinny_linear_packed_lstm_bwd_pair_fwd: bad arguments");
    PkArgs a[2];
    PkPw q[2];
    for (int d = 0; d < 2; ++d) {
        memset(&a[d], 0, sizeof(PkArgs));
        int rc = pk_fill(a[d], packed_w2[d], &x2[d], K, "st_skinny_linear_packed_lstm_bwd_pair_fwd");
        if (rc) return rc;
        a[d].B = B; a[d].N = N; a[d].H = 0; a[d].act = ST_ACT_NONE;
        a[d].y = y2 ? y2[d] : nullptr; a[d].ldy = ldy;
        rc = pk_pw_fill(q[d], &job2[d], N, ldy, a[d].y, "st_skinny_linear_packed_lstm_bwd_pair_fwd");
        if (rc) return rc;
    }
    const int tiles = (N + 15) / 16, BT = (B + 15) >> 4;
    hipLaunchKernelGGL((pk_pw_pair_kernel<1, 8, 2>), dim3(2 * tiles * BT), dim3(8 * 64), 0, (hipStream_t)stream, tiles * BT, tiles, a[0], q[0], a[1], q[1]);
    ST_LAUNCH_CHECK();
    return 0;
}

// st_skinny_linear_packed_lstm_bwd_fwd (job may be NULL: the plain product) with one attention-step backward (the arguments of
// st_attn_step_bwd_t16 as a struct; the forward must have kept S: s_in != NULL) in the same launch
extern "C" int st_skinny_linear_packed_lstm_bwd_attn_bwd(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                                         const st_lstm_pw_job* job, const st_attn_bwd_job* ab, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(y && ab && B > 0 && N > 0 && ldy >= N, "st_skinny_linear_packed_lstm_bwd_attn_bwd: bad arguments");
    PkArgs a;
    memset(&a, 0, sizeof(a));
    int rc = pk_fill(a, packed_w, x, K, "st_skinny_linear_packed_lstm_bwd_attn_bwd");
    if (rc) return rc;
    a.B = B; a.N = N; a.H = 0; a.act = ST_ACT_NONE;
    a.y = y; a.ldy = ldy;
    PkPw q;
    memset(&q, 0, sizeof(q));
    if (job) {
        const int H = job->H;
        ST_CHECK_ARG(H > 0 && H % 4 == 0 && job->n0 >= 0 && job->n0 % 16 == 0 && job->n0 + H <= N && (N % 16 == 0 || job->n0 + H <= (N & ~15)),
                     "st_skinny_linear_packed_lstm_bwd_attn_bwd: the cell's columns [%d, %d) must be whole tiles of the %d outputs", job->n0, job->n0 + H, N);
        ST_CHECK_ARG(job->gates && job->c && job->dc && job->dgates && job->ldg >= 4 * H && job->ldg % 4 == 0 && job->ldc % 4 == 0 &&
                     (!job->c_prev || job->ldcp % 4 == 0) && (!job->dh1 || job->ld1 % 4 == 0) && (!job->dh2 || job->ld2 % 4 == 0) && ldy % 4 == 0,
                     "st_skinny_linear_packed_lstm_bwd_attn_bwd: null operand or a row stride that is not a multiple of 4");
        ST_CHECK_ARG(st_aligned16(job->gates) && st_aligned16(job->c) && st_aligned16(job->dc) && st_aligned16(job->dgates) && st_aligned16(y) &&
                     (!job->c_prev || st_aligned16(job->c_prev)) && (!job->dh1 || st_aligned16(job->dh1)) && (!job->dh2 || st_aligned16(job->dh2)) &&
                     (!job->scale2 || st_aligned16(job->scale2)) && (!job->mask || st_aligned16(job->mask)) &&
                     (!job->dgates_t16.base || st_aligned16(job->dgates_t16.base)), "st_skinny_linear_packed_lstm_bwd_attn_bwd: operands must be 16-byte aligned");
        q.n0 = job->n0; q.H = H; q.dh1 = job->dh1; q.ld1 = job->ld1; q.dh1_slabs = job->dh1_slabs; q.dh1_slab_stride = job->dh1_slab_stride; q.dh2 = job->dh2; q.ld2 = job->ld2; q.scale2 = job->scale2; q.mask = job->mask;
        q.gates = job->gates; q.c = job->c; q.ldc = job->ldc; q.c_prev = job->c_prev; q.ldcp = job->ldcp; q.dc = job->dc;
        q.dgates = job->dgates; q.ldg = job->ldg; q.dg_t16 = pk_out(&job->dgates_t16);
    }      // (no job: H = 0, no column belongs to a cell -- the plain product)
    AbArgs t;
    if (ab_fill_job(t, ab)) return -1;
    ST_CHECK_ARG(t.s_in, "st_skinny_linear_packed_lstm_bwd_attn_bwd: the hosted attention backward starts from the forward's S (s_in)");
    const size_t red_bytes = (size_t)8 * 1 * 64 * sizeof(f32x4);          // the product's static LDS in the same workgroup
    const bool wide = ab_wide(t, red_bytes);
    // parts > 1: the split form (see pk_pw_ab_kernel); the caller then runs st_skinny_linear_packed_lstm_bwd_attn_hist next
    const int parts = ab->parts > 1 ? ab->parts : 1;
    const size_t lds = ab_lds_bytes(t, wide || parts > 1, parts);      // (the split form always runs the 48-position block, on its lean image)
    if (parts > 1) {
        ST_CHECK_ARG(parts == 2 || parts == 4, "st_skinny_linear_packed_lstm_bwd_attn_bwd: parts = %d (1, 2 or 4)", parts);
        ST_CHECK_ARG(ab->dloc_part && !ab->dcum_add && t.s_in && lds + red_bytes <= 160 * 1024 && t.A % parts == 0 && AB_THREADS % (t.A / parts) == 0 &&
                     (t.A / parts) % 16 == 0 && AB_THREADS / (t.A / parts) >= 2 * parts && t.A / parts <= AB_THREADS / (2 * parts),
                     "st_skinny_linear_packed_lstm_bwd_attn_bwd: parts = %d needs dloc_part, no dcum_add (the history job keeps dcum), the wide block "
                     "and A = %d splitting into parts of a multiple of 16 dims that divide %d", parts, t.A, AB_THREADS);
        t.dloc_part = ab->dloc_part;
    }
    if (lds + red_bytes > 160 * 1024) {
        // a text so long that the attention backward needs (nearly) all the LDS of a compute unit for itself: the two launches one after
        // the other (they are independent: any order)
        if (job) rc = st_skinny_linear_packed_lstm_bwd_fwd(packed_w, x, K, y, ldy, B, N, job, stream);
        else rc = st_skinny_linear_packed_fwd(packed_w, x, K, nullptr, ST_ACT_NONE, nullptr, 0, y, ldy, nullptr, 0, nullptr, 0, 0, 0, 0, nullptr, 0,
                                              nullptr, B, N, stream);
        if (rc) return rc;
        return st_attn_step_bwd_t16(ab->pq, ab->pm, ab->memory, ab->w_prev, ab->ld_wprev, ab->w_cum_prev, ab->w, ab->ld_w, ab->loc_conv_w,
                                    ab->loc_lin_w, ab->v, t.dctx, t.ld_dctx, AB_NDCTX, ab->dw_direct, ab->ld_dw, ab->n_dw, ab->dcum,
                                    ab->dcum_add, ab->ld_dcum_add, ab->dpq, &ab->dpq_t16, ab->dhist, ab->ds_t, ab->loc_t, ab->dloc_t,
                                    ab->hist_t, ab->dctx_t, ab->dv_t, ab->s_in, ab->B, ab->L, ab->A, ab->E, ab->F, ab->K, stream);
    }
    const int tiles = (N + 15) / 16, BT = (B + 15) >> 4;
    if (parts == 2 && BT == 2 && tiles + 2 * t.B <= 256 && getenv("ST_AB_NB2")) {
        // both batch tiles of a row tile in ONE eight-wave workgroup (pk_body NB = 2: every weight fragment feeds two MFMAs): N / 16 product
        // workgroups + 2 B attention workgroups <= 256: one round, a compute unit each
        auto k2 = pk_pw_ab_kernel<2, 8, 2, AB_LBLK_MAX, 2>;
        static size_t lds_nb2 = 0;
        if (lds > 32 * 1024 && lds > lds_nb2) {
            ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_nb2 = lds;
        }
        hipLaunchKernelGGL(k2, dim3(2 * t.B + tiles), dim3(8 * 64), lds, (hipStream_t)stream, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.N, tiles,
                           2 * t.B, a, q, t);
        ST_LAUNCH_CHECK();
        return 0;
    }
    if (parts == 2 && BT == 2 && tiles + 2 * t.B <= 256 && !getenv("ST_AB_NO_DUAL")) {
        auto kd = pk_pw_ab_dual_kernel<8, 2, AB_LBLK_MAX, 2>;
        static size_t lds_dual = 0;
        if (lds > 32 * 1024 && lds > lds_dual) {
            ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_dual = lds;
        }
        hipLaunchKernelGGL(kd, dim3(2 * t.B + tiles), dim3(2 * 8 * 64), lds, (hipStream_t)stream, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.N,
                           2 * t.B, a, q, t, st_exp_flag());
        ST_LAUNCH_CHECK();
        return 0;
    }
    auto kern = parts == 4 ? pk_pw_ab_kernel<1, 8, 2, AB_LBLK_MAX, 4> : parts == 2 ? pk_pw_ab_kernel<1, 8, 2, AB_LBLK_MAX, 2>
                : wide ? pk_pw_ab_kernel<1, 8, 2, AB_LBLK_MAX, 1> : pk_pw_ab_kernel<1, 8, 2, 16, 1>;
    static size_t lds_set[4] = {0, 0, 0, 0};
    const int ki = parts == 4 ? 3 : parts == 2 ? 2 : wide ? 1 : 0;
    if (lds > 48 * 1024 && lds > lds_set[ki]) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set[ki] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(parts * t.B + tiles * BT), dim3(8 * 64), lds, (hipStream_t)stream, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.N, tiles,
                       parts * t.B, a, q, t);
    ST_LAUNCH_CHECK();
    return 0;
}

// st_skinny_linear_packed_lstm_bwd_fwd with the history part of a split attention backward (st_attn_hist_job) in the same launch
static int pk_hist_sum_impl(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                            const st_lstm_pw_job* job, const st_attn_hist_job* hj, const st_partial_sum_job* sj, void* stream);

extern "C" int st_skinny_linear_packed_lstm_bwd_attn_hist(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                                          const st_lstm_pw_job* job, const st_attn_hist_job* hj, void* stream) {
    ST_CHECK_ARG(job && hj, "st_skinny_linear_packed_lstm_bwd_attn_hist: null job");
    return pk_hist_sum_impl(packed_w, x, K, y, ldy, B, N, job, hj, nullptr, stream);
}

// ... and with the sum of a K-split partial product (st_partial_sum_job: the slabs st_skinny_partial_attn_bwd wrote in the launch before);
// hj may be NULL here (the history part can ride in another launch: st_skinny_linear_packed_attn_hist)
extern "C" int st_skinny_linear_packed_lstm_bwd_attn_hist_sum(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                                              const st_lstm_pw_job* job, const st_attn_hist_job* hj,
                                                              const st_partial_sum_job* sj, void* stream) {
    ST_CHECK_ARG(job && sj, "st_skinny_linear_packed_lstm_bwd_attn_hist_sum: null job");
    return pk_hist_sum_impl(packed_w, x, K, y, ldy, B, N, job, hj, sj, stream);
}

// the plain product y = x W^T (st_skinny_linear_packed_fwd without epilogue) with an st_attn_hist_job beside it
extern "C" int st_skinny_linear_packed_attn_hist(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                                 const st_attn_hist_job* hj, void* stream) {
    ST_CHECK_ARG(hj, "st_skinny_linear_packed_attn_hist: null history job");
    return pk_hist_sum_impl(packed_w, x, K, y, ldy, B, N, nullptr, hj, nullptr, stream);
}

static int pk_sum_fill(PkSumArgs& sa, const st_partial_sum_job* sj, int B, const char* who) {
    memset(&sa, 0, sizeof(sa));
    ST_CHECK_ARG(sj->part && sj->S >= 1 && sj->N > 0 && sj->N % 4 == 0 && sj->y && sj->ldy >= sj->N && sj->ldy % 4 == 0 && st_aligned16(sj->part) &&
                 st_aligned16(sj->y), "%s: bad sum job (N and ldy multiples of 4, 16-byte aligned slabs and output)", who);
    sa.part = sj->part; sa.S = sj->S; sa.B = B; sa.N = sj->N; sa.y = sj->y; sa.ldy = sj->ldy;
    if (sj->pw) {
        int rc = pk_pw_fill(sa.pw, sj->pw, sj->N, sj->ldy, sj->y, who);
        if (rc) return rc;
    }
    return 0;
}

static int pk_hist_sum_impl(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                            const st_lstm_pw_job* job, const st_attn_hist_job* hj, const st_partial_sum_job* sj, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(y && (hj || sj) && B > 0 && N > 0 && ldy >= N, "st_skinny_linear_packed_*_attn_hist: bad arguments");
    PkArgs a;
    memset(&a, 0, sizeof(a));
    int rc = pk_fill(a, packed_w, x, K, "st_skinny_linear_packed_*_attn_hist");
    if (rc) return rc;
    a.B = B; a.N = N; a.H = 0; a.act = ST_ACT_NONE;
    a.y = y; a.ldy = ldy;
    PkPw q;
    memset(&q, 0, sizeof(q));
    if (job) {
        rc = pk_pw_fill(q, job, N, ldy, y, "st_skinny_linear_packed_lstm_bwd_attn_hist");
        if (rc) return rc;
    }
    AbHistArgs h;
    memset(&h, 0, sizeof(h));
    size_t lds = 0;
    if (hj) {
        ST_CHECK_ARG(hj->dloc_part && hj->parts >= 1 && hj->parts <= 4 && hj->loc_conv_w && hj->w_cum_prev && hj->dloc_t && hj->hist_t && hj->dhist &&
                     hj->B > 0 && hj->L > 0 && hj->F > 0 && hj->K > 0 && (hj->K & 1) && hj->K <= 31,
                     "st_skinny_linear_packed_*_attn_hist: bad history job (odd K <= 31)");
        h.dloc_part = hj->dloc_part; h.parts = hj->parts; h.loc_conv_w = hj->loc_conv_w; h.w_prev = hj->w_prev; h.ld_wprev = hj->ld_wprev;
        h.w_cum_prev = hj->w_cum_prev; h.dloc_t = hj->dloc_t; h.hist_t = hj->hist_t; h.dhist = hj->dhist; h.dcum = hj->dcum;
        h.B = hj->B; h.L = hj->L; h.F = hj->F; h.K = hj->K;
        lds = (size_t)ab_hist_lds_floats(h.L, h.F, h.K) * sizeof(float);
    }
    const size_t red_bytes = (size_t)8 * 1 * 64 * sizeof(f32x4);
    ST_CHECK_ARG(lds + red_bytes <= 160 * 1024, "st_skinny_linear_packed_*_attn_hist: L=%d needs %zu bytes of LDS", h.L, lds);
    auto kern = job ? pk_pw_hist_kernel<2, 1, 8, 2> : pk_pw_hist_kernel<1, 1, 8, 2>;
    static size_t lds_sets[2] = {0, 0};
    size_t& lds_set = lds_sets[job ? 1 : 0];
    if (lds > 48 * 1024 && lds > lds_set) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    PkSumArgs sa;
    memset(&sa, 0, sizeof(sa));
    int n_sum = 0;
    if (sj) {
        rc = pk_sum_fill(sa, sj, B, "st_skinny_linear_packed_lstm_bwd_attn_hist_sum");
        if (rc) return rc;
        n_sum = (B * (sa.N >> 2) + 8 * 64 - 1) / (8 * 64);
    }
    const int tiles = (N + 15) / 16, BT = (B + 15) >> 4;
    const int n_h = hj ? h.B : 0;
    hipLaunchKernelGGL(kern, dim3(n_h + n_sum + tiles * BT), dim3(8 * 64), lds, (hipStream_t)stream, a.w, a.x, a.w_kbs, a.x_kbs, a.KB, a.B, a.N, tiles,
                       n_h, n_sum, a, q, h, sa, st_exp_flag());
    ST_LAUNCH_CHECK();
    return 0;
}

// st_skinny_partial_attn_bwd's product with an st_attn_hist_job beside it instead of the attention backward
extern "C" int st_skinny_partial_attn_hist(const float* packed_w, const st_t16_view* x, int K, float* part, int S, int B, int N,
                                           const st_attn_hist_job* hj, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(packed_w && x && x->base && part && hj && K > 0 && K % 16 == 0 && S >= 1 && (K / 16) % S == 0 && B > 16 && B <= 32 && N > 0 && N % 32 == 0 &&
                 st_aligned16(part), "st_skinny_partial_attn_hist: needs 16 < B <= 32 (two batch tiles), N %% 32 == 0, K %% (16 S) == 0 (B=%d N=%d K=%d S=%d)", B, N, K, S);
    ST_CHECK_ARG(hj->dloc_part && hj->parts >= 1 && hj->parts <= 4 && hj->loc_conv_w && hj->w_cum_prev && hj->dloc_t && hj->hist_t && hj->dhist &&
                 hj->B > 0 && hj->L > 0 && hj->F > 0 && hj->K > 0 && (hj->K & 1) && hj->K <= 31, "st_skinny_partial_attn_hist: bad history job (odd K <= 31)");
    PkPartArgs p;
    p.w = reinterpret_cast<const f32x4*>(packed_w); p.w_kbs = K / 16;
    p.x = reinterpret_cast<const f32x4*>(x->base) + (size_t)x->kb0 * 64; p.x_kbs = x->kb_stride;
    p.KB = K / 16; p.S = S; p.B = B; p.N = N; p.part = part;
    AbHistArgs h;
    memset(&h, 0, sizeof(h));
    h.dloc_part = hj->dloc_part; h.parts = hj->parts; h.loc_conv_w = hj->loc_conv_w; h.w_prev = hj->w_prev; h.ld_wprev = hj->ld_wprev;
    h.w_cum_prev = hj->w_cum_prev; h.dloc_t = hj->dloc_t; h.hist_t = hj->hist_t; h.dhist = hj->dhist; h.dcum = hj->dcum;
    h.B = hj->B; h.L = hj->L; h.F = hj->F; h.K = hj->K;
    const size_t lds = (size_t)ab_hist_lds_floats(h.L, h.F, h.K) * sizeof(float);
    ST_CHECK_ARG(lds + 8 * 4 * 64 * sizeof(f32x4) <= 160 * 1024, "st_skinny_partial_attn_hist: L=%d needs %zu bytes of LDS", h.L, lds);
    auto kern = pk_part_hist_kernel<8, 2>;
    static size_t lds_set = 0;
    if (lds > 32 * 1024 && lds > lds_set) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(h.B + (N / 32) * S), dim3(8 * 64), lds, (hipStream_t)stream, p, h.B, h);
    ST_LAUNCH_CHECK();
    return 0;
}

// K-split partial product part[s] (B, N) = x[:, K-range s] . W[:, K-range s]^T (see pk_part_body; B <= 32, N % 32 == 0, (K / 16) % S == 0) with the
// split attention backward of a BPTT step (ab->parts = 2, as st_skinny_linear_packed_lstm_bwd_attn_bwd) beside it in ONE launch; ab NULL: the
// partial product alone.  The caller adds the slabs with an st_partial_sum_job in its next launch.
extern "C" int st_skinny_partial_attn_bwd(const float* packed_w, const st_t16_view* x, int K, float* part, int S, int B, int N,
                                          const st_attn_bwd_job* ab, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(packed_w && x && x->base && part && K > 0 && K % 16 == 0 && S >= 1 && (K / 16) % S == 0 && B > 16 && B <= 32 && N > 0 && N % 32 == 0 &&
                 st_aligned16(part), "st_skinny_partial_attn_bwd: needs 16 < B <= 32 (two batch tiles), N %% 32 == 0, K %% (16 S) == 0 (B=%d N=%d K=%d S=%d)", B, N, K, S);
    PkPartArgs p;
    p.w = reinterpret_cast<const f32x4*>(packed_w); p.w_kbs = K / 16;
    p.x = reinterpret_cast<const f32x4*>(x->base) + (size_t)x->kb0 * 64; p.x_kbs = x->kb_stride;
    p.KB = K / 16; p.S = S; p.B = B; p.N = N; p.part = part;
    const int n_prod = (N / 32) * S;
    if (!ab) {
        hipLaunchKernelGGL((pk_part_kernel<8, 2>), dim3(n_prod), dim3(8 * 64), 0, (hipStream_t)stream, p);
        ST_LAUNCH_CHECK();
        return 0;
    }
    AbArgs t;
    if (ab_fill_job(t, ab)) return -1;
    const size_t red_bytes = (size_t)8 * 4 * 64 * sizeof(f32x4);
    ST_CHECK_ARG(ab->parts == 2 && ab->dloc_part && !ab->dcum_add && t.s_in && ab_lds_bytes(t, true, 2) + red_bytes <= 160 * 1024 && t.A % 32 == 0 &&
                 AB_THREADS % (t.A / 2) == 0 && AB_THREADS / (t.A / 2) >= 4, "st_skinny_partial_attn_bwd: the attention job must be the two-part form (parts = 2, dloc_part, S kept, "
                 "no dcum_add) with A = %d splitting into halves that divide %d", t.A, AB_THREADS);
    t.dloc_part = ab->dloc_part;
    const size_t lds = ab_lds_bytes(t, true, 2);
    const int kw16 = getenv("ST_PART_KW16") ? 1 : 0;
    auto kern = kw16 ? pk_part_ab_kernel<16, 2, AB_LBLK_MAX, 2> : pk_part_ab_kernel<8, 2, AB_LBLK_MAX, 2>;      // (TRIP = 3 / 4, i.e. deeper groups in flight: 14.4 / 13.8 us against 12.8)
    const int trip = 2 + kw16;
    static size_t lds_set[5] = {0, 0, 0, 0, 0};
    if (lds > 32 * 1024 && lds > lds_set[trip]) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set[trip] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(2 * t.B + n_prod), dim3((kw16 ? 16 : 8) * 64), lds, (hipStream_t)stream, p, 2 * t.B, t, st_exp_flag());
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_skinny_linear_packed_attnpre_fwd(const float* packed_w, const st_t16_view* x, int K,
                                                   const float* bias, int act, const float* mask, int ldmask,
                                                   float* y, int ldy, const st_t16_view* y_dst,
                                                   int n_split, float* y2, int ldy2, int rep,
                                                   int n_split2, int act2, const float* mask2, int ldmask2,
                                                   const st_t16_view* y3_dst,
                                                   int B, int N, const st_attn_pre_job* pre, void* stream) {
    (void)hipGetLastError();
    return pk_linear_impl(packed_w, x, K, bias, act, mask, ldmask, y, ldy, y_dst, n_split, y2, ldy2, rep,
                          n_split2, act2, mask2, ldmask2, y3_dst, B, N, pre, stream);
}
