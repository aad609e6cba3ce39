// rnn.hip -- full-sequence recurrent layers for gfx950 (MI355X).
//
//  * nn.GRU of the CBHG (H = n_mels = 80, T = 258 steps): the recurrent matrix is only
//    3H x H = 77 KB, so one workgroup per (utterance, direction) keeps its rows in registers
//    (thread j owns row j of W_hh), the hidden state in LDS, and runs all T steps without
//    touching HBM except for the precomputed input projection gi[t] (prefetched one step
//    ahead) and the output row.
//  * nn.LSTM of the text encoder (H = 256 per direction, L = 43 steps): W_hh is 1 MB, too
//    large for one CU, so each time step is one launch of the weight-streaming LSTM-cell
//    kernel (skinny.hip) with the precomputed input projection as the `pre` addend.
#include "st_common.h"

namespace {

constexpr int GRU_HMAX = 128;

struct GruArgs {
    const float* gi[2]; const float* w_hh[2]; const float* b_hh[2];
    float* out; int ldo; int B, T, H;
    float* tape;     // training: [dir][b][t][4][H] = (r, z, n, W_hn h + b_hn); NULL in inference
};

// H <= 128: the three gate rows of hidden unit j live in three ADJACENT lanes (thread 4j + g, g = 0..2 = r, z, n; lane 3 of the
// quad idles), each with its W_hh row in registers.  A step is: 80-term dot per lane as four independent fma chains over the
// LDS-resident h (ds_read_b128 broadcasts), the two other gates fetched from the neighbour lanes with DPP quad permutes (no LDS
// exchange), the pointwise update on lane g = 0, h written to the OTHER of two LDS buffers -- ONE LDS-only barrier per step
// (the first form of this kernel: thread j = row j, gates exchanged through LDS, two barriers, one 80-deep fma chain: 1.42 us per
// step of the 258-step CBHG GRU = 43 % of the postnet).
// KQ = compile-time number of 4-float pieces of h (H <= 4 KQ): the dot product is a branch-free run of KQ ds_read_b128 + 4 KQ FMAs
// (with a run-time `k < H` test inside the unrolled loop the compiler emitted a branch, a wait and SGPR-spill traffic per piece:
// 1.4-1.6 us per step whatever the arithmetic; weights past H are zero and the padded h entries are zero).
template <int KQ>
__global__ __launch_bounds__(512) void gru_seq_quad_kernel(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.H, H3 = 3 * H, T = a.T;
    constexpr int HP = 4 * KQ;
    float* hb[2] = {lds, lds + HP};
    const int b = blockIdx.x, d = blockIdx.y;
    const int tid = threadIdx.x, j = tid >> 2, g = tid & 3;
    const float* __restrict__ gi = a.gi[d] + (size_t)b * T * H3;
    const bool row_ok = j < H && g < 3;
    float wreg[HP];
    float bh;
    {   // branch-free: idle lanes and columns past H read a valid address and are zeroed by a 0 / 1 factor (a select on a loaded value
        // becomes a branch around the load with a full wait behind it)
        const float rok = row_ok ? 1.0f : 0.0f;
        const float* wr = a.w_hh[d] + (size_t)(row_ok ? g * H + j : 0) * H;
        bh = a.b_hh[d][row_ok ? g * H + j : 0] * rok;
#pragma unroll
        for (int k = 0; k < HP; ++k) wreg[k] = wr[k < H ? k : H - 1] * (k < H ? rok : 0.0f);
    }
    for (int k = tid; k < 2 * HP; k += blockDim.x) lds[k] = 0.0f;
    __syncthreads();
    const bool upd = j < H && g == 0;
    float hprev = 0.f;
    // The input projections are read GRU_PB steps ahead: a step takes ~0.3 us, a load from HBM / L2 1-2 us, so a one-step
    // prefetch (the first form of this kernel) left every step waiting for memory.  Blocks of GRU_PB steps: the next block's
    // three gate inputs per step are requested before the current block is computed out of registers.
    constexpr int GRU_PB = 8;
    float cur[GRU_PB][3], nxt[GRU_PB][3];
    auto load_block = [&](float (&dst)[GRU_PB][3], int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GRU_PB; ++i) {
            const int s = s0 + i;
            const int tt = max(min(d ? T - 1 - s : s, T - 1), 0);
            // every lane reads three valid addresses (idle lanes and steps past T: values nobody uses).  No condition on the loads: a
            // `cond ? p[H] : p[0]` becomes a branch with a full s_waitcnt per step, and the "prefetch" is eight serial round trips
            const float* p = gi + (size_t)tt * H3 + (upd ? j : 0);
            dst[i][0] = p[0]; dst[i][1] = p[H]; dst[i][2] = p[2 * H];
        }
    };
    load_block(cur, 0);
    for (int s0 = 0; s0 < T; s0 += GRU_PB) {
        load_block(nxt, s0 + GRU_PB);
#pragma unroll
        for (int i = 0; i < GRU_PB; ++i) {
            const int s = s0 + i;
            if (s >= T) break;                                        // (uniform)
            const int t = d ? T - 1 - s : s;
            const float* hcur = hb[s & 1];
            float a0 = bh, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int k = 0; k < HP; k += 4) {          // padded h entries are zero, weights past H are zero
                const f32x4 h4 = *reinterpret_cast<const f32x4*>(hcur + k);
                a0 = fmaf(wreg[k], h4[0], a0); a1 = fmaf(wreg[k + 1], h4[1], a1);
                a2 = fmaf(wreg[k + 2], h4[2], a2); a3 = fmaf(wreg[k + 3], h4[3], a3);
            }
            const float gh = (a0 + a1) + (a2 + a3);                   // W_hh[g*H + j] . h + b_hh
            const float ghz = st_dpp<0x55>(gh), ghn = st_dpp<0xAA>(gh);      // quad_perm [1,1,1,1] / [2,2,2,2]: lanes g = 1, 2 of the quad
            if (upd) {
                const float r = st_sigmoid_fast(cur[i][0] + gh);
                const float z = st_sigmoid_fast(cur[i][1] + ghz);
                const float n = st_tanh_fast(cur[i][2] + r * ghn);
                const float hn = (1.0f - z) * n + z * hprev;
                hprev = hn;
                hb[(s + 1) & 1][j] = hn;
                a.out[((size_t)b * T + t) * a.ldo + d * H + j] = hn;
                if (a.tape) {
                    float* tp = a.tape + ((((size_t)d * a.B + b) * T + t) * 4) * H + j;
                    tp[0] = r; tp[H] = z; tp[2 * H] = n; tp[3 * H] = ghn;
                }
            }
            st_lds_barrier();      // (LDS-only: __syncthreads would also wait for the block of input projections just requested)
        }
#pragma unroll
        for (int i = 0; i < GRU_PB; ++i) { cur[i][0] = nxt[i][0]; cur[i][1] = nxt[i][1]; cur[i][2] = nxt[i][2]; }
    }
}

// H <= 84 (the CBHG GRU: 80): FOUR waves, one per SIMD; a wave owns 21 hidden units, three ADJACENT lanes each (lane 3u + s; lane
// 63 idles).  Round 4: the three lanes of a unit split the REDUCTION, not the gates -- lane s holds columns [28 s, 28 s + 28) of ALL
// three gate rows of its unit (3 x 28 weights in registers) and reads only ITS 28 entries of h: seven ds_read_b128 per lane instead
// of twenty.  Phase stamps of the gate-per-lane form (tools/exp_gru_stamps.py): 572 of a step's 1644 cycles were the 80 broadcast
// reads of h queuing on the LDS pipe (4 waves x 20 b128 x 4 cycles), 520 the multiply-adds.  The three partial sums of a gate meet
// on lane s = 0 with whole-wave DPP shifts (wave_shl:1, :2), which then does the pointwise update.  One LDS-only barrier per step.
// The output / tape / input-projection addresses are running pointers (the 64-bit index arithmetic per step was ~25 scalar
// instructions in front of the stores).
#ifdef GRU_STAMPS     // phase stamps of workgroup (0, 0), wave 0, step 100 (experiment builds only: tools/exp_gru_stamps.py)
__device__ unsigned long long gru_stamps[16];
#define GRU_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && s == 100) gru_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define GRU_STAMP(i) do { } while (0)
#endif
constexpr int GRU_SL = 28;            // columns per lane (3 x 28 = 84 >= H)
__global__ __launch_bounds__(256) void gru_seq_tri_kernel(const GruArgs a) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][3 * GRU_SL];
    const int H = a.H, H3 = 3 * H, T = a.T;
    const int b = blockIdx.x, d = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = lane / 3, sl = lane - 3 * u, j = wave * 21 + u;
    const float* __restrict__ gi = a.gi[d] + (size_t)b * T * H3;
    const bool unit_ok = lane < 63 && j < H;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 wreg[3][GRU_SL / 2];           // (pairs of adjacent columns: the multiply-adds are v_pk_fma_f32 -- a lone wave issues one VALU
    float bh[3];                         //  instruction per ~5 cycles, so halving their count is what shortens the phase)
    {   // branch-free: idle lanes and columns past H read a valid address and are zeroed by a 0 / 1 FACTOR.  (Not by a select: the
        // compiler turns `cond ? loaded : 0` into a branch around the load with a full s_waitcnt behind it -- 84 serialised round trips.)
        const float uok = unit_ok ? 1.0f : 0.0f;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float* wr = a.w_hh[d] + (size_t)(unit_ok ? g * H + j : 0) * H;
            bh[g] = a.b_hh[d][unit_ok ? g * H + j : 0] * uok;
#pragma unroll
            for (int k = 0; k < GRU_SL; ++k) {
                const int col = GRU_SL * sl + k;
                wreg[g][k >> 1][k & 1] = wr[col < H ? col : H - 1] * (col < H ? uok : 0.0f);
            }
        }
    }
    for (int k = tid; k < 2 * 3 * GRU_SL; k += blockDim.x) (&hbuf[0][0])[k] = 0.0f;
    __syncthreads();
    const bool upd = unit_ok && sl == 0;
    float hprev = 0.f;
    constexpr int GRU_PB = 8;      // input projections read GRU_PB steps ahead (see gru_seq_quad_kernel)
    float cur[GRU_PB][3], nxt[GRU_PB][3];
    auto load_block = [&](float (&dst)[GRU_PB][3], int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GRU_PB; ++i) {
            const int s = s0 + i;
            const int tt = max(min(d ? T - 1 - s : s, T - 1), 0);
            // every lane reads three valid addresses (idle lanes and steps past T: values nobody uses).  No condition on the loads: a
            // `cond ? p[H] : p[0]` becomes a branch with a full s_waitcnt per step, and the "prefetch" is eight serial round trips
            const float* p = gi + (size_t)tt * H3 + (upd ? j : 0);
            dst[i][0] = p[0]; dst[i][1] = p[H]; dst[i][2] = p[2 * H];
        }
    };
    load_block(cur, 0);
    // running pointers of this lane's output element and tape record (step s -> frame t = s or T - 1 - s)
    const ptrdiff_t ostep = (ptrdiff_t)(d ? -1 : 1) * a.ldo, tstep = (ptrdiff_t)(d ? -1 : 1) * 4 * H;
    float* op = a.out + ((size_t)b * T + (d ? T - 1 : 0)) * a.ldo + d * H + (upd ? j : 0);
    float* tp = a.tape ? a.tape + ((((size_t)d * a.B + b) * T + (d ? T - 1 : 0)) * 4) * H + (upd ? j : 0) : nullptr;
    for (int s0 = 0; s0 < T; s0 += GRU_PB) {
        load_block(nxt, s0 + GRU_PB);
#pragma unroll
        for (int i = 0; i < GRU_PB; ++i) {
            const int s = s0 + i;
            if (s >= T) break;                                        // (uniform)
            const float* hcur = hbuf[s & 1] + GRU_SL * sl;
            GRU_STAMP(0);
            f32x4 hreg[GRU_SL / 4];
#pragma unroll
            for (int k = 0; k < GRU_SL / 4; ++k) hreg[k] = *reinterpret_cast<const f32x4*>(hcur + 4 * k);
            asm volatile("" ::: "memory");             // (keeps the reads above the multiply-adds)
#ifdef GRU_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            GRU_STAMP(1);
            f32x2 acc[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = f32x2{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < GRU_SL / 4; ++k) {     // three independent chains of packed multiply-adds (x: even columns, y: odd columns)
                const f32x2 h01 = {hreg[k][0], hreg[k][1]}, h23 = {hreg[k][2], hreg[k][3]};
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_elementwise_fma(wreg[g][2 * k], h01, acc[g]);
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_elementwise_fma(wreg[g][2 * k + 1], h23, acc[g]);
            }
            float gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {              // the unit's three column slices meet on lane s = 0 (wave_shl:1 = the value of lane + 1)
                const float part = acc[g].x + acc[g].y;
                const float p1 = st_dpp<0x130>(part), p2 = st_dpp<0x130>(p1);
                gh[g] = ((part + p1) + p2) + bh[g];    // W_hh[g*H + j] . h + b_hh
            }
            GRU_STAMP(2);
            if (upd) {
                const float r = st_sigmoid_fast(cur[i][0] + gh[0]);
                const float z = st_sigmoid_fast(cur[i][1] + gh[1]);
                const float n = st_tanh_fast(cur[i][2] + r * gh[2]);
                const float hn = (1.0f - z) * n + z * hprev;
                hprev = hn;
                GRU_STAMP(3);
                hbuf[(s + 1) & 1][j] = hn;
                *op = hn;
                if (tp) { tp[0] = r; tp[H] = z; tp[2 * H] = n; tp[3 * H] = gh[2]; }
            }
            op += ostep;
            if (tp) tp += tstep;
            GRU_STAMP(4);
            st_lds_barrier();
            GRU_STAMP(5);
        }
#pragma unroll
        for (int i = 0; i < GRU_PB; ++i) { cur[i][0] = nxt[i][0]; cur[i][1] = nxt[i][1]; cur[i][2] = nxt[i][2]; }
    }
}

template <bool REG>
__global__ __launch_bounds__(REG ? 384 : 1024) void gru_seq_kernel(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.H, H3 = 3 * H, T = a.T;
    const int H4 = (H + 3) & ~3;
    float* hbuf = lds;            // [H4]
    float* ghs = lds + H4;        // [3H]
    const int b = blockIdx.x, d = blockIdx.y;
    const int j = threadIdx.x;
    const float* __restrict__ gi = a.gi[d] + (size_t)b * T * H3;
    const float* __restrict__ whh = a.w_hh[d];
    const bool row_ok = j < H3;
    float wreg[REG ? GRU_HMAX : 4];
    float bh = 0.0f;
    if (row_ok) {
        bh = a.b_hh[d][j];
        if (REG) {
#pragma unroll
            for (int k = 0; k < GRU_HMAX; ++k) wreg[k] = k < H ? whh[(size_t)j * H + k] : 0.0f;
        }
    }
    for (int k = j; k < H4; k += blockDim.x) hbuf[k] = 0.0f;
    __syncthreads();

    const bool upd = j < H;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    int t = d ? T - 1 : 0;
    if (upd) { gr = gi[(size_t)t * H3 + j]; gz = gi[(size_t)t * H3 + H + j]; gn = gi[(size_t)t * H3 + 2 * H + j]; }
    for (int s = 0; s < T; ++s) {
        // prefetch the next step's input projection while this step computes
        const int tn = d ? t - 1 : t + 1;
        float nr = 0.f, nz = 0.f, nn = 0.f;
        if (upd && s + 1 < T) {
            nr = gi[(size_t)tn * H3 + j]; nz = gi[(size_t)tn * H3 + H + j]; nn = gi[(size_t)tn * H3 + 2 * H + j];
        }
        if (row_ok) {
            float acc = bh;
            if (REG) {
#pragma unroll
                for (int k = 0; k < GRU_HMAX; k += 4) {
                    if (k < H) {   // H4-padded hbuf entries are zero, wreg beyond H is zero
                        const f32x4 h4 = *reinterpret_cast<const f32x4*>(hbuf + k);
                        acc = fmaf(wreg[k], h4[0], acc); acc = fmaf(wreg[k + 1], h4[1], acc);
                        acc = fmaf(wreg[k + 2], h4[2], acc); acc = fmaf(wreg[k + 3], h4[3], acc);
                    }
                }
            } else {
                for (int k = 0; k < H; ++k) acc = fmaf(whh[(size_t)j * H + k], hbuf[k], acc);
            }
            ghs[j] = acc;
        }
        st_lds_barrier();      // (LDS-only: __syncthreads would also wait for the next step's input projection just requested)
        if (upd) {
            const float r = st_sigmoid_fast(gr + ghs[j]);
            const float z = st_sigmoid_fast(gz + ghs[H + j]);
            const float n = st_tanh_fast(gn + r * ghs[2 * H + j]);
            const float hn = (1.0f - z) * n + z * hbuf[j];
            hbuf[j] = hn;
            a.out[((size_t)b * T + t) * a.ldo + d * H + j] = hn;
            if (a.tape) {
                float* tp = a.tape + ((((size_t)d * a.B + b) * T + t) * 4) * H + j;
                tp[0] = r; tp[H] = z; tp[2 * H] = n; tp[3 * H] = ghs[2 * H + j];
            }
            gr = nr; gz = nz; gn = nn;
        }
        st_lds_barrier();
        t = tn;
    }
}

// ---- GRU backward through time: one workgroup per (utterance, direction), thread (g, k) keeps column k of
// gate block g of W_hh in registers (the transposed mat-vec dh += W_hh^T dgh), dh carried in LDS.
struct GruBwdArgs {
    const float* dout; int ldd;       // (B, T, >= ndir*H) gradient w.r.t. out
    const float* out; int ldo;        // forward output (h_t)
    const float* tape;                // [dir][b][t][4][H]
    const float* w_hh[2];
    float* dgi[2];                    // (B, T, 3H) gradient w.r.t. gi (input projection incl. b_ih)
    float* dgh[2];                    // (B, T, 3H) gradient w.r.t. W_hh h + b_hh
    int B, T, H;
};

template <bool REG, int KQ = 32>      // REG: H <= 4 KQ, the mat-vec is a branch-free run over 4 KQ zero-padded entries (see gru_seq_quad_kernel)
__global__ __launch_bounds__(REG ? 384 : 1024) void gru_seq_bwd_kernel(const GruBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.H, H3 = 3 * H, T = a.T;
    const int H4 = REG ? 4 * KQ : ((H + 3) & ~3);
    float* dghs = lds;                 // [3][H4]  (pad entries stay zero)
    float* part = lds + 3 * H4;        // [3H]
    float* dhc = part + H3;            // [H]
    const int b = blockIdx.x, d = blockIdx.y;
    const int j = threadIdx.x;
    const bool row_ok = j < H3;
    const int g = row_ok ? j / H : 0, k = row_ok ? j - g * H : 0;
    const float* __restrict__ whh = a.w_hh[d];
    float wreg[REG ? 4 * KQ : 4];
    if (REG) {
        // branch-free: clamped address, zeroed by a 0 / 1 factor (a select on the loaded value becomes a branch around the load with a
        // full wait behind it: 4 KQ serialised round trips, a quarter of this kernel's time at T = 258)
        const float rok = row_ok ? 1.0f : 0.0f;
#pragma unroll
        for (int jj = 0; jj < 4 * KQ; ++jj) wreg[jj] = whh[(size_t)(g * H + (jj < H ? jj : H - 1)) * H + k] * (jj < H ? rok : 0.0f);
    }
    for (int i = j; i < 3 * H4; i += blockDim.x) dghs[i] = 0.0f;
    if (j < H) dhc[j] = 0.0f;
    __syncthreads();
    const bool upd = j < H;
    const float* tape = a.tape + (((size_t)d * a.B + b) * T) * 4 * H;
    const float* outb = a.out + (size_t)b * T * a.ldo + d * H;
    const float* doutb = a.dout + (size_t)b * T * a.ldd + d * H;
    float* dgi = a.dgi[d] + (size_t)b * T * H3;
    float* dgh = a.dgh[d] + (size_t)b * T * H3;
    // processing order of the forward was t = 0..T-1 (d = 0) or T-1..0 (d = 1); walk it backwards: step s handles t = T-1-s / s.
    // The six operands of a step (r, z, n, W_hn h + b_hn, h_{t-1}, dout_t) are read GB_PB steps ahead, a block at a time, with
    // unconditional loads from clamped addresses (see gru_seq_tri_kernel): a step is ~0.5 us, a first-touch load 1-2 us.
    constexpr int GB_PB = 4;
    float cur[GB_PB][6], nxt[GB_PB][6];
    const int jc = upd ? j : 0;
    auto load_block = [&](float (&dst)[GB_PB][6], int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GB_PB; ++i) {
            const int s = min(s0 + i, T - 1);
            const int tt = d ? s : T - 1 - s;
            const float* tp = tape + (size_t)tt * 4 * H + jc;
            dst[i][0] = tp[0]; dst[i][1] = tp[H]; dst[i][2] = tp[2 * H]; dst[i][3] = tp[3 * H];
            const int tprev = d ? tt + 1 : tt - 1;
            const bool inr = tprev >= 0 && tprev < T;
            dst[i][4] = outb[(size_t)(inr ? tprev : tt) * a.ldo + jc] * (inr ? 1.0f : 0.0f);        // h before the first step is zero
            dst[i][5] = doutb[(size_t)tt * a.ldd + jc];
        }
    };
    load_block(cur, 0);
    for (int s0 = 0; s0 < T; s0 += GB_PB) {
        load_block(nxt, s0 + GB_PB);
#pragma unroll
        for (int i = 0; i < GB_PB; ++i) {
            const int s = s0 + i;
            if (s >= T) break;                                        // (uniform)
            const int t = d ? s : T - 1 - s;
            float dhp_direct = 0.0f;
            if (upd) {
                const float r = cur[i][0], z = cur[i][1], n = cur[i][2], ghn = cur[i][3], hp = cur[i][4], go = cur[i][5];
                const float dh = dhc[j] + go;
                const float dn = dh * (1.0f - z), dz = dh * (hp - n);
                dhp_direct = dh * z;
                const float dnp = dn * (1.0f - n * n);
                const float dzp = dz * z * (1.0f - z);
                const float drp = dnp * ghn * r * (1.0f - r);
                float* gi_ = dgi + (size_t)t * H3 + j;
                float* gh_ = dgh + (size_t)t * H3 + j;
                gi_[0] = drp; gi_[H] = dzp; gi_[2 * H] = dnp;
                gh_[0] = drp; gh_[H] = dzp; gh_[2 * H] = dnp * r;
                dghs[j] = drp; dghs[H4 + j] = dzp; dghs[2 * H4 + j] = dnp * r;
            }
            st_lds_barrier();      // LDS-only (the next block's operand loads stay in flight)
            if (row_ok) {
                float acc = 0.0f;
                const float* dg = dghs + g * H4;
                if (REG) {
                    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                    for (int jj = 0; jj < 4 * KQ; jj += 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(dg + jj);
                        acc = fmaf(wreg[jj], v[0], acc); a1 = fmaf(wreg[jj + 1], v[1], a1);
                        a2 = fmaf(wreg[jj + 2], v[2], a2); a3 = fmaf(wreg[jj + 3], v[3], a3);
                    }
                    acc = (acc + a1) + (a2 + a3);
                } else {
                    for (int jj = 0; jj < H; ++jj) acc = fmaf(whh[(size_t)(g * H + jj) * H + k], dg[jj], acc);
                }
                part[j] = acc;
            }
            st_lds_barrier();
            if (upd) dhc[j] = dhp_direct + part[j] + part[H + j] + part[2 * H + j];
            st_lds_barrier();
        }
#pragma unroll
        for (int i = 0; i < GB_PB; ++i)
#pragma unroll
            for (int c = 0; c < 6; ++c) cur[i][c] = nxt[i][c];
    }
}

// ---- the same for H <= 84 with the transposed mat-vec sliced like gru_seq_tri_kernel: the three gate blocks of output column k live in
// three ADJACENT lanes (lane = 3u + g, column k = 21 wave + u), each with its 84 zero-padded weights W[g H + j][k] as 42 register
// pairs (v_pk_fma_f32), the three partial sums meet on lane g = 0 through two DPP shifts, and THAT lane runs the pointwise part of
// the next step: dh never goes through LDS, the step's 3 x H gate gradients do (two buffers, by step parity) -- ONE LDS-only barrier
// per step instead of three (0.77 -> see DESIGN.md us per step at H = 80).
constexpr int GB_LD = 84;
__global__ __launch_bounds__(256) void gru_seq_bwd_tri_kernel(const GruBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float dgb[2][3 * GB_LD];
    const int H = a.H, H3 = 3 * H, T = a.T;
    const int b = blockIdx.x, d = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = lane / 3, g = lane - 3 * u, k = wave * 21 + u;
    const bool unit_ok = lane < 63 && k < H;
    const int kc = unit_ok ? k : 0;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 wreg[GB_LD / 2];
    {
        const float uok = unit_ok ? 1.0f : 0.0f;
        const float* wc = a.w_hh[d] + (size_t)g * H * H + kc;           // column k of gate block g: W[g H + j][k], j = 0 .. H-1
#pragma unroll
        for (int jj = 0; jj < GB_LD; ++jj) wreg[jj >> 1][jj & 1] = wc[(size_t)(jj < H ? jj : H - 1) * H] * (jj < H ? uok : 0.0f);
    }
    for (int i = tid; i < 2 * 3 * GB_LD; i += blockDim.x) (&dgb[0][0])[i] = 0.0f;
    __syncthreads();
    const bool upd = unit_ok && g == 0;
    const float* tape = a.tape + (((size_t)d * a.B + b) * T) * 4 * H;
    const float* outb = a.out + (size_t)b * T * a.ldo + d * H;
    const float* doutb = a.dout + (size_t)b * T * a.ldd + d * H;
    float* dgi = a.dgi[d] + (size_t)b * T * H3;
    float* dgh = a.dgh[d] + (size_t)b * T * H3;
    constexpr int GB_PB = 4;
    float cur[GB_PB][6], nxt[GB_PB][6];
    const int jc = upd ? k : 0;
    auto load_block = [&](float (&dst)[GB_PB][6], int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GB_PB; ++i) {
            const int s = min(s0 + i, T - 1);
            const int tt = d ? s : T - 1 - s;
            const float* tp = tape + (size_t)tt * 4 * H + jc;
            dst[i][0] = tp[0]; dst[i][1] = tp[H]; dst[i][2] = tp[2 * H]; dst[i][3] = tp[3 * H];
            const int tprev = d ? tt + 1 : tt - 1;
            const bool inr = tprev >= 0 && tprev < T;
            dst[i][4] = outb[(size_t)(inr ? tprev : tt) * a.ldo + jc] * (inr ? 1.0f : 0.0f);
            dst[i][5] = doutb[(size_t)tt * a.ldd + jc];
        }
    };
    load_block(cur, 0);
    float dhc = 0.0f;                                   // dL/dh flowing into the step from the later ones (lane g = 0)
    for (int s0 = 0; s0 < T; s0 += GB_PB) {
        load_block(nxt, s0 + GB_PB);
#pragma unroll
        for (int i = 0; i < GB_PB; ++i) {
            const int s = s0 + i;
            if (s >= T) break;                                        // (uniform)
            const int t = d ? s : T - 1 - s;
            float* buf = dgb[s & 1];
            float dhp_direct = 0.0f;
            if (upd) {
                const float r = cur[i][0], z = cur[i][1], n = cur[i][2], ghn = cur[i][3], hp = cur[i][4], go = cur[i][5];
                const float dh = dhc + go;
                const float dn = dh * (1.0f - z), dz = dh * (hp - n);
                dhp_direct = dh * z;
                const float dnp = dn * (1.0f - n * n);
                const float dzp = dz * z * (1.0f - z);
                const float drp = dnp * ghn * r * (1.0f - r);
                float* gi_ = dgi + (size_t)t * H3 + k;
                float* gh_ = dgh + (size_t)t * H3 + k;
                gi_[0] = drp; gi_[H] = dzp; gi_[2 * H] = dnp;
                gh_[0] = drp; gh_[H] = dzp; gh_[2 * H] = dnp * r;
                buf[k] = drp; buf[GB_LD + k] = dzp; buf[2 * GB_LD + k] = dnp * r;
            }
            st_lds_barrier();      // LDS-only (the next block's operand loads stay in flight)
            const float* dg = buf + g * GB_LD;
            f32x4 hreg[GB_LD / 4];
#pragma unroll
            for (int q = 0; q < GB_LD / 4; ++q) hreg[q] = *reinterpret_cast<const f32x4*>(dg + 4 * q);
            f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < GB_LD / 4; ++q) {
                acc0 = __builtin_elementwise_fma(wreg[2 * q], f32x2{hreg[q][0], hreg[q][1]}, acc0);
                acc1 = __builtin_elementwise_fma(wreg[2 * q + 1], f32x2{hreg[q][2], hreg[q][3]}, acc1);
            }
            const float part = (acc0.x + acc0.y) + (acc1.x + acc1.y);
            const float p1 = st_dpp<0x130>(part), p2 = st_dpp<0x130>(p1);     // wave_shl:1 = the value of lane + 1: gates z, n of this column
            dhc = dhp_direct + ((part + p1) + p2);
        }
#pragma unroll
        for (int i = 0; i < GB_PB; ++i)
#pragma unroll
            for (int c = 0; c < 6; ++c) cur[i][c] = nxt[i][c];
    }
}

// ---- LSTM cell backward, pointwise part: from dh_t (sum of up to three addends, the third optionally scaled
// element-wise, then the dropout mask) and the carried dc_t to the pre-activation gate gradients and dc_{t-1}.
struct LstmPwArgs {
    const float* dh0; int ld0; const float* dh1; int ld1; const float* dh2; int ld2; const float* scale2;
    const float* mask;                 // (B, H) contiguous, may be NULL
    const float* gates;                // (B, 4, H) activated (i, f, g, o)
    const float* c; int ldc;           // c_t
    const float* c_prev; int ldcp;     // c_{t-1}, NULL = zeros
    float* dc;                         // (B, H) in: dL/dc_t carried from step t+1, out: dL/dc_{t-1}
    float* dgates; int ldg;            // (B, 4H) torch gate order
    float* dg_t16; int t16_kbs; int t16_kb0;   // optional second copy in the T16 tile layout (operand of the packed GEMM)
    int B, H;
};

__device__ __forceinline__ size_t lstm_t16_off(int b, int k, int KB) {
    return (((size_t)(b >> 4) * KB + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (b & 15)) * 4 + (k & 3);
}

__device__ __forceinline__ void lstm_bwd_pw_body(const LstmPwArgs& a) {
    const int total = a.B * a.H, H = a.H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / H, u = i - b * H;
        // all operands are requested before any of them is consumed (absent ones read a valid dummy address and are
        // replaced by their neutral value): a chain of `if (p) dh += p[i]` would wait for every load in turn
        const float* dummy = a.dh0 + (size_t)b * a.ld0 + u;
        const float l0 = dummy[0];
        const float l1 = (a.dh1 ? a.dh1 + (size_t)b * a.ld1 + u : dummy)[0];
        const float l2 = (a.dh2 ? a.dh2 + (size_t)b * a.ld2 + u : dummy)[0];
        const float ls = (a.scale2 ? a.scale2 + i : dummy)[0];
        const float lm = (a.mask ? a.mask + i : dummy)[0];
        const float* gp = a.gates + (size_t)b * 4 * H + u;
        const float gi = gp[0], gf = gp[H], gg = gp[2 * H], go = gp[3 * H];
        const float craw = a.c[(size_t)b * a.ldc + u];
        const float cp_l = (a.c_prev ? a.c_prev + (size_t)b * a.ldcp + u : dummy)[0];
        const float dc_in = a.dc[i];
        float dh = l0;
        if (a.dh1) dh += l1;
        if (a.dh2) dh += l2 * (a.scale2 ? ls : 1.0f);
        if (a.mask) dh *= lm;
        const float tc = tanhf(craw);
        const float cp = a.c_prev ? cp_l : 0.0f;
        const float dc = dc_in + dh * go * (1.0f - tc * tc);
        float* dg = a.dgates + (size_t)b * a.ldg + u;
        const float d0 = dc * gg * gi * (1.0f - gi), d1 = dc * cp * gf * (1.0f - gf);
        const float d2 = dc * gi * (1.0f - gg * gg), d3 = dh * tc * go * (1.0f - go);
        dg[0] = d0; dg[H] = d1; dg[2 * H] = d2; dg[3 * H] = d3;
        if (a.dg_t16) {
            const int k0 = a.t16_kb0 * 16 + u;
            a.dg_t16[lstm_t16_off(b, k0, a.t16_kbs)] = d0;
            a.dg_t16[lstm_t16_off(b, k0 + H, a.t16_kbs)] = d1;
            a.dg_t16[lstm_t16_off(b, k0 + 2 * H, a.t16_kbs)] = d2;
            a.dg_t16[lstm_t16_off(b, k0 + 3 * H, a.t16_kbs)] = d3;
        }
        a.dc[i] = dc * gf;
    }
}

// ---- both directions of a bidirectional LSTM layer, ALL time steps in ONE launch -------------------------------------------------------
// The per-step form above is one launch of ~5.4 us per step for 1.05 MB of recurrent weights per direction (43 launches for the text
// encoder): the launch, not the work, is the step.  Here the layer is one launch of 2 * H/4 workgroups (128 of the 256 compute units at
// H = 256), all resident at once.  Workgroup (tile, d) owns four hidden units of direction d: its 16 rows of W_hh (gates i, f, g, o of
// the four units) stay in REGISTERS for the whole sequence (H/16 floats per lane: wave w holds the k-blocks w NKB .. w NKB + NKB - 1),
// its cell states stay in registers, and a step is
//   1. poll h_{t-1} of ALL units of the direction straight out of the layer's OUTPUT tensor: the launcher fills it with a sentinel bit
//      pattern (0xFFFFFFFF, a NaN no arithmetic here produces) and every h is written exactly once with a write-through agent-scope
//      store, so "the data is the flag" -- no tags, no fences, no extra buffer (the {value, tag} granules of the decode step cost twice
//      the bytes).  Each wave re-reads its own K quarter of the B x H block (agent-scope loads) until no word is the sentinel;
//   2. RT x NKB x 4 fp32 MFMAs per wave (D[gate row][batch] += W[gate row][k] h[batch][k]), the four waves' partial sums through LDS,
//      ONE LDS-only barrier per step (two LDS buffers: a wave can run at most one step ahead of the workgroup's slowest wave);
//   3. waves 0 .. RT-1: gates + the precomputed input projection (requested before the wait), the pointwise update, h to the output.
// A step is then one cross-workgroup round trip (~2.3 us) + ~0.7 us of work instead of a launch.  A wait that does not complete
// (workgroups not co-resident: another tenant holds compute units) gives up after LP_SPINS polls: bit 1 of *status is set, the rows of
// that workgroup come out NaN, and every later wait of the launch polls once -- a starved launch ends, poisoned, instead of hanging.
constexpr unsigned LP_SENTINEL = 0xFFFFFFFFu;
constexpr int LP_SPINS = 1 << 16;
struct LpArgs {
    const float* xproj[2]; const float* w_hh[2]; const float* b_hh[2];
    float* out; int ldo; int ocol[2];
    float* gates_tape[2]; float* c_tape[2];
    int B, T, H; unsigned* status; int xcd_map;
};

__global__ __launch_bounds__(256) void lp_fill_kernel(float* out, int ldo, int ocol0, int ocol1, int H, int rows) {
    const int h4 = H >> 2, per = 2 * h4;
    const size_t total = (size_t)rows * per;
    const f32x4 s = {__uint_as_float(LP_SENTINEL), __uint_as_float(LP_SENTINEL), __uint_as_float(LP_SENTINEL), __uint_as_float(LP_SENTINEL)};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / per;
        const int c = (int)(i - row * per);
        const int col = c < h4 ? ocol0 + 4 * c : ocol1 + 4 * (c - h4);
        *reinterpret_cast<f32x4*>(out + row * ldo + col) = s;
    }
}

template <int RT, int NKB>
__global__ __launch_bounds__(256) void lstm_seq2_persist_kernel(const LpArgs a) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    typedef __attribute__((address_space(1))) unsigned gu32;
    __shared__ f32x4 red[2][4 * RT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile = blockIdx.x, d = blockIdx.y;
    if (a.xcd_map) {      // the workgroups of a direction on one half of the XCDs (workgroups go to the XCDs round-robin in launch order): the
        // h blocks a workgroup polls then come from four L2s' worth of producers instead of eight -- 3.62 -> 3.51 us per step; a quarter
        // or one XCD per direction loses (164 / 289 us per layer: two workgroups per unit, all hand-offs through one or two L2s).  Only
        // speed depends on the placement: the hand-off itself is agent-scope stores and loads.
        const int i = blockIdx.x + gridDim.x * blockIdx.y, xcd = i & 7, slot = i >> 3;
        d = xcd >> 2; tile = slot * 4 + (xcd & 3);
    }
    const int H = a.H, T = a.T, B = a.B;
    const int ar = lane & 15, q = lane >> 4;
    // MFMA A operand: lane supplies row `ar` of the 16 x 4 block = (unit tile*4 + ar/4, gate ar%4), k = 16 kb + 4 q + cc for the cc-th of
    // the four MFMAs of a k-block (the B operand uses the same k assignment, so the product is the plain one)
    f32x4 wreg[NKB];
    {
        const float* wr = a.w_hh[d] + (size_t)((ar & 3) * H + tile * 4 + (ar >> 2)) * H + 4 * q;
#pragma unroll
        for (int i = 0; i < NKB; ++i) wreg[i] = st_ld4(wr + 16 * (wave * NKB + i));
    }
    const float* hp[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) hp[rt] = a.out + (size_t)min(rt * 16 + ar, B - 1) * T * a.ldo + a.ocol[d] + 16 * (wave * NKB) + 4 * q;
    // pointwise role (waves 0 .. RT-1): batch row pb, hidden unit pu; D[row = 4 (lane >> 4) + r][col = lane & 15] -> r = gate
    const int pb = wave * 16 + (lane & 15), pu = tile * 4 + (lane >> 4);
    const bool pw = wave < RT && pb < B;
    const int pbc = min(pb, B - 1);
    float bh[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.b_hh[d]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bh[r] = a.b_hh[d][r * H + pu];
    }
    const float* xp = a.xproj[d] + (size_t)pbc * T * 4 * H + pu;
    float* op = a.out + (size_t)pbc * T * a.ldo + a.ocol[d] + pu;
    const size_t bhs = (size_t)B * H;
    float creg = 0.0f;
    bool dead = false;
    for (int s = 0; s < T; ++s) {
        const int t = d ? T - 1 - s : s, tp = d ? t + 1 : t - 1;
        // the input projection of this step does not depend on the recurrence: in flight during the wait.  EVERY wave loads (waves past
        // RT read a valid row and ignore it): with `pre = 0; if (wave < RT) load` the registers are zeroed first, and the compiler
        // drains the memory counter in front of that -- the previous step's write-through h store had to be acknowledged before the poll
        float pre[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pre[r] = xp[(size_t)t * 4 * H + r * H];
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            f32x4 x[RT][NKB];
            bool ok = false;
            const int spins = dead ? 1 : LP_SPINS;
            // (a) one CANARY word per producer workgroup and row tile of this wave's K range (its row 16 rt, first unit): a poll round
            //     of the launch is 128 KB instead of the whole B x H block per workgroup (4 MB: the rounds, not the latency, were the step)
            constexpr int NCAN = 4 * NKB * RT;
            for (int sp = 0; sp < spins; ++sp) {
                bool mine = true;
#pragma unroll
                for (int c0 = 0; c0 < NCAN; c0 += 64) {
                    const int c = min(c0 + lane, NCAN - 1);
                    const int crt = c / (4 * NKB), cj = c - crt * (4 * NKB);
                    gu32* cp = (gu32*)(a.out + ((size_t)(crt * 16) * T + tp) * a.ldo + a.ocol[d] + 16 * (wave * NKB) + 4 * cj);
                    mine = mine && __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != LP_SENTINEL;
                }
                if (__all(mine)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            // (b) the block itself with plain loads (shared through the L2 by the workgroups of an XCD).  A word that still holds the
            //     sentinel (the canary overtook a sibling store, or a line was cached half-written) sends the wave to (c)
            {
                // ALL the loads first, then the checks, and those without short-circuit operators: with `mine = mine && x != ...` after
                // each load the compiler emitted load, s_waitcnt vmcnt(0), compare, branch -- RT * NKB serialised L2 round trips per step
                // (2.2 of the 4.3 us of a step at RT = 2, NKB = 4)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int i = 0; i < NKB; ++i) x[rt][i] = st_ld4(hp[rt] + (size_t)tp * a.ldo + 16 * i);
                }
                unsigned bad = 0;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int i = 0; i < NKB; ++i) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) bad |= (unsigned)(__float_as_uint(x[rt][i][c]) == LP_SENTINEL);
                    }
                }
                ok = __all(bad == 0);
            }
            // (c) agent-scope re-reads of the block until it is complete
            for (int sp = 0; sp < spins && !ok; ++sp) {
                unsigned bad = 0;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int i = 0; i < NKB; ++i) {
                        gu64* gp = (gu64*)(hp[rt] + (size_t)tp * a.ldo + 16 * i);
                        const unsigned long long lo = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long hi = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned u0 = (unsigned)lo, u1 = (unsigned)(lo >> 32), u2 = (unsigned)hi, u3 = (unsigned)(hi >> 32);
                        bad |= (unsigned)(u0 == LP_SENTINEL) | (unsigned)(u1 == LP_SENTINEL) | (unsigned)(u2 == LP_SENTINEL) | (unsigned)(u3 == LP_SENTINEL);
                        x[rt][i] = f32x4{__uint_as_float(u0), __uint_as_float(u1), __uint_as_float(u2), __uint_as_float(u3)};
                    }
                }
                ok = __all(bad == 0);
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && !dead) {
                dead = true;       // (a word still holding the sentinel is a NaN: the rows of this workgroup are poisoned from here on)
                if (lane == 0 && a.status) __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int i = 0; i < NKB; ++i) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[i][cc], x[rt][i][cc], acc[rt], 0, 0, 0);
                }
            }
        }
        f32x4* rb = red[s & 1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) rb[(wave * RT + rt) * 64 + lane] = acc[rt];
        st_lds_barrier();
        if (wave < RT) {
            f32x4 g4 = rb[wave * 64 + lane];                               // (wave 0's partial of row tile `wave`)
#pragma unroll
            for (int w = 1; w < 4; ++w) { const f32x4 p = rb[(w * RT + wave) * 64 + lane]; g4[0] += p[0]; g4[1] += p[1]; g4[2] += p[2]; g4[3] += p[3]; }
            float g[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r] = g4[r] + (bh[r] + pre[r]);
            const float gi = st_sigmoid_fast(g[0]), gf = st_sigmoid_fast(g[1]), gg = st_tanh_fast(g[2]), go = st_sigmoid_fast(g[3]);
            const float c2 = gf * creg + gi * gg;
            float h2 = go * st_tanh_fast(c2);
            if (h2 != h2) h2 = __uint_as_float(0x7FC00000u);      // (a NaN must not look like the sentinel: the consumers would wait for it)
            creg = c2;
            if (pw) {
                __hip_atomic_store((gu32*)(op + (size_t)t * a.ldo), __float_as_uint(h2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.c_tape[d]) a.c_tape[d][(size_t)t * bhs + (size_t)pb * H + pu] = c2;
                if (a.gates_tape[d]) {
                    float* gp = a.gates_tape[d] + (size_t)t * 4 * bhs + (size_t)pb * 4 * H + pu;
                    gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void lstm_bwd_pw_kernel(const LstmPwArgs a) { lstm_bwd_pw_body(a); }
// two cells of the same shape in one launch (blockIdx.y picks the job): both directions of a bidirectional layer
__global__ __launch_bounds__(256) void lstm_bwd_pw_pair_kernel(const LstmPwArgs a0, const LstmPwArgs a1) {
    if (blockIdx.y == 0) lstm_bwd_pw_body(a0); else lstm_bwd_pw_body(a1);
}

}  // namespace

#ifdef GRU_STAMPS
extern "C" int st_gru_debug_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gru_stamps), sizeof(gru_stamps)); }
#endif

extern "C" int st_gru_seq_fwd(const float* gi_fwd, const float* gi_bwd, const float* w_hh_fwd, const float* w_hh_bwd,
                              const float* b_hh_fwd, const float* b_hh_bwd, float* out, int ldo,
                              float* tape, int B, int T, int H, int ndir, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(ndir == 1 || ndir == 2, "st_gru_seq_fwd: ndir=%d", ndir);
    ST_CHECK_ARG(gi_fwd && w_hh_fwd && b_hh_fwd && out && B > 0 && T > 0 && H > 0, "st_gru_seq_fwd: bad arguments");
    ST_CHECK_ARG(ndir == 1 || (gi_bwd && w_hh_bwd && b_hh_bwd), "st_gru_seq_fwd: missing reverse-direction pointers");
    ST_CHECK_ARG(3 * H <= 1024, "st_gru_seq_fwd: H=%d too large (3H must be <= 1024)", H);
    ST_CHECK_ARG(ldo >= ndir * H, "st_gru_seq_fwd: ldo=%d < ndir*H", ldo);
    GruArgs a;
    a.gi[0] = gi_fwd; a.gi[1] = gi_bwd; a.w_hh[0] = w_hh_fwd; a.w_hh[1] = w_hh_bwd;
    a.b_hh[0] = b_hh_fwd; a.b_hh[1] = b_hh_bwd; a.out = out; a.ldo = ldo; a.B = B; a.T = T; a.H = H; a.tape = tape;
    const int threads = ((3 * H + 63) / 64) * 64;
    const size_t lds = (size_t)(((H + 3) & ~3) + 3 * H) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (H <= GRU_HMAX) {
        const int qthreads = ((4 * H + 63) / 64) * 64;             // <= 512
        if (H <= 3 * GRU_SL && H <= 84) hipLaunchKernelGGL(gru_seq_tri_kernel, dim3(B, ndir), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(gru_seq_quad_kernel<32>, dim3(B, ndir), dim3(qthreads), (size_t)2 * 128 * sizeof(float), st, a);
    } else hipLaunchKernelGGL((gru_seq_kernel<false>), dim3(B, ndir), dim3(threads), lds, st, a);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_lstm_seq_fwd(const float* xproj, const float* w_hh, const float* b_hh, float* out, int ldo, int ocol,
                               float* ws, float* gates_tape, float* c_tape, int B, int T, int H, int reverse, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(xproj && w_hh && out && ws && B > 0 && T > 0 && H > 0, "st_lstm_seq_fwd: bad arguments");
    ST_CHECK_ARG(ldo >= ocol + H, "st_lstm_seq_fwd: ldo=%d < ocol+H", ldo);
    hipStream_t st = (hipStream_t)stream;
    const size_t bh = (size_t)B * H;
    float* cbuf[2] = {ws, ws + bh};
    float* zero = ws + 2 * bh;
    ST_HIP(hipMemsetAsync(zero, 0, bh * sizeof(float), st));
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? T - 1 - s : s;
        const int tp = reverse ? t + 1 : t - 1;
        st_seg seg;
        seg.w = w_hh; seg.ldw = H; seg.k = H;
        if (s == 0) { seg.x = zero; seg.ldx = H; }
        else { seg.x = out + (size_t)tp * ldo + ocol; seg.ldx = T * ldo; }
        // training keeps every c_t and the activated gates (tapes indexed by t, not by processing order)
        const float* cprev = s == 0 ? zero : (c_tape ? c_tape + (size_t)tp * bh : cbuf[(s - 1) & 1]);
        float* cnew = c_tape ? c_tape + (size_t)t * bh : cbuf[s & 1];
        int rc = st_lstm_cell_fwd(&seg, 1, nullptr, b_hh, xproj + (size_t)t * 4 * H, T * 4 * H,
                                  cprev, H, nullptr, out + (size_t)t * ldo + ocol, T * ldo,
                                  cnew, H, gates_tape ? gates_tape + (size_t)t * 4 * bh : nullptr, B, H, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int st_lstm_cell_pair_fwd(const st_seg* segs2, const float* const* b_hh2, const float* const* pre2, int ldpre,
                                     const float* const* c_prev2, int ldc_prev, float* const* h_out2, int ldh,
                                     float* const* c_out2, int ldc, float* const* gates_out2, int B, int H, void* stream);

// Both directions of a bidirectional LSTM layer, one launch per time step for the two of them (direction 0 walks t = s,
// direction 1 walks t = T-1-s).  Arguments as st_lstm_seq_fwd, in arrays of two; ws: 6*B*H floats.
extern "C" int st_lstm_seq2_fwd(const float* const* xproj2, const float* const* w_hh2, const float* const* b_hh2, float* out, int ldo,
                                const int* ocol2, float* ws, float* const* gates_tape2, float* const* c_tape2,
                                int B, int T, int H, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(xproj2 && w_hh2 && out && ws && ocol2 && xproj2[0] && xproj2[1] && w_hh2[0] && w_hh2[1] && B > 0 && T > 0 && H > 0,
                 "st_lstm_seq2_fwd: bad arguments");
    ST_CHECK_ARG(ldo >= ocol2[0] + H && ldo >= ocol2[1] + H, "st_lstm_seq2_fwd: ldo=%d too small", ldo);
    hipStream_t st = (hipStream_t)stream;
    const size_t bh = (size_t)B * H;
    float* zero = ws + 4 * bh;
    ST_HIP(hipMemsetAsync(zero, 0, bh * sizeof(float), st));
    for (int s = 0; s < T; ++s) {
        st_seg seg[2];
        const float *bh2[2], *pre[2], *cprev[2];
        float *hout[2], *cnew[2], *gout[2];
        for (int d = 0; d < 2; ++d) {
            const int t = d ? T - 1 - s : s, tp = d ? t + 1 : t - 1;
            float* cbuf[2] = {ws + (size_t)(2 * d) * bh, ws + (size_t)(2 * d + 1) * bh};
            float* ct = c_tape2 ? c_tape2[d] : nullptr;
            float* gt = gates_tape2 ? gates_tape2[d] : nullptr;
            seg[d].w = w_hh2[d]; seg[d].ldw = H; seg[d].k = H;
            if (s == 0) { seg[d].x = zero; seg[d].ldx = H; }
            else { seg[d].x = out + (size_t)tp * ldo + ocol2[d]; seg[d].ldx = T * ldo; }
            bh2[d] = b_hh2 ? b_hh2[d] : nullptr;
            pre[d] = xproj2[d] + (size_t)t * 4 * H;
            cprev[d] = s == 0 ? zero : (ct ? ct + (size_t)tp * bh : cbuf[(s - 1) & 1]);
            cnew[d] = ct ? ct + (size_t)t * bh : cbuf[s & 1];
            hout[d] = out + (size_t)t * ldo + ocol2[d];
            gout[d] = gt ? gt + (size_t)t * 4 * bh : nullptr;
        }
        int rc = st_lstm_cell_pair_fwd(seg, bh2, pre, T * 4 * H, cprev, H, hout, T * ldo, cnew, H, gout, B, H, stream);
        if (rc) return rc;
    }
    return 0;
}

// The whole bidirectional layer as ONE launch (lstm_seq2_persist_kernel): shapes it takes
extern "C" int st_lstm_seq2_persist_supported(int B, int T, int H, int ldo, int ocol0, int ocol1) {
    const int nkb = H / 64;
    return B > 0 && B <= 64 && T > 0 && H > 0 && H % 64 == 0 && (nkb == 1 || nkb == 2 || nkb == 4 || nkb == 8) &&
           ldo % 4 == 0 && ocol0 % 4 == 0 && ocol1 % 4 == 0 && ocol0 >= 0 && ocol1 >= 0 && ldo >= ocol0 + H && ldo >= ocol1 + H &&
           (ocol0 + H <= ocol1 || ocol1 + H <= ocol0) && 2 * (H / 4) <= st_device_cus();
}

template <int RT>
static void lp_launch(const LpArgs& a, hipStream_t st) {
    const dim3 grid(a.H / 4, 2), block(256);
    switch (a.H / 64) {
        case 1: hipLaunchKernelGGL((lstm_seq2_persist_kernel<RT, 1>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((lstm_seq2_persist_kernel<RT, 2>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((lstm_seq2_persist_kernel<RT, 4>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((lstm_seq2_persist_kernel<RT, 8>), grid, block, 0, st, a); break;
    }
}

extern "C" int st_lstm_seq2_persist_fwd(const float* const* xproj2, const float* const* w_hh2, const float* const* b_hh2, float* out, int ldo,
                                        const int* ocol2, float* const* gates_tape2, float* const* c_tape2,
                                        int B, int T, int H, unsigned* status, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(xproj2 && w_hh2 && out && ocol2 && xproj2[0] && xproj2[1] && w_hh2[0] && w_hh2[1], "st_lstm_seq2_persist_fwd: bad arguments");
    ST_CHECK_ARG(st_lstm_seq2_persist_supported(B, T, H, ldo, ocol2[0], ocol2[1]),
                 "st_lstm_seq2_persist_fwd: unsupported shape B=%d T=%d H=%d ldo=%d (see st_lstm_seq2_persist_supported)", B, T, H, ldo);
    ST_CHECK_ARG(st_aligned16(out) && st_aligned16(w_hh2[0]) && st_aligned16(w_hh2[1]), "st_lstm_seq2_persist_fwd: out / w_hh must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    LpArgs a;
    for (int d = 0; d < 2; ++d) {
        a.xproj[d] = xproj2[d]; a.w_hh[d] = w_hh2[d]; a.b_hh[d] = b_hh2 ? b_hh2[d] : nullptr; a.ocol[d] = ocol2[d];
        a.gates_tape[d] = gates_tape2 ? gates_tape2[d] : nullptr; a.c_tape[d] = c_tape2 ? c_tape2[d] : nullptr;
    }
    a.out = out; a.ldo = ldo; a.B = B; a.T = T; a.H = H; a.status = status;
    a.xcd_map = (H / 4) % 4 == 0 ? 1 : 0;
    const size_t pieces = (size_t)B * T * (H / 2);
    hipLaunchKernelGGL(lp_fill_kernel, dim3((unsigned)((pieces + 255) / 256 < 1024 ? (pieces + 255) / 256 : 1024)), dim3(256), 0, st,
                       out, ldo, ocol2[0], ocol2[1], H, B * T);
    ST_LAUNCH_CHECK();
    const int rt = (B + 15) / 16;
    if (rt == 1) lp_launch<1>(a, st); else if (rt == 2) lp_launch<2>(a, st); else if (rt == 3) lp_launch<3>(a, st); else lp_launch<4>(a, st);
    ST_LAUNCH_CHECK();
    return 0;
}

// ---- BPTT of both directions of a bidirectional LSTM layer, ALL time steps in ONE launch ----------------------------------------------------
// The per-step form (st_lstm_seq2_bwd_packed) is one launch of ~7.4 us per step: dh_rec = dgates(t+1) . W_hh with the pointwise backward
// of step t in its epilogue.  Here the loop is one launch of (H/16) x ceil(B/RG) x 2 workgroups of eight waves, all resident at once.
// Workgroup (tile, bg, d) owns 16 hidden units and RG = 8 or 16 batch rows of direction d: the 4H x 16 slice of W_hh it multiplies by stays in
// REGISTERS (wave w holds the k-blocks w NKB .. w NKB + NKB - 1 of K = 4H), the carried dL/dc of its 16 x 16 cells stays in
// registers, and a step is
//   1. poll the gate gradients of the step processed before (all 4H of them, for the workgroup's 16 batch rows) straight out of the
//      dxproj tensor the layer returns: the launcher fills it with the sentinel bit pattern and every word is written exactly once with
//      an agent-scope store -- the data is the flag, exactly as in lstm_seq2_persist_kernel (canary word per producer, then the block
//      with plain loads, then agent-scope re-reads of what is still missing);
//   2. NKB x 4 fp32 MFMAs per wave (D[unit][batch] += W_hh[k][unit] dgates[batch][k]), the eight waves' partial sums through LDS, one
//      LDS-only barrier per step (two buffers);
//   3. waves 0 .. 3: dh = dout_t + dh_rec, the pointwise backward of one cell per lane (operands requested before the wait), the four
//      gate gradients to dxproj.
// What a step exchanges is 4H floats per batch row (the forward exchanges H): 64 KB per workgroup and step at H = 256, spread over eight
// waves that each issue the same eight 16-byte loads a wave of the forward kernel does.
namespace {
struct LbArgs {
    const float* dout; int ldd; int dcol[2];
    const float* gates_tape[2]; const float* c_tape[2]; const float* w_hh[2];
    float* dxp[2];
    int B, T, H; unsigned* status;
};

template <int NKB, int RG>
__global__ __launch_bounds__(512) void lstm_seq2_bwd_persist_kernel(const LbArgs a) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    typedef __attribute__((address_space(1))) unsigned gu32;
    __shared__ f32x4 red[2][8 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, bg = blockIdx.y, d = blockIdx.z;
    const int H = a.H, T = a.T, B = a.B, H4 = 4 * a.H;
    const int ar = lane & 15, q = lane >> 4;
    // MFMA A operand: lane supplies row `ar` (unit tile*16 + ar) of the 16 x 4 block, k = 16 kb + 4 q + cc: W_hh[k][unit]
    f32x4 wreg[NKB];
#pragma unroll
    for (int i = 0; i < NKB; ++i) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) wreg[i][cc] = a.w_hh[d][(size_t)(16 * (wave * NKB + i) + 4 * q + cc) * H + tile * 16 + ar];
    }
    // B operand: column `ar` = batch row bg*16 + ar (rows past B read row B - 1; their columns of D are dropped), the same k
    // (RG = 8: eight batch rows per workgroup -- columns 8 .. 15 of the MFMA repeat rows 0 .. 7 (the same addresses: no extra traffic) and
    // are dropped; twice the workgroups, half the gate gradients taken in per workgroup and step)
    const int rr = ar & (RG - 1);
    const float* xp = a.dxp[d] + (size_t)min(bg * RG + rr, B - 1) * T * H4 + 16 * (wave * NKB) + 4 * q;
    // pointwise role (waves 0 .. 3): D[row = 4 (lane >> 4) + r][col = lane & 15]; wave w takes r = w: unit ju of batch row pb
    const int pb = bg * RG + rr, ju = tile * 16 + 4 * q + (wave & 3);
    const bool pw = pb < B && ar < RG;
    const int pbc = min(pb, B - 1);
    const size_t bhs = (size_t)B * H;
    const float* gp = a.gates_tape[d] + (size_t)pbc * 4 * H + ju;
    const float* cp_ = a.c_tape[d] + (size_t)pbc * H + ju;
    const float* dop = a.dout + (size_t)pbc * T * a.ldd + a.dcol[d] + ju;
    float* dgp = a.dxp[d] + (size_t)pbc * T * H4 + ju;
    float dcreg = 0.0f;
    bool dead = false;
    for (int s = T - 1; s >= 0; --s) {                        // s = processing index of the forward
        const int t = d ? T - 1 - s : s, tp = d ? t + 1 : t - 1, tn = d ? t - 1 : t + 1;
        // the tapes and dout of this step do not depend on the recurrence: in flight during the wait
        float g4[4], c4, cp4, do4;
        if (wave < 4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) g4[g] = gp[(size_t)t * 4 * bhs + g * H];
            c4 = cp_[(size_t)t * bhs];
            cp4 = cp_[(size_t)(s > 0 ? tp : t) * bhs];
            do4 = dop[(size_t)t * a.ldd];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (s < T - 1) {
            f32x4 x[NKB];
            bool ok = false;
            const int spins = dead ? 1 : LP_SPINS;
            // (a) one canary word per producer workgroup of this wave's K range (batch row bg*16, first unit of the block)
            for (int sp = 0; sp < spins; ++sp) {
                const int c = min(lane, NKB - 1);
                gu32* cw = (gu32*)(a.dxp[d] + ((size_t)(bg * RG) * T + tn) * H4 + 16 * (wave * NKB + c));
                const bool mine = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != LP_SENTINEL;
                if (__all(mine)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            // (b) the block itself with plain loads; a word that still holds the sentinel sends the wave to (c)
            {
#pragma unroll
                for (int i = 0; i < NKB; ++i) x[i] = st_ld4(xp + (size_t)tn * H4 + 16 * i);
                unsigned bad = 0;
#pragma unroll
                for (int i = 0; i < NKB; ++i) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) bad |= (unsigned)(__float_as_uint(x[i][c]) == LP_SENTINEL);
                }
                ok = __all(bad == 0);
            }
            // (c) agent-scope re-reads of the block until it is complete
            for (int sp = 0; sp < spins && !ok; ++sp) {
                unsigned bad = 0;
#pragma unroll
                for (int i = 0; i < NKB; ++i) {
                    gu64* g64 = (gu64*)(xp + (size_t)tn * H4 + 16 * i);
                    const unsigned long long lo = __hip_atomic_load(g64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long hi = __hip_atomic_load(g64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned u0 = (unsigned)lo, u1 = (unsigned)(lo >> 32), u2 = (unsigned)hi, u3 = (unsigned)(hi >> 32);
                    bad |= (unsigned)(u0 == LP_SENTINEL) | (unsigned)(u1 == LP_SENTINEL) | (unsigned)(u2 == LP_SENTINEL) | (unsigned)(u3 == LP_SENTINEL);
                    x[i] = f32x4{__uint_as_float(u0), __uint_as_float(u1), __uint_as_float(u2), __uint_as_float(u3)};
                }
                ok = __all(bad == 0);
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && !dead) {
                dead = true;       // (a word still holding the sentinel is a NaN: the rows of this workgroup are poisoned from here on)
                if (lane == 0 && a.status) __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int i = 0; i < NKB; ++i) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[i][cc], x[i][cc], acc, 0, 0, 0);
            }
        }
        float* rb = reinterpret_cast<float*>(red[s & 1]);          // [wave][r][lane]: written and read without bank conflicts
#pragma unroll
        for (int r = 0; r < 4; ++r) rb[(wave * 4 + r) * 64 + lane] = acc[r];
        st_lds_barrier();
        if (wave < 4) {
            float dh = do4;
#pragma unroll
            for (int w = 0; w < 8; ++w) dh += rb[(w * 4 + wave) * 64 + lane];
            const float gi = g4[0], gf = g4[1], gg = g4[2], go = g4[3];
            const float tc = st_tanh_fast(c4);                          // (the function the forward applied to this c)
            const float dc = dcreg + dh * go * (1.0f - tc * tc);
            float dg[4];
            dg[0] = dc * gg * gi * (1.0f - gi);
            dg[1] = s > 0 ? dc * cp4 * gf * (1.0f - gf) : 0.0f;
            dg[2] = dc * gi * (1.0f - gg * gg);
            dg[3] = dh * tc * go * (1.0f - go);
            dcreg = dc * gf;
            if (pw) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float v = dg[g];
                    const unsigned u = v != v ? 0x7FC00000u : __float_as_uint(v);      // (a NaN must not look like the sentinel)
                    __hip_atomic_store((gu32*)(dgp + (size_t)t * H4 + g * H), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

}  // namespace

extern "C" int st_lstm_seq2_bwd_persist_supported(int B, int T, int H, int ldd, int dcol0, int dcol1) {
    const int nkb = H / 32;
    return B > 0 && T > 0 && H > 0 && H % 32 == 0 && (nkb == 1 || nkb == 2 || nkb == 4 || nkb == 8) && ldd % 4 == 0 && dcol0 % 4 == 0 &&
           dcol1 % 4 == 0 && dcol0 >= 0 && dcol1 >= 0 && ldd >= dcol0 + H && ldd >= dcol1 + H && 2 * (H / 16) * ((B + 15) / 16) <= st_device_cus();
}

extern "C" int st_lstm_seq2_bwd_persist(const float* dout, int ldd, const int* dcol2, const float* const* gates_tape2, const float* const* c_tape2,
                                        const float* const* w_hh2, float* const* dxproj2, int B, int T, int H, unsigned* status, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dout && dcol2 && gates_tape2 && c_tape2 && w_hh2 && dxproj2 && gates_tape2[0] && gates_tape2[1] && c_tape2[0] && c_tape2[1] &&
                 w_hh2[0] && w_hh2[1] && dxproj2[0] && dxproj2[1], "st_lstm_seq2_bwd_persist: bad arguments");
    ST_CHECK_ARG(st_lstm_seq2_bwd_persist_supported(B, T, H, ldd, dcol2[0], dcol2[1]),
                 "st_lstm_seq2_bwd_persist: unsupported shape B=%d T=%d H=%d ldd=%d (see st_lstm_seq2_bwd_persist_supported)", B, T, H, ldd);
    ST_CHECK_ARG(st_aligned16(dout) && st_aligned16(gates_tape2[0]) && st_aligned16(gates_tape2[1]) && st_aligned16(c_tape2[0]) &&
                 st_aligned16(c_tape2[1]) && st_aligned16(dxproj2[0]) && st_aligned16(dxproj2[1]), "st_lstm_seq2_bwd_persist: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    LbArgs a;
    for (int d = 0; d < 2; ++d) {
        a.dcol[d] = dcol2[d]; a.gates_tape[d] = gates_tape2[d]; a.c_tape[d] = c_tape2[d]; a.w_hh[d] = w_hh2[d]; a.dxp[d] = dxproj2[d];
    }
    // every word = LP_SENTINEL (one fill when the two tensors lie back to back: ops.lstm_seq2_bwd allocates them so)
    const size_t words = (size_t)B * T * 4 * H;
    if (dxproj2[1] == dxproj2[0] + words) ST_HIP(hipMemsetAsync(dxproj2[0], 0xFF, 2 * words * sizeof(float), st));
    else {
        ST_HIP(hipMemsetAsync(dxproj2[0], 0xFF, words * sizeof(float), st));
        ST_HIP(hipMemsetAsync(dxproj2[1], 0xFF, words * sizeof(float), st));
    }
    a.dout = dout; a.ldd = ldd; a.B = B; a.T = T; a.H = H; a.status = status;
    // eight batch rows per workgroup while the workgroups fill at most half the device (the hand-off is bound by what a workgroup takes in
    // per step: 220 -> 192 us per C2 layer; four rows -- every compute unit of the device -- gave 188 and leaves no room to be resident)
    const int rg = 2 * (H / 16) * ((B + 7) / 8) <= st_device_cus() / 2 ? 8 : 16;
    const dim3 grid(H / 16, (B + rg - 1) / rg, 2), block(512);
#define LB_LAUNCH(NKB_) do { \
        if (rg == 8) hipLaunchKernelGGL((lstm_seq2_bwd_persist_kernel<NKB_, 8>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((lstm_seq2_bwd_persist_kernel<NKB_, 16>), grid, block, 0, st, a); } while (0)
    switch (H / 32) {
        case 1: LB_LAUNCH(1); break;
        case 2: LB_LAUNCH(2); break;
        case 4: LB_LAUNCH(4); break;
        default: LB_LAUNCH(8); break;
    }
#undef LB_LAUNCH
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_lstm_cell_bwd_pointwise(const float* dh0, int ld0, const float* dh1, int ld1, const float* dh2, int ld2,
                                          const float* scale2, const float* mask, const float* gates, const float* c, int ldc,
                                          const float* c_prev, int ldcp, float* dc, float* dgates, int ldg,
                                          const st_t16_view* dgates_t16, int B, int H, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dh0 && gates && c && dc && dgates && B > 0 && H > 0, "st_lstm_cell_bwd_pointwise: bad arguments");
    LstmPwArgs a;
    a.dh0 = dh0; a.ld0 = ld0; a.dh1 = dh1; a.ld1 = ld1; a.dh2 = dh2; a.ld2 = ld2; a.scale2 = scale2; a.mask = mask;
    a.gates = gates; a.c = c; a.ldc = ldc; a.c_prev = c_prev; a.ldcp = ldcp; a.dc = dc; a.dgates = dgates; a.ldg = ldg;
    a.dg_t16 = dgates_t16 ? dgates_t16->base : nullptr;
    a.t16_kbs = dgates_t16 ? dgates_t16->kb_stride : 0; a.t16_kb0 = dgates_t16 ? dgates_t16->kb0 : 0;
    a.B = B; a.H = H;
    const int blocks = (B * H + 255) / 256;
    hipLaunchKernelGGL(lstm_bwd_pw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_lstm_seq_bwd(const float* dout, int ldd, int dcol, const float* gates_tape, const float* c_tape,
                               const float* w_hh_t, float* dxproj, float* ws, int B, int T, int H, int reverse, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dout && gates_tape && c_tape && w_hh_t && dxproj && ws && B > 0 && T > 0 && H > 0, "st_lstm_seq_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t bh = (size_t)B * H;
    float* dc = ws;            // carried dL/dc
    float* dhrec = ws + bh;    // dL/dh_{t-1} through W_hh
    ST_HIP(hipMemsetAsync(ws, 0, 2 * bh * sizeof(float), st));
    for (int s = T - 1; s >= 0; --s) {                  // s = processing index of the forward
        const int t = reverse ? T - 1 - s : s;
        const int tp = reverse ? t + 1 : t - 1;
        float* dg = dxproj + (size_t)t * 4 * H;          // row stride T*4H
        int rc = st_lstm_cell_bwd_pointwise(dout + (size_t)t * ldd + dcol, T * ldd, dhrec, H, nullptr, 0, nullptr, nullptr,
                                            gates_tape + (size_t)t * 4 * bh, c_tape + (size_t)t * bh, H,
                                            s == 0 ? nullptr : c_tape + (size_t)tp * bh, H, dc, dg, T * 4 * H, nullptr, B, H, stream);
        if (rc) return rc;
        if (s == 0) break;
        st_seg seg;
        seg.x = dg; seg.ldx = T * 4 * H; seg.w = w_hh_t; seg.ldw = 4 * H; seg.k = 4 * H;
        rc = st_skinny_linear_fwd(&seg, 1, nullptr, ST_ACT_NONE, nullptr, 0, dhrec, H, 0, nullptr, 0, 0, B, H, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int st_skinny_linear_pair_fwd(const st_seg* segs2, float* const* y2, int ldy, int B, int N, void* stream);

// Backward through time of both directions of a bidirectional LSTM layer: per time step ONE pointwise launch and ONE W_hh^T
// launch serve the two directions (st_lstm_seq_bwd: two launches per step and direction).  ws: 4*B*H floats.
extern "C" int st_lstm_seq2_bwd(const float* dout, int ldd, const int* dcol2, const float* const* gates_tape2, const float* const* c_tape2,
                                const float* const* w_hh_t2, float* const* dxproj2, float* ws, int B, int T, int H, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dout && dcol2 && gates_tape2 && c_tape2 && w_hh_t2 && dxproj2 && ws && B > 0 && T > 0 && H > 0, "st_lstm_seq2_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t bh = (size_t)B * H;
    ST_HIP(hipMemsetAsync(ws, 0, 4 * bh * sizeof(float), st));
    const int blocks = (B * H + 255) / 256;
    for (int s = T - 1; s >= 0; --s) {                  // s = processing index of the forward
        LstmPwArgs a[2];
        st_seg seg[2];
        float* dh2[2];
        for (int d = 0; d < 2; ++d) {
            const int t = d ? T - 1 - s : s, tp = d ? t + 1 : t - 1;
            float* dc = ws + (size_t)(2 * d) * bh;
            float* dhrec = ws + (size_t)(2 * d + 1) * bh;
            float* dg = dxproj2[d] + (size_t)t * 4 * H;          // row stride T*4H
            memset(&a[d], 0, sizeof(LstmPwArgs));
            a[d].dh0 = dout + (size_t)t * ldd + dcol2[d]; a[d].ld0 = T * ldd; a[d].dh1 = dhrec; a[d].ld1 = H;
            a[d].gates = gates_tape2[d] + (size_t)t * 4 * bh; a[d].c = c_tape2[d] + (size_t)t * bh; a[d].ldc = H;
            a[d].c_prev = s == 0 ? nullptr : c_tape2[d] + (size_t)tp * bh; a[d].ldcp = H;
            a[d].dc = dc; a[d].dgates = dg; a[d].ldg = T * 4 * H; a[d].B = B; a[d].H = H;
            seg[d].x = dg; seg[d].ldx = T * 4 * H; seg[d].w = w_hh_t2[d]; seg[d].ldw = 4 * H; seg[d].k = 4 * H;
            dh2[d] = dhrec;
        }
        hipLaunchKernelGGL(lstm_bwd_pw_pair_kernel, dim3(blocks, 2), dim3(256), 0, st, a[0], a[1]);
        ST_LAUNCH_CHECK();
        if (s == 0) break;
        int rc = st_skinny_linear_pair_fwd(seg, dh2, H, B, H, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int st_lstm_seq2_bwd_packed(const float* dout, int ldd, const int* dcol2, const float* const* gates_tape2, const float* const* c_tape2,
                                       const float* const* w_hh_t_p16_2, float* const* dxproj2, float* ws, float* dg_t16_ws,
                                       int B, int T, int H, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dout && dcol2 && gates_tape2 && c_tape2 && w_hh_t_p16_2 && dxproj2 && ws && dg_t16_ws && B > 0 && T > 0 && H > 0 && H % 16 == 0,
                 "st_lstm_seq2_bwd_packed: bad arguments (H must be a multiple of 16)");
    hipStream_t st = (hipStream_t)stream;
    const size_t bh = (size_t)B * H;
    const size_t t16 = st_t16_floats(B, 4 * H);
    const int kbs = (4 * H + 15) >> 4;
    ST_HIP(hipMemsetAsync(ws, 0, 2 * bh * sizeof(float), st));
    ST_HIP(hipMemsetAsync(dg_t16_ws, 0, 4 * t16 * sizeof(float), st));       // (rows past B of the tiles stay zero)
    auto buf = [&](int d, int par) { return dg_t16_ws + (size_t)(2 * d + par) * t16; };
    const int blocks = (B * H + 255) / 256;
    {   // the last processed step: nothing recurrent arrives yet
        const int s = T - 1;
        LstmPwArgs a[2];
        for (int d = 0; d < 2; ++d) {
            const int t = d ? T - 1 - s : s, tp = d ? t + 1 : t - 1;
            memset(&a[d], 0, sizeof(LstmPwArgs));
            a[d].dh0 = dout + (size_t)t * ldd + dcol2[d]; a[d].ld0 = T * ldd;
            a[d].gates = gates_tape2[d] + (size_t)t * 4 * bh; a[d].c = c_tape2[d] + (size_t)t * bh; a[d].ldc = H;
            a[d].c_prev = s == 0 ? nullptr : c_tape2[d] + (size_t)tp * bh; a[d].ldcp = H;
            a[d].dc = ws + (size_t)d * bh; a[d].dgates = dxproj2[d] + (size_t)t * 4 * H; a[d].ldg = T * 4 * H;
            a[d].dg_t16 = buf(d, s & 1); a[d].t16_kbs = kbs; a[d].t16_kb0 = 0;
            a[d].B = B; a[d].H = H;
        }
        hipLaunchKernelGGL(lstm_bwd_pw_pair_kernel, dim3(blocks, 2), dim3(256), 0, st, a[0], a[1]);
        ST_LAUNCH_CHECK();
    }
    for (int s = T - 1; s >= 1; --s) {        // dh_rec(s-1) = dgates(s) . W_hh, and the pointwise backward of step s-1 in the epilogue
        st_t16_view xv[2];
        st_lstm_pw_job job[2];
        for (int d = 0; d < 2; ++d) {
            const int s1 = s - 1;
            const int t = d ? T - 1 - s1 : s1, tp = d ? t + 1 : t - 1;
            xv[d].base = buf(d, s & 1); xv[d].kb_stride = kbs; xv[d].kb0 = 0;
            memset(&job[d], 0, sizeof(st_lstm_pw_job));
            job[d].n0 = 0; job[d].H = H;
            job[d].dh1 = dout + (size_t)t * ldd + dcol2[d]; job[d].ld1 = T * ldd;
            job[d].gates = gates_tape2[d] + (size_t)t * 4 * bh; job[d].c = c_tape2[d] + (size_t)t * bh; job[d].ldc = H;
            job[d].c_prev = s1 == 0 ? nullptr : c_tape2[d] + (size_t)tp * bh; job[d].ldcp = H;
            job[d].dc = ws + (size_t)d * bh; job[d].dgates = dxproj2[d] + (size_t)t * 4 * H; job[d].ldg = T * 4 * H;
            job[d].dgates_t16.base = buf(d, s1 & 1); job[d].dgates_t16.kb_stride = kbs; job[d].dgates_t16.kb0 = 0;
        }
        int rc = st_skinny_linear_packed_lstm_bwd_pair_fwd(w_hh_t_p16_2, xv, 4 * H, nullptr, H, B, H, job, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int st_gru_seq_bwd(const float* dout, int ldd, const float* out, int ldo, const float* tape,
                              const float* w_hh_fwd, const float* w_hh_bwd, float* dgi_fwd, float* dgi_bwd,
                              float* dgh_fwd, float* dgh_bwd, int B, int T, int H, int ndir, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(ndir == 1 || ndir == 2, "st_gru_seq_bwd: ndir=%d", ndir);
    ST_CHECK_ARG(dout && out && tape && w_hh_fwd && dgi_fwd && dgh_fwd && B > 0 && T > 0 && H > 0, "st_gru_seq_bwd: bad arguments");
    ST_CHECK_ARG(ndir == 1 || (w_hh_bwd && dgi_bwd && dgh_bwd), "st_gru_seq_bwd: missing reverse-direction pointers");
    ST_CHECK_ARG(3 * H <= 1024, "st_gru_seq_bwd: H=%d too large (3H must be <= 1024)", H);
    GruBwdArgs a;
    a.dout = dout; a.ldd = ldd; a.out = out; a.ldo = ldo; a.tape = tape;
    a.w_hh[0] = w_hh_fwd; a.w_hh[1] = w_hh_bwd; a.dgi[0] = dgi_fwd; a.dgi[1] = dgi_bwd; a.dgh[0] = dgh_fwd; a.dgh[1] = dgh_bwd;
    a.B = B; a.T = T; a.H = H;
    const int threads = ((3 * H + 63) / 64) * 64;
    const int hpad = H <= 80 ? 80 : (H <= GRU_HMAX ? GRU_HMAX : ((H + 3) & ~3));      // the REG forms pad the dgh rows to 4 KQ entries
    const size_t lds = (size_t)(3 * hpad + 3 * H + H) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (H <= GB_LD) hipLaunchKernelGGL(gru_seq_bwd_tri_kernel, dim3(B, ndir), dim3(256), 0, st, a);
    else if (H <= 80) hipLaunchKernelGGL((gru_seq_bwd_kernel<true, 20>), dim3(B, ndir), dim3(threads), lds, st, a);
    else if (H <= GRU_HMAX) hipLaunchKernelGGL((gru_seq_bwd_kernel<true, 32>), dim3(B, ndir), dim3(threads), lds, st, a);
    else hipLaunchKernelGGL((gru_seq_bwd_kernel<false>), dim3(B, ndir), dim3(threads), lds, st, a);
    ST_LAUNCH_CHECK();
    return 0;
}
