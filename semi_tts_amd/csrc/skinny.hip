// skinny.hip -- small-batch, weight-streaming matrix kernels for gfx950 (MI355X).
//
// One decode step multiplies a (B<=64, K) activation block by LSTM/Linear weights whose
// bytes dominate everything else (75.5 MB per step at the headline shape), so these
// kernels are designed around ONE pass over the weights at HBM speed:
//   * the weight matrix is the MFMA *A* operand: a workgroup owns 16 weight rows
//     (for the LSTM: the i,f,g,o rows of 4 hidden units, so the cell update is local),
//   * the batch is the MFMA *N* dimension (16 columns per v_mfma_f32_16x16x4_f32, NB tiles),
//   * K is split over the KW waves of the workgroup in 32-float chunks (each lane loads two
//     16-byte pieces so a wave touches whole 128-byte lines of 16 rows), partial tiles are
//     reduced through LDS in a fixed order (deterministic),
//   * v_mfma_f32_16x16x4_f32 is exact fp32 (one rounding per fma), so results differ from a
//     CPU BLAS only by summation order.
// Lane mapping of the 16x16x4 MFMA: A[i = lane&15][k = lane>>4], B[k = lane>>4][n = lane&15],
// D[row = 4*(lane>>4) + r][col = lane&15].  A lane loads 4 consecutive k of its row as one
// float4 and feeds component c to the c-th of 4 MFMAs; A and B use the same k permutation.
#include "st_common.h"

namespace {

constexpr int SK_MAXSEG = 3;

struct SkArgs {
    st_seg seg[SK_MAXSEG];
    int nseg;
    int B, N, H;
    // LSTM epilogue
    const float* b_ih; const float* b_hh; const float* pre; int ldpre;
    const float* c_prev; int ldc_prev; const float* mask;
    float* h_out; int ldh; float* c_out; int ldc; float* gates_out;
    // linear epilogue
    const float* bias; int act; const float* lmask; int ldmask;
    float* y; int ldy; int n_split; float* y2; int ldy2; int rep;
};

template <bool VEC>
__device__ __forceinline__ f32x4 sk_load(const float* p, int k0, int klim) {
    // 4 floats at p[k0..k0+3]; elements at or beyond klim read as zero
    if (VEC) {
        if (k0 + 4 <= klim) return st_ld4(p + k0);
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        return z;
    } else {
        return st_ld4_guard(p + k0, klim - k0);
    }
}

// MODE 0: LSTM cell, MODE 1: linear
template <int MODE, int NB, int KW, bool VEC>
__device__ __forceinline__ void sk_body(const SkArgs& a, f32x4* red) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;   // weight row within the tile / batch column within a batch tile
    const int kq = lane >> 4;  // which 4-float group of a 16-float block
    const int tile = blockIdx.x;
    const int bbase = blockIdx.y * (NB * 16);

    int wrow;
    if (MODE == 0) wrow = (i & 3) * a.H + tile * 4 + (i >> 2);
    else { wrow = tile * 16 + i; if (wrow >= a.N) wrow = a.N - 1; }

    int brow[NB];
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
        int b = bbase + bt * 16 + i;
        brow[bt] = b < a.B ? b : a.B - 1;
    }

    f32x4 acc[NB];
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) acc[bt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // epilogue operands of the threads that run the epilogue: requested now (independent, branch-free loads; absent
    // operands read a valid dummy address and are dropped by a select when consumed), consumed after the K loop
    const int eb_ = bbase + (tid >> 6) * 16 + (lane & 15);
    const bool e_on = tid < NB * 64 && eb_ < a.B;
    float e_bi[4] = {0.f, 0.f, 0.f, 0.f}, e_bh[4] = {0.f, 0.f, 0.f, 0.f}, e_pr[4] = {0.f, 0.f, 0.f, 0.f};
    float e_c = 0.f, e_m = 1.f, l_bias[4] = {0.f, 0.f, 0.f, 0.f}, l_m1[4] = {1.f, 1.f, 1.f, 1.f};
    if (e_on) {
        const float* dummy = a.seg[0].w;
        if (MODE == 0) {
            const int u = tile * 4 + (lane >> 4);
            const float* pbi = a.b_ih ? a.b_ih + u : dummy;
            const float* pbh = a.b_hh ? a.b_hh + u : dummy;
            const float* ppr = a.pre ? a.pre + (size_t)eb_ * a.ldpre + u : dummy;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e_bi[r] = pbi[a.b_ih ? r * a.H : 0];
                e_bh[r] = pbh[a.b_hh ? r * a.H : 0];
                e_pr[r] = ppr[a.pre ? r * a.H : 0];
            }
            e_c = (a.c_prev ? a.c_prev + (size_t)eb_ * a.ldc_prev + u : dummy)[0];
            e_m = (a.mask ? a.mask + (size_t)eb_ * a.H + u : dummy)[0];
        } else {
            const int nb = tile * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + r < a.N ? nb + r : a.N - 1;
                l_bias[r] = (a.bias ? a.bias + n : dummy)[0];
                l_m1[r] = (a.lmask ? a.lmask + (size_t)eb_ * a.ldmask + n : dummy)[0];
            }
        }
    }

    int cstart = wave;  // rotates so that waves stay balanced across segments
    for (int s = 0; s < a.nseg; ++s) {
        const float* __restrict__ wp = a.seg[s].w + (size_t)wrow * a.seg[s].ldw;
        const float* xp[NB];
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) xp[bt] = a.seg[s].x + (size_t)brow[bt] * a.seg[s].ldx;
        const int K = a.seg[s].k;
        const int nchunk = (K + 31) >> 5;
        int c = cstart;
        for (; c < nchunk; c += 2 * KW) {
            // two 32-float chunks per trip: 4 weight + 4*NB activation 16-byte loads in flight
            const int k0 = c * 32 + kq * 4;
            const int k1 = (c + KW) * 32 + kq * 4;  // may lie beyond K: loads return zeros
            f32x4 w00 = sk_load<VEC>(wp, k0, K), w01 = sk_load<VEC>(wp, k0 + 16, K);
            f32x4 w10 = sk_load<VEC>(wp, k1, K), w11 = sk_load<VEC>(wp, k1 + 16, K);
            f32x4 x00[NB], x01[NB], x10[NB], x11[NB];
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                x00[bt] = sk_load<VEC>(xp[bt], k0, K);
                x01[bt] = sk_load<VEC>(xp[bt], k0 + 16, K);
                x10[bt] = sk_load<VEC>(xp[bt], k1, K);
                x11[bt] = sk_load<VEC>(xp[bt], k1 + 16, K);
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w00[cc], x00[bt][cc], acc[bt], 0, 0, 0);
                    acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w01[cc], x01[bt][cc], acc[bt], 0, 0, 0);
                    acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w10[cc], x10[bt][cc], acc[bt], 0, 0, 0);
                    acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w11[cc], x11[bt][cc], acc[bt], 0, 0, 0);
                }
            }
        }
        // next segment: the wave that would have taken chunk `nchunk` takes its chunk 0
        cstart = (cstart + KW - (nchunk % KW)) % KW;
    }

#pragma unroll
    for (int bt = 0; bt < NB; ++bt) red[(wave * NB + bt) * 64 + lane] = acc[bt];
    __syncthreads();
    if (tid >= NB * 64) return;

    const int bt = tid >> 6;
    f32x4 s = red[bt * 64 + lane];
#pragma unroll
    for (int w = 1; w < KW; ++w) {
        f32x4 t = red[(w * NB + bt) * 64 + lane];
        s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    const int b = bbase + bt * 16 + (lane & 15);
    if (b >= a.B) return;

    if (MODE == 0) {
        const int H = a.H;
        const int u = tile * 4 + (lane >> 4);
        float g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            g[r] = s[r] + (((a.b_ih ? e_bi[r] : 0.0f) + (a.b_hh ? e_bh[r] : 0.0f)) + (a.pre ? e_pr[r] : 0.0f));
        }
        const float gi = st_sigmoid_fast(g[0]), gf = st_sigmoid_fast(g[1]), gg = st_tanh_fast(g[2]), go = st_sigmoid_fast(g[3]);
        const float cp = a.c_prev ? e_c : 0.0f;
        const float c2 = gf * cp + gi * gg;
        float h2 = go * st_tanh_fast(c2);
        if (a.mask) h2 *= e_m;
        a.c_out[(size_t)b * a.ldc + u] = c2;
        a.h_out[(size_t)b * a.ldh + u] = h2;
        if (a.gates_out) {
            float* gp = a.gates_out + (size_t)b * 4 * H + u;
            gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = tile * 16 + 4 * (lane >> 4) + r;
            if (n >= a.N) continue;
            float v = s[r] + (a.bias ? l_bias[r] : 0.0f);
            v = st_act(v, a.act);
            if (a.lmask) v *= l_m1[r];
            if (a.n_split > 0 && n >= a.n_split) {
                float* p = a.y2 + (size_t)b * a.ldy2 + (size_t)(n - a.n_split) * a.rep;
                for (int j = 0; j < a.rep; ++j) p[j] = v;
            } else {
                a.y[(size_t)b * a.ldy + n] = v;
            }
        }
    }
}

template <int MODE, int NB, int KW, bool VEC>
__global__ __launch_bounds__(KW * 64) void sk_kernel(const SkArgs a) {
    __shared__ f32x4 red[KW * NB * 64];
    sk_body<MODE, NB, KW, VEC>(a, red);
}

// two independent jobs of the same shape in one launch (blockIdx.z picks the job): the two directions of a bidirectional LSTM
// advance one step each per launch instead of one launch per direction and step
template <int MODE, int NB, int KW, bool VEC>
__global__ __launch_bounds__(KW * 64) void sk_pair_kernel(const SkArgs a0, const SkArgs a1) {
    __shared__ f32x4 red[KW * NB * 64];
    if (blockIdx.z == 0) sk_body<MODE, NB, KW, VEC>(a0, red);
    else sk_body<MODE, NB, KW, VEC>(a1, red);
}

template <int MODE, int NB, bool VEC>
int sk_launch_pair(const SkArgs& a0, const SkArgs& a1, int tiles, hipStream_t st) {
    constexpr int KW = 8;
    dim3 grid(tiles, (a0.B + NB * 16 - 1) / (NB * 16), 2);
    hipLaunchKernelGGL((sk_pair_kernel<MODE, NB, KW, VEC>), grid, dim3(KW * 64), 0, st, a0, a1);
    ST_LAUNCH_CHECK();
    return 0;
}

template <int MODE>
int sk_dispatch_pair(const SkArgs (&a)[2], int tiles, hipStream_t st) {
    bool vec = true;
    for (int j = 0; j < 2; ++j) {
        const st_seg& g = a[j].seg[0];
        vec = vec && st_aligned16(g.x) && st_aligned16(g.w) && (g.ldx % 4 == 0) && (g.ldw % 4 == 0) && (g.k % 4 == 0);
    }
    const int nb = a[0].B <= 16 ? 1 : (a[0].B <= 32 ? 2 : 4);
    if (vec) {
        if (nb == 1) return sk_launch_pair<MODE, 1, true>(a[0], a[1], tiles, st);
        if (nb == 2) return sk_launch_pair<MODE, 2, true>(a[0], a[1], tiles, st);
        return sk_launch_pair<MODE, 4, true>(a[0], a[1], tiles, st);
    }
    if (nb == 1) return sk_launch_pair<MODE, 1, false>(a[0], a[1], tiles, st);
    if (nb == 2) return sk_launch_pair<MODE, 2, false>(a[0], a[1], tiles, st);
    return sk_launch_pair<MODE, 4, false>(a[0], a[1], tiles, st);
}

template <int MODE, int NB, bool VEC>
int sk_launch(const SkArgs& a, int tiles, hipStream_t st) {
    constexpr int KW = 8;
    dim3 grid(tiles, (a.B + NB * 16 - 1) / (NB * 16));
    hipLaunchKernelGGL((sk_kernel<MODE, NB, KW, VEC>), grid, dim3(KW * 64), 0, st, a);
    ST_LAUNCH_CHECK();
    return 0;
}

template <int MODE>
int sk_dispatch(const SkArgs& a, int tiles, hipStream_t st) {
    bool vec = true;
    for (int s = 0; s < a.nseg; ++s) {
        const st_seg& g = a.seg[s];
        vec = vec && st_aligned16(g.x) && st_aligned16(g.w) && (g.ldx % 4 == 0) && (g.ldw % 4 == 0) && (g.k % 4 == 0);
    }
    const int nb = a.B <= 16 ? 1 : (a.B <= 32 ? 2 : 4);
    if (vec) {
        if (nb == 1) return sk_launch<MODE, 1, true>(a, tiles, st);
        if (nb == 2) return sk_launch<MODE, 2, true>(a, tiles, st);
        return sk_launch<MODE, 4, true>(a, tiles, st);
    }
    if (nb == 1) return sk_launch<MODE, 1, false>(a, tiles, st);
    if (nb == 2) return sk_launch<MODE, 2, false>(a, tiles, st);
    return sk_launch<MODE, 4, false>(a, tiles, st);
}

int sk_fill_segs(SkArgs& a, const st_seg* segs, int nseg) {
    ST_CHECK_ARG(nseg >= 1 && nseg <= SK_MAXSEG, "skinny op: nseg=%d not in [1,%d]", nseg, SK_MAXSEG);
    a.nseg = nseg;
    for (int s = 0; s < nseg; ++s) {
        ST_CHECK_ARG(segs[s].x && segs[s].w && segs[s].k > 0, "skinny op: segment %d has null pointer or k<=0", s);
        ST_CHECK_ARG(segs[s].ldx >= segs[s].k && segs[s].ldw >= segs[s].k, "skinny op: segment %d stride < k", s);
        a.seg[s] = segs[s];
    }
    return 0;
}

}  // namespace

extern "C" int st_lstm_cell_fwd(const st_seg* segs, int nseg, const float* b_ih, const float* b_hh,
                                const float* pre, int ldpre, const float* c_prev, int ldc_prev,
                                const float* mask, float* h_out, int ldh, float* c_out, int ldc,
                                float* gates_out, int B, int H, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(B > 0 && H > 0, "st_lstm_cell_fwd: B=%d H=%d", B, H);
    ST_CHECK_ARG(H % 4 == 0, "st_lstm_cell_fwd: H=%d must be a multiple of 4", H);
    ST_CHECK_ARG(h_out && c_out, "st_lstm_cell_fwd: null output");
    SkArgs a;
    memset(&a, 0, sizeof(a));
    int rc = sk_fill_segs(a, segs, nseg);
    if (rc) return rc;
    a.B = B; a.N = 4 * H; a.H = H;
    a.b_ih = b_ih; a.b_hh = b_hh; a.pre = pre; a.ldpre = ldpre;
    a.c_prev = c_prev; a.ldc_prev = ldc_prev; a.mask = mask;
    a.h_out = h_out; a.ldh = ldh; a.c_out = c_out; a.ldc = ldc; a.gates_out = gates_out;
    return sk_dispatch<0>(a, H / 4, (hipStream_t)stream);
}

// Two LSTM cells of the same shape (e.g. the two directions of nn.LSTM(bidirectional=True) at their own time steps) in ONE launch.
// Arrays of two: segs2[j] (one segment each), b_hh2, pre2, c_prev2, h_out2, c_out2, gates_out2 (entries may be NULL where optional).
extern "C" int st_lstm_cell_pair_fwd(const st_seg* segs2, const float* const* b_hh2, const float* const* pre2, int ldpre,
                                     const float* const* c_prev2, int ldc_prev, float* const* h_out2, int ldh,
                                     float* const* c_out2, int ldc, float* const* gates_out2, int B, int H, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(B > 0 && H > 0 && H % 4 == 0 && segs2 && h_out2 && c_out2, "st_lstm_cell_pair_fwd: bad arguments");
    SkArgs a[2];
    bool vec = true;
    for (int j = 0; j < 2; ++j) {
        memset(&a[j], 0, sizeof(SkArgs));
        int rc = sk_fill_segs(a[j], segs2 + j, 1);
        if (rc) return rc;
        ST_CHECK_ARG(h_out2[j] && c_out2[j], "st_lstm_cell_pair_fwd: null output");
        a[j].B = B; a[j].N = 4 * H; a[j].H = H;
        a[j].b_hh = b_hh2 ? b_hh2[j] : nullptr; a[j].pre = pre2 ? pre2[j] : nullptr; a[j].ldpre = ldpre;
        a[j].c_prev = c_prev2 ? c_prev2[j] : nullptr; a[j].ldc_prev = ldc_prev;
        a[j].h_out = h_out2[j]; a[j].ldh = ldh; a[j].c_out = c_out2[j]; a[j].ldc = ldc;
        a[j].gates_out = gates_out2 ? gates_out2[j] : nullptr;
        const st_seg& g = a[j].seg[0];
        vec = vec && st_aligned16(g.x) && st_aligned16(g.w) && (g.ldx % 4 == 0) && (g.ldw % 4 == 0) && (g.k % 4 == 0);
    }
    (void)vec;
    return sk_dispatch_pair<0>(a, H / 4, (hipStream_t)stream);
}

// Two plain linears y_j = x_j W_j^T of the same shape (one segment each, no bias / activation) in one launch.
extern "C" int st_skinny_linear_pair_fwd(const st_seg* segs2, float* const* y2, int ldy, int B, int N, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(B > 0 && N > 0 && segs2 && y2 && y2[0] && y2[1], "st_skinny_linear_pair_fwd: bad arguments");
    SkArgs a[2];
    for (int j = 0; j < 2; ++j) {
        memset(&a[j], 0, sizeof(SkArgs));
        int rc = sk_fill_segs(a[j], segs2 + j, 1);
        if (rc) return rc;
        a[j].B = B; a[j].N = N; a[j].H = 0; a[j].act = ST_ACT_NONE; a[j].y = y2[j]; a[j].ldy = ldy;
    }
    return sk_dispatch_pair<1>(a, (N + 15) / 16, (hipStream_t)stream);
}

extern "C" int st_skinny_linear_fwd(const st_seg* segs, int nseg, const float* bias, int act,
                                    const float* mask, int ldmask, float* y, int ldy,
                                    int n_split, float* y2, int ldy2, int rep,
                                    int B, int N, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(B > 0 && N > 0 && y, "st_skinny_linear_fwd: B=%d N=%d y=%p", B, N, (void*)y);
    ST_CHECK_ARG(n_split <= 0 || (y2 && rep >= 1), "st_skinny_linear_fwd: n_split without y2/rep");
    SkArgs a;
    memset(&a, 0, sizeof(a));
    int rc = sk_fill_segs(a, segs, nseg);
    if (rc) return rc;
    a.B = B; a.N = N; a.H = 0;
    a.bias = bias; a.act = act; a.lmask = mask; a.ldmask = ldmask;
    a.y = y; a.ldy = ldy; a.n_split = n_split; a.y2 = y2; a.ldy2 = ldy2; a.rep = rep;
    return sk_dispatch<1>(a, (N + 15) / 16, (hipStream_t)stream);
}
