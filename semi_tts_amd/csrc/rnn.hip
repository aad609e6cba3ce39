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
};

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
        __syncthreads();
        if (upd) {
            const float r = st_sigmoid(gr + ghs[j]);
            const float z = st_sigmoid(gz + ghs[H + j]);
            const float n = tanhf(gn + r * ghs[2 * H + j]);
            const float hn = (1.0f - z) * n + z * hbuf[j];
            hbuf[j] = hn;
            a.out[((size_t)b * T + t) * a.ldo + d * H + j] = hn;
            gr = nr; gz = nz; gn = nn;
        }
        __syncthreads();
        t = tn;
    }
}

}  // namespace

extern "C" int st_gru_seq_fwd(const float* gi_fwd, const float* gi_bwd, const float* w_hh_fwd, const float* w_hh_bwd,
                              const float* b_hh_fwd, const float* b_hh_bwd, float* out, int ldo,
                              int B, int T, int H, int ndir, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(ndir == 1 || ndir == 2, "st_gru_seq_fwd: ndir=%d", ndir);
    ST_CHECK_ARG(gi_fwd && w_hh_fwd && b_hh_fwd && out && B > 0 && T > 0 && H > 0, "st_gru_seq_fwd: bad arguments");
    ST_CHECK_ARG(ndir == 1 || (gi_bwd && w_hh_bwd && b_hh_bwd), "st_gru_seq_fwd: missing reverse-direction pointers");
    ST_CHECK_ARG(3 * H <= 1024, "st_gru_seq_fwd: H=%d too large (3H must be <= 1024)", H);
    ST_CHECK_ARG(ldo >= ndir * H, "st_gru_seq_fwd: ldo=%d < ndir*H", ldo);
    GruArgs a;
    a.gi[0] = gi_fwd; a.gi[1] = gi_bwd; a.w_hh[0] = w_hh_fwd; a.w_hh[1] = w_hh_bwd;
    a.b_hh[0] = b_hh_fwd; a.b_hh[1] = b_hh_bwd; a.out = out; a.ldo = ldo; a.B = B; a.T = T; a.H = H;
    const int threads = ((3 * H + 63) / 64) * 64;
    const size_t lds = (size_t)(((H + 3) & ~3) + 3 * H) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (H <= GRU_HMAX) hipLaunchKernelGGL((gru_seq_kernel<true>), dim3(B, ndir), dim3(threads), lds, st, a);
    else hipLaunchKernelGGL((gru_seq_kernel<false>), dim3(B, ndir), dim3(threads), lds, st, a);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_lstm_seq_fwd(const float* xproj, const float* w_hh, const float* b_hh, float* out, int ldo, int ocol,
                               float* ws, int B, int T, int H, int reverse, void* stream) {
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
        const float* cprev = s == 0 ? zero : cbuf[(s - 1) & 1];
        int rc = st_lstm_cell_fwd(&seg, 1, nullptr, b_hh, xproj + (size_t)t * 4 * H, T * 4 * H,
                                  cprev, H, nullptr, out + (size_t)t * ldo + ocol, T * ldo,
                                  cbuf[s & 1], H, nullptr, B, H, stream);
        if (rc) return rc;
    }
    return 0;
}
