// gemm_lab2.hip -- round-4 experiment bench for the forward fp32 MFMA GEMM / implicit-GEMM conv core.
// Not part of the library: it measures tile geometries, the full-line loader and split-K factors on the shapes of one C2 forward
// before they go into csrc/gemm.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab2.hip -o tools/gemm_lab2 && tools/gemm_lab2
// What differs from csrc/gemm.hip's gm_pipe_kernel (r03):
//   * k-chunk of 32 floats: 8 adjacent lanes load ONE 128-byte line of a row (r03: 4 lanes x 16 bytes = half a line, 16 rows per
//     wave instruction);
//   * tile geometries up to 128 x 128 with 4 or 8 waves (bytes per flop of the operand stream halve from 64 x 64 to 128 x 128);
//   * the columns of a wave's MFMA tiles are INTERLEAVED (tile nt, column r <-> output column NT r + nt), W rows are permuted while
//     they are written to LDS, so a lane ends up with NT consecutive output columns per row: 8- or 16-byte stores;
//   * the reduction axis of a conv is the flat (tap, channel) index of tap-major weights; a 16-float half chunk never straddles taps
//     (Cin % 16 == 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct G2Args {
    const float* A; int lda; const float* W; float* C; int ldc;
    int Bn, Tin, Tout, Cin, N, KT, pad;
    int M, K;                 // Bn * Tout, KT * Cin
    int chunks_per_split;     // > 0: blockIdx.z takes chunks [z cps, (z + 1) cps) and writes its raw partial to ws[z][m][n]
    float* ws;
    const float* bias;
};

constexpr int G2_LD = 40;     // LDS row stride (floats) of a 32-float chunk: the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte slots

// ABL (timing ablations, results garbage): 1 = no global loads, 2 = no LDS stores, 3 = no LDS reads, 4 = no MFMAs, 5 = linear addressing
// (no tap / frame arithmetic: valid for KT = 1 only)
template <int BM, int BN, int NWM, int NWN, int ABL = 0>
__global__ __launch_bounds__(64 * NWM * NWN) void g2_kernel(const G2Args g) {
    constexpr int NTH = 64 * NWM * NWN;
    constexpr int MT = BM / NWM / 16, NT = BN / NWN / 16;
    constexpr int RPP = NTH / 8;                  // rows one pass of the workgroup's threads covers (8 lanes per row)
    constexpr int LA = BM / RPP, LB = BN / RPP;   // 16-byte loads per thread and chunk
    static_assert(BM % RPP == 0 && BN % RPP == 0 && MT >= 1 && NT >= 1, "geometry");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As[2] = {lds, lds + BM * G2_LD};
    float* Bs[2] = {lds + 2 * BM * G2_LD, lds + 2 * BM * G2_LD + BN * G2_LD};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int srow = tid >> 3, sp = tid & 7;

    // staging rows of this thread
    const float* arow[LA]; int atb[LA]; bool aok[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int m = m0 + srow + i * RPP;
        aok[i] = m < g.M;
        const int mc = min(m, g.M - 1);
        const int b = mc / g.Tout, t = mc - b * g.Tout;
        arow[i] = g.A + (size_t)b * g.Tin * g.lda;
        atb[i] = t - g.pad;
    }
    const float* wrow[LB]; bool wok[LB]; int wlds[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int r = srow + i * RPP;             // row of the W tile = output column n0 + r
        wok[i] = n0 + r < g.N;
        wrow[i] = g.W + (size_t)min(n0 + r, g.N - 1) * g.K;
        const int wb = r / (16 * NT), j = r % (16 * NT);
        wlds[i] = (wb * 16 * NT + (j % NT) * 16 + j / NT) * G2_LD + sp * 4;      // interleaved columns: see the header
    }
    const int nch = (g.K + 31) / 32;
    const int c_lo = g.chunks_per_split > 0 ? (int)blockIdx.z * g.chunks_per_split : 0;
    const int c_hi = g.chunks_per_split > 0 ? min(nch, c_lo + g.chunks_per_split) : nch;
    // flat k of this thread's piece in the next chunk to request, as (tap, channel)
    int k_n = c_lo * 32 + sp * 4;
    int tap_n = k_n / g.Cin, ci_n = k_n - tap_n * g.Cin;
    int left = c_hi - c_lo;

    struct Regs { f32x4 a[LA], w[LB]; unsigned va, vw; };
    auto issue = [&](Regs& r) __attribute__((always_inline)) {
        const bool in_k = left > 0 && k_n < g.K;
        --left;
        const int kc = min(k_n, g.K - 4);
        const int tapc = min(tap_n, g.KT - 1);
        const int cic = in_k ? ci_n : 0;
        r.va = 0; r.vw = 0;
        if (ABL == 1) { r.va = r.vw = in_k ? ~0u : 0u; k_n += 32; return; }
        if (ABL == 5) {
#pragma unroll
            for (int i = 0; i < LA; ++i) { r.a[i] = *reinterpret_cast<const f32x4*>(arow[i] + (size_t)(atb[i] + g.pad) * g.lda + kc); if (in_k && aok[i]) r.va |= 1u << i; }
#pragma unroll
            for (int i = 0; i < LB; ++i) { r.w[i] = *reinterpret_cast<const f32x4*>(wrow[i] + kc); if (in_k && wok[i]) r.vw |= 1u << i; }
            k_n += 32;
            return;
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int ti = atb[i] + tapc;
            const int tic = min(max(ti, 0), g.Tin - 1);
            r.a[i] = *reinterpret_cast<const f32x4*>(arow[i] + (size_t)tic * g.lda + cic);
            if (in_k && aok[i] && ti >= 0 && ti < g.Tin) r.va |= 1u << i;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            r.w[i] = *reinterpret_cast<const f32x4*>(wrow[i] + kc);
            if (in_k && wok[i]) r.vw |= 1u << i;
        }
        k_n += 32; ci_n += 32;
        if (ci_n >= g.Cin) { ci_n -= g.Cin; ++tap_n; }
        if (ci_n >= g.Cin) { ci_n -= g.Cin; ++tap_n; }      // (Cin = 16)
    };
    auto commit = [&](const Regs& r, int buf) __attribute__((always_inline)) {
        if (ABL == 2) { if (r.va == 0x12345 && r.a[0][0] == 1.5f && r.w[0][0] == 2.5f) As[buf][tid] = 0.f; return; }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(As[buf] + (srow + i * RPP) * G2_LD + sp * 4) = (r.va >> i) & 1 ? r.a[i] : z;
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<f32x4*>(Bs[buf] + wlds[i]) = (r.vw >> i) & 1 ? r.w[i] : z;
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a4[MT], b4[NT];
            if (ABL == 3) {
#pragma unroll
                for (int t = 0; t < MT; ++t) a4[t] = f32x4{(float)buf, (float)h, (float)t, 1.f};
#pragma unroll
                for (int t = 0; t < NT; ++t) b4[t] = f32x4{(float)buf, (float)h, (float)t, 2.f};
            } else {
#pragma unroll
            for (int t = 0; t < MT; ++t) a4[t] = *reinterpret_cast<const f32x4*>(As[buf] + (wm * 16 * MT + t * 16 + fr) * G2_LD + h * 16 + fq * 4);
#pragma unroll
            for (int t = 0; t < NT; ++t) b4[t] = *reinterpret_cast<const f32x4*>(Bs[buf] + (wn * 16 * NT + t * 16 + fr) * G2_LD + h * 16 + fq * 4);
            }
            if (ABL == 4) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += a4[mt] * b4[nt];
                continue;
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][cc], b4[nt][cc], acc[mt][nt], 0, 0, 0);
        }
    };
    const int n_my = c_hi - c_lo;
    Regs r0, r1;
    issue(r0); issue(r1);
    commit(r0, 0);
    issue(r0);
    lds_barrier();
    for (int c = 0; c < n_my; c += 2) {
        commit(r1, 1);
        issue(r1);
        compute(0);
        lds_barrier();
        if (c + 1 >= n_my) break;
        commit(r0, 0);
        issue(r0);
        compute(1);
        lds_barrier();
    }
    // epilogue: lane (fr, fq) holds rows 4 fq + e of each row tile and the NT consecutive columns nb + NT fr + nt
    float* out = g.chunks_per_split > 0 ? g.ws + (size_t)blockIdx.z * g.M * g.N : g.C;
    const int ldo = g.chunks_per_split > 0 ? g.N : g.ldc;
    const int nb = n0 + wn * 16 * NT + NT * fr;
    const bool vec_ok = (g.N % NT == 0) && (ldo % NT == 0);
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (g.bias && g.chunks_per_split == 0 && nb + nt < g.N) ? g.bias[nb + nt] : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm * 16 * MT + mt * 16 + 4 * fq + e;
            if (m >= g.M) continue;
            float* p = out + (size_t)m * ldo + nb;
            if (vec_ok && nb + NT <= g.N) {
                if (NT == 4) *reinterpret_cast<f32x4*>(p) = f32x4{acc[mt][0][e] + bv[0], acc[mt][1 % NT][e] + bv[1 % NT], acc[mt][2 % NT][e] + bv[2 % NT], acc[mt][3 % NT][e] + bv[3 % NT]};
                else if (NT == 2) *reinterpret_cast<f32x2*>(p) = f32x2{acc[mt][0][e] + bv[0], acc[mt][1 % NT][e] + bv[1 % NT]};
                else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) p[nt] = acc[mt][nt][e] + bv[nt];
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) if (nb + nt < g.N) p[nt] = acc[mt][nt][e] + bv[nt];
            }
        }
}


// ---- g3: both operands by LDS-DMA (global_load_lds_dwordx4), XOR-swizzled lane-linear LDS image, three buffers, one barrier per chunk ----
// A wave instruction moves 8 rows x 128 bytes (one full line per row) straight into LDS: no staging registers, no ds_write pass, no
// selects.  LDS row L (128 bytes, 8 slots of 16 bytes) holds in slot s the row's piece s ^ (L & 7) -- the source address is swizzled,
// the destination is lane-linear -- and a fragment read of piece p goes to slot p ^ (L & 7): the 16 lanes of every ds_read_b128 group
// fall on 16 distinct 16-byte slots.  Rows / pieces that do not exist (frames outside the utterance, k past K, rows past M or N) read
// 16 zero bytes from a constant instead.  The reduction axis is the flat (tap, channel) index: with channels-last activations whose
// row stride is Cin, the taps of one output row are ONE contiguous run of KT * Cin floats.
__device__ const f32x4 g3_zero4 = {0.f, 0.f, 0.f, 0.f};
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// OPT bit 0: XCD-aware tile order (workgroup b runs on XCD b % 8: XCD x takes the x-th contiguous eighth of the tiles in column-block-
// major order, so the W rows of a column block stay in ONE XCD's L2); bit 1: the fragments of both 16-float halves are read before the
// chunk's MFMAs (one exposed LDS latency per chunk instead of two)
template <int BM, int BN, int NWM, int NWN, int ABL = 0, int OPT = 0>
__global__ __launch_bounds__(64 * NWM * NWN) void g3_kernel(const G2Args g) {
    constexpr int NTH = 64 * NWM * NWN, NW = NWM * NWN;
    constexpr int MT = BM / NWM / 16, NT = BN / NWN / 16;
    constexpr int RPP = NTH / 8;
    constexpr int LA = BM / RPP, LB = BN / RPP;
    constexpr int BUF = (BM + BN) * 32;           // floats per buffer
    constexpr int NBUF = (OPT & 4) ? 4 : 3;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "geometry");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    int bx = blockIdx.x, by = blockIdx.y;
    if (OPT & 1) {
        const int gx = gridDim.x, T = gx * gridDim.y;
        const int b = bx + gx * by;
        const int per = (T + 7) / 8;
        const int tile = (b % 8) * per + b / 8;
        if (tile >= T || b / 8 >= per) return;      // (uniform; grids are padded to 8 * per by the launcher)
        by = tile / gx; bx = tile - by * gx;
    }
    const int m0 = bx * BM, n0 = by * BN;
    const int lrow = lane >> 3, slot = lane & 7;
    const float* zp = reinterpret_cast<const float*>(&g3_zero4);
    // per load: the row's run start (flat k = 0), its valid k range, the piece this lane fetches
    const float* abase[LA]; int aklo[LA], akhi[LA], apk[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int L = i * RPP + wave * 8 + lrow;              // LDS row = tile row
        const int m = m0 + L;
        const int mc = min(m, g.M - 1);
        const int b = mc / g.Tout, t = mc - b * g.Tout;
        abase[i] = g.A + ((size_t)b * g.Tin + (t - g.pad)) * g.lda;     // (may point in front of the utterance: only valid pieces are read)
        const int tlo = max(0, g.pad - t), thi = min(g.KT, g.Tin + g.pad - t);
        aklo[i] = m < g.M ? tlo * g.Cin : 0x7fffffff;
        akhi[i] = thi * g.Cin;
        apk[i] = (slot ^ (L & 7)) * 4;
    }
    const float* wbase[LB]; int wkhi[LB], wpk[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int L = i * RPP + wave * 8 + lrow;              // LDS row; the W-tile row it holds: interleaved columns
        const int wb = L / (16 * NT), within = L % (16 * NT);
        const int r = wb * 16 * NT + NT * (within % 16) + within / 16;
        wbase[i] = g.W + (size_t)min(n0 + r, g.N - 1) * g.K;
        wkhi[i] = n0 + r < g.N ? g.K : 0;
        wpk[i] = (slot ^ (L & 7)) * 4;
    }
    const int nch = (g.K + 31) / 32;
    const int c_lo = g.chunks_per_split > 0 ? (int)blockIdx.z * g.chunks_per_split : 0;
    const int c_hi = g.chunks_per_split > 0 ? min(nch, c_lo + g.chunks_per_split) : nch;
    const int n_my = c_hi - c_lo;
    int kreq = c_lo * 32;                          // flat k of the next chunk to request
    auto issue = [&](int bufi) __attribute__((always_inline)) {
        if (ABL == 1) return;
        float* ab = lds + bufi * BUF;
        float* bb = ab + BM * 32;
        const bool live = kreq < c_hi * 32;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int k = kreq + apk[i];
            const bool ok = live && k >= aklo[i] && k < akhi[i];
            const float* src = ok ? abase[i] + k : zp;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ab + (i * RPP + wave * 8) * 32), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int k = kreq + wpk[i];
            const bool ok = live && k < wkhi[i];
            const float* src = ok ? wbase[i] + k : zp;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(bb + (i * RPP + wave * 8) * 32), 16, 0, 0);
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
        if ((OPT & 2) && ABL != 4) {
            f32x4 a4[2][MT], b4[2][NT];
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
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a4[MT], b4[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) { const int L = wm * 16 * MT + t * 16 + fr; a4[t] = *reinterpret_cast<const f32x4*>(ab + L * 32 + (((h * 4 + fq) ^ (L & 7)) * 4)); }
#pragma unroll
            for (int t = 0; t < NT; ++t) { const int L = wn * 16 * NT + t * 16 + fr; b4[t] = *reinterpret_cast<const f32x4*>(bb + L * 32 + (((h * 4 + fq) ^ (L & 7)) * 4)); }
            if (ABL == 4) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += a4[mt] * b4[nt];
                continue;
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][cc], b4[nt][cc], acc[mt][nt], 0, 0, 0);
        }
    };
    auto compute_half = [&](int bufi, int h) __attribute__((always_inline)) {
        const float* ab = lds + bufi * BUF;
        const float* bb = ab + BM * 32;
        f32x4 a4[MT], b4[NT];
#pragma unroll
        for (int t = 0; t < MT; ++t) { const int L = wm * 16 * MT + t * 16 + fr; a4[t] = *reinterpret_cast<const f32x4*>(ab + L * 32 + (((h * 4 + fq) ^ (L & 7)) * 4)); }
#pragma unroll
        for (int t = 0; t < NT; ++t) { const int L = wn * 16 * NT + t * 16 + fr; b4[t] = *reinterpret_cast<const f32x4*>(bb + L * 32 + (((h * 4 + fq) ^ (L & 7)) * 4)); }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][cc], b4[nt][cc], acc[mt][nt], 0, 0, 0);
    };
    // chunk c lives in buffer c % NBUF; the requests of chunk c + NBUF - 1 go out after the barrier of iteration c (every wave has left
    // compute(c - 1), whose buffer they overwrite); at the top of iteration c chunks c + 1 .. c + NBUF - 2 may still be in flight
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) issue(i);
    int bi = 0, bn = NBUF - 1;
    for (int c = 0; c < n_my; ++c) {
        if (ABL != 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * (LA + LB)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (OPT & 8) {            // requests between the chunk's two MFMA halves: their address arithmetic issues in the MFMAs' shadow
            compute_half(bi, 0);
            issue(bn);
            compute_half(bi, 1);
        } else {
            issue(bn);
            compute(bi);
        }
        bi = bi == NBUF - 1 ? 0 : bi + 1;
        bn = bn == NBUF - 1 ? 0 : bn + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* out = g.chunks_per_split > 0 ? g.ws + (size_t)blockIdx.z * g.M * g.N : g.C;
    const int ldo = g.chunks_per_split > 0 ? g.N : g.ldc;
    const int nb = n0 + wn * 16 * NT + NT * fr;
    const bool vec_ok = (g.N % NT == 0) && (ldo % NT == 0);
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (g.bias && g.chunks_per_split == 0 && nb + nt < g.N) ? g.bias[nb + nt] : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm * 16 * MT + mt * 16 + 4 * fq + e;
            if (m >= g.M) continue;
            float* p = out + (size_t)m * ldo + nb;
            if (vec_ok && nb + NT <= g.N) {
                if (NT == 4) *reinterpret_cast<f32x4*>(p) = f32x4{acc[mt][0][e] + bv[0], acc[mt][1 % NT][e] + bv[1 % NT], acc[mt][2 % NT][e] + bv[2 % NT], acc[mt][3 % NT][e] + bv[3 % NT]};
                else if (NT == 2) *reinterpret_cast<f32x2*>(p) = f32x2{acc[mt][0][e] + bv[0], acc[mt][1 % NT][e] + bv[1 % NT]};
                else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) p[nt] = acc[mt][nt][e] + bv[nt];
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) if (nb + nt < g.N) p[nt] = acc[mt][nt][e] + bv[nt];
            }
        }
}
#define G3_EPILOGUE_END 1

__global__ void finish_kernel(const float* __restrict__ part, float* __restrict__ out, const float* bias, int N, size_t n4, int S) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
    for (int z = 1; z < S; ++z) { const f32x4 v = reinterpret_cast<const f32x4*>(part)[i + (size_t)z * n4]; s += v; }
    if (bias) { const int n = (int)((i * 4) % N); s[0] += bias[n]; s[1] += bias[n + 1]; s[2] += bias[n + 2]; s[3] += bias[n + 3]; }
    reinterpret_cast<f32x4*>(out)[i] = s;
}

// reference: the conv as written (channels-last, tap-major weights), one thread per output element
__global__ void naive_kernel(const G2Args g) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= g.N) return;
    const int b = m / g.Tout, t = m - b * g.Tout;
    float s = 0.f;
    for (int tap = 0; tap < g.KT; ++tap) {
        const int ti = t + tap - g.pad;
        if (ti < 0 || ti >= g.Tin) continue;
        const float* a = g.A + ((size_t)b * g.Tin + ti) * g.lda;
        const float* w = g.W + (size_t)n * g.K + (size_t)tap * g.Cin;
        for (int c = 0; c < g.Cin; ++c) s = fmaf(a[c], w[c], s);
    }
    g.C[(size_t)m * g.ldc + n] = s + (g.bias ? g.bias[n] : 0.0f);
}

struct Shape { const char* name; int Bn, T, Cin, N, KT, pad; };

template <typename F>
static float time_us(F launch, int iters = 40) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1000.f / iters;
}

static float *dA, *dW, *dC, *dR, *dP, *dBias;
static std::vector<float> hC, hR;

static double check(size_t n) {
    CK(hipMemcpy(hC.data(), dC, n * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t i = 0; i < n; i += 13) worst = fmax(worst, fabs((double)hC[i] - hR[i]) / (1.0 + fabs((double)hR[i])));
    return worst;
}

struct Best { float us = 1e30f; char what[96] = ""; };

template <int BM, int BN, int NWM, int NWN, int ABL = 0>
static void run(const Shape& s, G2Args g, int S, Best& best, bool verbose) {
    const int nch = (g.K + 31) / 32;
    if (S > nch) return;
    dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN, S);
    if ((long)grid.x * grid.y * S > 6000) return;
    const size_t lds = (size_t)2 * (BM + BN) * G2_LD * 4;
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(g2_kernel<BM, BN, NWM, NWN, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    g.chunks_per_split = S > 1 ? (nch + S - 1) / S : 0;
    g.ws = dP;
    CK(hipMemset(dC, 0, (size_t)g.M * g.N * 4));
    const size_t n4 = (size_t)g.M * g.N / 4;
    auto f = [&] {
        hipLaunchKernelGGL((g2_kernel<BM, BN, NWM, NWN, ABL>), grid, dim3(64 * NWM * NWN), lds, 0, g);
        if (S > 1) hipLaunchKernelGGL(finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, dP, dC, g.bias, g.N, n4, S);
    };
    const float us = time_us(f);
    const double tf = 2.0 * g.M * g.N * g.K / us / 1e6;
    const double err = check((size_t)g.M * g.N);
    char what[96];
    snprintf(what, sizeof what, "%dx%d w%dx%d S=%d wgs=%d%s", BM, BN, NWM, NWN, S, (int)(grid.x * grid.y * S),
             ABL == 1 ? " -loads" : ABL == 2 ? " -ldsW" : ABL == 3 ? " -ldsR" : ABL == 4 ? " -mfma" : ABL == 5 ? " lin" : "");
    if (verbose || err > 1e-4 || ABL) printf("    %-34s %8.2f us  %6.1f TF  frac %.3f  err %.1e\n", what, us, tf, tf / 157.3, err);
    if (ABL == 0 && err <= 1e-4 && us < best.us) { best.us = us; snprintf(best.what, sizeof best.what, "%s", what); }
}


template <int BM, int BN, int NWM, int NWN, int ABL = 0, int OPT = 0>
static void run3(const Shape& s, G2Args g, int S, Best& best, bool verbose) {
    const int nch = (g.K + 31) / 32;
    if (S > nch || g.lda != g.Cin) return;
    dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN, S);
    if ((long)grid.x * grid.y * S > 6000) return;
    if (OPT & 1) {      // pad the tile count to a multiple of 8 (as extra column blocks' worth of workgroups, dropped in the kernel)
        const int T = grid.x * grid.y, per = (T + 7) / 8;
        while ((int)(grid.x * grid.y) < 8 * per) ++grid.y;
    }
    const size_t lds = (size_t)((OPT & 4) ? 4 : 3) * (BM + BN) * 32 * 4;
    if (lds > 160 * 1024) return;
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(g3_kernel<BM, BN, NWM, NWN, ABL, OPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    g.chunks_per_split = S > 1 ? (nch + S - 1) / S : 0;
    g.ws = dP;
    CK(hipMemset(dC, 0, (size_t)g.M * g.N * 4));
    const size_t n4 = (size_t)g.M * g.N / 4;
    auto f = [&] {
        hipLaunchKernelGGL((g3_kernel<BM, BN, NWM, NWN, ABL, OPT>), grid, dim3(64 * NWM * NWN), lds, 0, g);
        if (S > 1) hipLaunchKernelGGL(finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, dP, dC, g.bias, g.N, n4, S);
    };
    const float us = time_us(f);
    const double tf = 2.0 * g.M * g.N * g.K / us / 1e6;
    const double err = check((size_t)g.M * g.N);
    char what[96];
    snprintf(what, sizeof what, "g3%s%s%s%s %dx%d w%dx%d S=%d wgs=%d%s", OPT & 1 ? "x" : "", OPT & 2 ? "f" : "", OPT & 4 ? "4" : "", OPT & 8 ? "i" : "", BM, BN, NWM, NWN, S, (int)(grid.x * grid.y * S),
             ABL == 1 ? " -loads" : ABL == 4 ? " -mfma" : "");
    if (verbose || err > 1e-4 || ABL) printf("    %-34s %8.2f us  %6.1f TF  frac %.3f  err %.1e\n", what, us, tf, tf / 157.3, err);
    if (ABL == 0 && err <= 1e-4 && us < best.us) { best.us = us; snprintf(best.what, sizeof best.what, "%s", what); }
}

int main(int argc, char** argv) {
    const bool verbose = argc > 1;
    const Shape shapes[] = {  // the forward products of one C2 pass (tools/bench_gemm_shapes.py)
        {"enc conv k5 512->512", 32, 43, 512, 512, 5, 2}, {"enc conv k5 64->512", 32, 43, 64, 512, 5, 2},
        {"enc lstm in-proj 512->1024", 32, 43, 512, 1024, 1, 0}, {"memory layer 512->256", 32, 43, 512, 256, 1, 0},
        {"bank conv k1 80->80", 32, 258, 80, 80, 1, 0}, {"bank conv k4 80->80", 32, 258, 80, 80, 4, 2}, {"bank conv k8 80->80", 32, 258, 80, 80, 8, 4},
        {"proj conv k3 640->128", 32, 258, 640, 128, 3, 1}, {"proj conv k3 128->80", 32, 258, 128, 80, 3, 1},
        {"highway 80->80", 32, 258, 80, 80, 1, 0}, {"gru in-proj 80->240", 32, 258, 80, 240, 1, 0}, {"linear 160->1024", 32, 258, 160, 1024, 1, 0},
        {"teacher prenet 240->256", 32, 86, 240, 256, 1, 0},
        {"dx enc conv (same shape)", 32, 43, 512, 512, 5, 2}, {"dx linear 1024->160", 32, 258, 1024, 160, 1, 0},
        {"c5 enc conv k5 512->512", 64, 171, 512, 512, 5, 2}, {"c5 proj conv 640->128", 64, 1066, 640, 128, 3, 1},
        {"big linear 1920->128", 64, 1066, 1920, 128, 1, 0}, {"big linear 2048->1024", 64, 1066, 2048, 1024, 1, 0}};
    size_t maxA = 0, maxW = 0, maxC = 0;
    for (const Shape& s : shapes) {
        const int Tout = s.T + 2 * s.pad - s.KT + 1;
        maxA = std::max(maxA, (size_t)s.Bn * s.T * s.Cin); maxW = std::max(maxW, (size_t)s.N * s.Cin * s.KT); maxC = std::max(maxC, (size_t)s.Bn * Tout * s.N);
    }
    CK(hipMalloc(&dA, maxA * 4)); CK(hipMalloc(&dW, maxW * 4)); CK(hipMalloc(&dC, maxC * 4)); CK(hipMalloc(&dR, maxC * 4));
    CK(hipMalloc(&dP, maxC * 4 * 8)); CK(hipMalloc(&dBias, 4096 * 4));
    std::vector<float> h(std::max(std::max(maxA, maxW), (size_t)4096));
    srand(1);
    // LAB_RANDN=1: unit normal data as the library's benchmarks use (more bit toggling than U(-0.5, 0.5): the clock the chip sustains
    // under MFMA load depends on it)
    const bool randn = getenv("LAB_RANDN") != nullptr;
    auto draw = [&]() -> float {
        if (!randn) return (float)rand() / RAND_MAX - 0.5f;
        const float u1 = ((float)rand() + 1.0f) / ((float)RAND_MAX + 2.0f), u2 = (float)rand() / RAND_MAX;
        return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    };
    for (float& v : h) v = draw();
    CK(hipMemcpy(dA, h.data(), maxA * 4, hipMemcpyHostToDevice));
    for (float& v : h) v = draw();
    CK(hipMemcpy(dW, h.data(), maxW * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBias, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    hC.resize(maxC); hR.resize(maxC);
    for (const Shape& s : shapes) {
        G2Args g;
        memset((void*)&g, 0, sizeof g);
        g.A = dA; g.lda = s.Cin; g.W = dW; g.C = dC; g.Bn = s.Bn; g.Tin = s.T; g.Tout = s.T + 2 * s.pad - s.KT + 1; g.Cin = s.Cin; g.N = s.N;
        g.ldc = s.N; g.KT = s.KT; g.pad = s.pad; g.M = g.Bn * g.Tout; g.K = s.KT * s.Cin; g.bias = dBias;
        printf("%s  M=%d N=%d K=%d  (%.2f GFLOP, floor %.1f us)\n", s.name, g.M, g.N, g.K, 2e-9 * g.M * g.N * g.K, 2.0 * g.M * g.N * g.K / 157.3e6);
        G2Args gr = g; gr.C = dR;
        hipLaunchKernelGGL(naive_kernel, dim3((g.N + 255) / 256, g.M), dim3(256), 0, 0, gr);
        CK(hipMemcpy(hR.data(), dR, (size_t)g.M * g.N * 4, hipMemcpyDeviceToHost));
        Best best;
        if (argc > 1 && argv[1][0] == 'q') {      // profiler run: the library's configuration of the first shape only
            if (&s != &shapes[0]) continue;
            run3<64, 64, 2, 2, 0, 2>(s, g, 4, best, true);
            continue;
        }
        if (argc > 1 && argv[1][0] == 'b') {      // 128-wide tiles at one workgroup per compute unit: buffers, interleave, split factors
            const bool want = (g.K >= 1536 && g.N >= 128) ;
            if (!want) continue;
            for (int S : {1, 2, 3, 4, 5, 6}) {
                if ((long)g.M * g.N * S > (long)maxC * 8) continue;
#define R4(BM, BN, A, B, O) run3<BM, BN, A, B, 0, O>(s, g, S, best, true);
                R4(128, 128, 4, 2, 0) R4(128, 128, 4, 2, 4) R4(128, 128, 4, 2, 8) R4(128, 128, 4, 2, 12) R4(128, 128, 4, 2, 14)
                R4(128, 128, 2, 2, 4) R4(128, 128, 2, 2, 12) R4(128, 64, 4, 2, 12) R4(128, 64, 4, 2, 4) R4(64, 64, 2, 2, 12) R4(64, 64, 2, 2, 2)
            }
            const double tfb = 2.0 * g.M * g.N * g.K / best.us / 1e6;
            printf("  BEST %-34s %8.2f us  %6.1f TF  frac %.3f\n", best.what, best.us, tfb, tfb / 157.3);
            continue;
        }
        if (argc > 1 && argv[1][0] == 'a') {      // ablations on the steady-state shapes
            if (g.M < 60000) continue;
            for (int S : {1}) {
#define ABLS(BM, BN, A, B) run<BM, BN, A, B, 0>(s, g, S, best, true); run<BM, BN, A, B, 1>(s, g, S, best, true); run<BM, BN, A, B, 2>(s, g, S, best, true); \
                           run<BM, BN, A, B, 3>(s, g, S, best, true); run<BM, BN, A, B, 4>(s, g, S, best, true);
                ABLS(64, 64, 2, 2) ABLS(128, 128, 2, 2) ABLS(128, 128, 4, 2) ABLS(128, 64, 4, 2)
#define ABL3(BM, BN, A, B) run3<BM, BN, A, B, 0>(s, g, S, best, true); run3<BM, BN, A, B, 1>(s, g, S, best, true); run3<BM, BN, A, B, 4>(s, g, S, best, true);
                ABL3(64, 64, 2, 2) ABL3(128, 128, 2, 2) ABL3(128, 128, 4, 2) ABL3(128, 64, 4, 2) ABL3(128, 64, 2, 2)
            }
            continue;
        }
        for (int S : {1, 2, 3, 4, 6, 8}) {
            const bool big = (size_t)g.M * g.N * S > maxC * 8;
            if (big) continue;
            run<64, 64, 2, 2>(s, g, S, best, verbose);
            run<128, 64, 2, 2>(s, g, S, best, verbose);
            run<128, 64, 4, 2>(s, g, S, best, verbose);
            run<128, 128, 2, 2>(s, g, S, best, verbose);
            run<128, 128, 4, 2>(s, g, S, best, verbose);
            run<64, 128, 2, 2>(s, g, S, best, verbose);
            run<32, 64, 2, 2>(s, g, S, best, verbose);
            run<64, 32, 2, 2>(s, g, S, best, verbose);
#define R3O(BM, BN, A, B) run3<BM, BN, A, B, 0, 1>(s, g, S, best, verbose); run3<BM, BN, A, B, 0, 2>(s, g, S, best, verbose); run3<BM, BN, A, B, 0, 3>(s, g, S, best, verbose);
            R3O(64, 64, 2, 2) R3O(128, 64, 4, 2) R3O(128, 128, 4, 2) R3O(32, 64, 2, 2) R3O(64, 32, 2, 2) R3O(128, 128, 2, 2)
            run3<64, 64, 2, 2>(s, g, S, best, verbose);
            run3<128, 64, 2, 2>(s, g, S, best, verbose);
            run3<128, 64, 4, 2>(s, g, S, best, verbose);
            run3<128, 128, 2, 2>(s, g, S, best, verbose);
            run3<128, 128, 4, 2>(s, g, S, best, verbose);
            run3<64, 128, 2, 2>(s, g, S, best, verbose);
            run3<32, 64, 2, 2>(s, g, S, best, verbose);
            run3<64, 32, 2, 2>(s, g, S, best, verbose);
        }
        const double tf = 2.0 * g.M * g.N * g.K / best.us / 1e6;
        printf("  BEST %-34s %8.2f us  %6.1f TF  frac %.3f\n", best.what, best.us, tf, tf / 157.3);
    }
    return 0;
}
