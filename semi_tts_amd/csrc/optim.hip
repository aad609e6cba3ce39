// optim.hip -- the optimiser half of the training step as multi-tensor kernels.
// ref: BaseSolver.backward src/solver.py:138-151 (clip_grad_norm_(5.0) then optimizer.step()), src/optim.py
// (torch.optim.Adam with its defaults).  ~100 parameter tensors are processed by ONE launch per pass: the tensor
// pointers and the block -> (tensor, chunk) map sit in a cached device-side table (mt_table; the addresses repeat from step
// to step).  During a stream capture, or when the table cannot be made, they travel in the kernel arguments instead,
// up to MT_T tensors / MT_NB blocks per launch.
#include <mutex>
#include "st_common.h"

namespace {

constexpr int MT_T = 24, MT_NB = 640, MT_CHUNK = 32768, MT_THREADS = 256;

struct MtOp {
    float* partial; int partial_base;            // sumsq
    const float* norm; float max_norm;           // scale
    float pre_scale;                             // scale: every gradient is first multiplied by this (1 / world: the ranks' SUM -> average)
    float b1, b2, eps, step_size, bc2_sqrt;      // adam
};
struct MtArgs {
    float* p[MT_T]; float* g[MT_T]; float* m[MT_T]; float* v[MT_T];
    long n[MT_T];
    unsigned char blk_tensor[MT_NB];
    unsigned short blk_chunk[MT_NB];         // chunk index inside the tensor (tensors up to 2^31 elements)
    MtOp op;
};
// The same map in DEVICE memory (st_mt_* keep a small cache of such tables keyed on the pointer lists: parameters, moments and -- in steady
// state, from the caching allocator or the reducer's bucket slots -- gradients sit at the same addresses every step): any number of
// tensors and blocks in ONE launch.  With the map in the kernel arguments an optimiser step over the model's ~110 tensors was five
// launches per pass, three of them a few dozen blocks on a 256-unit chip.
struct MtTab { float* const* p; float* const* g; float* const* m; float* const* v; const long* n; const unsigned* blk; };

// OP 0: partial[block] = sum g^2;  OP 1: g *= pre_scale * min(1, max_norm / (norm + 1e-6));  OP 2: Adam update
template <int OP>
__device__ __forceinline__ void mt_body(float* __restrict__ tp, float* __restrict__ g, float* __restrict__ tm, float* __restrict__ tv,
                                        const long n, const long beg, const MtOp& a, float* red) {
    const long end = beg + MT_CHUNK < n ? beg + MT_CHUNK : n;
    // 16-byte accesses when the chunk is aligned (torch allocations are; chunk starts are multiples of 32768 floats)
    const bool vec = (((uintptr_t)g | (uintptr_t)tp | (uintptr_t)tm | (uintptr_t)tv) & 15u) == 0;
    const long nvec = vec ? ((end - beg) >> 2) : 0;
    if (OP == 0) {
        float acc = 0.0f, acc2 = 0.0f;
        const f32x4* g4 = reinterpret_cast<const f32x4*>(g + beg);
        long i = threadIdx.x;
        // (eight pieces in flight per thread; the two running sums keep the pairing of the two-piece form: pieces 0, 2, 4, 6 of a round
        // go to acc, 1, 3, 5, 7 to acc2 -- the same additions in the same order as four two-piece rounds)
        for (; i + 7 * MT_THREADS < nvec; i += 8 * MT_THREADS) {
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = g4[i + u * MT_THREADS];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc = fmaf(x[u][0], x[u][0], acc); acc = fmaf(x[u][1], x[u][1], acc); acc = fmaf(x[u][2], x[u][2], acc); acc = fmaf(x[u][3], x[u][3], acc);
                acc2 = fmaf(x[u + 1][0], x[u + 1][0], acc2); acc2 = fmaf(x[u + 1][1], x[u + 1][1], acc2); acc2 = fmaf(x[u + 1][2], x[u + 1][2], acc2); acc2 = fmaf(x[u + 1][3], x[u + 1][3], acc2);
            }
        }
        for (; i + MT_THREADS < nvec; i += 2 * MT_THREADS) {
            const f32x4 x = g4[i], y = g4[i + MT_THREADS];
            acc = fmaf(x[0], x[0], acc); acc = fmaf(x[1], x[1], acc); acc = fmaf(x[2], x[2], acc); acc = fmaf(x[3], x[3], acc);
            acc2 = fmaf(y[0], y[0], acc2); acc2 = fmaf(y[1], y[1], acc2); acc2 = fmaf(y[2], y[2], acc2); acc2 = fmaf(y[3], y[3], acc2);
        }
        for (; i < nvec; i += MT_THREADS) { const f32x4 x = g4[i]; acc = fmaf(x[0], x[0], acc); acc = fmaf(x[1], x[1], acc); acc = fmaf(x[2], x[2], acc); acc = fmaf(x[3], x[3], acc); }
        for (long j = beg + nvec * 4 + threadIdx.x; j < end; j += MT_THREADS) { const float x = g[j]; acc = fmaf(x, x, acc); }
        acc = st_wave_sum(acc + acc2);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.0f;
            for (int w = 0; w < MT_THREADS / 64; ++w) s += red[w];
            a.partial[a.partial_base + blockIdx.x] = s;
        }
    } else if (OP == 1) {
        // torch.nn.utils.clip_grad_norm_ on gradients that are still the SUM over the ranks: *norm is the norm of the averaged
        // gradients (st_mt_grad_norm_scaled), and the 1 / world of the average rides in this launch (pre_scale = 1: the plain clip)
        float coef = a.max_norm / (*a.norm + 1e-6f);
        coef = (coef < 1.0f ? coef : 1.0f) * a.pre_scale;
        if (coef != 1.0f) {
            f32x4* g4 = reinterpret_cast<f32x4*>(g + beg);
            for (long i = threadIdx.x; i < nvec; i += MT_THREADS) { f32x4 x = g4[i]; x[0] *= coef; x[1] *= coef; x[2] *= coef; x[3] *= coef; g4[i] = x; }
            for (long j = beg + nvec * 4 + threadIdx.x; j < end; j += MT_THREADS) g[j] *= coef;
        }
    } else {
        // guarded form (st_mt_adam_guarded): no update at all when the gradient norm of this step is NaN / inf -- the decision the
        // reference takes on the host (`if math.isnan(grad_norm)`: skip optimizer.step(), src/solver.py:147-150) without a host round trip
        if (a.norm && !(fabsf(*a.norm) <= 3.0e38f)) return;
        float* __restrict__ p = tp; float* __restrict__ m = tm; float* __restrict__ v = tv;
        const float omb1 = 1.0f - a.b1, omb2 = 1.0f - a.b2;
        // no contraction left to the compiler (the fused multiply-adds are the explicit ones): the 16-byte loop and the scalar loop
        // (unaligned tensors, tails) must round alike -- a gradient that lives in an all-reduce bucket slot and one in a tensor of
        // its own then give the same weights bit for bit (r04: the two loops were contracted differently; invisible in a one-step test)
        auto upd = [&](float gr, float& pi, float& mi, float& vi) {
#pragma clang fp contract(off)
            mi = __builtin_fmaf(gr - mi, omb1, mi);                               // exp_avg.lerp_(grad, 1 - beta1)
            const float g2 = gr * gr, vb = vi * a.b2;
            vi = __builtin_fmaf(omb2, g2, vb);                                    // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
            const float sq = sqrtf(vi) / a.bc2_sqrt;
            const float denom = sq + a.eps;
            const float q = mi / denom;
            pi = __builtin_fmaf(-a.step_size, q, pi);                             // param.addcdiv_(exp_avg, denom, value=-step_size)
        };
        const f32x4* g4 = reinterpret_cast<const f32x4*>(g + beg);
        f32x4* p4 = reinterpret_cast<f32x4*>(p + beg); f32x4* m4 = reinterpret_cast<f32x4*>(m + beg); f32x4* v4 = reinterpret_cast<f32x4*>(v + beg);
        // four 16-byte pieces per thread and round: sixteen loads in flight before the first update (one piece per round left a
        // compute unit with 4 x 256 x 16 B = 16 KB in flight: 3.0 TB/s over the 680 MB of an update)
        long i = threadIdx.x;
        for (; i + 3 * MT_THREADS < nvec; i += 4 * MT_THREADS) {
            f32x4 gg[4], pp[4], mm[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { gg[u] = g4[i + u * MT_THREADS]; pp[u] = p4[i + u * MT_THREADS]; mm[u] = m4[i + u * MT_THREADS]; vv[u] = v4[i + u * MT_THREADS]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { float pi = pp[u][c], mi = mm[u][c], vi = vv[u][c]; upd(gg[u][c], pi, mi, vi); pp[u][c] = pi; mm[u][c] = mi; vv[u][c] = vi; }
                p4[i + u * MT_THREADS] = pp[u]; m4[i + u * MT_THREADS] = mm[u]; v4[i + u * MT_THREADS] = vv[u];
            }
        }
        for (; i < nvec; i += MT_THREADS) {
            const f32x4 gg = g4[i];
            f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) { float pi = pp[c], mi = mm[c], vi = vv[c]; upd(gg[c], pi, mi, vi); pp[c] = pi; mm[c] = mi; vv[c] = vi; }
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
        }
        for (long j = beg + nvec * 4 + threadIdx.x; j < end; j += MT_THREADS) upd(g[j], p[j], m[j], v[j]);
    }
}

template <int OP>
__global__ __launch_bounds__(MT_THREADS) void mt_kernel(const MtArgs a) {
    __shared__ float red[MT_THREADS / 64];
    const int t = a.blk_tensor[blockIdx.x];
    mt_body<OP>(a.p[t], a.g[t], a.m[t], a.v[t], a.n[t], (long)a.blk_chunk[blockIdx.x] * MT_CHUNK, a.op, red);
}

template <int OP>
__global__ __launch_bounds__(MT_THREADS) void mt_tab_kernel(const MtTab tb, const MtOp op) {
    __shared__ float red[MT_THREADS / 64];
    const unsigned e = tb.blk[blockIdx.x];
    const int t = (int)(e >> 16);
    mt_body<OP>(tb.p ? tb.p[t] : nullptr, tb.g[t], tb.m ? tb.m[t] : nullptr, tb.v ? tb.v[t] : nullptr, tb.n[t], (long)(e & 0xFFFFu) * MT_CHUNK, op, red);
}

__global__ __launch_bounds__(64) void mt_norm_final_kernel(const float* partial, int n, float* out, float pre_scale) {
    // fixed-order sum of the per-block partials (a few hundred) by one wave (lane j: partials j, j + 64, ...; then a fixed tree over
    // the lanes), then the square root.  (One thread walking them all took 29 us.)
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = st_wave_sum(s);
    if (threadIdx.x == 0) *out = pre_scale == 1.0f ? sqrtf(s) : sqrtf(s) * pre_scale;
}

// Many SMALL tensors copied by one launch (the gradients autograd allocated outside their all-reduce bucket: BatchNorm / bias / highway
// parameters, ~50 tensors of 80 ... 370k floats): up to MC_T tensors per launch, 4096 floats per workgroup (the optimiser's 32768-float
// chunks make such a launch as long as its largest tensor on ONE compute unit: 11 us).
constexpr int MC_T = 96, MC_NB = 512, MC_CHUNK = 4096;
struct McArgs {
    float* dst[MC_T]; const float* src[MC_T]; int n[MC_T];
    unsigned char blk_tensor[MC_NB]; unsigned short blk_chunk[MC_NB];
};

__global__ __launch_bounds__(MT_THREADS) void mc_kernel(const McArgs a) {
    const int t = a.blk_tensor[blockIdx.x];
    const int beg = (int)a.blk_chunk[blockIdx.x] * MC_CHUNK;
    const int end = min(beg + MC_CHUNK, a.n[t]);
    const float* __restrict__ s = a.src[t];
    float* __restrict__ d = a.dst[t];
    const bool vec = (((uintptr_t)s | (uintptr_t)d) & 15u) == 0;
    const int nvec = vec ? ((end - beg) >> 2) : 0;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(s + beg);
    f32x4* d4 = reinterpret_cast<f32x4*>(d + beg);
    for (int i = threadIdx.x; i < nvec; i += MT_THREADS) d4[i] = s4[i];
    for (int j = beg + nvec * 4 + threadIdx.x; j < end; j += MT_THREADS) d[j] = s[j];
}

// ---- the block map in device memory --------------------------------------------------------------------------------------------------
struct MtTabEntry { unsigned long long key; int dev; void* buf; void* host; size_t cap; hipEvent_t ev; bool ev_set; int nt, blocks; unsigned long long age; hipStream_t stream; int present; };
constexpr int MT_TABS = 16;
static MtTabEntry mt_tabs[MT_TABS];
static unsigned long long mt_age = 0;
static long mt_misses = 0;
static std::mutex mt_mutex;       // the table is process-wide state: two host threads (two streams) must not rebuild entries at once

static unsigned long long mt_hash(unsigned long long h, const void* data, size_t bytes) {
    const unsigned char* c = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}

// the device table of (p, g, m, v, n) (lists that are absent stay null): cached; else built in the entry's pinned staging buffer and
// uploaded by an asynchronous copy ON THE LAUNCH STREAM (a few KB; the host does not wait for the device: gradients that autograd
// allocates afresh change their addresses from step to step and miss every time).  An entry's buffers are reused in place -- the
// copy is ordered behind the launches that read the old contents, and the staging buffer is only rewritten after the event behind its
// last copy has passed.  All launches of a process that share tables must be on one stream (they are: torch's current stream).
// nullptr: no table (a stream capture is in progress, or a HIP call failed): the caller uses the kernel-argument form.
static const MtTabEntry* mt_table(float* const* p, float* const* g, float* const* m, float* const* v, const long* n, int nt, hipStream_t st, MtTab& tb) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    unsigned long long key = 1469598103934665603ull;
    const int present = (p ? 1 : 0) | (m ? 2 : 0) | (v ? 4 : 0);
    key = mt_hash(key, &present, sizeof(present));
    key = mt_hash(key, &nt, sizeof(nt));
    if (p) key = mt_hash(key, p, sizeof(float*) * nt);
    key = mt_hash(key, g, sizeof(float*) * nt);
    if (m) key = mt_hash(key, m, sizeof(float*) * nt);
    if (v) key = mt_hash(key, v, sizeof(float*) * nt);
    key = mt_hash(key, n, sizeof(long) * nt);
    std::lock_guard<std::mutex> lock(mt_mutex);
    MtTabEntry* hit = nullptr;
    MtTabEntry* victim = &mt_tabs[0];
    const size_t off_g = (size_t)nt * 8, off_m = 2 * off_g, off_v = 3 * off_g, off_n = 4 * off_g, off_blk = 5 * off_g;
    for (int i = 0; i < MT_TABS; ++i) {
        MtTabEntry& e = mt_tabs[i];
        // A hit is the same LISTS, not just the same hash: the entry's host staging copy holds exactly what was uploaded (advisor, round 5:
        // a 64-bit collision would let Adam write through stale pointers) -- and the same stream: the upload is ordered on the stream of
        // the miss only, a launch on another stream could run before it.
        if (e.buf && e.key == key && e.dev == dev && e.nt == nt && e.stream == st && e.present == present) {
            const unsigned char* h = static_cast<const unsigned char*>(e.host);
            const bool same = (!p || memcmp(h, p, off_g) == 0) && memcmp(h + off_g, g, off_g) == 0 && (!m || memcmp(h + off_m, m, off_g) == 0) &&
                              (!v || memcmp(h + off_v, v, off_g) == 0) && memcmp(h + off_n, n, off_g) == 0;
            if (same) { hit = &e; break; }
        }
        if (e.age < victim->age) victim = &e;
    }
    if (!hit) {
        ++mt_misses;
        size_t blocks = 0;
        for (int t = 0; t < nt; ++t) if (n[t] > 0) {
            const size_t ch = (size_t)((n[t] + MT_CHUNK - 1) / MT_CHUNK);
            if (ch > 65535 || t > 65535) return nullptr;
            blocks += ch;
        }
        if (blocks == 0 || blocks > (1u << 30)) return nullptr;
        const size_t bytes = off_blk + blocks * sizeof(unsigned);
        MtTabEntry& e = *victim;
        if (e.ev_set && hipEventSynchronize(e.ev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (e.dev != dev || e.cap < bytes) {          // (first use of the entry, another device, or a longer list: new buffers)
            if (e.buf) (void)hipFree(e.buf);
            if (e.host) (void)hipHostFree(e.host);
            if (e.ev_set) (void)hipEventDestroy(e.ev);
            e.buf = e.host = nullptr; e.ev_set = false; e.cap = 0;
            const size_t cap = (bytes + 65535) / 65536 * 65536;
            if (hipMalloc(&e.buf, cap) != hipSuccess || hipHostMalloc(&e.host, cap, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                if (e.buf) (void)hipFree(e.buf);
                if (e.host) (void)hipHostFree(e.host);
                e.buf = e.host = nullptr;
                return nullptr;
            }
            e.cap = cap; e.ev_set = true; e.dev = dev;
        }
        unsigned char* host = static_cast<unsigned char*>(e.host);
        memset(host, 0, off_blk);
        if (p) memcpy(host, p, off_g);
        memcpy(host + off_g, g, off_g);
        if (m) memcpy(host + off_m, m, off_g);
        if (v) memcpy(host + off_v, v, off_g);
        memcpy(host + off_n, n, off_g);
        unsigned* blk = reinterpret_cast<unsigned*>(host + off_blk);
        size_t bi = 0;
        for (int t = 0; t < nt; ++t) if (n[t] > 0) {
            const unsigned ch = (unsigned)((n[t] + MT_CHUNK - 1) / MT_CHUNK);
            for (unsigned c = 0; c < ch; ++c) blk[bi++] = ((unsigned)t << 16) | c;
        }
        if (hipMemcpyAsync(e.buf, e.host, bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(e.ev, st) != hipSuccess) {
            (void)hipGetLastError();
            e.key = 0; e.nt = -1;
            return nullptr;
        }
        e.key = key; e.nt = nt; e.blocks = (int)blocks; e.stream = st; e.present = present;
        hit = &e;
    }
    hit->age = ++mt_age;
    unsigned char* b = static_cast<unsigned char*>(hit->buf);
    tb.p = p ? reinterpret_cast<float* const*>(b) : nullptr;
    tb.g = reinterpret_cast<float* const*>(b + off_g);
    tb.m = m ? reinterpret_cast<float* const*>(b + off_m) : nullptr;
    tb.v = v ? reinterpret_cast<float* const*>(b + off_v) : nullptr;
    tb.n = reinterpret_cast<const long*>(b + off_n);
    tb.blk = reinterpret_cast<const unsigned*>(b + off_blk);
    return hit;
}

template <int OP>
int mt_run(MtArgs& a, float* const* p, float* const* g, float* const* m, float* const* v, const long* n, int nt, hipStream_t st,
           int* blocks_total) {
    {   // ONE launch over a device-side block map (same block order, hence the same partial-sum order, as the launches below)
        MtTab tb;
        const MtTabEntry* e = mt_table(p, g, m, v, n, nt, st, tb);
        if (e) {
            a.op.partial_base = 0;
            hipLaunchKernelGGL((mt_tab_kernel<OP>), dim3(e->blocks), dim3(MT_THREADS), 0, st, tb, a.op);
            ST_LAUNCH_CHECK();
            if (blocks_total) *blocks_total = e->blocks;
            return 0;
        }
    }
    int ti = 0, bl = 0, base = 0;
    auto flush = [&]() -> int {
        if (bl == 0) { ti = 0; return 0; }
        a.op.partial_base = base;
        hipLaunchKernelGGL((mt_kernel<OP>), dim3(bl), dim3(MT_THREADS), 0, st, a);
        ST_LAUNCH_CHECK();
        base += bl; ti = 0; bl = 0;
        return 0;
    };
    for (int t = 0; t < nt; ++t) {
        if (n[t] <= 0) continue;
        const int chunks = (int)((n[t] + MT_CHUNK - 1) / MT_CHUNK);
        ST_CHECK_ARG(chunks <= 65535, "multi-tensor op: tensor %d has %ld elements (> 2^31)", t, n[t]);
        int c = 0;
        while (c < chunks) {
            if (ti == MT_T || bl == MT_NB) { int rc = flush(); if (rc) return rc; }
            // (re)register this tensor in the current launch
            a.p[ti] = p ? p[t] : nullptr; a.g[ti] = g[t]; a.m[ti] = m ? m[t] : nullptr; a.v[ti] = v ? v[t] : nullptr; a.n[ti] = n[t];
            while (c < chunks && bl < MT_NB) { a.blk_tensor[bl] = (unsigned char)ti; a.blk_chunk[bl] = (unsigned short)c; ++bl; ++c; }
            ++ti;
        }
    }
    int rc = flush();
    if (rc) return rc;
    if (blocks_total) *blocks_total = base;
    return 0;
}

}  // namespace

// (diagnostics: how often a block map had to be built and uploaded -- once per set of tensor addresses)
extern "C" long st_mt_table_misses(void) { return mt_misses; }

extern "C" size_t st_mt_blocks(const long* n, int nt) {
    size_t b = 0;
    for (int t = 0; t < nt; ++t) if (n[t] > 0) b += (size_t)((n[t] + MT_CHUNK - 1) / MT_CHUNK);
    return b;
}

extern "C" int st_mt_grad_norm_scaled(float* const* g, const long* n, int nt, float* partials, float* norm_out, float pre_scale, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(g && n && nt > 0 && partials && norm_out && pre_scale > 0.0f, "st_mt_grad_norm: bad arguments");
    MtArgs a;
    memset(&a, 0, sizeof(a));
    a.op.partial = partials;
    int total = 0;
    int rc = mt_run<0>(a, nullptr, g, nullptr, nullptr, n, nt, (hipStream_t)stream, &total);
    if (rc) return rc;
    hipLaunchKernelGGL(mt_norm_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, total, norm_out, pre_scale);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_mt_grad_norm(float* const* g, const long* n, int nt, float* partials, float* norm_out, void* stream) {
    return st_mt_grad_norm_scaled(g, n, nt, partials, norm_out, 1.0f, stream);
}

extern "C" int st_mt_clip_scale_pre(float* const* g, const long* n, int nt, const float* norm, float max_norm, float pre_scale, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(g && n && nt > 0 && norm && max_norm > 0.0f && pre_scale > 0.0f, "st_mt_clip_scale: bad arguments");
    MtArgs a;
    memset(&a, 0, sizeof(a));
    a.op.norm = norm; a.op.max_norm = max_norm; a.op.pre_scale = pre_scale;
    return mt_run<1>(a, nullptr, g, nullptr, nullptr, n, nt, (hipStream_t)stream, nullptr);
}

extern "C" int st_mt_clip_scale(float* const* g, const long* n, int nt, const float* norm, float max_norm, void* stream) {
    return st_mt_clip_scale_pre(g, n, nt, norm, max_norm, 1.0f, stream);
}

extern "C" int st_mt_copy(float* const* dst, float* const* src, const long* n, int nt, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dst && src && n && nt > 0, "st_mt_copy: bad arguments");
    McArgs a;
    int ti = 0, bl = 0;
    auto flush = [&]() -> int {
        if (bl == 0) { ti = 0; return 0; }
        hipLaunchKernelGGL(mc_kernel, dim3(bl), dim3(MT_THREADS), 0, (hipStream_t)stream, a);
        ST_LAUNCH_CHECK();
        ti = 0; bl = 0;
        return 0;
    };
    for (int t = 0; t < nt; ++t) {
        if (n[t] <= 0) continue;
        ST_CHECK_ARG(n[t] < (long)65535 * MC_CHUNK, "st_mt_copy: tensor %d has %ld elements", t, n[t]);
        const int chunks = (int)((n[t] + MC_CHUNK - 1) / MC_CHUNK);
        int c = 0;
        while (c < chunks) {
            if (ti == MC_T || bl == MC_NB) { int rc = flush(); if (rc) return rc; }
            a.dst[ti] = dst[t]; a.src[ti] = src[t]; a.n[ti] = (int)n[t];
            while (c < chunks && bl < MC_NB) { a.blk_tensor[bl] = (unsigned char)ti; a.blk_chunk[bl] = (unsigned short)c; ++bl; ++c; }
            ++ti;
        }
    }
    return flush();
}
extern "C" int st_mt_adam_guarded(float* const* p, float* const* g, float* const* m, float* const* v, const long* n, int nt,
                                  float beta1, float beta2, float eps, float step_size, float bias_correction2_sqrt,
                                  const float* guard_norm, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(p && g && m && v && n && nt > 0 && bias_correction2_sqrt > 0.0f, "st_mt_adam: bad arguments");
    MtArgs a;
    memset(&a, 0, sizeof(a));
    a.op.b1 = beta1; a.op.b2 = beta2; a.op.eps = eps; a.op.step_size = step_size; a.op.bc2_sqrt = bias_correction2_sqrt;
    a.op.norm = guard_norm;
    return mt_run<2>(a, p, g, m, v, n, nt, (hipStream_t)stream, nullptr);
}

extern "C" int st_mt_adam(float* const* p, float* const* g, float* const* m, float* const* v, const long* n, int nt,
                          float beta1, float beta2, float eps, float step_size, float bias_correction2_sqrt, void* stream) {
    return st_mt_adam_guarded(p, g, m, v, n, nt, beta1, beta2, eps, step_size, bias_correction2_sqrt, nullptr, stream);
}
