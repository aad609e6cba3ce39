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

extern "C" int st_vq_l2_fwd(const float* x, const float* table, const float* temp, float* p_code,
                            int64_t* idx, float* out, int n, int D, int V, void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(x && table && temp && p_code && idx && out && n > 0 && D > 0 && V > 0, "st_vq_l2_fwd: bad arguments");
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
