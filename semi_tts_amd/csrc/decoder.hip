// decoder.hip -- the autoregressive decode loop of Decoder.forward, issued natively.
//
// ref: src/module.py:184-206 (loop + next-input policy) and :216-288 (one step).
// Once per forward the six weight matrices the loop streams every step are packed into MFMA
// lane order (skinny_packed.hip, "P16"); all per-step activations live in the tiled "T16"
// layout, written that way by their producers.  Per step the loop enqueues, in dependency order:
//   1. query LSTM cell      x = [dec_in | ctx_{t-1} | h_q] (ONE tiled buffer per step), fused AdaIN of the new h_q
//   2. query projection     pq = W_q h_q
//   3. attention step       energies, softmax, context, cumulative weights   (attention.hip)
//   4. decoder LSTM cell    x = [ctx_t | adapted h_q | h_d]
//   5. proj + gate          [h_d | ctx_t] -> r mel frames + stop logit
//   6. next input           teacher frame (tiled copy) and/or prenet(own output) (2 launches)
// There is no host synchronisation, allocation or blocking copy inside, so the whole loop
// can be captured into a hipGraph (st_graph_begin/st_graph_end) and replayed as one launch.
// State lives in caller-owned tapes indexed by step; nothing is overwritten, so the same
// buffers are the saved tensors of the backward pass.
#include "st_common.h"
#include "loop_graph.h"
#include <cstdlib>

namespace {

// zero up to twelve buffers in one launch (blockIdx.y = buffer): the tapes' initial slots and the hand-off words of a forward
struct ZeroArgs { float* p[12]; size_t floats[12]; int n; };
__global__ __launch_bounds__(256) void zero_regions_kernel(const ZeroArgs a) {
    float* __restrict__ p = a.p[blockIdx.y];
    const size_t n = a.floats[blockIdx.y];
    const size_t n4 = (((uintptr_t)p & 15u) == 0) ? n / 4 : 0;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) reinterpret_cast<f32x4*>(p)[i] = z;
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0f;
}

}  // namespace

extern "C" size_t st_packed_weight_floats(const int* k, int nseg, int N, int lstm_H);
extern "C" size_t st_t16_floats(int B, int K);

namespace {

struct PackedLayout {   // offsets (floats) of the six packed matrices inside the packed buffer
    size_t q, pq, d, pg, p0, p1, total;
};

PackedLayout packed_layout(const st_decoder_dims* d) {
    PackedLayout o;
    const int in_dim = d->r * d->n_mels;
    size_t p = 0;
    { int k[3] = {d->P, d->E, d->Q}; o.q = p; p += st_packed_weight_floats(k, 3, 4 * d->Q, d->Q); }
    { int k[1] = {d->Q}; o.pq = p; p += st_packed_weight_floats(k, 1, d->A, 0); }
    { int k[3] = {d->E, d->Q, d->D}; o.d = p; p += st_packed_weight_floats(k, 3, 4 * d->D, d->D); }
    { int k[2] = {d->D, d->E}; o.pg = p; p += st_packed_weight_floats(k, 2, in_dim + 1 + (d->fuse_pre0 ? d->P : 0), 0); }
    { int k[1] = {in_dim}; o.p0 = p; p += st_packed_weight_floats(k, 1, d->P, 0); }
    { int k[1] = {d->P}; o.p1 = p; p += st_packed_weight_floats(k, 1, d->P, 0); }
    o.total = p;
    return o;
}

inline int kb16(int k) { return (k + 15) >> 4; }

struct StepViews {   // k-block geometry of the three step-input buffers
    int q_kbs, q_ctx, q_h;        // xq = [dec_in | ctx | h_q]
    int d_kbs, d_ha, d_h;         // xd = [ctx | hadapt | h_d]
    int o_kbs, o_ctx;             // xo = [h_d | ctx]
    size_t q_floats, d_floats, o_floats;
};

StepViews step_views(const st_decoder_dims* d) {
    StepViews v;
    const int bt = (d->B + 15) >> 4;
    v.q_ctx = kb16(d->P); v.q_h = v.q_ctx + kb16(d->E); v.q_kbs = v.q_h + kb16(d->Q);
    v.d_ha = kb16(d->E); v.d_h = v.d_ha + kb16(d->Q); v.d_kbs = v.d_h + kb16(d->D);
    v.o_ctx = kb16(d->D); v.o_kbs = v.o_ctx + kb16(d->E);
    v.q_floats = (size_t)bt * v.q_kbs * 256; v.d_floats = (size_t)bt * v.d_kbs * 256; v.o_floats = (size_t)bt * v.o_kbs * 256;
    return v;
}

// dec_in_{s} = teacher frame min(s-1, Tt-1) for every step s = 1..steps-1 and row b < Bt, written straight into the
// dec_in k-blocks of the xq tape (slot s) in T16 order: one launch instead of one st_tile_rows per step
__global__ __launch_bounds__(256) void tile_teacher_kernel(const float* teacher_pre, int Tt, float* xq_tape, size_t slot_floats,
                                                           int q_kbs, int steps, int Bt, int P) {
    const size_t total = (size_t)(steps - 1) * Bt * P;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % P);
        const size_t sb = i / P;
        const int b = (int)(sb % Bt), s = (int)(sb / Bt) + 1;
        const int frame = s - 1 < Tt ? s - 1 : Tt - 1;
        const float v = teacher_pre[((size_t)b * Tt + frame) * P + k];
        xq_tape[(size_t)s * slot_floats + (((size_t)(b >> 4) * q_kbs + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (b & 15)) * 4 + (k & 3)] = v;
    }
}

int prenet_own(const st_decoder_weights* w, const st_decoder_dims* d, const st_decoder_io* io, const PackedLayout& pl, const StepViews& sv,
               int t, bool layer1_done, void* stream) {
    // dec_in_{t+1} = prenet(mel_t) for every row                ref: src/module.py:192,:197-198,:205-206
    const int in_dim = d->r * d->n_mels;
    const size_t BP = (size_t)d->B * d->P;
    const float* m1 = io->prenet_mask ? io->prenet_mask + ((size_t)t * 2 + 0) * BP : nullptr;
    const float* m2 = io->prenet_mask ? io->prenet_mask + ((size_t)t * 2 + 1) * BP : nullptr;
    st_t16_view mel = {io->mel_t16, kb16(in_dim), 0};
    st_t16_view pre1 = {io->pre1_t16 + (size_t)t * io->pre1_step_floats, kb16(d->P), 0};
    int rc = 0;
    if (d->prenet_norm) {
        // normalised prenet (Linear -> LayerNorm / BatchNorm1d -> ReLU -> dropout, src/module.py:337-339,508-521): the Linear's
        // output stays in natural layout, the norm launch applies norm + ReLU + mask and writes the T16 operand of the next product
        st_t16_view next_n = {io->xq_tape + (size_t)(t + 1) * sv.q_floats, sv.q_kbs, 0};
        const st_t16_view* dsts[2] = {&pre1, &next_n};
        const st_t16_view* srcs[2] = {&mel, &pre1};
        const float* pw[2] = {io->packed + pl.p0, io->packed + pl.p1};
        const int Ks[2] = {16 * kb16(in_dim), 16 * kb16(d->P)};
        const float* ms[2] = {m1, m2};
        // the rows that feed back: all of them on an own-output step, the ones without a teacher otherwise -- the batch of a
        // BatchNorm1d step (the products run over every row; rows < b0 of the next input come from the teacher afterwards)
        const int b0 = io->step_src[t] == -1 ? 0 : io->Bt;
        for (int l = 0; l < 2; ++l) {
            float* y = io->pre_nat_tape ? io->pre_nat_tape + ((size_t)t * 2 + l) * BP : io->pre_nat;
            rc = st_skinny_linear_packed_fwd(pw[l], srcs[l], Ks[l], nullptr, ST_ACT_NONE, nullptr, 0, y, d->P, nullptr, 0,
                                             nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, d->B, d->P, stream);
            if (rc) return rc;
            rc = st_prenet_norm_fwd(y, d->P, d->prenet_norm, w->pre_norm_w[l], w->pre_norm_b[l], w->pre_norm_rm[l], w->pre_norm_rv[l],
                                    w->pre_norm_nbt[l], w->pre_norm_eps, w->pre_norm_momentum, ms[l], d->P, dsts[l], b0, d->B, d->P, stream);
            if (rc) return rc;
        }
        return 0;
    }
    if (!layer1_done)
        rc = st_skinny_linear_packed_fwd(io->packed + pl.p0, &mel, 16 * kb16(in_dim), nullptr, ST_ACT_RELU, m1, d->P,
                                         nullptr, 0, &pre1, 0, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, d->B, d->P, stream);
    if (rc) return rc;
    st_t16_view next = {io->xq_tape + (size_t)(t + 1) * sv.q_floats, sv.q_kbs, 0};
    return st_skinny_linear_packed_fwd(io->packed + pl.p1, &pre1, 16 * kb16(d->P), nullptr, ST_ACT_RELU, m2, d->P,
                                       nullptr, 0, &next, 0, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, d->B, d->P, stream);
}

}  // namespace

extern "C" int st_decoder_gate_split_k(const st_decoder_dims* d) {
    if (!d) return 0;
    const StepViews sv = step_views(d);
    int hosted = (sv.d_kbs / 2) / 16 * 16;                 // k-blocks that ride beside pq / fin: half the reduction, whole rounds of 16
    if (hosted < 16 || hosted > sv.d_kbs - sv.d_ha) hosted = sv.d_kbs - sv.d_ha;
    return 16 * (sv.d_kbs - hosted);
}

extern "C" size_t st_decoder_packed_floats(const st_decoder_dims* d) { return d ? packed_layout(d).total : 0; }

extern "C" int st_decoder_pack(const st_decoder_weights* w, const st_decoder_dims* d, float* packed, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(w && d && packed, "st_decoder_pack: null pointer");
    const PackedLayout pl = packed_layout(d);
    const int in_dim = d->r * d->n_mels;
    st_pack_job j[6];
    memset(j, 0, sizeof(j));
    auto seg = [&](st_pack_job& q, int s, const float* wp, int ld, int k) { q.w[s] = wp; q.ldw[s] = ld; q.k[s] = k; q.nseg = s + 1; };
    // query LSTM: K = [dec_in (P) | ctx (E) | h_q (Q)]                       ref: src/module.py:227-228
    seg(j[0], 0, w->q_w_ih, d->P + d->E, d->P); seg(j[0], 1, w->q_w_ih + d->P, d->P + d->E, d->E); seg(j[0], 2, w->q_w_hh, d->Q, d->Q);
    j[0].N = 4 * d->Q; j[0].lstm_H = d->Q; j[0].packed = packed + pl.q;
    seg(j[1], 0, w->attn_query_w, d->Q, d->Q); j[1].N = d->A; j[1].packed = packed + pl.pq;
    // decoder LSTM: K = [ctx (E) | adapted h_q (Q) | h_d (D)]                 ref: src/module.py:275-277
    seg(j[2], 0, w->d_w_ih, d->E + d->Q, d->E); seg(j[2], 1, w->d_w_ih + d->E, d->E + d->Q, d->Q); seg(j[2], 2, w->d_w_hh, d->D, d->D);
    j[2].N = 4 * d->D; j[2].lstm_H = d->D; j[2].packed = packed + pl.d;
    // proj (+) gate: K = [h_d (D) | ctx (E)]                                  ref: src/module.py:282-287
    seg(j[3], 0, w->projgate_w, d->D + d->E, d->D); seg(j[3], 1, w->projgate_w + d->D, d->D + d->E, d->E);
    j[3].N = in_dim + 1 + (d->fuse_pre0 ? d->P : 0); j[3].packed = packed + pl.pg;
    seg(j[4], 0, w->prenet_w0, in_dim, in_dim); j[4].N = d->P; j[4].packed = packed + pl.p0;
    seg(j[5], 0, w->prenet_w1, d->P, d->P); j[5].N = d->P; j[5].packed = packed + pl.p1;
    return st_pack_weight_batch(j, 6, stream);       // one launch for the six (they were ~10 us each)
}

extern "C" size_t st_decoder_tape_floats(const st_decoder_dims* d, int which) {
    if (!d) return 0;
    const StepViews sv = step_views(d);
    return which == 0 ? sv.q_floats : (which == 1 ? sv.d_floats : sv.o_floats);
}

static int decoder_forward_issue(const st_decoder_weights* w, const st_decoder_dims* d, const st_decoder_io* io, void* stream);

static stlg::Cache g_fwd_graphs;
extern "C" void st_loop_graph_stats(long* fwd3, long* bwd3);      // (defined in decoder_bwd.hip: {replays, captures, eager} of both loops)
extern "C" void stx_fwd_loop_graph_stats(long* out3) {
    std::lock_guard<std::mutex> lock(g_fwd_graphs.mu);
    out3[0] = g_fwd_graphs.replays; out3[1] = g_fwd_graphs.captures; out3[2] = g_fwd_graphs.eager;
}

extern "C" int st_decoder_forward(const st_decoder_weights* w, const st_decoder_dims* d, const st_decoder_io* io,
                                  void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(w && d && io, "st_decoder_forward: null struct pointer");
    // the teacher-forced loop of a training step (tapes kept, projection deferred) comes back with bit-identical arguments step after step:
    // captured at the second sighting, replayed with one launch afterwards (loop_graph.h)
    if (io->defer_proj && io->steps > 0 && io->step_src && stlg::enabled(d->B)) {
        st_decoder_io key_io = *io;
        key_io.step_src = nullptr;                       // (a host array: its CONTENTS are part of the key, not its address)
        uint64_t key = stlg::fnv(stlg::FNV0, w, sizeof(*w));
        key = stlg::fnv(key, d, sizeof(*d));
        key = stlg::fnv(key, &key_io, sizeof(key_io));
        key = stlg::fnv(key, io->step_src, sizeof(int) * (size_t)io->steps);
        stlg::Entry* e = nullptr;
        hipStream_t issue_on = (hipStream_t)stream;
        const int mode = stlg::begin(g_fwd_graphs, key, (hipStream_t)stream, &e, &issue_on);
        if (mode == 1) return 0;
        int rc = decoder_forward_issue(w, d, io, (void*)issue_on);
        if (mode == 2) {
            rc = stlg::end(g_fwd_graphs, e, (hipStream_t)stream, rc);
            if (rc == -5) st_set_error("st_decoder_forward: capturing the loop into a hipGraph failed");
        }
        return rc;
    }
    return decoder_forward_issue(w, d, io, stream);
}

static int decoder_forward_issue(const st_decoder_weights* w, const st_decoder_dims* d, const st_decoder_io* io, void* stream) {
    ST_CHECK_ARG(w && d && io, "st_decoder_forward: null struct pointer");
    const int B = d->B, L = d->L, E = d->E, P = d->P, Q = d->Q, D = d->D, A = d->A;
    const int steps = io->steps;
    const int in_dim = d->r * d->n_mels;
    ST_CHECK_ARG(B > 0 && L > 0 && steps > 0 && in_dim > 0, "st_decoder_forward: B=%d L=%d steps=%d", B, L, steps);
    ST_CHECK_ARG(io->memory && io->pm && io->ada_std && io->ada_mean && io->step_src && io->packed,
                 "st_decoder_forward: null input");
    ST_CHECK_ARG(io->mel_out && io->align_out && io->stop_out, "st_decoder_forward: null output");
    ST_CHECK_ARG(io->xq_tape && io->xd_tape && io->xo_tape && io->cq_tape && io->cd_tape && io->wcum_tape &&
                 io->pq_buf && io->pre1_t16 && io->mel_t16 && io->zero_row, "st_decoder_forward: null tape/scratch");
    ST_CHECK_ARG(d->prenet_norm >= 0 && d->prenet_norm <= 3 && (!d->prenet_norm || (!d->fuse_pre0 && io->pre_nat && w->pre_norm_w[0] &&
                 w->pre_norm_w[1] && w->pre_norm_b[0] && w->pre_norm_b[1])), "st_decoder_forward: normalised prenet needs pre_nat, the norm weights and fuse_pre0 = 0");
    for (int t = 0; t + 1 < steps; ++t) {
        const int src = io->step_src[t];
        ST_CHECK_ARG(!d->fuse_pre0 || src == -1, "st_decoder_forward: fuse_pre0 needs every next input to be the own output");
        ST_CHECK_ARG(src >= -2 && (src < 0 || (io->teacher_pre && src < io->Tt)), "st_decoder_forward: step_src[%d]=%d invalid", t, src);
        ST_CHECK_ARG(src != -2 || io->teacher_mean, "st_decoder_forward: step_src[%d]=-2 without teacher_mean", t);
        ST_CHECK_ARG(src == -1 || (io->Bt > 0 && io->Bt <= B), "st_decoder_forward: Bt=%d invalid", io->Bt);
    }
    // pure teacher forcing (every next input is teacher frame min(t, Tt-1), all rows have a teacher): the inputs of all
    // steps are tiled into the tape up front, and with io->defer_proj the projection is left to the caller (one GEMM
    // over all steps instead of one launch per step -- no step's input depends on an earlier output)
    bool pure_tf = io->teacher_pre && io->Bt == B && io->Tt > 0;
    for (int t = 0; t + 1 < steps && pure_tf; ++t) pure_tf = io->step_src[t] == (t < io->Tt ? t : io->Tt - 1);
    const bool defer = io->defer_proj != 0;
    ST_CHECK_ARG(!defer || (pure_tf && !d->fuse_pre0), "st_decoder_forward: defer_proj needs pure teacher forcing");
    hipStream_t st = (hipStream_t)stream;
    const PackedLayout pl = packed_layout(d);
    const StepViews sv = step_views(d);
    const size_t BQ = (size_t)B * Q, BD = (size_t)B * D, BL = (size_t)B * L;
    // The tiled tapes are handed in zero-filled (torch.zeros): slot 0 of xq / the h_d part of xd slot 0
    // are the initial zero state (src/module.py:290-303) and dec_in_0 = prenet(go frame) = 0 (:161,:183).
    // (one launch for all of them, together with the hand-off words further down: eight memset nodes were 25 us of every forward)
    ZeroArgs za;
    memset(&za, 0, sizeof(za));
    auto zero = [&](void* p, size_t bytes) { za.p[za.n] = (float*)p; za.floats[za.n] = bytes / sizeof(float); ++za.n; };
    zero(io->cq_tape, BQ * sizeof(float));
    zero(io->cd_tape, BD * sizeof(float));
    zero(io->wcum_tape, BL * sizeof(float));
    zero(io->zero_row, BL * sizeof(float));
    zero(io->xq_tape, sv.q_floats * sizeof(float));
    zero(io->xd_tape, sv.d_floats * sizeof(float));
    if (io->attn_loc_tape) zero(io->attn_loc_tape, BL * d->F * sizeof(float));      // (slot 0: no step writes it -- there is no history yet)

    const size_t ldmel = (size_t)steps * in_dim;
    const int ldal = steps * L;
    const int Kq = 16 * sv.q_kbs, Kd = 16 * sv.d_kbs, Ko = 16 * sv.o_kbs;
    int rc = 0;
    // Attention in two parts (io->attn_s_buf set by the host): the location part of step t+1 ("pre": conv + W_l + processed
    // memory -> S) only needs the attention weights of step t, so it rides as extra workgroups of a small launch that runs
    // anyway -- the proj launch of step t when the loop has one, otherwise (deferred projection = teacher-forced training) the
    // query-projection launch of step t+1 itself.  The attention launch then starts from S ("fin").
    // Measured and removed: early LSTM inputs on a second stream, as side jobs of the small launches, and as
    // distributed side jobs over four launches of the step -- all slower than the plain chain of dependent launches.
    const bool split_attn = io->attn_s_buf != nullptr;
    const bool pre_in_pq = split_attn && defer;
    const int fp_req = io->attn_fin_parts;
    const int fin_parts = (fp_req == 2 || fp_req == 4 || fp_req == 8) && E % (4 * fp_req) == 0 ? fp_req : 1;
#ifdef ST_ABLATE   // timing experiments only (tools/gpu_ablate.sh builds a SEPARATE library with this macro): skip launches by bit mask
    const int skip = getenv("ST_SKIP") ? atoi(getenv("ST_SKIP")) : 0;
#define ST_SKIPPED(bit) (skip & (1 << (bit)))
#else
#define ST_SKIPPED(bit) 0
#endif
    // query projection + attention fin part as one launch with an in-launch hand-off of pq (st_query_attn_fin_fwd): needs the split
    // attention step, every workgroup of the launch resident at once, and the granule words zeroed before the first step (a memset
    // node when the loop is captured: a replay must not see the tags of the previous one)
    // long texts: the fin part split over position ranges + a combine launch (see at_split_kernel)
    const int sp_parts = io->attn_split_parts;
    const bool fin_split = split_attn && io->attn_split_ws && sp_parts >= 2 && sp_parts <= 64 && A % 4 == 0 && A <= 256 && E % 4 == 0 &&
                           E / 4 <= 512 && 512 % (E / 4) == 0 && (L + sp_parts - 1) / sp_parts <= 512;
    // ... or, when the device holds all its workgroups at once, query projection + fin part over position ranges + combine as ONE launch
    const bool fin_rng = fin_split && !pre_in_pq && io->pq_granules && io->attn_xchg && sp_parts <= 8 && A % 16 == 0 && E % (4 * sp_parts) == 0 &&
                         E / sp_parts <= 256 && (sp_parts - 1) * ((L + sp_parts - 1) / sp_parts) < L && st_query_attn_rng_fits(B, A, sp_parts);
    const bool fuse_pq_fin = !fin_split && split_attn && !pre_in_pq && io->pq_granules && A % 16 == 0 && A <= 256 && E % 4 == 0 &&
                             (A / 16) * ((B + 15) / 16) + B * fin_parts <= st_device_cus();
    if (fin_rng || fuse_pq_fin) zero(io->pq_granules, (size_t)B * A * sizeof(unsigned long long));
    // prenet layer 2 of the next input inside the proj (+) gate (+) prenet-layer-1 launch (st_attn_pre_job.p2_*): free-running steps with
    // the folded layer 1, the split attention step (that launch exists), every workgroup of the launch resident at once
    const int pre_parts_eff = (io->attn_pre_parts >= 2 && io->attn_pre_parts <= 64 && (io->attn_pre_parts & (io->attn_pre_parts - 1)) == 0) ? io->attn_pre_parts : 1;
    const bool fuse_p2 = d->fuse_pre0 && split_attn && !defer && io->pre1_granules && P % 16 == 0 && P <= 512 &&
                         ((in_dim + 1 + P + 15) / 16) * ((B + 15) / 16) + B * pre_parts_eff + (P / 16) * ((B + 15) / 16) <= st_device_cus();
    if (fuse_p2) zero(io->pre1_granules, (size_t)B * P * sizeof(unsigned long long));
    if (fin_rng) zero(io->attn_xchg, st_attn_rng_xchg_words(B, E, sp_parts) * sizeof(unsigned long long));
    {
        size_t most = 0;
        for (int i = 0; i < za.n; ++i) most = za.floats[i] > most ? za.floats[i] : most;
        size_t blocks = (most / 4 + 255) / 256 + 1;
        if (blocks > 256) blocks = 256;
        hipLaunchKernelGGL(zero_regions_kernel, dim3((unsigned)blocks, za.n), dim3(256), 0, st, za);
        ST_LAUNCH_CHECK();
    }
    if (io->dec_in0) {      // dec_in_0 = prenet(go frame) of a normalised prenet (a plain one gives the zeros the tape holds already)
        st_t16_view xq0 = {io->xq_tape, sv.q_kbs, 0};
        int rc0 = st_tile_rows(io->dec_in0, P, &xq0, B, P, stream);
        if (rc0) return rc0;
    }
    if (pure_tf && steps > 1) {
        const size_t total = (size_t)(steps - 1) * B * P;
        size_t blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(tile_teacher_kernel, dim3((unsigned)blocks), dim3(256), 0, st, io->teacher_pre, io->Tt, io->xq_tape,
                           sv.q_floats, sv.q_kbs, steps, B, P);
        ST_LAUNCH_CHECK();
    }
    // teacher-forced training (deferred projection): [decoder cell of t | query cell of t+1] as one launch
    const bool pair_cells = defer && pure_tf && io->pair_cells;
    // io->gate_part: the decoder cell's gate products over [AdaIN(h_q(t)) | h_d(t-1)] -- known before the attention of step t runs -- ride
    // beside the pq / fin launch; the cell launch reduces the context columns and adds the slab.  Measured (round 6, two boxes, A/B in
    // one process): 2.71 -> 2.79 and 2.78 -> 2.84 M mel-frames/s.  The same for the query cell, or the two halves at two sites, LOSE
    // (DESIGN.md section 3.5): a hosted product costs ~7 us + 3 us per 1024 columns whatever its host's length.
    // (the other forms of the attention step issue the partial product as a launch of its own: same arithmetic bit for bit, so a starved
    // hand-off that degrades to the two-launch form changes nothing but the speed)
    // Teacher-forced training (deferred projection) has the same split: there the product rides beside the query projection + attention pre
    // part of the step (pre_in_pq; its operands come from the paired cell launch of the step before), the paired launch adds the slab.
    const bool split_d = io->gate_part && (!defer || pre_in_pq) && B > 16 && B <= 32 && D % 16 == 0 && (D / 4) % 2 == 0 &&
                         E % 16 == 0 && Q % 16 == 0 && st_aligned16(io->gate_part);
    const bool split_hosted = split_d && fuse_pq_fin && (A / 16) * ((B + 15) / 16) + B * fin_parts + (4 * D) / 32 <= st_device_cus();
    // (two attention-pre workgroups per utterance then: with four, 16 + 128 + 128 workgroups would not fit the device in one round)
    const int pre_parts_split = io->attn_pre_parts > 2 ? 2 : io->attn_pre_parts;
    const bool split_pre = split_d && pre_in_pq && A <= 16 * 128 &&
                           ((A + 15) / 16) * ((B + 15) / 16) + B * (pre_parts_split > 1 ? pre_parts_split : 1) + (4 * D) / 32 <= st_device_cus();
    st_partial_product_job pj_d;
    memset(&pj_d, 0, sizeof(pj_d));
    // ... how much of it: measured at C2 (hosted 512 / 768 / 1024 / 1280 / 1536 / 2048 columns: 2.78 / 2.83 / 2.88 / 2.89 / 2.86 / 2.81 M
    // mel-frames/s, whole rounds of 16 k-blocks for the 8 waves x 2 only -- 1088 ... 1408 in steps of 64 all lose to 1280): the cell keeps half.
    ST_CHECK_ARG(io->gate_part_k == 0 || !io->gate_part || (io->gate_part_k % 16 == 0 && io->gate_part_k >= 16 * sv.d_ha && io->gate_part_k < 16 * sv.d_kbs),
                 "st_decoder_forward: gate_part_k = %d must be a multiple of 16 in [%d, %d)", io->gate_part_k, 16 * sv.d_ha, 16 * sv.d_kbs);
    const int d_k0 = (io->gate_part_k ? io->gate_part_k : st_decoder_gate_split_k(d)) / 16;
    pj_d.packed_w = io->packed + pl.d; pj_d.w_kbs = sv.d_kbs; pj_d.kb0 = d_k0; pj_d.KB = sv.d_kbs - d_k0; pj_d.N = 4 * D; pj_d.part = io->gate_part;
    for (int t = 0; t < steps; ++t) {
        float* xq = io->xq_tape + (size_t)t * sv.q_floats;
        float* xq_next = io->xq_tape + (size_t)(t + 1) * sv.q_floats;
        float* xd = io->xd_tape + (size_t)t * sv.d_floats;
        float* xd_next = io->xd_tape + (size_t)(t + 1) * sv.d_floats;
        float* xo = io->xo_tape + (size_t)t * sv.o_floats;
        bool prod_done = false;          // the hosted part of the decoder cell's gate product of this step has been issued

        // 1. query LSTM (+ AdaIN of the new hidden state)                ref: :227-231, :267-269
        //    h_q_t -> xq_{t+1}[h part] (next step's recurrent input, also the query projection's input)
        //    adapted h_q_t -> xd_t[hadapt part]
        st_t16_view xq_v = {xq, sv.q_kbs, 0};
        st_t16_view hq_dst = {xq_next, sv.q_kbs, sv.q_h};
        st_t16_view ha_dst = {xd, sv.d_kbs, sv.d_ha};
        // (pair_cells: the query cell of step t > 0 ran beside the decoder cell of step t-1, see step 4)
        if (!ST_SKIPPED(0) && !(pair_cells && t > 0))
                            rc = st_lstm_cell_packed_fwd(io->packed + pl.q, &xq_v, Kq, w->q_b_ih, w->q_b_hh,
                                     io->cq_tape + (size_t)t * BQ, Q, io->q_mask ? io->q_mask + (size_t)t * BQ : nullptr,
                                     &hq_dst, nullptr, io->cq_tape + (size_t)(t + 1) * BQ, Q,
                                     io->gates_q_tape ? io->gates_q_tape + (size_t)t * 4 * BQ : nullptr,
                                     io->ada_std, io->ada_mean, &ha_dst, B, Q, stream);
        if (rc) return rc;

        // 2. processed query                                             ref: :380
        if (fuse_pq_fin || fin_rng) rc = 0;    // (rides with step 3)
        else if (pre_in_pq && t > 0) {   // attention pre part of THIS step rides along (needs only the weights of step t-1)
            st_attn_pre_job job = {io->pm, io->align_out + (size_t)(t - 1) * L, ldal, io->wcum_tape + (size_t)t * BL,
                                   w->attn_loc_conv_w, w->attn_loc_lin_w, io->attn_s_buf + (size_t)t * io->attn_s_step_floats, L, A, d->F,
                                   d->K, split_pre ? pre_parts_split : io->attn_pre_parts,
                                   io->attn_loc_tape ? io->attn_loc_tape + (size_t)t * BL * d->F : nullptr};
            if (split_pre) {             // ... and the tail of the decoder cell's gate product of this step
                pj_d.x = st_t16_view{xd, sv.d_kbs, d_k0};
                job.part = &pj_d;
                prod_done = true;
            }
            rc = st_skinny_linear_packed_attnpre_fwd(io->packed + pl.pq, &hq_dst, 16 * kb16(Q), nullptr, ST_ACT_NONE, nullptr, 0,
                                                     io->pq_buf, A, nullptr, 0, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, B, A,
                                                     &job, stream);
        } else if (!ST_SKIPPED(1))
            rc = st_skinny_linear_packed_fwd(io->packed + pl.pq, &hq_dst, 16 * kb16(Q), nullptr, ST_ACT_NONE, nullptr, 0,
                                             io->pq_buf, A, nullptr, 0, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, B, A, stream);
        if (rc) return rc;

        // 3. attention + state update                                    ref: :256-264, :371-407
        //    ctx_t -> xq_{t+1}[ctx part], xd_t[ctx part], xo_t[ctx part]
        const float* w_prev = t == 0 ? io->zero_row : io->align_out + (size_t)(t - 1) * L;
        st_t16_view ctx_dst[3] = {{xq_next, sv.q_kbs, sv.q_ctx}, {xd, sv.d_kbs, 0}, {xo, sv.o_kbs, sv.o_ctx}};
        if (fin_rng && !(ST_SKIPPED(1) || ST_SKIPPED(2))) {   // 2 + 3 + combine as one launch (long texts)
            st_attn_fin_job fj;
            memset(&fj, 0, sizeof(fj));
            fj.s_buf = t == 0 ? io->pm : io->attn_s_buf + (size_t)t * io->attn_s_step_floats;
            fj.memory = io->memory; fj.w_cum_prev = io->wcum_tape + (size_t)t * BL;
            fj.w_out = io->align_out + (size_t)t * L; fj.ld_wout = ldal; fj.w_cum_out = io->wcum_tape + (size_t)(t + 1) * BL; fj.v = w->attn_v;
            for (int c = 0; c < 3; ++c) fj.ctx_dst[c] = ctx_dst[c];
            fj.status = io->handoff_status;
            fj.n_ctx_dst = 3; fj.parts = sp_parts; fj.L = L; fj.A = A; fj.E = E; fj.F = d->F; fj.K = d->K;
            rc = st_query_attn_rng_fwd(io->packed + pl.pq, &hq_dst, 16 * kb16(Q), io->pq_granules, io->attn_xchg, (unsigned)(t + 1), &fj, B, stream);
        }
        else if (fin_rng) rc = 0;
        else if (fin_split && !ST_SKIPPED(2))
            rc = st_attn_fin_split_fwd(io->pq_buf, t == 0 ? io->pm : io->attn_s_buf + (size_t)t * io->attn_s_step_floats, io->memory,
                                       io->wcum_tape + (size_t)t * BL, io->align_out + (size_t)t * L, ldal,
                                       io->wcum_tape + (size_t)(t + 1) * BL, w->attn_v, ctx_dst, 3, nullptr, 0, io->attn_split_ws, sp_parts,
                                       B, L, A, E, stream);
        else if (fin_split) rc = 0;
        else if (fuse_pq_fin && (ST_SKIPPED(1) || ST_SKIPPED(2))) rc = 0;
        else if (fuse_pq_fin) {   // 2 + 3 as one launch: the fin workgroups wait for pq inside the launch (granule hand-off)
            st_attn_fin_job fj;
            memset(&fj, 0, sizeof(fj));
            fj.s_buf = t == 0 ? io->pm : io->attn_s_buf + (size_t)t * io->attn_s_step_floats;
            fj.memory = io->memory; fj.w_cum_prev = io->wcum_tape + (size_t)t * BL;
            fj.w_out = io->align_out + (size_t)t * L; fj.ld_wout = ldal; fj.w_cum_out = io->wcum_tape + (size_t)(t + 1) * BL; fj.v = w->attn_v;
            for (int c = 0; c < 3; ++c) fj.ctx_dst[c] = ctx_dst[c];
            fj.status = io->handoff_status;
            fj.n_ctx_dst = 3; fj.parts = fin_parts; fj.L = L; fj.A = A; fj.E = E; fj.F = d->F; fj.K = d->K;
            if (split_hosted && !ST_SKIPPED(3)) {
                pj_d.x = st_t16_view{xd, sv.d_kbs, d_k0};
                prod_done = true;
                rc = st_query_attn_fin_part_fwd(io->packed + pl.pq, &hq_dst, 16 * kb16(Q), io->pq_granules, (unsigned)(t + 1), &fj, B, &pj_d, stream);
            } else
            rc = st_query_attn_fin_fwd(io->packed + pl.pq, &hq_dst, 16 * kb16(Q), io->pq_granules, (unsigned)(t + 1), &fj, B, stream);
        }
        else if (ST_SKIPPED(2)) rc = 0;
        else if (split_attn)     // S of this step was written by the pre part (step 0: no history yet, S = pm)
            rc = st_attn_fin_t16_fwd(io->pq_buf, t == 0 ? io->pm : io->attn_s_buf + (size_t)t * io->attn_s_step_floats, io->memory,
                                     io->wcum_tape + (size_t)t * BL,
                                     io->align_out + (size_t)t * L, ldal, io->wcum_tape + (size_t)(t + 1) * BL, w->attn_v,
                                     ctx_dst, 3, nullptr, 0, fin_parts, B, L, A, E, d->F, d->K, stream);
        else
            rc = st_attn_step_t16_fwd(io->pq_buf, io->pm, io->memory, w_prev, t == 0 ? L : ldal,
                                      io->wcum_tape + (size_t)t * BL, io->align_out + (size_t)t * L, ldal,
                                      io->wcum_tape + (size_t)(t + 1) * BL,
                                      w->attn_loc_conv_w, w->attn_loc_lin_w, w->attn_v, ctx_dst, 3, nullptr, 0,
                                      B, L, A, E, d->F, d->K, stream);
        if (rc) return rc;

        // 4. decoder LSTM: h_d_t -> xd_{t+1}[h part], xo_t[h part]        ref: :275-280
        st_t16_view xd_v = {xd, sv.d_kbs, 0};
        st_t16_view hd_dst0 = {xd_next, sv.d_kbs, sv.d_h};
        st_t16_view hd_dst1 = {xo, sv.o_kbs, 0};
        if (split_d && !prod_done && !ST_SKIPPED(3) && !(split_hosted && (ST_SKIPPED(1) || ST_SKIPPED(2)))) {       // (no launch of this step could host the product: one of its own, same arithmetic)
            pj_d.x = st_t16_view{xd, sv.d_kbs, d_k0};
            rc = st_partial_product_fwd(&pj_d, B, stream);
            if (rc) return rc;
        }
        if (pair_cells && t + 1 < steps) {
            // teacher forcing: the query cell of step t+1 needs ctx_t, h_q_t and a teacher frame -- not h_d_t.  Both cells in one launch.
            st_lstm_cell_packed_job jd, jq;
            memset(&jd, 0, sizeof(jd)); memset(&jq, 0, sizeof(jq));
            jd.packed_w = io->packed + pl.d; jd.x = xd_v; jd.K = Kd; jd.b_ih = w->d_b_ih; jd.b_hh = w->d_b_hh;
            if (split_d) { jd.K = 16 * d_k0; jd.part = io->gate_part; jd.w_kbs = sv.d_kbs; }
            jd.c_prev = io->cd_tape + (size_t)t * BD; jd.ldc_prev = D; jd.mask = io->d_mask ? io->d_mask + (size_t)t * BD : nullptr;
            jd.h_dst0 = hd_dst0; jd.h_dst1 = hd_dst1; jd.c_out = io->cd_tape + (size_t)(t + 1) * BD; jd.ldc = D;
            jd.gates_out = io->gates_d_tape ? io->gates_d_tape + (size_t)t * 4 * BD : nullptr;
            jd.B = B; jd.H = D;
            float* xq_next2 = io->xq_tape + (size_t)(t + 2) * sv.q_floats;
            jq.packed_w = io->packed + pl.q; jq.x = st_t16_view{xq_next, sv.q_kbs, 0}; jq.K = Kq; jq.b_ih = w->q_b_ih; jq.b_hh = w->q_b_hh;
            jq.c_prev = io->cq_tape + (size_t)(t + 1) * BQ; jq.ldc_prev = Q; jq.mask = io->q_mask ? io->q_mask + (size_t)(t + 1) * BQ : nullptr;
            jq.h_dst0 = st_t16_view{xq_next2, sv.q_kbs, sv.q_h}; jq.c_out = io->cq_tape + (size_t)(t + 2) * BQ; jq.ldc = Q;
            jq.gates_out = io->gates_q_tape ? io->gates_q_tape + (size_t)(t + 1) * 4 * BQ : nullptr;
            jq.ada_std = io->ada_std; jq.ada_mean = io->ada_mean; jq.hadapt_dst = st_t16_view{xd_next, sv.d_kbs, sv.d_ha};
            jq.B = B; jq.H = Q;
            rc = st_lstm_cell_packed_pair_fwd(&jd, &jq, stream);
        } else {
        if (split_d && !ST_SKIPPED(3) && !(split_hosted && (ST_SKIPPED(1) || ST_SKIPPED(2))))
            rc = st_lstm_cell_packed_part_fwd(io->packed + pl.d, sv.d_kbs, &xd_v, 16 * d_k0, io->gate_part, w->d_b_ih, w->d_b_hh,
                                              io->cd_tape + (size_t)t * BD, D, io->d_mask ? io->d_mask + (size_t)t * BD : nullptr,
                                              &hd_dst0, &hd_dst1, io->cd_tape + (size_t)(t + 1) * BD, D,
                                              io->gates_d_tape ? io->gates_d_tape + (size_t)t * 4 * BD : nullptr, B, D, stream);
        else
        if (!ST_SKIPPED(3)) rc = st_lstm_cell_packed_fwd(io->packed + pl.d, &xd_v, Kd, w->d_b_ih, w->d_b_hh,
                                     io->cd_tape + (size_t)t * BD, D, io->d_mask ? io->d_mask + (size_t)t * BD : nullptr,
                                     &hd_dst0, &hd_dst1, io->cd_tape + (size_t)(t + 1) * BD, D,
                                     io->gates_d_tape ? io->gates_d_tape + (size_t)t * 4 * BD : nullptr,
                                     nullptr, nullptr, nullptr, B, D, stream);
        }
        if (rc) return rc;

        // 5. mel frames + stop logit (+ prenet layer 1 of the next input when fused)   ref: :282-287
        st_t16_view xo_v = {xo, sv.o_kbs, 0};
        st_t16_view mel_dst = {io->mel_t16, kb16(in_dim), 0};
        st_t16_view pre1_dst = {io->pre1_t16 + (size_t)t * io->pre1_step_floats, kb16(P), 0};
        const bool fuse = d->fuse_pre0 != 0;
        if (defer) continue;         // mel / stop of all steps come from one GEMM over the xo tape (caller)
        st_attn_pre_job job = {io->pm, io->align_out + (size_t)t * L, ldal, io->wcum_tape + (size_t)(t + 1) * BL,
                               w->attn_loc_conv_w, w->attn_loc_lin_w,
                               split_attn ? io->attn_s_buf + (size_t)(t + 1) * io->attn_s_step_floats : nullptr, L, A, d->F, d->K,
                               io->attn_pre_parts, io->attn_loc_tape ? io->attn_loc_tape + (size_t)(t + 1) * BL * d->F : nullptr};
        const bool p2_now = fuse_p2 && t + 1 < steps && !ST_SKIPPED(6) && !ST_SKIPPED(5);
        if (p2_now) {
            job.p2_packed_w = io->packed + pl.p1; job.p2_K = P; job.p2_N = P; job.p2_act = ST_ACT_RELU;
            job.p2_mask = io->prenet_mask ? io->prenet_mask + ((size_t)t * 2 + 1) * B * P : nullptr; job.p2_ldmask = P;
            job.p2_dst = st_t16_view{xq_next, sv.q_kbs, 0};
            job.p2_gran = io->pre1_granules; job.p2_epoch = (unsigned)(t + 1); job.p2_status = io->handoff_status;
        }
        if (!ST_SKIPPED(4)) rc = st_skinny_linear_packed_attnpre_fwd(io->packed + pl.pg, &xo_v, ST_SKIPPED(7) ? Ko / 4 : Ko, w->projgate_b, ST_ACT_NONE, nullptr, 0,
                                                 io->mel_out + (size_t)t * in_dim, (int)ldmel, fuse ? nullptr : &mel_dst, in_dim,
                                                 io->stop_out + (size_t)t * d->r, steps * d->r, d->r,
                                                 fuse ? in_dim + 1 : 0, ST_ACT_RELU,
                                                 io->prenet_mask ? io->prenet_mask + (size_t)t * 2 * B * P : nullptr, P,
                                                 fuse ? &pre1_dst : nullptr, B, in_dim + 1 + (fuse ? P : 0),
                                                 split_attn && t + 1 < steps && !ST_SKIPPED(6) ? &job : nullptr, stream);
        if (rc) return rc;

        // 6. next decoder input -> xq_{t+1}[dec_in part]                 ref: :190-206
        if (t + 1 < steps) {
            const int src = io->step_src[t];
            st_t16_view next = {xq_next, sv.q_kbs, 0};
            if ((src == -1 || io->Bt < B) && !ST_SKIPPED(5) && !p2_now) {   // rows without a teacher feed their own output back
                rc = prenet_own(w, d, io, pl, sv, t, fuse, stream);             // (p2_now: layer 2 rode in the launch above)
                if (rc) return rc;
            }
            if (pure_tf) rc = 0;             // tiled for all steps before the loop
            else if (src >= 0) rc = st_tile_rows(io->teacher_pre + (size_t)src * P, io->Tt * P, &next, io->Bt, P, stream);
            else if (src == -2) rc = st_tile_rows(io->teacher_mean, P, &next, io->Bt, P, stream);
            if (rc) return rc;
        } else if (d->prenet_norm == 3 && (io->step_src[t] == -1 || io->Bt < B)) {
            // the reference forms the next input after the LAST step too (:192,:197-198,:205-206).  Nothing reads it, but a training-mode
            // BatchNorm1d moves its running statistics (and counts the batch) once more: same launches into slot `steps` of the tape
            rc = prenet_own(w, d, io, pl, sv, t, false, stream);
            if (rc) return rc;
        }
    }
    return 0;
}
