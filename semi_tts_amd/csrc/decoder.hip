// decoder.hip -- the autoregressive decode loop of Decoder.forward, issued natively.
//
// ref: src/module.py:184-206 (loop + next-input policy) and :216-288 (one step).
// Once per forward the six weight matrices the loop streams every step are packed into MFMA
// lane order (skinny_packed.hip, "P16"); all per-step activations live in the tiled "T16"
// layout, written that way by their producers.  Per step the loop enqueues, in dependency order:
//   1. query LSTM cell      x = [dec_in | ctx_{t-1} | h_q], fused AdaIN of the new h_q
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
    { int k[2] = {d->D, d->E}; o.pg = p; p += st_packed_weight_floats(k, 2, in_dim + 1, 0); }
    { int k[1] = {in_dim}; o.p0 = p; p += st_packed_weight_floats(k, 1, d->P, 0); }
    { int k[1] = {d->P}; o.p1 = p; p += st_packed_weight_floats(k, 1, d->P, 0); }
    o.total = p;
    return o;
}

int prenet_own(const st_decoder_weights* w, const st_decoder_dims* d, const st_decoder_io* io, const PackedLayout& pl,
               int t, void* stream) {
    // dec_in_{t+1} = prenet(mel_t) for every row                ref: src/module.py:192,:197-198,:205-206
    const int in_dim = d->r * d->n_mels;
    const size_t BP = (size_t)d->B * d->P;
    const size_t tP = st_t16_floats(d->B, d->P);
    const float* m1 = io->prenet_mask ? io->prenet_mask + ((size_t)t * 2 + 0) * BP : nullptr;
    const float* m2 = io->prenet_mask ? io->prenet_mask + ((size_t)t * 2 + 1) * BP : nullptr;
    st_pseg s1 = {io->mel_t16, in_dim};
    int rc = st_skinny_linear_packed_fwd(io->packed + pl.p0, &s1, 1, nullptr, ST_ACT_RELU, m1, d->P,
                                         nullptr, 0, io->pre1_t16, 0, nullptr, 0, 0, d->B, d->P, stream);
    if (rc) return rc;
    st_pseg s2 = {io->pre1_t16, d->P};
    return st_skinny_linear_packed_fwd(io->packed + pl.p1, &s2, 1, nullptr, ST_ACT_RELU, m2, d->P,
                                       nullptr, 0, io->decin_tape + (size_t)(t + 1) * tP, 0, nullptr, 0, 0,
                                       d->B, d->P, stream);
}

}  // namespace

extern "C" size_t st_decoder_packed_floats(const st_decoder_dims* d) { return d ? packed_layout(d).total : 0; }

extern "C" int st_decoder_pack(const st_decoder_weights* w, const st_decoder_dims* d, float* packed, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(w && d && packed, "st_decoder_pack: null pointer");
    const PackedLayout pl = packed_layout(d);
    const int in_dim = d->r * d->n_mels;
    int rc;
    {   // query LSTM: K = [dec_in (P) | ctx (E) | h_q (Q)]                  ref: src/module.py:227-228
        const float* ws[3] = {w->q_w_ih, w->q_w_ih + d->P, w->q_w_hh};
        int ld[3] = {d->P + d->E, d->P + d->E, d->Q}, k[3] = {d->P, d->E, d->Q};
        if ((rc = st_pack_weight(ws, ld, k, 3, 4 * d->Q, d->Q, packed + pl.q, stream))) return rc;
    }
    {   const float* ws[1] = {w->attn_query_w}; int ld[1] = {d->Q}, k[1] = {d->Q};
        if ((rc = st_pack_weight(ws, ld, k, 1, d->A, 0, packed + pl.pq, stream))) return rc; }
    {   // decoder LSTM: K = [ctx (E) | adapted h_q (Q) | h_d (D)]            ref: src/module.py:275-277
        const float* ws[3] = {w->d_w_ih, w->d_w_ih + d->E, w->d_w_hh};
        int ld[3] = {d->E + d->Q, d->E + d->Q, d->D}, k[3] = {d->E, d->Q, d->D};
        if ((rc = st_pack_weight(ws, ld, k, 3, 4 * d->D, d->D, packed + pl.d, stream))) return rc;
    }
    {   // proj (+) gate: K = [h_d (D) | ctx (E)]                             ref: src/module.py:282-287
        const float* ws[2] = {w->projgate_w, w->projgate_w + d->D};
        int ld[2] = {d->D + d->E, d->D + d->E}, k[2] = {d->D, d->E};
        if ((rc = st_pack_weight(ws, ld, k, 2, in_dim + 1, 0, packed + pl.pg, stream))) return rc;
    }
    {   const float* ws[1] = {w->prenet_w0}; int ld[1] = {in_dim}, k[1] = {in_dim};
        if ((rc = st_pack_weight(ws, ld, k, 1, d->P, 0, packed + pl.p0, stream))) return rc; }
    {   const float* ws[1] = {w->prenet_w1}; int ld[1] = {d->P}, k[1] = {d->P};
        if ((rc = st_pack_weight(ws, ld, k, 1, d->P, 0, packed + pl.p1, stream))) return rc; }
    return 0;
}

extern "C" int st_decoder_forward(const st_decoder_weights* w, const st_decoder_dims* d, const st_decoder_io* io,
                                  void* stream) {
    (void)hipGetLastError();  // drop stale errors left by other HIP users of this thread
    ST_CHECK_ARG(w && d && io, "st_decoder_forward: null struct pointer");
    const int B = d->B, L = d->L, E = d->E, P = d->P, Q = d->Q, D = d->D, A = d->A;
    const int steps = io->steps;
    const int in_dim = d->r * d->n_mels;
    ST_CHECK_ARG(B > 0 && L > 0 && steps > 0 && in_dim > 0, "st_decoder_forward: B=%d L=%d steps=%d", B, L, steps);
    ST_CHECK_ARG(io->memory && io->pm && io->ada_std && io->ada_mean && io->step_src && io->packed,
                 "st_decoder_forward: null input");
    ST_CHECK_ARG(io->mel_out && io->align_out && io->stop_out, "st_decoder_forward: null output");
    ST_CHECK_ARG(io->hq_tape && io->cq_tape && io->hd_tape && io->cd_tape && io->ctx_tape && io->wcum_tape &&
                 io->hadapt_tape && io->decin_tape && io->pq_buf && io->pre1_t16 && io->mel_t16 && io->zero_row,
                 "st_decoder_forward: null tape/scratch");
    for (int t = 0; t + 1 < steps; ++t) {
        const int src = io->step_src[t];
        ST_CHECK_ARG(src >= -2 && (src < 0 || (io->teacher_pre && src < io->Tt)), "st_decoder_forward: step_src[%d]=%d invalid", t, src);
        ST_CHECK_ARG(src != -2 || io->teacher_mean, "st_decoder_forward: step_src[%d]=-2 without teacher_mean", t);
        ST_CHECK_ARG(src == -1 || (io->Bt > 0 && io->Bt <= B), "st_decoder_forward: Bt=%d invalid", io->Bt);
    }
    hipStream_t st = (hipStream_t)stream;
    const PackedLayout pl = packed_layout(d);
    const size_t BQ = (size_t)B * Q, BD = (size_t)B * D, BL = (size_t)B * L;
    const size_t tQ = st_t16_floats(B, Q), tD = st_t16_floats(B, D), tE = st_t16_floats(B, E), tP = st_t16_floats(B, P);
    // Tiled tapes must be zero wherever a pad lane/row lives: the CALLER hands them in zero-filled
    // (torch.zeros); slot 0 (= initial zero state, src/module.py:290-303) is therefore already set.
    ST_HIP(hipMemsetAsync(io->cq_tape, 0, BQ * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->cd_tape, 0, BD * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->wcum_tape, 0, BL * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->zero_row, 0, BL * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->hq_tape, 0, tQ * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->hd_tape, 0, tD * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->ctx_tape, 0, tE * sizeof(float), st));
    // dec_in of step 0 = prenet(go frame of zeros) = relu(0) * mask = 0     ref: :161,:183
    ST_HIP(hipMemsetAsync(io->decin_tape, 0, tP * sizeof(float), st));

    const size_t ldmel = (size_t)steps * in_dim;
    const int ldal = steps * L;
    int rc;
    for (int t = 0; t < steps; ++t) {
        const float* decin = io->decin_tape + (size_t)t * tP;
        const float* ctx_prev = io->ctx_tape + (size_t)t * tE;
        float* ctx_new = io->ctx_tape + (size_t)(t + 1) * tE;
        const float* hq_prev = io->hq_tape + (size_t)t * tQ;
        float* hq_new = io->hq_tape + (size_t)(t + 1) * tQ;
        const float* hd_prev = io->hd_tape + (size_t)t * tD;
        float* hd_new = io->hd_tape + (size_t)(t + 1) * tD;
        float* hadapt = io->hadapt_tape + (size_t)t * tQ;

        // 1. query LSTM (+ AdaIN of the new hidden state)                ref: :227-231, :267-269
        st_pseg sq[3] = {{decin, P}, {ctx_prev, E}, {hq_prev, Q}};
        rc = st_lstm_cell_packed_fwd(io->packed + pl.q, sq, 3, w->q_b_ih, w->q_b_hh, nullptr, 0,
                                     io->cq_tape + (size_t)t * BQ, Q, io->q_mask ? io->q_mask + (size_t)t * BQ : nullptr,
                                     hq_new, io->cq_tape + (size_t)(t + 1) * BQ, Q,
                                     io->gates_q_tape ? io->gates_q_tape + (size_t)t * 4 * BQ : nullptr,
                                     io->ada_std, io->ada_mean, hadapt, B, Q, stream);
        if (rc) return rc;

        // 2. processed query                                             ref: :380
        st_pseg sp = {hq_new, Q};
        rc = st_skinny_linear_packed_fwd(io->packed + pl.pq, &sp, 1, nullptr, ST_ACT_NONE, nullptr, 0,
                                         io->pq_buf, A, nullptr, 0, nullptr, 0, 0, B, A, stream);
        if (rc) return rc;

        // 3. attention + state update                                    ref: :256-264, :371-407
        const float* w_prev = t == 0 ? io->zero_row : io->align_out + (size_t)(t - 1) * L;
        rc = st_attn_step_t16_fwd(io->pq_buf, io->pm, io->memory, w_prev, t == 0 ? L : ldal,
                                  io->wcum_tape + (size_t)t * BL, io->align_out + (size_t)t * L, ldal,
                                  io->wcum_tape + (size_t)(t + 1) * BL,
                                  w->attn_loc_conv_w, w->attn_loc_lin_w, w->attn_v, ctx_new, nullptr, 0,
                                  B, L, A, E, d->F, d->K, stream);
        if (rc) return rc;

        // 4. decoder LSTM                                                ref: :275-280
        st_pseg sd[3] = {{ctx_new, E}, {hadapt, Q}, {hd_prev, D}};
        rc = st_lstm_cell_packed_fwd(io->packed + pl.d, sd, 3, w->d_b_ih, w->d_b_hh, nullptr, 0,
                                     io->cd_tape + (size_t)t * BD, D, io->d_mask ? io->d_mask + (size_t)t * BD : nullptr,
                                     hd_new, io->cd_tape + (size_t)(t + 1) * BD, D,
                                     io->gates_d_tape ? io->gates_d_tape + (size_t)t * 4 * BD : nullptr,
                                     nullptr, nullptr, nullptr, B, D, stream);
        if (rc) return rc;

        // 5. mel frames + stop logit                                     ref: :282-287
        st_pseg so[2] = {{hd_new, D}, {ctx_new, E}};
        rc = st_skinny_linear_packed_fwd(io->packed + pl.pg, so, 2, w->projgate_b, ST_ACT_NONE, nullptr, 0,
                                         io->mel_out + (size_t)t * in_dim, (int)ldmel, io->mel_t16, in_dim,
                                         io->stop_out + (size_t)t * d->r, steps * d->r, d->r, B, in_dim + 1, stream);
        if (rc) return rc;

        // 6. next decoder input                                          ref: :190-206
        if (t + 1 < steps) {
            const int src = io->step_src[t];
            float* next = io->decin_tape + (size_t)(t + 1) * tP;
            if (src == -1 || io->Bt < B) {   // rows without a teacher feed their own output back
                rc = prenet_own(w, d, io, pl, t, stream);
                if (rc) return rc;
            }
            if (src >= 0) rc = st_tile_rows(io->teacher_pre + (size_t)src * P, io->Tt * P, next, io->Bt, P, stream);
            else if (src == -2) rc = st_tile_rows(io->teacher_mean, P, next, io->Bt, P, stream);
            if (rc) return rc;
        }
    }
    return 0;
}
