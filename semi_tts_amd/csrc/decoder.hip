// decoder.hip -- the autoregressive decode loop of Decoder.forward, issued natively.
//
// ref: src/module.py:184-206 (loop + next-input policy) and :216-288 (one step).
// Per step the loop enqueues, on ONE stream, in dependency order:
//   1. query LSTM cell      x = [dec_in | ctx_{t-1}], h = h_q                    (skinny.hip)
//   2. query projection     pq = W_q h_q                                         (skinny.hip)
//   3. attention step       energies, softmax, context, cumulative weights, AdaIN (attention.hip)
//   4. decoder LSTM cell    x = [ctx_t | adapted h_q], h = h_d                   (skinny.hip)
//   5. proj + gate          [h_d | ctx_t] -> r mel frames + stop logit           (skinny.hip)
//   6. next input           teacher frame copy, and/or prenet(own output) (2 launches)
// There is no host synchronisation, allocation or blocking copy inside, so the whole loop
// can be captured into a hipGraph (st_graph_begin/st_graph_end) and replayed as one launch.
// State lives in caller-owned tapes indexed by step; nothing is overwritten, so the same
// buffers are the saved tensors of the backward pass.
#include "st_common.h"

namespace {

int prenet_own(const st_decoder_weights* w, const st_decoder_dims* d, const st_decoder_io* io, int t,
               int row0, int rows, void* stream) {
    // dec_in_{t+1}[row0:row0+rows] = prenet(mel_t[row0:...])   ref: src/module.py:192,:197-198,:205-206
    const int in_dim = d->r * d->n_mels;
    const size_t ldmel = (size_t)io->steps * in_dim;
    const size_t BP = (size_t)d->B * d->P;
    st_seg s1;
    s1.x = io->mel_out + (size_t)row0 * ldmel + (size_t)t * in_dim; s1.ldx = (int)ldmel;
    s1.w = w->prenet_w0; s1.ldw = in_dim; s1.k = in_dim;
    const float* m1 = io->prenet_mask ? io->prenet_mask + ((size_t)t * 2 + 0) * BP + (size_t)row0 * d->P : nullptr;
    const float* m2 = io->prenet_mask ? io->prenet_mask + ((size_t)t * 2 + 1) * BP + (size_t)row0 * d->P : nullptr;
    float* h1 = io->pre1_buf + (size_t)row0 * d->P;
    int rc = st_skinny_linear_fwd(&s1, 1, nullptr, ST_ACT_RELU, m1, d->P, h1, d->P, 0, nullptr, 0, 0, rows, d->P, stream);
    if (rc) return rc;
    st_seg s2;
    s2.x = h1; s2.ldx = d->P; s2.w = w->prenet_w1; s2.ldw = d->P; s2.k = d->P;
    float* out = io->decin_tape + (size_t)(t + 1) * BP + (size_t)row0 * d->P;
    return st_skinny_linear_fwd(&s2, 1, nullptr, ST_ACT_RELU, m2, d->P, out, d->P, 0, nullptr, 0, 0, rows, d->P, stream);
}

}  // namespace

extern "C" int st_decoder_forward(const st_decoder_weights* w, const st_decoder_dims* d, const st_decoder_io* io,
                                  void* stream) {
    ST_CHECK_ARG(w && d && io, "st_decoder_forward: null struct pointer");
    const int B = d->B, L = d->L, E = d->E, P = d->P, Q = d->Q, D = d->D, A = d->A;
    const int steps = io->steps;
    const int in_dim = d->r * d->n_mels;
    ST_CHECK_ARG(B > 0 && L > 0 && steps > 0 && in_dim > 0, "st_decoder_forward: B=%d L=%d steps=%d", B, L, steps);
    ST_CHECK_ARG(io->memory && io->pm && io->ada_std && io->ada_mean && io->step_src, "st_decoder_forward: null input");
    ST_CHECK_ARG(io->mel_out && io->align_out && io->stop_out, "st_decoder_forward: null output");
    ST_CHECK_ARG(io->hq_tape && io->cq_tape && io->hd_tape && io->cd_tape && io->ctx_tape && io->wcum_tape &&
                 io->hadapt_tape && io->decin_tape && io->pq_buf && io->pre1_buf && io->zero_row,
                 "st_decoder_forward: null tape/scratch");
    for (int t = 0; t + 1 < steps; ++t) {
        const int src = io->step_src[t];
        ST_CHECK_ARG(src >= -2 && (src < 0 || (io->teacher_pre && src < io->Tt)), "st_decoder_forward: step_src[%d]=%d invalid", t, src);
        ST_CHECK_ARG(src != -2 || io->teacher_mean, "st_decoder_forward: step_src[%d]=-2 without teacher_mean", t);
        ST_CHECK_ARG(src == -1 || (io->Bt > 0 && io->Bt <= B), "st_decoder_forward: Bt=%d invalid", io->Bt);
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t BQ = (size_t)B * Q, BD = (size_t)B * D, BE = (size_t)B * E, BL = (size_t)B * L, BP = (size_t)B * P;
    // slot 0 of every tape = initial zero state                      ref: src/module.py:290-303
    ST_HIP(hipMemsetAsync(io->hq_tape, 0, BQ * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->cq_tape, 0, BQ * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->hd_tape, 0, BD * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->cd_tape, 0, BD * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->ctx_tape, 0, BE * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->wcum_tape, 0, BL * sizeof(float), st));
    ST_HIP(hipMemsetAsync(io->zero_row, 0, BL * sizeof(float), st));
    // dec_in of step 0 = prenet(go frame of zeros) = relu(0) * mask = 0     ref: :161,:183
    ST_HIP(hipMemsetAsync(io->decin_tape, 0, BP * sizeof(float), st));

    const size_t ldmel = (size_t)steps * in_dim;
    const int ldal = steps * L;
    int rc;
    for (int t = 0; t < steps; ++t) {
        const float* decin = io->decin_tape + (size_t)t * BP;
        const float* ctx_prev = io->ctx_tape + (size_t)t * BE;
        float* ctx_new = io->ctx_tape + (size_t)(t + 1) * BE;
        const float* hq_prev = io->hq_tape + (size_t)t * BQ;
        float* hq_new = io->hq_tape + (size_t)(t + 1) * BQ;
        const float* hd_prev = io->hd_tape + (size_t)t * BD;
        float* hd_new = io->hd_tape + (size_t)(t + 1) * BD;
        float* hadapt = io->hadapt_tape + (size_t)t * BQ;

        // 1. query LSTM                                                  ref: :227-231
        st_seg sq[3];
        sq[0].x = decin;    sq[0].ldx = P; sq[0].w = w->q_w_ih;     sq[0].ldw = P + E; sq[0].k = P;
        sq[1].x = ctx_prev; sq[1].ldx = E; sq[1].w = w->q_w_ih + P; sq[1].ldw = P + E; sq[1].k = E;
        sq[2].x = hq_prev;  sq[2].ldx = Q; sq[2].w = w->q_w_hh;     sq[2].ldw = Q;     sq[2].k = Q;
        rc = st_lstm_cell_fwd(sq, 3, w->q_b_ih, w->q_b_hh, nullptr, 0, io->cq_tape + (size_t)t * BQ, Q,
                              io->q_mask ? io->q_mask + (size_t)t * BQ : nullptr, hq_new, Q,
                              io->cq_tape + (size_t)(t + 1) * BQ, Q,
                              io->gates_q_tape ? io->gates_q_tape + (size_t)t * 4 * BQ : nullptr, B, Q, stream);
        if (rc) return rc;

        // 2. processed query                                             ref: :380
        st_seg sp;
        sp.x = hq_new; sp.ldx = Q; sp.w = w->attn_query_w; sp.ldw = Q; sp.k = Q;
        rc = st_skinny_linear_fwd(&sp, 1, nullptr, ST_ACT_NONE, nullptr, 0, io->pq_buf, A, 0, nullptr, 0, 0, B, A, stream);
        if (rc) return rc;

        // 3. attention + state update + AdaIN                            ref: :256-269, :371-407
        const float* w_prev = t == 0 ? io->zero_row : io->align_out + (size_t)(t - 1) * L;
        rc = st_attn_step_fwd(io->pq_buf, io->pm, io->memory, w_prev, t == 0 ? L : ldal,
                              io->wcum_tape + (size_t)t * BL, io->align_out + (size_t)t * L, ldal,
                              io->wcum_tape + (size_t)(t + 1) * BL,
                              w->attn_loc_conv_w, w->attn_loc_lin_w, w->attn_v, ctx_new, E,
                              hq_new, Q, io->ada_std, io->ada_mean, hadapt, Q,
                              B, L, A, E, d->F, d->K, stream);
        if (rc) return rc;

        // 4. decoder LSTM                                                ref: :275-280
        st_seg sd[3];
        sd[0].x = ctx_new; sd[0].ldx = E; sd[0].w = w->d_w_ih;     sd[0].ldw = E + Q; sd[0].k = E;
        sd[1].x = hadapt;  sd[1].ldx = Q; sd[1].w = w->d_w_ih + E; sd[1].ldw = E + Q; sd[1].k = Q;
        sd[2].x = hd_prev; sd[2].ldx = D; sd[2].w = w->d_w_hh;     sd[2].ldw = D;     sd[2].k = D;
        rc = st_lstm_cell_fwd(sd, 3, w->d_b_ih, w->d_b_hh, nullptr, 0, io->cd_tape + (size_t)t * BD, D,
                              io->d_mask ? io->d_mask + (size_t)t * BD : nullptr, hd_new, D,
                              io->cd_tape + (size_t)(t + 1) * BD, D,
                              io->gates_d_tape ? io->gates_d_tape + (size_t)t * 4 * BD : nullptr, B, D, stream);
        if (rc) return rc;

        // 5. mel frames + stop logit                                     ref: :282-287
        st_seg so[2];
        so[0].x = hd_new;  so[0].ldx = D; so[0].w = w->projgate_w;     so[0].ldw = D + E; so[0].k = D;
        so[1].x = ctx_new; so[1].ldx = E; so[1].w = w->projgate_w + D; so[1].ldw = D + E; so[1].k = E;
        rc = st_skinny_linear_fwd(so, 2, w->projgate_b, ST_ACT_NONE, nullptr, 0,
                                  io->mel_out + (size_t)t * in_dim, (int)ldmel, in_dim,
                                  io->stop_out + (size_t)t * d->r, steps * d->r, d->r, B, in_dim + 1, stream);
        if (rc) return rc;

        // 6. next decoder input                                          ref: :190-206
        if (t + 1 < steps) {
            const int src = io->step_src[t];
            float* next = io->decin_tape + (size_t)(t + 1) * BP;
            if (src == -1) {
                rc = prenet_own(w, d, io, t, 0, B, stream);
                if (rc) return rc;
            } else {
                if (src >= 0) rc = st_copy2d(next, P, io->teacher_pre + (size_t)src * P, io->Tt * P, io->Bt, P, stream);
                else rc = st_copy2d(next, P, io->teacher_mean, P, io->Bt, P, stream);
                if (rc) return rc;
                if (io->Bt < B) {   // unpaired-text rows have no teacher: feed their own output back
                    rc = prenet_own(w, d, io, t, io->Bt, B - io->Bt, stream);
                    if (rc) return rc;
                }
            }
        }
    }
    return 0;
}
