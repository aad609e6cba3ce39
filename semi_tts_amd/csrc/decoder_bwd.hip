// decoder_bwd.hip -- backward through time of the teacher-forced decode loop (training).
//
// ref: what torch autograd derives for Decoder.forward / decode_one_step, src/module.py:184-288.
// The caller (semi_tts_amd/autograd.py) has already
//   * un-tiled the forward tapes and recomputed pq_t = W_q h_q_t for all steps in one GEMM,
//   * pushed the output gradients through proj/gate for all steps in one GEMM (dxo), which is possible
//     because with teacher forcing no step's input depends on an earlier step's output.
// Per step, from the last to the first, this file enqueues (no host sync, graph-capturable):
//   a. decoder LSTM, pointwise part        dh_d = dxo_t[:, :D] + (W_hh_d^T dgates_d)_{t+1}
//   b. dxd_t = dgates_d_t [W_ih_d | W_hh_d]     -> dctx, d(adapted h_q), dh_d carried to t-1
//   c. attention backward (attention_bwd.hip)  -> dpq_t, d[w_{t-1}; cum_{t-1}], per-step tape slices for dpm, dmem, dv, dW_l, dW_c
//   d. dh_q += W_q^T dpq_t
//   e. query LSTM, pointwise part          dh_q = d + (W_hh_q^T dgates_q)_{t+1} + std * d(adapted h_q)
//   f. dxq_t = dgates_q_t [W_ih_q | W_hh_q]     -> d(dec_in_t), dctx_{t-1}, dh_q carried to t-1
// Weight gradients are the caller's TN GEMMs over the tapes written here (dgates_q/d, dpq) afterwards.
#include "st_common.h"
#include "loop_graph.h"

extern "C" int st_attn_step_bwd(const float* pq, const float* pm, const float* memory,
                                const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                                const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                const float* const* dctx, const int* ld_dctx, int n_dctx,
                                const float* const* dw_direct, const int* ld_dw, int n_dw,
                                float* dcum, const float* dcum_add, int ld_dcum_add,
                                float* dpq, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                                float* dctx_t, float* dv_t,
                                int B, int L, int A, int E, int F, int K, void* stream);
extern "C" int st_attn_step_bwd_s(const float* pq, const float* pm, const float* memory,
                                  const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                                  const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                  const float* const* dctx, const int* ld_dctx, int n_dctx,
                                  const float* const* dw_direct, const int* ld_dw, int n_dw,
                                  float* dcum, const float* dcum_add, int ld_dcum_add,
                                  float* dpq, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                                  float* dctx_t, float* dv_t, const float* s_in,
                                  int B, int L, int A, int E, int F, int K, void* stream);

extern "C" int st_attn_step_bwd_t16(const float* pq, const float* pm, const float* memory,
                                    const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                                    const float* loc_conv_w, const float* loc_lin_w, const float* v,
                                    const float* const* dctx, const int* ld_dctx, int n_dctx,
                                    const float* const* dw_direct, const int* ld_dw, int n_dw,
                                    float* dcum, const float* dcum_add, int ld_dcum_add,
                                    float* dpq, const st_t16_view* dpq_t16, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                                    float* dctx_t, float* dv_t, const float* s_in,
                                    int B, int L, int A, int E, int F, int K, void* stream);

extern "C" int st_attn_bwd_wide_fits(int L, int A, int E, int F, int K);

namespace {
// which forms the fused BPTT loop takes (see st_decoder_bwd_forms): parts of the hosted attention backward, the two partial products
struct BwdForms { int parts; bool partial_d, partial_q; int dsplits, qsplits; };
BwdForms bwd_forms(const st_decoder_dims* d, const st_decoder_bwd_io* io) {
    BwdForms f = {1, false, false, 2, 4};
    const int B = d->B, L = d->L, E = d->E, P = d->P, Q = d->Q, D = d->D, A = d->A;
    const int XQ = P + E + Q, XD = E + Q + D;
    const bool overlap = io->overlap_attn && io->attn_s_tape && io->fuse_pw;
    // the hosted attention backward as `parts` workgroups per utterance over slices of the attention dims; what needs the sum of their
    // partial location-feature gradients (dloc_t, hist_t, the conv-transpose to dhist, the carried dcum) rides in the step's dgates_q
    // launch (st_attn_hist_job).  A = 256, 32 filters, texts whose wide block fits the LDS -- else the whole step per workgroup
    f.parts = overlap && io->dloc_part && (io->attn_parts == 2 || io->attn_parts == 4) ? io->attn_parts : 1;
    if (f.parts > 1 && !(A % (16 * f.parts) == 0 && 512 % (A / f.parts) == 0 && 512 / (A / f.parts) >= 2 * f.parts && d->F == 32 && d->K <= 31 &&
                         st_attn_bwd_wide_fits(L, A, E, d->F, d->K))) f.parts = 1;
    // ... and, in that form, the decoder cell's product of the hosted launch K-split into partial slabs (two row tiles and both batch tiles
    // per workgroup: half the bytes through the compute units; 2 B + N / 32 * S workgroups: one round) that the step's W_q^T dpq launch sums
    f.dsplits = io->dxd_splits > 0 ? io->dxd_splits : 2;
    f.partial_d = f.parts == 2 && io->dxd_part && B > 16 && B <= 32 && XD % 32 == 0 && ((4 * D) / 16) % f.dsplits == 0 && io->Bp == B;
    // ... and the query cell's product likewise (N / 32 * S + B history workgroups); its consumers take the slabs as addends
    f.qsplits = io->dxq_splits > 0 ? io->dxq_splits : 4;
    f.partial_q = f.partial_d && io->dxq_part && f.qsplits <= 4 && XQ % 32 == 0 && ((4 * Q) / 16) % f.qsplits == 0;
    return f;
}
}  // namespace

// the dimensions the fused BPTT loop (io.fuse_pw: pointwise halves in the products' epilogues) takes; anything else runs the six-launch loop
static bool bwd_fuse_dims_ok(const st_decoder_dims* d) {
    const int E = d->E, P = d->P, Q = d->Q, D = d->D, A = d->A;
    const int XQ = P + E + Q, XD = E + Q + D, XO = D + E;
    return Q % 16 == 0 && D % 16 == 0 && (E + Q) % 16 == 0 && A % 4 == 0 && XD % 4 == 0 && XQ % 4 == 0 && XO % 4 == 0;
}

extern "C" int st_decoder_bwd_fuse_dims(const st_decoder_dims* d) { return d && bwd_fuse_dims_ok(d) ? 1 : 0; }

extern "C" int st_decoder_bwd_forms(const st_decoder_dims* d, const st_decoder_bwd_io* io) {
    if (!d || !io) return 0;
    const BwdForms f = bwd_forms(d, io);
    return (f.parts > 1 ? 1 : 0) | (f.partial_d ? 2 : 0) | (f.partial_q ? 4 : 0);
}

static int decoder_backward_issue(const st_decoder_bwd_weights* w, const st_decoder_dims* d, const st_decoder_bwd_io* io, void* stream);

static stlg::Cache g_bwd_graphs;
namespace stlg { int& enabled_flag() { static int v = -1; return v; } }
extern "C" int st_loop_graphs_enable(int on) { int& v = stlg::enabled_flag(); const int old = v; v = on < 0 ? -1 : (on ? 1 : 0); return old; }
extern "C" void stx_fwd_loop_graph_stats(long* out3);
extern "C" void st_loop_graph_stats(long* fwd3, long* bwd3) {
    if (fwd3) stx_fwd_loop_graph_stats(fwd3);
    if (bwd3) {
        std::lock_guard<std::mutex> lock(g_bwd_graphs.mu);
        bwd3[0] = g_bwd_graphs.replays; bwd3[1] = g_bwd_graphs.captures; bwd3[2] = g_bwd_graphs.eager;
    }
}

extern "C" int st_decoder_backward(const st_decoder_bwd_weights* w, const st_decoder_dims* d, const st_decoder_bwd_io* io,
                                   void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(w && d && io, "st_decoder_backward: null struct pointer");
    if (io->steps > 0 && stlg::enabled(d->B)) {              // (see st_decoder_forward: the loop of a training step repeats its arguments)
        st_decoder_bwd_io key_io = *io;
        key_io.step_src = nullptr;
        uint64_t key = stlg::fnv(stlg::FNV0, w, sizeof(*w));
        key = stlg::fnv(key, d, sizeof(*d));
        key = stlg::fnv(key, &key_io, sizeof(key_io));
        if (io->step_src) key = stlg::fnv(key, io->step_src, sizeof(int) * (size_t)io->steps);
        stlg::Entry* e = nullptr;
        hipStream_t issue_on = (hipStream_t)stream;
        const int mode = stlg::begin(g_bwd_graphs, key, (hipStream_t)stream, &e, &issue_on);
        if (mode == 1) return 0;
        int rc = decoder_backward_issue(w, d, io, (void*)issue_on);
        if (mode == 2) {
            rc = stlg::end(g_bwd_graphs, e, (hipStream_t)stream, rc);
            if (rc == -5) st_set_error("st_decoder_backward: capturing the loop into a hipGraph failed");
        }
        return rc;
    }
    return decoder_backward_issue(w, d, io, stream);
}

static int decoder_backward_issue(const st_decoder_bwd_weights* w, const st_decoder_dims* d, const st_decoder_bwd_io* io, void* stream) {
    ST_CHECK_ARG(w && d && io, "st_decoder_backward: null struct pointer");
    const int B = d->B, L = d->L, E = d->E, P = d->P, Q = d->Q, D = d->D, A = d->A;
    const int steps = io->steps, Bp = io->Bp;
    ST_CHECK_ARG(B > 0 && steps > 0 && Bp >= B, "st_decoder_backward: B=%d steps=%d Bp=%d", B, steps, Bp);
    ST_CHECK_ARG(w->attn_v && w->attn_loc_conv_w && w->attn_loc_lin_w, "st_decoder_backward: null weight");
    ST_CHECK_ARG(io->memory && io->pm && io->ada_std && io->align && io->wcum_tape && io->cq_tape && io->cd_tape &&
                 io->gates_q_tape && io->gates_d_tape && io->pq_all && io->dxo, "st_decoder_backward: null saved tensor");
    ST_CHECK_ARG(io->dgq && io->dgd && io->dxq && io->dxd && io->dpq && io->ds_tape && io->loc_tape && io->dloc_tape &&
                 io->hist_tape && io->dctx_tape && io->dv_tape && io->dcq && io->dcd && io->dhist[0] && io->dhist[1] && io->dcum &&
                 io->dhq_attn,
                 "st_decoder_backward: null output/scratch");
    const int XQ = P + E + Q, XD = E + Q + D, XO = D + E;
    const int in_dim = d->r * d->n_mels, YW = in_dim + 1;
    // own-output feedback anywhere in the plan?
    bool own = false;
    if (io->step_src)
        for (int t = 0; t + 1 < steps; ++t)     // (-2 = teacher mean: like a teacher frame, no feedback for rows < Bt; the caller spreads
            own = own || io->step_src[t] == -1 || io->Bt < B;     //  the input gradient of such a step over the teacher's frames)
    ST_CHECK_ARG(!own || (io->dY && io->dxo_rw && io->wpg_t && io->pre_w1_t && io->pre_w0_t && io->xq_nat && io->pre1_nat &&
                          io->d2_tape && io->dp1_tape && io->tmp_p && io->tmp_in && io->Bt > 0 && io->Bt <= B),
                 "st_decoder_backward: own-output feedback needs the prenet tapes / scratch");
    ST_CHECK_ARG(!own || !io->prenet_norm || (io->prenet_norm >= 1 && io->prenet_norm <= 3 && io->pre_y_tape && io->pre_norm_w[0] && io->pre_norm_w[1] &&
                 io->dpre_norm_w[0] && io->dpre_norm_w[1] && io->dpre_norm_b[0] && io->dpre_norm_b[1]),
                 "st_decoder_backward: own-output feedback through a normalised prenet needs pre_y_tape, the norm weights and the gradient accumulators");
    const bool packed = w->q_w_cat_t_p16 && w->d_w_cat_t_p16 && io->dgq_t16 && io->dgd_t16;
    st_t16_view dgq_v = {io->dgq_t16, (4 * Q + 15) >> 4, 0}, dgd_v = {io->dgd_t16, (4 * D + 15) >> 4, 0};
    const size_t BQ = (size_t)B * Q, BD = (size_t)B * D, BL = (size_t)B * L;
    const int ldal = steps * L;
    int rc;
    const bool fuse_pw = io->fuse_pw && !own && packed && w->attn_query_w_t_p16 && io->dgd_t16_b && io->dpq_t16 && bwd_fuse_dims_ok(d) &&
                         (io->q_mask == nullptr || st_aligned16(io->q_mask)) && st_aligned16(io->dxo) && st_aligned16(io->dxd) && st_aligned16(io->dxq);
    // A caller that asks for the fused loop may hand its step tapes over UNINITIALISED (the fused launches write every element; the
    // six-launch loop reads the slot behind the last step as zeros): asking for it where it cannot run is an error, not a silent fall-back
    // (st_decoder_bwd_fuse_dims answers the question about the dimensions beforehand)
    ST_CHECK_ARG(!io->fuse_pw || fuse_pw, "st_decoder_backward: io.fuse_pw requested but the fused loop cannot run (own-output feedback, missing packed "
                 "operands, dimensions st_decoder_bwd_fuse_dims rejects, or a step tape that is not 16-byte aligned)");
    ST_CHECK_ARG(packed || (w->q_w_cat_t && w->d_w_cat_t), "st_decoder_backward: neither packed nor natural [W_ih | W_hh]^T");
    ST_CHECK_ARG(fuse_pw || w->attn_query_w_t, "st_decoder_backward: natural W_q^T missing (needed without fuse_pw)");
    if (fuse_pw) {
        // Four launches per step instead of six: the pointwise half of each cell's backward step runs in the epilogue of the product
        // that makes its dh -- dgates_d(t) . [W_ih_d | W_hh_d] makes dh_d(t-1) in its last D columns (so the decoder cell's pointwise
        // step of t-1 rides there; the T16 copy of dgates_d ping-pongs between two buffers because step t's is still being read), and
        // W_q^T dpq(t), now a packed product over the T16 copy of dpq the attention backward writes, makes the attention's share of dh_q(t).
        float* dgd_buf[2] = {io->dgd_t16, io->dgd_t16_b};
        const int kbd = (4 * D + 15) >> 4;
        st_t16_view dpq_v = {io->dpq_t16, (A + 15) >> 4, 0};
        {   // the last step's decoder-cell pointwise part has no product in front of it
            const int t = steps - 1;
            st_t16_view v0 = {dgd_buf[t & 1], kbd, 0};
            rc = st_lstm_cell_bwd_pointwise(io->dxo + (size_t)t * Bp * XO, XO, nullptr, 0, nullptr, 0, nullptr,
                                            io->d_mask ? io->d_mask + (size_t)t * BD : nullptr,
                                            io->gates_d_tape + (size_t)t * 4 * BD, io->cd_tape + (size_t)(t + 1) * BD, D,
                                            io->cd_tape + (size_t)t * BD, D, io->dcd, io->dgd + (size_t)t * Bp * 4 * D, 4 * D, &v0, B, D, stream);
            if (rc) return rc;
        }
        // launch "b" of step t: dxd_t = dgates_d_t [W_ih_d | W_hh_d] (+ the decoder cell's pointwise step of t-1 in its epilogue: columns
        // [E+Q, E+Q+D) + dxo_{t-1}[:, :D] = dh_d(t-1)), optionally with the attention backward `ab` of the step AFTER it beside it
        // the decoder cell's pointwise step of t-1 that rides behind the product of step t (t > 0)
        auto pw_d = [&](int t, st_lstm_pw_job& j) {
            memset(&j, 0, sizeof(j));
            j.n0 = E + Q; j.H = D;
            j.dh1 = io->dxo + (size_t)(t - 1) * Bp * XO; j.ld1 = XO;
            j.mask = io->d_mask ? io->d_mask + (size_t)(t - 1) * BD : nullptr;
            j.gates = io->gates_d_tape + (size_t)(t - 1) * 4 * BD;
            j.c = io->cd_tape + (size_t)t * BD; j.ldc = D; j.c_prev = io->cd_tape + (size_t)(t - 1) * BD; j.ldcp = D;
            j.dc = io->dcd; j.dgates = io->dgd + (size_t)(t - 1) * Bp * 4 * D; j.ldg = 4 * D;
            j.dgates_t16.base = dgd_buf[(t - 1) & 1]; j.dgates_t16.kb_stride = kbd; j.dgates_t16.kb0 = 0;
        };
        auto product_d = [&](int t, const st_attn_bwd_job* ab) -> int {
            float* dxd = io->dxd + (size_t)t * Bp * XD;
            st_t16_view x_v = {dgd_buf[t & 1], kbd, 0};
            st_lstm_pw_job j;
            memset(&j, 0, sizeof(j));
            if (t > 0) {
                j.n0 = E + Q; j.H = D;
                j.dh1 = io->dxo + (size_t)(t - 1) * Bp * XO; j.ld1 = XO;
                j.mask = io->d_mask ? io->d_mask + (size_t)(t - 1) * BD : nullptr;
                j.gates = io->gates_d_tape + (size_t)(t - 1) * 4 * BD;
                j.c = io->cd_tape + (size_t)t * BD; j.ldc = D; j.c_prev = io->cd_tape + (size_t)(t - 1) * BD; j.ldcp = D;
                j.dc = io->dcd; j.dgates = io->dgd + (size_t)(t - 1) * Bp * 4 * D; j.ldg = 4 * D;
                j.dgates_t16.base = dgd_buf[(t - 1) & 1]; j.dgates_t16.kb_stride = kbd; j.dgates_t16.kb0 = 0;
            }
            if (ab) return st_skinny_linear_packed_lstm_bwd_attn_bwd(w->d_w_cat_t_p16, &x_v, 4 * D, dxd, XD, B, XD, t > 0 ? &j : nullptr, ab, stream);
            if (t > 0) return st_skinny_linear_packed_lstm_bwd_fwd(w->d_w_cat_t_p16, &x_v, 4 * D, dxd, XD, B, XD, &j, stream);
            return st_skinny_linear_packed_fwd(w->d_w_cat_t_p16, &x_v, 4 * D, nullptr, ST_ACT_NONE, nullptr, 0, dxd, XD, nullptr,
                                               0, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, B, XD, stream);
        };
        // the decoder cell's recurrence (dgates_d(t-1) from dgates_d(t) . W_hh_d and the output gradients) does not touch the attention /
        // query chain of step t: with overlap_attn its product runs ONE STEP AHEAD, beside the attention backward of step t
        const bool overlap = io->overlap_attn && io->attn_s_tape;
        const BwdForms forms = bwd_forms(d, io);
        const int parts = forms.parts, dsplits = forms.dsplits, qsplits = forms.qsplits;
        const bool partial = forms.partial_d, partial_q = forms.partial_q;
        const size_t qslab = (size_t)B * XQ;                 // one slab of dxq_part; a step holds qsplits of them
        if (overlap) { rc = product_d(steps - 1, nullptr); if (rc) return rc; }
        for (int t = steps - 1; t >= 0; --t) {
            const float* dxo = io->dxo + (size_t)t * Bp * XO;
            float* dxd = io->dxd + (size_t)t * Bp * XD;
            float* dxq = io->dxq + (size_t)t * Bp * XQ;
            // (no step behind the last one: absent addends instead of a zero slot -- the tapes need not start from zeros)
            const float* dxq_next = t + 1 < steps ? (partial_q ? io->dxq_part + (size_t)(t + 1) * qsplits * qslab : io->dxq + (size_t)(t + 1) * Bp * XQ) : nullptr;
            float* dpq = io->dpq + (size_t)t * Bp * A;
            float* dhist_cur = io->dhist[t & 1];
            const float* dhist_next = io->dhist[(t + 1) & 1];
            if (!overlap) { rc = product_d(t, nullptr); if (rc) return rc; }
            // c. attention (dpq also in T16)
            st_attn_bwd_job ab;
            memset(&ab, 0, sizeof(ab));
            ab.pq = io->pq_all + (size_t)t * Bp * A; ab.pm = io->pm; ab.memory = io->memory;
            ab.w_prev = t > 0 ? io->align + (size_t)(t - 1) * L : nullptr; ab.ld_wprev = ldal; ab.w_cum_prev = io->wcum_tape + (size_t)t * BL;
            ab.w = io->align + (size_t)t * L; ab.ld_w = ldal;
            ab.loc_conv_w = w->attn_loc_conv_w; ab.loc_lin_w = w->attn_loc_lin_w; ab.v = w->attn_v;
            ab.dctx[0] = dxo + D; ab.dctx[1] = dxd; ab.dctx[2] = dxq_next ? dxq_next + P : nullptr; ab.ld_dctx[0] = XO; ab.ld_dctx[1] = XD; ab.ld_dctx[2] = XQ; ab.n_dctx = 3;
            if (partial_q && dxq_next) {     // the other slabs of dxq_{t+1} (slab order = addend order: fixed)
                for (int sl = 1; sl < qsplits; ++sl) { ab.dctx_more[sl - 1] = dxq_next + (size_t)sl * qslab + P; ab.ld_dctx_more[sl - 1] = XQ; }
                ab.n_dctx_more = qsplits - 1;
            }
            ab.dw_direct[0] = dhist_next; ab.dw_direct[1] = io->dalign ? io->dalign + (size_t)t * L : nullptr; ab.ld_dw[0] = 2 * L; ab.ld_dw[1] = ldal;
            ab.n_dw = io->dalign ? 2 : 1;
            // (split form: the history job of step t+1 has already added dhist(t+1)[1] into dcum)
            ab.dcum = io->dcum; ab.dcum_add = parts > 1 ? nullptr : dhist_next + L; ab.ld_dcum_add = 2 * L;
            const bool split = parts > 1 && overlap && t > 0;
            if (split) { ab.parts = parts; ab.dloc_part = io->dloc_part; }
            ab.dpq = dpq; ab.dpq_t16 = dpq_v; ab.dhist = dhist_cur; ab.ds_t = io->ds_tape + (size_t)t * BL * A;
            ab.loc_t = io->loc_tape + (size_t)t * BL * d->F; ab.dloc_t = io->dloc_tape + (size_t)t * BL * d->F;
            ab.hist_t = io->hist_tape + (size_t)t * BL * 2; ab.dctx_t = io->dctx_tape + (size_t)t * B * E; ab.dv_t = io->dv_tape + (size_t)t * B * A;
            ab.s_in = io->attn_s_tape ? (t == 0 ? io->pm : io->attn_s_tape + (size_t)t * BL * A) : nullptr;
            ab.B = B; ab.L = L; ab.A = A; ab.E = E; ab.F = d->F; ab.K = d->K;
            if (split && partial) {      // [attention backward of t, two parts | K-split partial product of the decoder cell, step t-1]
                st_t16_view x_v = {dgd_buf[(t - 1) & 1], kbd, 0};
                rc = st_skinny_partial_attn_bwd(w->d_w_cat_t_p16, &x_v, 4 * D, io->dxd_part, dsplits, B, XD, &ab, stream);
            } else if (overlap && t > 0) rc = product_d(t - 1, &ab);        // [attention backward of t | decoder cell product of t-1]: one launch
            else {
                const float* dcx[6]; int ldx[6]; int ncx = 0;       // (the standalone call takes the slabs of dxq_{t+1} in one addend list)
                for (int q_ = 0; q_ < ab.n_dctx; ++q_) { dcx[ncx] = ab.dctx[q_]; ldx[ncx++] = ab.ld_dctx[q_]; }
                for (int q_ = 0; q_ < ab.n_dctx_more; ++q_) { dcx[ncx] = ab.dctx_more[q_]; ldx[ncx++] = ab.ld_dctx_more[q_]; }
                rc = st_attn_step_bwd_t16(ab.pq, ab.pm, ab.memory, ab.w_prev, ab.ld_wprev, ab.w_cum_prev, ab.w, ab.ld_w, ab.loc_conv_w, ab.loc_lin_w,
                                           ab.v, dcx, ldx, ncx, ab.dw_direct, ab.ld_dw, ab.n_dw, ab.dcum, ab.dcum_add, ab.ld_dcum_add,
                                           ab.dpq, &ab.dpq_t16, ab.dhist, ab.ds_t, ab.loc_t, ab.dloc_t, ab.hist_t, ab.dctx_t, ab.dv_t, ab.s_in,
                                           B, L, A, E, d->F, d->K, stream);
            }
            if (rc) return rc;
            // d + e. dh_q = W_q^T dpq + (W_hh_q^T dgates_q)_{t+1} + std * d(adapted h_q): the query cell's pointwise part in the epilogue
            {
                st_lstm_pw_job j;
                memset(&j, 0, sizeof(j));
                j.n0 = 0; j.H = Q;
                j.dh1 = dxq_next ? dxq_next + P + E : nullptr; j.ld1 = XQ;
                if (partial_q && dxq_next) { j.dh1_slabs = qsplits; j.dh1_slab_stride = (long)qslab; }
                j.dh2 = dxd + E; j.ld2 = XD; j.scale2 = io->ada_std;
                j.mask = io->q_mask ? io->q_mask + (size_t)t * BQ : nullptr;
                j.gates = io->gates_q_tape + (size_t)t * 4 * BQ;
                j.c = io->cq_tape + (size_t)(t + 1) * BQ; j.ldc = Q; j.c_prev = io->cq_tape + (size_t)t * BQ; j.ldcp = Q;
                j.dc = io->dcq; j.dgates = io->dgq + (size_t)t * Bp * 4 * Q; j.ldg = 4 * Q;
                j.dgates_t16 = dgq_v;
                if (split && partial) {
                    // ... with the slabs of launch 1's partial product beside it: dxd_{t-1} + the decoder cell's pointwise step of t-2
                    st_lstm_pw_job jd;
                    st_partial_sum_job sj;
                    memset(&sj, 0, sizeof(sj));
                    sj.part = io->dxd_part; sj.S = dsplits; sj.N = XD; sj.y = io->dxd + (size_t)(t - 1) * Bp * XD; sj.ldy = XD;
                    if (t - 1 > 0) { pw_d(t - 1, jd); sj.pw = &jd; }
                    rc = st_skinny_linear_packed_lstm_bwd_attn_hist_sum(w->attn_query_w_t_p16, &dpq_v, A, io->dhq_attn, Q, B, Q, &j, nullptr, &sj, stream);
                } else rc = st_skinny_linear_packed_lstm_bwd_fwd(w->attn_query_w_t_p16, &dpq_v, A, io->dhq_attn, Q, B, Q, &j, stream);
                if (rc) return rc;
            }
            // f. gradient w.r.t. [dec_in_t | ctx_{t-1} | h_q_{t-1}]
            if (split) {
                // ... with what the split attention backward of this step left behind (the sum of its partial dloc, the history tape, the
                // conv-transpose to dhist, the carried dcum) on the compute units this product leaves idle
                st_attn_hist_job hj;
                memset(&hj, 0, sizeof(hj));
                hj.dloc_part = io->dloc_part; hj.parts = parts; hj.loc_conv_w = w->attn_loc_conv_w;
                hj.w_prev = ab.w_prev; hj.ld_wprev = ab.ld_wprev; hj.w_cum_prev = ab.w_cum_prev;
                hj.dloc_t = ab.dloc_t; hj.hist_t = ab.hist_t; hj.dhist = dhist_cur; hj.dcum = io->dcum;
                hj.B = B; hj.L = L; hj.F = d->F; hj.K = d->K;
                if (partial_q) rc = st_skinny_partial_attn_hist(w->q_w_cat_t_p16, &dgq_v, 4 * Q, io->dxq_part + (size_t)t * qsplits * qslab, qsplits, B, XQ, &hj, stream);
                else rc = st_skinny_linear_packed_attn_hist(w->q_w_cat_t_p16, &dgq_v, 4 * Q, dxq, XQ, B, XQ, &hj, stream);
                if (rc) return rc;
            } else if (t > 0 || io->need_dxq0) {
                rc = st_skinny_linear_packed_fwd(w->q_w_cat_t_p16, &dgq_v, 4 * Q, nullptr, ST_ACT_NONE, nullptr, 0, dxq, XQ, nullptr,
                                                 0, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, B, XQ, stream);
                if (rc) return rc;
            }
        }
        return 0;
    }
    for (int t = steps - 1; t >= 0; --t) {
        const float* dxo = io->dxo + (size_t)t * Bp * XO;
        float* dxd = io->dxd + (size_t)t * Bp * XD;
        const float* dxd_next = io->dxd + (size_t)(t + 1) * Bp * XD;
        float* dxq = io->dxq + (size_t)t * Bp * XQ;
        const float* dxq_next = io->dxq + (size_t)(t + 1) * Bp * XQ;
        float* dgd = io->dgd + (size_t)t * Bp * 4 * D;
        float* dgq = io->dgq + (size_t)t * Bp * 4 * Q;
        float* dpq = io->dpq + (size_t)t * Bp * A;
        float* dhist_cur = io->dhist[t & 1];
        const float* dhist_next = io->dhist[(t + 1) & 1];

        if (own) {
            // 0. own-output feedback: dec_in_{t+1}[rows] = prenet(mel_t[rows]) -> add its gradient to dmel_t, then
            //    push [dmel_t | dstop_t] through proj (+) gate (done for all steps at once by the caller otherwise)
            float* dY = io->dY + (size_t)t * Bp * YW;
            const int r0 = (t + 1 < steps) ? (io->step_src[t] == -1 ? 0 : io->Bt) : B;      // first own row (B = none)
            const int nr = B - r0;
            if (nr > 0) {
                const size_t BP = (size_t)B * P;
                const float* m1 = io->own_mask ? io->own_mask + ((size_t)t * 2 + 0) * BP + (size_t)r0 * P : nullptr;
                const float* m2 = io->own_mask ? io->own_mask + ((size_t)t * 2 + 1) * BP + (size_t)r0 * P : nullptr;
                float* d2 = io->d2_tape + ((size_t)t * Bp + r0) * P;
                float* dp1 = io->dp1_tape + ((size_t)t * Bp + r0) * P;
                rc = st_act_bwd(dxq_next + (size_t)r0 * XQ, XQ, io->xq_nat + ((size_t)(t + 1) * Bp + r0) * XQ, XQ, ST_ACT_RELU,
                                m2, P, d2, P, nr, P, stream);
                if (rc) return rc;
                if (io->prenet_norm) {     // through the norm of layer 2: d2 becomes the gradient at the Linear's output
                    rc = st_prenet_norm_bwd(d2, P, io->pre_y_tape + (((size_t)t * 2 + 1) * B + r0) * P, P, io->prenet_norm, io->pre_norm_w[1],
                                            io->pre_norm_rm[1], io->pre_norm_rv[1], io->pre_norm_eps, io->dpre_norm_w[1], io->dpre_norm_b[1],
                                            nr, P, stream);
                    if (rc) return rc;
                }
                st_seg sg;
                sg.x = d2; sg.ldx = P; sg.w = io->pre_w1_t; sg.ldw = P; sg.k = P;
                rc = st_skinny_linear_fwd(&sg, 1, nullptr, ST_ACT_NONE, nullptr, 0, io->tmp_p, P, 0, nullptr, 0, 0, nr, P, stream);
                if (rc) return rc;
                rc = st_act_bwd(io->tmp_p, P, io->pre1_nat + ((size_t)t * Bp + r0) * P, P, ST_ACT_RELU, m1, P, dp1, P, nr, P, stream);
                if (rc) return rc;
                if (io->prenet_norm) {
                    rc = st_prenet_norm_bwd(dp1, P, io->pre_y_tape + (((size_t)t * 2 + 0) * B + r0) * P, P, io->prenet_norm, io->pre_norm_w[0],
                                            io->pre_norm_rm[0], io->pre_norm_rv[0], io->pre_norm_eps, io->dpre_norm_w[0], io->dpre_norm_b[0],
                                            nr, P, stream);
                    if (rc) return rc;
                }
                sg.x = dp1; sg.ldx = P; sg.w = io->pre_w0_t; sg.ldw = P; sg.k = P;
                rc = st_skinny_linear_fwd(&sg, 1, nullptr, ST_ACT_NONE, nullptr, 0, io->tmp_in, in_dim, 0, nullptr, 0, 0, nr, in_dim, stream);
                if (rc) return rc;
                rc = st_copy3d(dY + (size_t)r0 * YW, YW, YW, io->tmp_in, in_dim, in_dim, nr, 1, in_dim, 1, stream);
                if (rc) return rc;
            }
            st_seg so;
            so.x = dY; so.ldx = YW; so.w = io->wpg_t; so.ldw = YW; so.k = YW;
            rc = st_skinny_linear_fwd(&so, 1, nullptr, ST_ACT_NONE, nullptr, 0, io->dxo_rw + (size_t)t * Bp * XO, XO, 0, nullptr, 0, 0,
                                      B, XO, stream);
            if (rc) return rc;
        }
        // a. decoder LSTM pointwise
        rc = st_lstm_cell_bwd_pointwise(dxo, XO, dxd_next + E + Q, XD, nullptr, 0, nullptr,
                                        io->d_mask ? io->d_mask + (size_t)t * BD : nullptr,
                                        io->gates_d_tape + (size_t)t * 4 * BD, io->cd_tape + (size_t)(t + 1) * BD, D,
                                        io->cd_tape + (size_t)t * BD, D, io->dcd, dgd, 4 * D, packed ? &dgd_v : nullptr, B, D, stream);
        if (rc) return rc;
        // b. gradient w.r.t. [ctx_t | adapted h_q_t | h_d_{t-1}]
        st_seg seg;
        if (packed) {   // W^T streamed in MFMA lane order (P16), dgates in T16
            rc = st_skinny_linear_packed_fwd(w->d_w_cat_t_p16, &dgd_v, 4 * D, nullptr, ST_ACT_NONE, nullptr, 0, dxd, XD, nullptr,
                                             0, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, B, XD, stream);
        } else {
            seg.x = dgd; seg.ldx = 4 * D; seg.w = w->d_w_cat_t; seg.ldw = 4 * D; seg.k = 4 * D;
            rc = st_skinny_linear_fwd(&seg, 1, nullptr, ST_ACT_NONE, nullptr, 0, dxd, XD, 0, nullptr, 0, 0, B, XD, stream);
        }
        if (rc) return rc;
        // c. attention
        const float* dctx[3] = {dxo + D, dxd, dxq_next + P};
        const int ld_dctx[3] = {XO, XD, XQ};
        const float* dwd[2] = {dhist_next, io->dalign ? io->dalign + (size_t)t * L : nullptr};
        const int ld_dw[2] = {2 * L, ldal};
        // S_t from the forward when it kept it (S_0 = pm: no history before the first step); loc_tape is then an input
        const float* s_in = io->attn_s_tape ? (t == 0 ? io->pm : io->attn_s_tape + (size_t)t * BL * A) : nullptr;
        rc = st_attn_step_bwd_s(io->pq_all + (size_t)t * Bp * A, io->pm, io->memory,
                              t > 0 ? io->align + (size_t)(t - 1) * L : nullptr, ldal, io->wcum_tape + (size_t)t * BL,
                              io->align + (size_t)t * L, ldal, w->attn_loc_conv_w, w->attn_loc_lin_w, w->attn_v,
                              dctx, ld_dctx, 3, dwd, ld_dw, io->dalign ? 2 : 1,
                              io->dcum, dhist_next + L, 2 * L,
                              dpq, dhist_cur, io->ds_tape + (size_t)t * BL * A, io->loc_tape + (size_t)t * BL * d->F,
                              io->dloc_tape + (size_t)t * BL * d->F, io->hist_tape + (size_t)t * BL * 2,
                              io->dctx_tape + (size_t)t * B * E, io->dv_tape + (size_t)t * B * A, s_in,
                              B, L, A, E, d->F, d->K, stream);
        if (rc) return rc;
        // d. through the query projection
        seg.x = dpq; seg.ldx = A; seg.w = w->attn_query_w_t; seg.ldw = A; seg.k = A;
        rc = st_skinny_linear_fwd(&seg, 1, nullptr, ST_ACT_NONE, nullptr, 0, io->dhq_attn, Q, 0, nullptr, 0, 0, B, Q, stream);
        if (rc) return rc;
        // e. query LSTM pointwise (AdaIN: adapted = std * (h_q - mean) -> dh_q += std * d adapted)
        rc = st_lstm_cell_bwd_pointwise(io->dhq_attn, Q, dxq_next + P + E, XQ, dxd + E, XD, io->ada_std,
                                        io->q_mask ? io->q_mask + (size_t)t * BQ : nullptr,
                                        io->gates_q_tape + (size_t)t * 4 * BQ, io->cq_tape + (size_t)(t + 1) * BQ, Q,
                                        io->cq_tape + (size_t)t * BQ, Q, io->dcq, dgq, 4 * Q, packed ? &dgq_v : nullptr, B, Q, stream);
        if (rc) return rc;
        // f. gradient w.r.t. [dec_in_t | ctx_{t-1} | h_q_{t-1}]  (step 0: go frame and zero initial state, nothing to do)
        if (t > 0 || io->need_dxq0) {
            if (packed) {
                rc = st_skinny_linear_packed_fwd(w->q_w_cat_t_p16, &dgq_v, 4 * Q, nullptr, ST_ACT_NONE, nullptr, 0, dxq, XQ, nullptr,
                                                 0, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, B, XQ, stream);
            } else {
                seg.x = dgq; seg.ldx = 4 * Q; seg.w = w->q_w_cat_t; seg.ldw = 4 * Q; seg.k = 4 * Q;
                rc = st_skinny_linear_fwd(&seg, 1, nullptr, ST_ACT_NONE, nullptr, 0, dxq, XQ, 0, nullptr, 0, 0, B, XQ, stream);
            }
            if (rc) return rc;
        }
    }
    return 0;
}

namespace {

// dY[t][b][:] = [dmel[b][t*r .. t*r+r][:] | sum_j dstop[b][t*r + j]]     rows b >= B of a slot are left untouched (zero)
__global__ __launch_bounds__(256) void pack_dout_kernel(const float* dmel, const float* dstop, float* dY,
                                                        int B, int Bp, int steps, int r, int n_mels, int ld) {
    const int in_dim = r * n_mels, W = ld;          // (all ld columns of a row: the pad columns past in_dim + 1 are written as zeros)
    const size_t total = (size_t)steps * B * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % W);
        const size_t row = i / W;
        const int b = (int)(row % B), t = (int)(row / B);
        float v;
        if (c < in_dim) v = dmel ? dmel[((size_t)b * steps + t) * in_dim + c] : 0.0f;
        else if (c > in_dim) v = 0.0f;
        else {
            v = 0.0f;
            if (dstop) for (int j = 0; j < r; ++j) v += dstop[(size_t)b * steps * r + (size_t)t * r + j];
        }
        dY[((size_t)t * Bp + b) * ld + c] = v;
    }
}

// the forward twin: Y[t][b][:] = [mel_t | stop_t] rows of the deferred proj (+) gate product -> mel[b][t*r .. ][:], stop[b][t*r + j] (the
// stop value of a step repeated r times, src/module.py:287): one launch instead of 1 + r strided copies
__global__ __launch_bounds__(256) void unpack_out_kernel(const float* Y, float* mel, float* stop, int B, int Bp, int steps, int r, int n_mels, int ld) {
    const int in_dim = r * n_mels, W = in_dim + r;
    const size_t total = (size_t)steps * B * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % W);
        const size_t row = i / W;
        const int t = (int)(row % steps), b = (int)(row / steps);
        const float* y = Y + ((size_t)t * Bp + b) * ld;
        if (c < in_dim) mel[((size_t)b * steps + t) * in_dim + c] = y[c];
        else stop[(size_t)b * steps * r + (size_t)t * r + (c - in_dim)] = y[in_dim];
    }
}

// gradient of prenet(teacher) under plain teacher forcing (step t + 1 read teacher frame t): dteacher[b][t][:] = the dec_in columns of
// dxq_{t+1}, the S slabs of the K-split product added in slab order; frames no step read (t >= steps - 1) get zeros.  One launch instead
// of a fill and one strided copy per slab.
__global__ __launch_bounds__(256) void dteacher_sum_kernel(const float* dxq_part, float* dteacher, int S, int Bp, int XQw, int Bt, int Tt, int P, int steps) {
    const size_t total = (size_t)Bt * Tt * P;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % P);
        const size_t bt = i / P;
        const int t = (int)(bt % Tt), b = (int)(bt / Tt);
        float v = 0.0f;
        if (t < steps - 1) {
            const float* src = dxq_part + (((size_t)(t + 1) * S) * Bp + b) * XQw + p;
            for (int sl = 0; sl < S; ++sl) v += src[(size_t)sl * Bp * XQw];
        }
        dteacher[i] = v;
    }
}

// AdaIN parameter gradients: adapted_t = std * (h_q_t - mean)
//   dstd[b][q] = sum_t da[t][b][q] * (h_q_t[b][q] - mean[b][q]);   dmean[b][q] = -std[b][q] * sum_t da[t][b][q]
__global__ __launch_bounds__(256) void adain_bwd_kernel(const float* da, long da_st, int da_ld, const float* hq, long hq_st, int hq_ld,
                                                        const float* std_, const float* mean, float* dstd, float* dmean,
                                                        int B, int Q, int steps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Q) return;
    const int b = i / Q, q = i - b * Q;
    const float mu = mean[i];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll 8
    for (int t = 0; t < steps; ++t) {      // (unrolled: sixteen loads in flight; rolled it is a chain of `steps` dependent round trips)
        const float g = da[(size_t)t * da_st + (size_t)b * da_ld + q];
        s1 += g;
        s2 = fmaf(g, hq[(size_t)t * hq_st + (size_t)b * hq_ld + q] - mu, s2);
    }
    dstd[i] = s2;
    dmean[i] = -std_[i] * s1;
}

}  // namespace

extern "C" int st_decoder_pack_dout(const float* dmel, const float* dstop, float* dY, int ld, int B, int Bp, int steps, int r,
                                    int n_mels, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dY && B > 0 && Bp >= B && steps > 0 && r > 0 && n_mels > 0 && ld >= r * n_mels + 1, "st_decoder_pack_dout: bad arguments");
    const size_t total = (size_t)steps * B * ld;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_dout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dmel, dstop, dY, B, Bp,
                       steps, r, n_mels, ld);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_decoder_unpack_out(const float* Y, float* mel, float* stop, int B, int Bp, int steps, int r, int n_mels, int ld, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(Y && mel && stop && B > 0 && Bp >= B && steps > 0 && r > 0 && n_mels > 0 && ld >= r * n_mels + 1, "st_decoder_unpack_out: bad arguments");
    const size_t total = (size_t)steps * B * (r * n_mels + r);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(unpack_out_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, Y, mel, stop, B, Bp, steps, r, n_mels, ld);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_decoder_dteacher_sum(const float* dxq_part, float* dteacher, int S, int Bp, int XQw, int Bt, int Tt, int P, int steps, void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dxq_part && dteacher && S > 0 && Bp >= Bt && Bt > 0 && Tt > 0 && P > 0 && XQw >= P && steps > 0, "st_decoder_dteacher_sum: bad arguments");
    const size_t total = (size_t)Bt * Tt * P;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dteacher_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dxq_part, dteacher, S, Bp, XQw, Bt, Tt, P, steps);
    ST_LAUNCH_CHECK();
    return 0;
}

extern "C" int st_adain_bwd(const float* dadapt, long da_step_stride, int da_ld, const float* hq, long hq_step_stride, int hq_ld,
                            const float* ada_std, const float* ada_mean, float* dstd, float* dmean, int B, int Q, int steps,
                            void* stream) {
    (void)hipGetLastError();
    ST_CHECK_ARG(dadapt && hq && ada_std && ada_mean && dstd && dmean && B > 0 && Q > 0 && steps > 0, "st_adain_bwd: bad arguments");
    hipLaunchKernelGGL(adain_bwd_kernel, dim3((B * Q + 255) / 256), dim3(256), 0, (hipStream_t)stream, dadapt, da_step_stride,
                       da_ld, hq, hq_step_stride, hq_ld, ada_std, ada_mean, dstd, dmean, B, Q, steps);
    ST_LAUNCH_CHECK();
    return 0;
}
