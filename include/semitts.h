/* semitts.h -- C ABI of libsemitts_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the Tacotron-style TTS decode hot path and the VQ codebook lookup of
 * ttaoREtw/semi-tts.  The reference has NO native/FFI layer (it is pure PyTorch, SURVEY.md
 * section 8b): each entry point below replaces the torch.nn / torch.nn.functional call(s)
 * of the reference cited next to it (`ref:` = path:line under the reference tree).  The
 * Python host (the .py files of semi_tts_amd/) binds these with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous row-major fp32 (or int64 where said);
 *    the caller (PyTorch) owns every buffer; the library allocates nothing persistent
 *    except graphs/streams explicitly created through st_graph_* / st_stream_*;
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and is
 *    hipGraph-capturable (no host sync, no malloc, no blocking memcpy inside);
 *  - return value 0 = success; negative = error, text via st_last_error() (thread-local);
 *  - "ld*" arguments are row strides in elements.
 */
#ifndef SEMITTS_H
#define SEMITTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST_ABI_VERSION 1

/* activation codes */
#define ST_ACT_NONE 0
#define ST_ACT_RELU 1
#define ST_ACT_TANH 2
#define ST_ACT_SIGMOID 3

const char* st_last_error(void);
int st_abi_version(void);
/* number of compute units / device name of the current device (diagnostics for bench.py) */
int st_device_info(int* n_cu, int* lds_bytes, char* name, int name_len);

/* ------------------------------------------------------------------ streams and graphs */
/* HIP stream + hipGraph plumbing so that a launch-bound sequence of st_* calls (the 86..355
 * step decode loop) is recorded once and replayed as ONE graph launch. */
int st_stream_create(void** stream_out);
int st_stream_destroy(void* stream);
int st_stream_sync(void* stream);
int st_graph_begin(void* stream);                       /* hipStreamBeginCapture            */
int st_graph_end(void* stream, void** graph_exec_out);  /* EndCapture + Instantiate         */
int st_graph_launch(void* graph_exec, void* stream);
int st_graph_destroy(void* graph_exec);
/* HIP-event timing on a given stream (bench.py measures kernels on the stream they run on) */
int st_event_create(void** ev_out);
int st_event_record(void* ev, void* stream);
int st_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* syncs on ev_stop */
int st_event_destroy(void* ev);

/* ------------------------------------------------------------------ skinny (small-batch) ops
 * Weight-streaming kernels for M = batch <= a few dozen rows: the weight matrix is the MFMA
 * A operand (16 rows per workgroup, K split over the waves of the workgroup), the batch is
 * the MFMA N dimension, so every weight byte is read from HBM exactly once per call.
 * The logical input is the concatenation of up to 3 segments x_s (B, k_s) with row stride
 * ldx_s, multiplied by W_s (N, k_s) with row stride ldw_s (slices of torch weights).
 */
typedef struct st_seg {
    const float* x; /* (B, k) activations                       */
    const float* w; /* (N, k) weight slice, torch [out][in]     */
    int ldx;
    int ldw;
    int k;
} st_seg;

/* One LSTMCell step, fused gate GEMM + pointwise (+ hidden-state dropout).
 * ref: nn.LSTMCell src/module.py:127-128,133-134 called :228,:277; nn.Dropout :230,:279;
 *      nn.LSTM time step src/module.py:458-460 (with `pre` = x W_ih^T + b_ih + b_hh).
 * gates(b, g*H+u) = sum_s x_s W_s^T + b_ih + b_hh + pre ; order (i,f,g,o)
 * c' = s(f) c + s(i) tanh(g);  h' = s(o) tanh(c');  h_out = h' * mask (mask already scaled by 1/(1-p))
 * b_ih, b_hh, pre, mask, gates_out may be NULL.  gates_out (B,4,H) receives the ACTIVATED
 * gates (i,f,g,o) for the backward pass.  c_prev/c_out and h_out are (B,H) with row strides
 * ldc / ldh (so they can point into a (B,T,H) sequence tensor). */
int st_lstm_cell_fwd(const st_seg* segs, int nseg, const float* b_ih, const float* b_hh,
                     const float* pre, int ldpre, const float* c_prev, int ldc_prev,
                     const float* mask, float* h_out, int ldh, float* c_out, int ldc,
                     float* gates_out, int B, int H, void* stream);

/* y(b, n) = act(sum_s x_s W_s^T + bias) * mask      (B small)
 * ref: Linear wrapper src/module.py:500-522 (prenet :337-339, query_layer :380,
 *      pseudo_latent_mean/std :268-269).
 * If n_split > 0 the output columns n >= n_split go to y2 instead, each value repeated
 * `rep` times: the proj (+) gate_layer pair of src/module.py:285-287 in one launch
 * (y = mel (B, r*n_mels), y2 = stop (B, r)). */
int st_skinny_linear_fwd(const st_seg* segs, int nseg, const float* bias, int act,
                         const float* mask, int ldmask, float* y, int ldy,
                         int n_split, float* y2, int ldy2, int rep,
                         int B, int N, void* stream);

/* ------------------------------------------------------------------ pre-packed operands (decode loop)
 * A global load whose lanes touch 16 different cache lines per quarter-wave (the natural MFMA
 * operand pattern over row-major data) sustains ~18 B/clk per CU on MI355X; 1 KiB-contiguous
 * wave loads sustain ~90-140 B/clk (tools/mb).  The decode loop therefore keeps its operands
 * in MFMA lane order in HBM:
 *   P16 (weights):      [row tile of 16][k block of 16][lane 0..63][4 floats]
 *   T16 (activations):  [batch tile of 16][k block of 16][lane 0..63][4 floats]
 *   element (r, k) of a block lives at lane 16*((k>>2)&3) + (r&15), component k&3.
 * Rows/columns beyond the logical shape are zero.  T16 destinations must be zero-filled by the
 * caller once (kernels only write logical elements). */
typedef struct st_t16_view {   /* a k-block range [kb0, kb0 + ceil(k/16)) of a T16 buffer */
    float* base;      /* start of the T16 buffer                                     */
    int kb_stride;    /* k-blocks per batch tile of the WHOLE buffer                 */
    int kb0;          /* first k-block of this view                                  */
} st_t16_view;
/* Several logical activations that one consumer multiplies with ONE packed weight matrix share one
 * T16 buffer (e.g. [dec_in | ctx | h_q] for the query LSTM): each producer writes its k-block range
 * of every consumer's buffer, so a consumer streams exactly one contiguous activation run. */

size_t st_packed_weight_floats(const int* k, int nseg, int N, int lstm_H); /* size of a P16 buffer */
size_t st_t16_floats(int B, int K);                                        /* size of a T16 buffer */
/* Pack up to 3 K-segments w[s] (N, k[s]) (row stride ldw[s], torch [out][in]) into one P16 buffer
 * whose K axis is the concatenation of the segments (each padded to a multiple of 16).
 * lstm_H > 0: N == 4*lstm_H and the 16 rows of tile t are the (i,f,g,o) rows of hidden units
 * 4t..4t+3 (so the cell update is workgroup-local). */
int st_pack_weight(const float* const* w, const int* ldw, const int* k, int nseg, int N, int lstm_H,
                   float* packed, void* stream);
/* ... for up to eight matrices in ONE launch (the six matrices of st_decoder_pack were six launches of ~10 us per forward / weight update) */
typedef struct st_pack_job { const float* w[3]; int ldw[3]; int k[3]; int nseg; int N; int lstm_H; float* packed; } st_pack_job;
int st_pack_weight_batch(const st_pack_job* jobs, int n, void* stream);
/* P16 image of the TRANSPOSE of the column concatenation [w[0] | w[1] | ...] (each (K, cols[s]), row stride ldw[s]): N = sum cols,
 * one segment of K columns -- what st_pack_weight would make of torch.cat(w, 1).t().contiguous(), without those two copies
 * (the BPTT loop's W^T operands: backward of nn.LSTMCell / Linear, src/module.py:247-283). */
int st_pack_weight_t(const float* const* w, const int* ldw, const int* cols, int nseg, int K, float* packed, void* stream);
int st_tile_rows(const float* src, int ld, const st_t16_view* dst, int B, int K, void* stream);   /* natural -> T16 */
int st_untile_rows(const st_t16_view* src, float* dst, int ld, int B, int K, void* stream);       /* T16 -> natural */
/* st_lstm_cell_fwd on packed operands: x = K logical columns starting at view x.  The new hidden
 * state goes to up to two T16 destinations; optionally also the AdaIN-adapted hidden state
 * hadapt = ada_std * (h_out - ada_mean)  (ada_* natural (B,H)).
 * ref: as st_lstm_cell_fwd, AdaIN src/module.py:268-269 */
int st_lstm_cell_packed_fwd(const float* packed_w, const st_t16_view* x, int K,
                            const float* b_ih, const float* b_hh,
                            const float* c_prev, int ldc_prev, const float* mask,
                            const st_t16_view* h_dst0, const st_t16_view* h_dst1,
                            float* c_out, int ldc, float* gates_out,
                            const float* ada_std, const float* ada_mean, const st_t16_view* hadapt_dst,
                            int B, int H, void* stream);
/* st_lstm_cell_packed_fwd (without AdaIN) over the LEADING K columns of a cell whose packed matrix has w_kbs k-blocks per row tile; the gate
 * products over the remaining columns come as `part` (B, 4H), written by an st_partial_product_job of an earlier launch, and are added to
 * the reduced gates.  16 < B <= 32, H / 4 even.  ref: nn.LSTMCell, src/module.py:275-280 */
int st_lstm_cell_packed_part_fwd(const float* packed_w, int w_kbs, const st_t16_view* x, int K, const float* part,
                                 const float* b_ih, const float* b_hh,
                                 const float* c_prev, int ldc_prev, const float* mask,
                                 const st_t16_view* h_dst0, const st_t16_view* h_dst1,
                                 float* c_out, int ldc, float* gates_out, int B, int H, void* stream);
/* The arguments of st_lstm_cell_packed_fwd as a struct, and two independent cells in ONE launch: under teacher forcing the decoder
 * cell of step t and the query cell of step t+1 both only wait for the attention of step t (src/module.py:216-288 with a teacher
 * frame as the next input).  Falls back to one launch per cell for shapes the 2-D tiled kernel does not take. */
typedef struct st_lstm_cell_packed_job {
    const float* packed_w; st_t16_view x; int K;
    const float* b_ih; const float* b_hh;
    const float* c_prev; int ldc_prev; const float* mask;
    st_t16_view h_dst0; st_t16_view h_dst1;      /* h_dst1.base may be NULL */
    float* c_out; int ldc; float* gates_out;
    const float* ada_std; const float* ada_mean; st_t16_view hadapt_dst;   /* hadapt_dst.base may be NULL */
    int B, H;
    const float* part; int w_kbs;   /* optional (as st_lstm_cell_packed_part_fwd): K covers the LEADING k-blocks of a matrix packed with w_kbs
                                     * k-blocks per tile; the gate products over the others arrive as the (B, 4 H) slab `part` */
} st_lstm_cell_packed_job;
int st_lstm_cell_packed_pair_fwd(const st_lstm_cell_packed_job* j0, const st_lstm_cell_packed_job* j1, void* stream);
/* st_skinny_linear_fwd on packed operands; output natural (y) and/or T16 (y_dst).
 * Optional third row range [n_split2, N): v = act2(v) * mask2(b, n - n_split2) -> y3_dst column
 * n - n_split2 (used to emit prenet layer 1 of the next step from the same launch as proj/gate). */
int st_skinny_linear_packed_fwd(const float* packed_w, const st_t16_view* x, int K,
                                const float* bias, int act, const float* mask, int ldmask,
                                float* y, int ldy, const st_t16_view* y_dst,
                                int n_split, float* y2, int ldy2, int rep,
                                int n_split2, int act2, const float* mask2, int ldmask2, const st_t16_view* y3_dst,
                                int B, int N, void* stream);

/* ------------------------------------------------------------------ location-sensitive attention
 * One decode step for the whole batch, one workgroup per utterance.
 * ref: Attention.energy/.forward src/module.py:371-407 and the state update :262-264,
 *      AdaIN :267-269 (std/mean hoisted out of the loop).
 *   loc   = W_loclin . conv1d(stack[w_prev, w_cum]; 2->F, k=K, pad (K-1)/2)      (L,A)
 *   e_l   = v . tanh(pq + loc_l + pm_l)
 *   w     = softmax_L(e)  (no mask)             -> w_out ; w_cum_out = w + w_cum_prev
 *   ctx   = sum_l w_l memory_l                  -> ctx (B,E)
 *   if h_q != NULL: h_adapt = ada_std * (h_q - ada_mean)                          (B,Q)
 * pq (B,A) = query_layer(h_q) is produced by st_skinny_linear_fwd. */
int st_attn_step_fwd(const float* pq, const float* pm, const float* memory,
                     const float* w_prev, int ld_wprev, const float* w_cum_prev,
                     float* w_out, int ld_wout, float* w_cum_out,
                     const float* loc_conv_w, const float* loc_lin_w, const float* v,
                     float* ctx, int ld_ctx,
                     const float* h_q, int ld_hq, const float* ada_std, const float* ada_mean,
                     float* h_adapt, int Q,
                     int B, int L, int A, int E, int F, int K, void* stream);
/* same step with the context written to up to 3 T16 destinations (and optionally natural); no
 * AdaIN (the decode loop fuses it into the query LSTM epilogue) */
int st_attn_step_t16_fwd(const float* pq, const float* pm, const float* memory,
                         const float* w_prev, int ld_wprev, const float* w_cum_prev,
                         float* w_out, int ld_wout, float* w_cum_out,
                         const float* loc_conv_w, const float* loc_lin_w, const float* v,
                         const st_t16_view* ctx_dst, int n_ctx_dst, float* ctx, int ld_ctx,
                         int B, int L, int A, int E, int F, int K, void* stream);
/* The same step in two launches.  `pre` only needs the PREVIOUS step's attention weights: location conv and
 * S(b,l,:) = pm(b,l,:) + W_l conv([w_prev; w_cum_prev])(l) -> s_buf (B,L,A); it can run while the rest of the decode step
 * does (st_skinny_linear_packed_attnpre_fwd runs it as extra workgroups of the proj launch).  `fin` = energies
 * v . tanh(pq + S), softmax, cumulative weights, context.  (pq + W_l cf) + pm becomes pq + (pm + W_l cf): fp32
 * re-association only. */
int st_attn_pre_fwd(const float* pm, const float* w_prev, int ld_wprev, const float* w_cum_prev,
                    const float* loc_conv_w, const float* loc_lin_w, float* s_buf, int parts,
                    int B, int L, int A, int F, int K, void* stream);   /* parts = workgroups per utterance: 1, 2 or 4 */
int st_attn_fin_t16_fwd(const float* pq, const float* s_buf, const float* memory, const float* w_cum_prev,
                        float* w_out, int ld_wout, float* w_cum_out, const float* v,
                        const st_t16_view* ctx_dst, int n_ctx_dst, float* ctx, int ld_ctx, int parts,
                        int B, int L, int A, int E, int F, int K, void* stream);
/* (parts = workgroups per utterance of the fin part: each takes E/parts context dims and repeats the energies + softmax;
 *  1, 2, 4 or 8 with E % (4*parts) == 0) */
/* st_skinny_linear_packed_fwd plus, as extra workgroups of the same launch (one per utterance), st_attn_pre_fwd for the NEXT
 * decode step: the proj (+) gate launch of step t leaves most compute units idle and the attention weights of step t are
 * already known, so S of step t+1 is ready when its attention launch starts. */
struct st_partial_product_job;
typedef struct st_attn_pre_job {
    const float* pm; const float* w_prev; int ld_wprev; const float* w_cum_prev;
    const float* loc_conv_w; const float* loc_lin_w; float* s_buf;
    int L, A, F, K;
    int parts;      /* workgroups per utterance (ranges of positions): 1, 2 or 4 */
    float* cf_out;  /* optional (B, L, F): the location features of the step, kept for the backward pass */
    /* optional (p2_packed_w != NULL): ONE MORE packed linear inside the same launch whose operand is the product's THIRD range
     * (act2 / mask2 applied) -- prenet layer 2 of the next decoder input behind proj (+) gate (+) prenet layer 1 (src/module.py:192,
     * :337-339).  The third range also leaves as 8-byte {value, tag = p2_epoch} granules, (B, p2_K) words in p2_gran (zeroed by the
     * caller before the first use of an epoch sequence), and (p2_N / 16) x batch-tile workgroups behind the attention ones wait for
     * them.  p2_K = N - n_split2, a multiple of 16, at most 512; every workgroup of the launch must be resident at once (checked
     * against the device's compute units).  A wait that does not complete sets bit 0 of *p2_status and leaves NaN. */
    const float* p2_packed_w; int p2_K, p2_N, p2_act; const float* p2_mask; int p2_ldmask;
    st_t16_view p2_dst;
    unsigned long long* p2_gran; unsigned p2_epoch; unsigned* p2_status;
    /* optional: a K-split partial product (st_partial_product_job, below) on the compute units the launch leaves idle -- teacher-forced
     * training hosts the tail of the decoder cell's gate reduction beside the query projection + attention pre part (B = 17..32; the
     * launch then has N / 16 x 2 + B x parts + job->N / 32 workgroups).  Not together with p2 */
    const struct st_partial_product_job* part;
} st_attn_pre_job;
int st_skinny_linear_packed_attnpre_fwd(const float* packed_w, const st_t16_view* x, int K,
                                        const float* bias, int act, const float* mask, int ldmask,
                                        float* y, int ldy, const st_t16_view* y_dst,
                                        int n_split, float* y2, int ldy2, int rep,
                                        int n_split2, int act2, const float* mask2, int ldmask2, const st_t16_view* y3_dst,
                                        int B, int N, const st_attn_pre_job* pre, void* stream);

/* st_skinny_linear_packed_fwd (no bias / activation, natural y) whose output columns [n0, n0 + H) are dh of an LSTM cell: the pointwise
 * half of that cell's backward step (st_lstm_cell_bwd_pointwise with dh0 = those columns of y) runs in the EPILOGUE of the product
 * instead of as a launch of its own.  The BPTT of the decode loop uses it twice per step: dgates_d(t) . [W_ih_d | W_hh_d] makes
 * dh_d(t-1) in its last D columns, and W_q^T dpq(t) makes the attention's share of dh_q(t).  Every row stride a multiple of 4, every
 * base 16-byte aligned, n0 a multiple of 16.  ref: backward of nn.LSTMCell src/module.py:228,:277 */
typedef struct st_lstm_pw_job {
    int n0, H;
    const float* dh1; int ld1;                        /* optional addend (B rows) */
    const float* dh2; int ld2; const float* scale2;   /* optional addend * scale2 (B, H) */
    const float* mask;                                /* optional (B, H) */
    const float* gates;                               /* (B, 4, H) activated (i, f, g, o) */
    const float* c; int ldc; const float* c_prev; int ldcp;
    float* dc;                                        /* (B, H) in: dL/dc_t, out: dL/dc_{t-1} */
    float* dgates; int ldg;                           /* (B, 4H) out */
    st_t16_view dgates_t16;                           /* optional second copy in T16 (K = 4H) */
    int dh1_slabs; long dh1_slab_stride;              /* > 1: dh1 is the first of that many slabs (dh1_slab_stride floats apart) of a K-split
                                                       * product (st_skinny_partial_attn_*): the addend is their sum, in slab order */
} st_lstm_pw_job;
int st_skinny_linear_packed_lstm_bwd_fwd(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                         const st_lstm_pw_job* job, void* stream);

/* The fin part for LONG texts: every utterance split over `parts` (2..64) ranges of positions -- local softmax statistics and an
 * un-normalised partial context per range (flash-decoding style), then a combine launch.  S and the memory rows are read once in
 * total; st_attn_fin_t16_fwd re-reads S in each of its context slices and one compute unit has to pull ~2 L A 4 bytes (12 us at
 * L = 171).  Same arithmetic as softmax(e) @ memory up to fp32 rounding (~1e-7).  workspace: B * parts * (4 + E) floats. */
size_t st_attn_fin_split_workspace_floats(int B, int E, int parts);
int st_attn_fin_split_fwd(const float* pq, const float* s_buf, const float* memory, const float* w_cum_prev,
                          float* w_out, int ld_wout, float* w_cum_out, const float* v,
                          const st_t16_view* ctx_dst, int n_ctx_dst, float* ctx, int ld_ctx, float* workspace, int parts,
                          int B, int L, int A, int E, void* stream);

/* Query projection + attention fin part in ONE launch: pq = W_q h_q (ref: src/module.py:380) is computed by the first workgroups
 * and handed to the fin workgroups of the same launch (st_attn_fin_t16_fwd's work: :389-406, :262-264) as 8-byte
 * {value, tag} words in `granules` ((B, A) 64-bit words, device memory, ZEROED by the caller before the first step of a
 * forward); `epoch` = decode step + 1 (never 0) is the tag this launch writes and waits for.  The fin workgroups request S, v
 * and the memory rows while pq is being computed: one kernel boundary and one exposed load round trip less per decode step.
 * Needs A % 16 == 0, A <= 256 and (A / 16) * ceil(B / 16) + B * parts workgroups resident at once (<= compute units); when that
 * does not hold the call fails and the caller uses st_skinny_linear_packed_fwd + st_attn_fin_t16_fwd. */
typedef struct st_attn_fin_job {
    const float* s_buf; const float* memory; const float* w_cum_prev;
    float* w_out; int ld_wout; float* w_cum_out; const float* v;
    st_t16_view ctx_dst[3]; int n_ctx_dst;
    int parts;          /* workgroups per utterance (slices of the context dims): 1, 2, 4, 8 */
    int L, A, E, F, K;
    unsigned* status;   /* optional device word: bit 0 is set when a fin workgroup gave up waiting for its granules (the query is
                         * then poisoned with NaN as well); never cleared by the library -- the caller zeroes and reads it */
} st_attn_fin_job;
int st_query_attn_fin_fwd(const float* packed_wq, const st_t16_view* h_q, int Q, unsigned long long* granules, unsigned epoch,
                          const st_attn_fin_job* job, int B, void* stream);
/* A partial product part (B, N) = x[:, kb0 .. kb0 + KB) W[:, kb0 .. kb0 + KB)^T over a k-block range of a P16 matrix (w_kbs k-blocks per row
 * tile; LSTM row order when the matrix is a cell's) and of a T16 activation buffer (x.kb0 = the range's first k-block in it): two row tiles
 * and both batch tiles per workgroup (16 < B <= 32, N % 32 == 0), N / 32 workgroups. */
typedef struct st_partial_product_job {
    const float* packed_w; int w_kbs; int kb0; int KB;
    st_t16_view x;
    int N;
    float* part;
} st_partial_product_job;
/* st_query_attn_fin_fwd with such a job beside it in the same launch (every workgroup on a compute unit of its own: (A / 16) ceil(B / 16) +
 * B * parts + N / 32 <= compute units).  The decode step's use: the part of nn.LSTMCell's gate product (ref: src/module.py:275-280) whose
 * operands do not depend on the attention of the step runs WHILE the attention runs. */
int st_query_attn_fin_part_fwd(const float* packed_wq, const st_t16_view* h_q, int Q, unsigned long long* granules, unsigned epoch,
                               const st_attn_fin_job* job, int B, const st_partial_product_job* part, void* stream);
/* ... and the job as a launch of its own: what the decode loop issues in its two-launch pq / fin form (a starved hand-off degrades to it),
 * so that both forms compute the same arithmetic bit for bit */
int st_partial_product_fwd(const st_partial_product_job* job, int B, void* stream);
/* Long texts (the fin part is bound by what ONE compute unit pulls in: S and the memory rows of an utterance): the same launch with the
 * fin part over `job->parts` (2..8) POSITION ranges per utterance -- local softmax statistics and an un-normalised partial context per
 * range, as st_attn_fin_split_fwd -- and the combine INSIDE the launch: the ranges of an utterance exchange (max, sum, partial context)
 * as 8-byte {value, tag} granules through `xchg` (st_attn_rng_xchg_words(B, E, parts) 64-bit words, zero before the first epoch of a
 * forward) and each finishes E / parts context dims and the weights of its own positions.  One launch instead of three.
 * Needs E % (4 * parts) == 0, E / parts <= 256 and every workgroup resident at once: st_query_attn_rng_fits(B, A, parts) != 0
 * (occupancy query of the kernel x compute units); otherwise use the two / three launch forms. */
int st_query_attn_rng_fits(int B, int A, int parts);
size_t st_attn_rng_xchg_words(int B, int E, int parts);
int st_query_attn_rng_fwd(const float* packed_wq, const st_t16_view* h_q, int Q, unsigned long long* granules,
                          unsigned long long* xchg, unsigned epoch, const st_attn_fin_job* job, int B, void* stream);
/* Diagnostics: one wave runs the consumer side of that hand-off on `granules` (A 64-bit {value, tag} words) as they are and writes
 * the A values to `out` -- or NaN, with bit 0 of *status set, after `max_spins` polls without every tag == epoch (the failure
 * path of a starved launch, which the scheduler never produces on an idle device). */
int st_handoff_wait_selftest(const unsigned long long* granules, unsigned epoch, int A, unsigned* status, float* out,
                             int max_spins, void* stream);

/* ------------------------------------------------------------------ dense GEMM / conv1d (many rows)
 * C(m, coff + n) = epilogue( sum_tap sum_ci A(row(m) * stride + tap - pad, ci) * W(n, ci, tap) )
 * channels-last activations: A is (Bn * Tin, Cin) with row stride lda, C is (Bn * Tout, N)
 * with row stride ldc; rows of one utterance never read across its [0,Tin) range (zero pad).
 * W is the torch Conv1d weight (N, Cin, KT) or, with KT == 1, the torch Linear weight (N, Cin).
 * epilogue: v += bias[n]; v = act_pre(v); if bn_mean: v = (v - bn_mean[n]) / sqrt(bn_var[n] + eps)
 *           * bn_w[n] + bn_b[n]; v = act_post(v); if highway_h: v = highway_h(m,n) * v +
 *           res(m,n) * (1 - v) else if res: v += res(m,n); if mask: v *= mask(m,n)
 * pool_prev != 0: A(row, ci) is replaced by max(A(row, ci), A(row - 1, ci)) (row-1 inside the
 * utterance), i.e. MaxPool1d(2, stride 1, padding 1)[:T] fused into the load.
 * ref: Conv1d+BatchNorm1d+ReLU src/module.py:421-431; BatchNormConv1d :527-538; CBHG bank /
 *      maxpool / projections / pre_highway / Highway :597-611, :541-555; nn.Linear src/tts.py:34;
 *      memory_layer src/module.py:306; prenet over the whole teacher :178-179; the strided / residual
 *      ConvLayer of the speech encoder src/module.py:627-648. */
typedef struct st_gemm_epilogue {
    const float* bias;
    int act_pre;
    const float* bn_mean; const float* bn_var; const float* bn_w; const float* bn_b; float bn_eps;
    int act_post;
    const float* res; int ldres;
    const float* highway_h; int ldhw;
    const float* mask; int ldmask;
    int w_tap_major;   /* KT > 1 only: W is given as (N, KT, Cin) -- taps outermost, so a k-block of weights is contiguous (16-byte
                        * loads) -- instead of the torch Conv1d layout (N, Cin, KT) */
    float* splitk_ws;  /* optional workspace of splitk_slabs * (Bn * Tout) * N floats: the reduction is split over splitk_slabs =
                        * st_gemm_splitk_slabs(...) (> 1) groups of workgroups whose partial products a finish pass adds in a
                        * fixed order before the epilogue (small grids with long reductions: the encoder convs) */
    int splitk_slabs;
} st_gemm_epilogue;
int st_gemm_splitk_slabs(int Bn, int Tout, int Cin, int N, int KT);   /* 1 = no split */

int st_gemm_fwd(const float* A, int lda, const float* W, float* C, int ldc, int coff,
                int Bn, int Tin, int Tout, int Cin, int N, int KT, int pad, int stride, int pool_prev,
                const st_gemm_epilogue* ep, void* stream);

/* Several independent st_gemm_fwd jobs in one launch: the K convolutions of the CBHG conv bank read the same input
 * (`[conv1d(x)[:, :, :T] for conv1d in self.conv1d_banks]`, src/module.py:590-598).  Each job holds the arguments of st_gemm_fwd
 * (the epilogue by value).  Jobs the pipelined kernel takes without pooling or split-K share launches (8 jobs each, longest
 * reduction first); otherwise the jobs run one after the other -- the results are those of n st_gemm_fwd calls either way. */
typedef struct st_gemm_job {
    const float* A; int lda; const float* W; float* C; int ldc; int coff;
    int Bn, Tin, Tout, Cin, N, KT, pad, stride, pool_prev;
    st_gemm_epilogue ep;
} st_gemm_job;
int st_gemm_fwd_batch(const st_gemm_job* jobs, int n, void* stream);

/* The highway stack of the CBHG in one launch (eval mode): n_layers times y = relu(W_H x + b_H) * T + x * (1 - T), T = sigmoid(W_T x + b_T),
 * x (M, C) rows; w_h / w_t: n_layers pointers to torch Linear weights (C, C), b_h / b_t: their biases (NULL entries = no bias).
 * Bit-identical to 2 n_layers st_gemm_fwd launches with the highway epilogue.  st_highway_stack_supported: C % 16 == 0, C <= 80, <= 8 layers.
 * ref: Highway.forward src/module.py:541-555, CBHG.forward :609-611. */
int st_highway_stack_supported(int C, int n_layers);
int st_highway_stack_fwd(const float* x, int ldx, const float* const* w_h, const float* const* b_h, const float* const* w_t,
                         const float* const* b_t, int n_layers, float* y, int ldy, int M, int C, void* stream);

/* per-column statistics over M rows (training-mode BatchNorm): mean, biased variance, and
 * the running-stat update  run = (1-mom) run + mom * {mean, unbiased var}; batches_tracked (optional, one int64 on the device:
 * nn.BatchNorm1d.num_batches_tracked) is incremented by the same launch.
 * ref: nn.BatchNorm1d in training mode, src/module.py:429,:531 */
int st_bn_stats(const float* X, int ldx, int coff, int M, int N, float* mean_out, float* var_out,
                float* run_mean, float* run_var, float momentum, long long* batches_tracked, float* ws, void* stream);
/* nn.LayerNorm over the last dimension of (M, N) rows: y = (x - mean) / sqrt(var + eps) * gamma + beta (biased variance, two passes
 * as ATen); mean_out / rstd_out (M) optional, kept for the backward.  ref: the speech encoder's `layer_norm` src/asr.py:38-39,58; the
 * normalised prenet Linear src/module.py:508-521 */
int st_layer_norm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* y, int ldy,
                      float* mean_out, float* rstd_out, int M, int N, void* stream);
/* dx of the above; dyxhat (optional, (M, N) contiguous) = dy * xhat, whose column sum is d gamma (d beta = column sum of dy) */
int st_layer_norm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* mean,
                      const float* rstd, float* dx, int lddx, float* dyxhat, int M, int N, void* stream);
/* log_softmax over the last dimension of (M, N) contiguous rows and its backward dx = dy - exp(y) * sum_n dy.
 * ref: ASRPostnet.forward src/asr.py:80 */
int st_log_softmax_fwd(const float* x, float* y, int M, int N, void* stream);
int st_log_softmax_bwd(const float* dy, const float* y, float* dx, int M, int N, void* stream);
/* One normalised prenet layer of a decode step after its Linear (the reference's Linear wrapper with norm_type, applied to the
 * (B, P) rows of ONE step: src/module.py:192,:508-521,:337-339): dst (T16) = relu(norm(y)) * mask.
 * mode 1: LayerNorm over the P columns; mode 2: BatchNorm1d with running statistics (eval); mode 3: BatchNorm1d with the
 * statistics of the step's rows, running statistics updated in place (momentum, unbiased variance), batches_tracked += 1.
 * Rows b0 .. B-1 are normalised and written (y, mask and dst are indexed by the absolute row): with a partial teacher the
 * reference's prenet sees the rows WITHOUT a teacher only (src/module.py:197-198,205-206), which is the batch of its BatchNorm1d. */
int st_prenet_norm_fwd(const float* y, int ldy, int mode, const float* gamma, const float* beta, float* run_mean,
                       float* run_var, long long* batches_tracked, float eps, float momentum, const float* mask, int ldmask,
                       const st_t16_view* dst, int b0, int B, int P, void* stream);
/* Backward of that layer, in place: dn (rows, P) = gradient at the norm's output (ReLU / mask backward applied) -> gradient at the
 * Linear's output y (rows, P); this step's sums are ADDED to dgamma / dbeta (P).  Statistics are recomputed from y (mode 2: the
 * running ones).  What torch autograd derives for nn.LayerNorm / nn.BatchNorm1d on a (rows, P) input. */
int st_prenet_norm_bwd(float* dn, int ld, const float* y, int ldy, int mode, const float* gamma, const float* run_mean,
                       const float* run_var, float eps, float* dgamma, float* dbeta, int rows, int P, void* stream);
/* X(m, coff+n) = act((X - mean) / sqrt(var + eps) * w + b), in place */
int st_bn_apply(float* X, int ldx, int coff, int M, int N, const float* mean, const float* var,
                const float* w, const float* b, float eps, int act, void* stream);

/* out-of-place variant of st_bn_apply (training keeps the pre-normalisation tensor for the backward) */
int st_bn_norm_fwd(const float* X, int ldx, int xoff, float* Y, int ldy, int yoff, int M, int N,
                   const float* mean, const float* var, const float* w, const float* b, float eps, int act,
                   void* stream);

/* ------------------------------------------------------------------ backward building blocks (training, H1)
 * Input gradient of st_gemm_fwd = st_gemm_fwd itself on dC with the weight transposed and tap-flipped
 * (W'(ci, n, t) = W(n, ci, KT-1-t), pad' = KT-1-pad, Tin/Tout swapped).
 * ref: what torch autograd derives for nn.Conv1d / nn.Linear / nn.BatchNorm1d / activations used at
 * src/module.py:421-431,:527-538,:541-555,:597-611 and src/tts.py:34. */
/* dW(n, ci, tap) (+)= sum_rows dC(row, dcoff+n) * A(row + tap - pad, ci)   [torch weight layout (N, Cin, KT)]
 * ws: st_gemm_wgrad_workspace_floats() floats of scratch (row-split partial slabs, added in fixed order) */
size_t st_gemm_wgrad_workspace_floats(int Bn, int Tout, int Cin, int N, int KT);
int st_gemm_wgrad(const float* dC, int lddc, int dcoff, const float* A, int lda, float* dW, float* ws,
                  int Bn, int Tin, int Tout, int Cin, int N, int KT, int pad, int pool_prev, int accumulate, void* stream);
/* st_gemm_wgrad + the bias gradient db[n] = sum over all (Bn * Tout) rows of dC[:, dcoff + n] in the same launches (no st_colsum
 * next to the product): backward of nn.Linear / nn.Conv1d with bias. */
int st_gemm_wgrad_db(const float* dC, int lddc, int dcoff, const float* A, int lda, float* dW, float* db, float* ws,
                     int Bn, int Tin, int Tout, int Cin, int N, int KT, int pad, int pool_prev, int accumulate, void* stream);
/* out(n) (+)= sum_m X(m, xoff+n) [* Y(m, yoff+n)]      (bias gradients, AdaIN statistics gradients) */
int st_colsum(const float* X, int ldx, int xoff, const float* Y, int ldy, int yoff, int M, int N,
              float* out, int accumulate, float* ws, void* stream);
/* scratch of the row-split column reductions st_bn_stats / st_colsum / st_bn_bwd (M rows, N columns) */
size_t st_colreduce_workspace_floats(int M, int N);
/* dpre = dout * mask * act'(out)   (out = the activated forward value; mask may be NULL) */
int st_act_bwd(const float* dout, int ldd, const float* out, int ldo, int act, const float* mask, int ldm,
               float* dpre, int ldp, int M, int N, void* stream);
/* BatchNorm backward with batch statistics, y = act((x - mean)/sqrt(var+eps)*w + b):
 * dx = w/sigma * (dyb - mean(dyb) - xhat * mean(dyb*xhat)), dw (+)= sum dyb*xhat, db (+)= sum dyb,
 * dyb = dy * act'(y).  ws: st_colreduce_workspace_floats(M, N) floats. */
int st_bn_bwd(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff, int act,
              const float* x, int ldx, int xoff, const float* mean, const float* var, const float* w, float eps,
              int M, int N, float* dx, int lddx, int dxoff, float* dw, float* db, int accumulate, float* ws, void* stream);
/* The two halves of st_bn_bwd, for data-parallel training with synchronised statistics: the caller all-reduces
 * s (2N floats: sum dyb, sum dyb*xhat over the local rows) across ranks between the calls and passes the global row
 * count as Mstat (mean / var must then be the global batch statistics too).  ws as for st_bn_bwd. */
int st_bn_bwd_reduce(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff, int act,
                     const float* x, int ldx, int xoff, const float* mean, const float* var, float eps,
                     int M, int N, float* s, float* ws, void* stream);
int st_bn_bwd_apply(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff, int act,
                    const float* x, int ldx, int xoff, const float* mean, const float* var, const float* w, float eps,
                    int M, int N, const float* s, int Mstat, float* dx, int lddx, int dxoff, void* stream);
/* ConvLayer's tail in training (ref: src/module.py:641-646: BatchNorm -> activation -> (+ x) -> dropout) as ONE launch behind the conv:
 * Tact (M, N) = act(BatchNorm(X)) with the given batch statistics, kept for the backward (NULL: not written), Y (M, N) = (Tact + res) * mask
 * (res, mask: NULL = absent).  Contiguous rows of N floats, N % 4 == 0, 16-byte aligned.  Replaces nn.BatchNorm1d + torch.tanh + `+` +
 * nn.Dropout of the reference (one launch instead of BatchNorm apply, add, bernoulli_, div_, mul). */
int st_bn_norm_res_mask_fwd(const float* X, float* Tact, float* Y, int M, int N, const float* mean, const float* var,
                            const float* w, const float* b, float eps, int act, const float* res, const float* mask, void* stream);
/* The backward of that tail in the two halves of st_bn_bwd_reduce / st_bn_bwd_apply: dy is multiplied by `mask` on the way in, `y` is the
 * kept Tact; the apply half also writes dres (M, N) = dy * mask, the gradient of the residual input (NULL: none); inv_total: NULL, or the
 * device scalar of st_bn_sync_merge (s then holds the sums over all ranks). */
int st_bn_bwd_reduce_masked(const float* dy, int ldd, const float* mask, int ldm, const float* y, int ldy, int act,
                            const float* x, int ldx, const float* mean, const float* var, float eps,
                            int M, int N, float* s, float* ws, void* stream);
int st_bn_bwd_apply_masked(const float* dy, int ldd, const float* mask, int ldm, const float* y, int ldy, int act,
                           const float* x, int ldx, const float* mean, const float* var, const float* w, float eps,
                           int M, int N, const float* s, int Mstat, const float* inv_total, float* dx, int lddx,
                           float* dres, int lddr, void* stream);
/* SyncBN (the data-parallel form of the BatchNorm1d layers above; the reference trains on one device, ref: src/module.py:434-455,
 * so this is the path's own multi-GPU extension) with the row counts kept on the device:
 *   st_bn_stats_record   rec (2N + 1) = (mean[N], M2[N], row count) of this rank's rows -- what the ranks all-gather;
 *   st_bn_sync_merge     the gathered records (world, 2N + 1) -> mean / biased variance of the global batch, merged in rank order
 *                        (identical on every rank), running statistics updated with the unbiased variance over the global count,
 *                        *inv_total_out = 1 / (global row count);
 *   st_bn_bwd_apply_sync st_bn_bwd_apply with s summed over all ranks and divided by the global count read from *inv_total. */
int st_bn_stats_record(const float* X, int ldx, int coff, int M, int N, float* rec, long long* batches_tracked, float* ws, void* stream);
int st_bn_sync_merge(const float* rec, int world, int N, float* mean_out, float* var_out, float* run_mean, float* run_var,
                     float momentum, float* inv_total_out, void* stream);
int st_bn_bwd_apply_sync(const float* dy, int ldd, int doff, const float* y, int ldy, int yoff, int act,
                         const float* x, int ldx, int xoff, const float* mean, const float* var, const float* w, float eps,
                         int M, int N, const float* s, const float* inv_total, float* dx, int lddx, int dxoff, void* stream);
/* Highway combine y = H*T + x*(1-T) and its backward dH = dy*T, dT = dy*(H-x), dx_direct = dy*(1-T)
 * ref: src/module.py:551-554 */
int st_highway_fwd(const float* H, const float* Tgate, const float* x, float* y, size_t total, void* stream);
/* One Highway layer on the PRE-activations of its two Linear layers side by side, ht (M, 2C) = [W_H x + b_H | W_T x + b_T] (one
 * N = 2C product instead of two): y = relu(h) * sigmoid(t) + x * (1 - sigmoid(t)) (src/module.py:551-554), and its backward:
 * dht (M, 2C) = gradients at the two pre-activations (the operand of ONE input-gradient product and ONE weight-gradient product),
 * dx_direct (M, C) = dy * (1 - T). */
/* The K BatchNorm1d layers of a conv bank (K independent layers of equal width N over K tensors of (Bn, T_k, N) rows, src/module.py:590-598)
 * as one launch per phase.  Segment k: x (Bn * T_k rows, stride ldx), statistics over ALL its rows (the reference normalises the T + 1
 * positions of an even-k conv before trimming to T); the normalised rows t < Tout go to columns [k N, (k + 1) N) of Y (Bn * Tout rows, stride
 * ldy) -- the concatenated bank, no torch.cat.  Backward: dY (same geometry) -> dx_k over all rows of each segment (rows t >= Tout see
 * dy = 0), optionally through the ReLU in FRONT of the norm (relu_in: x_k = relu(conv), so the factor is x > 0), and (s1, s2) = (d bias, d weight).
 * ws: st_bn_bank_workspace_floats(nseg, max rows, N). */
#define ST_BN_BANK_MAX 16
typedef struct st_bn_bank_seg {
    const float* x; int ldx; int T;                /* rows b * T + t, b < Bn */
    const float* w; const float* b;                /* weight, bias (N) or NULL */
    float* run_mean; float* run_var; long long* batches_tracked;   /* updated in place by st_bn_bank_fwd (NULL = none) */
    float momentum, eps;
    float* mean; float* var;                       /* (N) out of st_bn_bank_fwd, in of st_bn_bank_bwd */
    float* dx; int lddx;                           /* st_bn_bank_bwd: out, Bn * T rows */
    float* sums;                                   /* st_bn_bank_bwd: out (2, N) = (sum dy, sum dy xhat) = (d bias, d weight) */
} st_bn_bank_seg;
size_t st_bn_bank_workspace_floats(int nseg, int max_rows, int N);
int st_bn_bank_fwd(const st_bn_bank_seg* segs, int nseg, int Bn, int N, float* Y, int ldy, int Tout, float* ws, void* stream);
int st_bn_bank_bwd(const st_bn_bank_seg* segs, int nseg, int Bn, int N, const float* dY, int lddy, int Tout, int relu_in, float* ws,
                   void* stream);
/* The same in SyncBN stages (data parallel; the collectives between the stages are the caller's -- ONE all-gather of the records and ONE
 * all-reduce of the sums for the whole bank): st_bn_bank_stats_record writes (mean, M2, row count) of this rank's rows of every segment
 * to rec (nseg, 2N + 1) and counts the batch; st_bn_bank_sync_merge turns the gathered allrec (world, nseg, 2N + 1) into the global mean /
 * var of every segment (running statistics updated) and inv_total (nseg) = 1 / global row count; st_bn_bank_norm normalises.  Backward:
 * st_bn_bank_bwd_reduce leaves this rank's sums in segs[k].sums; st_bn_bank_bwd_apply takes the all-reduced sums there and inv_total. */
int st_bn_bank_stats_record(const st_bn_bank_seg* segs, int nseg, int Bn, int N, float* rec, float* ws, void* stream);
int st_bn_bank_sync_merge(const st_bn_bank_seg* segs, int nseg, int Bn, int N, const float* allrec, int world, float* inv_total, void* stream);
int st_bn_bank_norm(const st_bn_bank_seg* segs, int nseg, int Bn, int N, float* Y, int ldy, int Tout, void* stream);
int st_bn_bank_bwd_reduce(const st_bn_bank_seg* segs, int nseg, int Bn, int N, const float* dY, int lddy, int Tout, float* ws, void* stream);
int st_bn_bank_bwd_apply(const st_bn_bank_seg* segs, int nseg, int Bn, int N, const float* dY, int lddy, int Tout, int relu_in,
                         const float* inv_total, void* stream);
int st_highway_ht_fwd(const float* ht, const float* x, float* y, int M, int C, void* stream);
int st_highway_ht_bwd(const float* dy, const float* ht, const float* x, float* dht, float* dx_direct, int M, int C, void* stream);
int st_highway_bwd(const float* dy, const float* H, const float* x, const float* Tgate,
                   float* dH, float* dT, float* dx_direct, size_t total, void* stream);
/* y(b,t,:) = max(x(b,t-1,:), x(b,t,:)), y(b,0,:) = x(b,0,:): nn.MaxPool1d(2, stride=1, padding=1)(x)[:, :, :T] of the CBHG,
 * src/module.py:600, on contiguous channels-last (Bn, T, C) tensors, C % 4 == 0 (the input of the LDS-DMA conv kernel; the
 * register-staged kernel fuses the same maximum into its loads: pool_prev of st_gemm_fwd) */
int st_pool_prev_fwd(const float* x, float* y, int Bn, int T, int C, void* stream);
/* backward of the fused MaxPool1d(2,1,1)[:T]: dx(b,t,c) from the gradient w.r.t. the pooled tensor */
int st_pool_prev_bwd(const float* dy_pooled, const float* x, float* dx, int Bn, int T, int C, void* stream);
/* dst(b,t,:) (+)= src(b,t,:) with arbitrary (b,t) strides (elements) */
int st_copy3d(float* dst, long dst_sb, long dst_st, const float* src, long src_sb, long src_st,
              int Bn, int T, int C, int accumulate, void* stream);
/* dtable(v, :) += sum of dout(r, :) over the rows with idx(r) == v, in row order (deterministic, no atomics)
 * ref: backward of F.embedding src/embed.py:97-101 */
int st_scatter_add_rows(const float* dout, const int64_t* idx, float* dtable, int n, int D, int V, void* stream);

/* ------------------------------------------------------------------ recurrent sequence layers */
/* One direction of nn.LSTM over a full sequence: xproj (B,T,4H) = x W_ih^T + b_ih
 * (from st_gemm_fwd; b_hh is added in the cell), out(b, t, ocol : ocol+H) = h_t.  Workspace ws: 3*B*H floats.
 * Training: gates_tape (T,B,4,H) receives the activated gates and c_tape (T,B,H) every cell state (both
 * indexed by time step); NULL in inference.
 * ref: nn.LSTM src/module.py:432-438,:458-460 (lengths ignored, zero initial state). */
int st_lstm_seq_fwd(const float* xproj, const float* w_hh, const float* b_hh, float* out, int ldo, int ocol,
                    float* ws, float* gates_tape, float* c_tape, int B, int T, int H, int reverse, void* stream);
/* Pointwise half of one LSTM cell backward step (shared by the encoder BiLSTM and the decoder's two cells):
 * dh = (dh0 + dh1 + dh2 * scale2) * mask;  from dh and the carried dc (B,H, updated in place to dL/dc_{t-1})
 * to dgates (B,4H) w.r.t. the pre-activation gates in torch order (i,f,g,o).  dh1, dh2, scale2, mask, c_prev
 * may be NULL; dgates_t16 (optional): a second copy of dgates in the T16 tile layout (K = 4H), the operand of
 * st_skinny_linear_packed_fwd over the packed W^T.  ref: backward of nn.LSTMCell src/module.py:228,:277 */
int st_lstm_cell_bwd_pointwise(const float* dh0, int ld0, const float* dh1, int ld1, const float* dh2, int ld2,
                               const float* scale2, const float* mask, const float* gates, const float* c, int ldc,
                               const float* c_prev, int ldcp, float* dc, float* dgates, int ldg,
                               const st_t16_view* dgates_t16, int B, int H, void* stream);
/* Both directions of nn.LSTM(bidirectional=True) in one pass: per time step ONE launch advances direction 0 at t = s and direction
 * 1 at t = T-1-s (st_lstm_cell_pair_fwd: two cells of the same shape, blockIdx.z picks the job).  Arguments as st_lstm_seq_fwd in
 * arrays of two; ws: 6*B*H floats.  ref: nn.LSTM src/module.py:432-438,458-460 */
int st_lstm_cell_pair_fwd(const st_seg* segs2, const float* const* b_hh2, const float* const* pre2, int ldpre,
                          const float* const* c_prev2, int ldc_prev, float* const* h_out2, int ldh,
                          float* const* c_out2, int ldc, float* const* gates_out2, int B, int H, void* stream);
int st_lstm_seq2_fwd(const float* const* xproj2, const float* const* w_hh2, const float* const* b_hh2, float* out, int ldo,
                     const int* ocol2, float* ws, float* const* gates_tape2, float* const* c_tape2,
                     int B, int T, int H, void* stream);
/* The same layer as ONE launch for all T steps (H % 64 == 0, H <= 512, B <= 64, 2 * H/4 workgroups resident at once): the recurrent
 * weights and the cell states stay in registers, h_{t-1} is polled straight out of `out`, which the entry first fills with a sentinel
 * (every h is written exactly once: the data is the flag).  *status (optional device word): bit 1 set when a wait timed out -- the launch
 * was starved of compute units, the affected rows are NaN.  st_lstm_seq2_persist_supported tells whether a shape is taken.
 * ref: nn.LSTM src/module.py:432-438,458-460 */
int st_lstm_seq2_persist_supported(int B, int T, int H, int ldo, int ocol0, int ocol1);
int st_lstm_seq2_persist_fwd(const float* const* xproj2, const float* const* w_hh2, const float* const* b_hh2, float* out, int ldo,
                             const int* ocol2, float* const* gates_tape2, float* const* c_tape2,
                             int B, int T, int H, unsigned* status, void* stream);
/* backward of both directions at once (see st_lstm_seq_bwd; arrays of two, ws: 4*B*H floats): one pointwise launch and one
 * W_hh^T launch per time step for the two directions */
int st_lstm_seq2_bwd(const float* dout, int ldd, const int* dcol2, const float* const* gates_tape2, const float* const* c_tape2,
                     const float* const* w_hh_t2, float* const* dxproj2, float* ws, int B, int T, int H, void* stream);
/* The same loop as ONE launch for all T steps (H % 32 == 0, H <= 256, 2 * H/16 * ceil(B/16) workgroups resident at once): the
 * 4H x 16 slices of W_hh (the plain (4H, H) parameters, no transposed copy) and the carried dL/dc stay in registers, the gate gradients
 * of the step before are polled straight out of dxproj2[d] (B, T, 4H), which the entry first fills with a sentinel (every word is
 * written exactly once).  *status: as st_lstm_seq2_persist_fwd.  ref: backward of nn.LSTM src/module.py:432-438,458-460 */
int st_lstm_seq2_bwd_persist_supported(int B, int T, int H, int ldd, int dcol0, int dcol1);
int st_lstm_seq2_bwd_persist(const float* dout, int ldd, const int* dcol2, const float* const* gates_tape2, const float* const* c_tape2,
                             const float* const* w_hh2, float* const* dxproj2, int B, int T, int H, unsigned* status, void* stream);
int st_skinny_linear_pair_fwd(const st_seg* segs2, float* const* y2, int ldy, int B, int N, void* stream);
/* st_lstm_seq2_bwd on packed operands: ONE launch per time step for the two directions -- dgates(s) . W_hh with the pointwise backward of
 * step s-1 in its epilogue (st_skinny_linear_packed_lstm_bwd_pair_fwd).  w_hh_t_p16_2: st_pack_weight_t of each direction's W_hh (N = H,
 * K = 4H); dg_t16_ws: 4 * st_t16_floats(B, 4H) floats (two ping-pong T16 copies of dgates per direction).  ws: 2*B*H floats (dc carries). */
int st_lstm_seq2_bwd_packed(const float* dout, int ldd, const int* dcol2, const float* const* gates_tape2, const float* const* c_tape2,
                            const float* const* w_hh_t_p16_2, float* const* dxproj2, float* ws, float* dg_t16_ws,
                            int B, int T, int H, void* stream);
/* Backward through time of st_lstm_seq_fwd: dout(b, t, dcol : dcol+H) -> dxproj (B,T,4H) (= gradient of the
 * input projection incl. both biases).  w_hh_t = W_hh^T (H, 4H).  ws: 2*B*H floats.  The caller finishes with
 * dW_hh = st_gemm_wgrad(dxproj, out shifted by one step) and db = st_colsum(dxproj). */
int st_lstm_seq_bwd(const float* dout, int ldd, int dcol, const float* gates_tape, const float* c_tape,
                    const float* w_hh_t, float* dxproj, float* ws, int B, int T, int H, int reverse, void* stream);
/* nn.GRU, ndir directions in one launch (direction 1 runs time-reversed), one workgroup per
 * (utterance, direction) keeps W_hh rows in registers and h in LDS for all T steps.
 * gi[d] (B,T,3H) = x W_ih[d]^T + b_ih[d] (from st_gemm_fwd); out(b, t, d*H : (d+1)*H) = h_t.
 * Training: tape (ndir,B,T,4,H) receives (r, z, n, W_hn h + b_hn) per step; NULL in inference.
 * ref: nn.GRU src/module.py:585-586,:617 */
int st_gru_seq_fwd(const float* gi_fwd, const float* gi_bwd, const float* w_hh_fwd, const float* w_hh_bwd,
                   const float* b_hh_fwd, const float* b_hh_bwd, float* out, int ldo, float* tape,
                   int B, int T, int H, int ndir, void* stream);
/* Backward through time of st_gru_seq_fwd, again one workgroup per (utterance, direction) with W_hh^T columns
 * in registers: dout (B,T,>=ndir*H) -> dgi[d] (B,T,3H) (gradient of gi) and dgh[d] (B,T,3H) (gradient of
 * W_hh h + b_hh).  The caller finishes with dW_hh = st_gemm_wgrad(dgh, out shifted by one step), db_hh = colsum. */
int st_gru_seq_bwd(const float* dout, int ldd, const float* out, int ldo, const float* tape,
                   const float* w_hh_fwd, const float* w_hh_bwd, float* dgi_fwd, float* dgi_bwd,
                   float* dgh_fwd, float* dgh_bwd, int B, int T, int H, int ndir, void* stream);

/* ------------------------------------------------------------------ VQ codebook */
/* table(v, :) = cat[learnable(v, 0:Dl), attr(v,:) W_attr^T + b_attr]      (V, Dl + Da)
 * ref: L2Embedding.embedding / forward src/embed.py:87-94,109-112 */
int st_vq_build_table(const float* learnable, int Dl, const float* attr, int n_attr,
                      const float* attr_w, const float* attr_b, int Da, float* table, int V, void* stream);
/* out(n, :) = table(idx(n), :)   ref: F.embedding src/embed.py:97-101,:134,:181-183 */
int st_gather_rows(const float* table, const int64_t* idx, float* out, int n, int D, int V, void* stream);
/* Nearest-code search.  x (n, D), table (V, D):
 *   sim = relu(temp) * -(|x|^2 + |e|^2 - 2 x.e) ; p = softmax_V(sim) ; idx = argmax_V(p)
 *   (first maximum wins) ; out = (x + table[idx]) - x   (straight-through forward value)
 * ref: L2Embedding.forward src/embed.py:105-147, neg_batch_l2 :208-213 */
int st_vq_l2_fwd(const float* x, const float* table, const float* temp, float* p_code,
                 int64_t* idx, float* out, float* workspace, int n, int D, int V, void* stream);
/* The same search with the table's MFMA-order copy made ONCE per table version instead of on every call (the table only changes
 * with the weights): st_vq_pack_table fills `packed` (st_vq_l2_workspace_floats(D, V) floats; D <= 64, D % 4 == 0, V <= 1024),
 * st_vq_l2_packed_fwd searches with it.  Same results as st_vq_l2_fwd bit for bit. */
int st_vq_pack_table(const float* table, float* packed, int D, int V, void* stream);
int st_vq_l2_packed_fwd(const float* x, const float* table, const float* packed, const float* temp, float* p_code,
                        int64_t* idx, float* out, int n, int D, int V, void* stream);
/* workspace: st_vq_l2_workspace_floats(D, V) floats (the table re-packed into MFMA operand order + |e|^2 per code) selects the
 * matrix-core kernel (D <= 64, D % 4 == 0, V <= 1024; exact-fp32 MFMA keeps the dimension-ascending dot product, so the indices
 * are those of the scalar kernel); NULL or another shape runs the scalar kernel (table staged in LDS). */
size_t st_vq_l2_workspace_floats(int D, int V);
/* Backward pieces of the codebook lookup (what autograd derives for src/embed.py:105-147 and :187-205):
 *   st_softmax_bwd:       dz(n, v) = s * p(n, v) * (dp(n, v) - sum_v' dp(n, v') p(n, v')),  s = scale * (relu_scale ? max(relu_scale[0], 0) : 1);
 *                         rowsum (optional) = sum_v dz(n, v)
 *   st_rowscale_combine:  out(m, d) = alpha * a(m, d) + beta * r(m) * x(m, d) + c(m, d)   (x/r and c optional) -- with a = dz E or
 *                         dz^T X from the GEMM entry points this forms dX = 2 (dz E) - 2 X rowsum + dnew_latent and
 *                         dE = 2 (dz^T X) - 2 E colsum + scatter_add(dnew_latent) of sim = -relu(temp) (|x|^2 + |e|^2 - 2 x.e). */
int st_softmax_bwd(const float* p, const float* dp, const float* relu_scale, float scale, float* dz, float* rowsum,
                   int n, int V, void* stream);
int st_rowscale_combine(const float* a, float alpha, const float* x, const float* r, float beta, const float* c,
                        float* out, int M, int D, void* stream);
/* Run-length merge of VQ codes with blank filtering (ref: VQVAE.mean_forward src/vqvae.py:218-257): per utterance,
 * consecutive frames with the same argmax code (runs capped at max_frames_per_phn+1 frames) are replaced by the
 * mean of their latents, runs of code 0 are dropped.  out (B, T, D) must be ZERO on entry (rows >= lens(b) stay
 * zero = the reference's pad_sequence); lens (B) int32 = kept segments per utterance (0 = all-blank utterance:
 * the reference returns None for the batch); frame_seg (B,T) int32 / frame_w (B,T) = segment of every frame
 * (-1 = dropped) and 1/len, the saved tensors of st_vq_mean_bwd. */
int st_vq_mean_fwd(const float* p_code, const float* latent, float* out, int* lens, int* frame_seg, float* frame_w,
                   int B, int T, int D, int V, int max_frames_per_phn, void* stream);
/* dlatent(b,t,:) = dout(b, seg(t), :) / len(seg(t)); dout is (B, n_seg_rows, D) */
int st_vq_mean_bwd(const float* dout, int n_seg_rows, const int* frame_seg, const float* frame_w, float* dlatent,
                   int B, int T, int D, void* stream);
/* softmax + argmax over rows of logits (n, V) -> p (n, V), idx (n)
 * ref: SeperateEmbedding.forward src/embed.py:190-193 */
int st_softmax_argmax(const float* logits, float* p, int64_t* idx, int n, int V, void* stream);

/* ------------------------------------------------------------------ the decode loop
 * Decoder.forward's `for t in range(decode_steps)` (src/module.py:184-206) issued natively:
 * per step  query LSTM -> query projection -> attention (+AdaIN) -> decoder LSTM -> proj/gate
 * [-> prenet of the own output when the next input is not a teacher frame].
 * All per-step state lives in caller-owned "tapes" indexed by step (slot 0 = initial zeros),
 * which double as the saved tensors of the backward pass.
 */
typedef struct st_decoder_weights {
    const float* prenet_w0;      /* decoder.prenet.layers.0.linear.weight (P, r*n_mels)   */
    const float* prenet_w1;      /* decoder.prenet.layers.1.linear.weight (P, P)          */
    const float* q_w_ih;         /* decoder.query_rnn.weight_ih (4Q, P+E)                  */
    const float* q_w_hh;         /* decoder.query_rnn.weight_hh (4Q, Q)                    */
    const float* q_b_ih; const float* q_b_hh;
    const float* attn_query_w;   /* decoder.attn.query_layer.linear.weight (A, Q)          */
    const float* attn_v;         /* decoder.attn.v.linear.weight (1, A)                    */
    const float* attn_loc_conv_w;/* decoder.attn.loc_conv.conv.weight (F, 2, K)            */
    const float* attn_loc_lin_w; /* decoder.attn.loc_linear.linear.weight (A, F)           */
    const float* d_w_ih;         /* decoder.dec_rnn.weight_ih (4D, E+Q)                    */
    const float* d_w_hh;         /* decoder.dec_rnn.weight_hh (4D, D)                      */
    const float* d_b_ih; const float* d_b_hh;
    const float* projgate_w;     /* cat[proj.linear.weight; gate_layer.linear.weight] (r*n_mels+1, D+E) */
    const float* projgate_b;     /* cat[proj.linear.bias; gate_layer.linear.bias]                      */
    /* normalised prenet (st_decoder_dims.prenet_norm != 0; the reference's Linear wrapper with norm_type, src/module.py:508-521):
     * per layer the norm's weight / bias and, for BatchNorm1d, its running statistics and batch counter */
    const float* pre_norm_w[2]; const float* pre_norm_b[2];
    float* pre_norm_rm[2]; float* pre_norm_rv[2]; long long* pre_norm_nbt[2];
    float pre_norm_eps, pre_norm_momentum;
} st_decoder_weights;

typedef struct st_decoder_dims {
    int B, L, E, n_mels, r, P, Q, D, A, F, K;
    /* fuse_pre0 != 0 (free-running inference only: every next input is prenet(own output)):
     * projgate_w/_b carry P extra rows  W_pre0 . W_proj  /  W_pre0 . b_proj, so the proj/gate launch
     * also emits relu(prenet layer 1) of the next step (same function, re-associated in fp32). */
    int fuse_pre0;
    /* 0: plain prenet.  1 LayerNorm / 2 BatchNorm1d (eval) / 3 BatchNorm1d (training: statistics of a step's B rows) between
     * each prenet Linear and its ReLU (st_prenet_norm_fwd): the own-output prenet of the loop then runs as Linear, norm launch,
     * Linear, norm launch; fuse_pre0 must be 0 and st_decoder_io.pre_nat given */
    int prenet_norm;
} st_decoder_dims;

typedef struct st_decoder_io {
    const float* memory;      /* (B, L, E)  encoder output                                  */
    const float* pm;          /* (B, L, A)  memory_layer(memory)        src/module.py:306   */
    const float* ada_std;     /* (B, Q) relu(pseudo_latent_std(spkr))   src/module.py:268   */
    const float* ada_mean;    /* (B, Q) pseudo_latent_mean(spkr)        src/module.py:269   */
    /* next-input policy, decided by the host exactly as src/module.py:190-206:
     * step_src[t] (HOST array, `steps` entries) tells where dec_in of step t+1 comes from:
     *   -1 = prenet(own output) for every row; -2 = teacher mean (rows < Bt) ; >= 0 = teacher
     *   frame index `take_frame` (rows < Bt); rows >= Bt always use prenet(own output).   */
    const int* step_src;
    const float* teacher_pre; /* (Bt, Tt, P) prenet(teacher) or NULL     src/module.py:178-179 */
    const float* teacher_mean;/* (Bt, P) teacher_pre.mean(1) or NULL     src/module.py:194   */
    int Bt, Tt;
    const float* prenet_mask; /* (steps, 2, B, P) scaled masks for prenet(own output) after step t, or NULL */
    const float* q_mask;      /* (steps, B, Q) scaled hidden-state dropout masks or NULL   */
    const float* d_mask;      /* (steps, B, D) */
    int steps;
    /* outputs */
    float* mel_out;           /* (B, steps*r, n_mels) */
    float* align_out;         /* (B, steps, L)        */
    float* stop_out;          /* (B, steps*r)         */
    /* packed weights: st_decoder_packed_floats() floats, filled by st_decoder_pack() */
    const float* packed;
    /* tapes.  The step inputs live in three T16 buffers per step, handed in ZERO-FILLED:
     *   xq_tape: steps+1 slots of T16 (B, P+E+Q)  slot t = [dec_in_t | ctx_{t-1} | h_q_{t-1}]
     *   xd_tape: steps+1 slots of T16 (B, E+Q+D)  slot t = [ctx_t | adapted h_q_t | h_d_{t-1}]
     *   xo_tape: steps   slots of T16 (B, D+E)    slot t = [h_d_t | ctx_t]
     * (each logical piece padded to a multiple of 16 columns; slot size = st_t16_floats(B, padded sum)).
     * cq/cd/wcum are natural (steps+1, B, dim); slot 0 = initial zero state. */
    float* xq_tape; float* xd_tape; float* xo_tape;
    float* cq_tape; float* cd_tape; float* wcum_tape;
    float* pq_buf;            /* (B, A) scratch, natural */
    float* pre1_t16;          /* T16 (B,P) scratch (prenet layer-1 output), zero-filled */
    float* mel_t16;           /* T16 (B, r*n_mels) scratch (own output as the prenet input), zero-filled */
    float* zero_row;          /* (B, max(L,1)) zeros (w_prev of step 0) -- zeroed by the callee */
    float* gates_q_tape;      /* (steps, B, 4, Q) or NULL (training) */
    float* gates_d_tape;      /* (steps, B, 4, D) or NULL */
    float* attn_s_buf;        /* (B,L,A) or NULL: split the attention step -- its location conv + W_l part for step t+1 runs inside
                               * the proj (+) gate launch of step t (st_skinny_linear_packed_attnpre_fwd), the attention launch
                               * itself starts from S (free-running fused-prenet inference) */
    int attn_pre_parts;       /* workgroups per utterance of the pre part (1, 2, 4) */
    int attn_fin_parts;       /* workgroups per utterance of the fin part (0/1, 2, 4, 8; E % (4*parts) == 0) */
    int defer_proj;           /* pure teacher forcing only: skip the per-step proj (+) gate launch; the caller computes mel /
                               * stop for all steps with one GEMM over xo_tape afterwards (mel_out / stop_out untouched) */
    int pre1_step_floats;     /* > 0: pre1_t16 is a tape of `steps` slots of that many floats (training keeps the prenet
                               * layer-1 output of every own-output feedback for the backward); 0: one scratch slot */
    int attn_s_step_floats;   /* > 0: attn_s_buf is a tape, slot t = S of step t (B*L*A floats apart), kept for the backward */
    float* attn_loc_tape;     /* optional (steps, B, L, F): location features of every step (slot 0, which no step writes, is zeroed by the callee) */
    float* attn_split_ws;     /* optional, st_attn_fin_split_workspace_floats(B, E, attn_split_parts) floats: long texts run the fin part
                               * split over attn_split_parts position ranges + a combine launch (st_attn_fin_split_fwd) */
    int attn_split_parts;
    unsigned long long* pq_granules;   /* optional (B, A) 64-bit words: with attn_s_buf, the query projection and the attention fin
                                        * part of a step run as ONE launch (st_query_attn_fin_fwd); zeroed by the callee per forward */
    const float* dec_in0;              /* optional (B, P) natural: the input of step 0 = prenet(go frame).  NULL = zeros, which is what a
                                        * plain prenet makes of the all-zero go frame (src/module.py:161,183); a normalised prenet does not */
    float* pre_nat;                    /* (B, P) scratch, natural layout: the Linear output of a normalised prenet layer (prenet_norm != 0) */
    unsigned long long* attn_xchg;     /* optional, st_attn_rng_xchg_words(B, E, attn_split_parts) 64-bit words: with pq_granules and
                                        * attn_split_parts in 2..8, long texts run query projection + fin part over position ranges +
                                        * combine as ONE launch per step (st_query_attn_rng_fwd) when it fits the device; zeroed by the
                                        * callee per forward */
    unsigned* handoff_status;          /* optional device word for pq_granules: bit 0 = an in-launch hand-off timed out (a starved launch:
                                        * the waiting workgroups were not co-resident with their producers).  Sticky; owned, zeroed and
                                        * read by the caller after the forward -- a time-out is an ERROR, not a NaN to find later */
    int pair_cells;                    /* != 0 (with defer_proj, i.e. pure teacher forcing): the decoder cell of step t and the query cell of
                                        * step t+1 share one launch (st_lstm_cell_packed_pair_fwd) -- neither needs the other's output */
    float* pre_nat_tape;               /* optional (steps, 2, B, P): with prenet_norm, the Linear outputs of both prenet layers of every
                                        * own-output feedback are kept here (instead of pre_nat) for the backward */
    unsigned long long* pre1_granules; /* optional (B, P) 64-bit words: with fuse_pre0 and the split attention step, prenet layer 2 of the next
                                        * input runs INSIDE the proj (+) gate (+) prenet-layer-1 launch (st_attn_pre_job.p2_*), its operand
                                        * handed over as granules -- one launch less per free-running decode step.  Zeroed by the callee per
                                        * forward; time-outs go to handoff_status */
    float* gate_part;                  /* optional (B, 4 D) scratch: in the one-launch pq + fin form (pq_granules) with 16 < B <= 32 the decoder
                                        * cell's gate products over operands that are known BEFORE the attention runs -- the tail of the cell's
                                        * reduction [ctx | AdaIN(h_q(t)) | h_d(t-1)] from column gate_part_k on -- ride beside the pq / fin launch
                                        * on compute units it leaves idle (st_query_attn_fin_part_fwd); the cell launch then reduces the first
                                        * gate_part_k columns and adds this slab (st_lstm_cell_packed_part_fwd).  fp32 re-association only.
                                        * NULL = off */
    int gate_part_k;                   /* the cell's own share of the reduction: a multiple of 16 in [kb16(E) * 16, K_d); 0 = the callee's rule
                                        * (st_decoder_gate_split_k: half of the reduction in whole rounds of 16 k-blocks -- C2: 1280 of 2560,
                                        * measured best of 512 ... 2048) */
} st_decoder_io;

/* the default of st_decoder_io.gate_part_k for these dimensions */
int st_decoder_gate_split_k(const st_decoder_dims* d);
size_t st_decoder_packed_floats(const st_decoder_dims* d);
/* floats of ONE slot of xq_tape (which = 0), xd_tape (1), xo_tape (2) */
size_t st_decoder_tape_floats(const st_decoder_dims* d, int which);
/* pack the six matrices the loop streams every step (once per forward / per weight update) */
int st_decoder_pack(const st_decoder_weights* w, const st_decoder_dims* d, float* packed, void* stream);
int st_decoder_forward(const st_decoder_weights* w, const st_decoder_dims* d, const st_decoder_io* io,
                       void* stream);

/* ------------------------------------------------------------------ decoder backward (training, teacher forcing)
 * ref: what torch autograd derives for Decoder.forward / decode_one_step, src/module.py:184-288 */
/* Backward of one attention step (recomputes the location features and tanh from the saved weights).
 * dctx / dw_direct: up to three addends each (row strides ld_*); dcum (B,L) carries dL/dcum_t across steps
 * (+ dcum_add).  Outputs needed by the recurrence: dpq (B,A) and dhist (B,2,L) = gradient w.r.t.
 * [w_{t-1}; cum_{t-1}] through the location conv.  Everything that is a sum over steps is left to the caller:
 * the kernel writes this step's slices ds_t (B,L,A), loc_t / dloc_t (B,L,F), hist_t (B,L,2) channels-last,
 * dctx_t (B,E), dv_t (B,A), from which dpm = sum_t ds, dW_l = ds^T loc, dW_c = conv weight gradient of
 * (dloc, hist), dmem[b] = w^T dctx, dv = sum dv_t. */
int st_attn_step_bwd(const float* pq, const float* pm, const float* memory,
                     const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                     const float* loc_conv_w, const float* loc_lin_w, const float* v,
                     const float* const* dctx, const int* ld_dctx, int n_dctx,
                     const float* const* dw_direct, const int* ld_dw, int n_dw,
                     float* dcum, const float* dcum_add, int ld_dcum_add,
                     float* dpq, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                     float* dctx_t, float* dv_t,
                     int B, int L, int A, int E, int F, int K, void* stream);

/* dmem(b,l,:) = sum_t align(b,t,l) * dctx_tape(t,b,:)   (gradient of the encoder memory through the contexts) */
int st_attn_dmem(const float* align, const float* dctx_tape, float* dmem, int B, int steps, int L, int E, void* stream);

typedef struct st_decoder_bwd_weights {   /* transposed copies prepared by the caller (layout only) */
    const float* q_w_cat_t;        /* [W_ih_q | W_hh_q]^T   (P+E+Q, 4Q)   (may be NULL when the packed form below is given) */
    const float* d_w_cat_t;        /* [W_ih_d | W_hh_d]^T   (E+Q+D, 4D)   (likewise) */
    const float* attn_query_w_t;   /* W_q^T                 (Q, A)        (may be NULL when the loop runs with fuse_pw) */
    const float* q_w_cat_t_p16;    /* optional: q_w_cat_t packed by st_pack_weight (N = P+E+Q, K = 4Q); NULL = natural kernels */
    const float* d_w_cat_t_p16;    /* optional: d_w_cat_t packed (N = E+Q+D, K = 4D) */
    const float* attn_query_w_t_p16;   /* optional: W_q^T packed (N = Q, K = A): with the two above and st_decoder_bwd_io.fuse_pw the loop runs
                                        * the cells' pointwise backward in the epilogues of its products (4 launches per step instead of 6) */
    const float* attn_v; const float* attn_loc_conv_w; const float* attn_loc_lin_w;
} st_decoder_bwd_weights;

/* All step-indexed tensors are natural row-major with Bp >= B rows per step slot (Bp = B rounded up to 16,
 * the row count of the un-tiled forward tapes; rows >= B are zero). */
typedef struct st_decoder_bwd_io {
    const float* memory; const float* pm; const float* ada_std;   /* (B,L,E) (B,L,A) (B,Q) */
    const float* align;            /* (B, steps, L) forward output            */
    const float* wcum_tape;        /* (steps+1, B, L)                         */
    const float* cq_tape; const float* cd_tape;             /* (steps+1, B, Q / D) */
    const float* gates_q_tape; const float* gates_d_tape;   /* (steps, B, 4, Q / D) activated gates */
    const float* q_mask; const float* d_mask;               /* (steps, B, Q / D) or NULL */
    const float* pq_all;           /* (steps, Bp, A)  = W_q h_q_t, recomputed by the caller */
    int steps; int Bp;
    const float* dxo;              /* (steps, Bp, D+E) = [dmel_t | dstop_t] [W_proj ; W_gate] */
    const float* dalign;           /* (B, steps, L) or NULL */
    float* dgq; float* dgd;        /* (steps, Bp, 4Q / 4D) out: pre-activation gate gradients */
    float* dxq; float* dxd;        /* (steps+1, Bp, P+E+Q / E+Q+D) out; slot `steps` must be zero */
    float* dpq;                    /* (steps, Bp, A) out */
    float* ds_tape; float* loc_tape; float* dloc_tape;   /* (steps, B, L, A / F / F) out, see st_attn_step_bwd */
    float* hist_tape; float* dctx_tape; float* dv_tape;  /* (steps, B, L, 2), (steps, B, E), (steps, B, A) out */
    float* dcq; float* dcd;        /* (B,Q) (B,D) scratch, zero on entry */
    float* dhist[2];               /* (B,2,L) x 2 scratch, zero on entry */
    float* dcum;                   /* (B,L) scratch, zero on entry */
    float* dhq_attn;               /* (B,Q) scratch */
    float* dgq_t16; float* dgd_t16;   /* st_t16_floats(B, 4Q / 4D) scratch, ZERO on entry; needed with the packed weights */
    /* own-output feedback (scheduled sampling: step_src[t] == -1 for all rows; unpaired rows: rows >= Bt always):
     * dec_in_{t+1} = prenet(mel_t) for those rows, so the gradient w.r.t. dec_in_{t+1} flows back into mel_t and the
     * projection gradient of step t can only be formed inside the loop.  All NULL / 0 for pure teacher forcing. */
    const int* step_src; int Bt;      /* host array (steps entries) as given to st_decoder_forward */
    float* dY;                        /* (steps, Bp, r*n_mels+1) [dmel_t | dstop_t]; gets the feedback gradient ADDED */
    float* dxo_rw;                    /* = dxo, written per step in this mode */
    const float* wpg_t;               /* [W_proj ; W_gate]^T   (D+E, r*n_mels+1) */
    const float* pre_w1_t; const float* pre_w0_t;   /* prenet W1^T (P,P), W0^T (r*n_mels, P) */
    const float* own_mask;            /* (steps, 2, B, P) or NULL */
    const float* xq_nat;              /* (steps+1, Bp, P+E+Q) un-tiled forward tape (dec_in = first P columns) */
    const float* pre1_nat;            /* (steps, Bp, P) un-tiled prenet layer-1 outputs */
    float* d2_tape; float* dp1_tape;  /* (steps, Bp, P) out, zero on entry: gradients at the two prenet layers (-> dW1, dW0) */
    float* tmp_p; float* tmp_in;      /* (B, P), (B, r*n_mels) scratch */
    int fuse_pw;                      /* != 0 (pure teacher forcing, packed weights incl. attn_query_w_t_p16): pointwise LSTM backward in the
                                       * epilogues of dgates_d . W (for step t-1) and of W_q^T dpq (for step t); needs dgd_t16_b and dpq_t16 */
    float* dgd_t16_b;                 /* second st_t16_floats(B, 4D) buffer, zero on entry (the epilogue writes step t-1 while step t is read) */
    float* dpq_t16;                   /* st_t16_floats(B, A) scratch, zero on entry */
    int need_dxq0;                    /* != 0: also form dxq of step 0 (the gradient of dec_in_0 = prenet(go frame), which only a
                                       * normalised prenet makes non-constant) */
    const float* attn_s_tape;         /* optional (steps, B, L, A): S_t = pm + W_l loc_t saved by the forward (slot 0 unused: S_0 =
                                       * pm).  Then loc_tape is an INPUT (the forward's attn_loc_tape) and the attention backward
                                       * neither recomputes the location conv nor the W_l product. */
    int overlap_attn;                 /* != 0 (with fuse_pw and attn_s_tape): the decoder cell's product of step t-1 and the attention backward
                                       * of step t share one launch -- the decoder cell's recurrence runs one step ahead of the attention /
                                       * query chain (3 launches per step on the critical path instead of 4) */
    /* own-output feedback through a NORMALISED prenet (st_decoder_dims.prenet_norm of the forward; 0 = plain): the forward's
     * pre_nat_tape, the norm weights (and, mode 2, running statistics) of the two layers, and (P) accumulators for their gradients,
     * zero on entry.  d2_tape / dp1_tape then hold the gradients at the two Linear outputs. */
    int prenet_norm;
    const float* pre_y_tape;          /* (steps, 2, B, P) */
    const float* pre_norm_w[2]; const float* pre_norm_rm[2]; const float* pre_norm_rv[2];
    float pre_norm_eps;
    float* dpre_norm_w[2]; float* dpre_norm_b[2];
    /* attn_parts = 2 or 4 (with overlap_attn; 0 / 1 = off): the hosted attention backward of a step runs as that many workgroups per
     * utterance (st_attn_bwd_job.parts) and the step's W_q^T dpq launch hosts the st_attn_hist_job; dloc_part (attn_parts, B, L, F) scratch */
    int attn_parts; float* dloc_part;
    /* with attn_parts = 2 and 16 < B <= 32: (dxd_splits, B, E+Q+D) scratch -- the decoder cell's product of the hosted launch runs K-split
     * (st_skinny_partial_attn_bwd) and is summed, with its pointwise epilogue, in the step's W_q^T dpq launch (st_partial_sum_job); NULL = off */
    float* dxd_part; int dxd_splits;
    /* ... and (steps + 1, dxq_splits, B, P+E+Q) slabs: the query cell's product dgates_q . [W_ih | W_hh] K-split as well; dxq then only receives
     * step 0 (need_dxq0) and the callers sum the slabs' first P columns for the teacher gradient.  NULL = off.  dxq_splits <= 4. */
    float* dxq_part; int dxq_splits;
} st_decoder_bwd_io;
/* 1 when the dimensions allow the fused BPTT loop (st_decoder_bwd_io.fuse_pw: the pointwise halves of the two LSTM cells' backward steps,
 * ref: src/module.py:227-231,275-280, in the epilogues of the loop's products): Q, D, E + Q multiples of 16, A and the input widths of the
 * three products multiples of 4.  A caller asks BEFORE it decides to hand its step tapes over uninitialised; st_decoder_backward then
 * fails (instead of falling back to the six-launch loop, which reads a zero slot) when fuse_pw is set and cannot be honoured. */
int st_decoder_bwd_fuse_dims(const st_decoder_dims* d);
/* which forms st_decoder_backward will take for these dimensions and buffers: bit 0 split attention backward, bit 1 partial decoder-cell
 * product, bit 2 partial query-cell product (dxq_part holds the gradient w.r.t. the query cell's inputs as slabs) */
int st_decoder_bwd_forms(const st_decoder_dims* d, const st_decoder_bwd_io* io);
/* st_attn_step_bwd with S = pm + W_l loc of the step given (s_in, (B,L,A)): loc_t is not written (may be NULL) */
int st_attn_step_bwd_s(const float* pq, const float* pm, const float* memory,
                       const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                       const float* loc_conv_w, const float* loc_lin_w, const float* v,
                       const float* const* dctx, const int* ld_dctx, int n_dctx,
                       const float* const* dw_direct, const int* ld_dw, int n_dw,
                       float* dcum, const float* dcum_add, int ld_dcum_add,
                       float* dpq, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                       float* dctx_t, float* dv_t, const float* s_in,
                       int B, int L, int A, int E, int F, int K, void* stream);
/* st_attn_step_bwd_s with a second copy of dpq in the T16 tile layout (K = A): the operand of the packed W_q^T product that follows */
int st_attn_step_bwd_t16(const float* pq, const float* pm, const float* memory,
                         const float* w_prev, int ld_wprev, const float* w_cum_prev, const float* w, int ld_w,
                         const float* loc_conv_w, const float* loc_lin_w, const float* v,
                         const float* const* dctx, const int* ld_dctx, int n_dctx,
                         const float* const* dw_direct, const int* ld_dw, int n_dw,
                         float* dcum, const float* dcum_add, int ld_dcum_add,
                         float* dpq, const st_t16_view* dpq_t16, float* dhist, float* ds_t, float* loc_t, float* dloc_t, float* hist_t,
                         float* dctx_t, float* dv_t, const float* s_in,
                         int B, int L, int A, int E, int F, int K, void* stream);
/* The arguments of st_attn_step_bwd_t16 as a struct, for the launch that hosts the attention backward of BPTT step t beside the
 * decoder cell's backward product of step t-1 (the decoder cell's recurrence does not depend on the attention / query chain of the
 * same step; backward of src/module.py:247-283): st_skinny_linear_packed_lstm_bwd_fwd + st_attn_step_bwd_t16 in ONE launch. */
typedef struct st_attn_bwd_job {
    const float* pq; const float* pm; const float* memory;
    const float* w_prev; int ld_wprev; const float* w_cum_prev; const float* w; int ld_w;
    const float* loc_conv_w; const float* loc_lin_w; const float* v;
    const float* dctx[3]; int ld_dctx[3]; int n_dctx;
    const float* dw_direct[3]; int ld_dw[3]; int n_dw;
    float* dcum; const float* dcum_add; int ld_dcum_add;
    float* dpq; st_t16_view dpq_t16; float* dhist; float* ds_t; float* loc_t; float* dloc_t; float* hist_t;
    float* dctx_t; float* dv_t; const float* s_in;
    int B, L, A, E, F, K;
    /* parts > 1 (2 or 4; 0 / 1 = the whole step in one workgroup per utterance): `parts` workgroups per utterance, each taking A / parts
     * attention dims of the energy gradient; ds_t / dpq / dv_t / dctx_t are complete afterwards, the location-feature gradient only as
     * `parts` partial sums in dloc_part (parts, B, L, F).  dhist / dloc_t / hist_t are NOT written and dcum is read-only (it must hold
     * the total gradient w.r.t. cum_t; dcum_add must be NULL): the caller runs st_attn_hist_job next, which forms them. */
    int parts; float* dloc_part;
    /* up to three more addends of the context gradient, behind dctx[0 .. n_dctx): the slabs of a K-split product (the gradient w.r.t.
     * ctx_{t} inside dxq_{t+1} when that launch ran in the partial form) */
    const float* dctx_more[3]; int ld_dctx_more[3]; int n_dctx_more;
} st_attn_bwd_job;
/* What a split attention backward (st_attn_bwd_job.parts > 1) leaves behind, one workgroup per utterance: dloc = the partial sums added in
 * part order -> dloc_t (B, L, F); hist_t (B, L, 2) = [w_{t-1}, cum_{t-1}]; dhist (B, 2, L) = the gradient w.r.t. that history through the
 * location conv (the transposed convolution of dloc with loc_conv_w (F, 2, K)); dcum (B, L; may be NULL) += dhist(:, 1, :), the gradient
 * carried along cum_t = cum_{t-1} + w_t.  Backward of src/module.py:235-237,384-385 (stack / conv1d / transpose). */
typedef struct st_attn_hist_job {
    const float* dloc_part; int parts;
    const float* loc_conv_w;
    const float* w_prev; int ld_wprev; const float* w_cum_prev;
    float* dloc_t; float* hist_t; float* dhist; float* dcum;
    int B, L, F, K;
} st_attn_hist_job;
/* job may be NULL (the plain product); ab->s_in must be given (the forward kept S) */
/* two st_skinny_linear_packed_lstm_bwd_fwd of one shape in one launch (arrays of two; y2 or its entries may be NULL) */
int st_skinny_linear_packed_lstm_bwd_pair_fwd(const float* const* packed_w2, const st_t16_view* x2, int K, float* const* y2, int ldy,
                                              int B, int N, const st_lstm_pw_job* job2, void* stream);
int st_skinny_linear_packed_lstm_bwd_attn_bwd(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                              const st_lstm_pw_job* job, const st_attn_bwd_job* ab, void* stream);
/* 1 when the attention backward's 48-position block (+ a hosting product's 8 KB) fits the LDS for these dims: what parts > 1 needs */
int st_attn_bwd_wide_fits(int L, int A, int E, int F, int K);
/* st_skinny_linear_packed_lstm_bwd_fwd with an st_attn_hist_job beside it in the same launch (the BPTT step's W_q^T dpq product, which
 * occupies half of the compute units, hosts the history part of the step's split attention backward) */
int st_skinny_linear_packed_lstm_bwd_attn_hist(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                               const st_lstm_pw_job* job, const st_attn_hist_job* hist, void* stream);
/* K-split partial product for 16 < B <= 32 rows: part(s, b, n) = sum over the s-th of S equal ranges of k of x(b, k) W(n, k) -- two row
 * tiles and both batch tiles per workgroup (half the bytes per output of st_skinny_linear_packed_fwd, whose workgroups each re-read a
 * whole batch tile of x), N / 32 * S workgroups; N % 32 == 0, (K / 16) % S == 0; part (S, B, N).  With `ab` (an st_attn_bwd_job with
 * parts = 2) the split attention backward of a BPTT step runs beside it in the same launch, as in
 * st_skinny_linear_packed_lstm_bwd_attn_bwd.  The slabs are added -- in split order -- by an st_partial_sum_job of the caller's next launch:
 *   y(b, n) = sum_s part(s, b, n), and for the columns [pw->n0, pw->n0 + pw->H) the pointwise LSTM backward of st_lstm_pw_job on them
 * (pw may be NULL).  Backward of nn.LSTMCell's product, ref: src/module.py:277 (dgates . [W_ih | W_hh]). */
typedef struct st_partial_sum_job {
    const float* part; int S; int N;
    float* y; int ldy;
    const st_lstm_pw_job* pw;
} st_partial_sum_job;
int st_skinny_partial_attn_bwd(const float* packed_w, const st_t16_view* x, int K, float* part, int S, int B, int N,
                               const st_attn_bwd_job* ab, void* stream);
/* the partial product with an st_attn_hist_job beside it (BPTT launch 3, dgates_q . [W_ih | W_hh], in the partial form: its consumers -- the
 * next attention backward's context addend and the query cell's dh1 addend -- take the S slabs as they are, st_attn_bwd_job.dctx_more and
 * st_lstm_pw_job.dh1_slabs) */
int st_skinny_partial_attn_hist(const float* packed_w, const st_t16_view* x, int K, float* part, int S, int B, int N,
                                const st_attn_hist_job* hist, void* stream);
/* st_skinny_linear_packed_lstm_bwd_attn_hist with an st_partial_sum_job as well (same launch); hist may be NULL */
int st_skinny_linear_packed_lstm_bwd_attn_hist_sum(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                                   const st_lstm_pw_job* job, const st_attn_hist_job* hist,
                                                   const st_partial_sum_job* sum, void* stream);
/* y = x W^T (no epilogue) with an st_attn_hist_job beside it: the BPTT step's dgates_q . [W_ih | W_hh] launch leaves 32 compute units idle
 * for 9 us -- the history part of the step's split attention backward hides there completely */
int st_skinny_linear_packed_attn_hist(const float* packed_w, const st_t16_view* x, int K, float* y, int ldy, int B, int N,
                                      const st_attn_hist_job* hist, void* stream);
int st_decoder_backward(const st_decoder_bwd_weights* w, const st_decoder_dims* d, const st_decoder_bwd_io* io, void* stream);
/* The two loops of a training step (st_decoder_forward with defer_proj, st_decoder_backward) are called with bit-identical arguments step after
 * step once the caller's allocator repeats its addresses: at the second sighting of an argument set the loop's launches are captured into a
 * hipGraph (thread-local capture on the calling stream) and replayed with one launch whenever the set comes back; other sets run eagerly.
 * OFF unless asked for (ST_LOOP_GRAPHS=1, st_loop_graphs_enable): it saves host time (0.8 ms of a C2 step), which the measured steps are not bound by.  st_loop_graph_stats: {replays, captures, eager calls} of the forward / backward loop (either may be NULL). */
void st_loop_graph_stats(long* fwd3, long* bwd3);
/* the replay at run time: 1 on, 0 off, -1 re-read ST_LOOP_GRAPHS; returns the previous setting */
int st_loop_graphs_enable(int on);
/* dY(t, b, :) = [dmel(b, t*r .. t*r+r-1, :) | sum_j dstop(b, t*r+j)]   (steps, Bp) rows of r*n_mels+1 values, row stride ld floats
 * (a stride rounded up to a multiple of 4, pad columns zero, puts the two products over dY on the 16-byte kernels); NULL = zeros */
int st_decoder_pack_dout(const float* dmel, const float* dstop, float* dY, int ld, int B, int Bp, int steps, int r, int n_mels,
                         void* stream);
/* the forward twin (teacher forcing with the deferred projection): Y (steps, Bp) rows [mel_t | stop_t] of stride ld -> mel (B, steps*r, n_mels)
 * and stop (B, steps*r), a step's stop value repeated r times.  ref: src/module.py:283-287 */
int st_decoder_unpack_out(const float* Y, float* mel, float* stop, int B, int Bp, int steps, int r, int n_mels, int ld, void* stream);
/* gradient of prenet(teacher) under plain teacher forcing (step t + 1 reads frame t, src/module.py:190-206): dteacher (Bt, Tt, P) = the dec_in
 * columns of dxq_{t+1}, summed over the S slabs of dxq_part (steps + 1, S, Bp, XQw) in slab order; frames t >= steps - 1: zeros */
int st_decoder_dteacher_sum(const float* dxq_part, float* dteacher, int S, int Bp, int XQw, int Bt, int Tt, int P, int steps, void* stream);
/* AdaIN statistics gradients (ref: src/module.py:267-269): adapted_t = std * (h_q_t - mean)
 * dstd = sum_t dadapt_t * (h_q_t - mean), dmean = -std * sum_t dadapt_t; *_step_stride / *_ld in elements */
int st_adain_bwd(const float* dadapt, long da_step_stride, int da_ld, const float* hq, long hq_step_stride, int hq_ld,
                 const float* ada_std, const float* ada_mean, float* dstd, float* dmean, int B, int Q, int steps, void* stream);

/* CTC loss of the paired ASR branch (ref: torch.nn.CTCLoss() as called by compute_ctcloss, bin/train_vqvae.py:430-444, with
 * paras.actual_len = False): lp = log(prob + eps), targets = the non-zero tokens of text (B, L) in order, every one of the T
 * frames counts, blank = 0, loss = mean_b(nll_b / max(S_b, 1)).  prob (B, T, V) posteriors over the codebook.  One call gives the
 * loss (device scalar) and, when dprob != NULL, d loss / d prob (the backward scales it by the incoming scalar: st_scale_by).
 * ws: st_ctc_workspace_floats(B, T) floats.  Transcripts of up to 127 tokens, at most 10240 classes.
 * log_input != 0: `prob` already holds log-probabilities (ASRPostnet's log_softmax, src/asr.py:80; compute_ctcloss(..., apply_log=False),
 * bin/train_vqvae.py:212) and the gradient is with respect to them; eps is ignored.
 * A token outside [0, V) makes that utterance's loss and gradient NaN (torch raises; the device cannot). */
size_t st_ctc_workspace_floats(int B, int T);
int st_ctc_loss(const float* prob, const int64_t* text, float eps, float* loss, float* dprob, float* ws,
                int B, int T, int V, int L, int log_input, void* stream);

/* The trainer's scalar arithmetic on loss values as one launch (ref: bin/train_vqvae.py:208-233: total_loss = asr_weight * asr_loss +
 * tts_weight * (mel_loss + linear_loss) + unpair_speech_weight * ... -- a chain of one-element torch kernels there):
 * *outs[j] = sum_i W[j * n + i] * *xs[i] for j < m (m <= 4 outputs, n <= ST_SCALAR_MAX terms, W on the host; in rows j > 0 a zero weight means the term is not a member of that sum).
 * st_scalar_fanout is the backward of output 0: out[i] = w[i] * *dout. */
#define ST_SCALAR_MAX 8
int st_scalar_combine(const float* const* xs, int n, const float* W, int m, float* const* outs, void* stream);
int st_scalar_fanout(const float* dout, const float* w, int n, float* out, void* stream);

/* ------------------------------------------------------------------ trainer loss
 * loss = w_all*crit(pred,label) + w_low*crit(low n_low bins) + w_diff*crit(first differences along T), crit = MSE or
 * (l1 != 0) mean absolute error; writes the scalar to *loss and d loss / d pred to dpred (B,T,D).  ws: st_freq_loss_workspace_floats() floats (one partial sum per workgroup).
 * ref: freq_loss src/util.py:80-126 (mel: w = 1, 0, 0.5; linear with low-band emphasis: w = 0.5, 0.5, 0) */
size_t st_freq_loss_workspace_floats(void);
int st_freq_loss(const float* pred, const float* label, float* loss, float* dpred, float* ws,
                 int B, int T, int D, int n_low, float w_all, float w_low, float w_diff, int l1, void* stream);
/* Several st_gemm_wgrad / st_gemm_wgrad_db (db may be NULL per job; no pooling, no accumulation) as ONE product launch and ONE slab-sum launch
 * when every job takes the LDS-DMA form (16-byte addressable rows, N and Cin multiples of 4, Cin >= 16) on 64-tiles; otherwise one after the
 * other.  Each job keeps its own slab count and summation order: bit for bit the results of the separate calls.  For the K weight
 * gradients of the CBHG conv bank (80 x 80 x k: eight launches that each leave most of the chip idle) and the two directions of a
 * recurrent layer.  ws: st_gemm_wgrad_batch_workspace_floats(jobs, n) floats.  ref: backward of src/module.py:590-598. */
typedef struct st_wgrad_job {
    const float* dC; int lddc, dcoff; const float* A; int lda; float* dW; float* db;
    int Bn, Tin, Tout, Cin, N, KT, pad;
} st_wgrad_job;
size_t st_gemm_wgrad_batch_workspace_floats(const st_wgrad_job* jobs, int n);
int st_gemm_wgrad_batch(const st_wgrad_job* jobs, int n, float* ws, void* stream);
/* st_gemm_wgrad[_db] of a Linear over CONCATENATED inputs (KT = 1; M rows), the result cut at input column `split`:
 * dW0 (N, split) and dW1 (N, Cin - split) are the gradients of the two weights whose columns the product saw side by side -- an
 * nn.LSTMCell fed with [x | h] has gates = [W_ih | W_hh] [x | h]^T (ref: src/module.py:227-231,275-280), so the BPTT weight gradient
 * over the step tapes lands in the two parameters' own gradient tensors (under data parallelism: in their all-reduce bucket slots)
 * instead of one tensor that two copies then cut up.  db (may be NULL) = column sums of dC; db_dup (may be NULL) receives the same
 * sums once more (b_ih and b_hh have equal gradients). */
int st_gemm_wgrad_split(const float* dC, int lddc, int dcoff, const float* A, int lda, float* dW0, int split, float* dW1,
                        float* db, float* db_dup, float* ws, int M, int Cin, int N, int accumulate, void* stream);
/* y = x * (*scalar)   (scalar on the device: the incoming gradient of a scalar loss) */
int st_scale_by(const float* x, const float* scalar, float* y, size_t n, void* stream);

/* ------------------------------------------------------------------ optimiser step (multi-tensor)
 * ref: BaseSolver.backward src/solver.py:138-151 (clip_grad_norm_ 5.0, optimizer.step()), torch.optim.Adam as built by
 * src/optim.py.  The pointer arrays are HOST arrays of device pointers (nt tensors, n[t] elements each); the
 * kernels take them through their arguments, a few launches for ~100 tensors. */
long st_mt_table_misses(void);   /* diagnostics: block maps built and uploaded so far (once per set of tensor addresses; the multi-tensor
                                    * launches below read their tensor list from a cached device-side table) */
size_t st_mt_blocks(const long* n, int nt);                   /* floats of `partials` st_mt_grad_norm needs */
/* *norm_out (device) = sqrt(sum_t sum_i g_t[i]^2), fixed summation order */
int st_mt_grad_norm(float* const* g, const long* n, int nt, float* partials, float* norm_out, void* stream);
/* g *= max_norm / (norm + 1e-6) when that is < 1 (torch.nn.utils.clip_grad_norm_); norm is a device scalar */
int st_mt_clip_scale(float* const* g, const long* n, int nt, const float* norm, float max_norm, void* stream);
/* The same pair for gradients that are still the SUM over `world` data-parallel ranks (pre_scale = 1 / world): *norm_out =
 * pre_scale * sqrt(sum g^2) = the norm of the AVERAGED gradients, and the clip launch multiplies every gradient by
 * pre_scale * min(1, max_norm / (norm + 1e-6)) -- the average of the all-reduce costs no launch of its own (ref: the reference
 * clips the single-process gradient, src/solver.py:145; data parallelism is this path's extension). */
int st_mt_grad_norm_scaled(float* const* g, const long* n, int nt, float* partials, float* norm_out, float pre_scale, void* stream);
int st_mt_clip_scale_pre(float* const* g, const long* n, int nt, const float* norm, float max_norm, float pre_scale, void* stream);
/* dst_t[i] = src_t[i] for nt tensors in a handful of launches (gradients that autograd allocated elsewhere are gathered into their
 * all-reduce bucket by one call per bucket) */
int st_mt_copy(float* const* dst, float* const* src, const long* n, int nt, void* stream);
/* torch.optim.Adam (no weight decay, no amsgrad): m = m + (g-m)(1-b1); v = b2 v + (1-b2) g^2;
 * p -= step_size * m / (sqrt(v)/bias_correction2_sqrt + eps), step_size = lr / (1 - b1^t) */
int st_mt_adam(float* const* p, float* const* g, float* const* m, float* const* v, const long* n, int nt,
               float beta1, float beta2, float eps, float step_size, float bias_correction2_sqrt, void* stream);
/* st_mt_adam that leaves parameters and moments untouched when *guard_norm (a device scalar: the gradient norm st_mt_grad_norm wrote)
 * is NaN or inf -- BaseSolver.backward's `if math.isnan(grad_norm): ... else: self.optimizer.step()` (src/solver.py:147-150) decided
 * on the device, so a training loop need not wait for the norm on the host every step.  guard_norm NULL = st_mt_adam. */
int st_mt_adam_guarded(float* const* p, float* const* g, float* const* m, float* const* v, const long* n, int nt,
                       float beta1, float beta2, float eps, float step_size, float bias_correction2_sqrt,
                       const float* guard_norm, void* stream);

/* ------------------------------------------------------------------ small utilities */
int st_fill(float* p, float v, size_t n, void* stream);

/* Re-layouts of many PARAMETERS in one launch: what `w.permute(0, 2, 1).contiguous()` (mode 0: Conv1d weight (N, Cin, KT) -> tap-major
 * (N, KT, Cin), the forward implicit-GEMM operand) and `w.permute(1, 0, 2).flip(2)` in tap-major form (mode 1: -> (Cin, KT, N) with the
 * taps reversed, the weight of the input-gradient conv; KT = 1: the transpose of a Linear weight) do one torch copy at a time.
 * `table_dev` is a DEVICE array of n descriptors ordered by blk0; descriptor d owns workgroups [blk0, blk0 + st_relayout_blocks(N, Cin, KT)).
 * Replaces the per-weight layout copies in front of torch's conv / linear backward (src/module.py: every Conv1d / Linear). */
typedef struct st_relayout_desc {
    const float* src; float* dst;
    int N, Cin, KT, mode;
    int blk0;
    int ld_dst;     /* mode 1 only: row stride of dst in floats (0 = N): the layout lands in a column block of a wider matrix (the
                     * transposes of several Linear weights side by side = the weight of ONE input-gradient product) */
} st_relayout_desc;
int st_relayout_blocks(int N, int Cin, int KT);
/* blk_desc_dev: optional DEVICE array of total_blocks ints, the descriptor index of every workgroup (NULL: each workgroup searches the table) */
int st_relayout_batch(const st_relayout_desc* table_dev, int n, int total_blocks, const int* blk_desc_dev, void* stream);
int st_copy2d(float* dst, int ldd, const float* src, int lds, int rows, int cols, void* stream);
/* dst(b, :) = mean over t of src(b, t, :)    ref: teacher.mean(dim=1) src/module.py:194 */
int st_mean_rows(const float* src, float* dst, int B, int T, int D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMITTS_H */
