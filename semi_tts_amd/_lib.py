"""ctypes binding of libsemitts_hip.so (the C ABI declared in include/semitts.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Nothing here imports the oracle.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ST_LIB_PATH') or os.path.join(HERE, 'lib', 'libsemitts_hip.so')   # ST_LIB_PATH: kernel experiments

c_float_p = C.c_void_p   # device pointers travel as integers
c_void_p = C.c_void_p


class StSeg(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('ldx', C.c_int), ('ldw', C.c_int), ('k', C.c_int)]


class StT16View(C.Structure):
    _fields_ = [('base', C.c_void_p), ('kb_stride', C.c_int), ('kb0', C.c_int)]


class StGemmEpilogue(C.Structure):
    _fields_ = [('bias', C.c_void_p), ('act_pre', C.c_int),
                ('bn_mean', C.c_void_p), ('bn_var', C.c_void_p), ('bn_w', C.c_void_p), ('bn_b', C.c_void_p),
                ('bn_eps', C.c_float), ('act_post', C.c_int),
                ('res', C.c_void_p), ('ldres', C.c_int),
                ('highway_h', C.c_void_p), ('ldhw', C.c_int),
                ('mask', C.c_void_p), ('ldmask', C.c_int), ('w_tap_major', C.c_int),
                ('splitk_ws', C.c_void_p), ('splitk_slabs', C.c_int)]


class StGemmJob(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int), ('W', C.c_void_p), ('C', C.c_void_p), ('ldc', C.c_int), ('coff', C.c_int),
                ('Bn', C.c_int), ('Tin', C.c_int), ('Tout', C.c_int), ('Cin', C.c_int), ('N', C.c_int), ('KT', C.c_int),
                ('pad', C.c_int), ('stride', C.c_int), ('pool_prev', C.c_int), ('ep', StGemmEpilogue)]


class StRelayoutDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('N', C.c_int), ('Cin', C.c_int), ('KT', C.c_int), ('mode', C.c_int),
                ('blk0', C.c_int), ('ld_dst', C.c_int)]


class StBnBankSeg(C.Structure):
    _fields_ = [('x', C.c_void_p), ('ldx', C.c_int), ('T', C.c_int), ('w', C.c_void_p), ('b', C.c_void_p), ('run_mean', C.c_void_p),
                ('run_var', C.c_void_p), ('batches_tracked', C.c_void_p), ('momentum', C.c_float), ('eps', C.c_float),
                ('mean', C.c_void_p), ('var', C.c_void_p), ('dx', C.c_void_p), ('lddx', C.c_int), ('sums', C.c_void_p)]


class StDecoderWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'prenet_w0', 'prenet_w1', 'q_w_ih', 'q_w_hh', 'q_b_ih', 'q_b_hh',
        'attn_query_w', 'attn_v', 'attn_loc_conv_w', 'attn_loc_lin_w',
        'd_w_ih', 'd_w_hh', 'd_b_ih', 'd_b_hh', 'projgate_w', 'projgate_b')] + [
        ('pre_norm_w', C.c_void_p * 2), ('pre_norm_b', C.c_void_p * 2), ('pre_norm_rm', C.c_void_p * 2), ('pre_norm_rv', C.c_void_p * 2),
        ('pre_norm_nbt', C.c_void_p * 2), ('pre_norm_eps', C.c_float), ('pre_norm_momentum', C.c_float)]


class StDecoderDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('B', 'L', 'E', 'n_mels', 'r', 'P', 'Q', 'D', 'A', 'F', 'K', 'fuse_pre0', 'prenet_norm')]


class StDecoderIO(C.Structure):
    _fields_ = [('memory', C.c_void_p), ('pm', C.c_void_p), ('ada_std', C.c_void_p), ('ada_mean', C.c_void_p),
                ('step_src', C.POINTER(C.c_int)), ('teacher_pre', C.c_void_p), ('teacher_mean', C.c_void_p),
                ('Bt', C.c_int), ('Tt', C.c_int),
                ('prenet_mask', C.c_void_p), ('q_mask', C.c_void_p), ('d_mask', C.c_void_p),
                ('steps', C.c_int),
                ('mel_out', C.c_void_p), ('align_out', C.c_void_p), ('stop_out', C.c_void_p),
                ('packed', C.c_void_p),
                ('xq_tape', C.c_void_p), ('xd_tape', C.c_void_p), ('xo_tape', C.c_void_p),
                ('cq_tape', C.c_void_p), ('cd_tape', C.c_void_p), ('wcum_tape', C.c_void_p),
                ('pq_buf', C.c_void_p), ('pre1_t16', C.c_void_p), ('mel_t16', C.c_void_p),
                ('zero_row', C.c_void_p),
                ('gates_q_tape', C.c_void_p), ('gates_d_tape', C.c_void_p), ('attn_s_buf', C.c_void_p), ('attn_pre_parts', C.c_int), ('attn_fin_parts', C.c_int), ('defer_proj', C.c_int),
                ('pre1_step_floats', C.c_int), ('attn_s_step_floats', C.c_int), ('attn_loc_tape', C.c_void_p),
                ('attn_split_ws', C.c_void_p), ('attn_split_parts', C.c_int), ('pq_granules', C.c_void_p),
                ('dec_in0', C.c_void_p), ('pre_nat', C.c_void_p), ('attn_xchg', C.c_void_p), ('handoff_status', C.c_void_p),
                ('pair_cells', C.c_int), ('pre_nat_tape', C.c_void_p), ('pre1_granules', C.c_void_p), ('gate_part', C.c_void_p), ('gate_part_k', C.c_int)]


class StDecoderBwdWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('q_w_cat_t', 'd_w_cat_t', 'attn_query_w_t', 'q_w_cat_t_p16', 'd_w_cat_t_p16', 'attn_query_w_t_p16',
                                          'attn_v', 'attn_loc_conv_w', 'attn_loc_lin_w')]


class StDecoderBwdIO(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ('memory', 'pm', 'ada_std', 'align', 'wcum_tape', 'cq_tape', 'cd_tape',
                                           'gates_q_tape', 'gates_d_tape', 'q_mask', 'd_mask', 'pq_all')] +
                [('steps', C.c_int), ('Bp', C.c_int)] +
                [(n, C.c_void_p) for n in ('dxo', 'dalign', 'dgq', 'dgd', 'dxq', 'dxd', 'dpq', 'ds_tape', 'loc_tape', 'dloc_tape',
                                           'hist_tape', 'dctx_tape', 'dv_tape', 'dcq', 'dcd')] +
                [('dhist', C.c_void_p * 2), ('dcum', C.c_void_p), ('dhq_attn', C.c_void_p), ('dgq_t16', C.c_void_p),
                 ('dgd_t16', C.c_void_p), ('step_src', C.POINTER(C.c_int)), ('Bt', C.c_int)] +
                [(n, C.c_void_p) for n in ('dY', 'dxo_rw', 'wpg_t', 'pre_w1_t', 'pre_w0_t', 'own_mask', 'xq_nat', 'pre1_nat',
                                           'd2_tape', 'dp1_tape', 'tmp_p', 'tmp_in')] +
                [('fuse_pw', C.c_int), ('dgd_t16_b', C.c_void_p), ('dpq_t16', C.c_void_p), ('need_dxq0', C.c_int), ('attn_s_tape', C.c_void_p),
                 ('overlap_attn', C.c_int), ('prenet_norm', C.c_int), ('pre_y_tape', C.c_void_p), ('pre_norm_w', C.c_void_p * 2),
                 ('pre_norm_rm', C.c_void_p * 2), ('pre_norm_rv', C.c_void_p * 2), ('pre_norm_eps', C.c_float),
                 ('dpre_norm_w', C.c_void_p * 2), ('dpre_norm_b', C.c_void_p * 2), ('attn_parts', C.c_int), ('dloc_part', C.c_void_p),
                 ('dxd_part', C.c_void_p), ('dxd_splits', C.c_int), ('dxq_part', C.c_void_p), ('dxq_splits', C.c_int)])



class StLstmPwJob(C.Structure):
    _fields_ = [('n0', C.c_int), ('H', C.c_int), ('dh1', C.c_void_p), ('ld1', C.c_int), ('dh2', C.c_void_p), ('ld2', C.c_int),
                ('scale2', C.c_void_p), ('mask', C.c_void_p), ('gates', C.c_void_p), ('c', C.c_void_p), ('ldc', C.c_int),
                ('c_prev', C.c_void_p), ('ldcp', C.c_int), ('dc', C.c_void_p), ('dgates', C.c_void_p), ('ldg', C.c_int),
                ('dgates_t16', StT16View), ('dh1_slabs', C.c_int), ('dh1_slab_stride', C.c_long)]


class StLstmCellPackedJob(C.Structure):
    _fields_ = [('packed_w', C.c_void_p), ('x', StT16View), ('K', C.c_int), ('b_ih', C.c_void_p), ('b_hh', C.c_void_p),
                ('c_prev', C.c_void_p), ('ldc_prev', C.c_int), ('mask', C.c_void_p), ('h_dst0', StT16View), ('h_dst1', StT16View),
                ('c_out', C.c_void_p), ('ldc', C.c_int), ('gates_out', C.c_void_p), ('ada_std', C.c_void_p), ('ada_mean', C.c_void_p),
                ('hadapt_dst', StT16View), ('B', C.c_int), ('H', C.c_int), ('part', C.c_void_p), ('w_kbs', C.c_int)]


class StAttnBwdJob(C.Structure):
    _fields_ = [('pq', C.c_void_p), ('pm', C.c_void_p), ('memory', C.c_void_p),
                ('w_prev', C.c_void_p), ('ld_wprev', C.c_int), ('w_cum_prev', C.c_void_p), ('w', C.c_void_p), ('ld_w', C.c_int),
                ('loc_conv_w', C.c_void_p), ('loc_lin_w', C.c_void_p), ('v', C.c_void_p),
                ('dctx', C.c_void_p * 3), ('ld_dctx', C.c_int * 3), ('n_dctx', C.c_int),
                ('dw_direct', C.c_void_p * 3), ('ld_dw', C.c_int * 3), ('n_dw', C.c_int),
                ('dcum', C.c_void_p), ('dcum_add', C.c_void_p), ('ld_dcum_add', C.c_int),
                ('dpq', C.c_void_p), ('dpq_t16', StT16View), ('dhist', C.c_void_p), ('ds_t', C.c_void_p), ('loc_t', C.c_void_p),
                ('dloc_t', C.c_void_p), ('hist_t', C.c_void_p), ('dctx_t', C.c_void_p), ('dv_t', C.c_void_p), ('s_in', C.c_void_p),
                ('B', C.c_int), ('L', C.c_int), ('A', C.c_int), ('E', C.c_int), ('F', C.c_int), ('K', C.c_int),
                ('parts', C.c_int), ('dloc_part', C.c_void_p),
                ('dctx_more', C.c_void_p * 3), ('ld_dctx_more', C.c_int * 3), ('n_dctx_more', C.c_int)]


class StAttnHistJob(C.Structure):
    _fields_ = [('dloc_part', C.c_void_p), ('parts', C.c_int), ('loc_conv_w', C.c_void_p),
                ('w_prev', C.c_void_p), ('ld_wprev', C.c_int), ('w_cum_prev', C.c_void_p),
                ('dloc_t', C.c_void_p), ('hist_t', C.c_void_p), ('dhist', C.c_void_p), ('dcum', C.c_void_p),
                ('B', C.c_int), ('L', C.c_int), ('F', C.c_int), ('K', C.c_int)]


class StWgradJob(C.Structure):
    _fields_ = [('dC', C.c_void_p), ('lddc', C.c_int), ('dcoff', C.c_int), ('A', C.c_void_p), ('lda', C.c_int), ('dW', C.c_void_p), ('db', C.c_void_p),
                ('Bn', C.c_int), ('Tin', C.c_int), ('Tout', C.c_int), ('Cin', C.c_int), ('N', C.c_int), ('KT', C.c_int), ('pad', C.c_int)]


class StPartialSumJob(C.Structure):
    _fields_ = [('part', C.c_void_p), ('S', C.c_int), ('N', C.c_int), ('y', C.c_void_p), ('ldy', C.c_int), ('pw', C.c_void_p)]


class StAttnPreJob(C.Structure):
    _fields_ = [('pm', C.c_void_p), ('w_prev', C.c_void_p), ('ld_wprev', C.c_int), ('w_cum_prev', C.c_void_p),
                ('loc_conv_w', C.c_void_p), ('loc_lin_w', C.c_void_p), ('s_buf', C.c_void_p),
                ('L', C.c_int), ('A', C.c_int), ('F', C.c_int), ('K', C.c_int), ('parts', C.c_int), ('cf_out', C.c_void_p),
                ('p2_packed_w', C.c_void_p), ('p2_K', C.c_int), ('p2_N', C.c_int), ('p2_act', C.c_int), ('p2_mask', C.c_void_p),
                ('p2_ldmask', C.c_int), ('p2_dst', StT16View), ('p2_gran', C.c_void_p), ('p2_epoch', C.c_uint), ('p2_status', C.c_void_p),
                ('part', C.c_void_p)]


class StAttnFinJob(C.Structure):
    _fields_ = [('s_buf', C.c_void_p), ('memory', C.c_void_p), ('w_cum_prev', C.c_void_p),
                ('w_out', C.c_void_p), ('ld_wout', C.c_int), ('w_cum_out', C.c_void_p), ('v', C.c_void_p),
                ('ctx_dst', StT16View * 3), ('n_ctx_dst', C.c_int), ('parts', C.c_int),
                ('L', C.c_int), ('A', C.c_int), ('E', C.c_int), ('F', C.c_int), ('K', C.c_int), ('status', C.c_void_p)]


P, I, F, Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/semitts.h
SIGNATURES = {
    'st_last_error': [],
    'st_abi_version': [],
    'st_device_info': [C.POINTER(I), C.POINTER(I), C.c_char_p, I],
    'st_stream_create': [C.POINTER(P)],
    'st_stream_destroy': [P],
    'st_stream_sync': [P],
    'st_graph_begin': [P],
    'st_graph_end': [P, C.POINTER(P)],
    'st_graph_launch': [P, P],
    'st_graph_destroy': [P],
    'st_event_create': [C.POINTER(P)],
    'st_event_record': [P, P],
    'st_event_elapsed_ms': [P, P, C.POINTER(F)],
    'st_event_destroy': [P],
    'st_lstm_cell_fwd': [C.POINTER(StSeg), I, P, P, P, I, P, I, P, P, I, P, I, P, I, I, P],
    'st_skinny_linear_fwd': [C.POINTER(StSeg), I, P, I, P, I, P, I, I, P, I, I, I, I, P],
    'st_attn_step_fwd': [P, P, P, P, I, P, P, I, P, P, P, P, P, I, P, I, P, P, P, I, I, I, I, I, I, I, P],
    'st_gemm_fwd': [P, I, P, P, I, I, I, I, I, I, I, I, I, I, I, C.POINTER(StGemmEpilogue), P],
    'st_gemm_fwd_batch': [C.POINTER(StGemmJob), I, P],
    'st_highway_stack_supported': [I, I],
    'st_highway_stack_fwd': [P, I, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), I, P, I, I, I, P],
    'st_gemm_splitk_slabs': [I, I, I, I, I],
    'st_bn_stats': [P, I, I, I, I, P, P, P, P, F, P, P, P],
    'st_colreduce_workspace_floats': [I, I],
    'st_bn_apply': [P, I, I, I, I, P, P, P, P, F, I, P],
    'st_lstm_seq_fwd': [P, P, P, P, I, I, P, P, P, I, I, I, I, P],
    'st_lstm_cell_bwd_pointwise': [P, I, P, I, P, I, P, P, P, P, I, P, I, P, P, I, C.POINTER(StT16View), I, I, P],
    'st_lstm_seq_bwd': [P, I, I, P, P, P, P, P, I, I, I, I, P],
    'st_gru_seq_fwd': [P, P, P, P, P, P, P, I, P, I, I, I, I, P],
    'st_gru_seq_bwd': [P, I, P, I, P, P, P, P, P, P, P, I, I, I, I, P],
    'st_vq_build_table': [P, I, P, I, P, P, I, P, I, P],
    'st_gather_rows': [P, P, P, I, I, I, P],
    'st_vq_l2_fwd': [P, P, P, P, P, P, P, I, I, I, P],
    'st_vq_pack_table': [P, P, I, I, P],
    'st_vq_l2_packed_fwd': [P, P, P, P, P, P, P, I, I, I, P],
    'st_vq_l2_workspace_floats': [I, I],
    'st_ctc_workspace_floats': [I, I],
    'st_ctc_loss': [P, P, C.c_float, P, P, P, I, I, I, I, I, P],
    'st_scalar_combine': [P, I, P, I, P, P],
    'st_scalar_fanout': [P, P, I, P, P],
    'st_softmax_bwd': [P, P, P, C.c_float, P, P, I, I, P],
    'st_rowscale_combine': [P, C.c_float, P, P, C.c_float, P, P, I, I, P],
    'st_vq_mean_fwd': [P, P, P, P, P, P, I, I, I, I, I, P],
    'st_vq_mean_bwd': [P, I, P, P, P, I, I, I, P],
    'st_softmax_argmax': [P, P, P, I, I, P],
    'st_packed_weight_floats': [C.POINTER(I), I, I, I],
    'st_t16_floats': [I, I],
    'st_pack_weight': [C.POINTER(P), C.POINTER(I), C.POINTER(I), I, I, I, P, P],
    'st_pack_weight_t': [C.POINTER(P), C.POINTER(I), C.POINTER(I), I, I, P, P],
    'st_relayout_blocks': [I, I, I],
    'st_relayout_batch': [P, I, I, P, P],
    'st_pack_weight_batch': [P, I, P],
    'st_tile_rows': [P, I, C.POINTER(StT16View), I, I, P],
    'st_untile_rows': [C.POINTER(StT16View), P, I, I, I, P],
    'st_lstm_cell_packed_fwd': [P, C.POINTER(StT16View), I, P, P, P, I, P, C.POINTER(StT16View), C.POINTER(StT16View),
                                P, I, P, P, P, C.POINTER(StT16View), I, I, P],
    'st_lstm_cell_packed_part_fwd': [P, I, C.POINTER(StT16View), I, P, P, P, P, I, P, C.POINTER(StT16View), C.POINTER(StT16View),
                                     P, I, P, I, I, P],
    'st_lstm_cell_packed_pair_fwd': [C.POINTER(StLstmCellPackedJob), C.POINTER(StLstmCellPackedJob), P],
    'st_skinny_linear_packed_fwd': [P, C.POINTER(StT16View), I, P, I, P, I, P, I, C.POINTER(StT16View), I, P, I, I,
                                    I, I, P, I, C.POINTER(StT16View), I, I, P],
    'st_attn_step_t16_fwd': [P, P, P, P, I, P, P, I, P, P, P, P, C.POINTER(StT16View), I, P, I, I, I, I, I, I, I, P],
    'st_skinny_linear_packed_attnpre_fwd': [P, C.POINTER(StT16View), I, P, I, P, I, P, I, C.POINTER(StT16View), I, P, I, I,
                                            I, I, P, I, C.POINTER(StT16View), I, I, C.POINTER(StAttnPreJob), P],
    'st_attn_pre_fwd': [P, P, I, P, P, P, P, I, I, I, I, I, I, P],
    'st_attn_fin_t16_fwd': [P, P, P, P, P, I, P, P, C.POINTER(StT16View), I, P, I, I, I, I, I, I, I, I, P],
    'st_attn_fin_split_workspace_floats': [I, I, I],
    'st_attn_fin_split_fwd': [P, P, P, P, P, I, P, P, C.POINTER(StT16View), I, P, I, P, I, I, I, I, I, P],
    'st_query_attn_fin_fwd': [P, C.POINTER(StT16View), I, P, C.c_uint, C.POINTER(StAttnFinJob), I, P],
    'st_query_attn_fin_part_fwd': [P, C.POINTER(StT16View), I, P, C.c_uint, C.POINTER(StAttnFinJob), I, P, P],
    'st_partial_product_fwd': [P, I, P],
    'st_layer_norm_fwd': [P, I, P, P, F, P, I, P, P, I, I, P],
    'st_layer_norm_bwd': [P, I, P, I, P, P, P, P, I, P, I, I, P],
    'st_log_softmax_fwd': [P, P, I, I, P],
    'st_log_softmax_bwd': [P, P, P, I, I, P],
    'st_prenet_norm_fwd': [P, I, I, P, P, P, P, P, F, F, P, I, C.POINTER(StT16View), I, I, I, P],
    'st_prenet_norm_bwd': [P, I, P, I, I, P, P, P, F, P, P, I, I, P],
    'st_handoff_wait_selftest': [P, C.c_uint, I, P, P, I, P],
    'st_query_attn_rng_fits': [I, I, I],
    'st_attn_rng_xchg_words': [I, I, I],
    'st_query_attn_rng_fwd': [P, C.POINTER(StT16View), I, P, P, C.c_uint, C.POINTER(StAttnFinJob), I, P],
    'st_decoder_packed_floats': [C.POINTER(StDecoderDims)],
    'st_decoder_gate_split_k': [C.POINTER(StDecoderDims)],
    'st_decoder_tape_floats': [C.POINTER(StDecoderDims), I],
    'st_decoder_pack': [C.POINTER(StDecoderWeights), C.POINTER(StDecoderDims), P, P],
    'st_decoder_forward': [C.POINTER(StDecoderWeights), C.POINTER(StDecoderDims), C.POINTER(StDecoderIO), P],
    'st_attn_step_bwd': [P, P, P, P, I, P, P, I, P, P, P, C.POINTER(P), C.POINTER(I), I, C.POINTER(P), C.POINTER(I), I,
                         P, P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    'st_attn_step_bwd_s': [P, P, P, P, I, P, P, I, P, P, P, C.POINTER(P), C.POINTER(I), I, C.POINTER(P), C.POINTER(I), I,
                           P, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    'st_attn_step_bwd_t16': [P, P, P, P, I, P, P, I, P, P, P, C.POINTER(P), C.POINTER(I), I, C.POINTER(P), C.POINTER(I), I,
                             P, P, I, P, C.POINTER(StT16View), P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    'st_skinny_linear_packed_lstm_bwd_fwd': [P, C.POINTER(StT16View), I, P, I, I, I, C.POINTER(StLstmPwJob), P],
    'st_skinny_linear_packed_lstm_bwd_attn_bwd': [P, C.POINTER(StT16View), I, P, I, I, I, C.POINTER(StLstmPwJob), C.POINTER(StAttnBwdJob), P],
    'st_attn_bwd_wide_fits': [I, I, I, I, I],
    'st_skinny_linear_packed_attn_hist': [P, C.POINTER(StT16View), I, P, I, I, I, C.POINTER(StAttnHistJob), P],
    'st_skinny_partial_attn_hist': [P, C.POINTER(StT16View), I, P, I, I, I, C.POINTER(StAttnHistJob), P],
    'st_decoder_bwd_forms': [C.POINTER(StDecoderDims), C.POINTER(StDecoderBwdIO)],
    'st_decoder_bwd_fuse_dims': [C.POINTER(StDecoderDims)],
    'st_loop_graph_stats': [P, P],
    'st_loop_graphs_enable': [I],
    'st_skinny_partial_attn_bwd': [P, C.POINTER(StT16View), I, P, I, I, I, C.POINTER(StAttnBwdJob), P],
    'st_skinny_linear_packed_lstm_bwd_attn_hist_sum': [P, C.POINTER(StT16View), I, P, I, I, I, C.POINTER(StLstmPwJob), C.POINTER(StAttnHistJob),
                                                       C.POINTER(StPartialSumJob), P],
    'st_skinny_linear_packed_lstm_bwd_attn_hist': [P, C.POINTER(StT16View), I, P, I, I, I, C.POINTER(StLstmPwJob), C.POINTER(StAttnHistJob), P],
    'st_lstm_seq2_fwd': [C.POINTER(P), C.POINTER(P), C.POINTER(P), P, I, C.POINTER(I), P, C.POINTER(P), C.POINTER(P), I, I, I, P],
    'st_lstm_seq2_persist_supported': [I, I, I, I, I, I],
    'st_lstm_seq2_persist_fwd': [C.POINTER(P), C.POINTER(P), C.POINTER(P), P, I, C.POINTER(I), C.POINTER(P), C.POINTER(P), I, I, I, P, P],
    'st_lstm_seq2_bwd_persist_supported': [I, I, I, I, I, I],
    'st_lstm_seq2_bwd_persist': [P, I, C.POINTER(I), C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), I, I, I, P, P],
    'st_lstm_seq2_bwd': [P, I, C.POINTER(I), C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), P, I, I, I, P],
    'st_skinny_linear_pair_fwd': [P, C.POINTER(P), I, I, I, P],
    'st_lstm_seq2_bwd_packed': [P, I, C.POINTER(I), C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), P, P, I, I, I, P],
    'st_skinny_linear_packed_lstm_bwd_pair_fwd': [C.POINTER(P), C.POINTER(StT16View), I, C.POINTER(P), I, I, I, C.POINTER(StLstmPwJob), P],
    'st_lstm_cell_pair_fwd': [P, C.POINTER(P), C.POINTER(P), I, C.POINTER(P), I, C.POINTER(P), I, C.POINTER(P), I, C.POINTER(P), I, I, P],
    'st_attn_dmem': [P, P, P, I, I, I, I, P],
    'st_decoder_backward': [C.POINTER(StDecoderBwdWeights), C.POINTER(StDecoderDims), C.POINTER(StDecoderBwdIO), P],
    'st_decoder_pack_dout': [P, P, P, I, I, I, I, I, I, P],
    'st_decoder_unpack_out': [P, P, P, I, I, I, I, I, I, P],
    'st_decoder_dteacher_sum': [P, P, I, I, I, I, I, I, I, P],
    'st_adain_bwd': [P, C.c_long, I, P, C.c_long, I, P, P, P, P, I, I, I, P],
    'st_mt_blocks': [P, I],
    'st_mt_table_misses': [],
    'st_mt_grad_norm': [P, P, I, P, P, P],
    'st_mt_clip_scale': [P, P, I, P, F, P],
    'st_mt_grad_norm_scaled': [P, P, I, P, P, F, P],
    'st_mt_clip_scale_pre': [P, P, I, P, F, F, P],
    'st_mt_copy': [P, P, P, I, P],
    'st_mt_adam': [P, P, P, P, P, I, F, F, F, F, F, P],
    'st_mt_adam_guarded': [P, P, P, P, P, I, F, F, F, F, F, P, P],
    'st_freq_loss': [P, P, P, P, P, I, I, I, I, F, F, F, I, P],
    'st_freq_loss_workspace_floats': [],
    'st_scale_by': [P, P, P, Z, P],
    'st_bn_norm_fwd': [P, I, I, P, I, I, I, I, P, P, P, P, F, I, P],
    'st_gemm_wgrad_workspace_floats': [I, I, I, I, I],
    'st_gemm_wgrad': [P, I, I, P, I, P, P, I, I, I, I, I, I, I, I, I, P],
    'st_gemm_wgrad_db': [P, I, I, P, I, P, P, P, I, I, I, I, I, I, I, I, I, P],
    'st_gemm_wgrad_batch_workspace_floats': [C.POINTER(StWgradJob), I],
    'st_gemm_wgrad_batch': [C.POINTER(StWgradJob), I, P, P],
    'st_gemm_wgrad_split': [P, I, I, P, I, P, I, P, P, P, P, I, I, I, I, P],
    'st_colsum': [P, I, I, P, I, I, I, I, P, I, P, P],
    'st_act_bwd': [P, I, P, I, I, P, I, P, I, I, I, P],
    'st_bn_bwd': [P, I, I, P, I, I, I, P, I, I, P, P, P, F, I, I, P, I, I, P, P, I, P, P],
    'st_bn_bwd_reduce': [P, I, I, P, I, I, I, P, I, I, P, P, F, I, I, P, P, P],
    'st_bn_norm_res_mask_fwd': [P, P, P, I, I, P, P, P, P, F, I, P, P, P],
    'st_bn_bwd_reduce_masked': [P, I, P, I, P, I, I, P, I, P, P, F, I, I, P, P, P],
    'st_bn_bwd_apply_masked': [P, I, P, I, P, I, I, P, I, P, P, P, F, I, I, P, I, P, P, I, P, I, P],
    'st_bn_bwd_apply': [P, I, I, P, I, I, I, P, I, I, P, P, P, F, I, I, P, I, P, I, I, P],
    'st_bn_stats_record': [P, I, I, I, I, P, P, P, P],
    'st_bn_sync_merge': [P, I, I, P, P, P, P, F, P, P],
    'st_bn_bwd_apply_sync': [P, I, I, P, I, I, I, P, I, I, P, P, P, F, I, I, P, P, P, I, I, P],
    'st_highway_fwd': [P, P, P, P, Z, P],
    'st_bn_bank_workspace_floats': [I, I, I],
    'st_bn_bank_fwd': [C.POINTER(StBnBankSeg), I, I, I, P, I, I, P, P],
    'st_bn_bank_bwd': [C.POINTER(StBnBankSeg), I, I, I, P, I, I, I, P, P],
    'st_bn_bank_stats_record': [C.POINTER(StBnBankSeg), I, I, I, P, P, P],
    'st_bn_bank_sync_merge': [C.POINTER(StBnBankSeg), I, I, I, P, I, P, P],
    'st_bn_bank_norm': [C.POINTER(StBnBankSeg), I, I, I, P, I, I, P],
    'st_bn_bank_bwd_reduce': [C.POINTER(StBnBankSeg), I, I, I, P, I, I, P, P],
    'st_bn_bank_bwd_apply': [C.POINTER(StBnBankSeg), I, I, I, P, I, I, I, P, P],
    'st_highway_ht_fwd': [P, P, P, I, I, P],
    'st_highway_ht_bwd': [P, P, P, P, P, I, I, P],
    'st_highway_bwd': [P, P, P, P, P, P, P, Z, P],
    'st_pool_prev_fwd': [P, P, I, I, I, P],
    'st_pool_prev_bwd': [P, P, P, I, I, I, P],
    'st_copy3d': [P, C.c_long, C.c_long, P, C.c_long, C.c_long, I, I, I, I, P],
    'st_scatter_add_rows': [P, P, P, I, I, I, P],
    'st_fill': [P, F, Z, P],
    'st_copy2d': [P, I, P, I, I, I, P],
    'st_mean_rows': [P, P, I, I, I, P],
}
_RESTYPES = {'st_last_error': C.c_char_p, 'st_packed_weight_floats': C.c_size_t, 'st_t16_floats': C.c_size_t,
             'st_decoder_packed_floats': C.c_size_t, 'st_vq_l2_workspace_floats': C.c_size_t, 'st_ctc_workspace_floats': C.c_size_t, 'st_decoder_tape_floats': C.c_size_t,
             'st_gemm_wgrad_workspace_floats': C.c_size_t, 'st_gemm_wgrad_batch_workspace_floats': C.c_size_t, 'st_freq_loss_workspace_floats': C.c_size_t, 'st_attn_fin_split_workspace_floats': C.c_size_t, 'st_attn_rng_xchg_words': C.c_size_t, 'st_colreduce_workspace_floats': C.c_size_t, 'st_mt_blocks': C.c_size_t, 'st_mt_table_misses': C.c_long,
             'st_bn_bank_workspace_floats': C.c_size_t}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64 (NEEDED as "libamdhip64.so"); ours is linked against
    # "libamdhip64.so.7".  Importing torch first makes the loader resolve both to ONE HIP
    # runtime -- loading ours first would put two runtimes in the process (hipErrorNoDevice).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libsemitts_hip.so not found at %s -- build it with `python -m semi_tts_amd.build` '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    if lib.st_abi_version() != 1:
        raise RuntimeError('libsemitts_hip.so ABI version mismatch')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().st_last_error()
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))


def check_header_symbols():
    """every function include/semitts.h declares is exported by the library and bound by SIGNATURES (and nothing else is)"""
    import re
    hdr = open(os.path.join(os.path.dirname(HERE), 'include', 'semitts.h')).read()
    declared = sorted(set(re.findall(r'\b(st_[a-z0-9_]+)\s*\(', hdr)))
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    if missing:
        raise RuntimeError('declared in include/semitts.h but not exported by %s: %s' % (LIB_PATH, missing))
    if sorted(SIGNATURES) != declared:
        raise RuntimeError('ctypes table and include/semitts.h disagree: %s' % sorted(set(SIGNATURES) ^ set(declared)))
    return declared
