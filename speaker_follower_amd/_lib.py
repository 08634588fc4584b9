"""ctypes binding of libsf_hip.so (C ABI in include/sf_hip.h).

There is deliberately NO fallback: if the library is missing or an entry point is
absent, importing this module raises.  A non-zero status from any call raises
SfError.  PyTorch only supplies device memory (`tensor.data_ptr()`) and the
current HIP stream.
"""
import ctypes as C
import os

# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as the system
# one).  It must be the first HIP runtime loaded into the process, otherwise libsf_hip.so pulls in
# /opt/rocm's copy and torch -- which provides device memory and streams -- runs on a runtime it was
# not built against.  Importing torch first makes both share torch's runtime.
import torch  # noqa: F401  (load order matters)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libsf_hip.so')

c_f = C.c_void_p          # device float*
c_p = C.c_void_p


class SfError(RuntimeError):
    pass


class Dropout(C.Structure):
    _fields_ = [('p', C.c_float), ('seed', C.c_uint32), ('row0', C.c_int32),
                ('site_dev', c_p), ('site_mul', C.c_uint32)]        # ABI 8: device-side site offset (optional)


class Pano(C.Structure):
    _fields_ = [('dense', c_p), ('table', c_p), ('loc_table', c_p), ('vp', c_p), ('view', c_p),
                ('V', C.c_int32), ('IMG', C.c_int32), ('LOC', C.c_int32)]


class Cands(C.Structure):
    _fields_ = [('dense', c_p), ('table', c_p), ('vp', c_p), ('cand_view', c_p),
                ('cand_sincos', c_p), ('a_num', c_p),
                ('A', C.c_int32), ('V', C.c_int32), ('IMG', C.c_int32), ('LOC', C.c_int32)]


def _ptr_struct(name, fields):
    return type(name, (C.Structure,), {'_fields_': [(f, c_p) for f in fields]})


LstmW = _ptr_struct('LstmW', ['w_ih', 'w_hh', 'b_ih', 'b_hh', 'w_ih_t', 'w_hh_t'])
VisualW = _ptr_struct('VisualW', ['w_h', 'b_h', 'w_v', 'b_v', 'w_v_t', 'w_h_t'])
SoftdotW = _ptr_struct('SoftdotW', ['w_in', 'w_out', 'w_in_t', 'w_out_t'])
ScoringW = _ptr_struct('ScoringW', ['w_h', 'b_h', 'w_a', 'b_a', 'w_out', 'b_out', 'w_a_t', 'w_h_t'])


DecoderFold = _ptr_struct('DecoderFold', ['m_v', 'c_v', 'm_a', 'c_a'])


class DecoderW(C.Structure):
    _fields_ = [('lstm', LstmW), ('visual', VisualW), ('text', SoftdotW), ('action', ScoringW),
                ('fold', C.POINTER(DecoderFold))]


DecoderGTape = _ptr_struct('DecoderGTape', ['dgates', 'dpre', 'dt_text', 'dt_v', 'dq', 'dwt', 'dta', 'dr',
                                            'dc', 'dcat2', 'ds', 'dh1d'])
DecoderTape = _ptr_struct('DecoderTape', ['t_v', 'q', 'alpha_v', 'xin', 'gates', 'c1', 'h1', 'cat2',
                                          't_text', 'alpha', 'h_tilde', 't_a', 'wt', 'r', 'logit'])


class FollowerGlue(C.Structure):
    _fields_ = [('is_valid', c_p), ('target', c_p), ('feedback', C.c_int32), ('ended', c_p),
                ('a_t', c_p), ('target_used', c_p), ('score', c_p), ('u_next', c_p),
                ('ld_u_next', C.c_int32), ('u_drop', C.POINTER(Dropout)), ('u_drop_stream', C.c_uint32),
                ('ce_term', c_p), ('live', c_p), ('sample_seed', C.c_uint32),
                ('sample_stream', C.c_uint32), ('row0', C.c_int32), ('sample_site_dev', c_p), ('nav', c_p)]


class FollowerEpisode(C.Structure):
    _fields_ = [('S', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('D', C.c_int32),
                ('L', C.c_int32), ('A', C.c_int32), ('X', Pano), ('U', Cands), ('h_init', c_p),
                ('c_init', c_p), ('ctx', c_p), ('ctx_mask', c_p), ('tape', DecoderTape),
                ('glue', FollowerGlue), ('drop', Dropout), ('step0', C.c_uint32),
                ('side_stream', C.c_void_p),
                ('ctx_q', c_p), ('ctx_o', c_p),                     # ABI 9: folded text attention (inference only)
                ('chain_fold', c_p)]                                # ... + folded query / scoring products (sf_decoder_fold*)


class EncoderW(C.Structure):
    _fields_ = [('embedding', c_p), ('lstm', LstmW), ('w_e2d', c_p), ('b_e2d', c_p), ('w_e2d_t', c_p),
                ('xw_table', c_p), ('flags', C.c_int32)]


class EncoderG(C.Structure):
    _fields_ = [('lstm', LstmW), ('w_e2d', c_p), ('b_e2d', c_p), ('embedding', c_p), ('seq', c_p),
                ('Lpad', C.c_int32), ('padding_idx', C.c_int32)]


EncoderTape = _ptr_struct('EncoderTape', ['emb', 'xg', 'gates', 'hs', 'cs'])


class SpkDecoderW(C.Structure):
    _fields_ = [('embedding', c_p), ('lstm', LstmW), ('attn', SoftdotW), ('w_out', c_p),
                ('b_out', c_p), ('xw_table', c_p), ('flags', C.c_int32), ('w_out_t', c_p)]


class SpkDecoderG(C.Structure):
    _fields_ = [('lstm', LstmW), ('attn', SoftdotW), ('w_out', c_p), ('b_out', c_p), ('embedding', c_p)]


class SpkDecoderGTape(C.Structure):
    """sf_spk_decoder_gtape: stacked [S][B][..] dY operands of the speaker's word-loop backward."""
    _fields_ = [(n, C.c_void_p) for n in ('dlogit', 'dpre', 'dt_text', 'dgates')]


class Sample(C.Structure):
    """sf_sample: counter-based draw of the speaker's `sample` feedback (seed, stream, row0)."""
    _fields_ = [('seed', C.c_uint32), ('stream', C.c_uint32), ('row0', C.c_int32), ('stream_dev', c_p)]


class VisualFold64(C.Structure):
    """sf_visual_fold64."""
    _fields_ = [('m_v', c_p), ('c_v', c_p)]


class RowMove(C.Structure):
    """sf_row_move."""
    _fields_ = [('src', c_p), ('dst', c_p), ('idx', c_p), ('ld_src', C.c_int32), ('ld_dst', C.c_int32),
                ('width', C.c_int32), ('scatter', C.c_int32)]


class FillRegion(C.Structure):
    """sf_fill_region."""
    _fields_ = [('ptr', c_p), ('count', C.c_uint64), ('value', C.c_uint64), ('width', C.c_int32)]


class NavTableS(C.Structure):
    _fields_ = [('a_num', c_p), ('next_row', c_p), ('cand_view', c_p), ('cand_sincos', c_p),
                ('feat_row', c_p), ('A', C.c_int32), ('V', C.c_int32)]


class NavIO(C.Structure):
    _fields_ = [('nav', NavTableS), ('row', c_p), ('view', c_p), ('goal_hop', c_p), ('ld_hop', C.c_int32),
                ('hop_base', c_p), ('row_next', c_p), ('vp_next', c_p), ('view_next', c_p),
                ('a_num_next', c_p), ('cand_view_next', c_p), ('sincos_next', c_p), ('target_next', c_p)]


SpkDecoderTape = _ptr_struct('SpkDecoderTape', ['emb', 'gates', 'c1', 'h1', 'cat2', 't_text',
                                                'alpha', 'h_tilde', 'logit'])

i32, u32, i64p = C.c_int, C.c_uint32, C.c_void_p
P = C.POINTER
WS = [c_p, C.c_size_t, c_p]          # ws, ws_bytes, stream

_SIGNATURES = {
    'sf_workspace_bytes': (C.c_size_t, []),
    'sf_wgrad_workspace_bytes': (C.c_size_t, []),
    'sf_abi_version': (C.c_int, []),
    'sf_build_id': (C.c_char_p, []),
    'sf_debug_persist_timeout': (None, [C.c_longlong]),
    'sf_debug_gate_product_f32': (None, [C.c_int]),
    'sf_debug_fold_merge_with_glue': (None, [C.c_int]),
    'sf_debug_fold_chain3': (None, [C.c_int]),
    'sf_debug_fold_build_overlap': (None, [C.c_int]),
    'sf_debug_precise_attention': (None, [C.c_int]),
    'sf_debug_many_row_product': (None, [C.c_int]),
    'sf_debug_grouped_weight_gradients': (None, [C.c_int]),
    'sf_debug_slab_consumers': (None, [C.c_int]),
    'sf_debug_fused_cell_backward': (None, [C.c_int]),
    'sf_debug_bptt_lookahead': (None, [C.c_int]),
    'sf_debug_bptt_flags': (None, [C.c_int]),
    'sf_debug_bptt_part': (None, [C.c_int]),
    'sf_site_advance': (C.c_int, [c_p, u32, c_p]),
    'sf_store_u32x4': (C.c_int, [c_p, u32, u32, u32, u32, c_p]),
    'sf_adam_step_dev': (C.c_int, [c_f, c_f, c_f, c_f, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_double, c_p, c_f, c_p, c_p]),
    'sf_debug_cotenant': (C.c_int, [i32, i32, i32, C.c_longlong, c_f, c_p]),
    'sf_gate_product_strict': (None, [C.c_int]),
    'sf_gate_product_is_strict': (C.c_int, []),
    'sf_debug_tn_split_min_rows': (None, [C.c_int]),
    'sf_workspace_fault_offset': (C.c_size_t, [C.c_size_t]),
    'sf_debug_trace': (None, [C.c_void_p]),
    'sf_debug_force_write_through': (None, [C.c_int]),
    'sf_status_string': (C.c_char_p, [C.c_int]),
    'sf_last_error_string': (C.c_char_p, []),
    'sf_linear_fwd': (C.c_int, [c_f, i32, c_f, c_f, i32, i32, i32, i32, c_f, i32] + WS),
    'sf_linear_slabs_fwd': (C.c_int, [c_f, i32, c_f, i32, c_f, i32, c_f, i32, i32, i32, P(C.c_int)] + WS),
    'sf_linear_bwd': (C.c_int, [c_f, i32, c_f, c_f, i32, c_f, i32, i32, i32, i32, i32, c_f, i32,
                                i32, c_f, c_f] + WS),
    'sf_lstm_cell_fwd': (C.c_int, [P(LstmW), i32, i32, i32, c_f, i32, c_f, c_f, c_f, c_f, c_f, c_f,
                                   i32, P(Dropout), u32] + WS),
    'sf_lstm_cell_bwd': (C.c_int, [P(LstmW), P(LstmW), i32, i32, i32, c_f, i32, c_f, c_f, c_f, c_f,
                                   c_f, c_f, c_f, i32, c_f, c_f] + WS),
    'sf_visual_attention_fwd': (C.c_int, [P(VisualW), P(Pano), i32, i32, i32, c_f, c_f, i32, c_f,
                                          c_f, c_f, P(Dropout), u32, i32] + WS),
    'sf_visual_attention_fwd_f64': (C.c_int, [P(VisualW), P(Pano), i32, i32, i32, c_f, c_f, i32, c_f,
                                              c_f, c_f, P(Dropout), u32, i32] + WS),
    'sf_linear_f64': (C.c_int, [c_f, i32, c_f, i32, c_f, i32, i32, i32, c_p, c_f, c_p]),
    'sf_visual_attention_bwd': (C.c_int, [P(VisualW), P(VisualW), P(Pano), i32, i32, i32, c_f, c_f,
                                          c_f, c_f, i32, P(Dropout), u32, i32, c_f] + WS),
    'sf_soft_dot_attention_fwd': (C.c_int, [P(SoftdotW), i32, i32, i32, c_f, i32, c_f, c_p, c_p,
                                            c_f, c_f, c_f, c_f] + WS),
    'sf_soft_dot_attention_bwd': (C.c_int, [P(SoftdotW), P(SoftdotW), i32, i32, i32, c_f, c_f, c_f,
                                            c_f, c_f, c_f, c_f, i32, c_f] + WS),
    'sf_text_attention_fwd': (C.c_int, [c_f, c_p, i32, i32, i32, c_f, i32, c_f, c_f, i32, c_p]),
    'sf_text_attention_bwd': (C.c_int, [c_f, i32, i32, i32, c_f, i32, c_f, i32, c_f, c_f, i32, c_f, c_p]),
    'sf_eltwise_prod_scoring_fwd': (C.c_int, [P(ScoringW), P(Cands), i32, i32, i32, c_f, c_f, c_f,
                                              c_f, c_f] + WS),
    'sf_eltwise_prod_scoring_bwd': (C.c_int, [P(ScoringW), P(ScoringW), P(Cands), i32, i32, i32,
                                              c_f, c_f, c_f, c_f, c_f] + WS),
    'sf_attn_decoder_fwd': (C.c_int, [P(DecoderW), P(Pano), P(Cands), i32, i32, i32, i32, c_f, c_f,
                                      c_f, c_f, c_p, c_p, P(DecoderTape), P(FollowerGlue),
                                      P(Dropout), u32] + WS),
    'sf_attn_decoder_head_fwd': (C.c_int, [P(DecoderW), P(Pano), i32, i32, i32, c_f, P(DecoderTape),
                                           P(Dropout), u32] + WS),
    'sf_attn_decoder_tail_fwd': (C.c_int, [P(DecoderW), P(Cands), i32, i32, i32, i32, c_f, c_f, c_f,
                                           c_f, c_p, c_p, P(DecoderTape), P(FollowerGlue), P(Dropout),
                                           u32, P(Pano), P(DecoderTape)] + WS),
    'sf_attn_decoder_attend_fwd': (C.c_int, [P(Pano), i32, P(DecoderTape), P(Dropout), u32] + WS),
    'sf_follower_episode_fwd': (C.c_int, [P(DecoderW), P(FollowerEpisode)] + WS),
    'sf_follower_episode_bwd': (C.c_int, [P(DecoderW), P(FollowerEpisode), P(DecoderGTape), c_f, c_f,
                                          c_f, c_f, c_f, c_f, c_f, P(C.c_int)] + WS),
    'sf_follower_episode_bwd_range': (C.c_int, [P(DecoderW), P(FollowerEpisode), P(DecoderGTape), c_f, c_f,
                                                c_f, c_f, c_f, c_f, c_f, P(C.c_int), i32, i32, c_f, c_f] + WS),
    'sf_attn_decoder_bwd': (C.c_int, [P(DecoderW), P(DecoderW), P(Pano), P(Cands), i32, i32, i32,
                                      i32, c_f, c_f, c_f, P(DecoderTape), P(DecoderGTape), c_f, c_f,
                                      c_f, c_f, c_f, c_f, P(Dropout), u32] + WS),
    'sf_attn_decoder_wgrad': (C.c_int, [P(DecoderW), P(DecoderW), i32, i32, i32, i32, c_f,
                                        P(DecoderTape), P(DecoderGTape)] + WS),
    'sf_decoder_fold_build': (C.c_int, [P(DecoderW), i32, i32, i32, c_f, c_f, c_f, c_f] + WS),
    'sf_follower_glue_fwd': (C.c_int, [P(Cands), i32, c_f, P(FollowerGlue), c_p]),
    'sf_follower_glue_bwd': (C.c_int, [i32, i32, c_f, i64p, c_f, c_f, c_p]),
    'sf_reduce_terms': (C.c_int, [c_f, c_f, i32, i32, c_f, c_p]),
    'sf_loss_finalize': (C.c_int, [c_f, i32, c_f, c_f, c_p]),
    'sf_encoder_lstm_fwd': (C.c_int, [P(EncoderW), i32, i32, i32, i32, i32, i64p, c_p, c_f, c_f,
                                      c_f, P(EncoderTape), P(Dropout), u32] + WS),
    'sf_encoder_lstm_bwd': (C.c_int, [P(EncoderW), P(EncoderG), i32, i32, i32, i32, c_p, c_f, c_f,
                                      c_f, c_f, P(EncoderTape), P(Dropout), u32] + WS),
    'sf_gather_panorama': (C.c_int, [P(Pano), i32, c_f, c_p]),
    'sf_gather_candidates': (C.c_int, [P(Cands), i32, c_f, c_f, c_p]),
    'sf_gather_actions': (C.c_int, [P(Cands), i32, c_p, c_f, c_p]),
    'sf_gather_actions_ld': (C.c_int, [P(Cands), i32, c_p, c_f, i32, c_p]),
    'sf_gather_path_actions': (C.c_int, [c_f, i32, i32, i32, c_p, c_p, c_f, c_p, i32, c_f, i32, c_p]),
    'sf_gather_rows': (C.c_int, [c_f, i32, c_p, i32, i32, c_f, i32, c_p]),
    'sf_scatter_rows': (C.c_int, [c_f, i32, c_p, i32, i32, c_f, i32, c_p]),
    'sf_logprob_topk': (C.c_int, [c_f, i32, i32, i32, c_p, i32, c_p, c_f, c_p]),
    'sf_speaker_decoder_fwd': (C.c_int, [P(SpkDecoderW), i32, i32, i32, i32, i32, i64p, c_f, c_f,
                                         c_f, c_p, c_p, P(SpkDecoderTape), P(Dropout), u32] + WS),
    'sf_speaker_decoder_bwd': (C.c_int, [P(SpkDecoderW), P(SpkDecoderG), i32, i32, i32, i32, i32, i64p,
                                         c_f, c_f, c_f, P(SpkDecoderTape), c_f, c_f, c_f, c_f, c_f,
                                         c_f, P(Dropout), u32] + WS),
    'sf_speaker_decode': (C.c_int, [P(SpkDecoderW), i32, i32, i32, i32, i32, i32, i32, i32, i64p, c_f, c_f, c_f, c_p,
                                   i64p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, P(Sample)] + WS),
    'sf_speaker_glue_fwd': (C.c_int, [i32, i32, i32, c_f, i64p, i32, i32, i32, c_p, i64p, c_f, c_f,
                                      c_f, P(Sample), c_p]),
    'sf_speaker_loss_finalize': (C.c_int, [c_f, i64p, i32, i32, i32, c_f, c_f, c_p]),
    'sf_speaker_glue_bwd': (C.c_int, [i32, i32, i32, c_f, i64p, i32, c_f, c_f, c_p]),
    'sf_speaker_encoder_fwd': (C.c_int, [P(VisualW), P(LstmW), c_f, c_f, P(Pano), i32, i32, i32, i32, c_f, c_f, c_f, c_f,
                                         c_f, c_f, c_f, c_f, c_f, c_f, P(Dropout), u32] + WS),
    'sf_speaker_encoder_fwd_folded': (C.c_int, [P(VisualFold64), P(VisualW), P(LstmW), c_f, c_f, P(Pano), i32, i32, i32, i32,
                                                c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, P(Dropout), u32] + WS),
    'sf_visual_query_fold_f64': (C.c_int, [P(VisualW), i32, i32, i32, c_p, c_p, c_p]),
    'sf_speaker_words_fwd': (C.c_int, [P(SpkDecoderW), i32, i32, i32, i32, i32, i32, i32, i32, i32, i64p, c_f, c_f, c_f,
                                       c_p, i64p, c_p, c_f, c_f, c_f, P(SpkDecoderTape), P(Dropout), u32, P(Sample)] + WS),
    'sf_speaker_words_bwd': (C.c_int, [P(SpkDecoderW), P(SpkDecoderG), i32, i32, i32, i32, i32, i32, i32, i64p, i64p, c_f,
                                       c_f, c_f, P(SpkDecoderTape), c_f, c_f, c_f, c_f, c_f, c_f, c_f, P(C.c_int),
                                       P(Dropout), u32, P(SpkDecoderGTape), c_f] + WS),
    'sf_speaker_teacher_fwd': (C.c_int, [P(SpkDecoderW), i32, i32, i32, i32, i32, i32, i32, i32, i64p, c_f, c_f, c_f, c_p,
                                         i64p, c_p, c_f, c_f, c_f, P(SpkDecoderTape), P(Dropout), u32] + WS),
    'sf_speaker_teacher_bwd': (C.c_int, [P(SpkDecoderW), P(SpkDecoderG), i32, i32, i32, i32, i32, i32, i32, i64p, i64p, c_f,
                                         c_f, c_f, P(SpkDecoderTape), c_f, c_f, c_f, c_f, P(Dropout), u32,
                                         P(SpkDecoderGTape), c_f, c_f, c_f] + WS),
    'sf_fill_f32': (C.c_int, [c_f, C.c_size_t, C.c_float, c_p]),
    'sf_fill_regions': (C.c_int, [c_p, C.c_int, c_p]),
    'sf_move_rows': (C.c_int, [c_p, C.c_int, C.c_int, c_p]),
    'sf_add_f32': (C.c_int, [c_f, c_f, C.c_size_t, c_p]),
    'sf_adam_step': (C.c_int, [c_f, c_f, c_f, c_f, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double,
                              C.c_double, C.c_int, c_p]),
    'sf_dropout_copy': (C.c_int, [c_f, i32, i32, i32, c_f, i32, P(Dropout), u32, i32, c_p]),
    'sf_embedding_fwd': (C.c_int, [c_f, i32, i64p, i32, c_f, c_p]),
    'sf_transpose': (C.c_int, [c_f, i32, i32, c_f, c_p]),
    'sf_nav_step': (C.c_int, [P(NavTableS), i32] + [c_p] * 5 + [i32] + [c_p] * 8 + [c_p]),
    'sf_profile_begin': (C.c_int, []),
    'sf_profile_end': (C.c_long, [C.c_char_p, C.c_size_t]),
}

SF_OK, SF_ERR_ARG, SF_ERR_UNSUPPORTED, SF_ERR_LAUNCH, SF_ERR_WORKSPACE = 0, 1, 2, 3, 4
SF_ENC_PER_STEP = 1           # sf_encoder_w.flags
SF_ENC_EMB_DROPOUT = 2
SF_ENC_RAW_STATE = 4
SF_ENC_REVERSED = 8
SF_SPK_EMB_DROPOUT = 1        # sf_spk_decoder_w.flags

EXPORTS = tuple(_SIGNATURES)
ABI_VERSION = 9


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libsf_hip.so is missing (%s). Build it with `python -m speaker_follower_amd.build`; '
            'there is no CPU fallback for the speaker/follower hot path.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sf_abi_version() != ABI_VERSION:
        raise ImportError('libsf_hip.so ABI version mismatch')
    # a library built from other sources than the ones on disk (stale .so after a checkout) is refused
    from .build import build_id, CSRC
    if os.path.isdir(CSRC) and os.environ.get('SF_SKIP_BUILD_ID_CHECK') != '1':
        have, want = lib.sf_build_id().decode(), build_id()
        if have != want:
            raise ImportError('libsf_hip.so is stale: built from sources %s, the tree holds %s; run '
                              '`python -m speaker_follower_amd.build`' % (have, want))
    return lib


lib = _load()

def check(status, what=''):
    if status != 0:
        detail = lib.sf_status_string(status).decode()
        if status == 3:
            detail += ': ' + lib.sf_last_error_string().decode()
        raise SfError('%s failed: %s (status %d)' % (what or 'libsf_hip call', detail, status))


def call(name, *args):
    check(getattr(lib, name)(*args), name)


class kernel_profile:
    """`with kernel_profile() as prof: ...` times every kernel this thread launches through the
    library (sf_profile_begin / sf_profile_end); afterwards prof.rows maps the kernel name to
    dict(calls, total_us, avg_us, min_us, max_us).  Eager issue only (not under graph capture)."""

    def __init__(self, library=None):
        self.library = library if library is not None else lib

    def __enter__(self):
        check(self.library.sf_profile_begin(), 'sf_profile_begin')
        self.rows = {}
        return self

    def __exit__(self, *exc):
        buf = C.create_string_buffer(1 << 16)
        n = self.library.sf_profile_end(buf, len(buf))
        if n < 0:
            raise SfError('sf_profile_end failed')
        for line in buf.value.decode().splitlines():
            name, calls, total, mn, mx = line.split('\t')
            name = name.strip('()')
            self.rows[name] = dict(calls=int(calls), total_us=float(total), avg_us=float(total) / int(calls),
                                   min_us=float(mn), max_us=float(mx))
        return False
