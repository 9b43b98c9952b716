"""ctypes binding of lib3dinfomax_hip.so - the C ABI declared in include/infomax3d_hip.h.

The product path has NO fallback: if the HIP library is missing or a call fails, it raises.
"""
import ctypes
import os
import re
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_long, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# test hooks (tests/test_gpu_dist.py), behind ONE flag and read ONCE, here: I3D_TESTING=1 makes I3D_TEST_PEER_SELFTEST_FAIL /
# I3D_TEST_PEER_FAIL (a rank number) and I3D_TEST_FORCE_EARLY_ALLREDUCE (1) effective; without the flag a stray variable changes nothing
_TESTING = os.environ.get('I3D_TESTING') == '1'
TEST_HOOKS = dict(peer_selftest_fail=os.environ.get('I3D_TEST_PEER_SELFTEST_FAIL') if _TESTING else None,
                  peer_fail=os.environ.get('I3D_TEST_PEER_FAIL') if _TESTING else None,
                  force_early_allreduce=_TESTING and os.environ.get('I3D_TEST_FORCE_EARLY_ALLREDUCE') == '1',
                  verbose_selftest=_TESTING and bool(os.environ.get('I3D_DEBUG_SELFTEST')))
LIB_PATH = os.environ.get('I3D_LIB_PATH') or os.path.join(HERE, 'lib', 'lib3dinfomax_hip.so')      # (override: kernel probes of tools/probes)
HEADER_PATH = os.path.join(os.path.dirname(HERE), 'include', 'infomax3d_hip.h')

# constants of include/infomax3d_hip.h
ACT = {'none': 0, None: 0, 'relu': 1, 'silu': 2, 'sigmoid': 3, 'leakyrelu': 4, 'tanh': 5, 'elu': 6, 'selu': 7, 'softplus': 8}
AGG = {'mean': 0, 'sum': 1, 'max': 2, 'min': 3, 'std': 4, 'var': 5}
SCALER = {'identity': 0, 'amplification': 1, 'attenuation': 2}

_P = c_void_p


# mirrors of the argument structs of include/infomax3d_hip.h (composites)
class BnTail(ctypes.Structure):
    _fields_ = [('act', c_int), ('post_act', c_int), ('eps', c_float), ('momentum', c_float), ('gamma', _P),
                ('beta', _P), ('running_mean', _P), ('running_var', _P), ('mean', _P), ('invstd', _P),
                ('workspace', _P), ('gemm_workspace', _P), ('gemm_workspace_bytes', c_long), ('num_batches_tracked', _P),
                ('bias_partial', _P)]


class FcArgs(ctypes.Structure):
    _fields_ = [('tail', BnTail), ('rows', c_int), ('f_in', c_int), ('f_out', c_int), ('ldw', c_int), ('x', _P),
                ('W', _P), ('bias', _P), ('residual', _P), ('xact', _P), ('pre_keep', _P), ('y', _P), ('grad_y', _P),
                ('grad_pre', _P), ('grad_gamma', _P), ('grad_beta', _P), ('grad_W', _P), ('grad_bias', _P),
                ('grad_x', _P), ('W_dgrad_panel', _P), ('W_fwd_panel', _P)]


class EdgeFcArgs(ctypes.Structure):
    _fields_ = [('tail', BnTail), ('num_nodes', c_int), ('num_edges', c_int), ('f_h', c_int), ('f_q', c_int),
                ('f_out', c_int), ('ldw', c_int), ('q_rows', c_int), ('v_pad', c_int), ('q_code', _P), ('onehot', _P),
                ('grad_Q', _P), ('h', _P), ('q', _P), ('W', _P), ('bias', _P), ('src_s', _P),
                ('dst_s', _P), ('in_ptr', _P), ('out_ptr', _P), ('out_epos', _P), ('P', _P), ('Q', _P), ('xact', _P),
                ('pre_keep', _P), ('y', _P), ('grad_y', _P), ('grad_pre', _P), ('grad_P', _P), ('grad_gamma', _P),
                ('grad_beta', _P), ('grad_W', _P), ('grad_bias', _P), ('grad_h', _P), ('grad_q', _P),
                ('grad_q_accumulate', c_int)]


class GroupedFcArgs(ctypes.Structure):
    _fields_ = [('tail', BnTail), ('num_nodes', c_int), ('f_h', c_int), ('f_out', c_int), ('agg_width', c_int),
                ('ldw', c_int), ('n_groups', c_int), ('n_scalers', c_int), ('m_padded', c_int),
                ('group_start', c_int * 32), ('group_count', c_int * 32), ('coef', c_float * 128), ('h', _P),
                ('agg', _P), ('W', _P), ('bias', _P), ('residual', _P), ('deg_rows', _P), ('deg_tile_group', _P),
                ('WD', _P), ('xact', _P), ('pre_keep', _P), ('y', _P), ('grad_y', _P), ('grad_pre', _P),
                ('grad_WD', _P), ('grad_gamma', _P), ('grad_beta', _P), ('grad_W', _P), ('grad_bias', _P),
                ('grad_h', _P), ('grad_agg', _P)]


class PnaLayerArgs(ctypes.Structure):
    _fields_ = [('edge', EdgeFcArgs), ('n_pre_extra', c_int), ('pre', FcArgs * 3), ('n_aggregators', c_int),
                ('n_scalers', c_int), ('force_scalers', c_int), ('aggregators', c_int * 8), ('scalers', c_int * 4),
                ('avg_d_log', c_float), ('msg', _P), ('grad_msg', _P), ('post', GroupedFcArgs), ('n_post_extra', c_int),
                ('postx', FcArgs * 3), ('residual', c_int), ('grad_out', _P), ('agg_event_start', _P), ('agg_event_stop', _P),
                ('fused_bn', c_int), ('defer_join', c_int), ('stats_ws', _P), ('aff', _P * 4), ('weights_ready', c_int),
                ('merge_h', c_int), ('Wcat', _P), ('bcat', _P), ('PL', _P), ('DL', _P), ('wgrad_split', c_int), ('eval_mode', c_int), ('msg_bf16', c_int),
                ('edge_bias_partial', _P), ('Wcat_panel', _P), ('Wcat_dgrad_panel', _P)]


class Net3dEdgeArgs(ctypes.Structure):
    _fields_ = [('tail_in', BnTail), ('tail_msg', BnTail), ('num_nodes', c_int), ('num_edges', c_int), ('hidden', c_int),
                ('n_enc', c_int), ('reduce_mean', c_int), ('ld_w_in', c_int), ('ld_w_msg', c_int), ('d_raw', _P), ('perm', _P),
                ('dst_s', _P), ('in_ptr', _P), ('emb', _P), ('W_in', _P), ('b_in', _P), ('W_msg', _P), ('b_msg', _P),
                ('w_gate', _P), ('b_gate', _P), ('stats', _P), ('aff_in', _P), ('aff_msg', _P), ('x_msg', _P), ('d_out', _P),
                ('msg', _P), ('m_sum', _P), ('grad_m_sum', _P), ('grad_ya', _P), ('grad_lin', _P), ('partial', _P), ('grad_W_in', _P),
                ('grad_b_in', _P), ('grad_gamma_in', _P), ('grad_beta_in', _P), ('grad_W_msg', _P), ('grad_b_msg', _P),
                ('grad_gamma_msg', _P), ('grad_beta_msg', _P), ('grad_w_gate', _P), ('grad_b_gate', _P), ('grad_emb', _P), ('store_bf16', c_int), ('x_center', _P)]


class FcParams(ctypes.Structure):
    _fields_ = [('W', _P), ('bias', _P), ('gamma', _P), ('beta', _P), ('running_mean', _P), ('running_var', _P),
                ('num_batches_tracked', _P), ('grad_W', _P), ('grad_bias', _P), ('grad_gamma', _P), ('grad_beta', _P),
                ('f_in', c_int), ('f_out', c_int), ('act', c_int), ('eps', c_float), ('momentum', c_float)]


class PnaModel(ctypes.Structure):
    _fields_ = [('training', c_int), ('n_layers', c_int), ('hidden', c_int), ('n_pre', c_int), ('residual', c_int), ('n_aggregators', c_int),
                ('aggregators', c_int * 8), ('n_scalers', c_int), ('scalers', c_int * 4), ('avg_d_log', c_float),
                ('pre', (FcParams * 4) * 16), ('post', FcParams * 16), ('n_atom_tables', c_int), ('atom_dims', c_int * 16),
                ('atom_tables', _P * 16), ('grad_atom_tables', _P), ('n_bond_tables', c_int), ('bond_dims', c_int * 16),
                ('bond_tables', _P * 16), ('grad_bond_tables', _P), ('n_readout', c_int), ('readout_ops', c_int * 4),
                ('n_head', c_int), ('head', FcParams * 4)]


class PnaBatch(ctypes.Structure):
    _fields_ = [('num_nodes', c_int), ('num_edges', c_int), ('num_graphs', c_int), ('atom_feat', _P), ('bond_feat', _P),
                ('in_ptr', _P), ('perm', _P), ('src_s', _P), ('dst_s', _P), ('out_ptr', _P), ('out_epos', _P),
                ('graph_ptr', _P), ('deg_rows', _P), ('deg_tile_group', _P), ('m_padded', c_int), ('n_groups', c_int),
                ('group_degree', c_int * 32), ('group_start', c_int * 32), ('group_count', c_int * 32), ('comb', _P),
                ('n_comb', c_int), ('v_pad', c_int)]


class BnEvalAff(ctypes.Structure):
    _fields_ = [('running_mean', _P), ('running_var', _P), ('gamma', _P), ('beta', _P), ('aff', _P), ('feat', c_int),
                ('eps', c_float)]


class CopyBlock(ctypes.Structure):
    _fields_ = [('src', _P), ('dst', _P), ('rows', c_int), ('cols', c_int), ('ld_src', c_long), ('ld_dst', c_long)]


class TowerLayerArgs(ctypes.Structure):
    _fields_ = ([(n, c_int) for n in ('num_nodes', 'num_edges', 'f_in', 'f_edge', 'f_msg', 'f_out', 'f_mix', 'ldp', 'ldq', 'ldgp',
                                      'ldgq', 'n_aggregators', 'n_scalers', 'residual', 'training', 'grad_e_accumulate')]
                + [('aggregators', c_int * 8), ('scalers', c_int * 4), ('avg_d_log', c_float), ('eps', c_float), ('momentum', c_float)]
                + [(n, _P) for n in ('h', 'e', 'snorm', 'Wp', 'bp', 'Wq', 'bq', 'gamma', 'beta', 'running_mean', 'running_var', 'Wm',
                                     'bm', 'src_s', 'dst_s', 'in_ptr', 'out_ptr', 'out_epos', 'saved', 'scratch', 'workspace',
                                     'gemm_workspace')]
                + [('gemm_workspace_bytes', c_long)]
                + [(n, _P) for n in ('out', 'grad_out', 'grad_h', 'grad_e', 'grad_Wp', 'grad_bp', 'grad_Wq', 'grad_bq', 'grad_gamma',
                                     'grad_beta', 'grad_Wm', 'grad_bm')]
                + [('n_towers', c_int), ('n_deg_groups', c_int), ('m_padded', c_int), ('group_start', c_int * 32),
                   ('group_count', c_int * 32), ('coef', c_float * 128), ('deg_rows', _P), ('deg_tile_group', _P)])


ALL_GATHER_F32 = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_long, c_void_p)
ALL_REDUCE_F64 = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_long, c_void_p)


class Collectives(ctypes.Structure):
    _fields_ = [('world', c_int), ('all_gather_f32', ALL_GATHER_F32), ('all_reduce_f64', ALL_REDUCE_F64), ('user', _P),
                ('scratch', _P), ('scratch_bytes', c_long)]


class WgradProblem(ctypes.Structure):
    _fields_ = [('A', _P), ('B', _P), ('rows', _P), ('rows_total', c_long), ('lda', c_int), ('ldb', c_int), ('M', c_int),
                ('N', c_int), ('k_begin', c_int), ('k_count', c_int)]


class WgradOutput(ctypes.Structure):
    _fields_ = [('kind', c_int), ('n_groups', c_int), ('first_problem', c_int), ('ldc', c_int), ('c_split', c_int),
                ('n_scalers', c_int), ('c_delta', c_long), ('scaler_stride', c_long), ('C', _P), ('aff', _P), ('row', _P),
                ('coef', POINTER(c_float))]


_SIGNATURES = {
    'i3d_bn_eval_aff_multi': (c_int, [POINTER(BnEvalAff), c_int, _P]),
    'i3d_set_collectives': (c_int, [POINTER(Collectives)]),
    'i3d_collectives_world': (c_int, []),
    'i3d_collectives_all_gather_f32': (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    'i3d_collectives_all_reduce_f64': (c_int, [c_void_p, c_long, c_void_p]),
    'i3d_rccl_available': (c_int, []),
    'i3d_rccl_unique_id': (c_int, [ctypes.c_char_p]),
    'i3d_rccl_init': (c_int, [ctypes.c_char_p, c_int, c_int, POINTER(c_void_p)]),
    'i3d_rccl_destroy': (c_int, [_P]),
    'i3d_set_collectives_rccl': (c_int, [_P, c_int, _P, c_long]),
    'i3d_block_copy': (c_int, [_P, c_int, c_int, _P]),
    'i3d_gru_gates_fwd': (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P]),
    'i3d_gru_gates_bwd': (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    'i3d_copy_cols': (c_int, [_P, c_int, c_int, _P, c_int, _P]),
    'i3d_edge_sqdist': (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, _P]),
    'i3d_tower_layer_saved_floats': (c_long, [POINTER(TowerLayerArgs)]),
    'i3d_tower_layer_scratch_floats': (c_long, [POINTER(TowerLayerArgs)]),
    'i3d_tower_layer_fwd': (c_int, [POINTER(TowerLayerArgs), _P]),
    'i3d_tower_layer_bwd': (c_int, [POINTER(TowerLayerArgs), _P]),
    'i3d_peer_mailbox_bytes': (c_long, []),
    'i3d_peer_handle_bytes': (c_int, []),
    'i3d_peer_alloc': (c_int, [POINTER(c_void_p), ctypes.c_char_p]),
    'i3d_peer_free': (c_int, [_P]),
    'i3d_peer_open': (c_int, [_P, ctypes.c_char_p, c_int, c_int, c_double, POINTER(c_void_p)]),
    'i3d_set_collectives_peer': (c_int, [_P, _P, c_long]),
    'i3d_peer_bind_stream': (c_int, [_P, _P, _P, c_long]),
    'i3d_peer_status': (c_int, [_P]),
    'i3d_peer_sequence': (ctypes.c_longlong, [_P]),
    'i3d_peer_close': (c_int, [_P]),
    'i3d_gemm_f32_fused_src': (c_int, [c_int, c_int, c_int, _P, c_int, c_long, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int,
                                       _P, _P, _P, c_long, _P]),
    'i3d_pna_pack_h_weights': (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, _P, _P, _P]),
    'i3d_bn_bwd_strided': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    'i3d_wgrad_multi_supported': (c_int, [POINTER(WgradProblem), c_int, POINTER(WgradOutput), c_int]),
    'i3d_wgrad_multi_workspace_bytes': (c_long, [c_int]),
    'i3d_wgrad_multi_min_workspace_bytes': (c_long, [POINTER(WgradProblem), c_int]),
    'i3d_wgrad_multi': (c_int, [POINTER(WgradProblem), c_int, POINTER(WgradOutput), c_int, _P, c_long, _P]),
    'i3d_pna_model_saved_floats': (c_long, [POINTER(PnaModel), POINTER(PnaBatch)]),
    'i3d_pna_model_scratch_floats': (c_long, [POINTER(PnaModel), POINTER(PnaBatch)]),
    'i3d_pna_model_fwd': (c_int, [POINTER(PnaModel), POINTER(PnaBatch), _P, _P, _P, _P, _P, _P, _P, POINTER(c_void_p)]),
    'i3d_pna_model_bwd': (c_int, [_P, POINTER(PnaModel), _P, _P, _P, _P, c_long, _P]),
    'i3d_pna_model_bwd_part': (c_int, [_P, POINTER(PnaModel), _P, _P, _P, _P, c_long, c_int, c_int, _P]),
    'i3d_pna_model_ctx_free': (c_int, [_P]),
    'i3d_pna_messages_normalized': (c_int, [_P, _P, c_long, c_int, _P, _P]),
    'i3d_pna_messages_normalized_ex': (c_int, [_P, c_int, _P, c_long, c_int, _P, _P]),
    'i3d_pna_model_debug_messages': (c_int, [_P, c_int, _P, _P]),
    'i3d_event_create': (c_int, [POINTER(c_void_p)]),
    'i3d_event_destroy': (c_int, [_P]),
    'i3d_event_record': (c_int, [_P, _P]),
    'i3d_event_elapsed_ms': (c_int, [_P, _P, POINTER(c_float)]),
    'i3d_set_matmul_precision': (c_int, [c_int]),
    'i3d_get_matmul_precision': (c_int, []),
    'i3d_set_fp32_products': (c_int, [c_int]),
    'i3d_get_fp32_products': (c_int, []),
    'i3d_net3d_edge_supported': (c_int, [c_int, c_int]),
    'i3d_net3d_edge_stats_floats': (c_long, [c_int, c_int]),
    'i3d_net3d_edge_bwd_floats': (c_long, [c_int, c_int, c_int]),
    'i3d_net3d_edge_fwd': (c_int, [POINTER(Net3dEdgeArgs), _P]),
    'i3d_net3d_edge_bwd': (c_int, [POINTER(Net3dEdgeArgs), _P]),
    'i3d_wgrad_stream_fork': (c_int, [_P, POINTER(_P)]),
    'i3d_wgrad_stream_peek': (c_int, [_P, POINTER(_P)]),
    'i3d_pna_layer_weights_fwd': (c_int, [POINTER(PnaLayerArgs), _P]),
    'i3d_pna_layer_fwd': (c_int, [POINTER(PnaLayerArgs), _P]),
    'i3d_pna_layer_bwd': (c_int, [POINTER(PnaLayerArgs), _P]),
    'i3d_fc_bn_fwd': (c_int, [POINTER(FcArgs), _P]),
    'i3d_fc_bn_bwd': (c_int, [POINTER(FcArgs), _P]),
    'i3d_fc_bn_bwd_chain': (c_int, [POINTER(FcArgs), _P]),
    'i3d_fc_bn_bwd_wgrad': (c_int, [POINTER(FcArgs), _P]),
    'i3d_edge_fc_bn_fwd': (c_int, [POINTER(EdgeFcArgs), _P]),
    'i3d_edge_fc_bn_bwd': (c_int, [POINTER(EdgeFcArgs), _P]),
    'i3d_grouped_fc_bn_fwd': (c_int, [POINTER(GroupedFcArgs), _P]),
    'i3d_grouped_fc_bn_bwd': (c_int, [POINTER(GroupedFcArgs), _P]),
    'i3d_pna_layer_stats_floats': (c_long, [c_int, c_int, c_int, c_int]),
    'i3d_bn_finalize_partials': (c_int, [_P, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'i3d_edge_stats_rows_per_tile': (c_int, [c_int]),
    'i3d_edge_combine_act_stats': (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    'i3d_gemm_f32_fused': (c_int, [c_int, c_int, c_int, _P, c_int, c_long, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, _P,
                                   _P, c_long, _P]),
    'i3d_gemm_f32_wgrad_bn': (c_int, [c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_long, _P]),
    'i3d_pna_aggregate_fwd_aff': (c_int, [_P, _P, _P, c_int, c_int, POINTER(c_int), c_int, POINTER(c_int), c_int, c_int, c_float,
                                          _P, _P]),
    'i3d_pna_aggregate_fwd_ex': (c_int, [_P, c_int, _P, _P, c_int, c_int, POINTER(c_int), c_int, POINTER(c_int), c_int, c_int, c_float, _P, _P]),
    'i3d_pna_aggregate_bwd_ex': (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, POINTER(c_int), c_int, POINTER(c_int), c_int, c_int, c_float, _P, _P]),
    'i3d_gemm_f32_fused_bf16out': (c_int, [c_int, c_int, c_int, _P, c_int, c_long, _P, c_int, _P, c_int, _P, _P, c_int, _P, _P]),
    'i3d_bn_bwd_x_bf16': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'i3d_pna_aggregate_bwd_aff': (c_int, [_P, _P, _P, _P, c_int, c_int, POINTER(c_int), c_int, POINTER(c_int), c_int, c_int,
                                          c_float, _P, _P]),
    'i3d_bn_bias_partial_floats': (c_long, [c_int]),
    'i3d_set_bn_bwd_one_launch': (c_int, [c_int]),
    'i3d_panel_packed_bytes': (c_long, [c_int, c_int]),
    'i3d_panel_pack': (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    'i3d_panel_pack_multi': (c_int, [_P, c_int, _P]),
    'i3d_panel_gemm': (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, c_int, _P]),
    'i3d_panel_stats_tiles': (c_int, [c_int]),
    'i3d_panel_gemm_fused': (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, _P, c_int, _P, _P]),
    'i3d_bn_bwd_one_launch_supported': (c_int, [c_int, c_int]),
    'i3d_bn_bwd_deferred_bias': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_long,
                                         _P, _P, _P]),
    'i3d_bn_bias_finalize': (c_int, [_P, c_int, c_int, _P, _P]),
    'i3d_bn_bwd_edge_sums': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, c_int, _P, _P]),
    'i3d_colsum_strided': (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    'i3d_wgrad_stream_join': (c_int, [_P]),
    'i3d_adam_chunk_elems': (c_int, []),
    'i3d_adam_chunk_bytes': (c_int, []),
    'i3d_adam_step': (c_int, [_P, c_int, _P, c_int, c_double, c_double, c_double, c_double, c_double, c_double, c_double, _P]),
    'i3d_ntxent_loss_scratch_floats': (c_long, [c_int, c_int]),
    'i3d_ntxent_loss_fwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P]),
    'i3d_ntxent_loss_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    'i3d_abi_version': (c_int, []),
    'i3d_last_error': (c_char_p, []),
    'i3d_embedding_sum_fwd': (c_int, [_P, _P, c_int, c_int, POINTER(c_void_p), c_int, _P, _P]),
    'i3d_embedding_sum_bwd': (c_int, [_P, _P, c_int, c_int, _P, c_int, POINTER(c_void_p), POINTER(c_int), _P]),
    'i3d_pna_aggregate_fwd_towers': (c_int, [_P, _P, c_int, c_int, c_int, POINTER(c_int), c_int, POINTER(c_int), c_int, c_int, c_float, _P, _P]),
    'i3d_pna_aggregate_bwd_towers': (c_int, [_P, _P, _P, c_int, c_int, c_int, POINTER(c_int), c_int, POINTER(c_int), c_int, c_int, c_float,
                                             _P, _P]),
    'i3d_pna_aggregate_fwd': (c_int, [_P, _P, c_int, c_int, POINTER(c_int), c_int, POINTER(c_int), c_int, c_int, c_float, _P,
                                      _P]),
    'i3d_pna_aggregate_bwd': (c_int, [_P, _P, _P, c_int, c_int, POINTER(c_int), c_int, POINTER(c_int), c_int, c_int, c_float,
                                      _P, _P]),
    'i3d_segment_readout_fwd': (c_int, [_P, _P, c_int, c_int, POINTER(c_int), c_int, _P, _P]),
    'i3d_segment_readout_bwd': (c_int, [_P, _P, _P, c_int, c_int, POINTER(c_int), c_int, _P, _P]),
    'i3d_gemm_f32': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    'i3d_gemm_f32_ex': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int,
                                _P, c_long, _P]),
    'i3d_gemm_f32_blocks': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, c_long, c_long, _P, c_int,
                                    c_int, c_long, c_int, _P, c_long, _P]),
    'i3d_gemm_f32_ws': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_long, _P]),
    'i3d_gemm_f32_grouped_batched': (c_int, [c_int, c_int, c_int, c_int, _P, c_int, c_long, c_long, _P, _P, _P, c_int, c_long, c_long, _P, c_int,
                                             c_long, c_int, c_int, _P]),
    'i3d_gemm_f32_batched': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, c_long, _P, c_int, c_long, _P, c_int, c_long, c_int,
                                     c_int, _P, c_long, _P]),
    'i3d_pna_combine_weights_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), _P, _P]),
    'i3d_pna_combine_weights_bwd': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), _P, _P]),
    'i3d_gemm_f32_grouped': (c_int, [c_int, c_int, c_int, c_int, _P, c_int, c_long, _P, _P, _P, c_int, c_long, _P, c_int,
                                     c_int, _P]),
    'i3d_gemm_f32_rowsubset': (c_int, [c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_long, _P, c_int, c_int, _P]),
    'i3d_gemm_f32_rowsubset_multi': (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), _P, c_int, _P, c_int, _P,
                                             c_long, _P, c_long, c_int, c_int, c_int, c_int, _P, c_long, _P]),
    'i3d_colreduce_workspace_bytes': (c_long, [c_int, c_int]),
    'i3d_act_stats_fwd': (c_int, [_P, c_int, c_int, c_int, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P]),
    'i3d_act_stats_fwd_counted': (c_int, [_P, c_int, c_int, c_int, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    'i3d_bn_finalize_stats': (c_int, [_P, c_int, c_float, c_float, _P, _P, _P, _P, _P]),
    'i3d_bn_apply_fwd': (c_int, [_P, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P]),
    'i3d_bn_eval_fwd': (c_int, [_P, c_int, c_int, _P, _P, c_float, _P, _P, c_int, _P, _P, _P]),
    'i3d_bn_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_long, _P, _P]),
    'i3d_bn_eval_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P]),
    'i3d_colsum': (c_int, [_P, _P, c_int, c_int, _P, _P, _P]),
    'i3d_act_fwd': (c_int, [_P, c_long, c_int, _P, _P]),
    'i3d_act_bwd': (c_int, [_P, _P, c_long, c_int, _P, _P]),
    'i3d_add_inplace': (c_int, [_P, _P, c_long, _P]),
    'i3d_add': (c_int, [_P, _P, c_long, _P, _P]),
    'i3d_mul': (c_int, [_P, _P, c_long, _P, _P]),
    'i3d_broadcast_row': (c_int, [_P, c_long, c_int, _P, _P]),
    'i3d_edge_combine_fwd': (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    'i3d_multihot': (c_int, [_P, _P, c_int, c_int, POINTER(c_int), c_int, _P, _P]),
    'i3d_edge_codes': (c_int, [_P, _P, c_int, c_int, POINTER(c_int), c_int, _P, _P, _P]),
    'i3d_segment_sum_bf16': (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    'i3d_segment_sum_pair': (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'i3d_segment_sum': (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    'i3d_segment_bcast': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P]),
    'i3d_gather_rows': (c_int, [_P, _P, c_int, c_int, _P, _P]),
    'i3d_fourier_encode': (c_int, [_P, c_int, c_int, _P, _P]),
    'i3d_soft_edge_fwd': (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P]),
    'i3d_soft_edge_bwd': (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    'i3d_row_norms': (c_int, [_P, c_int, c_int, _P, _P]),
    'i3d_ntxent_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P]),
    'i3d_ntxent_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P, _P]),
    'i3d_row_axpy': (c_int, [_P, _P, c_int, c_int, _P, _P]),
    'i3d_row_scale': (c_int, [_P, _P, c_int, c_int, _P, _P]),
    'i3d_contrastive_rowstats': (c_int, [_P, _P, _P, c_int, c_int, c_float, c_float, c_float, _P, _P]),
    'i3d_cov_rowstats': (c_int, [_P, _P, c_int, c_int, _P, _P]),
    'i3d_complete_graph_build': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def declared_symbols():
    """Function names declared in include/infomax3d_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(i3d_[a-z0-9_]+)\s*\(', text)))


def load():
    """Load the C-ABI library (once) and attach argtypes.  Raises HipLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # I3D_MATMUL_PRECISION=bf16: the trainer of the reference has no knob for it (ops.set_matmul_precision otherwise)
    prec = os.environ.get('I3D_MATMUL_PRECISION', 'fp32').lower()
    if prec not in ('fp32', 'bf16'):
        raise HipLibraryError(f'I3D_MATMUL_PRECISION={prec!r}: fp32 or bf16')
    lib.i3d_set_matmul_precision(int(prec == 'bf16'))
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        msg = load().i3d_last_error()
        raise HipLibraryError(f'{name} failed (rc={rc}): {msg.decode() if msg else ""}')


def int_array(values):
    return (c_int * len(values))(*values)


def ptr_array(ptrs):
    return (c_void_p * len(ptrs))(*ptrs)
