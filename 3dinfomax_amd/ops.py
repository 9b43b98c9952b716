"""Tensor-level wrappers over the C ABI (include/infomax3d_hip.h).

torch is plumbing here: it owns device memory (caching allocator) and the current HIP stream; every
computation below is one of the hand-written gfx950 kernels in csrc/.  No autograd in this file - the
autograd.Functions in layers.py / pna.py / net3d.py / losses.py chain these calls explicitly.
"""
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import ACT, AGG, SCALER, check, int_array, ptr_array

_tls = threading.local()
KERNEL_TIMERS = None      # set to a dict by bench.py to collect (start, end) HIP events of the roofline kernel


# The wrappers below are on the host's critical path (~210 calls per training step): pointers and the stream are
# passed to ctypes as plain ints (argtypes are c_void_p), the raw current stream comes from one C call.
_raw_stream = torch._C._cuda_getCurrentRawStream


def _stream():
    return _raw_stream(torch.cuda.current_device())


def _p(t):
    return t.data_ptr() if t is not None else None


def _chk(t, dtype=torch.float32):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise AssertionError(f'expected a contiguous {dtype} HIP tensor, got {t.device} {t.dtype} '
                             f'contiguous={t.is_contiguous()} (there is no CPU fallback)')
    return t


_ws_need = {}


def _workspace(feat, device):
    """Thread-local scratch for the two-stage column reductions (stream-ordered reuse), one per stream (the raw stream
    handle is the key: torch.cuda.current_stream() builds a Python object, ~10 us, and this runs ~60 times per step)."""
    need = _ws_need.get(feat)
    if need is None:
        need = _ws_need[feat] = _lib.load().i3d_colreduce_workspace_bytes(0, feat)
    key = (device.index, _raw_stream(device.index))
    ws = getattr(_tls, 'ws', None)
    if ws is None:
        ws = _tls.ws = {}
    buf = ws.get(key)
    if buf is None or buf.numel() < need:
        # zero-initialised: the arrival counters of the in-kernel finalisation live at its head (include/infomax3d_hip.h)
        buf = ws[key] = torch.zeros(max(need, 1 << 22), dtype=torch.uint8, device=device)
    return buf


def _gemm_workspace(device):
    """Thread-local scratch of the weight-gradient GEMMs (split-K / row-segment slices, include/infomax3d_hip.h
    i3d_gemm_f32_ws), one per stream."""
    key = (device.index, _raw_stream(device.index))
    ws = getattr(_tls, 'gws', None)
    if ws is None:
        ws = _tls.gws = {}
    buf = ws.get(key)
    if buf is None:
        buf = ws[key] = torch.empty(max(GEMM_WORKSPACE_BYTES, 16), dtype=torch.uint8, device=device)
    return buf


# I3D_GEMM_SCRATCH=0: fp32 atomics on top of a zero-fill instead of the two-stage reduction (A/B switch)
GEMM_WORKSPACE_BYTES = (160 << 20) if os.environ.get('I3D_GEMM_SCRATCH', '1') != '0' else 0


class RawEvent:
    """hipEvent_t created through the C ABI (for measurements around a kernel that a composite launches)"""

    def __init__(self):
        h = ctypes.c_void_p()
        check(_lib.load().i3d_event_create(ctypes.byref(h)), 'i3d_event_create')
        self.handle = h.value

    def record(self):
        """on the current stream"""
        check(_lib.load().i3d_event_record(self.handle, _stream()), 'i3d_event_record')

    def elapsed_time(self, stop):
        ms = ctypes.c_float()
        check(_lib.load().i3d_event_elapsed_ms(self.handle, stop.handle, ctypes.byref(ms)), 'i3d_event_elapsed_ms')
        return ms.value

    def __del__(self):
        try:
            _lib.load().i3d_event_destroy(self.handle)
        except Exception:
            pass


def agg_codes(names):
    return [AGG[n] for n in names]


def scaler_codes(names):
    return [SCALER[n] for n in names]


# ---- K1 --------------------------------------------------------------------------------------------------
def embedding_sum_fwd(idx, tables, row_perm=None):
    _chk(idx, torch.int64)
    rows, n_cols = idx.shape
    feat = tables[0].shape[1]
    out = torch.empty(rows, feat, dtype=torch.float32, device=idx.device)
    L = _lib.load()
    check(L.i3d_embedding_sum_fwd(_p(idx), _p(row_perm), rows, n_cols, ptr_array([_chk(t).data_ptr() for t in tables]), feat,
                                  _p(out), _stream()), 'i3d_embedding_sum_fwd')
    return out


def embedding_sum_bwd(idx, grad_out, dims, row_perm=None):
    _chk(idx, torch.int64)
    _chk(grad_out)
    rows, n_cols = idx.shape
    feat = grad_out.shape[1]
    total = sum(dims)
    L = _lib.load()
    if MULTIHOT_EMB_BWD and rows > 0 and total <= 1024 and n_cols <= 16 and (feat % 4 == 0 or MULTIHOT_UNALIGNED):
        # all tables at once: multi-hot^T dY, split-K through the scratch (deterministic; the LDS-atomics kernel below
        # needs ~115 us for the 9 atom tables of a 512-molecule batch)
        v_pad = (total + 31) // 32 * 32
        offsets, o = [], 0
        for d in dims:
            offsets.append(o)
            o += d
        hot = torch.empty(rows, v_pad, dtype=torch.float32, device=idx.device)
        check(L.i3d_multihot(_p(idx), _p(row_perm), rows, n_cols, int_array(offsets), v_pad, _p(hot), _stream()), 'i3d_multihot')
        flat = gemm(hot, grad_out, trans_a=True)
    else:
        flat = torch.zeros(total, feat, dtype=torch.float32, device=idx.device)     # one fill for all tables
        ptrs, o = [], 0
        for d in dims:
            ptrs.append(flat.data_ptr() + 4 * o * feat)
            o += d
        check(L.i3d_embedding_sum_bwd(_p(idx), _p(row_perm), rows, n_cols, _p(grad_out), feat, ptr_array(ptrs),
                                      int_array(list(dims)), _stream()), 'i3d_embedding_sum_bwd')
    grads, o = [], 0
    for d in dims:
        grads.append(flat[o:o + d])
        o += d
    return grads


# I3D_MULTIHOT_EMB_BWD=0: embedding-table gradients by LDS-privatised atomics instead of the multi-hot GEMM
MULTIHOT_EMB_BWD = True
# I3D_MULTIHOT_UNALIGNED=0: table widths that are not multiples of 4 (the tower variant's 90 / 70) through the LDS-atomics kernel
MULTIHOT_UNALIGNED = True


# ---- K4 / K6 ---------------------------------------------------------------------------------------------
def pna_aggregate_fwd(e, in_ptr, num_nodes, aggregators, scalers, avg_d_log=1.0, force_scalers=False, tower_feat=0):
    """tower_feat > 0: the output row is [tower][block][feature of the tower] (include/infomax3d_hip.h: i3d_pna_aggregate_fwd_towers)"""
    _chk(e)
    _chk(in_ptr, torch.int32)
    feat = e.shape[1]
    n_sc = len(scalers) if (len(scalers) > 1 or force_scalers) else 1
    out = torch.empty(num_nodes, n_sc * len(aggregators) * feat, dtype=torch.float32, device=e.device)
    L = _lib.load()
    if tower_feat:
        check(L.i3d_pna_aggregate_fwd_towers(_p(e), _p(in_ptr), num_nodes, feat, tower_feat, int_array(aggregators), len(aggregators),
                                             int_array(scalers), len(scalers), int(force_scalers), float(avg_d_log), _p(out), _stream()),
              'i3d_pna_aggregate_fwd_towers')
        return out
    timed = KERNEL_TIMERS is not None
    if timed:   # bench.py: HIP events on the launch stream around the roofline kernel
        t0, t1 = RawEvent(), RawEvent()
        t0.record()
    check(L.i3d_pna_aggregate_fwd(_p(e), _p(in_ptr), num_nodes, feat, int_array(aggregators), len(aggregators),
                                  int_array(scalers), len(scalers), int(force_scalers), float(avg_d_log), _p(out), _stream()),
          'i3d_pna_aggregate_fwd')
    if timed:
        t1.record()
        KERNEL_TIMERS.setdefault('pna_aggregate_fwd', []).append((t0, t1, num_nodes, e.shape[0], feat, out.shape[1]))
    return out


def pna_aggregate_bwd(grad_out, e, in_ptr, num_nodes, aggregators, scalers, avg_d_log=1.0, force_scalers=False, tower_feat=0):
    _chk(grad_out)
    _chk(e)
    grad_e = torch.empty_like(e)
    L = _lib.load()
    if tower_feat:
        check(L.i3d_pna_aggregate_bwd_towers(_p(grad_out), _p(e), _p(in_ptr), num_nodes, e.shape[1], tower_feat, int_array(aggregators),
                                             len(aggregators), int_array(scalers), len(scalers), int(force_scalers), float(avg_d_log),
                                             _p(grad_e), _stream()), 'i3d_pna_aggregate_bwd_towers')
        return grad_e
    check(L.i3d_pna_aggregate_bwd(_p(grad_out), _p(e), _p(in_ptr), num_nodes, e.shape[1], int_array(aggregators),
                                  len(aggregators), int_array(scalers), len(scalers), int(force_scalers), float(avg_d_log),
                                  _p(grad_e), _stream()), 'i3d_pna_aggregate_bwd')
    return grad_e


def segment_readout_fwd(x, graph_ptr, num_graphs, ops):
    _chk(x)
    _chk(graph_ptr, torch.int32)
    feat = x.shape[1]
    out = torch.empty(num_graphs, len(ops) * feat, dtype=torch.float32, device=x.device)
    L = _lib.load()
    check(L.i3d_segment_readout_fwd(_p(x), _p(graph_ptr), num_graphs, feat, int_array(ops), len(ops), _p(out), _stream()),
          'i3d_segment_readout_fwd')
    return out


def segment_readout_bwd(grad_out, x, graph_ptr, num_graphs, ops):
    _chk(grad_out)
    _chk(x)
    grad_x = torch.empty_like(x)
    L = _lib.load()
    check(L.i3d_segment_readout_bwd(_p(grad_out), _p(x), _p(graph_ptr), num_graphs, x.shape[1], int_array(ops), len(ops),
                                    _p(grad_x), _stream()), 'i3d_segment_readout_bwd')
    return grad_x


# ---- GEMM ------------------------------------------------------------------------------------------------
def _ld(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def set_matmul_precision(dtype):
    """'fp32' (default): exact fp32 products; 'bf16': bf16-rounded operands on the bf16 matrix pipe, fp32 accumulation
    (every GEMM of the library, process-level).  Returns the previous setting."""
    L = _lib.load()
    prev = 'bf16' if L.i3d_get_matmul_precision() else 'fp32'
    name = {torch.float32: 'fp32', torch.bfloat16: 'bf16'}.get(dtype, dtype)
    if name not in ('fp32', 'bf16'):
        raise ValueError(f'matmul precision {dtype!r}: fp32 or bf16')
    check(L.i3d_set_matmul_precision(int(name == 'bf16')), 'i3d_set_matmul_precision')
    return prev


def get_matmul_precision():
    return 'bf16' if _lib.load().i3d_get_matmul_precision() else 'fp32'


def set_fp32_products(mode):
    """'native': v_mfma_f32_32x32x2_f32; 'split': three-part bf16 split of both operands, six part products on the bf16 pipe
    (include/infomax3d_hip.h: i3d_set_fp32_products).  fp32 mode only.  Returns the previous setting."""
    L = _lib.load()
    prev = 'split' if L.i3d_get_fp32_products() else 'native'
    if mode not in ('native', 'split'):
        raise ValueError(f'fp32 products {mode!r}: native or split')
    check(L.i3d_set_fp32_products(int(mode == 'split')), 'i3d_set_fp32_products')
    return prev


def get_fp32_products():
    return 'split' if _lib.load().i3d_get_fp32_products() else 'native'


def gemm(A, B, trans_a=False, trans_b=False, out=None, bias=None, accumulate=False):
    """out[M,N] = (accumulate ? out : 0) + op(A) op(B) + bias.  A, B, out: 2-D fp32, unit inner stride
    (row slices / column slices of a contiguous tensor are fine: the row stride is the leading dimension)."""
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    K2, N = (B.shape[1], B.shape[0]) if trans_b else B.shape
    assert K == K2, (A.shape, B.shape, trans_a, trans_b)
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    L = _lib.load()
    if trans_a and K >= 1024:     # weight gradient: the split-K slices are combined through the scratch (deterministic)
        check(L.i3d_gemm_f32_ws(int(trans_a), int(trans_b), M, N, K, _p(A), _ld(A), _p(B), _ld(B), _p(out), _ld(out),
                                _p(bias), int(accumulate), _p(_gemm_workspace(A.device)), GEMM_WORKSPACE_BYTES, _stream()),
              'i3d_gemm_f32_ws')
        return out
    check(L.i3d_gemm_f32(int(trans_a), int(trans_b), M, N, K, _p(A), _ld(A), _p(B), _ld(B), _p(out), _ld(out),
                         _p(bias), int(accumulate), _stream()), 'i3d_gemm_f32')
    return out


# ---- BatchNorm / activations -----------------------------------------------------------------------------
def act_stats_fwd(pre, act, eps, momentum, running_mean=None, running_var=None, sums_out=None, out=None):
    """x = act(pre) (in place unless `out` is given), returns (x, mean, invstd)."""
    _chk(pre)
    rows, feat = pre.shape
    x = pre if out is None else out
    mean = torch.empty(feat, dtype=torch.float32, device=pre.device)
    invstd = torch.empty(feat, dtype=torch.float32, device=pre.device)
    L = _lib.load()
    check(L.i3d_act_stats_fwd(_p(pre), rows, feat, ACT[act], _p(x), float(eps), float(momentum), _p(mean), _p(invstd),
                              _p(running_mean), _p(running_var), _p(sums_out), _p(_workspace(feat, pre.device)),
                              _stream()), 'i3d_act_stats_fwd')
    return x, mean, invstd


def bn_finalize_stats(sums, feat, eps, momentum, running_mean=None, running_var=None):
    mean = torch.empty(feat, dtype=torch.float32, device=sums.device)
    invstd = torch.empty(feat, dtype=torch.float32, device=sums.device)
    L = _lib.load()
    check(L.i3d_bn_finalize_stats(_p(sums), feat, float(eps), float(momentum), _p(mean), _p(invstd), _p(running_mean),
                                  _p(running_var), _stream()), 'i3d_bn_finalize_stats')
    return mean, invstd


def bn_apply_fwd(x, mean, invstd, gamma, beta, post_act=None, residual=None, out=None):
    _chk(x)
    rows, feat = x.shape
    y = torch.empty_like(x) if out is None else out
    L = _lib.load()
    check(L.i3d_bn_apply_fwd(_p(x), rows, feat, _p(mean), _p(invstd), _p(gamma), _p(beta), ACT[post_act], _p(residual),
                             _p(y), _stream()), 'i3d_bn_apply_fwd')
    return y


def bn_eval_fwd(x, running_mean, running_var, eps, gamma, beta, post_act=None, residual=None, out=None):
    _chk(x)
    rows, feat = x.shape
    y = torch.empty_like(x) if out is None else out
    L = _lib.load()
    check(L.i3d_bn_eval_fwd(_p(x), rows, feat, _p(running_mean), _p(running_var), float(eps), _p(gamma), _p(beta),
                            ACT[post_act], _p(residual), _p(y), _stream()), 'i3d_bn_eval_fwd')
    return y


def bn_bwd(grad_y, x, pre, act, post_act, mean, invstd, gamma, beta, sums_out=None, sums_in=None, total_rows=0,
           grad_gamma=None, grad_beta=None, out=None, grad_bias=None):
    """Returns (grad_pre, grad_gamma, grad_beta).  grad_pre may alias grad_y (out=grad_y); grad_bias (optional, [feat])
    receives the column sums of grad_pre from the same pass."""
    _chk(grad_y)
    _chk(x)
    rows, feat = x.shape
    if grad_gamma is None:
        grad_gamma = torch.empty(feat, dtype=torch.float32, device=x.device)
        grad_beta = torch.empty(feat, dtype=torch.float32, device=x.device)
    grad_pre = torch.empty_like(x) if out is None else out
    L = _lib.load()
    check(L.i3d_bn_bwd(_p(grad_y), _p(x), _p(pre), rows, feat, ACT[act], ACT[post_act], _p(mean), _p(invstd), _p(gamma),
                       _p(beta), _p(grad_gamma), _p(grad_beta), _p(grad_pre), _p(grad_bias), _p(sums_out), _p(sums_in),
                       int(total_rows), _p(_workspace(feat, x.device)), _stream()), 'i3d_bn_bwd')
    return grad_pre, grad_gamma, grad_beta


def bn_eval_bwd(grad_y, x, pre, act, post_act, running_mean, running_var, eps, gamma, beta, out=None):
    _chk(grad_y)
    _chk(x)
    rows, feat = x.shape
    grad_gamma = torch.empty(feat, dtype=torch.float32, device=x.device)
    grad_beta = torch.empty(feat, dtype=torch.float32, device=x.device)
    grad_pre = torch.empty_like(x) if out is None else out
    L = _lib.load()
    check(L.i3d_bn_eval_bwd(_p(grad_y), _p(x), _p(pre), rows, feat, ACT[act], ACT[post_act], _p(running_mean),
                            _p(running_var), float(eps), _p(gamma), _p(beta), _p(grad_gamma), _p(grad_beta), _p(grad_pre),
                            _p(_workspace(feat, x.device)), _stream()), 'i3d_bn_eval_bwd')
    return grad_pre, grad_gamma, grad_beta


def set_bn_bwd_one_launch(on):
    """The BatchNorm backward as one launch (reduction, finalisation and data gradient; csrc/bn.hip: bn_bwd_fused_kernel) - process
    wide, on by default; off = the two-pass kernels (what larger tensors take anyway).  Switch it off when several processes share
    one GPU (include/infomax3d_hip.h: i3d_set_bn_bwd_one_launch).  Returns the previous setting."""
    return bool(_lib.load().i3d_set_bn_bwd_one_launch(int(bool(on))))


def colsum(x, w=None, out=None):
    _chk(x)
    rows, feat = x.shape
    if out is None:
        out = torch.empty(feat, dtype=torch.float32, device=x.device)
    L = _lib.load()
    check(L.i3d_colsum(_p(x), _p(w), rows, feat, _p(out), _p(_workspace(feat, x.device)), _stream()), 'i3d_colsum')
    return out


def act_fwd(x, act, out=None):
    _chk(x)
    y = torch.empty_like(x) if out is None else out
    check(_lib.load().i3d_act_fwd(_p(x), x.numel(), ACT[act], _p(y), _stream()), 'i3d_act_fwd')
    return y


def act_bwd(grad_y, x, act, out=None):
    _chk(x)
    g = torch.empty_like(x) if out is None else out
    check(_lib.load().i3d_act_bwd(_p(grad_y), _p(x), x.numel(), ACT[act], _p(g), _stream()), 'i3d_act_bwd')
    return g


def add_inplace(dst, src):
    _chk(dst)
    _chk(src)
    assert dst.numel() == src.numel()
    check(_lib.load().i3d_add_inplace(_p(dst), _p(src), dst.numel(), _stream()), 'i3d_add_inplace')
    return dst


def mul(a, b, out=None):
    """out = a * b elementwise (same shape; out may be a)"""
    _chk(a)
    _chk(b)
    assert a.numel() == b.numel()
    if out is None:
        out = torch.empty_like(a)
    check(_lib.load().i3d_mul(_p(a), _p(b), a.numel(), _p(out), _stream()), 'i3d_mul')
    return out


def dropout_mask(like, p):
    """the scaled keep-mask nn.Dropout(p) would apply to a tensor shaped like `like` at this point of torch's generator
    stream: torch's own dropout kernel on a tensor of ones (same shape -> same Philox offsets -> the SAME mask the reference's
    module draws on this device with this seed), values 0 or 1 / (1 - p)"""
    return torch.nn.functional.dropout(torch.ones_like(like), float(p), True)


# ---- edge kernels ----------------------------------------------------------------------------------------
def edge_combine_fwd(P, Q, bias, src_s, dst_s, q_code=None):
    _chk(P)
    E, feat = src_s.shape[0], P.shape[1] // 2
    pre = torch.empty(E, feat, dtype=torch.float32, device=P.device)
    check(_lib.load().i3d_edge_combine_fwd(_p(P), P.shape[1], _p(Q), _p(q_code), _p(bias), _p(src_s), _p(dst_s), E, feat,
                                           _p(pre), _stream()), 'i3d_edge_combine_fwd')
    return pre


def edge_codes(idx, row_perm, dims, v_pad):
    """codes[j] = joint category of row (row_perm[j] if given else j) of idx [rows, C]; onehot [rows, v_pad]."""
    _chk(idx, torch.int64)
    rows, n_cols = idx.shape
    strides, s = [], 1
    for d in dims:
        strides.append(s)
        s *= d
    assert s <= v_pad and v_pad % 4 == 0 and len(dims) == n_cols
    codes = torch.empty(rows, dtype=torch.int32, device=idx.device)
    onehot = torch.empty(rows, v_pad, dtype=torch.float32, device=idx.device)
    check(_lib.load().i3d_edge_codes(_p(idx), _p(row_perm), rows, n_cols, int_array(strides), v_pad, _p(codes), _p(onehot),
                                     _stream()), 'i3d_edge_codes')
    return codes, onehot


def segment_sum(x, ptr, idx, num_segments, mean=False, out=None):
    """out[v] = sum_{j in [ptr[v],ptr[v+1])} x[idx[j] if idx is not None else j]  (x, out may be column slices)."""
    feat = x.shape[1]
    if out is None:
        out = torch.empty(num_segments, feat, dtype=torch.float32, device=x.device)
    check(_lib.load().i3d_segment_sum(_p(x), _ld(x), _p(ptr), _p(idx), num_segments, feat, int(mean), _p(out), _ld(out),
                                      _stream()), 'i3d_segment_sum')
    return out


def segment_bcast(g, ptr, seg_of_row, rows, mean=False):
    _chk(g)
    feat = g.shape[1]
    out = torch.empty(rows, feat, dtype=torch.float32, device=g.device)
    check(_lib.load().i3d_segment_bcast(_p(g), _p(ptr), _p(seg_of_row), rows, feat, int(mean), _p(out), _stream()),
          'i3d_segment_bcast')
    return out


def gather_rows(x, idx):
    _chk(x)
    _chk(idx, torch.int32)
    out = torch.empty(idx.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
    check(_lib.load().i3d_gather_rows(_p(x), _p(idx), idx.shape[0], x.shape[1], _p(out), _stream()), 'i3d_gather_rows')
    return out


def fourier_encode(d, n_enc):
    _chk(d)
    E = d.shape[0]
    out = torch.empty(E, 2 * n_enc + 1, dtype=torch.float32, device=d.device)
    check(_lib.load().i3d_fourier_encode(_p(d), E, n_enc, _p(out), _stream()), 'i3d_fourier_encode')
    return out


def soft_edge_fwd(m, ws, bs):
    _chk(m)
    E, feat = m.shape
    msg = torch.empty_like(m)
    w = torch.empty(E, dtype=torch.float32, device=m.device)
    check(_lib.load().i3d_soft_edge_fwd(_p(m), _p(ws), _p(bs), E, feat, _p(msg), _p(w), _stream()), 'i3d_soft_edge_fwd')
    return msg, w


def soft_edge_bwd(grad_msg, m, w, ws):
    _chk(grad_msg)
    E, feat = m.shape
    gm = torch.empty_like(m)
    gg = torch.empty(E, dtype=torch.float32, device=m.device)
    check(_lib.load().i3d_soft_edge_bwd(_p(grad_msg), _p(m), _p(w), _p(ws), E, feat, _p(gm), _p(gg), _stream()),
          'i3d_soft_edge_bwd')
    return gm, gg


# ---- NT-Xent ---------------------------------------------------------------------------------------------
def row_norms(z):
    _chk(z)
    n = torch.empty(z.shape[0], dtype=torch.float32, device=z.device)
    check(_lib.load().i3d_row_norms(_p(z), z.shape[0], z.shape[1], _p(n), _stream()), 'i3d_row_norms')
    return n


def ntxent_fwd(sim, n1, n2, b1, b2, conf, pos_offset, tau, eps, loss_scale=1.0):
    row_sum = torch.empty(b1, dtype=torch.float32, device=sim.device)
    row_pos = torch.empty(b1, dtype=torch.float32, device=sim.device)
    loss_sum = torch.empty(1, dtype=torch.float32, device=sim.device)
    check(_lib.load().i3d_ntxent_fwd(_p(sim), _p(n1), _p(n2), b1, b2, conf, pos_offset, float(tau), float(eps),
                                     float(loss_scale), _p(row_sum), _p(row_pos), _p(loss_sum), _stream()), 'i3d_ntxent_fwd')
    return row_sum, row_pos, loss_sum


def ntxent_bwd(sim, n1, n2, row_sum, row_pos, b1, b2, conf, pos_offset, tau, eps, grad_scale, grad_scale_dev=None):
    dsim = torch.empty_like(sim)
    ca = torch.empty(b1, dtype=torch.float32, device=sim.device)
    cb = torch.empty(b2 * conf, dtype=torch.float32, device=sim.device)
    check(_lib.load().i3d_ntxent_bwd(_p(sim), _p(n1), _p(n2), _p(row_sum), _p(row_pos), b1, b2, conf, pos_offset,
                                     float(tau), float(eps), float(grad_scale), _p(grad_scale_dev), _p(dsim), _p(ca), _p(cb),
                                     _stream()),
          'i3d_ntxent_bwd')
    return dsim, ca, cb


def add(a, b):
    """a + b (new tensor) without a torch op: clone-free"""
    out = torch.empty_like(a)
    check(_lib.load().i3d_add(_p(a), _p(b), a.numel(), _p(out), _stream()), 'i3d_add')
    return out


def broadcast_row(row, rows):
    """[rows, F] with every row = `row` [F]"""
    out = torch.empty(rows, row.shape[0], dtype=torch.float32, device=row.device)
    check(_lib.load().i3d_broadcast_row(_p(row), rows, row.shape[0], _p(out), _stream()), 'i3d_broadcast_row')
    return out


def row_axpy(z, coef, out):
    _chk(z)
    _chk(out)
    check(_lib.load().i3d_row_axpy(_p(z), _p(coef), z.shape[0], z.shape[1], _p(out), _stream()), 'i3d_row_axpy')
    return out


def row_scale(z, coef):
    _chk(z)
    out = torch.empty_like(z)
    check(_lib.load().i3d_row_scale(_p(z), _p(coef), z.shape[0], z.shape[1], _p(out), _stream()), 'i3d_row_scale')
    return out


# ---- degree-grouped posttrans ------------------------------------------------------------------------------
def _float_array(values):
    from ctypes import c_float
    return (c_float * len(values))(*values)


def combine_weights_fwd(W, f_in, agg_width, coef, n_groups, n_scalers):
    """WD[g] = sum_s coef[g][s] * W[:, f_in + s*agg_width : f_in + (s+1)*agg_width]  -> [n_groups, f_out, agg_width]."""
    _chk(W)
    f_out = W.shape[0]
    WD = torch.empty(n_groups, f_out, agg_width, dtype=torch.float32, device=W.device)
    check(_lib.load().i3d_pna_combine_weights_fwd(_p(W), W.shape[1], f_in, f_out, agg_width, n_groups, n_scalers,
                                                  _float_array(coef), _p(WD), _stream()), 'i3d_pna_combine_weights_fwd')
    return WD


def combine_weights_bwd(dWD, dW, f_in, agg_width, coef, n_groups, n_scalers):
    """dW[:, f_in + s*agg_width ...] = sum_g coef[g][s] * dWD[g]   (writes the aggregate blocks of dW in place)."""
    _chk(dWD)
    _chk(dW)
    check(_lib.load().i3d_pna_combine_weights_bwd(_p(dWD), dW.shape[1], f_in, dW.shape[0], agg_width, n_groups,
                                                  n_scalers, _float_array(coef), _p(dW), _stream()),
          'i3d_pna_combine_weights_bwd')
    return dW


def gemm_grouped(A, m_rows, tile_group, Bg, out, trans_b, accumulate):
    """out[r] (+)= A[r] @ op(Bg[group of r])  for the rows listed in m_rows (64-padded per group, -1 = padding)."""
    _chk(A)
    _chk(Bg)
    K = A.shape[1]
    N = Bg.shape[1] if trans_b else Bg.shape[2]
    assert (Bg.shape[2] if trans_b else Bg.shape[1]) == K and out.shape[1] == N
    check(_lib.load().i3d_gemm_f32_grouped(int(trans_b), m_rows.shape[0], N, K, _p(A), A.shape[1], A.shape[0], _p(m_rows),
                                           _p(tile_group), _p(Bg), Bg.shape[2], Bg.shape[1] * Bg.shape[2], _p(out),
                                           _ld(out), int(accumulate), _stream()), 'i3d_gemm_f32_grouped')
    return out


def gemm_rowsubset_multi(A, B, k_rows, group_start, group_count, out, accumulate=False, tile_cfg=-1, seg_rows=0,
                         use_workspace=True):
    """out[g] = sum over j in [group_start[g], +group_count[g]) of A[k_rows[j],:]^T B[k_rows[j],:]; out: [G, M, N]."""
    _chk(A)
    _chk(B)
    _chk(out)
    G = len(group_start)
    assert out.shape == (G, A.shape[1], B.shape[1])
    check(_lib.load().i3d_gemm_f32_rowsubset_multi(A.shape[1], B.shape[1], G, int_array(list(group_start)),
                                                   int_array(list(group_count)), _p(A), A.shape[1], _p(B), B.shape[1],
                                                   _p(k_rows), A.shape[0], _p(out), out.shape[1] * out.shape[2],
                                                   out.shape[2], int(accumulate), tile_cfg, seg_rows,
                                                   _p(_gemm_workspace(A.device)) if use_workspace else None,
                                                   GEMM_WORKSPACE_BYTES if use_workspace else 0, _stream()),
          'i3d_gemm_f32_rowsubset_multi')
    return out


def gemm_rowsubset(A, B, k_rows, out):
    """out[M,N] = sum over rows r in k_rows of A[r,:M]^T B[r,:N]."""
    _chk(A)
    _chk(B)
    check(_lib.load().i3d_gemm_f32_rowsubset(A.shape[1], B.shape[1], k_rows.shape[0], _p(A), A.shape[1], _p(B), B.shape[1],
                                             _p(k_rows), A.shape[0], _p(out), _ld(out), 0, _stream()),
          'i3d_gemm_f32_rowsubset')
    return out


# ---- BatchNorm out of the memory path (csrc/fused_bn.hip) --------------------------------------------------
def bn_finalize_partials(partial, n_tiles, feat, eps, momentum, gamma=None, beta=None, running_mean=None, running_var=None,
                         num_batches_tracked=None, want_aff=True):
    """per-tile partials [n_tiles, 3, feat] -> (mean, invstd, aff [3, feat] = mean | gamma invstd | beta or None)"""
    _chk(partial)
    dev = partial.device
    mean = torch.empty(feat, dtype=torch.float32, device=dev)
    invstd = torch.empty(feat, dtype=torch.float32, device=dev)
    aff = torch.empty(3, feat, dtype=torch.float32, device=dev) if want_aff else None
    check(_lib.load().i3d_bn_finalize_partials(_p(partial), n_tiles, feat, float(eps), float(momentum), _p(gamma), _p(beta),
                                               _p(mean), _p(invstd), _p(running_mean), _p(running_var),
                                               _p(num_batches_tracked), _p(aff), _stream()), 'i3d_bn_finalize_partials')
    return mean, invstd, aff


def edge_combine_act_stats(P, Q, bias, src_s, dst_s, act=None, q_code=None):
    """x = act(P[src,:F] + P[dst,F:] + Q[code or row] + bias) [E, F] and its per-tile statistics -> (x, partial, n_tiles)"""
    _chk(P)
    E, feat = src_s.shape[0], P.shape[1] // 2
    L = _lib.load()
    rpt = L.i3d_edge_stats_rows_per_tile(feat)
    n_tiles = (E + rpt - 1) // rpt
    x = torch.empty(E, feat, dtype=torch.float32, device=P.device)
    partial = torch.empty(n_tiles, 3, feat, dtype=torch.float32, device=P.device)
    check(L.i3d_edge_combine_act_stats(_p(P), P.shape[1], _p(Q), _p(q_code), _p(bias), _p(src_s), _p(dst_s), E, feat, ACT[act],
                                       _p(x), _p(partial), _stream()), 'i3d_edge_combine_act_stats')
    return x, partial, n_tiles


def gemm_fused(A, W, bias=None, a_aff=None, act=None, want_stats=True, out=None, accumulate=False, m_rows=None,
               tile_group=None):
    """out = act(A' W^T + bias (+ out)), A' = (A - mean) * scale + shift per column when a_aff [3, K] is given;
    -> (out, partial [tiles, 3, N] or None, tiles).  W: [N, K], or [G, N, K] with m_rows / tile_group (grouped).
    act: None, 'relu' or 'leakyrelu' only (since round 4 the fused epilogue carries the ReLU class; SiLU / Sigmoid / Tanh ... run as a
    pass of their own behind a plain product - the C entry point refuses their codes)."""
    _chk(A)
    K = A.shape[1]
    if m_rows is not None:
        M, N, stride, rows_total = m_rows.shape[0], W.shape[1], W.shape[1] * W.shape[2], A.shape[0]
    else:
        M, N, stride, rows_total = A.shape[0], W.shape[0], 0, A.shape[0]
    if out is None:
        assert not accumulate
        out = torch.empty(rows_total, N, dtype=torch.float32, device=A.device)
    tiles = (M + 63) // 64
    partial = torch.empty(tiles, 3, N, dtype=torch.float32, device=A.device) if want_stats else None
    check(_lib.load().i3d_gemm_f32_fused(M, N, K, _p(A), K, rows_total, _p(W), K, _p(out), N, _p(bias), int(accumulate),
                                         _p(a_aff), ACT[act], _p(partial), _p(m_rows), _p(tile_group), stride, _stream()),
          'i3d_gemm_f32_fused')
    return out, partial, tiles


def panel_pack(W, trans):
    """The weight of a Linear packed for the row-panel products (csrc/panel.hip): its three bf16 images in the LDS layout of the product
    kernel.  trans=True: B[n][k] = W[n, k] (forward, W stored [out, in]); False: B[n][k] = W[k, n] (data gradient).  -> uint8 tensor"""
    _chk(W)
    N, K = (W.shape[0], W.shape[1]) if trans else (W.shape[1], W.shape[0])
    L = _lib.load()
    packed = torch.empty(L.i3d_panel_packed_bytes(N, K), dtype=torch.uint8, device=W.device)
    check(L.i3d_panel_pack(_p(W), W.stride(0), N, K, int(bool(trans)), _p(packed), _stream()), 'i3d_panel_pack')
    return packed, N, K


def panel_gemm(A, packed, N, bias=None, out=None, accumulate=False):
    """out (+)= A B^T (+ bias) with B packed by panel_pack: every 64-row (or 32-row) slab of A read and split once per 208 columns"""
    _chk(A)
    M, K = A.shape
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    check(_lib.load().i3d_panel_gemm(M, N, K, _p(A), A.stride(0), _p(packed), _p(out), out.stride(0), _p(bias), int(accumulate), _stream()),
          'i3d_panel_gemm')
    return out


def panel_gemm_fused(A, packed, N, bias=None, a_aff=None, act=None, want_stats=True):
    """gemm_fused in row-panel form -> (out, partial [tiles, 3, N] of 32-row tiles or None, tiles)"""
    _chk(A)
    M, K = A.shape
    L = _lib.load()
    out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    tiles = L.i3d_panel_stats_tiles(M)
    partial = torch.empty(tiles, 3, N, dtype=torch.float32, device=A.device) if want_stats else None
    check(L.i3d_panel_gemm_fused(M, N, K, _p(A), A.stride(0), _p(packed), _p(out), N, _p(bias), _p(a_aff), ACT[act], _p(partial), _stream()),
          'i3d_panel_gemm_fused')
    return out, partial, tiles


def gemm_wgrad_bn(dY, x, grad_bias, aff):
    """dW = dY^T ((x - mean) * scale + shift) from the raw x (aff [3, f_in]); grad_bias = column sums of dY"""
    _chk(dY), _chk(x)
    dW = torch.empty(dY.shape[1], x.shape[1], dtype=torch.float32, device=x.device)
    check(_lib.load().i3d_gemm_f32_wgrad_bn(dY.shape[1], x.shape[1], x.shape[0], _p(dY), dY.shape[1], _p(x), x.shape[1], _p(dW),
                                            x.shape[1], _p(grad_bias), _p(aff), _p(_gemm_workspace(x.device)),
                                            GEMM_WORKSPACE_BYTES, _stream()), 'i3d_gemm_f32_wgrad_bn')
    return dW


def pna_aggregate_fwd_aff(e, aff, in_ptr, num_nodes, aggregators, scalers, avg_d_log, force_scalers=False):
    if e.dtype == torch.bfloat16:      # messages stored as bf16 (the bf16 mode's storage form): i3d_pna_aggregate_fwd_ex
        n_s = len(scalers) if (len(scalers) > 1 or force_scalers) else 1
        out = torch.empty(num_nodes, n_s * len(aggregators) * e.shape[1], dtype=torch.float32, device=e.device)
        check(_lib.load().i3d_pna_aggregate_fwd_ex(e.data_ptr(), 1, _p(aff), _p(in_ptr), num_nodes, e.shape[1], int_array(aggregators),
                                                   len(aggregators), int_array(scalers), len(scalers), int(force_scalers),
                                                   float(avg_d_log), _p(out), _stream()), 'i3d_pna_aggregate_fwd_ex')
        return out
    _chk(e)
    n_s = len(scalers) if (len(scalers) > 1 or force_scalers) else 1
    out = torch.empty(num_nodes, n_s * len(aggregators) * e.shape[1], dtype=torch.float32, device=e.device)
    check(_lib.load().i3d_pna_aggregate_fwd_aff(_p(e), _p(aff), _p(in_ptr), num_nodes, e.shape[1], int_array(aggregators),
                                                len(aggregators), int_array(scalers), len(scalers), int(force_scalers),
                                                float(avg_d_log), _p(out), _stream()), 'i3d_pna_aggregate_fwd_aff')
    return out


def pna_aggregate_bwd_aff(grad_out, e, aff, in_ptr, num_nodes, aggregators, scalers, avg_d_log, force_scalers=False):
    if e.dtype == torch.bfloat16:
        _chk(grad_out)
        ge = torch.empty(e.shape, dtype=torch.float32, device=e.device)      # (every edge row has a destination: all rows are written)
        check(_lib.load().i3d_pna_aggregate_bwd_ex(_p(grad_out), e.data_ptr(), 1, _p(aff), _p(in_ptr), num_nodes, e.shape[1],
                                                   int_array(aggregators), len(aggregators), int_array(scalers), len(scalers),
                                                   int(force_scalers), float(avg_d_log), _p(ge), _stream()), 'i3d_pna_aggregate_bwd_ex')
        return ge
    _chk(grad_out), _chk(e)
    ge = torch.empty_like(e)
    check(_lib.load().i3d_pna_aggregate_bwd_aff(_p(grad_out), _p(e), _p(aff), _p(in_ptr), num_nodes, e.shape[1],
                                                int_array(aggregators), len(aggregators), int_array(scalers), len(scalers),
                                                int(force_scalers), float(avg_d_log), _p(ge), _stream()),
          'i3d_pna_aggregate_bwd_aff')
    return ge


WGRAD_PLAIN, WGRAD_BN, WGRAD_COMBINE = 0, 1, 2


def wgrad_multi(problems, outputs):
    """All weight gradients of a layer from one launch + one fixed-order reduction (csrc/wgrad.hip).

    problems: dicts {A [rows, M], B [rows, N], rows (int32 index or None), k_begin, k_count};
    outputs: dicts {kind, first_problem, n_groups, C (tensor, written in place), ldc, c_split, c_delta, aff, row, coef
    (list of n_groups * n_scalers floats), n_scalers, scaler_stride}."""
    L = _lib.load()
    pa = (_lib.WgradProblem * len(problems))()
    keep = []
    for d, p in zip(problems, pa):
        A, B = _chk(d['A']), _chk(d['B'])
        p.A, p.B = A.data_ptr(), B.data_ptr()
        p.rows = d['rows'].data_ptr() if d.get('rows') is not None else None
        p.rows_total = A.shape[0]
        p.lda, p.ldb = d.get('lda', A.shape[1]), d.get('ldb', B.shape[1])
        p.M, p.N = d.get('M', A.shape[1]), d.get('N', B.shape[1])
        p.k_begin = d.get('k_begin', 0)
        p.k_count = d.get('k_count', A.shape[0] if d.get('rows') is None else d['rows'].shape[0])
    oa = (_lib.WgradOutput * len(outputs))()
    for d, o in zip(outputs, oa):
        o.kind, o.n_groups, o.first_problem = d.get('kind', WGRAD_PLAIN), d.get('n_groups', 1), d['first_problem']
        C = _chk(d['C'])
        o.C, o.ldc = C.data_ptr() + 4 * d.get('c_offset', 0), d.get('ldc', C.shape[-1])
        o.c_split, o.c_delta = d.get('c_split', 0), d.get('c_delta', 0)
        o.n_scalers, o.scaler_stride = d.get('n_scalers', 1), d.get('scaler_stride', 0)
        o.aff = d['aff'].data_ptr() if d.get('aff') is not None else None
        o.row = d['row'].data_ptr() if d.get('row') is not None else None
        if d.get('coef') is not None:
            arr = _float_array(d['coef'])
            keep.append(arr)
            o.coef = arr
    dev = problems[0]['A'].device
    check(L.i3d_wgrad_multi(pa, len(problems), oa, len(outputs), _p(_gemm_workspace(dev)), GEMM_WORKSPACE_BYTES, _stream()),
          'i3d_wgrad_multi')
