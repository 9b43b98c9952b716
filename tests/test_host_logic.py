"""Host-side logic that needs no GPU: the hot caches of FCLayer, the per-model tape state, gradient-pool bookkeeping."""
import copy
import importlib
import pickle

import pytest

import torch

amd = importlib.import_module('3dinfomax_amd')
layers = importlib.import_module('3dinfomax_amd.layers')
tape = importlib.import_module('3dinfomax_amd.tape')


def test_fclayer_hot_cache_follows_mode_device_casts_and_sync_group():
    fc = layers.FCLayer(6, 4, activation='relu', batch_norm=True, batch_norm_momentum=0.93)
    W, b, gamma, beta, spec = fc.hot()[:5]
    assert W is fc.linear.weight and b is fc.linear.bias and gamma is fc.batch_norm.weight and beta is fc.batch_norm.bias
    assert spec.bn.training and spec.bn.running_mean is fc.batch_norm.running_mean and spec.act == 'relu'
    assert fc.hot() is fc.hot()                              # cached
    assert fc.hot('silu')[4].post_act == 'silu' and fc.hot()[4].post_act is None
    fc.eval()
    assert not fc.hot()[4].bn.training
    fc.train()
    first = fc.hot()
    fc.double()                                              # _apply replaces the buffer objects
    again = fc.hot()
    assert again is not first and again[4].bn.running_mean is fc.batch_norm.running_mean
    fc.float()
    fc.sync_group = object()
    assert fc.hot()[4].bn.sync_group is fc.sync_group
    fc.sync_group = None
    assert fc.hot()[4].bn.sync_group is None
    # state_dict round trip keeps the cached tensors valid (load_state_dict copies in place)
    sd = {k: v.clone() + 1 for k, v in fc.state_dict().items()}
    cached = fc.hot()
    fc.load_state_dict(sd)
    assert fc.hot() is cached and torch.equal(cached[0], sd['linear.weight'])
    fc.linear.weight = torch.nn.Parameter(torch.zeros(4, 6))   # a re-assigned Parameter object is noticed
    assert fc.hot()[0] is fc.linear.weight and fc.hot() is not cached
    plain = layers.FCLayer(6, 4, activation='none', batch_norm=False)
    assert plain.hot()[2] is None and plain.hot()[4].bn is None


def test_model_state_is_per_module_and_does_not_travel_with_copies():
    m = torch.nn.Linear(3, 3)
    st = tape.model_state(m)
    assert tape.model_state(m) is st and st.module() is m
    m2 = copy.deepcopy(m)
    assert m2.__dict__.get('_i3d_state') is None
    assert tape.model_state(m2).module() is m2 and tape.model_state(m2) is not st
    m3 = pickle.loads(pickle.dumps(m))
    assert m3.__dict__.get('_i3d_state') is None


def test_grad_pool_hands_out_each_view_once_and_knows_the_bias_of_a_weight():
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3), torch.nn.Linear(3, 2))
    params = list(net.parameters())
    pool = tape.model_state(net).pool_for(params)
    assert tape.model_state(net).pool_for(params) is pool                 # cached for the same parameter list
    assert set(pool.view_of) == {id(p) for p in params}
    for p in params:
        assert pool.view_of[id(p)].shape == p.shape
    assert pool.bias_of[id(net[0].weight)] is net[0].bias and pool.bias_of[id(net[2].weight)] is net[2].bias
    # outside a model backward nothing is pooled: plain new tensors
    g = tape.grad_like(net[0].weight)
    assert g.shape == net[0].weight.shape and g is not pool.view_of[id(net[0].weight)]
    tape._tls.pool = pool
    try:
        pool.used.clear()
        a = tape.grad_like(net[0].weight)
        b = tape.grad_like(net[0].weight)                                 # a parameter used twice: second buffer is separate
        assert a is pool.view_of[id(net[0].weight)] and b is not a and b.shape == a.shape
        assert tape.grad_for_bias_of(net[0].weight, 3) is pool.view_of[id(net[0].bias)]
        assert tape.grad_for_bias_of(net[0].weight, 7).shape == (7,)       # size mismatch: not that bias
    finally:
        tape._tls.pool = None
    # a different parameter list (e.g. some frozen) gets a new pool
    net[2].weight.requires_grad_(False)
    pool2 = tape.model_state(net).pool_for([p for p in net.parameters() if p.requires_grad])
    assert pool2 is not pool and id(net[2].weight) not in pool2.view_of
    # a data-parallel sink brings its own views
    views = {id(p): torch.zeros_like(p) for p in params}
    tape.register_grad_sink(net, lambda ps, gs: gs, views)
    pool3 = tape.model_state(net).pool_for(params)
    assert all(pool3.view_of[id(p)] is views[id(p)] for p in params)


def test_adam_accepts_a_parameter_generator():
    """`Adam(model.parameters())` (a generator) used to be exhausted by the device check."""
    lin = torch.nn.Linear(4, 3)
    opt = amd.Adam(lin.parameters(), lr=1e-3)
    assert len(opt.param_groups[0]['params']) == 2
    lin(torch.randn(5, 4)).sum().backward()
    opt.step()
    opt2 = amd.Adam([{'params': [lin.weight]}, {'params': [lin.bias], 'weight_decay': 0}], lr=1e-3)
    assert len(opt2.param_groups) == 2


def test_param_list_cache_follows_replaced_heads_and_parameters():
    """the tape's cached leaf list must notice a swapped sub-module / re-assigned Parameter."""
    tape = importlib.import_module('3dinfomax_amd.tape')
    m = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
    a = tape._param_list(m)
    assert tape._param_list(m) is a and len(a) == 4
    m[1] = torch.nn.Linear(4, 3)                       # fine-tuning: new output head
    b = tape._param_list(m)
    assert b is not a and any(p is m[1].weight for p in b) and not any(p is a[2] for p in b)
    m[0].weight = torch.nn.Parameter(torch.zeros(4, 4))
    c = tape._param_list(m)
    assert c is not b and any(p is m[0].weight for p in c)
    m.add_module('extra', torch.nn.Linear(2, 2))
    assert len(tape._param_list(m)) == 6


def test_metrics_cache_requires_the_very_tensor_objects():
    """(address, version, shape) is not an identity - a new tensor at a recycled address must miss."""
    M = importlib.import_module('3dinfomax_amd.metrics')
    x = torch.zeros(2, 2)
    import weakref
    r = weakref.ref(x)
    assert M._same_object(r, x) and not M._same_object(r, torch.zeros(2, 2)) and not M._same_object(None, x)
    y = torch.zeros(2, 2)
    r2 = weakref.ref(y)
    del y
    assert not M._same_object(r2, x)


def test_fclayer_hot_cache_follows_an_in_place_change_of_the_dropout_probability():
    """the cached FCSpec bakes in dropout.p; `layer.dropout.p = q` (no attribute re-assignment) must invalidate it"""
    import importlib
    layers = importlib.import_module('3dinfomax_amd.layers')
    fc = layers.FCLayer(8, 8, activation='relu', dropout=0.3, batch_norm=True).train()
    assert fc.spec().dropout == pytest.approx(0.3)
    fc.dropout.p = 0.5
    assert fc.spec().dropout == pytest.approx(0.5)
    fc.eval()
    assert fc.spec().dropout == 0.0
    plain = layers.FCLayer(8, 8).train()
    assert plain.spec().dropout == 0.0 and plain.hot() is plain.hot()      # (no dropout module: the entry is reused)
