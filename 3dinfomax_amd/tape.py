"""A define-by-run tape over the block Functions of this package: a whole model forward/backward as ONE autograd node.

Why: the pre-training step is ~330 small kernel launches and the host is as much on the critical path as the GPU
(DESIGN.md section 7).  torch's autograd costs ~25-30 us of host time per node in each direction and ~35 us per
gradient accumulation (`at::add`) on this stack; PNA + Net3D are ~35 nodes.  Under a tape the same `forward` /
`backward` static methods of the block Functions (layers.py, net3d.py, pna.py - unchanged) run back to back from plain
Python: the tape records (Function, context, inputs, output), walks the records in reverse for the backward pass and
sums multiple gradient contributions with the in-place add kernel.  torch.autograd only sees the model's parameters
going in and the model output coming out (`ModelFn`).

Gradient-buffer ownership: a backward may return the incoming gradient itself (residual branches, `_AddFn`) or the
same tensor for two inputs; such aliases are never written in place - the accumulator is either a buffer a kernel just
produced ("fresh") or a new one.
"""
import contextlib
import os
import threading

import torch

from . import ops, streams

_tls = threading.local()


class SubCtx:
    """Stand-in for the autograd context of a block Function whose forward/backward runs as a step of a tape (or of a
    fused layer Function)."""

    def __init__(self, needs_input_grad=(True,) * 16):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def active():
    return getattr(_tls, 'tape', None)


def apply(fn, *args):
    """fn.apply(*args) - through the active tape if there is one."""
    tape = getattr(_tls, 'tape', None)
    if tape is None:
        return fn.apply(*args)
    return tape.run(fn, args)


@contextlib.contextmanager
def paused():
    """Code whose results are not differentiated (side-effect tensors): nothing is recorded."""
    tape = getattr(_tls, 'tape', None)
    _tls.tape = None
    try:
        yield
    finally:
        _tls.tape = tape


class Tape:
    def __init__(self, leaves):
        self.req = {id(t) for t in leaves}        # tensors a gradient has to reach
        self.nodes = []

    def run(self, fn, args):
        req = self.req
        needs = tuple(torch.is_tensor(a) and id(a) in req for a in args)
        ctx = SubCtx(needs)
        out = fn.forward(ctx, *args)
        if True in needs:
            req.add(id(out))
            self.nodes.append([fn, ctx, args, needs, id(out), out])     # holding `out` keeps the ids unique
        return out

    def release(self, out):
        """The model output gets a grad_fn that owns this tape: drop the tape's own references to it (no cycle)."""
        for node in self.nodes:
            if node[5] is out:
                node[5] = None

    def backward(self, out_id, grad):
        grads = {out_id: (grad, False)}
        for fn, ctx, args, needs, oid, _ in reversed(self.nodes):
            ent = grads.pop(oid, None)
            if ent is None:                       # output never used downstream
                continue
            g = ent[0]
            res = fn.backward(ctx, g)
            if not isinstance(res, tuple):
                res = (res,)
            for i, need in enumerate(needs):
                if not need:
                    continue
                ga = res[i]
                if ga is None:
                    continue
                fresh = ga is not g
                if fresh:
                    for j, other in enumerate(res):
                        if other is ga and j != i:
                            fresh = False
                            break
                key = id(args[i])
                cur = grads.get(key)
                if cur is None:
                    grads[key] = (ga, fresh)
                elif cur[1]:
                    ops.add_inplace(cur[0], ga.contiguous())
                elif fresh:
                    grads[key] = (ops.add_inplace(ga, cur[0].contiguous()), True)
                else:
                    grads[key] = (ops.add(cur[0].contiguous(), ga.contiguous()), True)
        return grads


class ModelFn(torch.autograd.Function):
    """run(*) under a tape as one autograd node; `params` are the leaves."""

    @staticmethod
    def forward(ctx, run, state, *params):
        ctx.state = state
        tape = Tape(params)
        prev = getattr(_tls, 'tape', None)
        _tls.tape = tape
        try:
            out = run()
        finally:
            _tls.tape = prev
        tape.release(out)
        ctx.tape, ctx.out_id, ctx.params = tape, id(out), params
        ctx.on_side_stream = out.is_cuda and streams.on_side_stream(out.device)
        return out

    @staticmethod
    def backward(ctx, grad):
        streams.invalidate_step()
        direct = DIRECT_PARAM_GRADS and _plain_leaves(ctx.params)
        if direct and ASYNC_SIDE_BACKWARD and ctx.on_side_stream:
            # The model ran next to another one on the side stream (streams.py: Net3D beside PNA) and autograd has made
            # that stream current for this node: hand the whole backward pass to a helper thread and return, so that
            # autograd's worker goes on with the other model's backward.  The two Python threads take turns at C-call
            # granularity (every kernel launch releases the GIL); kernels, streams and their order are as before.  A
            # callback at the end of the backward pass joins the helper and orders the caller's stream after it.
            job = _Job(torch.cuda.current_stream(grad.device), ctx, grad.contiguous())
            _helper().put(job)
            torch.autograd.Variable._execution_engine.queue_callback(job.join)
            return (None,) * (2 + len(ctx.params))
        out = _model_backward(ctx, grad.contiguous(), direct)
        if direct:
            # `.grad` is empty and nothing hooks the parameters: store the gradients here instead of sending them
            # through ~110 AccumulateGrad nodes (each a task of the autograd engine; and a gradient that is a view of
            # the all-reduce buffer would be CLONED there, because the buffer's own views keep it alive)
            for p, g in zip(ctx.params, out):
                if g is not None:
                    p.grad = g
            return (None,) * (2 + len(ctx.params))
        return (None, None) + tuple(out)


class _GradPool:
    """Persistent gradient buffers of one model: views of one flat buffer (the data-parallel all-reduce buffer when a
    GradReducer is attached), handed to the weight-gradient kernels as their outputs, so a backward pass allocates
    nothing per parameter and - data parallel - copies nothing into the all-reduce buffer."""

    def __init__(self, module, params, views=None):
        self.key = tuple(id(p) for p in params)
        if views is None:
            p0 = params[0]
            sizes = [p.numel() for p in params]
            flat = torch.empty(sum(sizes), dtype=p0.dtype, device=p0.device)
            views = {id(p): v.view_as(p) for p, v in zip(params, flat.split(sizes))
                     if p.dtype == p0.dtype and p.device == p0.device}
        self.view_of = views
        self.bias_of = {}       # id(weight) -> bias parameter of the same Linear / BatchNorm module
        for m in module.modules():
            w, b = getattr(m, 'weight', None), getattr(m, 'bias', None)
            if isinstance(w, torch.nn.Parameter) and isinstance(b, torch.nn.Parameter):
                self.bias_of[id(w)] = b
        self.used = set()


class _ModelState:
    """what this file keeps per model; lives in the module's __dict__ (so it dies with the module: nothing is keyed by
    the id() of an object that may be gone)"""

    def __init__(self, module):
        import weakref
        self.module = weakref.ref(module)
        self.pool = None
        self.sink = None          # callable(params, grads) -> grads (dist.GradReducer)
        self.sink_views = None    # {id(parameter): its view in the sink's flat buffer}
        self.reducer = None       # the dist.GradReducer itself (early, overlapped all-reduce of finished gradients)

    def __reduce__(self):          # copy.deepcopy(model) / pickling: the copy starts without state (model_state rebuilds it)
        return (type(None), ())

    def pool_for(self, params):
        pool = self.pool
        if pool is None or len(pool.key) != len(params) or pool.key != tuple(id(p) for p in params):
            views = self.sink_views
            if views is not None and any(id(p) not in views or views[id(p)].shape != p.shape for p in params):
                views = None
            pool = self.pool = _GradPool(self.module(), params, views)
        return pool


def model_state(module):
    st = module.__dict__.get('_i3d_state')
    if st is None:
        st = module.__dict__['_i3d_state'] = _ModelState(module)
    return st


def grad_like(p):
    """output buffer for the gradient of parameter `p`: its persistent view inside a model backward, else a new tensor"""
    pool = getattr(_tls, 'pool', None)
    if pool is not None:
        v = pool.view_of.get(id(p))
        if v is not None and id(p) not in pool.used:      # a parameter used twice gets a second, separate buffer
            pool.used.add(id(p))
            return v
    return torch.empty_like(p)


def grad_for_bias_of(weight, n):
    """as grad_like for the bias that belongs to `weight` (the block Functions do not hold the bias parameter)"""
    pool = getattr(_tls, 'pool', None)
    if pool is not None:
        b = pool.bias_of.get(id(weight))
        if b is not None and b.numel() == n:
            return grad_like(b)
    return torch.empty(n, dtype=weight.dtype, device=weight.device)


# I3D_PERSISTENT_GRADS=0: every backward pass allocates its parameter gradients
PERSISTENT_GRADS = os.environ.get('I3D_PERSISTENT_GRADS', '1') != '0'


def _model_backward(ctx, grad, direct=False):
    if direct and PERSISTENT_GRADS:
        # `.grad` of every parameter is empty (nothing to accumulate into, nobody holds last step's buffers through it)
        pool = ctx.state.pool_for(ctx.params)
        pool.used.clear()
        _tls.pool = pool
        try:
            return _model_backward(ctx, grad)
        finally:
            _tls.pool = None
    grads = ctx.tape.backward(ctx.out_id, grad)
    out = [grads[id(p)][0] if id(p) in grads else None for p in ctx.params]
    sink = ctx.state.sink
    if sink is not None:         # data parallel: the gradients go straight into the all-reduce buffer (dist.GradReducer)
        out = sink(ctx.params, out)
    return out


# I3D_ASYNC_SIDE_BACKWARD=1: the side-stream model's backward pass is enqueued by a helper thread while autograd's worker
# goes on with the other model.  EXPERIMENTAL, off: measured on MI355X (tools/step_segments.py, tools/ab.sh) the two
# Python threads hand the GIL back and forth at every kernel launch and the backward segment gets SLOWER (1.85-2.0 ms
# against 1.46-1.6 ms); it needs the enqueue loop itself out of Python (a native whole-model composite) to pay off.
ASYNC_SIDE_BACKWARD = False


class _Job:
    """one model backward pass for the helper thread"""

    def __init__(self, stream, ctx, grad):
        self.stream, self.ctx, self.grad = stream, ctx, grad
        self.done = threading.Event()
        self.error = None
        self.event = None

    def run(self):
        try:
            with torch.no_grad(), torch.cuda.stream(self.stream):      # grad mode and current stream are per thread
                out = _model_backward(self.ctx, self.grad, True)
                for p, g in zip(self.ctx.params, out):
                    if g is not None:
                        p.grad = g
                self.event = torch.cuda.Event()
                self.event.record(self.stream)
        except BaseException as e:      # re-raised in the thread that called backward()
            self.error = e
        finally:
            self.ctx = self.grad = None
            self.done.set()

    def join(self):
        """end-of-backward callback (runs with the caller's streams current)"""
        self.done.wait()
        if self.error is not None:
            raise self.error
        torch.cuda.current_stream(self.stream.device).wait_event(self.event)


_helper_queue = None
_helper_lock = threading.Lock()


def _helper():
    global _helper_queue
    if _helper_queue is None:
        with _helper_lock:
            if _helper_queue is None:
                import queue
                q = queue.SimpleQueue()

                def loop():
                    while True:
                        q.get().run()
                threading.Thread(target=loop, name='i3d-side-backward', daemon=True).start()
                _helper_queue = q
    return _helper_queue


# I3D_DIRECT_PARAM_GRADS=0: hand the parameter gradients to autograd's AccumulateGrad nodes instead
DIRECT_PARAM_GRADS = os.environ.get('I3D_DIRECT_PARAM_GRADS', '1') != '0'


def _plain_leaves(params):
    """no gradient to accumulate into, no tensor hooks, no post-accumulate hooks: `p.grad = g` is what autograd would do"""
    for p in params:
        if p.grad is not None or p._backward_hooks is not None or p._post_accumulate_grad_hooks is not None:
            return False
    return True


def register_grad_sink(module, fn, views=None):
    """`fn(params, grads)` receives the parameter gradients of `module` at the end of its backward pass and returns the
    tensors autograd should store in `.grad`.  `views` ({id(parameter): tensor}): the sink's own per-parameter buffers -
    the weight-gradient kernels then write into them directly (_GradPool)."""
    st = model_state(module)
    st.sink, st.sink_views, st.pool = fn, views, None


# I3D_FUSED_MODEL=0: one autograd node per block (or per PNA layer) instead of one per model
FUSED_MODEL = True


def _param_list(module):
    """`list(module.parameters())`, cached in the module's __dict__ - with a cheap validity check (~15 us for PNA): every
    cached Parameter must still be the object its owner's `_parameters` dict holds under that name, every sub-module the
    object its parent's `_modules` dict holds, and both dicts must have kept their sizes (a head swapped for fine-tuning,
    a parameter re-assigned or added after the first forward would otherwise silently get no gradient)."""
    ent = module.__dict__.get('_i3d_param_list')
    if ent is not None:
        plist, owned, sizes = ent
        ok = True
        for d, name, obj in owned:
            if d.get(name) is not obj:
                ok = False
                break
        if ok:
            for d, n in sizes:
                if len(d) != n:
                    ok = False
                    break
        if ok:
            return plist
    owned, sizes, seen, plist = [], [], set(), []
    for m in module.modules():
        sizes.append((m._modules, len(m._modules)))
        sizes.append((m._parameters, len(m._parameters)))
        for name, child in m._modules.items():
            owned.append((m._modules, name, child))
        for name, p in m._parameters.items():
            owned.append((m._parameters, name, p))
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                plist.append(p)
    module.__dict__['_i3d_param_list'] = (plist, owned, sizes)
    return plist


def run_model(module, run):
    """module-level entry: `run()` is the plain forward of `module`.  Falls back to per-block autograd nodes when
    gradients are off, a tape is already recording, or nothing is trainable."""
    if not FUSED_MODEL or not torch.is_grad_enabled() or getattr(_tls, 'tape', None) is not None:
        return run()
    cached = _param_list(module)
    params = [p for p in cached if p.requires_grad]
    if not params or not params[0].is_cuda:
        return run()
    return ModelFn.apply(run, model_state(module), *params)
