"""HIP-stream helpers: run independent kernels of one backward node on a second stream.

At batch 512 most GEMMs of the step launch 130-520 workgroups on a 256-CU chip that could hold >1000 of them, so
the weight-gradient GEMMs (dW = dY^T X, which nothing downstream of the node depends on) are enqueued on a side
stream next to the data-gradient GEMMs of the same node and joined before the node returns.  Ordering rules:
  * the side stream first waits for the main stream (its inputs - grad_pre, saved activations - were produced there);
  * every tensor the side kernels read is `record_stream`ed so the caching allocator does not recycle it early;
  * outputs are allocated on the main stream's pool BEFORE switching streams and the main stream waits for the side
    stream before the node returns, so autograd (AccumulateGrad, the next node) only ever sees completed tensors.
EXPERIMENTAL, off by default (I3D_OVERLAP=1 enables it) and only wired into the per-kernel (I3D_COMPOSITE=0) backward
paths: at batch 512 the host's enqueue rate is as much on the critical path as the GPU, where the extra stream
bookkeeping costs more than the overlap wins.  (A large-batch gradient check that disagreed with it switched on
turned out to be arg-max routing flipping on near-ties between two summation orders - DESIGN.md section 6 - not a
race.)
"""
import os
import threading

import torch

ENABLED = False
_tls = threading.local()


def _side(device):
    pool = getattr(_tls, 'pool', None)
    if pool is None:
        pool = _tls.pool = {}
    s = pool.get(device.index)
    if s is None:
        s = pool[device.index] = torch.cuda.Stream(device=device)
    return s


# thread the peer exchange's side-stream context belongs to (dist.enable_native_sync / disable_native_sync), or None
BOUND_THREAD = None


def side_stream(device):
    """this thread's side stream on `device` (created on first use): the stream the 3D network runs on beside the 2D network"""
    return _side(torch.device(device))


class fork:
    """with fork(t1, t2, ...) as f:  kernels launched inside run on the side stream; `f.join()` (or leaving the
    `with` and calling join later in the same node) makes the main stream wait for them."""

    def __init__(self, *reads):
        self.reads = [t for t in reads if t is not None]
        self.active = ENABLED and len(self.reads) > 0 and self.reads[0].is_cuda
        self.main = self.side = self.ctx = None

    def __enter__(self):
        if self.active:
            self.main = torch.cuda.current_stream()
            self.side = _side(self.reads[0].device)
            self.side.wait_stream(self.main)
            for t in self.reads:
                t.record_stream(self.side)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.active:
            self.main.wait_stream(self.side)


# ---- the 3D network next to the 2D network ---------------------------------------------------------------------------
# Net3D is ~85 small launches per step (140k edges x 20 features: 5-15 us kernels that occupy a fraction of the chip)
# and is independent of PNA until the loss.  When the caller runs `model(g2d)` and then `model3d(g3d)` - the reference's
# forward_pass, trainer/self_supervised_trainer.py:24-29 - Net3D's kernels go to a side stream that only waits for (a)
# everything the main stream held when PNA.forward was ENTERED (so the previous optimizer step is in, PNA's own kernels
# are not) and (b) the event the batch carries from its own assembly (graph.py: BatchedMolGraph.ready_event).  The main
# stream waits for the side stream before Net3D.forward returns, so every consumer of its output is ordered as before;
# autograd runs the backward of Net3D on the same side stream and joins it.  Any other call pattern (no step-start
# mark, a backward pass in between, a foreign graph object without the event) keeps Net3D on the caller's stream.
# I3D_NET3D_STREAM=0 switches it off.
NET3D_STREAM = os.environ.get('I3D_NET3D_STREAM', '1') != '0'


def note_step_start(device):
    """PNA.forward entry: remember the main stream's position."""
    if not NET3D_STREAM:
        return
    ev = getattr(_tls, 'step_event', None)
    if ev is None:
        ev = _tls.step_event = torch.cuda.Event()
    stream = torch.cuda.current_stream(device)
    ev.record(stream)
    _tls.step_valid = (device.index, stream.cuda_stream, _generation[0])


_generation = [0]      # bumped by every model backward pass (any thread): a mark taken before it is stale


def invalidate_step():
    """a backward pass ran: parameters may change before the next forward"""
    _generation[0] += 1


def on_side_stream(device):
    """is this thread's current stream its side stream?"""
    pool = getattr(_tls, 'pool', None)
    if not pool:
        return False
    s = pool.get(device.index)
    return s is not None and s.cuda_stream == torch._C._cuda_getCurrentRawStream(device.index)


def side_stream_for(graph_event, device):
    """The side stream, already ordered after the step-start mark and the batch's own event - or None."""
    if not NET3D_STREAM or graph_event is None:
        return None
    if BOUND_THREAD is not None and BOUND_THREAD != threading.get_ident():
        # dist.enable_native_sync bound the peer exchange's second context to the side stream of ANOTHER thread: this thread's
        # side stream has no context (its collectives would share the default one with the 2D network's stream)
        raise RuntimeError('3dinfomax_amd: synchronised BatchNorm (peer exchange) was set up on another thread; run the '
                           'training loop on the thread that called dist.setup / enable_native_sync')
    main = torch.cuda.current_stream(device)
    if getattr(_tls, 'step_valid', None) != (device.index, main.cuda_stream, _generation[0]):
        return None
    _tls.step_valid = None                      # one consumer per mark
    side = _side(device)
    side.wait_event(_tls.step_event)
    side.wait_event(graph_event)
    return side
