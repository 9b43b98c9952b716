"""Host-side fast path of torch.optim.Adam for the pre-training step.

The reference builds `Adam(param_groups, **optimizer_params)` by name (train.py:189, trainer/trainer.py:216-238) and
calls `optim.step()` once per batch (trainer/trainer.py:120).  SURVEY.md K12 keeps torch's optimizer - and this IS
torch's optimizer: the same state layout (`step`, `exp_avg`, `exp_avg_sq`, so `state_dict()` round-trips with
torch.optim.Adam), the same fused multi-tensor kernel (`torch._fused_adam_`).  The only difference is the Python in
front of the kernel: torch re-derives the tensor lists of every group on every call (~0.6 ms for the 110 parameter
tensors of PNA + Net3D, as much as a sixth of the whole step on this host); here they are cached and only the gradient
list is rebuilt.  Anything outside the plain case (closure, amsgrad, maximize, capturable, tensor lr, CPU or mixed
devices, a parameter without gradient) goes through torch's own `step()`.
"""
import ctypes
import math
import os
import struct

import torch

# I3D_NATIVE_ADAM=0: torch._fused_adam_ (three multi-tensor launches of ~50 workgroups) instead of csrc/adam.hip
NATIVE_ADAM = True


class _NativeTable:
    """device-resident chunk table of csrc/adam.hip for one set of (param, grad, exp_avg, exp_avg_sq) tensors"""

    def __init__(self, ps, grads, exp_avgs, exp_avg_sqs):
        from . import _lib
        L = _lib.load()
        ch, rec = L.i3d_adam_chunk_elems(), L.i3d_adam_chunk_bytes()
        assert rec == 40
        buf = bytearray()
        for p, g, m, v in zip(ps, grads, exp_avgs, exp_avg_sqs):
            n = p.numel()
            for o in range(0, n, ch):
                buf += struct.pack('<QQQQiI', p.data_ptr() + 4 * o, g.data_ptr() + 4 * o, m.data_ptr() + 4 * o,
                                   v.data_ptr() + 4 * o, min(ch, n - o), 0)
        self.n_chunks = len(buf) // rec
        self.table = torch.frombuffer(buf, dtype=torch.uint8).clone().to(ps[0].device)
        self.grads = list(grads)                                  # identity of these objects = validity of the table
        self.grad_ptrs = [g.data_ptr() for g in grads]
        self.param_ptrs = [p.data_ptr() for p in ps]

    def valid_for(self, ps, grads):
        if len(grads) != len(self.grads) or [p.data_ptr() for p in ps] != self.param_ptrs:
            return False                  # (a parameter whose storage moved: the table points at the old memory)
        for a, b in zip(grads, self.grads):
            if a is not b:
                # a different tensor object may still be the same memory (views re-created by autograd)
                return ([g.data_ptr() for g in grads] == self.grad_ptrs and [p.data_ptr() for p in ps] == self.param_ptrs
                        and all(g.is_contiguous() for g in grads))
        return True


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kwargs):
        params = list(params)      # a generator (`Adam(model.parameters())`) must survive the look at its devices below
        if 'fused' not in kwargs and 'foreach' not in kwargs:
            kwargs['fused'] = self._all_cuda(params)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kwargs)
        self._lists = None
        self._steps_flat = None
        self._merged = None
        self._native = None          # (_NativeTable, host step count)
        self._host_step = None
        self._steps_unequal = False
        self._gradless = []

    @staticmethod
    def _all_cuda(params):
        flat = []
        for p in params:
            flat += list(p['params']) if isinstance(p, dict) else [p]
        return len(flat) > 0 and all(t.is_cuda and t.dtype == torch.float32 for t in flat)

    def _plain(self, group):
        return (group.get('fused') and not group['amsgrad'] and not group['maximize'] and not group['capturable']
                and not group['differentiable'] and not isinstance(group['lr'], torch.Tensor)
                and not group.get('decoupled_weight_decay', False))

    def _build_lists(self):
        self._steps_unequal = False
        lists = []
        self._gradless = []          # parameters torch has never seen a gradient for (no state): skipped, as torch.optim.Adam does -
        #                              e.g. the reference's PNAGNNOriginal.MLP_layer, built but not used by forward
        for group in self.param_groups:
            ps = [p for p in group['params'] if p in self.state and len(self.state[p]) > 0]
            self._gradless += [p for p in group['params'] if not (p in self.state and len(self.state[p]) > 0)]
            if not ps:
                lists.append(None)
                continue
            if not self._plain(group) or any((not p.is_cuda) or p.dtype != torch.float32 or p.device != ps[0].device for p in ps):
                return None
            st = [self.state[p] for p in ps]
            if any((not torch.is_tensor(s['step'])) or (not s['step'].is_cuda) for s in st):
                return None
            lists.append((list(ps), [s['exp_avg'] for s in st], [s['exp_avg_sq'] for s in st], [s['step'] for s in st], len(group['params'])))
        # the step counters of all parameters become views of ONE tensor: a single add per step() instead of a
        # multi-tensor add per group (same values, same dtype; state_dict() keeps working, load_state_dict() rebuilds)
        every = [s for lst in lists if lst is not None for s in lst[3]]
        if every and all(s.dtype == torch.float32 and s.device == every[0].device and s.numel() == 1 for s in every):
            flat = torch.stack([s.reshape(()) for s in every])
            i = 0
            for lst in lists:
                if lst is None:
                    continue
                for j, p in enumerate(lst[0]):
                    view = flat[i].view(self.state[p]['step'].shape)
                    self.state[p]['step'] = lst[3][j] = view
                    i += 1
            self._steps_flat = flat
        else:
            self._steps_flat = None
        return lists

    def _native_fast(self):
        """steady state of the one-launch kernel: same gradient tensor objects as last step, same hyper-parameters ->
        straight to the launch (no tensor lists, no torch.no_grad context: nothing here touches autograd)"""
        nat = self._native
        for p in self._gradless:
            if p.grad is not None:       # a parameter that had no gradient so far has one now: torch creates its state
                return False
        for p, g, ptr in zip(nat.params, nat.grads, nat.param_ptrs):
            # same gradient objects AND the parameter's storage is the one the chunk table points into (a swap of
            # `p.data` - .to(), a checkpoint surgery - keeps p and p.grad alive but moves the memory)
            if p.grad is not g or p.data_ptr() != ptr:
                return False
        key = nat.key
        for group in self.param_groups:
            if (group['lr'], group['betas'], group['weight_decay'], group['eps']) != key or len(group['params']) != nat.group_sizes[id(group)]:
                return False
        t = self._host_step + 1
        from . import _lib, ops
        beta1, beta2 = key[1]
        _lib.check(_lib.load().i3d_adam_step(nat.table.data_ptr(), nat.n_chunks, self._steps_flat.data_ptr(),
                                             self._steps_flat.numel(), float(key[0]), float(beta1), float(beta2), float(key[2]),
                                             float(key[3]), 1 - beta1 ** t, math.sqrt(1 - beta2 ** t), ops._stream()),
                   'i3d_adam_step')
        self._host_step = t
        return True

    def step(self, closure=None):
        if (closure is None and NATIVE_ADAM and self._native is not None and self._host_step is not None
                and self._steps_flat is not None and self._native_fast()):
            return None
        return self._step_general(closure)

    @torch.no_grad()
    def _step_general(self, closure=None):
        if closure is not None:
            return super().step(closure)
        if self._lists is None or len(self._lists) != len(self.param_groups):
            out = super().step()              # torch initialises the state on its first call
            self._lists = self._build_lists()
            self._merged = None
            self._native = None
            self._host_step = None
            return out
        work = []
        if any(p.grad is not None for p in self._gradless):
            self._lists = None
            return super().step()
        for group, lst in zip(self.param_groups, self._lists):
            if lst is None:
                if any(p.grad is not None for p in group['params']):
                    self._lists = None
                    return super().step()
                continue
            ps = lst[0]
            if lst[4] != len(group['params']) or not self._plain(group):
                self._lists = None
                return super().step()
            grads = [p.grad for p in ps]
            if any(g is None for g in grads):
                self._lists = None
                return super().step()
            work.append((group, lst, grads))
        if NATIVE_ADAM and self._steps_flat is not None and self._native_step(work):
            return None
        if self._steps_flat is not None:
            self._steps_flat.add_(1)
        self._host_step = None
        if len(work) > 1 and self._steps_flat is not None:
            # groups with identical hyper-parameters (the reference's BatchNorm / other split with weight_decay 0 in both,
            # trainer/self_supervised_trainer.py:78-86) are ONE multi-tensor launch: same arithmetic per tensor
            g0 = work[0][0]
            key = (g0['lr'], g0['betas'], g0['weight_decay'], g0['eps'])
            if all((g['lr'], g['betas'], g['weight_decay'], g['eps']) == key for g, _, _ in work[1:]):
                merged = self._merged
                if merged is None:
                    merged = self._merged = tuple([t for _, lst, _ in work for t in lst[i]] for i in range(4))
                grads = [t for _, _, gr in work for t in gr]
                torch._fused_adam_(merged[0], grads, merged[1], merged[2], [], merged[3], amsgrad=False, lr=key[0],
                                   beta1=key[1][0], beta2=key[1][1], weight_decay=key[2], eps=key[3], maximize=False,
                                   grad_scale=None, found_inf=None)
                return None
        for group, (ps, exp_avgs, exp_avg_sqs, steps, _n), grads in work:
            beta1, beta2 = group['betas']
            if self._steps_flat is None:
                torch._foreach_add_(steps, 1)
            torch._fused_adam_(ps, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False, lr=group['lr'], beta1=beta1,
                               beta2=beta2, weight_decay=group['weight_decay'], eps=group['eps'], maximize=False,
                               grad_scale=None, found_inf=None)
        return None

    def _native_step(self, work):
        """every group with the same hyper-parameters, contiguous fp32 tensors: ONE launch of csrc/adam.hip for all
        parameters (and their step counters).  Returns False when it does not apply (torch's kernel runs instead)."""
        if self._steps_unequal:
            return False
        g0 = work[0][0]
        key = (g0['lr'], g0['betas'], g0['weight_decay'], g0['eps'])
        for g, _, _ in work[1:]:
            if (g['lr'], g['betas'], g['weight_decay'], g['eps']) != key:
                return False
        ps = [t for _, lst, _ in work for t in lst[0]]
        grads = [t for _, _, gr in work for t in gr]
        nat = self._native
        if nat is None or not nat.valid_for(ps, grads):
            ms = [t for _, lst, _ in work for t in lst[1]]
            vs = [t for _, lst, _ in work for t in lst[2]]
            if not all(t.is_contiguous() and t.dtype == torch.float32 for t in ps + grads + ms + vs):
                return False
            nat = self._native = _NativeTable(ps, grads, ms, vs)
            nat.params = ps
        # the table holds pointers only: hyper-parameters (an LR scheduler moves group['lr'] - the reference's WarmUpWrapper
        # / ReduceLROnPlateau) and re-created gradient views refresh the fast path's keys instead of switching it off
        nat.key, nat.grads = key, list(grads)
        nat.group_sizes = {id(g): len(g['params']) for g in self.param_groups}
        if self._host_step is None:                 # one read-back after construction / load_state_dict / a torch step
            # ONE step count for the whole launch: torch's kernel uses each parameter's own `step` for the bias
            # corrections, and they differ after add_param_group on a trained optimizer, a checkpoint in which some
            # parameters were skipped, unfreezing during fine-tuning - then torch's per-parameter kernel runs
            steps = self._steps_flat.cpu()
            if not bool((steps == steps[0]).all()):
                self._steps_unequal = True          # (remembered: no read-back per step; cleared with the tensor lists)
                return False
            self._host_step = int(round(float(steps[0].item())))
        t = self._host_step + 1
        from . import _lib, ops
        beta1, beta2 = key[1]
        bc1, bc2s = 1 - beta1 ** t, math.sqrt(1 - beta2 ** t)
        _lib.check(_lib.load().i3d_adam_step(nat.table.data_ptr(), nat.n_chunks, self._steps_flat.data_ptr(),
                                             self._steps_flat.numel(), float(key[0]), float(beta1), float(beta2), float(key[2]),
                                             float(key[3]), bc1, bc2s, ops._stream()), 'i3d_adam_step')
        self._host_step = t
        return True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._lists = None
        self._merged = None
        self._native = None
        self._host_step = None

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._lists = None
        self._merged = None
        self._native = None
