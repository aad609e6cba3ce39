"""Optimizer wrapper with the reference's interface (src/optim.py:4-54): a torch.optim optimiser whose
learning rate is set before every step from a Noam-style schedule ('warmup' 4000 / 'decay' 1000 steps,
'fixed' otherwise), plus the linear teacher-forcing schedule `pre_step` returns.  'Adam' on device
tensors runs on the multi-tensor HIP kernels (FusedAdam below); any other torch.optim
name is torch's own optimiser."""
import torch

_SCHEDULE_STEPS = {'warmup': 4000.0, 'decay': 1000.0}


class Optimizer:
    def __init__(self, parameters, optimizer, lr, lr_scheduler, tf_start=1, tf_end=1, tf_step=1,
                 recon_init_weight=1.0, recon_decay=0.0, **kwargs):
        self.tf_start, self.tf_end, self.tf_step = tf_start, tf_end, tf_step
        self.tf_type = tf_end != 1
        self.recon_sch = recon_init_weight != 1.0
        self.opt_type, self.sch_type, self.init_lr = optimizer, lr_scheduler, lr
        self.knee = _SCHEDULE_STEPS.get(lr_scheduler)
        # with a schedule the optimiser is built with lr 1.0 and the rate is overwritten every step
        parameters = list(parameters)
        on_device = bool(parameters) and all(p.is_cuda for p in parameters)
        cls = FusedAdam if (optimizer == 'Adam' and on_device) else getattr(torch.optim, optimizer)
        self.opt = cls(parameters, lr=1.0 if self.knee else lr)

    def tf_rate(self, step):
        return max(self.tf_end, self.tf_start - (self.tf_start - self.tf_end) * step / self.tf_step)

    def lr_at(self, step):
        if not self.knee:
            return self.init_lr
        return self.init_lr * self.knee ** 0.5 * min((step + 1) * self.knee ** -1.5, (step + 1) ** -0.5)

    def get_opt_state_dict(self):
        return self.opt.state_dict()

    def load_opt_state_dict(self, state_dict):
        self.opt.load_state_dict(state_dict)

    def pre_step(self, step):
        if self.knee:
            for group in self.opt.param_groups:
                group['lr'] = self.lr_at(step)
        self.opt.zero_grad()
        return self.tf_rate(step)

    def step(self, guard_norm=None):
        if guard_norm is not None and isinstance(self.opt, FusedAdam):
            self.opt.step(guard_norm=guard_norm)
        else:
            self.opt.step()

    def create_msg(self):
        return ['Optim.spec.| Algo. = {}\t| Lr/sampling/rec.loss scheduler = {}/{}/{}'.format(
            self.opt_type, self.sch_type, self.tf_type, self.recon_sch)]


# --------------------------------------------------------------------------------------------- multi-tensor HIP path
def _tables(tensors):
    import ctypes as C
    n = len(tensors)
    return (C.c_void_p * n)(*[t.data_ptr() for t in tensors]), (C.c_long * n)(*[t.numel() for t in tensors])


def clip_grad_norm_(parameters, max_norm, pre_scale=1.0):
    """torch.nn.utils.clip_grad_norm_ (L2) on the multi-tensor HIP kernels: returns the total norm as a 0-dim device
    tensor and scales every .grad in place by max_norm / (norm + 1e-6) when that is below 1.
    pre_scale (1 / world under data parallelism, parallel.GradReducer(defer_average=True)): the gradients are still the SUM over
    the ranks -- the norm returned is that of the averaged gradients and the average itself rides in the scale launch.
    ref: BaseSolver.backward src/solver.py:145"""
    from . import _lib, ops
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    for g in grads:
        if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
            raise RuntimeError('clip_grad_norm_: gradients must be contiguous fp32 device tensors (no CPU fallback)')
    lib = _lib.load()
    ptrs, sizes = _tables(grads)
    dev = grads[0].device
    partials = torch.empty(int(lib.st_mt_blocks(sizes, len(grads))), device=dev, dtype=torch.float32)
    norm = torch.empty((), device=dev, dtype=torch.float32)
    _lib.check(lib.st_mt_grad_norm_scaled(ptrs, sizes, len(grads), ops._p(partials), ops._p(norm), float(pre_scale), ops.stream_handle()),
               'st_mt_grad_norm')
    _lib.check(lib.st_mt_clip_scale_pre(ptrs, sizes, len(grads), ops._p(norm), float(max_norm), float(pre_scale), ops.stream_handle()),
               'st_mt_clip_scale')
    return norm


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) with the update of all
    parameter tensors in a handful of multi-tensor launches (st_mt_adam).  State keys (`step`, `exp_avg`,
    `exp_avg_sq`) and param_group keys are torch's, so optimizer state_dicts are interchangeable with the reference's
    checkpoints (src/solver.py:211-216)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError('FusedAdam implements the configuration the reference uses (no weight decay / amsgrad)')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None))

    def state_dict(self):
        # the per-parameter `step` tensors torch's Adam keeps are materialised only here (102 tiny host-tensor
        # updates per step would cost more than the update kernels)
        for group in self.param_groups:
            for p in group['params']:
                if p in self.state and '_step' in self.state[p]:
                    self.state[p]['step'] = torch.tensor(float(self.state[p]['_step']))
        sd = super().state_dict()
        # the packed state maps to the live per-parameter dicts: return copies without the private counter
        sd['state'] = {k: {n: v for n, v in st.items() if n != '_step'} for k, st in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if 'step' in st:
                st['_step'] = int(float(st['step']))
        self._cache = {}

    def rollback_step(self, which=None):
        """a guarded step turned out to have been skipped on the device (TtsTrainer._skipped_on_device): take it off the host-side
        step counts, so that the bias corrections of later steps and the checkpointed `step` agree with torch's Adam again.
        which: the value of `guarded_steps` just before that step() call -- only the parameters that took part in it (had a gradient)
        are rolled back; None (or a step too long ago to be remembered): every parameter that has ever stepped"""
        plist = self.__dict__.get('_guarded_log', {}).pop(which, None) if which is not None else None
        for st in ([self.state[p] for p in plist] if plist is not None else self.state.values()):
            if st.get('_step', 0) > 0:
                st['_step'] -= 1

    guarded_steps = 0        # step() calls with a guard so far (the key of rollback_step)

    @torch.no_grad()
    def step(self, closure=None, guard_norm=None):
        """guard_norm: optional 0-dim device tensor (the gradient norm of this step): a NaN / inf value makes the update a no-op ON THE
        DEVICE (the reference's host-side `if math.isnan(grad_norm)` skip without waiting for the norm).  The host-side step counts
        advance either way; the trainer calls rollback_step() when it reads the non-finite norm (at most STATS_WINDOW steps later;
        the bias corrections of the steps in between ran one step ahead)."""
        from . import _lib, ops
        lib = _lib.load()
        cache = self.__dict__.setdefault('_cache', {})
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            b1, b2 = group['betas']
            buckets = {}
            for p in ps:
                st = self.state[p]
                if 'exp_avg' not in st:
                    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                        raise RuntimeError('FusedAdam: parameters must be contiguous fp32 device tensors')
                    st['_step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['_step'] = st.get('_step', 0) + 1
                buckets.setdefault(st['_step'], []).append(p)
            for step, plist in buckets.items():        # parameters share one step count unless some were frozen for a while
                key = (gi, tuple(id(p) for p in plist))
                if key not in cache:                   # pointer tables of parameters and moments are stable across steps
                    pp, sizes = _tables(plist)
                    mp, _ = _tables([self.state[p]['exp_avg'] for p in plist])
                    vp, _ = _tables([self.state[p]['exp_avg_sq'] for p in plist])
                    cache[key] = (pp, mp, vp, sizes)
                pp, mp, vp, sizes = cache[key]
                grads = [p.grad for p in plist]
                if not all(g.is_contiguous() and g.dtype == torch.float32 for g in grads):
                    raise RuntimeError('FusedAdam: gradients must be contiguous fp32 tensors')
                gp, _ = _tables(grads)
                bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
                _lib.check(lib.st_mt_adam_guarded(pp, gp, mp, vp, sizes, len(plist), float(b1), float(b2), float(group['eps']),
                                                  float(group['lr'] / bc1), float(bc2 ** 0.5),
                                                  ops._p(guard_norm) if guard_norm is not None else None, ops.stream_handle()),
                           'st_mt_adam')
                # the kernel wrote the parameters behind torch's back: bump their version counters so that everything keyed on
                # them (tap-major / packed weight copies, autograd's saved-tensor checks) sees the in-place update
                for p in plist:
                    torch.autograd.graph.increment_version(p)
        if guard_norm is not None:       # which parameters this (possibly skipped) update advanced: what a rollback has to undo
            log = self.__dict__.setdefault('_guarded_log', {})
            log[self.guarded_steps] = [p for group in self.param_groups for p in group['params'] if p.grad is not None]
            if len(log) > 256:
                del log[min(log)]
            self.guarded_steps += 1
        return None
