"""Optimizer wrapper with the reference's interface (src/optim.py:4-54): a torch.optim optimiser whose
learning rate is set before every step from a Noam-style schedule ('warmup' 4000 / 'decay' 1000 steps,
'fixed' otherwise), plus the linear teacher-forcing schedule `pre_step` returns.  The optimiser
itself is torch's (ROCm) -- SURVEY.md lists it as reused, not rebuilt."""
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
        self.opt = getattr(torch.optim, optimizer)(parameters, lr=1.0 if self.knee else lr)

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

    def step(self):
        self.opt.step()

    def create_msg(self):
        return ['Optim.spec.| Algo. = {}\t| Lr/sampling/rec.loss scheduler = {}/{}/{}'.format(
            self.opt_type, self.sch_type, self.tf_type, self.recon_sch)]
