"""GECO Lagrange-multiplier objective -- device-resident mirror of the reference's `utils/geco.py`
(GECO.__init__ :19-28, to_cuda :30-33, loss :35-51), same constructor and attributes (`beta`,
`err_ema`, `goal`, `step_size`, `alpha`, `beta_min`, `beta_max`, `speedup`).

Difference in mechanism, not in arithmetic: the state {beta, err_ema} lives in one small device
tensor and is updated by a HIP kernel (gx_geco_update), so `loss()` never synchronises with the host
(the reference's `constraint.item()`, geco.py:45, costs a device->host sync per iteration) and the
whole step can be captured in a HIP graph."""
import ctypes

import torch

from . import _lib


class GECO(object):

    def __init__(self, goal, step_size, alpha=0.99, beta_init=1.0, beta_min=1e-10, speedup=None,
                 device='cuda'):
        self.goal = float(goal)
        self.step_size = float(step_size)
        self.alpha = float(alpha)
        self.speedup = speedup
        self._beta_min = float(beta_min)
        self._beta_max = 1e10
        # {beta, err_ema, initialised}
        self.state = torch.tensor([float(beta_init), 0.0, 0.0], dtype=torch.float32, device=device)

    # -- reference-compatible attribute surface (train.py:198-204,268-272 read / restore these)
    @property
    def beta(self):
        return self.state[0]

    @beta.setter
    def beta(self, value):
        self.state[0] = float(value)

    @property
    def err_ema(self):
        return self.state[1] if float(self.state[2]) != 0.0 else None

    @err_ema.setter
    def err_ema(self, value):
        if value is None:
            self.state[2] = 0.0
        else:
            self.state[1] = float(value)
            self.state[2] = 1.0

    @property
    def beta_min(self):
        return torch.tensor(self._beta_min)

    @property
    def beta_max(self):
        return torch.tensor(self._beta_max)

    def to_cuda(self):
        self.state = self.state.cuda()

    def update(self, err, step=None):
        """geco.py:39-49 with `err` a device scalar (batch-mean reconstruction error).  step: optional device int64
        counter (the optimiser's) incremented by the same launch."""
        err = err.detach().reshape(1).to(torch.float32).contiguous()
        args = (ctypes.c_void_p(self.state.data_ptr()), ctypes.c_void_p(err.data_ptr()),
                self.goal, self.step_size, self.alpha, float(self.speedup or 0.0),
                int(self.speedup is not None), self._beta_min, self._beta_max)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if step is None:
            _lib.call('gx_geco_update', *args, stream)
        else:
            _lib.call('gx_geco_update_step', *args, ctypes.c_void_p(step.data_ptr()), stream)

    def loss(self, err, kld):
        # loss with the CURRENT beta (geco.py:37), then the no-grad multiplier update
        beta = self.state[0].clone()
        loss = err + beta * kld
        self.update(err)
        return loss


def make_geco(img_size, g_goal=0.5655, g_lr=1e-5, g_alpha=0.99, g_init=1.0, g_min=1e-10, g_speedup=10,
              device='cuda'):
    """GECO configured as train.py:159-167 does from its flags."""
    return GECO(g_goal * 3 * img_size ** 2, g_lr * (64 ** 2 / img_size ** 2), g_alpha, g_init, g_min, g_speedup,
                device=device)
