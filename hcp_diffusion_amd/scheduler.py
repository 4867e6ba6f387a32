"""Seam 4 (``model.noise_scheduler``, train_base.yaml:76 -> train_ac.py:211): the slice of diffusers' DDPMScheduler the training
step uses — ``config.num_train_timesteps`` and ``add_noise`` (train_ac.py:437-447) — over the native kernel.

``add_noise(x0, noise, t) = sqrt(acp_t) x0 + sqrt(1 - acp_t) noise`` with the scaled-linear beta schedule of Stable Diffusion
(constants as in the reference's loggers/preview/image_previewer.py:28)."""
from types import SimpleNamespace

import torch

from . import kernels as K


class NativeDDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear"):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type="epsilon")
        # the three schedules of diffusers' DDPMScheduler (published definitions; Stable Diffusion trains with 'scaled_linear')
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "squaredcos_cap_v2":         # Nichol & Dhariwal's cosine schedule, betas capped at 0.999
            import math
            bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
            betas = torch.tensor([min(1 - bar((i + 1) / num_train_timesteps) / bar(i / num_train_timesteps), 0.999)
                                  for i in range(num_train_timesteps)], dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule!r}: 'scaled_linear', 'linear' and 'squaredcos_cap_v2' exist")
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self._acp_dev = {}

    def _acp(self, device):
        key = str(device)
        if key not in self._acp_dev:
            self._acp_dev[key] = self.alphas_cumprod.to(device)
        return self._acp_dev[key]

    def add_noise(self, original_samples, noise, timesteps):
        x0 = original_samples.float().contiguous()
        return K.add_noise(x0, noise.float().contiguous(), timesteps.long(), self._acp(x0.device)).to(original_samples.dtype)
