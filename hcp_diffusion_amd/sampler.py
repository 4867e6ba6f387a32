"""Inference over the native UNet: the classifier-free-guidance batched forward of the reference's pipeline
hook (utils/pipe_hook.py:120-140: ``torch.cat([latents] * 2)`` -> ONE UNet call -> ``uncond + scale (text - uncond)``) and a DDIM
(eta = 0) step, fused into one kernel per step (``hcp_cfg_ddim_step``).  This is what the in-training previewer
(loggers/preview/image_previewer.py:97-149) needs from the UNet side; the VAE decoder and the prompt pipeline stay the reference's.

``NativeDDIMSampler.sample`` runs the whole denoising loop on the device with no host synchronisation."""
import torch

from . import kernels as K
from .trainer import ddpm_alphas_cumprod


class NativeDDIMSampler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.acp = ddpm_alphas_cumprod(num_train_timesteps, beta_start, beta_end)       # host copy: the step coefficients are scalars
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset

    def timesteps(self, num_inference_steps):
        """Leading spacing with the Stable Diffusion offset (diffusers DDIMScheduler, timestep_spacing='leading', steps_offset=1)."""
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (torch.arange(num_inference_steps) * ratio).flip(0) + self.steps_offset
        return ts.clamp(max=self.num_train_timesteps - 1)

    @torch.no_grad()
    def sample(self, unet, latents, cond, uncond=None, guidance_scale=7.5, num_inference_steps=20, encoder_attention_mask=None,
               added_cond_kwargs=None, uncond_added_cond_kwargs=None):
        """latents [B,4,h,w] fp32 unit-variance noise; cond / uncond [B,L,D] text states.  encoder_attention_mask: [B,L] (the same key
        mask for both halves of the guided batch) or [2B,L] in the reference's order [negative prompts; prompts] — what the previewer
        hands the pipeline when ``encoder_attention_mask`` is on (image_previewer.py:133-135,147).  Returns the denoised latents."""
        dev = latents.device
        x = latents.float().contiguous().clone()
        B = x.shape[0]
        guided = uncond is not None and guidance_scale != 1.0
        ehs = torch.cat([uncond, cond]) if guided else cond
        added = None
        if added_cond_kwargs is not None:
            u = uncond_added_cond_kwargs or added_cond_kwargs
            added = {k: torch.cat([u[k], v]) if guided else v for k, v in added_cond_kwargs.items()}
        mask = None
        if encoder_attention_mask is not None:
            rows = encoder_attention_mask.shape[0]
            if rows == 2 * B:                                      # reference order [negative prompts; prompts]
                mask = encoder_attention_mask if guided else encoder_attention_mask[B:]      # unguided: the prompts' rows only
            elif rows == B:
                mask = torch.cat([encoder_attention_mask] * 2) if guided else encoder_attention_mask
            else:
                raise ValueError(f"encoder_attention_mask has {rows} rows for a batch of {B}: expected [B, L] or [2B, L] "
                                 "([negative prompts; prompts])")
        ts = self.timesteps(num_inference_steps)
        ratio = self.num_train_timesteps // num_inference_steps
        for t in ts.tolist():
            xin = torch.cat([x, x]) if guided else x                      # pipe_hook.py:121
            tt = torch.full((xin.shape[0],), t, dtype=torch.long, device=dev)
            kw = {}
            if mask is not None:
                kw["encoder_attention_mask"] = mask
            if added is not None:
                kw["added_cond_kwargs"] = added
            eps2 = unet(xin, tt, ehs, **kw).sample.float().contiguous()
            prev = t - ratio
            a_t = float(self.acp[t]); a_prev = float(self.acp[prev]) if prev >= 0 else float(self.acp[0])
            K.cfg_ddim_step(x, eps2, a_t, a_prev, guidance_scale, out=x)
        return x
