"""ORACLE / test infrastructure: plain fp32 PyTorch restatement of classifier-free guidance + one DDIM (eta = 0) step, as the
reference's pipeline hook drives its diffusers scheduler (hcpdiff/utils/pipe_hook.py:120-140: cat([latents] * 2), one UNet call,
``noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)``, ``scheduler.step``).  The DDIM update itself lives in
the un-vendored diffusers (DDIMScheduler.step, prediction_type 'epsilon', eta 0, clip_sample False, set_alpha_to_one False,
timestep_spacing 'leading', steps_offset 1 — Stable Diffusion's scheduler config) [ext]: parity unpinned for that part."""
import torch


def cfg_ddim_step(x, eps_uncond, eps_text, a_t, a_prev, guidance_scale):
    eps = eps_uncond + guidance_scale * (eps_text - eps_uncond) if eps_text is not None else eps_uncond
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


@torch.no_grad()
def sample(unet, latents, cond, uncond, alphas_cumprod, guidance_scale=7.5, num_inference_steps=20, steps_offset=1):
    T = alphas_cumprod.numel()
    ratio = T // num_inference_steps
    ts = ((torch.arange(num_inference_steps) * ratio).flip(0) + steps_offset).clamp(max=T - 1)
    x = latents.clone()
    for t in ts.tolist():
        tt = torch.full((x.shape[0],), t, dtype=torch.long)
        eu = unet(x, tt, uncond).sample
        ec = unet(x, tt, cond).sample
        prev = t - ratio
        a_prev = alphas_cumprod[prev] if prev >= 0 else alphas_cumprod[0]
        x = cfg_ddim_step(x, eu, ec, float(alphas_cumprod[t]), float(a_prev), guidance_scale)
    return x
