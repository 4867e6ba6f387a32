"""f4: CFG-batched forward + fused DDIM step vs the oracle restatement (oracle/sampler_ref.py)."""
import pytest
import torch

from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd.sampler import NativeDDIMSampler
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
from oracle.sampler_ref import cfg_ddim_step, sample
from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel, ddpm_alphas_cumprod, seeded_init_


def test_cfg_ddim_step_kernel(backend):
    g = torch.Generator().manual_seed(0)
    x, eu, ec = (torch.randn(3, 4, 8, 8, generator=g) for _ in range(3))
    ref = cfg_ddim_step(x, eu, ec, 0.37, 0.52, 6.0)
    out = K.cfg_ddim_step(backend.to(x), backend.to(torch.cat([eu, ec])), 0.37, 0.52, 6.0).cpu()
    assert (out - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    ref1 = cfg_ddim_step(x, eu, None, 0.37, 1.0, 1.0)                   # unguided, final step (a_prev = 1 -> x0)
    out1 = K.cfg_ddim_step(backend.to(x), backend.to(eu), 0.37, 1.0).cpu()
    assert (out1 - ref1).abs().max().item() < 2e-5 * ref1.abs().max().item()
    xin = backend.to(x.clone())
    K.cfg_ddim_step(xin, backend.to(torch.cat([eu, ec])), 0.37, 0.52, 6.0, out=xin)         # in place
    assert torch.equal(xin.cpu(), out)


def test_sampler_loop_vs_oracle(backend):
    torch.manual_seed(0)
    ora = seeded_init_(OracleUNet2DConditionModel(**MICRO_CONFIG), 1)
    nat = NativeUNet2DConditionModel(**MICRO_CONFIG)
    nat.load_state_dict(ora.state_dict())
    nat.to(backend.device)
    g = torch.Generator().manual_seed(2)
    lat = torch.randn(2, 4, 8, 8, generator=g); cond = torch.randn(2, 24, 32, generator=g); unc = torch.randn(2, 24, 32, generator=g)
    ref = sample(ora, lat, cond, unc, ddpm_alphas_cumprod(), guidance_scale=5.0, num_inference_steps=4)
    s = NativeDDIMSampler()
    assert s.timesteps(4).tolist() == [751, 501, 251, 1]
    out = s.sample(nat, backend.to(lat), backend.to(cond), backend.to(unc), guidance_scale=5.0, num_inference_steps=4).cpu()
    assert ((out - ref).norm() / ref.norm()).item() < 3e-2            # bf16 UNet, 4 accumulated guided steps
    out1 = s.sample(nat, backend.to(lat), backend.to(cond), None, num_inference_steps=2).cpu()       # no guidance: B-row forward
    assert torch.isfinite(out1).all() and out1.shape == lat.shape
    # a [2B, L] key mask in the reference's order [negative prompts; prompts] on an UNGUIDED call uses the prompts' rows
    # (not the negatives', not a shape error); any other row count is refused with a clear message
    m_neg = torch.ones(2, 24, dtype=torch.bool); m_pos = torch.ones(2, 24, dtype=torch.bool); m_pos[:, 16:] = False
    a = s.sample(nat, backend.to(lat), backend.to(cond), None, num_inference_steps=2, encoder_attention_mask=backend.to(torch.cat([m_neg, m_pos]))).cpu()
    b = s.sample(nat, backend.to(lat), backend.to(cond), None, num_inference_steps=2, encoder_attention_mask=backend.to(m_pos)).cpu()
    assert torch.equal(a, b) and not torch.equal(a, out1)
    with pytest.raises(ValueError, match="encoder_attention_mask"):
        s.sample(nat, backend.to(lat), backend.to(cond), None, num_inference_steps=2, encoder_attention_mask=backend.to(torch.ones(3, 24, dtype=torch.bool)))


def test_ddpm_scheduler_beta_schedules(backend):
    """Seam 4 (`model.noise_scheduler`, train_base.yaml:76): the three beta schedules of the DDPMScheduler the reference instantiates —
    published definitions, checked against closed forms — and add_noise over each table through the native kernel."""
    import math
    from hcp_diffusion_amd.scheduler import NativeDDPMScheduler
    sl = NativeDDPMScheduler()
    assert torch.allclose(sl.alphas_cumprod, ddpm_alphas_cumprod(), atol=0, rtol=0)
    lin = NativeDDPMScheduler(beta_schedule="linear", beta_start=1e-4, beta_end=0.02)
    ref = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000, dtype=torch.float64), 0)
    assert (lin.alphas_cumprod.double() - ref).abs().max().item() < 1e-6
    cos = NativeDDPMScheduler(beta_schedule="squaredcos_cap_v2")
    bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    # uncapped betas telescope: alphas_cumprod[i] = bar((i + 1) / T) / bar(0) until the 0.999 cap bites (only the very last steps)
    for i in (0, 10, 500, 900):
        assert abs(cos.alphas_cumprod[i].item() - bar((i + 1) / 1000) / bar(0)) < 1e-5
    assert cos.alphas_cumprod[-1].item() > 0 and (cos.alphas_cumprod[1:] < cos.alphas_cumprod[:-1]).all()
    g = torch.Generator().manual_seed(1)
    x0, noise = torch.randn(3, 4, 8, 8, generator=g), torch.randn(3, 4, 8, 8, generator=g)
    t = torch.tensor([0, 417, 999])
    for sch in (sl, lin, cos):
        a = sch.alphas_cumprod[t].view(-1, 1, 1, 1)
        want = a.sqrt() * x0 + (1 - a).sqrt() * noise
        got = sch.add_noise(backend.to(x0), backend.to(noise), backend.to(t)).cpu()
        assert (got - want).abs().max().item() < 1e-5
    import pytest
    with pytest.raises(NotImplementedError):
        NativeDDPMScheduler(beta_schedule="sigmoid")
