"""oracle/pin_with_diffusers.py is the recipe that pins the UNet / VAE / scheduler restatements the day `diffusers==0.26.1` is importable
(it is not, here: no network).  Untested code is a liability, so the recipe itself runs here against a MOCKED `diffusers` namespace:
once with stand-ins that agree with the oracle (every comparison must pass, parameter names must load, exit code 0) and once with a
UNet that deviates (the script must fail loudly, exit code 1).  The mock proves nothing about diffusers — only that the script's
argument mapping, state-dict loading, comparisons and exit codes work."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MOCK = r'''
import sys, types, torch
sys.path.insert(0, ROOT)
from oracle import unet_sd15 as U
from oracle.vae_ref import OracleVAEEncoder

class UNet2DConditionModel(U.OracleUNet2DConditionModel):
    def __init__(self, sample_size=64, in_channels=4, out_channels=4, layers_per_block=2, block_out_channels=(320,), down_block_types=(), up_block_types=(),
                 cross_attention_dim=768, norm_num_groups=32, attention_head_dim=8, transformer_layers_per_block=1, use_linear_projection=False,
                 addition_embed_type=None, addition_time_embed_dim=None, projection_class_embeddings_input_dim=None):
        super().__init__(in_channels=in_channels, out_channels=out_channels, layers_per_block=layers_per_block, block_out_channels=block_out_channels,
                         down_block_types=down_block_types, up_block_types=up_block_types, cross_attention_dim=cross_attention_dim,
                         norm_num_groups=norm_num_groups, num_attention_heads=attention_head_dim, transformer_layers_per_block=transformer_layers_per_block,
                         use_linear_projection=use_linear_projection, addition_embed_type=addition_embed_type, addition_time_embed_dim=addition_time_embed_dim,
                         projection_class_embeddings_input_dim=projection_class_embeddings_input_dim)
    def forward(self, *a, **k):
        out = super().forward(*a, **k)
        if DEVIATE:
            out.sample = out.sample * 1.001
        return out

class DDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear"):
        self.acp = U.ddpm_alphas_cumprod(num_train_timesteps, beta_start, beta_end)
    def add_noise(self, x0, n, t):
        return U.add_noise(x0, n, t, self.acp)

class AutoencoderKL(torch.nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(), layers_per_block=1, down_block_types=(), up_block_types=(), norm_num_groups=32):
        super().__init__()
        enc = OracleVAEEncoder(in_channels=in_channels, latent_channels=latent_channels, block_out_channels=block_out_channels,
                               layers_per_block=layers_per_block, norm_num_groups=norm_num_groups)
        for k, v in enc.named_children():
            self.add_module(k, v)
        self._enc = [enc]
    def encode(self, img):
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(parameters=self._enc[0].moments(img)))

d = types.ModuleType("diffusers")
d.__version__, d.__hcp_mock__ = "0.26.1", True
d.UNet2DConditionModel, d.DDPMScheduler, d.AutoencoderKL = UNet2DConditionModel, DDPMScheduler, AutoencoderKL
sys.modules["diffusers"] = d
sys.argv = ["pin_with_diffusers.py"]
import runpy
runpy.run_path(ROOT + "/oracle/pin_with_diffusers.py", run_name="__main__")
'''


def _run(deviate):
    code = "ROOT = %r\nDEVIATE = %r\n" % (ROOT, deviate) + MOCK
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)


def test_pinning_recipe_passes_against_an_agreeing_namespace():
    r = _run(False)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "recipe ran clean" in r.stdout and "MISMATCH" not in r.stdout and "parameter names differ" not in r.stdout
    for tag in ("[micro]", "[tiny]", "[tiny-sdxl]", "[add_noise]", "[vae encoder]"):
        assert tag in r.stdout, r.stdout
    assert not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "PINNED")) or "mock" not in open(os.path.join(ROOT, "oracle", "_ref", "PINNED")).read()


def test_pinning_recipe_fails_loudly_on_a_deviating_unet():
    r = _run(True)
    assert r.returncode == 1 and "MISMATCH" in r.stdout and "PINNING FAILED" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
