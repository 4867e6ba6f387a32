"""VAE encoder: native gfx950 path vs the CPU fp32 oracle (oracle/vae_ref.py — public diffusers AutoencoderKL
architecture, parity unpinned like the UNet) on seeded weights; diffusers-layout loading; the latent-cache file format of
the reference's PairDataset.cache_latents."""
import json
import os

import pytest
import torch

from hcp_diffusion_amd.vae import NativeAutoencoderKL, NativeVAEEncoder, build_latent_cache
from oracle.unet_sd15 import seeded_init_
from oracle.vae_ref import SD_VAE_CONFIG, TINY_VAE_CONFIG, OracleAutoencoderKL, OracleVAEEncoder

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pair(cfg, dev, seed=3):
    ora = seeded_init_(OracleVAEEncoder(**cfg), seed)
    nat = NativeVAEEncoder(**cfg)
    nat.load_state_dict(ora.state_dict())
    return ora, nat.to(dev)


def test_native_vae_names_match_oracle():
    with torch.device("meta"):
        a, b = OracleVAEEncoder(**SD_VAE_CONFIG), NativeVAEEncoder(**SD_VAE_CONFIG)
    sa, sb = a.state_dict(), b.state_dict()
    assert {k: tuple(v.shape) for k, v in sa.items()} == {k: tuple(v.shape) for k, v in sb.items()}
    assert sum(v.numel() for v in sa.values()) == 34163664                      # SD VAE encoder + quant_conv
    assert "encoder.mid_block.attentions.0.to_out.0.weight" in sa and "encoder.down_blocks.2.downsamplers.0.conv.bias" in sa


@pytest.mark.parametrize("hw", [(32, 32), (48, 24)])
def test_tiny_vae_encode_vs_oracle(backend, hw):
    """Latents (sampled with the same noise, and the mode) vs the fp32 oracle: relative L2 <= 2e-2 (bf16 activations)."""
    ora, nat = _pair(TINY_VAE_CONFIG, backend.device)
    g = torch.Generator().manual_seed(11)
    img = torch.rand(2, 3, *hw, generator=g) * 2 - 1
    noise = torch.randn(2, 4, hw[0] // 2, hw[1] // 2, generator=g)
    with torch.no_grad():
        zo, zo_mode = ora.encode(img, noise), ora.encode(img, None)
    zn = nat.encode_latents(backend.to(img), noise=backend.to(noise)).cpu()
    zn_mode = nat.encode_latents(backend.to(img), sample=False).cpu()
    # the diffusers contract the reference calls (pair_dataset.py:74-75): encode(...).latent_dist.sample() * config.scaling_factor
    dist = nat.encode(backend.to(img)).latent_dist
    assert torch.allclose(dist.sample(noise=backend.to(noise)).cpu() * nat.config.scaling_factor, zn, rtol=1e-6, atol=1e-7)
    assert torch.allclose(dist.mode().cpu() * nat.config.scaling_factor, zn_mode, rtol=1e-6, atol=1e-7)
    assert dist.sample().shape == zn.shape
    assert zn.shape == zo.shape and zn.dtype == torch.float32
    assert ((zn - zo).norm() / zo.norm()).item() < 2e-2
    assert ((zn_mode - zo_mode).norm() / zo_mode.norm()).item() < 2e-2
    with pytest.raises(ValueError):
        nat.encode(backend.to(img[:, :2]))
    with pytest.raises(NotImplementedError):
        nat.encode(backend.to(torch.zeros(1, 3, 20, 32)))


def test_latent_cache_format(backend, tmp_path):
    """{img_name: {'img': [L,h,w] cpu fp32 (already x scaling_factor), 'mask': [h,w]}} via torch.save, re-read on the next call
    (data/pair_dataset.py:60-79)."""
    _, nat = _pair(TINY_VAE_CONFIG, backend.device)
    g = torch.Generator().manual_seed(2)
    items = [("a.png", torch.rand(3, 32, 32, generator=g) * 2 - 1, None), ("b.png", torch.rand(3, 16, 48, generator=g) * 2 - 1, torch.zeros(8, 24)),
             ("a.png", torch.zeros(3, 32, 32), None)]
    path = str(tmp_path / "latents.pth")
    cache = build_latent_cache(nat, items, cache_path=path)
    assert set(cache) == {"a.png", "b.png"}
    assert tuple(cache["a.png"]["img"].shape) == (4, 16, 16) and cache["a.png"]["img"].device.type == "cpu" and cache["a.png"]["img"].dtype == torch.float32
    assert torch.equal(cache["a.png"]["mask"], torch.ones(16, 16)) and torch.equal(cache["b.png"]["mask"], torch.zeros(8, 24))
    again = build_latent_cache(nat, [], cache_path=path)
    assert torch.equal(again["b.png"]["img"], cache["b.png"]["img"])


def test_vae_from_pretrained_reads_diffusers_layout(backend, tmp_path):
    from safetensors.torch import save_file
    ora, _ = _pair(TINY_VAE_CONFIG, "cpu")
    root = tmp_path / "model" / "vae"
    root.mkdir(parents=True)
    json.dump(dict(TINY_VAE_CONFIG, _class_name="AutoencoderKL", out_channels=3), open(root / "config.json", "w"))
    sd = {k: v.contiguous() for k, v in ora.state_dict().items()}
    sd["decoder.conv_in.weight"] = torch.zeros(4, 4, 3, 3)                    # decoder keys are ignored
    save_file(sd, str(root / "diffusion_pytorch_model.safetensors"))
    nat = NativeVAEEncoder.from_pretrained(str(tmp_path / "model"), device=backend.device)
    assert nat.config["block_out_channels"] == (32, 64)
    for k, v in nat.state_dict().items():
        assert torch.equal(v.cpu(), ora.state_dict()[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("side", [256, 512, 1024])
def test_sd_vae_full_size_encode_vs_golden(side):
    """Full SD VAE encoder (34.2 M parameters, seeded weights) on one seeded image vs the oracle's latents committed in
    tests/golden/vae_full_oracle.pt (oracle/make_golden.py vae)."""
    g = torch.load(os.path.join(GOLD, "vae_full_oracle.pt"))[side]
    nat = seeded_init_(NativeVAEEncoder(**SD_VAE_CONFIG), g["seed"]).to("cuda")
    gen = torch.Generator().manual_seed(g["input_seed"])
    img = torch.rand(1, 3, side, side, generator=gen) * 2 - 1
    noise = torch.randn(1, 4, side // 8, side // 8, generator=gen)
    z = nat.encode_latents(img.cuda(), noise=noise.cuda()).cpu()
    assert ((z - g["latents"]).norm() / g["latents"].norm()).item() < 2e-2


# ---------------------------------------------------------------------------------------------------------------- decoder (f4)
def test_native_autoencoder_names_match_oracle():
    with torch.device("meta"):
        a, b = OracleAutoencoderKL(**SD_VAE_CONFIG), NativeAutoencoderKL(**SD_VAE_CONFIG)
    sa, sb = a.state_dict(), b.state_dict()
    assert {k: tuple(v.shape) for k, v in sa.items()} == {k: tuple(v.shape) for k, v in sb.items()}
    assert sum(v.numel() for v in sa.values()) == 83653863                      # the SD VAE (diffusers AutoencoderKL), encoder + decoder
    assert "decoder.up_blocks.0.upsamplers.0.conv.weight" in sa and "decoder.up_blocks.3.resnets.2.conv2.bias" in sa and "post_quant_conv.bias" in sa
    assert "decoder.up_blocks.2.resnets.0.conv_shortcut.weight" in sa and "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sa


@pytest.mark.parametrize("hw", [(16, 16), (24, 8)])
def test_tiny_vae_decode_vs_oracle(backend, hw):
    """``vae.decode(z, return_dict=False)[0]`` (pipe_hook.py:155) vs the fp32 oracle: relative L2 <= 2e-2 (bf16 activations); the
    folded post_quant_conv is exact at the image border (compared on the border rows alone); slicing = the same images."""
    ora = seeded_init_(OracleAutoencoderKL(**TINY_VAE_CONFIG), 4)
    nat = NativeAutoencoderKL(**TINY_VAE_CONFIG)
    nat.load_state_dict(ora.state_dict())
    nat = nat.to(backend.device)
    with torch.no_grad():
        ora.post_quant_conv.bias.add_(0.5); nat.post_quant_conv.bias.add_(0.5)      # a bias large enough to show a border error
    g = torch.Generator().manual_seed(21)
    z = torch.randn(2, 4, *hw, generator=g)
    with torch.no_grad():
        ref = ora.decode(z)
    out = nat.decode(backend.to(z), return_dict=False)[0].cpu()
    assert out.shape == ref.shape == (2, 3, hw[0] * 2, hw[1] * 2) and out.dtype == torch.float32
    assert ((out - ref).norm() / ref.norm()).item() < 2e-2
    border = torch.cat([(out - ref)[..., :2, :].flatten(), (out - ref)[..., -2:, :].flatten(), (out - ref)[..., :, :2].flatten()])
    rb = torch.cat([ref[..., :2, :].flatten(), ref[..., -2:, :].flatten(), ref[..., :, :2].flatten()])
    assert (border.norm() / rb.norm()).item() < 2e-2
    assert nat.decode(backend.to(z)).sample.shape == ref.shape
    nat.enable_slicing()
    assert torch.equal(nat.decode(backend.to(z)).sample.cpu(), out)
    nat.disable_slicing(); nat.enable_tiling(); nat.disable_tiling()
    with pytest.raises(ValueError):
        nat.decode(backend.to(z[:, :3]))


@pytest.mark.gpu
@pytest.mark.parametrize("side", [256, 512])
def test_sd_vae_full_size_decode_vs_golden(side):
    """Full SD VAE decoder (49.5 M parameters, seeded weights) vs the oracle images committed in tests/golden/vae_decode_oracle.pt
    (oracle/make_golden.py vae_dec): 16384 seeded pixel samples + the image norm (512 px), the whole image (256 px)."""
    from oracle.make_golden import boundary_sample
    g = torch.load(os.path.join(GOLD, "vae_decode_oracle.pt"))[side]
    nat = seeded_init_(NativeAutoencoderKL(**SD_VAE_CONFIG), g["seed"]).to("cuda")
    gen = torch.Generator().manual_seed(g["input_seed"])
    z = torch.randn(1, 4, side // 8, side // 8, generator=gen)
    img = nat.decode(z.cuda(), return_dict=False)[0].cpu()
    vals, norm = boundary_sample(f"vae_dec_{side}", img, n=16384)
    assert ((vals - g["samples"]).norm() / g["samples"].norm()).item() < 2e-2 and abs(norm - g["norm"]) / g["norm"] < 1e-2
    if g["image"] is not None:
        assert ((img - g["image"].float()).norm() / g["image"].float().norm()).item() < 2e-2
