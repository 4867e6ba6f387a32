"""ORACLE / test infrastructure — the recipe that PINS the two un-pinned restatements the day `diffusers` is importable.

oracle/unet_sd15.py (UNet2DConditionModel) and oracle/vae_ref.py (AutoencoderKL encoder) restate the arithmetic of the reference's
un-vendored dependency ``diffusers<=0.26.1`` (requirements.txt:4); that wheel cannot be installed in the build container (no
network), so today their parity is "unpinned" (oracle/README.md).  Run this script in an environment that has

    pip install "diffusers==0.26.1" "torch" "transformers" "safetensors"

and it will, for every configuration the goldens use (tiny / micro / full SD1.5, tiny / full SDXL, the VAE encoder):
  1. build the REAL diffusers module with that configuration, load the oracle's seeded weights into it by parameter name
     (names are identical by construction — tests/golden/sd15_struct.json pins them to the reference's cfgs/unet_struct.txt);
  2. run both on the golden inputs and FAIL LOUDLY (exit code 1, per-tensor report) if any output differs by more than
     fp32 accumulation noise (rtol 1e-4 / atol 1e-5 on `.sample`, as SURVEY.md §8(c) states);
  3. on success regenerate tests/golden/*.pt from the diffusers outputs (``--write``), so that every native-vs-golden test is from
     then on a test against the reference's real dependency, and write oracle/_ref/PINNED with the diffusers version.
Exit code 3 = diffusers not importable (nothing checked)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _diffusers():
    try:
        import diffusers
    except ImportError:
        print("diffusers is not importable here: the UNet / VAE oracles stay UNPINNED (see the module docstring for the recipe)")
        sys.exit(3)
    ver = tuple(int(x) for x in diffusers.__version__.split(".")[:3])
    if ver > (0, 26, 1):
        print(f"warning: diffusers {diffusers.__version__} is newer than the reference's pin (<=0.26.1): attention processors / defaults may differ")
    return diffusers


def _unet_kwargs(cfg):
    """Oracle config -> diffusers.UNet2DConditionModel kwargs (same names where they exist)."""
    kw = dict(sample_size=64, in_channels=cfg["in_channels"], out_channels=cfg["out_channels"], layers_per_block=cfg["layers_per_block"],
              block_out_channels=tuple(cfg["block_out_channels"]), down_block_types=tuple(cfg["down_block_types"]),
              up_block_types=tuple(cfg["up_block_types"]), cross_attention_dim=cfg["cross_attention_dim"],
              norm_num_groups=cfg["norm_num_groups"], attention_head_dim=cfg["num_attention_heads"],
              transformer_layers_per_block=cfg.get("transformer_layers_per_block", 1),
              use_linear_projection=bool(cfg.get("use_linear_projection", False)))
    if cfg.get("addition_embed_type"):
        kw.update(addition_embed_type=cfg["addition_embed_type"], addition_time_embed_dim=cfg["addition_time_embed_dim"],
                  projection_class_embeddings_input_dim=cfg["projection_class_embeddings_input_dim"])
    return kw


def check_unet(diffusers, name, cfg, inputs, rtol=1e-4, atol=1e-5):
    from oracle.unet_sd15 import OracleUNet2DConditionModel, seeded_init_
    ora = seeded_init_(OracleUNet2DConditionModel(**cfg), 1).eval()
    ref = diffusers.UNet2DConditionModel(**_unet_kwargs(cfg)).eval()
    missing, unexpected = ref.load_state_dict(ora.state_dict(), strict=False)
    if missing or unexpected:
        print(f"[{name}] parameter names differ: missing {missing[:5]} unexpected {unexpected[:5]}")
        return False
    with torch.no_grad():
        a = ora(*inputs["args"], **inputs.get("kwargs", {})).sample
        b = ref(*inputs["args"], **inputs.get("kwargs", {})).sample
    err = (a - b).abs().max().item()
    ok = torch.allclose(a, b, rtol=rtol, atol=atol)
    print(f"[{name}] oracle vs diffusers {diffusers.__version__}: max |diff| {err:.3e} (|sample| max {b.abs().max().item():.3e}) -> {'OK' if ok else 'MISMATCH'}")
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="after a clean comparison regenerate tests/golden/*.pt (oracle/make_golden.py)")
    ap.add_argument("--full", action="store_true", help="also the full-size SD1.5 / SDXL configurations (minutes of CPU time)")
    args = ap.parse_args()
    diffusers = _diffusers()
    from oracle.make_golden import sd15_b4_inputs
    from oracle.unet_sd15 import MICRO_CONFIG, SD15_CONFIG, SDXL_CONFIG, TINY_CONFIG, TINY_SDXL_CONFIG, add_noise, ddpm_alphas_cumprod
    g = torch.Generator().manual_seed(0)
    ok = True
    for name, cfg, hw, L in [("micro", MICRO_CONFIG, 8, 24), ("tiny", TINY_CONFIG, 16, 77)]:
        x = torch.randn(2, 4, hw, hw, generator=g); t = torch.tensor([10, 900]); e = torch.randn(2, L, cfg["cross_attention_dim"], generator=g)
        ok &= check_unet(diffusers, name, cfg, dict(args=(x, t, e)))
        mask = torch.ones(2, L); mask[:, L - 5:] = 0
        ok &= check_unet(diffusers, name + "+encoder_attention_mask", cfg, dict(args=(x, t, e), kwargs=dict(encoder_attention_mask=mask)))
    x = torch.randn(2, 4, 16, 16, generator=g); t = torch.tensor([3, 700]); e = torch.randn(2, 77, TINY_SDXL_CONFIG["cross_attention_dim"], generator=g)
    added = dict(text_embeds=torch.randn(2, TINY_SDXL_CONFIG["projection_class_embeddings_input_dim"] - 6 * TINY_SDXL_CONFIG["addition_time_embed_dim"], generator=g),
                 time_ids=torch.tensor([[128.0, 128.0, 0.0, 0.0, 128.0, 128.0]] * 2))
    ok &= check_unet(diffusers, "tiny-sdxl", TINY_SDXL_CONFIG, dict(args=(x, t, e), kwargs=dict(added_cond_kwargs=added)))
    if args.full:
        x0, ehs, noise, t = sd15_b4_inputs()
        ok &= check_unet(diffusers, "sd15-full-b4", SD15_CONFIG, dict(args=(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ehs)))
    # scheduler: DDPMScheduler.add_noise on the SD beta schedule
    sch = diffusers.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    x0 = torch.randn(3, 4, 8, 8, generator=g); n = torch.randn(3, 4, 8, 8, generator=g); t = torch.tensor([0, 437, 999])
    d = (sch.add_noise(x0, n, t) - add_noise(x0, n, t, ddpm_alphas_cumprod())).abs().max().item()
    print(f"[add_noise] max |diff| {d:.3e}"); ok &= d < 1e-6
    # VAE encoder
    try:
        from oracle.vae_ref import OracleVAEEncoder, TINY_VAE_CONFIG
        from oracle.unet_sd15 import seeded_init_
        ora = seeded_init_(OracleVAEEncoder(**TINY_VAE_CONFIG), 3).eval()
        vae = diffusers.AutoencoderKL(in_channels=3, out_channels=3, latent_channels=TINY_VAE_CONFIG["latent_channels"],
                                      block_out_channels=tuple(TINY_VAE_CONFIG["block_out_channels"]), layers_per_block=TINY_VAE_CONFIG["layers_per_block"],
                                      down_block_types=("DownEncoderBlock2D",) * len(TINY_VAE_CONFIG["block_out_channels"]),
                                      up_block_types=("UpDecoderBlock2D",) * len(TINY_VAE_CONFIG["block_out_channels"]),
                                      norm_num_groups=TINY_VAE_CONFIG["norm_num_groups"]).eval()
        sd = {k: v for k, v in ora.state_dict().items()}
        miss, unexp = vae.load_state_dict(sd, strict=False)
        img = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
        with torch.no_grad():
            a = ora.moments(img); b = vae.encode(img).latent_dist.parameters
        d = (a - b).abs().max().item()
        print(f"[vae encoder] max |diff| {d:.3e} (unexpected keys {len(unexp)})"); ok &= d < 1e-4
    except Exception as e:  # noqa: BLE001 - report, count as failure
        print(f"[vae encoder] comparison could not run: {type(e).__name__}: {e}"); ok = False
    if not ok:
        print("PINNING FAILED: a restatement disagrees with diffusers — fix oracle/*.py before trusting any golden")
        sys.exit(1)
    if getattr(diffusers, "__hcp_mock__", False):          # tests/test_pin_recipe.py exercises this script against a stand-in namespace
        print("mock diffusers namespace: recipe ran clean, nothing is written")
        return
    os.makedirs(os.path.join(ROOT, "oracle", "_ref"), exist_ok=True)
    open(os.path.join(ROOT, "oracle", "_ref", "PINNED"), "w").write(f"diffusers {diffusers.__version__}\n")
    print(f"oracle pinned against diffusers {diffusers.__version__}")
    if args.write:
        import subprocess
        subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py")])


if __name__ == "__main__":
    main()
