"""Checkpoint wire format: files written by the REFERENCE load into the native modules, files written
natively load into the reference's HCPModelLoader, and every section (base / lora / plugin, + _ema, both containers,
old LoRA key scheme) round-trips.  The fixture tests/golden/ref_lora_unet-7.safetensors was produced by the reference's
own make_hcpdiff + CkptManagerSafe (oracle/make_golden.py ckpt)."""
import os

import pytest
import torch

from hcp_diffusion_amd import kernels as K
from hcp_diffusion_amd.ckpt import CkptManagerNative, NativeModelLoader, fold_dict, unfold_dict
from hcp_diffusion_amd.controlnet import make_controlnet
from hcp_diffusion_amd.trainer import NativeTrainer
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel, seeded_init_

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF_CKPT = os.path.join(GOLD, "ref_lora_unet-7.safetensors")
HAVE_REFERENCE = os.path.isdir("/root/reference/hcpdiff")


def _native(dev, seed=1):
    nat = seeded_init_(NativeUNet2DConditionModel(**MICRO_CONFIG), seed)
    return nat.to(dev)


def test_fold_unfold_are_inverse_and_flatten_lists():
    t = [torch.full((2,), float(i)) for i in range(4)]
    nested = {"lora": {"a.___.layer.W_up": t[0], "a.___.alpha": t[1]}, "base": {"conv_in.weight": t[2]}, "seq": [t[3]]}
    flat = unfold_dict(nested)
    assert set(flat) == {"lora:a.___.layer.W_up", "lora:a.___.alpha", "base:conv_in.weight", "seq:0"}
    back = fold_dict(flat)
    assert back["lora"]["a.___.layer.W_up"] is t[0] and back["seq"]["0"] is t[3]


def test_native_loads_reference_written_lora_ckpt(backend):
    """LoRA trained/saved by the reference (Linear r4 on attn/ff + Conv r8 on resnets.0.conv1) -> native blocks; the native
    forward reproduces the prediction the reference's LoraPatchContainer forward gave (bf16 vs fp32: rel. L2 <= 2e-2)."""
    exp = torch.load(os.path.join(GOLD, "ref_lora_ckpt_expect.pt"))
    nat = _native(backend.device, exp["host_seed"])
    nat.requires_grad_(False)
    group, bucket = NativeModelLoader(nat).load_lora([dict(path=REF_CKPT, alpha=2.0)])
    # load_lora keys its group '{layer}.{block name}' (cfg_net_tools.py:290), make_hcpdiff '{layer}' (:117): mirrored as is
    assert sorted(k.replace(".lora_block_0.___.", ".___.") for k in group.state_dict()) == exp["keys"]
    raw = CkptManagerNative.load_ckpt(REF_CKPT)["lora"]
    for k, v in group.state_dict().items():
        if not k.endswith("alpha"):
            assert torch.equal(v.detach().cpu(), raw[k.replace(".lora_block_0.___.", ".___.")]), k
    some = group.plugin_dict["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.lora_block_0"]
    assert abs(float(some.alpha) - 2.0 / 4) < 1e-7 and some.rank == 4
    conv = group.plugin_dict["down_blocks.0.resnets.0.conv1.lora_block_0"]
    assert conv.rank == 8 and conv.host_type == "conv" and tuple(conv.layer.W_down.shape) == (8, 40, 3, 3)
    to = backend.to
    with torch.no_grad():
        y = nat(to(exp["x"]), to(exp["t"]), to(exp["ehs"])).sample.float().cpu()
    base_only = _native(backend.device, exp["host_seed"])
    with torch.no_grad():
        y0 = base_only(to(exp["x"]), to(exp["t"]), to(exp["ehs"])).sample.float().cpu()
    err = ((y - exp["pred"]).norm() / exp["pred"].norm()).item()
    assert err < 2e-2, err
    assert ((y0 - exp["pred"]).norm() / exp["pred"].norm()).item() > 3 * err       # the loaded LoRA matters


def _trained_native(backend, cfg, **kw):
    nat = _native(backend.device)
    tr = NativeTrainer(nat, cfg, lr=1e-3, **kw)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for blk in tr.lora_group.plugin_dict.values():
            blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=g) * 0.05)
    tr.bucket.pack()
    return nat, tr


LORA_CFG = [dict(layers=[r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=4, alpha=2.0), dict(layers=[r"re:.*\.resnets\.0\.conv1$"], rank=8, alpha=2.0)]


@pytest.mark.parametrize("fmt", ["safetensors", "ckpt"])
def test_native_lora_ckpt_round_trip(backend, tmp_path, fmt):
    nat, tr = _trained_native(backend, LORA_CFG, ema=dict(decay_max=0.9))
    mgr = CkptManagerNative(fmt=fmt)
    mgr.set_save_dir(str(tmp_path))
    (path,) = tr.save_model(mgr, step=3)
    assert path.endswith(f"unet-3.{fmt}")
    sd = mgr.load_ckpt(path)
    assert set(sd) == {"base", "lora", "base_ema", "lora_ema"} or set(sd) == {"lora", "lora_ema"}   # empty dicts vanish in safetensors
    assert set(sd["lora"]) == set(sd["lora_ema"]) == set(torch.load(os.path.join(GOLD, "ref_lora_ckpt_expect.pt"))["keys"])
    fresh = _native(backend.device)
    fresh.requires_grad_(False)
    group, bucket = NativeModelLoader(fresh).load_lora([dict(path=path, alpha=2.0)])
    saved = tr.lora_group.state_dict()
    for k, v in group.state_dict().items():                                      # (bucket order follows the file's key order)
        assert torch.equal(v.detach().cpu(), saved[k.replace(".lora_block_0.___.", ".___.")].detach().cpu()), k
    assert bucket.params.numel() == tr.bucket.params.numel()
    to = backend.to
    g = torch.Generator().manual_seed(8)
    x, ehs, t = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 77, 32, generator=g), torch.tensor([123])
    with torch.no_grad():
        assert torch.equal(nat(to(x), to(t), to(ehs)).sample, fresh(to(x), to(t), to(ehs)).sample)
    # selected layers only (cfg_net_tools.py:268-277)
    part = _native(backend.device)
    g2, _ = NativeModelLoader(part).load_lora([dict(path=path, alpha=2.0, layers=["re:down_blocks\\.0\\.attentions.*"])])
    assert g2.plugin_dict and all(k.startswith("down_blocks.0.attentions") for k in g2.plugin_dict)


def test_old_lora_key_scheme_is_converted(backend, tmp_path):
    """'<host>.lora_block.layer.lora_down.weight' files (tools/convert_old_lora.py; cfg_net_tools.py:262-264,283-284)."""
    raw = CkptManagerNative.load_ckpt(REF_CKPT)["lora"]
    host = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q"
    old = {f"{host}.lora_block.layer.lora_down.weight": raw[f"{host}.___.layer.W_down"],
           f"{host}.lora_block.layer.lora_up.weight": raw[f"{host}.___.layer.W_up"]}
    mgr = CkptManagerNative()
    path = mgr._save_ckpt({"lora": old}, save_path=str(tmp_path / "old.safetensors"))
    with pytest.warns(DeprecationWarning):
        group, _ = NativeModelLoader(_native(backend.device)).load_lora([dict(path=path)])
    (blk,) = group.plugin_dict.values()
    assert torch.equal(blk.layer.W_up.detach().cpu(), raw[f"{host}.___.layer.W_up"])
    bad = mgr._save_ckpt({"lora": {f"{host}.___.layer.weird": raw[f"{host}.___.layer.W_up"]}}, save_path=str(tmp_path / "bad.safetensors"))
    with pytest.raises(ValueError):
        NativeModelLoader(_native(backend.device)).load_lora([dict(path=bad)])


def test_base_part_round_trip_and_merge(backend, tmp_path):
    """Full fine-tuning: 'base' holds the trainable masters by diffusers name (channels-last conv masters are written
    contiguous in their logical shape); load_part computes base_model_alpha * p + alpha * ckpt (cfg_net_tools.py:232-246)."""
    nat = _native(backend.device)
    tr = NativeTrainer(nat, None, train_cfg=[dict(layers=["re:down_blocks\\.0\\..*"], lr=1e-5)], ema=dict(decay_max=0.5))
    with torch.no_grad():
        for st in tr.host_buckets:
            st.bucket.params.add_(0.01)
    mgr = CkptManagerNative()
    mgr.set_save_dir(str(tmp_path))
    (path,) = tr.save_model(mgr, step=11)
    sd = mgr.load_ckpt(path)
    want = {k for k, p in nat.named_parameters() if p.requires_grad}
    assert set(sd["base"]) == set(sd["base_ema"]) == want and all(k.startswith("down_blocks.0.") for k in want)
    w = sd["base"]["down_blocks.0.resnets.0.conv1.weight"]
    assert w.is_contiguous() and torch.equal(w, nat.state_dict()["down_blocks.0.resnets.0.conv1.weight"].cpu())
    other = _native(backend.device, seed=2)
    before = {k: v.detach().cpu().clone() for k, v in other.named_parameters()}
    NativeModelLoader(other).load_part([dict(path=path, alpha=0.25)], base_model_alpha=0.5)
    for k, v in other.named_parameters():
        ref = 0.5 * before[k] + 0.25 * sd["base"][k] if k in want else before[k]
        assert torch.allclose(v.detach().cpu(), ref, atol=1e-7), k
    lay = dict(other.named_modules())["down_blocks.0.resnets.0.conv1"]
    pk = lay.packed()                                                           # operand copy follows the merged master
    assert torch.allclose(pk.w.float().cpu().view(-1)[:32], lay.weight.detach().permute(0, 2, 3, 1).reshape(-1)[:32].cpu().to(torch.bfloat16).float())


def test_controlnet_plugin_ckpt_round_trip(backend, tmp_path):
    nat = _native(backend.device)
    cn = make_controlnet(nat)
    tr = NativeTrainer(nat, None, plugins=[(cn, 1e-4)])
    with torch.no_grad():
        tr.host_buckets[0].bucket.params.add_(0.02)
    mgr = CkptManagerNative()
    mgr.set_save_dir(str(tmp_path))
    paths = tr.save_model(mgr, step=5)
    assert [os.path.basename(p) for p in paths] == ["unet-5.safetensors", "unet-controlnet1-5.safetensors"]
    sd = mgr.load_ckpt(paths[1])["plugin"]
    assert set(sd) == {f".___.{k}" for k in cn.state_dict()}
    assert "down_blocks.0.resnets.0.conv1.weight" in {k[len(".___."):] for k in sd}
    other = _native(backend.device)
    cn2 = make_controlnet(other)
    NativeModelLoader(other).load_plugin({"controlnet1": dict(path=paths[1])})
    for (k, a), (_, b) in zip(cn.state_dict().items(), cn2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k


@pytest.mark.skipif(not HAVE_REFERENCE, reason="reference tree only exists in the build container")
def test_reference_loader_reads_native_written_lora_ckpt(backend, tmp_path):
    """The other direction: a natively trained/saved LoRA file goes through the reference's OWN auto_manager +
    HCPModelLoader.load_lora (cfg_net_tools.py:248-292) onto the oracle UNet; both models then predict the same."""
    from oracle.make_golden import _Item
    from oracle.ref_shims import load_reference_ckpt
    _, tools = load_reference_ckpt()
    nat, tr = _trained_native(backend, LORA_CFG)
    mgr = CkptManagerNative()
    mgr.set_save_dir(str(tmp_path))
    (path,) = tr.save_model(mgr, step=1)
    ora = seeded_init_(OracleUNet2DConditionModel(**MICRO_CONFIG), 1)
    ora.requires_grad_(False)
    ora.device = torch.device("cpu")
    group = tools.HCPModelLoader(ora).load_lora([_Item(path=path, alpha=2.0)])
    assert sorted(k.replace(".lora_block_0.___.", ".___.") for k in group.state_dict()) == sorted(tr.lora_group.state_dict())
    g = torch.Generator().manual_seed(6)
    x, ehs, t = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 77, 32, generator=g), torch.tensor([40, 800])
    to = backend.to
    with torch.no_grad():
        yo = ora(x, t, ehs).sample
        yn = nat(to(x), to(t), to(ehs)).sample.float().cpu()
    assert ((yn - yo).norm() / yo.norm()).item() < 2e-2


@pytest.mark.skipif(not HAVE_REFERENCE, reason="reference tree only exists in the build container")
def test_ckpt_fixture_is_reproducible_from_reference(tmp_path):
    from oracle.make_golden import ref_lora_ckpt_fixture
    ref_lora_ckpt_fixture(str(tmp_path))
    new, old = CkptManagerNative.load_ckpt(str(tmp_path / "ref_lora_unet-7.safetensors")), CkptManagerNative.load_ckpt(REF_CKPT)
    assert set(new) == set(old) == {"lora"} and set(new["lora"]) == set(old["lora"])
    assert all(torch.equal(new["lora"][k], old["lora"][k]) for k in old["lora"])
    assert torch.equal(torch.load(tmp_path / "ref_lora_ckpt_expect.pt")["pred"], torch.load(os.path.join(GOLD, "ref_lora_ckpt_expect.pt"))["pred"])


def test_text_encoder_lora_ckpt_round_trip(backend, tmp_path):
    """Trainer.save_model writes the text encoder's LoRA to its own ``text_encoder-{step}`` file (train_ac.py:529-533); the loader
    rebuilds the blocks on a fresh encoder and the conditioning states are bit-identical."""
    from hcp_diffusion_amd.text_encoder import NativeCLIPTextModel
    from oracle.clip_ref import OracleCLIPTextModel
    tcfg = dict(vocab_size=100, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=1, max_position_embeddings=77)
    tcfg["hidden_size"] = 64                                            # one 64-wide head; MICRO UNet with a matching context width
    ucfg = dict(MICRO_CONFIG, cross_attention_dim=64)
    te_sd = seeded_init_(OracleCLIPTextModel(**tcfg), 2).state_dict()
    def fresh_te():
        m = NativeCLIPTextModel(**tcfg); m.load_state_dict(te_sd); return m.to(backend.device)
    nat = seeded_init_(NativeUNet2DConditionModel(**ucfg), 1).to(backend.device)
    te = fresh_te()
    te_cfg = [dict(layers=[r"re:.*self_attn$", r"re:.*mlp$"], rank=4, alpha=2.0)]
    tr = NativeTrainer(nat, [dict(layers=[r"re:.*\.attn.?$"], rank=4)], lr=1e-3, text_encoder=te, lora_te_cfg=te_cfg)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for blk in tr.te_bucket.blocks:
            blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=g) * 0.05)
    tr.te_bucket.pack()
    mgr = CkptManagerNative()
    mgr.set_save_dir(str(tmp_path))
    paths = tr.save_model(mgr, step=9)
    assert [os.path.basename(p) for p in paths] == ["unet-9.safetensors", "text_encoder-9.safetensors"]
    sd = mgr.load_ckpt(paths[1])["lora"]
    assert len(sd) == 2 * 6 * 3 and "text_model.encoder.layers.1.mlp.fc2.___.layer.W_up" in sd
    te2 = fresh_te()
    te2.requires_grad_(False)
    group, _ = NativeModelLoader(te2).load_lora([dict(path=paths[1], alpha=2.0)])
    ids = backend.to(torch.randint(0, 100, (2, 77), generator=g))
    with torch.no_grad():
        assert torch.equal(te(ids), te2(ids))


@pytest.mark.skipif(not HAVE_REFERENCE, reason="reference tree only exists in the build container")
@pytest.mark.parametrize("fmt", ["safetensors", "ckpt"])
def test_reference_auto_manager_reads_native_files(backend, tmp_path, fmt):
    """Both containers through the reference's OWN auto_manager (ckpt_manager/__init__.py): a natively written file yields the same
    nested dict (sections, keys, tensors) as CkptManagerNative.load_ckpt — and a file the reference's manager writes from that dict
    is read back natively."""
    from oracle.ref_shims import load_reference_ckpt
    load_reference_ckpt()
    import hcpdiff.ckpt_manager as ref_mgr
    nat, tr = _trained_native(backend, LORA_CFG, ema=dict(decay_max=0.9))
    mgr = CkptManagerNative(fmt=fmt)
    mgr.set_save_dir(str(tmp_path))
    (path,) = tr.save_model(mgr, step=2)
    theirs, ours = ref_mgr.auto_manager(path).load_ckpt(path), mgr.load_ckpt(path)
    assert set(theirs) == set(ours) >= {"lora", "lora_ema"}
    for sec in ours:
        assert set(theirs[sec]) == set(ours[sec])
        assert all(torch.equal(theirs[sec][k], ours[sec][k]) for k in ours[sec])
    back = str(tmp_path / f"back.{fmt}")
    ref_mgr.auto_manager(back)._save_ckpt({k: v for k, v in theirs.items() if v}, save_path=back)
    again = mgr.load_ckpt(back)
    assert all(torch.equal(again["lora"][k], ours["lora"][k]) for k in ours["lora"])


def test_webui_round_trip_loads_into_the_native_model(backend, tmp_path):
    """A natively trained LoRA file -> webui layout (lora_convert CLI) -> back -> NativeModelLoader.load_lora on a fresh UNet: the same
    prediction as the model that was trained (the whole f2 chain a user of the reference's tools/lora_convert.py walks)."""
    from hcp_diffusion_amd import lora_convert as LC
    nat, tr = _trained_native(backend, LORA_CFG)
    mgr = CkptManagerNative(fmt="safetensors")
    mgr.set_save_dir(str(tmp_path))
    (path,) = tr.save_model(mgr, step=5)
    LC.main(["--lora_path", path, "--dump_path", str(tmp_path / "webui.safetensors"), "--to_webui"])
    web = mgr.load_ckpt(str(tmp_path / "webui.safetensors"))
    assert web and all(k.startswith("lora_unet_") and k.rsplit(".", 2)[-2:] in (["lora_down", "weight"], ["lora_up", "weight"]) or k.endswith(".alpha")
                       for k in web)
    LC.main(["--lora_path", str(tmp_path / "webui.safetensors"), "--dump_path", str(tmp_path / "back"), "--from_webui"])
    fresh = _native(backend.device)
    fresh.requires_grad_(False)
    group, _ = NativeModelLoader(fresh).load_lora([dict(path=str(tmp_path / "back" / "unet-webui.safetensors"), alpha=2.0)])
    assert len(group.plugin_dict) == len(tr.lora_group.plugin_dict)
    to = backend.to
    g = torch.Generator().manual_seed(8)
    x, ehs, t = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 77, 32, generator=g), torch.tensor([123])
    with torch.no_grad():
        assert torch.equal(nat(to(x), to(t), to(ehs)).sample, fresh(to(x), to(t), to(ehs)).sample)
