"""CLIP text encoder (host of the reference's text-encoder LoRA): the oracle restatement is pinned against the
installed transformers CLIPTextModel and against the reference's structure dump; the native gfx950 path against the oracle —
forward, and LoRA gradients through the encoder."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from hcp_diffusion_amd.lora import make_lora
from hcp_diffusion_amd.text_encoder import NativeCLIPTextModel
from oracle.clip_ref import CLIP_L_CONFIG, TINY_CLIP_CONFIG, OracleCLIPTextModel
from oracle.lora_ref import wrap_lora
from oracle.unet_sd15 import seeded_init_

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TE_LORA = [r"re:.*self_attn$", r"re:.*mlp$"]                  # cfgs/train/examples/lora_conventional.yaml:14-19


def test_text_encoder_names_match_reference_struct_dump():
    """Every parameter name and shape of oracle and native == the reference's cfgs/te_struct.txt (123.06 M parameters)."""
    ref = json.load(open(os.path.join(GOLD, "te_struct.json")))
    with torch.device("meta"):
        a, b = OracleCLIPTextModel(**CLIP_L_CONFIG), NativeCLIPTextModel(**CLIP_L_CONFIG)
    for m in (a, b):
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref["shapes"]
    assert sum(v.numel() for v in a.state_dict().values()) == ref["n_params"] == 123060480


def test_oracle_matches_installed_transformers_clip():
    """The restated arithmetic (pre-LN blocks, causal mask, scale, quick_gelu) vs transformers' own CLIPTextModel on the same
    weights: last_hidden_state and every hidden state to fp32 rounding."""
    tr = pytest.importorskip("transformers")
    cfg = tr.CLIPTextConfig(vocab_size=100, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                            max_position_embeddings=77, hidden_act="quick_gelu", bos_token_id=98, eos_token_id=99, pad_token_id=0)
    theirs = tr.CLIPTextModel(cfg).eval()
    ours = seeded_init_(OracleCLIPTextModel(**TINY_CLIP_CONFIG), 5)
    sd = ours.state_dict()
    keys = list(theirs.state_dict())
    prefixed = keys[0].startswith("text_model.")
    missing, unexpected = theirs.load_state_dict({(k if prefixed else k[len("text_model."):]): v for k, v in sd.items()}, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 98, (3, 77), generator=g); ids[:, 0] = 98; ids[:, 40:] = 99
    with torch.no_grad():
        out = theirs(input_ids=ids, output_hidden_states=True)
        hs = ours.hidden_states(ids)
        assert len(out.hidden_states) == len(hs) == 3
        for a, b in zip(out.hidden_states, hs):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)
        assert torch.allclose(out.last_hidden_state, ours.encode(ids), rtol=1e-4, atol=1e-5)
        assert torch.allclose(ours.text_model.final_layer_norm(out.hidden_states[-2]), ours.encode(ids, clip_skip=1), rtol=1e-4, atol=1e-5)
        mask = torch.ones(3, 77, dtype=torch.long); mask[0, 41:] = 0; mask[2, 60:] = 0          # padding masked out (tokenizer attention_mask)
        masked = theirs(input_ids=ids, attention_mask=mask).last_hidden_state
        assert torch.allclose(masked, ours.encode(ids, attention_mask=mask), rtol=1e-4, atol=1e-5)
        assert not torch.allclose(masked[0], out.last_hidden_state[0], atol=1e-3)


def _pair(dev, seed=5, **kw):
    ora = seeded_init_(OracleCLIPTextModel(**TINY_CLIP_CONFIG), seed)
    nat = NativeCLIPTextModel(**TINY_CLIP_CONFIG, **kw)
    nat.load_state_dict(ora.state_dict())
    return ora, nat.to(dev)


@pytest.mark.parametrize("clip_skip", [0, 1])
def test_tiny_text_encoder_forward_vs_oracle(backend, clip_skip):
    ora, nat = _pair(backend.device, clip_skip=clip_skip)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 100, (2, 77), generator=g)
    with torch.no_grad():
        ref = ora.encode(ids, clip_skip=clip_skip)
        out = nat(backend.to(ids)).float().cpu()
    assert out.shape == ref.shape
    assert ((out - ref).norm() / ref.norm()).item() < 2e-2
    mask = torch.ones(2, 77); mask[0, 50:] = 0
    with torch.no_grad():
        refm = ora.encode(ids, clip_skip=clip_skip, attention_mask=mask)
        outm = nat(backend.to(ids), attention_mask=backend.to(mask)).float().cpu()
    assert ((outm - refm).norm() / refm.norm()).item() < 2e-2 and ((refm - ref).norm() / ref.norm()).item() > 1e-2
    with pytest.raises(ValueError):
        nat(backend.to(ids), attention_mask=torch.ones(2, 70))


def test_text_encoder_prompt_repeats(backend):
    """tokenizer_repeats = 2 (TEEXHook, textencoder_ex.py:57-72): [B, 2 x 77] ids -> [B, 2 x 75 + 2, C] with one BOS and one EOS;
    the native model against the oracle restatement, and the stitching against its definition."""
    ora, _ = _pair(backend.device)
    nat = NativeCLIPTextModel(**TINY_CLIP_CONFIG, N_repeats=2)
    nat.load_state_dict(ora.state_dict()); nat.to(backend.device)
    g = torch.Generator().manual_seed(8)
    ids = torch.randint(0, 100, (2, 154), generator=g)
    with torch.no_grad():
        ref = ora.encode(ids, n_repeats=2)
        parts = ora.encode(ids.reshape(4, 77)).reshape(2, 2, 77, -1)
        out = nat(backend.to(ids)).float().cpu()
    assert ref.shape == out.shape == (2, 152, 128)
    assert torch.equal(ref[:, 0], parts[:, 0, 0]) and torch.equal(ref[:, -1], parts[:, 1, -1]) and torch.equal(ref[:, 76], parts[:, 1, 1])
    assert ((out - ref).norm() / ref.norm()).item() < 2e-2
    with pytest.raises(ValueError):
        nat(backend.to(ids[:, :153]))


def test_tiny_text_encoder_lora_gradients_vs_oracle(backend):
    """lora_text_encoder (rank 4 on self_attn + mlp Linears): gradients of a scalar loss on the conditioning states w.r.t. every
    W_down / W_up vs autograd through the oracle + the reference's LoRA restatement (cosine >= 0.995)."""
    dev = backend.device
    ora, nat = _pair(dev)
    ora.requires_grad_(False); nat.requires_grad_(False)
    wr = wrap_lora(ora, TE_LORA, rank=4)
    groups, group, bucket = make_lora(nat, [dict(layers=TE_LORA, rank=4)])
    assert sorted(k for k in ora.state_dict() if "lora" in k) == sorted(k for k in nat.state_dict() if "lora" in k)
    assert len(wr) == 2 * 6                                              # q,k,v,out + fc1,fc2 per layer
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for path, w in wr.items():
            blk = group.plugin_dict[path]
            w.lora_block_0.layer.W_up.copy_(torch.randn(w.lora_block_0.layer.W_up.shape, generator=gen) * 0.05)
            blk.layer.W_down.copy_(w.lora_block_0.layer.W_down); blk.layer.W_up.copy_(w.lora_block_0.layer.W_up)
    bucket.pack()
    ids = torch.randint(0, 100, (2, 77), generator=gen)
    target = torch.randn(2, 77, 128, generator=gen)
    lo = F.mse_loss(ora.encode(ids), target)
    lo.backward()
    out = nat(backend.to(ids))
    ln = F.mse_loss(out.float(), backend.to(target))
    ln.backward()
    assert abs(lo.item() - ln.item()) / lo.item() < 2e-2
    go = torch.cat([p.grad.flatten() for w in wr.values() for p in (w.lora_block_0.layer.W_down, w.lora_block_0.layer.W_up)])
    order = [group.plugin_dict[path] for path in wr]
    gn = torch.cat([p.grad.flatten().float().cpu() for blk in order for p in (blk.layer.W_down, blk.layer.W_up)])
    assert F.cosine_similarity(go, gn, dim=0).item() > 0.995
    assert (gn.norm() / go.norm()).item() == pytest.approx(1.0, abs=3e-2)


@pytest.mark.parametrize("with_mask", [False, True])
def test_unet_plus_text_encoder_lora_step_vs_oracle(backend, with_mask):
    """The reference's default LoRA example (lora_unet + lora_text_encoder, lora_conventional.yaml:7-19): the prompt is encoded
    inside the step, the UNet's cross-attention K/V projections hand a gradient back to the encoder's LoRA blocks, and ONE
    global-norm clip covers both buckets (train_ac.py:485-490).  Loss, both gradient sets and the updated parameters vs the
    oracle pair (UNet + CLIP restatements, reference LoRA restatement, torch clip_grad_norm_ + AdamW)."""
    from hcp_diffusion_amd import kernels as K
    from hcp_diffusion_amd.trainer import NativeTrainer
    from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
    from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel, add_noise, ddpm_alphas_cumprod
    dev = backend.device
    ucfg = dict(MICRO_CONFIG, cross_attention_dim=64)
    tcfg = dict(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, max_position_embeddings=77)
    ou = seeded_init_(OracleUNet2DConditionModel(**ucfg), 1); ot = seeded_init_(OracleCLIPTextModel(**tcfg), 2)
    nu = NativeUNet2DConditionModel(**ucfg); nu.load_state_dict(ou.state_dict()); nu.to(dev)
    nt = NativeCLIPTextModel(**tcfg); nt.load_state_dict(ot.state_dict()); nt.to(dev)
    ou.requires_grad_(False); ot.requires_grad_(False)
    UNET_LORA = [r"re:.*\.attn.?$", r"re:.*\.ff$"]
    wu, wt = wrap_lora(ou, UNET_LORA, rank=4), wrap_lora(ot, TE_LORA, rank=4)
    tr = NativeTrainer(nu, [dict(layers=UNET_LORA, rank=4)], lr=1e-3, text_encoder=nt, lora_te_cfg=[dict(layers=TE_LORA, rank=4)])
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for wr, group in ((wu, tr.lora_group), (wt, tr.lora_te_group)):
            for path, w in wr.items():
                blk = group.plugin_dict[path]
                w.lora_block_0.layer.W_up.copy_(torch.randn(w.lora_block_0.layer.W_up.shape, generator=gen) * 0.05)
                blk.layer.W_down.copy_(w.lora_block_0.layer.W_down); blk.layer.W_up.copy_(w.lora_block_0.layer.W_up)
    tr.bucket.pack(); tr.te_bucket.pack()
    x0 = torch.randn(2, 4, 8, 8, generator=gen); noise = torch.randn(2, 4, 8, 8, generator=gen)
    t = torch.tensor([100, 800]); ids = torch.randint(0, 100, (2, 77), generator=gen)
    amask = None
    if with_mask:                                    # the batch's attn_mask goes to BOTH models (wrapper.py:20,29)
        amask = torch.ones(2, 77); amask[0, 30:] = 0; amask[1, 70:] = 0
    pred = ou(add_noise(x0, noise, t, ddpm_alphas_cumprod()), t, ot.encode(ids, attention_mask=amask), encoder_attention_mask=amask).sample
    lo = F.mse_loss(pred, noise)
    lo.backward()
    tr.make_noise = lambda lat: (K.add_noise(lat, noise.to(dev), t.to(dev), tr.acp), noise.to(dev), t.to(dev))
    ln = tr.forward_backward(x0.to(dev), None, prompt_ids=ids.to(dev), attn_mask=amask.to(dev) if with_mask else None)
    assert abs(lo.item() - ln.item()) / lo.item() < 2e-2

    def flat(wr, group, grads):
        po = [p for w in wr.values() for p in (w.lora_block_0.layer.W_down, w.lora_block_0.layer.W_up)]
        pn = [p for path in wr for p in (group.plugin_dict[path].layer.W_down, group.plugin_dict[path].layer.W_up)]
        pick = (lambda p: p.grad) if grads else (lambda p: p.detach())
        return po, torch.cat([pick(p).flatten() for p in po]), torch.cat([pick(p).flatten().float().cpu() for p in pn])
    pu, gu_o, gu_n = flat(wu, tr.lora_group, True)
    pt, gt_o, gt_n = flat(wt, tr.lora_te_group, True)
    assert F.cosine_similarity(gu_o, gu_n, dim=0).item() > 0.995
    assert F.cosine_similarity(gt_o, gt_n, dim=0).item() > 0.99 and gt_o.norm().item() > 0       # gradient reached the encoder
    assert (gt_n.norm() / gt_o.norm()).item() == pytest.approx(1.0, abs=5e-2)
    # one clip over BOTH parameter sets, then AdamW: feed the oracle the native gradients so only the optimizer arithmetic is compared
    with torch.no_grad():
        off = 0
        for p in pu:
            p.grad = gu_n[off:off + p.numel()].view_as(p).clone(); off += p.numel()
        off = 0
        for p in pt:
            p.grad = gt_n[off:off + p.numel()].view_as(p).clone(); off += p.numel()
    opt = torch.optim.AdamW(pu + pt, lr=1e-3, weight_decay=1e-3)
    torch.nn.utils.clip_grad_norm_(pu + pt, 1.0)
    opt.step()
    tr.all_reduce(); tr.optimizer_step()
    for wr, group in ((wu, tr.lora_group), (wt, tr.lora_te_group)):
        _, po, pn = flat(wr, group, False)
        assert ((po - pn).abs().max() / po.abs().max()).item() < 1e-5


@pytest.mark.gpu
def test_clip_l_full_size_forward_and_lora_grads_vs_oracle():
    """Full CLIP-L text encoder (123 M parameters, seeded weights, 12 layers x 12 heads x 64, 77 tokens): conditioning states and
    rank-4 LoRA gradients vs the fp32 oracle computed here on the CPU."""
    ora = seeded_init_(OracleCLIPTextModel(**CLIP_L_CONFIG), 4)
    nat = NativeCLIPTextModel(**CLIP_L_CONFIG)
    nat.load_state_dict(ora.state_dict())
    nat.to("cuda")
    ora.requires_grad_(False); nat.requires_grad_(False)
    wr = wrap_lora(ora, TE_LORA, rank=4)
    _, group, bucket = make_lora(nat, [dict(layers=TE_LORA, rank=4)])
    gen = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for path, w in wr.items():
            blk = group.plugin_dict[path]
            w.lora_block_0.layer.W_up.copy_(torch.randn(w.lora_block_0.layer.W_up.shape, generator=gen) * 0.02)
            blk.layer.W_down.copy_(w.lora_block_0.layer.W_down); blk.layer.W_up.copy_(w.lora_block_0.layer.W_up)
    bucket.pack()
    ids = torch.randint(0, 49408, (4, 77), generator=gen); ids[:, 0] = 49406; ids[:, 30:] = 49407
    target = torch.randn(4, 77, 768, generator=gen)
    ref = ora.encode(ids)
    F.mse_loss(ref, target).backward()
    out = nat(ids.cuda())
    assert ((out.float().cpu() - ref.detach()).norm() / ref.norm()).item() < 2e-2
    F.mse_loss(out.float(), target.cuda()).backward()
    go = torch.cat([p.grad.flatten() for w in wr.values() for p in (w.lora_block_0.layer.W_down, w.lora_block_0.layer.W_up)])
    gn = torch.cat([p.grad.flatten().float().cpu() for path in wr for p in (group.plugin_dict[path].layer.W_down, group.plugin_dict[path].layer.W_up)])
    assert F.cosine_similarity(go, gn, dim=0).item() > 0.99


@pytest.mark.skipif(not os.path.isdir("/root/reference/hcpdiff"), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("n_repeats,clip_skip,final_norm", [(1, 0, True), (2, 1, True), (3, 0, False), (2, 2, True)])
def test_output_selection_matches_the_reference_teexhook(n_repeats, clip_skip, final_norm):
    """oracle.encode(clip_skip, final_norm, n_repeats) == the reference's OWN TEEXHook (hcpdiff/models/textencoder_ex.py:19-79: input
    re-chunking pre-hook, hidden_states[-clip_skip-1] + final_layer_norm, BOS / EOS stitching) hooked onto the oracle text model
    exposed with the transformers output fields the hook reads."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_hcp_ref_teex", "/root/reference/hcpdiff/models/textencoder_ex.py")
    teex = importlib.util.module_from_spec(spec); spec.loader.exec_module(teex)
    ora = seeded_init_(OracleCLIPTextModel(**TINY_CLIP_CONFIG), 5)

    class _Out(dict):
        pooler_output = None

    class HFLike(torch.nn.Module):                      # CLIPTextModel's surface as the hook uses it: .text_model.final_layer_norm, output fields
        def __init__(self, m):
            super().__init__()
            self.m, self.text_model = m, m.text_model

        def forward(self, input_ids, **kw):
            hs = self.m.hidden_states(input_ids)
            return _Out(hidden_states=hs, last_hidden_state=self.text_model.final_layer_norm(hs[-1]))

    host = HFLike(ora).eval()
    teex.TEEXHook(host, tokenizer=None, N_repeats=n_repeats, clip_skip=clip_skip, clip_final_norm=final_norm, device="cpu")
    ids = torch.randint(0, 100, (2, 77 * n_repeats), generator=torch.Generator().manual_seed(n_repeats))
    with torch.no_grad():
        ref, pooled = host(ids)
        ours = ora.encode(ids, clip_skip=clip_skip, final_norm=final_norm, n_repeats=n_repeats)
    assert pooled is None and ref.shape == ours.shape == (2, 75 * n_repeats + 2, 128)
    assert torch.equal(ref, ours)


@pytest.mark.skipif(not os.path.isdir("/root/reference/hcpdiff"), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("with_mask", [False, True])
def test_step_composition_matches_the_reference_wrapper(with_mask):
    """The reference's OWN TEUnetWrapper.forward (hcpdiff/models/wrapper.py:14-30, with pad_attn_bias from utils/utils.py:154-162 and
    TEEXHook on the text model) run over the oracle text encoder + oracle UNet == the composition NativeTrainer.forward_backward
    restates: states = TE(ids, attention_mask=m); pred = unet(x, t, states, encoder_attention_mask=m).  (With a mask the wrapper pads
    77 -> 80 keys with mask 0: the padded keys carry no weight, so the unpadded call must agree.)"""
    import importlib
    import importlib.util
    from oracle.ref_shims import load_reference_lora
    from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel
    load_reference_lora()                                    # registers the hcpdiff stub packages (hcpdiff.utils.pad_attn_bias)
    wrapper = importlib.import_module("hcpdiff.models.wrapper")
    spec = importlib.util.spec_from_file_location("_hcp_ref_teex2", "/root/reference/hcpdiff/models/textencoder_ex.py")
    teex = importlib.util.module_from_spec(spec); spec.loader.exec_module(teex)
    tcfg = dict(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, max_position_embeddings=77)
    ot = seeded_init_(OracleCLIPTextModel(**tcfg), 2)
    ou = seeded_init_(OracleUNet2DConditionModel(**dict(MICRO_CONFIG, cross_attention_dim=64)), 1)

    class _Out(dict):
        pooler_output = None

    class HFLike(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m, self.text_model = m, m.text_model

        def forward(self, input_ids, position_ids=None, attention_mask=None, output_hidden_states=True):
            hs = self.m.hidden_states(input_ids, position_ids, attention_mask)
            return _Out(hidden_states=hs, last_hidden_state=self.text_model.final_layer_norm(hs[-1]))

    te = HFLike(ot).eval()
    teex.TEEXHook(te, tokenizer=None, N_repeats=1, clip_skip=0, device="cpu")
    w = wrapper.TEUnetWrapper(ou, te)
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 100, (2, 77), generator=g); x = torch.randn(2, 4, 8, 8, generator=g); t = torch.tensor([10, 600])
    m = None
    if with_mask:
        m = torch.ones(2, 77); m[0, 33:] = 0; m[1, 70:] = 0
    with torch.no_grad():
        ref = w(ids, x, t, attn_mask=m)
        ours = ou(x, t, ot.encode(ids, attention_mask=m), encoder_attention_mask=m).sample
    assert torch.allclose(ref, ours, rtol=1e-5, atol=1e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/hcpdiff"), reason="reference tree only exists in the build container")
def test_sdxl_call_contract_matches_the_reference_wrapper():
    """SDXLTEUnetWrapper.forward (models/wrapper.py:57-74): added_cond_kwargs = {text_embeds: pooled_output[-1], time_ids: crop_info} —
    the contract NativeTrainer / bench.py feed the SDXL UNet with.  The reference wrapper over the oracle SDXL-structured UNet and a
    stand-in text encoder == the direct oracle call."""
    import importlib
    from oracle.ref_shims import load_reference_lora
    from oracle.unet_sd15 import TINY_SDXL_CONFIG, OracleUNet2DConditionModel
    load_reference_lora()
    wrapper = importlib.import_module("hcpdiff.models.wrapper")
    ou = seeded_init_(OracleUNet2DConditionModel(**TINY_SDXL_CONFIG), 1)
    g = torch.Generator().manual_seed(5)
    ctx_dim = TINY_SDXL_CONFIG["cross_attention_dim"]
    pooled_dim = TINY_SDXL_CONFIG["projection_class_embeddings_input_dim"] - 6 * TINY_SDXL_CONFIG["addition_time_embed_dim"]
    ehs = torch.randn(2, 77, ctx_dim, generator=g); pooled = torch.randn(2, pooled_dim, generator=g)

    class TE(torch.nn.Module):
        def forward(self, prompt_ids, position_ids=None, attention_mask=None, output_hidden_states=True):
            return ehs, [torch.zeros_like(pooled), pooled]               # (states, per-encoder pooled outputs): the wrapper takes the last

    x = torch.randn(2, 4, 8, 8, generator=g); t = torch.tensor([5, 500])
    crop = torch.tensor([[64.0, 64.0, 0.0, 0.0, 64.0, 64.0]] * 2)
    with torch.no_grad():
        ref = wrapper.SDXLTEUnetWrapper(ou, TE())(torch.zeros(2, 77, dtype=torch.long), x, t, crop_info=crop)
        ours = ou(x, t, ehs, added_cond_kwargs=dict(text_embeds=pooled, time_ids=crop)).sample
    assert torch.equal(ref, ours)


@pytest.mark.skipif(not os.path.isdir("/root/reference/hcpdiff"), reason="reference tree only exists in the build container")
def test_native_models_inside_the_reference_wrapper(backend):
    """Drop-in check of seam 1 + the text encoder: the reference's OWN TEUnetWrapper (models/wrapper.py:6-30) constructed over the
    NATIVE text encoder and NATIVE UNet, called the way Trainer.forward calls it (train_ac.py:454), gives the same prediction as the
    native trainer's own composition — and tracks the oracle pair."""
    import importlib
    from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
    from oracle.ref_shims import load_reference_lora
    from oracle.unet_sd15 import MICRO_CONFIG, OracleUNet2DConditionModel
    load_reference_lora()
    wrapper = importlib.import_module("hcpdiff.models.wrapper")
    dev = backend.device
    tcfg = dict(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, max_position_embeddings=77)
    ucfg = dict(MICRO_CONFIG, cross_attention_dim=64)
    ot = seeded_init_(OracleCLIPTextModel(**tcfg), 2); ou = seeded_init_(OracleUNet2DConditionModel(**ucfg), 1)
    nt = NativeCLIPTextModel(**tcfg); nt.load_state_dict(ot.state_dict()); nt.to(dev)
    nu = NativeUNet2DConditionModel(**ucfg); nu.load_state_dict(ou.state_dict()); nu.to(dev)
    w = wrapper.TEUnetWrapper(nu, nt)
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 100, (2, 77), generator=g); x = torch.randn(2, 4, 8, 8, generator=g); t = torch.tensor([10, 600])
    m = torch.ones(2, 77); m[0, 33:] = 0
    to = backend.to
    with torch.no_grad():
        for mask in (None, m):
            ref = w(to(ids), to(x), to(t), attn_mask=to(mask) if mask is not None else None)
            ours = nu(to(x), to(t), nt(to(ids), attention_mask=to(mask) if mask is not None else None),
                      encoder_attention_mask=to(mask) if mask is not None else None).sample
            assert torch.allclose(ref.float().cpu(), ours.float().cpu(), rtol=1e-3, atol=1e-3)       # (the wrapper pads 77 -> 80 masked keys)
            oracle = ou(x, t, ot.encode(ids, attention_mask=mask), encoder_attention_mask=mask).sample
            assert ((ref.float().cpu() - oracle).norm() / oracle.norm()).item() < 2e-2
