"""Seams 2 and 3 under the REAL reference plugin API: in a fresh interpreter the reference's own modules are made
importable first (oracle/ref_shims.py), so hcp_diffusion_amd.patch_api binds to hcpdiff.models.plugin instead of its restatement;
then the reference's OWN builders — make_hcpdiff with ``type: lora_hip`` and make_plugin with the native ControlNet class — assemble
the native modules, and a forward runs through the interpreted kernels.  Only where /root/reference exists."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import functools, sys, torch
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from oracle.ref_shims import load_reference_ckpt
_, tools = load_reference_ckpt()                         # BEFORE the package: patch_api must find hcpdiff importable
import hcp_diffusion_amd.patch_api as pa
assert pa.USING_REFERENCE_PLUGIN_API
import hcpdiff.models.plugin as ref_plugin
import hcpdiff.models.lora_layers_patch as llp
from conftest import emu_cdll
from hcp_diffusion_amd import kernels as K
K._set_backend_for_tests(emu_cdll())
from hcp_diffusion_amd.controlnet import ControlNetHipPlugin
from hcp_diffusion_amd.lora import LoraBucket, LoraHipLayer
from hcp_diffusion_amd.unet import NativeUNet2DConditionModel
from oracle.make_golden import _Item
from oracle.unet_sd15 import MICRO_CONFIG, seeded_init_

assert issubclass(LoraHipLayer, ref_plugin.PatchPluginBlock) and llp.lora_layer_map["lora_hip"] is LoraHipLayer
assert issubclass(ControlNetHipPlugin, ref_plugin.MultiPluginBlock)
u = seeded_init_(NativeUNet2DConditionModel(**MICRO_CONFIG), 1); u.requires_grad_(False)
# --- seam 2: the reference's make_hcpdiff builds the native LoRA blocks (cfg_net_tools.py:107-123)
groups, group = tools.make_hcpdiff(u, None, [_Item(layers=[r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=4, type="lora_hip", lr=1e-4)])
assert type(group).__name__ == "LoraGroup" and len(group.plugin_dict) == 40 and len(groups[0]["params"]) == 80
assert type(u.down_blocks[0].attentions[0].transformer_blocks[0].attn1.to_q).__name__ == "LoraHipContainer"
assert all(k.count(".___.") == 1 for k in group.state_dict())
g = torch.Generator().manual_seed(3)
with torch.no_grad():
    for blk in group.plugin_dict.values():
        blk.layer.W_up.copy_(torch.randn(blk.layer.W_up.shape, generator=g) * 0.05)
LoraBucket(list(group.plugin_dict.values()))
x, t, ehs = torch.randn(1, 4, 8, 8, generator=g), torch.tensor([77]), torch.randn(1, 9, 32, generator=g)
y = u(x, t, ehs).sample
y.square().mean().backward()
grads = [p.grad for p in groups[0]["params"]]
assert all(gr is not None and torch.isfinite(gr).all() for gr in grads) and sum(float(gr.abs().sum()) for gr in grads) > 0
# --- checkpoint: the reference's CkptManagerSafe saves this group; the native loader rebuilds identical blocks on a fresh UNet
import tempfile, os
from hcpdiff.ckpt_manager import CkptManagerSafe
from hcp_diffusion_amd.ckpt import NativeModelLoader
d = tempfile.mkdtemp()
mgr = CkptManagerSafe(); mgr.set_save_dir(d)
mgr.save_model_with_lora(u, group, name="unet", step=1)
u2 = seeded_init_(NativeUNet2DConditionModel(**MICRO_CONFIG), 1); u2.requires_grad_(False)
g2, _ = NativeModelLoader(u2).load_lora([dict(path=os.path.join(d, "unet-1.safetensors"), alpha=1.0)])
def same(a, b):        # (u runs the fused q|k|v group — q pre-scaled for the attention kernel —, the freshly loaded blocks the per-layer
    return ((a - b).norm() / b.norm()).item() < 2e-2      # path until they are bucketed: same model, two bf16 evaluation orders (1.3e-2 measured))
with torch.no_grad():
    assert same(u2(x, t, ehs).sample, y.detach())
# ... and the reference's OWN HCPModelLoader.load_lora does too once its registry's default entry points at the native class
# (get_lora_rank_and_cls hard-codes lora_layer_map['lora'], cfg_net_tools.py:77-88)
llp.lora_layer_map["lora"] = LoraHipLayer
u3 = seeded_init_(NativeUNet2DConditionModel(**MICRO_CONFIG), 1); u3.requires_grad_(False)
g3 = tools.HCPModelLoader(u3).load_lora([_Item(path=os.path.join(d, "unet-1.safetensors"), alpha=1.0)])
assert len(g3.plugin_dict) == 40 and all(isinstance(b, LoraHipLayer) for b in g3.plugin_dict.values())
with torch.no_grad():
    assert same(u3(x, t, ehs).sample, y.detach())
group.remove()                                            # PluginGroup.remove -> PatchPluginBlock.remove restores the plain hosts
assert type(u.down_blocks[0].attentions[0].transformer_blocks[0].attn1.to_q).__name__ == "HipLinear"
# --- seam 3: the reference's make_plugin builds the native ControlNet (cfg_net_tools.py:130-162, plugin_controlnet.yaml)
builder = functools.partial(ControlNetHipPlugin, lr=1e-4, from_layers=["pre_hook:", "pre_hook:conv_in"],
                            to_layers=["down_blocks.0", "down_blocks.1", "mid_block", "pre_hook:up_blocks.1.resnets.1"],
                            cond_block_channels=(3, 8, 8, 16, 16, 40), layers_per_block=1, block_out_channels=MICRO_CONFIG["block_out_channels"])
train_params, plugin_groups = tools.make_plugin(u, {"controlnet1": builder})
plug = plugin_groups["controlnet1"].plugin_dict[""]
assert isinstance(plug, ControlNetHipPlugin) and len(train_params[0]["params"]) == len(list(plug.parameters())) and train_params[0]["lr"] == 1e-4
for feeder in u.input_feeder:
    feeder(dict(cond=torch.rand(1, 3, 64, 64, generator=g)))
with torch.no_grad():
    y2 = u(x, t, ehs).sample
assert y2.shape == y.shape and torch.isfinite(y2).all()
print("REFERENCE_API_OK")
'''


@pytest.mark.skipif(not os.path.isdir("/root/reference/hcpdiff"), reason="reference tree only exists in the build container")
def test_reference_builders_assemble_the_native_modules():
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + SCRIPT], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "REFERENCE_API_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
