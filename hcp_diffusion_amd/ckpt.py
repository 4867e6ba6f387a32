"""Checkpoint wire format of the path's trainable state — SURVEY.md §8 row f2.

What the reference writes every ``save_step`` (``Trainer.save_model``, hcpdiff/train_ac.py:523-544) and reads back for
inference / resume (``HCPModelLoader``, hcpdiff/utils/cfg_net_tools.py:225-322), restated for the native modules so that
files interchange in BOTH directions with the reference's ``CkptManagerSafe`` / ``CkptManagerPKL``:

  ``{name}-{step}.safetensors``  (ckpt_manager/ckpt_pkl.py:56-72, ckpt_safetensor.py:20-27)
      base          trainable host parameters by their diffusers names, plugin keys stripped (plugin.py:43-55)
      base_ema      the same names from the EMA copy
      lora          ``{host path}.___.layer.W_down | layer.W_up | alpha``          (plugin.py:337-342)
      lora_ema
  ``{name}-{plugin}-{step}.safetensors``  (ckpt_pkl.py:45-53)
      plugin / plugin_ema   ``{block path}.___.{plugin state key}``
  nested dicts are flattened with ':' for safetensors (ckpt_safetensor.py:34-63); ``.ckpt`` files hold the nested dict
  itself (``torch.save``).

The loader side builds native ``LoraHipLayer`` blocks (one flat ``LoraBucket``) from a LoRA file — including the
deprecated ``lora_down/lora_up`` key scheme (tools/convert_old_lora.py) — merges ``base`` parts into the fp32 masters
in place (the bf16 operand copies are refreshed through the parameters' version counters) and restores plugin weights.
Host logic only: no kernel is involved except the bucket's pack launch.
"""
import os
import warnings

import torch

from .lora import LoraBucket, LoraHipLayer, get_match_layers
from .patch_api import BasePluginBlock, PluginGroup

SPLIT_KEY = ":"


def unfold_dict(data, split_key=SPLIT_KEY):
    """{'lora': {'a.___.layer.W_up': t}} -> {'lora:a.___.layer.W_up': t}; lists/tuples by index (ckpt_safetensor.py:34-48)."""
    flat = {}

    def walk(prefix, node):
        for k, v in node.items():
            key = f"{k}" if prefix == "" else f"{prefix}{split_key}{k}"
            if isinstance(v, dict):
                walk(key, v)
            elif isinstance(v, (list, tuple)):
                walk(key, dict(enumerate(v)))
            else:
                flat[key] = v

    walk("", data)
    return flat


def fold_dict(flat, split_key=SPLIT_KEY):
    """Inverse of unfold_dict (ckpt_safetensor.py:50-63)."""
    out = {}
    for k, v in flat.items():
        node = out
        parts = k.split(split_key)
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = v
    return out


def trainable_state_without_plugins(model):
    """BasePluginBlock.extract_state_without_plugin(model, trainable=True) (plugin.py:43-55)."""
    trainable = {k for k, v in model.named_parameters() if v.requires_grad}
    plugin_names = [k for k, v in model.named_modules() if isinstance(v, BasePluginBlock)]
    return {k: v for k, v in model.state_dict().items()
            if k in trainable and not any(k.startswith(n) for n in plugin_names)}


class _EMAView:
    """What the reference passes as ``model_ema``: an object whose state_dict() maps full parameter/buffer names to the
    EMA tensors (utils/ema.py:8-10,46-47).  Buffers (LoRA ``alpha``) are not averaged: they come from the live model."""

    def __init__(self, ema_params, model):
        self._sd = dict(model.state_dict())
        self._sd.update(ema_params)

    def state_dict(self):
        return self._sd


class CkptManagerNative:
    """Method surface of CkptManagerPKL / CkptManagerSafe for the UNet half (ckpt_pkl.py:22-79); ``fmt`` picks the
    container: 'safetensors' (CkptManagerSafe) or 'ckpt' (CkptManagerPKL)."""

    def __init__(self, plugin_from_raw=False, fmt="safetensors", **kwargs):
        if fmt not in ("safetensors", "ckpt"):
            raise ValueError(f"Unknown checkpoint format {fmt}")
        self.plugin_from_raw, self.fmt = plugin_from_raw, fmt
        self.save_dir = None

    def set_save_dir(self, save_dir, emb_dir=None):
        os.makedirs(save_dir, exist_ok=True)
        self.save_dir, self.emb_dir = save_dir, emb_dir

    @staticmethod
    def exclude_state(state, key):
        return state if key is None else {k: v for k, v in state.items() if key not in k}

    # ---- writers
    def save_model_with_lora(self, model, lora_blocks, name, step, model_ema=None, exclude_key=None):
        sd = {"base": self.exclude_state(trainable_state_without_plugins(model), exclude_key)} if model is not None else {}
        has_lora = lora_blocks is not None and not lora_blocks.empty()
        if has_lora:
            sd["lora"] = lora_blocks.state_dict(model if self.plugin_from_raw else None)
        if model_ema is not None:
            ema = model_ema.state_dict()
            if model is not None:
                sd["base_ema"] = self.exclude_state({k: ema[k] for k in sd["base"]}, exclude_key)
            if has_lora:
                sd["lora_ema"] = lora_blocks.state_dict(model_ema)
        return self._save_ckpt(sd, name, step)

    def save_plugins(self, host_model, plugins, name, step, model_ema=None):
        paths = []
        for plugin_name, group in plugins.items():
            sd = {"plugin": group.state_dict(host_model if self.plugin_from_raw else None)}
            if model_ema is not None:
                sd["plugin_ema"] = group.state_dict(model_ema)
            paths.append(self._save_ckpt(sd, f"{name}-{plugin_name}", step))
        return paths

    def _save_ckpt(self, sd_model, name=None, step=None, save_path=None):
        if save_path is None:
            save_path = os.path.join(self.save_dir, f"{name}-{step}.{self.fmt}")
        if save_path.endswith(".safetensors"):
            from safetensors.torch import save_file
            flat = {k: v.detach().to("cpu").contiguous() for k, v in unfold_dict(sd_model).items()}   # channels-last masters, views
            save_file({k: v.clone() for k, v in flat.items()}, save_path)                             # no shared storage
        else:
            torch.save(fold_dict({k: v.detach().to("cpu").contiguous().clone() for k, v in unfold_dict(sd_model).items()}), save_path)
        return save_path

    # ---- reader (either container, chosen by extension like ckpt_manager/__init__.py auto_manager)
    @staticmethod
    def load_ckpt(ckpt_path, map_location="cpu"):
        if ckpt_path.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(ckpt_path, framework="pt", device=map_location) as f:
                return fold_dict({k: f.get_tensor(k) for k in f.keys()})
        return torch.load(ckpt_path, map_location=map_location)


def _convert_old_lora_state(state):
    """'layer.lora_down.weight' / 'layer.lora_up.weight' (+bias) -> W_down / W_up (tools/convert_old_lora.py:4-14)."""
    new = {"layer.W_down": state["layer.lora_down.weight"], "layer.W_up": state["layer.lora_up.weight"]}
    if "layer.lora_up.bias" in state:
        new["layer.bias"] = state["layer.lora_up.bias"]
    if "alpha" in state:
        new["alpha"] = state["alpha"]
    return new


class NativeModelLoader:
    """HCPModelLoader (cfg_net_tools.py:225-322) for a native host.  cfg items are mappings with the reference's keys:
    ``path`` (+ ``alpha``, ``layers``, ``alpha_auto_scale``, ``dropout``)."""

    def __init__(self, host):
        self.host = host
        self.named_modules = dict(host.named_modules())
        self.named_params = dict(host.named_parameters())

    @staticmethod
    def _get(item, key, default):
        return item.get(key, default) if isinstance(item, dict) else getattr(item, key, default)

    @torch.no_grad()
    def load_part(self, cfg, base_model_alpha=0.0, load_ema=False):
        """p <- base_model_alpha * p + item.alpha * ckpt[p]  for the (selected) 'base' entries (cfg_net_tools.py:232-246)."""
        for item in cfg or []:
            path = self._get(item, "path", None)
            state = CkptManagerNative.load_ckpt(path)["base_ema" if load_ema else "base"]
            layers = self._get(item, "layers", "all")
            if layers != "all":
                blocks = get_match_layers(layers, self.named_modules)
                state = {k: v for blk in blocks for k, v in state.items() if k.startswith(blk)}
            for k, v in state.items():
                p = self.named_params[k]
                p.mul_(base_model_alpha).add_(v.to(p.device, p.dtype), alpha=float(self._get(item, "alpha", 1.0)))   # in place:
                # bucket views stay views, and the version bump re-packs the layer's bf16 operand copies on next use

    @torch.no_grad()
    def load_lora(self, cfg, base_model_alpha=1.0, load_ema=False):
        """Wrap every host named in the file(s) with a native LoRA block and load its factors (cfg_net_tools.py:248-292).
        Returns (PluginGroup keyed ``{layer path}.{block name}``, LoraBucket) — None, None for an empty cfg."""
        if not cfg:
            return None, None
        all_blocks = {}
        for lora_id, item in enumerate(cfg):
            state = CkptManagerNative.load_ckpt(self._get(item, "path", None))["lora_ema" if load_ema else "lora"]
            per_layer = {}
            for name, p in state.items():
                sep = ".___." if name.rfind("lora_block.") == -1 else ".lora_block."           # old key scheme
                prefix, block_key = name.split(sep, 1)
                per_layer.setdefault(prefix, {})[block_key] = p
            layers = self._get(item, "layers", "all")
            if layers != "all":
                match = get_match_layers(layers, self.named_modules)
                per_layer = {k: v for k, v in per_layer.items() if any(k.startswith(m) for m in match)}
            for layer_name, lstate in per_layer.items():
                parent_name, _, host_name = layer_name.rpartition(".")
                if "layer.lora_down.weight" in lstate:
                    warnings.warn("The old lora format is deprecated.", DeprecationWarning)
                    lstate = _convert_old_lora_state(lstate)
                elif "layer.W_down" not in lstate:
                    raise ValueError("Unknown lora format.")
                lstate = {k: v for k, v in lstate.items() if k != "alpha"}                      # alpha comes from the cfg item
                rank = lstate["layer.W_down"].shape[0]
                blk = LoraHipLayer.wrap_layer(lora_id, self.named_modules[layer_name], rank=rank,
                                              dropout=self._get(item, "dropout", 0.0), alpha=self._get(item, "alpha", 1.0),
                                              bias="layer.bias" in lstate, alpha_auto_scale=self._get(item, "alpha_auto_scale", True),
                                              parent_block=self.named_modules[parent_name], host_name=host_name)
                missing, unexpected = blk.load_state_dict(lstate, strict=False)
                if unexpected:
                    raise ValueError(f"lora checkpoint entries {unexpected} have no counterpart in {type(blk).__name__}")
                all_blocks[f"{layer_name}.{blk.name}"] = blk
        self.named_modules = dict(self.host.named_modules())
        self.named_params = dict(self.host.named_parameters())
        return PluginGroup(all_blocks), LoraBucket(all_blocks.values())

    @torch.no_grad()
    def load_plugin(self, cfg, load_ema=False):
        """{plugin name: {path, layers?, ...}}: the '___' placeholder of every key becomes the plugin's name and the result is
        loaded non-strictly (cfg_net_tools.py:294-315).  A MultiPluginBlock (ControlNet) is saved under the empty block
        path, i.e. as '.___.<key>': those keys address the plugin object ``host.<name>`` directly."""
        for name, item in (cfg or {}).items():
            state = CkptManagerNative.load_ckpt(self._get(item, "path", None))["plugin_ema" if load_ema else "plugin"]
            layers = self._get(item, "layers", "all")
            if layers != "all":
                match = get_match_layers(layers, self.named_modules)
                state = {k: v for blk in match for k, v in state.items() if k.startswith(blk)}
            multi = getattr(self.host, name, None)
            own = {k[len(".___."):]: v for k, v in state.items() if k.startswith(".___.")}
            if own:
                if multi is None:
                    raise ValueError(f"checkpoint holds a whole-model plugin but the host has no plugin named {name!r}")
                multi.load_state_dict(own, strict=True)
            rest = {k.replace("___", name): v for k, v in state.items() if not k.startswith(".___.")}
            if rest:
                self.host.load_state_dict(rest, strict=False)
            # every remaining cfg key is a hyper-parameter of the plugin(s) (cfg_net_tools.py:307-315: e.g. a ControlNet's scale)
            hyper = {k: v for k, v in dict(item).items() if k not in ("path", "layers")}
            if hyper:
                targets = [multi] if multi is not None else [self.named_modules[k.split("___", 1)[0] + name] for k in
                                                             {k.split("___", 1)[0] for k in state if not k.startswith(".___.")}]
                for tgt in targets:
                    if not hasattr(tgt, "set_hyper_params"):
                        raise ValueError(f"plugin {name!r}: cfg keys {sorted(hyper)} given but {type(tgt).__name__} has no set_hyper_params")
                    tgt.set_hyper_params(**hyper)

    def load_all(self, cfg_merge, load_ema=False):
        self.load_part(cfg_merge.get("part", []), base_model_alpha=cfg_merge.get("base_model_alpha", 0.0), load_ema=load_ema)
        group = self.load_lora(cfg_merge.get("lora", []), base_model_alpha=cfg_merge.get("base_model_alpha", 1.0), load_ema=load_ema)
        self.load_plugin(cfg_merge.get("plugin", {}), load_ema=load_ema)
        return group
