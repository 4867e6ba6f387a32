"""Host-side mirror of the slice of the reference plugin API the LoRA hot path sits behind.

When ``hcpdiff`` itself is importable (a real HCP-Diffusion install) the native LoRA classes subclass the
reference's own ``PatchPluginBlock`` / ``PatchPluginContainer`` (hcpdiff/models/plugin.py:223-315) so
``make_hcpdiff`` (utils/cfg_net_tools.py:90-128), ``PluginGroup.state_dict`` (plugin.py:317-348) and the
checkpoint managers treat them as first-class plugins.  In this container ``import hcpdiff`` fails
(diffusers/hydra absent), so the same contract is restated here: same class roles, attribute
names (`_host`, `plugin_names`, `host_name`, `name`), child-replacement semantics and `{host}.___.{param}`
checkpoint keys.
"""
import re
import weakref

from torch import nn

try:  # pragma: no cover - only on a machine with the reference installed
    from hcpdiff.models.plugin import BasePluginBlock, MultiPluginBlock, PatchPluginBlock, PatchPluginContainer, PluginGroup  # noqa: F401
    USING_REFERENCE_PLUGIN_API = True
except Exception:  # noqa: BLE001
    USING_REFERENCE_PLUGIN_API = False

    def _split(path):
        parent, _, leaf = path.rpartition(".")
        return parent, leaf

    class BasePluginBlock(nn.Module):
        def __init__(self, name):
            super().__init__()
            self.name = name

        def remove(self):
            pass

        def get_trainable_parameters(self):
            return self.parameters()

        def set_hyper_params(self, **kwargs):          # reference plugin.py:39-41
            for k, v in kwargs.items():
                setattr(self, k, v)

    class MultiPluginBlock(BasePluginBlock):
        """Role marker of a whole-model plugin (reference plugin.py:175-222): make_plugin builds subclasses with
        (name, host_model, from_layers, to_layers) (cfg_net_tools.py:148-162)."""

    class PatchPluginContainer(nn.Module):
        """Takes the host's place inside its parent; plugins become children named ``plugin_names[i]``."""

        def __init__(self, host_name, host, parent_block):
            super().__init__()
            self._host = host
            self.host_name = host_name
            self.parent_block = weakref.ref(parent_block)
            self.plugin_names = []
            delattr(parent_block, host_name)
            setattr(parent_block, host_name, self)

        def add_plugin(self, name, plugin):
            setattr(self, name, plugin)
            self.plugin_names.append(name)

        def remove_plugin(self, name):
            delattr(self, name)
            self.plugin_names.remove(name)
            if not self.plugin_names:
                self.remove()

        def remove(self):
            parent = self.parent_block()
            delattr(parent, self.host_name)
            setattr(parent, self.host_name, self._host)

        def __getitem__(self, name):
            return getattr(self, name)

        def __iter__(self):
            return ((n, getattr(self, n)) for n in self.plugin_names)

        def forward(self, *args, **kwargs):
            for _, plugin in self:
                args, kwargs = plugin.pre_forward(*args, **kwargs)
            out = self._host(*args, **kwargs)
            for _, plugin in self:
                out = plugin.post_forward(out, *args, **kwargs)
            return out

    class PatchPluginBlock(BasePluginBlock):
        container_cls = PatchPluginContainer
        wrapable_classes = ()

        def __init__(self, name, host, host_model=None, parent_block=None, host_name=None):
            super().__init__(name)
            real_host = host._host if isinstance(host, self.container_cls) else host
            self.host = weakref.ref(real_host)
            self.parent_block = weakref.ref(parent_block)
            self.host_name = host_name
            container = host if isinstance(host, self.container_cls) else self.container_cls(host_name, host, parent_block)
            container.add_plugin(name, self)
            self.container = weakref.ref(container)

        def pre_forward(self, *args, **kwargs):
            return args, kwargs

        def post_forward(self, output, *args, **kwargs):
            return output

        def remove(self):
            self.container().remove_plugin(self.name)

        @classmethod
        def wrap_layer(cls, name, layer, **kwargs):
            return cls(name, layer, **kwargs)

        @classmethod
        def _walk(cls, module, prefix, skip_key, skip_classes, seen):
            if module in seen:
                return
            seen.add(module)
            if (skip_key and re.search(skip_key, prefix)) or isinstance(module, skip_classes):
                return
            yield prefix, module
            for child_name, child in module._modules.items():
                if child is not None:
                    yield from cls._walk(child, f"{prefix}.{child_name}" if prefix else child_name, skip_key, skip_classes, seen)

        @classmethod
        def wrap_model(cls, name, host, exclude_key=None, exclude_classes=tuple(), **kwargs):
            """{relative layer path: plugin}.  Wraps `host` itself when it is wrapable, else every wrapable
            descendant (not descending into already-wrapped `_host` modules)."""
            if isinstance(host, cls.wrapable_classes):
                return {"": cls.wrap_layer(name, host, **kwargs)}
            modules = dict(cls._walk(host, "", exclude_key or "_host", tuple(exclude_classes), set()))
            out = {}
            for path, layer in modules.items():
                if isinstance(layer, cls.wrapable_classes) or isinstance(layer, cls.container_cls):
                    if "parent_block" in kwargs:
                        parent_path, leaf = _split(path)
                        kwargs["parent_block"] = modules[parent_path]
                        kwargs["host_name"] = leaf
                    out[path] = cls.wrap_layer(name, layer, **kwargs)
            return out

    class PluginGroup:
        """{host path: plugin}; checkpoint keys are ``{host path}.___.{plugin state key}``."""

        def __init__(self, plugin_dict):
            self.plugin_dict = plugin_dict

        def __getitem__(self, k):
            return self.plugin_dict[k]

        def __setitem__(self, k, v):
            self.plugin_dict[k] = v

        def empty(self):
            return len(self.plugin_dict) == 0

        @property
        def plugin_name(self):
            return None if self.empty() else next(iter(self.plugin_dict.values())).name

        def remove(self):
            for p in self.plugin_dict.values():
                p.remove()

        def state_dict(self, model=None):
            if model is None:
                return {f"{k}.___.{ks}": vs for k, v in self.plugin_dict.items() for ks, vs in v.state_dict().items()}
            sd = model.state_dict()
            return {f"{k}.___.{ks}": sd[f"{k}.{v.name}.{ks}"] for k, v in self.plugin_dict.items() for ks in v.state_dict()}

        def state_keys_raw(self):
            return [f"{k}.{v.name}.{ks}" for k, v in self.plugin_dict.items() for ks in v.state_dict()]
