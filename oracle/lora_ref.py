"""ORACLE (test infrastructure): CPU fp32 restatement of the reference's LoRA arithmetic.

Follows hcpdiff/models/lora_base_patch.py:20-35 (LoraPatchContainer.forward: sum of get_weight() over blocks, ONE host
op with the merged weight), :59 (alpha = cfg_alpha / rank), :61-62 (get_weight = W_up @ W_down * alpha), :68-74
(post_forward: layer(x, W_host + dW, bias)), and hcpdiff/models/lora_layers_patch.py:31-57 (LinearLayer: W_down [r,in]
kaiming_uniform(a=sqrt 5), W_up [out,r] zeros, forward = mm(x2d, W^T) + bias).
PARITY STATUS: pinned — tests/test_oracle.py checks this file against tests/golden/lora_reference.pt, produced by
oracle/make_golden.py from the reference's own code (oracle/ref_shims.py).
"""
import math
import re

import torch
from torch import nn


class OracleLoraLinear(nn.Module):
    """Takes the place of an nn.Linear; parameter names mirror the reference container:
    `_host.weight`, `lora_block_0.layer.W_down`, `lora_block_0.layer.W_up`, buffer `lora_block_0.alpha`."""

    class _Layer(nn.Module):
        def __init__(self, in_f, out_f, rank):
            super().__init__()
            self.W_down = nn.Parameter(torch.empty(rank, in_f))
            self.W_up = nn.Parameter(torch.empty(out_f, rank))
            nn.init.kaiming_uniform_(self.W_down, a=math.sqrt(5))
            nn.init.zeros_(self.W_up)

    class _Block(nn.Module):
        def __init__(self, in_f, out_f, rank, alpha):
            super().__init__()
            self.layer = OracleLoraLinear._Layer(in_f, out_f, rank)
            self.register_buffer("alpha", torch.tensor(alpha / rank))

    def __init__(self, host: nn.Linear, rank, alpha=1.0):
        super().__init__()
        self._host = host
        self.lora_block_0 = self._Block(host.in_features, host.out_features, rank, alpha)

    def forward(self, x):
        blk = self.lora_block_0
        w = self._host.weight + torch.mm(blk.layer.W_up, blk.layer.W_down) * blk.alpha
        y = torch.mm(x.reshape(-1, x.shape[-1]), w.t()).view(*x.shape[:-1], -1)
        return y if self._host.bias is None else y + self._host.bias


class OracleLoraConv2d(nn.Module):
    """Conv2d host (LoCon): hcpdiff/models/lora_layers_patch.py:64-100 — W_down [r,Cin,kh,kw] kaiming_uniform(a=sqrt 5),
    W_up [Cout,r,1,1] zeros, merged weight = einsum('o r ..., r i ... -> o i ...') * alpha added to the host weight, ONE conv."""

    class _Layer(nn.Module):
        def __init__(self, host, rank):
            super().__init__()
            self.W_down = nn.Parameter(torch.empty(rank, host.in_channels, *host.kernel_size))
            self.W_up = nn.Parameter(torch.empty(host.out_channels, rank, 1, 1))
            nn.init.kaiming_uniform_(self.W_down, a=math.sqrt(5))
            nn.init.zeros_(self.W_up)

    class _Block(nn.Module):
        def __init__(self, host, rank, alpha):
            super().__init__()
            self.layer = OracleLoraConv2d._Layer(host, rank)
            self.register_buffer("alpha", torch.tensor(alpha / rank))

    def __init__(self, host: nn.Conv2d, rank, alpha=1.0):
        super().__init__()
        self._host = host
        self.lora_block_0 = self._Block(host, rank, alpha)

    def forward(self, x):
        blk, h = self.lora_block_0, self._host
        dw = torch.einsum("or,rikl->oikl", blk.layer.W_up[:, :, 0, 0], blk.layer.W_down) * blk.alpha
        return torch.nn.functional.conv2d(x, h.weight + dw, h.bias, h.stride, h.padding, h.dilation, h.groups)


def wrap_lora(model, patterns, rank, alpha=1.0, conv=False):
    """Restates make_hcpdiff's layer selection (utils/cfg_net_tools.py:30-75,108-123): `re:` patterns are
    `re.match`-anchored on module paths; every nn.Linear (conv=True: and nn.Conv2d, lora_base_patch.py:39) under a
    matched module is wrapped."""
    named = dict(model.named_modules())
    hits = []
    for pat in patterns:
        rx = re.compile(pat[3:]) if pat.startswith("re:") else None
        for name in named:
            if (rx.match(name) if rx else name == pat):
                hits.append(name)
    wrapped = {}
    for top in sorted(set(hits), key=hits.index):
        for sub, mod in list(named[top].named_modules()):
            if isinstance(mod, (nn.Linear, nn.Conv2d) if conv else nn.Linear) and "_host" not in sub:
                path = f"{top}.{sub}" if sub else top
                parent_path, _, leaf = path.rpartition(".")
                parent = dict(model.named_modules())[parent_path]
                w = OracleLoraLinear(mod, rank, alpha) if isinstance(mod, nn.Linear) else OracleLoraConv2d(mod, rank, alpha)
                setattr(parent, leaf, w)
                wrapped[path] = w
    return wrapped
