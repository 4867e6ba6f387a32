"""Native CLIP text encoder — the host of the reference's text-encoder LoRA.

The reference's default LoRA example trains ``lora_text_encoder`` next to ``lora_unet``
(cfgs/train/examples/lora_conventional.yaml:14-19: rank-4 blocks on ``re:.*self_attn$`` and ``re:.*mlp$``), which makes the
whole ``TEUnetWrapper`` differentiable (hcpdiff/models/wrapper.py:14-30, train_ac.py:61,160-162): the prompt is encoded inside
the step, and the UNet's cross-attention K/V projections pass a gradient back into the encoder's LoRA factors.

Same module tree, parameter names and shapes as transformers' ``CLIPTextModel`` as dumped in the reference's cfgs/te_struct.txt
(``text_model.embeddings.token_embedding`` ... ``text_model.final_layer_norm``), so checkpoints load by name and the
``re:.*self_attn$`` / ``re:.*mlp$`` selectors wrap the same Linear leaves.  Every leaf is a ``HipLinear`` / ``HipLayerNorm``:
LoRA blocks are the UNet's own ``LoraHipLayer`` (one flat bucket, grouped weight-gradient launch, fused clip + AdamW).
Kernels: ``hcp_embedding_bf16``, LayerNorm, fused-LoRA GEMM, flash attention with ``causal=1`` (12 x 64 heads, 77 tokens),
``hcp_quick_gelu``.  Output selection follows ``TEEXHook.forward_hook`` (textencoder_ex.py:62-79) for N_repeats = 1:
``final_layer_norm(hidden_states[-clip_skip-1])``; ``N_repeats`` > 1 (``tokenizer_repeats``) encodes [B, r x 77] ids as B r prompts and stitches
the chunks back with one BOS and one EOS, as the hook does.
"""
import json
import os

import torch
from torch import nn

from . import kernels as K
from . import ops
from .layers import HipLayerNorm, HipLinear

BF16 = torch.bfloat16
CLIP_L_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                     max_position_embeddings=77)


class _Config(dict):
    __getattr__ = dict.__getitem__          # transformers configs answer cfg.hidden_size as well as cfg['hidden_size']


def _call(m, x, residual=None):
    return m(x, residual=residual) if residual is not None else m(x)


class CLIPAttention(nn.Module):
    def __init__(self, c, heads):
        super().__init__()
        self.k_proj = HipLinear(c, c); self.v_proj = HipLinear(c, c); self.q_proj = HipLinear(c, c); self.out_proj = HipLinear(c, c)
        self.heads = heads

    def forward(self, x, residual, key_bias=None):
        q, k, v = self.q_proj(x), self.k_proj(x), self.v_proj(x)
        return _call(self.out_proj, ops.attention(q, k, v, self.heads, key_bias=key_bias, causal=True), residual)


class CLIPMLP(nn.Module):
    def __init__(self, c, inner):
        super().__init__()
        self.fc1 = HipLinear(c, inner); self.fc2 = HipLinear(inner, c)

    def forward(self, x, residual):
        return _call(self.fc2, ops.quick_gelu(self.fc1(x)), residual)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, c, heads, inner):
        super().__init__()
        self.self_attn = CLIPAttention(c, heads)
        self.layer_norm1 = HipLayerNorm(c, eps=1e-5)
        self.mlp = CLIPMLP(c, inner)
        self.layer_norm2 = HipLayerNorm(c, eps=1e-5)

    def forward(self, x, key_bias=None):
        h, x = self.layer_norm1(x, fork=True)                 # fork: backward adds the residual-path gradient inside the LN kernel
        x = self.self_attn(h, x, key_bias)                    # residual add fused into the out_proj GEMM epilogue
        h, x = self.layer_norm2(x, fork=True)
        return self.mlp(h, x)


class _Embeddings(nn.Module):
    def __init__(self, vocab, c, npos):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, c)
        self.position_embedding = nn.Embedding(npos, c)


class _Encoder(nn.Module):
    def __init__(self, c, heads, inner, n):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(c, heads, inner) for _ in range(n)])


class _TextTransformer(nn.Module):
    def __init__(self, vocab_size, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, max_position_embeddings):
        super().__init__()
        self.embeddings = _Embeddings(vocab_size, hidden_size, max_position_embeddings)
        self.encoder = _Encoder(hidden_size, num_attention_heads, intermediate_size, num_hidden_layers)
        self.final_layer_norm = HipLayerNorm(hidden_size, eps=1e-5)


class NativeCLIPTextModel(nn.Module):
    def __init__(self, clip_skip=0, clip_final_norm=True, N_repeats=1, **cfg):
        super().__init__()
        keys = tuple(CLIP_L_CONFIG)
        self.config = _Config({**CLIP_L_CONFIG, **{k: v for k, v in cfg.items() if k in keys}})
        if self.config["hidden_size"] // self.config["num_attention_heads"] not in (40, 64, 80, 160):
            raise NotImplementedError("hcp_diffusion_amd: text-encoder head width must be one of 40/64/80/160")
        self.text_model = _TextTransformer(**self.config)
        self.clip_skip, self.clip_final_norm, self.N_repeats = clip_skip, clip_final_norm, N_repeats

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    @property
    def dtype(self):                                   # utils/pipe_hook.py:25-26 casts the prompt states to text_encoder.dtype
        return self.text_model.final_layer_norm.weight.dtype

    def enable_hip_graph(self, on=True):
        """As NativeUNet2DConditionModel.enable_hip_graph: under an ordinary eager trainer loop the encoder's forward and backward
        (text-encoder LoRA training, lora_conventional.yaml:14-19) replay captured hipGraphs, one pair per input signature."""
        self._hip_graph, self._hip_graphs = bool(on), {}
        self._hcp_capturable = None

    def forward(self, input_ids, position_ids=None, attention_mask=None, output_hidden_states=None):
        if (getattr(self, "_hip_graph", False) and torch.is_grad_enabled() and input_ids.is_cuda
                and not torch.cuda.is_current_stream_capturing()):
            from . import graphed
            if graphed.capturable_cached(self)[0]:
                ins = [input_ids, position_ids, attention_mask]
                key = tuple(None if t is None else (tuple(t.shape), t.dtype) for t in ins)
                x = graphed.call(self, ins, lambda i_, p_, m_: self._forward_impl(i_, p_, m_), self._hip_graphs, key)
                return (x, None) if output_hidden_states is not None else x
        return self._forward_impl(input_ids, position_ids, attention_mask, output_hidden_states)

    def _forward_impl(self, input_ids, position_ids=None, attention_mask=None, output_hidden_states=None):
        """int64 [B, L] token ids -> bf16 [B, L, C] conditioning states (TEEXHook's selection).  attention_mask [B, L] (1 = attend) is
        combined with the causal mask like transformers' CLIPTextTransformer does; token 0 must stay visible.
        Called the way the reference's wrapper calls its hooked text encoder — ``TE(ids, position_ids=..., attention_mask=...,
        output_hidden_states=True)[0]`` (models/wrapper.py:20,64) — it answers with the hook's tuple ``(states, pooled_output)``
        (pooled_output None: CLIP-L's pooled vector is not used by the SD1.x path)."""
        tm = self.text_model
        B, r = input_ids.shape[0], self.N_repeats
        if r > 1:                                                 # TEEXHook.forward_hook_input (textencoder_ex.py:57-59): 'b (r w) -> (b r) w'
            if input_ids.dim() != 2 or input_ids.shape[1] % r:
                raise ValueError(f"token ids [B, {r} x L] expected for N_repeats={r}, got {tuple(input_ids.shape)}")
            input_ids = input_ids.reshape(B * r, -1)
            attention_mask = attention_mask.reshape(B * r, -1) if attention_mask is not None else None
            position_ids = position_ids.reshape(B * r, -1) if position_ids is not None else None
        if input_ids.dim() != 2 or input_ids.shape[1] > self.config["max_position_embeddings"]:
            raise ValueError(f"expected token ids [B, L <= {self.config['max_position_embeddings']}], got {tuple(input_ids.shape)}")
        if attention_mask is not None and tuple(attention_mask.shape) != tuple(input_ids.shape):
            raise ValueError(f"attention_mask {tuple(attention_mask.shape)} does not match input_ids {tuple(input_ids.shape)}")
        emb = tm.embeddings
        if torch.is_grad_enabled() and (emb.token_embedding.weight.requires_grad or emb.position_embedding.weight.requires_grad):
            raise NotImplementedError("hcp_diffusion_amd: training the embedding tables (prompt tuning) is not implemented")
        x = K.embedding(emb.token_embedding.weight.detach(), input_ids.contiguous(), emb.position_embedding.weight.detach(), position_ids)
        key_bias = None
        if attention_mask is not None:       # [B, L], 1 = attend (wrapper.py:20 passes the tokenizer's mask when encoder_attention_mask is on)
            key_bias = ((1.0 - attention_mask.to(torch.float32)) * -1.0e9).contiguous()       # additive on the keys, on top of the causal mask
        layers = tm.encoder.layers
        for layer in layers[:len(layers) - self.clip_skip]:
            x = layer(x, key_bias)
        x = tm.final_layer_norm(x) if self.clip_final_norm else x
        if r > 1:       # textencoder_ex.py:68-72: one BOS (first chunk), every chunk's inner tokens, one EOS (last chunk) -> [B, r*(L-2)+2, C]
            x = x.reshape(B, r, *x.shape[1:])
            x = torch.cat([x[:, 0, :1, :], x[:, :, 1:-1, :].flatten(1, 2), x[:, -1, -1:, :]], dim=1)
        return (x, None) if output_hidden_states is not None else x

    @classmethod
    def from_pretrained(cls, path=None, subfolder="text_encoder", device="cuda", pretrained_model_name_or_path=None, hip_graph=False, **kw):
        """A diffusers / transformers directory (config.json + model.safetensors) by parameter name (hip_graph: enable_hip_graph())."""
        from safetensors.torch import load_file
        path = path if path is not None else pretrained_model_name_or_path
        root = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
        cfg = json.load(open(os.path.join(root, "config.json")))
        model = cls(**kw, **{k: cfg[k] for k in CLIP_L_CONFIG if k in cfg})
        sd = load_file(os.path.join(root, "model.safetensors"))
        own = model.state_dict()
        sd = {k: v for k, v in sd.items() if k in own}                   # position_ids buffer etc. are ignored
        missing = [k for k in own if k not in sd]
        if missing:
            raise ValueError(f"text-encoder checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
        model.load_state_dict(sd)
        model = model.to(device)
        if hip_graph:
            model.enable_hip_graph()
        return model
