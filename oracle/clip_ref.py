"""ORACLE / test infrastructure (never imported by the product path): CPU fp32 restatement of the CLIP text encoder the
reference conditions on (``CLIPTextModel`` from the un-vendored ``transformers``; loaded at hcpdiff/train_ac.py:209-218, called at
hcpdiff/models/wrapper.py:20, post-processed by ``TEEXHook.forward_hook`` hcpdiff/models/textencoder_ex.py:62-79).

Structure, parameter names and shapes follow the reference's own dump /root/reference/cfgs/te_struct.txt (token_embedding
49408x768, position_embedding 77x768, 12 x CLIPEncoderLayer {self_attn q/k/v/out_proj 768, layer_norm1, mlp fc1 768->3072
QuickGELU fc2, layer_norm2}, final_layer_norm).  The arithmetic — pre-LN residual blocks, attention scale d^-0.5 applied to
the scores, the causal mask, quick_gelu(x) = x sigmoid(1.702 x), LayerNorm eps 1e-5 — is [ext] transformers knowledge; it is
PINNED numerically by tests/test_text_encoder.py against the CLIPTextModel of the transformers build installed in this image
(same weights through a key-name map), so this oracle is checked against real third-party code, not only restated.

``encode(ids, clip_skip, final_norm)`` is TEEXHook's selection for N_repeats = 1: final_layer_norm(hidden_states[-clip_skip-1])
(textencoder_ex.py:63-65; the BOS/EOS re-concatenation of :71-72 is the identity for one repeat).
"""
import torch
from torch import nn

CLIP_L_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                     max_position_embeddings=77)
TINY_CLIP_CONFIG = dict(vocab_size=100, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                        max_position_embeddings=77)


class CLIPAttention(nn.Module):
    def __init__(self, c, heads):
        super().__init__()
        self.k_proj = nn.Linear(c, c); self.v_proj = nn.Linear(c, c); self.q_proj = nn.Linear(c, c); self.out_proj = nn.Linear(c, c)
        self.heads = heads

    def forward(self, x, attention_mask=None):
        B, L, C = x.shape
        d = C // self.heads
        q, k, v = (p(x).view(B, L, self.heads, d).transpose(1, 2) for p in (self.q_proj, self.k_proj, self.v_proj))
        s = q @ k.transpose(-1, -2) * d ** -0.5
        if attention_mask is not None:                       # [B, L], 1 = attend: additive on the keys, on top of the causal mask
            s = s + ((1.0 - attention_mask.to(s.dtype)) * torch.finfo(s.dtype).min)[:, None, None, :]
        s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool, device=x.device), 1), float("-inf"))
        return self.out_proj((torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, C))


class CLIPMLP(nn.Module):
    def __init__(self, c, inner):
        super().__init__()
        self.fc1 = nn.Linear(c, inner); self.fc2 = nn.Linear(inner, c)

    def forward(self, x):
        h = self.fc1(x)
        return self.fc2(h * torch.sigmoid(1.702 * h))


class CLIPEncoderLayer(nn.Module):
    def __init__(self, c, heads, inner):
        super().__init__()
        self.self_attn = CLIPAttention(c, heads)
        self.layer_norm1 = nn.LayerNorm(c, eps=1e-5)
        self.mlp = CLIPMLP(c, inner)
        self.layer_norm2 = nn.LayerNorm(c, eps=1e-5)

    def forward(self, x, attention_mask=None):
        x = x + self.self_attn(self.layer_norm1(x), attention_mask)
        return x + self.mlp(self.layer_norm2(x))


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
        self.final_layer_norm = nn.LayerNorm(hidden_size, eps=1e-5)


class OracleCLIPTextModel(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        self.config = {**CLIP_L_CONFIG, **cfg}
        self.text_model = _TextTransformer(**self.config)

    def hidden_states(self, input_ids, position_ids=None, attention_mask=None):
        tm = self.text_model
        L = input_ids.shape[-1]
        pos = position_ids if position_ids is not None else torch.arange(L, device=input_ids.device)[None]
        x = tm.embeddings.token_embedding(input_ids) + tm.embeddings.position_embedding(pos)
        hs = [x]
        for layer in tm.encoder.layers:
            x = layer(x, attention_mask)
            hs.append(x)
        return hs

    def encode(self, input_ids, position_ids=None, clip_skip=0, final_norm=True, attention_mask=None, n_repeats=1):
        """TEEXHook (textencoder_ex.py:57-72): ids [B, r*77] are encoded as B*r prompts of 77 tokens (forward_hook_input :57-59), the
        selected hidden state goes through final_layer_norm (:63-65), then the r chunks are stitched back together keeping ONE BOS
        (first chunk) and ONE EOS (last chunk): [B, r*75 + 2, C] (:68-72)."""
        B = input_ids.shape[0]
        if n_repeats > 1:
            input_ids = input_ids.reshape(B * n_repeats, -1)
            attention_mask = attention_mask.reshape(B * n_repeats, -1) if attention_mask is not None else None
        h = self.hidden_states(input_ids, position_ids, attention_mask)[-clip_skip - 1]
        h = self.text_model.final_layer_norm(h) if final_norm else h
        if n_repeats > 1:
            h = h.reshape(B, n_repeats, *h.shape[1:])
            h = torch.cat([h[:, 0, :1, :], h[:, :, 1:-1, :].flatten(1, 2), h[:, -1, -1:, :]], dim=1)
        return h

    def forward(self, input_ids, position_ids=None):
        return self.encode(input_ids, position_ids)
