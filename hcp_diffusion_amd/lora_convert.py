"""LoRA checkpoint key conversion between the reference's layout and the webui (kohya) layout — SURVEY §8 f2, the wire format's last
piece (reference ``hcpdiff/tools/lora_convert.py:9-196``, CLI ``:198-236``).  Pure key / scale bookkeeping on CPU state dicts; no kernel.

reference layout (``ckpt.py``)        ``{module path}.___.layer.W_down | .___.layer.W_up | .___.alpha``   (sections ``lora`` of the UNet file
                                      and of the text-encoder file)
webui layout                          ``lora_unet_{path with '_' for '.'}.lora_down.weight | .lora_up.weight | .alpha`` and ``lora_te_…``
                                      (SDXL: ``lora_te1_…`` / ``lora_te2_…`` for clip_B / clip_bigG, and the UNet in the original
                                      ``input_blocks / middle_block / output_blocks`` numbering)

A webui module name has lost the dots, and some path components contain underscores themselves: those components are the fixed vocabulary
below (the reference's, ``lora_convert.py:10-11``); everything else splits at '_'.
"""
import argparse
import math
import os
import re

UNET_WORDS = ("down_blocks", "up_blocks", "mid_block", "transformer_blocks", "to_q", "to_k", "to_v", "to_out", "proj_in", "proj_out",
              "input_blocks", "middle_block", "output_blocks")
TE_WORDS = ("self_attn", "q_proj", "v_proj", "k_proj", "out_proj", "text_model")
WEIGHT_OF = {"lora_down.weight": "W_down", "lora_up.weight": "W_up"}
WEBUI_OF = {"W_down": "lora_down.weight", "W_up": "lora_up.weight"}

# SDXL in the original numbering -> diffusers blocks (lora_convert.py:131-146): input_blocks.{4,5,7,8}.1 are the attentions of down
# blocks 1 and 2, output_blocks.{0..5}.1 those of up blocks 0 and 1, middle_block.1 the mid attention
_XL_DOWN = {"4": (1, 0), "5": (1, 1), "7": (2, 0), "8": (2, 1)}
_XL_UP = {str(i): divmod(i, 3) for i in range(6)}


def _dotted(name, words):
    """'down_blocks_0_attentions_0_to_q' -> 'down_blocks.0.attentions.0.to_q'."""
    for w in words:
        name = name.replace(w, w.replace("_", "%"))
    return name.replace("_", ".").replace("%", "_")


def _from_webui_part(state, prefix, words, rename=None):
    out = {}
    for k, v in state.items():
        if not k.startswith(prefix):
            continue
        module, what = k[len(prefix):].split(".", 1)
        module = _dotted(module, words)
        if rename is not None:
            module = rename(module)
        out[f"{module}.___.alpha" if what == "alpha" else f"{module}.___.layer.{WEIGHT_OF[what]}"] = v
    return out


def _xl_unet_name(module):
    m = re.match(r"input_blocks\.(\d+)\.1\.(.+)", module)
    if m:
        b, a = _XL_DOWN[m.group(1)]
        return f"down_blocks.{b}.attentions.{a}.{m.group(2)}"
    m = re.match(r"middle_block\.1\.(.+)", module)
    if m:
        return f"mid_block.attentions.0.{m.group(1)}"
    m = re.match(r"output_blocks\.(\d+)\.(\d+)\.(.+)", module)
    if m:
        b, a = _XL_UP[m.group(1)]
        return f"up_blocks.{b}.attentions.{a}.{m.group(3)}"
    raise NotImplementedError(f"SDXL webui module {module!r}: only attention layers are mapped (reference lora_convert.py:166-174)")


def _rescale(state, is_up, is_down):
    """`auto_scale_alpha` (lora_convert.py:187-205): both factors times sqrt(rank), so that alpha / rank conventions agree."""
    for k, v in state.items():
        if is_up(k):
            state[k] = v * math.sqrt(v.shape[1])
        elif is_down(k):
            state[k] = v * math.sqrt(v.shape[0])
    return state


def from_webui(state, auto_scale_alpha=False, sdxl=False):
    """webui state dict -> ({'lora': text-encoder part}, {'lora': UNet part}) in the reference's key layout (lora_convert.py:23-37)."""
    if sdxl:
        unet = _from_webui_part(state, "lora_unet_", UNET_WORDS, rename=_xl_unet_name)
        te = _from_webui_part(state, "lora_te1_", TE_WORDS, rename=lambda m: f"clip_B.{m}")
        te.update(_from_webui_part(state, "lora_te2_", TE_WORDS, rename=lambda m: f"clip_bigG.{m}"))
    else:
        unet = _from_webui_part(state, "lora_unet_", UNET_WORDS)
        te = _from_webui_part(state, "lora_te_", TE_WORDS)
    if auto_scale_alpha:
        for part in (unet, te):
            _rescale(part, lambda k: "W_up" in k, lambda k: "W_down" in k)
    return {"lora": te}, {"lora": unet}


def _to_webui_part(state, prefix, xl_te=False):
    out = {}
    for k, v in state.items():
        module, what = k.split(".___.", 1)
        if what.startswith("layer."):                 # layer.W_down / layer.W_up (the old format's layer.lora_down.weight passes through)
            what = WEBUI_OF.get(what[len("layer."):], what[len("layer."):])
        key = f"{prefix}{module.replace('.', '_')}.{what}"
        if xl_te and "clip" in key:                   # lora_te_clip_B_... -> lora_te1_..., lora_te_clip_bigG_... -> lora_te2_...
            key = key.replace("_clip_B", "1") if "clip_B" in key else key.replace("_clip_bigG", "2")
        out[key] = v
    return out


def to_webui(sd_unet, sd_te=None, auto_scale_alpha=False, sdxl=False):
    """The 'lora' sections of a UNet file and (optionally) a text-encoder file -> one webui state dict (lora_convert.py:39-49).  The
    UNet keeps diffusers block names, as the reference's converter leaves them (webui resolves both numberings)."""
    out = _to_webui_part(sd_unet, "lora_unet_")
    out.update(_to_webui_part(sd_te or {}, "lora_te_", xl_te=sdxl))
    if auto_scale_alpha:
        _rescale(out, lambda k: "lora_up" in k, lambda k: "lora_down" in k)
    return out


def main(argv=None):
    """``python -m hcp_diffusion_amd.lora_convert --lora_path … --dump_path … (--from_webui | --to_webui)`` — the reference CLI's
    arguments and file names (lora_convert.py:198-236), through ckpt.CkptManagerNative's readers / writers."""
    from .ckpt import CkptManagerNative
    load_ckpt_file = CkptManagerNative.load_ckpt
    save_ckpt_file = lambda sd, path: CkptManagerNative()._save_ckpt(sd, save_path=path)   # noqa: E731
    ap = argparse.ArgumentParser()
    ap.add_argument("--lora_path", required=True)
    ap.add_argument("--lora_path_TE", default=None)
    ap.add_argument("--dump_path", required=True)
    ap.add_argument("--from_webui", action="store_true")
    ap.add_argument("--to_webui", action="store_true")
    ap.add_argument("--auto_scale_alpha", action="store_true")
    ap.add_argument("--sdxl", action="store_true")
    args = ap.parse_args(argv)
    name = os.path.basename(args.lora_path)
    if args.from_webui:
        sd_te, sd_unet = from_webui(load_ckpt_file(args.lora_path), auto_scale_alpha=args.auto_scale_alpha, sdxl=args.sdxl)
        os.makedirs(args.dump_path, exist_ok=True)
        save_ckpt_file(sd_te, os.path.join(args.dump_path, "TE-" + name))
        save_ckpt_file(sd_unet, os.path.join(args.dump_path, "unet-" + name))
    elif args.to_webui:
        sd_unet = load_ckpt_file(args.lora_path)
        sd_te = load_ckpt_file(args.lora_path_TE) if args.lora_path_TE else {"lora": {}}
        save_ckpt_file(to_webui(sd_unet["lora"], sd_te["lora"], auto_scale_alpha=args.auto_scale_alpha, sdxl=args.sdxl), args.dump_path)
    else:
        ap.error("one of --from_webui / --to_webui")


if __name__ == "__main__":
    main()
