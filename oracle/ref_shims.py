"""ORACLE / test infrastructure: import the reference's OWN LoRA + plugin code, unmodified, from /root/reference.

`import hcpdiff` fails in this container (hcpdiff/__init__.py:1 -> train_ac.py:18 imports diffusers, hydra, ...;
SURVEY.md §8c), so stub *packages* are registered whose ``__path__`` points into the reference tree (their
``__init__.py`` files are thereby skipped) plus two tiny shim modules for names imported at module scope:
``diffusers.optimization`` (needed by hcpdiff/utils/net_utils.py:6) and ``omegaconf`` (hcpdiff/utils/utils.py:7).
Nothing is copied: the reference files are executed where they lie.  Only available where /root/reference exists
(NOT on the GPU box) — used to generate tests/golden/*.pt and to pin oracle/lora_ref.py.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("HCP_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "hcpdiff", "models"))


def _stub_pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def load_reference_lora():
    """Returns (lora_layers_patch module, plugin module) of the reference, executed from REFERENCE_ROOT."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if "hcpdiff.models.lora_layers_patch" in sys.modules:
        return sys.modules["hcpdiff.models.lora_layers_patch"], sys.modules["hcpdiff.models.plugin"]
    if "diffusers" not in sys.modules:
        d = types.ModuleType("diffusers"); d.__path__ = []
        sys.modules["diffusers"] = d
    if "diffusers.optimization" not in sys.modules:       # (whichever loader ran first may have made the bare package stub)
        opt = types.ModuleType("diffusers.optimization")
        opt.SchedulerType = type("SchedulerType", (), {})
        opt.TYPE_TO_SCHEDULER_FUNCTION = {}
        opt.Optimizer = object
        sys.modules["diffusers.optimization"] = opt
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        oc.OmegaConf = type("OmegaConf", (), {}); oc.ListConfig = list
        sys.modules["omegaconf"] = oc
    base = os.path.join(REFERENCE_ROOT, "hcpdiff")
    _stub_pkg("hcpdiff", base)
    utils = _stub_pkg("hcpdiff.utils", os.path.join(base, "utils"))
    _stub_pkg("hcpdiff.models", os.path.join(base, "models"))
    net_utils = importlib.import_module("hcpdiff.utils.net_utils")
    u = importlib.import_module("hcpdiff.utils.utils")
    for k in dir(u):                      # `from hcpdiff.utils import ...` style imports in the reference
        if not k.startswith("_"):
            setattr(utils, k, getattr(u, k))
    utils.net_utils = net_utils
    plugin = importlib.import_module("hcpdiff.models.plugin")
    importlib.import_module("hcpdiff.models.lora_base_patch")
    layers = importlib.import_module("hcpdiff.models.lora_layers_patch")
    return layers, plugin


def load_reference_loss():
    """The reference's hcpdiff/loss/min_snr_loss.py, executed where it lies (its one third-party import,
    ``from diffusers import SchedulerMixin``, is an annotation only -> an empty shim class)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    import importlib.util
    d = sys.modules.get("diffusers")
    if d is None:
        d = types.ModuleType("diffusers"); d.__path__ = []
        sys.modules["diffusers"] = d
    if not hasattr(d, "SchedulerMixin"):
        d.SchedulerMixin = type("SchedulerMixin", (), {})
    spec = importlib.util.spec_from_file_location("_hcp_ref_min_snr_loss", os.path.join(REFERENCE_ROOT, "hcpdiff", "loss", "min_snr_loss.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_ckpt():
    """(CkptManagerSafe class, cfg_net_tools module) of the reference, executed from REFERENCE_ROOT: the checkpoint writer
    (ckpt_manager/ckpt_safetensor.py, ckpt_pkl.py) and the loader (utils/cfg_net_tools.py HCPModelLoader).  Extra shims:
    ``diffusers.StableDiffusionPipeline`` / ``diffusers.models.lora.LoRACompatibleLinear`` (annotations in
    ckpt_manager/base.py:1-2) and the names hcpdiff/models/__init__.py would have re-exported."""
    layers, plugin = load_reference_lora()
    d = sys.modules["diffusers"]
    if not hasattr(d, "StableDiffusionPipeline"):
        d.StableDiffusionPipeline = type("StableDiffusionPipeline", (), {})
        dm = types.ModuleType("diffusers.models"); dm.__path__ = []
        dml = types.ModuleType("diffusers.models.lora")
        dml.LoRACompatibleLinear = type("LoRACompatibleLinear", (), {})
        sys.modules["diffusers.models"] = dm; sys.modules["diffusers.models.lora"] = dml
    models = sys.modules["hcpdiff.models"]
    patch = sys.modules["hcpdiff.models.lora_base_patch"]
    models.LoraBlock, models.LoraGroup, models.lora_layer_map = patch.LoraBlock, patch.LoraGroup, layers.lora_layer_map
    _stub_pkg("hcpdiff.tools", os.path.join(REFERENCE_ROOT, "hcpdiff", "tools"))
    ckpt = importlib.import_module("hcpdiff.ckpt_manager")          # its real 4-line __init__ runs
    tools = importlib.import_module("hcpdiff.utils.cfg_net_tools")
    return ckpt.CkptManagerSafe, tools


def load_reference_trainer():
    """(train_ac module, train_ac_single module) of the reference, executed where they lie, with every dependency that the
    inner loop does NOT touch replaced by an empty stand-in: diffusers / hydra / loguru (not installable here) and the
    reference's own data, logger, visualizer and config-converter packages (host-side I/O, out of scope of the hot path).
    What stays REAL: hcpdiff/train_ac.py itself (Trainer.train_one_step / forward / make_noise / get_loss / get_latents,
    train_ac.py:428-515), hcpdiff/train_ac_single.py (TrainerSingleCard.init_context builds a real accelerate.Accelerator),
    hcpdiff/models/wrapper.py (TEUnetWrapper), hcpdiff/models/cfg_context.py, hcpdiff/utils/cfg_net_tools.py (make_hcpdiff)."""
    load_reference_ckpt()
    d = sys.modules["diffusers"]
    for name in ("AutoencoderKL", "UNet2DConditionModel", "DDPMScheduler", "SchedulerMixin"):
        if not hasattr(d, name):
            setattr(d, name, type(name, (), {}))
    if "diffusers.utils" not in sys.modules:
        du = types.ModuleType("diffusers.utils"); du.__path__ = []
        dui = types.ModuleType("diffusers.utils.import_utils"); dui.is_xformers_available = lambda: False
        sys.modules["diffusers.utils"] = du; sys.modules["diffusers.utils.import_utils"] = dui
        d.utils = du; du.import_utils = dui

    def stub(modname, **attrs):
        if modname not in sys.modules:
            m = types.ModuleType(modname); m.__path__ = []
            sys.modules[modname] = m
        for k, v in attrs.items():
            setattr(sys.modules[modname], k, v)
        return sys.modules[modname]

    def placeholder(name):
        return type(name, (), {})

    stub("hydra", utils=stub("hydra.utils", instantiate=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("hydra is shimmed"))))
    if "loguru" not in sys.modules:
        import logging
        stub("loguru", logger=logging.getLogger("hcpdiff-shim"))
    sys.modules["omegaconf"].OmegaConf = getattr(sys.modules["omegaconf"], "OmegaConf", placeholder("OmegaConf"))
    stub("hcpdiff.data", RatioBucket=placeholder("RatioBucket"), DataGroup=placeholder("DataGroup"), get_sampler=lambda *a, **k: None)
    stub("hcpdiff.deprecated"); stub("hcpdiff.deprecated.cfg_converter", TrainCFGConverter=placeholder("TrainCFGConverter"))
    stub("hcpdiff.loggers", LoggerGroup=placeholder("LoggerGroup"))
    stub("hcpdiff.visualizer", Visualizer=placeholder("Visualizer"))
    stub("hcpdiff.models.compose", ComposeEmbPTHook=placeholder("ComposeEmbPTHook"), ComposeTEEXHook=placeholder("ComposeTEEXHook"),
         SDXLTextEncoder=placeholder("SDXLTextEncoder"))
    models = sys.modules["hcpdiff.models"]
    ctx = importlib.import_module("hcpdiff.models.cfg_context")
    wrapper = importlib.import_module("hcpdiff.models.wrapper")
    models.CFGContext, models.DreamArtistPTContext = ctx.CFGContext, ctx.DreamArtistPTContext
    models.TEUnetWrapper, models.SDXLTEUnetWrapper = wrapper.TEUnetWrapper, wrapper.SDXLTEUnetWrapper
    train_ac = importlib.import_module("hcpdiff.train_ac")
    single = importlib.import_module("hcpdiff.train_ac_single")
    return train_ac, single


class ShimDDIMScheduler:
    """Stand-in for the diffusers scheduler object the reference's pipeline drives (pipe_hook.py:85-87,121-122,139-140): Stable
    Diffusion's DDIM configuration (prediction_type 'epsilon', eta 0, clip_sample False, set_alpha_to_one False, timestep_spacing
    'leading', steps_offset 1) [ext] — the arithmetic of oracle/sampler_ref.cfg_ddim_step behind diffusers' method names."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000):
        import torch
        from oracle.unet_sd15 import ddpm_alphas_cumprod
        self.alphas_cumprod = ddpm_alphas_cumprod(num_train_timesteps)
        self.T = num_train_timesteps
        self.timesteps = torch.zeros(0, dtype=torch.long)

    def set_timesteps(self, num_inference_steps, device=None):
        import torch
        self.ratio = self.T // num_inference_steps
        self.timesteps = ((torch.arange(num_inference_steps) * self.ratio).flip(0) + 1).clamp(max=self.T - 1).to(device)

    def scale_model_input(self, sample, t):
        return sample

    def step(self, noise_pred, t, latents, **kwargs):
        import types
        t = int(t)
        prev = t - self.ratio
        a_t = float(self.alphas_cumprod[t]); a_prev = float(self.alphas_cumprod[prev] if prev >= 0 else self.alphas_cumprod[0])
        x0 = (latents - (1 - a_t) ** 0.5 * noise_pred) / a_t ** 0.5
        return types.SimpleNamespace(prev_sample=a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * noise_pred)


def load_reference_pipe():
    """hcpdiff/utils/pipe_hook.py executed where it lies: HookPipe_T2I.__call__ (the denoising loop :116-140 with the CFG-doubled batch,
    ``encoder_attention_mask``, SDXL ``added_cond_kwargs``) is the reference's own code.  Its diffusers base class is replaced by a
    stand-in that provides exactly the helpers the loop calls (prepare_latents, prepare_extra_step_kwargs, progress_bar, the image
    processor's postprocess); the inpainting pipeline it imports next to it is stubbed (out of scope)."""
    import contextlib
    import torch
    load_reference_lora()
    d = sys.modules["diffusers"]

    class _Pipe:                                           # StableDiffusionPipeline's surface as HookPipe_T2I.__call__ uses it
        vae_scale_factor = 8

        def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, scheduler=None, **kw):
            self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
            self.image_processor = types.SimpleNamespace(postprocess=lambda image, output_type=None, do_denormalize=None: image)

        def prepare_extra_step_kwargs(self, generator, eta):
            return {}

        def prepare_latents(self, batch_size, channels, height, width, dtype, device, generator, latents=None):
            if latents is None:
                shape = (batch_size, channels, height // 8, width // 8)
                if isinstance(generator, (list, tuple)):   # diffusers' randn_tensor: one generator per sample (the previewer's seeds)
                    latents = torch.cat([torch.randn((1,) + shape[1:], generator=g_, dtype=dtype) for g_ in generator])
                else:
                    latents = torch.randn(shape, generator=generator, dtype=dtype)
            return latents.to(device) * self.scheduler.init_noise_sigma

        @contextlib.contextmanager
        def progress_bar(self, total=None):
            yield types.SimpleNamespace(update=lambda *a: None)

    d.StableDiffusionPipeline = _Pipe
    d.StableDiffusionImg2ImgPipeline = type("StableDiffusionImg2ImgPipeline", (_Pipe,), {})
    for modname, attrs in (("diffusers.image_processor", dict(VaeImageProcessor=type("VaeImageProcessor", (), {}))),
                           ("diffusers.pipelines", {}),
                           ("diffusers.pipelines.stable_diffusion", dict(StableDiffusionPipelineOutput=lambda images=None, nsfw_content_detected=None:
                                                                      types.SimpleNamespace(images=images)))):
        if modname not in sys.modules:
            m = types.ModuleType(modname); m.__path__ = []
            sys.modules[modname] = m
        for k, v in attrs.items():
            setattr(sys.modules[modname], k, v)
    if "PIL" not in sys.modules:
        import PIL  # noqa: F401
    ip = types.ModuleType("hcpdiff.utils.inpaint_pipe")     # inpainting: out of scope; names only
    ip.preprocess_mask = ip.preprocess_image = lambda *a, **k: None
    ip.StableDiffusionInpaintPipelineLegacy = type("StableDiffusionInpaintPipelineLegacy", (_Pipe,), {})
    sys.modules["hcpdiff.utils.inpaint_pipe"] = ip
    return importlib.import_module("hcpdiff.utils.pipe_hook")


def load_reference_previewer():
    """(image_previewer module, pipe_hook module, TEEXHook class, TokenizerHook class) of the reference, executed where they lie:
    ``ImagePreviewer.preview`` / ``vis_images`` (loggers/preview/image_previewer.py:97-149) and everything they call that is the
    reference's own — HookPipe_T2I.__call__ (utils/pipe_hook.py), TokenizerHook.parse_attn_mult (models/tokenizer_ex.py),
    TEEXHook.encode_prompt_to_emb / mult_attn (models/textencoder_ex.py).  Replaced by stand-ins: hydra (config instantiation, not
    installable here), the diffusers pipeline base class (load_reference_pipe), ``Visualizer`` (hcpdiff/visualizer.py drags in the
    model loaders, compose hooks and the config converter; the two helpers the previewer inherits from it and uses, get_ex_input()
    and inter_callback, are restated for the no-condition / no-interface case), and ``prepare_seed`` where no CUDA device exists
    (utils/utils.py:135 builds torch.Generator(device='cuda'))."""
    import random
    import torch
    pipe_hook = load_reference_pipe()
    d = sys.modules["diffusers"]
    if not hasattr(d, "PNDMScheduler"):
        d.PNDMScheduler = type("PNDMScheduler", (), {"__init__": lambda self, **kw: None})
    if "hydra" not in sys.modules:
        h = types.ModuleType("hydra"); h.__path__ = []
        hu = types.ModuleType("hydra.utils"); hu.instantiate = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("hydra is shimmed"))
        h.utils = hu
        sys.modules["hydra"] = h; sys.modules["hydra.utils"] = hu
    models = sys.modules["hcpdiff.models"]
    tok = importlib.import_module("hcpdiff.models.tokenizer_ex")
    models.TokenizerHook = tok.TokenizerHook
    teex = importlib.import_module("hcpdiff.models.textencoder_ex")

    class Visualizer:                                      # hcpdiff/visualizer.py:22,171-183,221-231 for condition None, no ex_input, no interface
        dtype_dict = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}

        def get_pipeline(self):
            return pipe_hook.HookPipe_T2I

        def get_ex_input(self):
            assert getattr(self.cfgs, "condition", None) is None and getattr(self.cfgs, "ex_input", None) is None
            return {}, {}

        def inter_callback(self, i, t, num_t, latents_x0, latents):
            return latents

    v = types.ModuleType("hcpdiff.visualizer"); v.Visualizer = Visualizer
    sys.modules["hcpdiff.visualizer"] = v
    _stub_pkg("hcpdiff.loggers", os.path.join(REFERENCE_ROOT, "hcpdiff", "loggers"))
    _stub_pkg("hcpdiff.loggers.preview", os.path.join(REFERENCE_ROOT, "hcpdiff", "loggers", "preview"))
    prev = importlib.import_module("hcpdiff.loggers.preview.image_previewer")
    if not torch.cuda.is_available():
        prev.prepare_seed = lambda seeds, device="cpu": [torch.Generator().manual_seed(s or random.randint(0, 1 << 30)) for s in seeds]
    return prev, pipe_hook, teex.TEEXHook, tok.TokenizerHook
