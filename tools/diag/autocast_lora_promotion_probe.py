import sys, torch, copy
sys.path.insert(0,'/root/repo')
from oracle.unet_sd15 import SDXL_CONFIG, TINY_SDXL_CONFIG, OracleUNet2DConditionModel, seeded_init_, add_noise, ddpm_alphas_cumprod
from oracle.lora_ref import wrap_lora, OracleLoraLinear
from oracle.make_golden import sd15_lora_init_
torch.manual_seed(0)
cfg = dict(TINY_SDXL_CONFIG, transformer_layers_per_block=(1, 2, 6))
def build(lora):
    m = seeded_init_(OracleUNet2DConditionModel(**cfg), 1); m.requires_grad_(False)
    if lora:
        wrap_lora(m, [r"re:.*\.attn.?$", r"re:.*\.ff$"], rank=16)
        sd15_lora_init_([(n, p) for n, p in m.named_parameters() if "lora_block_" in n])
    return m
g2 = torch.Generator().manual_seed(1)
x0 = torch.randn(2,4,32,32,generator=g2); ehs = torch.randn(2,77,64,generator=g2); t=torch.tensor([91,707])
added = dict(text_embeds=torch.randn(2,64,generator=g2), time_ids=torch.tensor([[256.0,256.0,0,0,256.0,256.0]]*2))
def run(m, ac):
    with torch.no_grad():
        if ac:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                return m(x0,t,ehs,added_cond_kwargs=added).sample.float()
        return m(x0,t,ehs,added_cond_kwargs=added).sample
rel = lambda a,b: ((a-b).norm()/b.norm()).item()
for lora in (False, True):
    m = build(lora)
    ref = run(m, False); ac = run(m, True)
    print("lora" if lora else "plain", "autocast rel-L2", rel(ac, ref))
    if lora:
        # variant: LoRA layers round their output to bf16 like nn.Linear under autocast (kills the fp32 promotion by the bias add)
        fwd = OracleLoraLinear.forward
        def fwd_bf16(self, x):
            y = fwd(self, x)
            return y.to(torch.bfloat16) if torch.is_autocast_enabled("cpu") else y
        OracleLoraLinear.forward = fwd_bf16
        print("lora, outputs forced to bf16 under autocast:", rel(run(m, True), ref))
        OracleLoraLinear.forward = fwd
