"""CLIP view classifier of the editing loop (`--clip_view`): counterpart of nerf/clip.py (class CLIP: `clip.load("ViT-B/32")`,
`transformCLIP`, `get_text_embeds`, `encode_img`) and of its use in Trainer_Nerf.get_pt / get_textz / prepare_text_embeddings
(nerf/utils_init_nerf.py:254-258, 268-279, 341-351): the cached pretrained render of a view is matched against the three prompts
"front / side / back face of an object" and the arg-max picks the direction-suffixed SD prompt for that view.

Both towers of OpenAI CLIP run on libcustomnerf_hip.so (patch-embedding and every projection through cnerf_sd_gemm, LayerNorm,
fused attention, quick-GELU epilogue).  Parameter names are the OpenAI `clip` package's state-dict keys (`visual.conv1.weight`,
`visual.transformer.resblocks.N.attn.in_proj_weight`, `token_embedding.weight`, `text_projection`, `logit_scale`, ...), so the
`ViT-B-32.pt` state dict loads as is.  Tokenisation (`clip.tokenize`: BPE vocabulary file) is host-side plumbing and stays outside:
token ids are passed in, exactly as for the SD text encoder."""
import math

import torch

from . import ops, pack

CLIP_VITB32 = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32, vision_heads=12,
                   context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12, eps=1e-5)
CLIP_TINY = dict(embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=16, vision_heads=2,
                 context_length=77, vocab_size=1000, transformer_width=128, transformer_heads=2, transformer_layers=2, eps=1e-5)

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
VIEW_PROMPTS = ["front face of an object", "side face of an object", "back face of an object"]     # utils_init_nerf.py:344
VIEW_SUFFIXES = ["front", "side", "back"]                                                           # utils_init_nerf.py:322


def _block_params(prefix, w):
    return [(prefix + "attn.in_proj_weight", (3 * w, w)), (prefix + "attn.in_proj_bias", (3 * w,)), (prefix + "attn.out_proj.weight", (w, w)),
            (prefix + "attn.out_proj.bias", (w,)), (prefix + "ln_1.weight", (w,)), (prefix + "ln_1.bias", (w,)), (prefix + "mlp.c_fc.weight", (4 * w, w)),
            (prefix + "mlp.c_fc.bias", (4 * w,)), (prefix + "mlp.c_proj.weight", (w, 4 * w)), (prefix + "mlp.c_proj.bias", (w,)),
            (prefix + "ln_2.weight", (w,)), (prefix + "ln_2.bias", (w,))]


def clip_params(cfg):
    """(name, shape) table of the OpenAI CLIP state dict (ViT image tower + text transformer); ViT-B/32 = 151,277,313 parameters."""
    vw, tw, p = cfg["vision_width"], cfg["transformer_width"], cfg["vision_patch_size"]
    n_tok = (cfg["image_resolution"] // p) ** 2 + 1
    out = [("positional_embedding", (cfg["context_length"], tw)), ("text_projection", (tw, cfg["embed_dim"])), ("logit_scale", ()),
           ("visual.class_embedding", (vw,)), ("visual.positional_embedding", (n_tok, vw)), ("visual.proj", (vw, cfg["embed_dim"])),
           ("visual.conv1.weight", (vw, 3, p, p)), ("visual.ln_pre.weight", (vw,)), ("visual.ln_pre.bias", (vw,))]
    for i in range(cfg["vision_layers"]):
        out += _block_params(f"visual.transformer.resblocks.{i}.", vw)
    out += [("visual.ln_post.weight", (vw,)), ("visual.ln_post.bias", (vw,))]
    for i in range(cfg["transformer_layers"]):
        out += _block_params(f"transformer.resblocks.{i}.", tw)
    out += [("token_embedding.weight", (cfg["vocab_size"], tw)), ("ln_final.weight", (tw,)), ("ln_final.bias", (tw,))]
    return out


def random_clip_state_dict(cfg, seed=0):
    """Seeded random weights of the CLIP shapes (no checkpoint offline), scaled like the OpenAI initialisation."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in clip_params(cfg):
        if name == "logit_scale":
            sd[name] = torch.tensor(math.log(1.0 / 0.07))
        elif name.endswith("ln_1.weight") or name.endswith("ln_2.weight") or name.endswith("ln_pre.weight") or name.endswith("ln_post.weight") or name == "ln_final.weight":
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            sd[name] = 0.02 * torch.randn(shape, generator=g)
        elif name == "visual.conv1.weight":
            sd[name] = torch.randn(shape, generator=g) / math.sqrt(shape[1] * shape[2] * shape[3])
        elif len(shape) == 2 and "embedding" not in name:
            fan_in = shape[0] if name in ("text_projection", "visual.proj") else shape[1]
            sd[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        else:
            sd[name] = 0.05 * torch.randn(shape, generator=g)
    return sd


class _Tower:
    """pre-LN residual attention blocks (OpenAI ResidualAttentionBlock: nn.MultiheadAttention + c_fc / QuickGELU / c_proj)"""

    def __init__(self, g, prefix, layers, width, heads, eps):
        self.width, self.heads, self.eps = width, heads, eps
        self.layers = []
        for i in range(layers):
            p = f"{prefix}resblocks.{i}."
            self.layers.append(dict(
                ln1=(pack.f32(g(p + "ln_1.weight")), pack.f32(g(p + "ln_1.bias"))), ln2=(pack.f32(g(p + "ln_2.weight")), pack.f32(g(p + "ln_2.bias"))),
                qkv=(pack.pack_linear(g(p + "attn.in_proj_weight")), pack.f32(g(p + "attn.in_proj_bias"))),
                out=(pack.pack_linear(g(p + "attn.out_proj.weight")), pack.f32(g(p + "attn.out_proj.bias"))),
                fc=(pack.pack_linear(g(p + "mlp.c_fc.weight")), pack.f32(g(p + "mlp.c_fc.bias"))),
                proj=(pack.pack_linear(g(p + "mlp.c_proj.weight")), pack.f32(g(p + "mlp.c_proj.bias")))))

    def __call__(self, h, causal):
        W = self.width
        for L in self.layers:
            n = ops.layernorm(h, *L["ln1"], self.eps)
            qkv = ops.linear(n, L["qkv"][0], bias=L["qkv"][1])
            a = ops.attention(qkv[..., :W], qkv[..., W:2 * W], qkv[..., 2 * W:], self.heads, causal=causal)
            h = ops.linear(a, L["out"][0], bias=L["out"][1], residual=h)
            n = ops.layernorm(h, *L["ln2"], self.eps)
            f = ops.linear(n, L["fc"][0], bias=L["fc"][1], act=ops.ACT_QUICK_GELU)
            h = ops.linear(f, L["proj"][0], bias=L["proj"][1], residual=h)
        return h


class CLIPModel:
    """OpenAI `clip.model.CLIP` forward surface: encode_image, encode_text, __call__(image, text) -> (logits_per_image, logits_per_text)."""

    def __init__(self, cfg, state_dict, device="cuda"):
        self.cfg = cfg
        dev = torch.device(device)
        g = lambda k: state_dict[k].to(dev)
        self.device = dev
        w = g("visual.conv1.weight")                                                                # [width, 3, P, P] -> [width, (kh, kw, c)] = ops.patchify's row order
        self.patch_w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(torch.float16).contiguous()
        self.class_embedding = g("visual.class_embedding").to(torch.float16)
        self.vis_pos = g("visual.positional_embedding").to(torch.float16)
        self.ln_pre = (pack.f32(g("visual.ln_pre.weight")), pack.f32(g("visual.ln_pre.bias")))
        self.ln_post = (pack.f32(g("visual.ln_post.weight")), pack.f32(g("visual.ln_post.bias")))
        self.vis_proj = pack.pack_linear_T(g("visual.proj"))                                        # x @ proj  ==  linear(x, proj^T)
        self.visual = _Tower(g, "visual.transformer.", cfg["vision_layers"], cfg["vision_width"], cfg["vision_heads"], cfg["eps"])
        self.tok = g("token_embedding.weight").to(torch.float16)
        self.txt_pos = g("positional_embedding").to(torch.float16)
        self.ln_final = (pack.f32(g("ln_final.weight")), pack.f32(g("ln_final.bias")))
        self.text_projection = pack.pack_linear_T(g("text_projection"))
        self.transformer = _Tower(g, "transformer.", cfg["transformer_layers"], cfg["transformer_width"], cfg["transformer_heads"], cfg["eps"])
        self.logit_scale = g("logit_scale").float()

    def parameters(self):
        return iter(())                                            # frozen: the reference only sets requires_grad=False on them (utils_init_nerf.py:170)

    @torch.no_grad()
    def encode_image(self, image):
        """image [B, 3, R, R] float32, already transformCLIP-ed -> [B, embed_dim] half (clip/model.py VisionTransformer.forward)."""
        cfg = self.cfg
        B = image.shape[0]
        assert image.shape[2] == image.shape[3] == cfg["image_resolution"], "CLIP image tower expects the transformCLIP output"
        x = ops.linear(ops.patchify(image, cfg["vision_patch_size"]), self.patch_w)                 # conv1 (kernel = stride = patch, no bias) as one GEMM
        h = torch.cat([self.class_embedding.expand(B, 1, -1), x], 1) + self.vis_pos[None]
        h = ops.layernorm(h.contiguous(), *self.ln_pre, cfg["eps"])
        h = self.visual(h, causal=False)
        cls = ops.layernorm(h[:, 0, :].contiguous(), *self.ln_post, cfg["eps"])
        return ops.linear(cls, self.vis_proj)

    @torch.no_grad()
    def encode_text(self, text):
        """text [n, 77] int64 token ids (clip.tokenize) -> [n, embed_dim] half: features at the EOT token (highest id) @ text_projection."""
        cfg = self.cfg
        n, T = text.shape
        h = (self.tok[text] + self.txt_pos[:T][None]).contiguous()
        h = self.transformer(h, causal=True)
        h = ops.layernorm(h, *self.ln_final, cfg["eps"])
        eot = h[torch.arange(n, device=h.device), text.argmax(dim=-1)].contiguous()
        return ops.linear(eot, self.text_projection)

    @torch.no_grad()
    def __call__(self, image, text):
        img, txt = self.encode_image(image).float(), self.encode_text(text).float()
        img = img / img.norm(dim=1, keepdim=True)
        txt = txt / txt.norm(dim=1, keepdim=True)
        logits_per_image = self.logit_scale.exp() * img @ txt.t()
        return logits_per_image, logits_per_image.t()


class CLIP:
    """nerf/clip.py:6-28 — `.model`, `.transformCLIP`, `get_text_embeds`, `encode_img`.  `tokenizer(list[str]) -> [n, 77] int64`
    is injected (clip.tokenize needs the BPE vocabulary file); without one, token-id tensors are passed directly."""

    def __init__(self, device, cfg=None, state_dict=None, tokenizer=None, seed=0):
        self.device = torch.device(device)
        self.cfg = cfg or CLIP_VITB32
        if state_dict is None:
            state_dict = random_clip_state_dict(self.cfg, seed)      # shapes of ViT-B/32, seeded random weights (no checkpoint offline)
        self.model = CLIPModel(self.cfg, state_dict, self.device)
        self.tokenizer = tokenizer

    def parameters(self):
        return self.model.parameters()

    def transformCLIP(self, img):
        return ops.clip_preprocess(img, self.cfg["image_resolution"], CLIP_MEAN, CLIP_STD)

    def tokenize(self, text):
        if torch.is_tensor(text):
            return text.to(self.device)
        if self.tokenizer is None:
            raise RuntimeError("CLIP.tokenize: no tokenizer injected (clip.tokenize's BPE vocabulary is not available offline); pass token ids")
        return self.tokenizer(text).to(self.device)

    def get_text_embeds(self, text):
        return self.model.encode_text(self.tokenize(text)).float()

    def encode_img(self, img):
        return self.model.encode_image(self.transformCLIP(img)).float()

    def match_view(self, img, match_text):
        """utils_init_nerf.py:255-258: softmax over the view prompts of the image-text logits -> [B, n_prompts]"""
        logits_per_image, _ = self.model(self.transformCLIP(img), match_text)
        return logits_per_image.softmax(dim=1)
