"""CLIP text encoder (the `text_encoder` of runwayml/stable-diffusion-v1-5: CLIPTextModel, ViT-L/14 text tower) on
libcustomnerf_hip.so — what StableDiffusion.get_text_embeds calls (nerf/sd.py:77-94: `self.text_encoder(input_ids)[0]`).

12 pre-LN transformer layers (width 768, 12 heads of 64, quick-GELU MLP 3072, causal attention), learned token + position
embeddings, final LayerNorm; returns last_hidden_state [B, 77, 768].  Parameter names are transformers' CLIPTextModel state-dict
keys.  Tokenisation (BPE vocabulary files) is host-side plumbing and stays outside: this module takes token ids."""
import torch

from . import ops, pack

CLIP_TEXT_SD15 = dict(vocab_size=49408, width=768, layers=12, heads=12, mlp=3072, max_position=77, eps=1e-5)
CLIP_TEXT_TINY = dict(vocab_size=1000, width=128, layers=2, heads=2, mlp=512, max_position=77, eps=1e-5)


def clip_text_params(cfg):
    w, m = cfg["width"], cfg["mlp"]
    out = [("text_model.embeddings.token_embedding.weight", (cfg["vocab_size"], w)), ("text_model.embeddings.position_embedding.weight", (cfg["max_position"], w))]
    for i in range(cfg["layers"]):
        p = f"text_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out += [(p + f"self_attn.{n}.weight", (w, w)), (p + f"self_attn.{n}.bias", (w,))]
        out += [(p + "layer_norm1.weight", (w,)), (p + "layer_norm1.bias", (w,)), (p + "layer_norm2.weight", (w,)), (p + "layer_norm2.bias", (w,)),
                (p + "mlp.fc1.weight", (m, w)), (p + "mlp.fc1.bias", (m,)), (p + "mlp.fc2.weight", (w, m)), (p + "mlp.fc2.bias", (w,))]
    out += [("text_model.final_layer_norm.weight", (w,)), ("text_model.final_layer_norm.bias", (w,))]
    return out


class CLIPTextEncoder:
    def __init__(self, cfg, state_dict, device="cuda"):
        self.cfg = cfg
        dev = torch.device(device)
        g = lambda k: state_dict[k].to(dev)
        self.tok = g("text_model.embeddings.token_embedding.weight").to(torch.float16)
        self.pos = g("text_model.embeddings.position_embedding.weight").to(torch.float16)
        self.layers = []
        for i in range(cfg["layers"]):
            p = f"text_model.encoder.layers.{i}."
            qkv_w = pack.pack_linear(torch.cat([g(p + f"self_attn.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")], 0))
            qkv_b = pack.f32(torch.cat([g(p + f"self_attn.{n}.bias") for n in ("q_proj", "k_proj", "v_proj")], 0))
            self.layers.append(dict(
                ln1=(pack.f32(g(p + "layer_norm1.weight")), pack.f32(g(p + "layer_norm1.bias"))), ln2=(pack.f32(g(p + "layer_norm2.weight")), pack.f32(g(p + "layer_norm2.bias"))),
                qkv=(qkv_w, qkv_b), out=(pack.pack_linear(g(p + "self_attn.out_proj.weight")), pack.f32(g(p + "self_attn.out_proj.bias"))),
                fc1=(pack.pack_linear(g(p + "mlp.fc1.weight")), pack.f32(g(p + "mlp.fc1.bias"))), fc2=(pack.pack_linear(g(p + "mlp.fc2.weight")), pack.f32(g(p + "mlp.fc2.bias")))))
        self.lnf = (pack.f32(g("text_model.final_layer_norm.weight")), pack.f32(g("text_model.final_layer_norm.bias")))

    def add_token_embedding(self, vector):
        """Textual inversion (`pipe.load_textual_inversion(model_id, weight_name="<new1>.bin")`, sd.py:58): appends one learned
        embedding row and returns its token id (the tokenizer maps the placeholder string to it)."""
        v = vector.reshape(1, -1).to(self.tok.device, torch.float16)
        assert v.shape[1] == self.cfg["width"]
        self.tok = torch.cat([self.tok, v], 0)
        return self.tok.shape[0] - 1

    def __call__(self, input_ids):
        """input_ids [B, T<=77] int64 on device -> (last_hidden_state [B, T, width] half,)   (tuple: callers index [0], sd.py:85)"""
        cfg = self.cfg
        B, T = input_ids.shape
        W = cfg["width"]
        # embedding lookup + position add: index plumbing on a [B, 77, 768] tensor (torch), everything after it is the library
        h = (self.tok[input_ids] + self.pos[:T][None]).contiguous()
        for L in self.layers:
            n = ops.layernorm(h, *L["ln1"], cfg["eps"])
            qkv = ops.linear(n, L["qkv"][0], bias=L["qkv"][1])                       # [B, T, 3W]
            a = ops.attention(qkv[..., :W], qkv[..., W:2 * W], qkv[..., 2 * W:], cfg["heads"], causal=True)
            h = ops.linear(a, L["out"][0], bias=L["out"][1], residual=h)
            n = ops.layernorm(h, *L["ln2"], cfg["eps"])
            f = ops.linear(n, L["fc1"][0], bias=L["fc1"][1], act=ops.ACT_QUICK_GELU)
            h = ops.linear(f, L["fc2"][0], bias=L["fc2"][1], residual=h)
        return (ops.layernorm(h, *self.lnf, cfg["eps"]),)
