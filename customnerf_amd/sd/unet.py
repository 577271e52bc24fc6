"""UNet2DConditionModel (SD-1.5 configuration) forward as a graph of libcustomnerf_hip.so primitives.

Replaces `self.unet(latent_model_input, t, encoder_hidden_states=text_embeddings).sample` (nerf/sd.py:140): eps-prediction,
inference only.  Module tree and parameter names are diffusers' (arch.unet_params); activations are NHWC float16, so every
convolution / projection is one implicit-GEMM launch with its bias, time-embedding bias and residual fused into the epilogue,
GroupNorm+SiLU is one fused pair of launches, the nearest-2x upsampling is folded into the following conv's loader.
The whole forward issues no host synchronisation and allocates only through torch's caching allocator, so it can be
captured into a HIP graph (`UNet.graphed`)."""
import torch

from . import ops, pack
from .arch import unet_up_channels


class _Resnet:
    def __init__(self, sd, p, dev):
        g = lambda k: sd[p + k].to(dev)
        self.n1w, self.n1b = pack.f32(g("norm1.weight")), pack.f32(g("norm1.bias"))
        self.n2w, self.n2b = pack.f32(g("norm2.weight")), pack.f32(g("norm2.bias"))
        self.c1w, self.c1b = pack.pack_conv(g("conv1.weight")), pack.f32(g("conv1.bias"))
        self.c2w, self.c2b = pack.pack_conv(g("conv2.weight")), pack.f32(g("conv2.bias"))
        self.has_temb = (p + "time_emb_proj.weight") in sd
        if self.has_temb:
            self.tw, self.tb = pack.pack_linear(g("time_emb_proj.weight")), pack.f32(g("time_emb_proj.bias"))
        self.has_sc = (p + "conv_shortcut.weight") in sd
        if self.has_sc:
            self.sw, self.sb = pack.pack_conv(g("conv_shortcut.weight")), pack.f32(g("conv_shortcut.bias"))

    def __call__(self, x, tb, groups, eps, pool, x_sums=None, out_gn=False):
        """tb: this block's time_emb_proj(silu(temb)) [B, Cout] float32 (all blocks' projections are one batched GEMM in UNet.forward).
        GroupNorm statistics ride on the producing GEMM's epilogue where there is one: x_sums = (sums, ready) for norm1's input, and with
        out_gn the block returns the same for its own output (for the norm that consumes it next).  -> (y, y_sums | None)"""
        hw = x.shape[1] * x.shape[2]
        if x_sums is not None:
            h, _ = ops.groupnorm(x, self.n1w, self.n1b, groups, eps, True, sums=x_sums[0], sums_ready=x_sums[1])
        else:
            h, _ = ops.groupnorm(x, self.n1w, self.n1b, groups, eps, True, pool)
        s2 = pool.take()
        h, ok = ops.conv2d(h, self.c1w, self.c1b, 3, bias_rows=tb, gn=(s2, groups, hw))
        h, _ = ops.groupnorm(h, self.n2w, self.n2b, groups, eps, True, sums=s2, sums_ready=ok)
        sc = ops.conv2d(x, self.sw, self.sb, 1, pad=0) if self.has_sc else x
        if not out_gn:
            return ops.conv2d(h, self.c2w, self.c2b, 3, residual=sc), None
        so = pool.take()
        y, ok = ops.conv2d(h, self.c2w, self.c2b, 3, residual=sc, gn=(so, groups, hw))
        return y, (so, ok)


class _Transformer:
    def __init__(self, sd, p, dev, heads):
        g = lambda k: sd[p + k].to(dev)
        t = "transformer_blocks.0."
        self.heads = heads
        self.nw, self.nb = pack.f32(g("norm.weight")), pack.f32(g("norm.bias"))
        self.piw, self.pib = pack.pack_conv(g("proj_in.weight")), pack.f32(g("proj_in.bias"))
        self.pow, self.pob = pack.pack_conv(g("proj_out.weight")), pack.f32(g("proj_out.bias"))
        self.ln = [(pack.f32(g(t + f"norm{i}.weight")), pack.f32(g(t + f"norm{i}.bias"))) for i in (1, 2, 3)]
        # self-attention: q, k, v projections share their input -> one GEMM with the three weight matrices stacked
        self.qkv1 = pack.pack_linear(torch.cat([g(t + "attn1.to_q.weight"), g(t + "attn1.to_k.weight"), g(t + "attn1.to_v.weight")], 0))
        self.o1w, self.o1b = pack.pack_linear(g(t + "attn1.to_out.0.weight")), pack.f32(g(t + "attn1.to_out.0.bias"))
        self.q2 = pack.pack_linear(g(t + "attn2.to_q.weight"))
        self.kv2 = pack.pack_linear(torch.cat([g(t + "attn2.to_k.weight"), g(t + "attn2.to_v.weight")], 0))
        self.o2w, self.o2b = pack.pack_linear(g(t + "attn2.to_out.0.weight")), pack.f32(g(t + "attn2.to_out.0.bias"))
        f1w, f1b = pack.interleave_geglu(g(t + "ff.net.0.proj.weight"), g(t + "ff.net.0.proj.bias"))       # GEGLU fused into the GEMM epilogue
        self.f1w, self.f1b = pack.pack_linear(f1w), pack.f32(f1b)
        self.f2w, self.f2b = pack.pack_linear(g(t + "ff.net.2.weight")), pack.f32(g(t + "ff.net.2.bias"))

    def load_custom_diffusion(self, sd, prefix, dev):
        """Custom-Diffusion attention processor of this block (diffusers CustomDiffusionAttnProcessor, loaded by the reference through
        `unet.load_attn_procs`, sd.py:56-58): fine-tuned cross-attention K/V projections (and optionally Q / out) replace the base ones."""
        k, v = prefix + "to_k_custom_diffusion.weight", prefix + "to_v_custom_diffusion.weight"
        if k in sd and v in sd:
            self.kv2 = pack.pack_linear(torch.cat([sd[k].to(dev), sd[v].to(dev)], 0))
        q = prefix + "to_q_custom_diffusion.weight"
        if q in sd:
            self.q2 = pack.pack_linear(sd[q].to(dev))
        o = prefix + "to_out_custom_diffusion.0.weight"
        if o in sd:
            self.o2w = pack.pack_linear(sd[o].to(dev))
            self.o2b = pack.f32(sd[prefix + "to_out_custom_diffusion.0.bias"].to(dev))
        return k in sd

    def context_kv(self, ctx):
        """cross-attention keys / values of a text embedding: they depend on the prompt only, not on the latents or t (views of the fused
        k|v projection: the attention kernel reads V row-major since round 4)"""
        C = self.q2.shape[0]
        kv = ops.linear(ctx, self.kv2)                                                   # [B, 77, 2C]
        return kv[..., :C], kv[..., C:]

    def __call__(self, x, ctx, groups, pool, x_sums=None, out_gn=False, kv=None):
        """-> (y, y_sums | None), see _Resnet.__call__.  kv: this block's context_kv(ctx) when the caller caches it per prompt."""
        B, H, W, C = x.shape
        if x_sums is not None:
            h, _ = ops.groupnorm(x, self.nw, self.nb, groups, 1e-6, False, sums=x_sums[0], sums_ready=x_sums[1])
        else:
            h, _ = ops.groupnorm(x, self.nw, self.nb, groups, 1e-6, False, pool)
        # (each LayerNorm rides on the tail kernel of the GEMM that produces its input when that GEMM runs split-K — the 8^2 / 16^2 / 32^2 levels —
        # and is its own launch otherwise: ops.linear(ln=))
        h, n = ops.linear(h.view(B, H * W, C), self.piw, bias=self.pib, ln=self.ln[0])   # 1x1 conv on NHWC = linear over channels
        qkv = ops.linear(n, self.qkv1)                                                   # [B, T, 3C]
        a = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.heads)
        h, n = ops.linear(a, self.o1w, bias=self.o1b, residual=h, ln=self.ln[1])
        q = ops.linear(n, self.q2)
        if kv is None:
            kv = self.context_kv(ctx)
        a = ops.attention(q, kv[0], kv[1], self.heads)
        h, n = ops.linear(a, self.o2w, bias=self.o2b, residual=h, ln=self.ln[2])
        f = ops.linear(n, self.f1w, bias=self.f1b, act=ops.ACT_GEGLU)                     # [B, T, 4C]
        h = ops.linear(f, self.f2w, bias=self.f2b, residual=h)
        if not out_gn:
            return ops.linear(h, self.pow, bias=self.pob, residual=x.view(B, H * W, C)).view(B, H, W, C), None
        so = pool.take()
        y, ok = ops.linear(h, self.pow, bias=self.pob, residual=x.view(B, H * W, C), gn=(so, groups, H * W))
        return y.view(B, H, W, C), (so, ok)


class UNet:
    def __init__(self, cfg, state_dict, device="cuda"):
        self.cfg = cfg
        dev = torch.device(device)
        sd = state_dict
        g = lambda k: sd[k].to(dev)
        boc = cfg["block_out_channels"]
        self.t1w, self.t1b = pack.pack_linear(g("time_embedding.linear_1.weight")), pack.f32(g("time_embedding.linear_1.bias"))
        self.t2w, self.t2b = pack.pack_linear(g("time_embedding.linear_2.weight")), pack.f32(g("time_embedding.linear_2.bias"))
        self.ciw, self.cib = pack.pack_conv(g("conv_in.weight")), pack.f32(g("conv_in.bias"))
        self.down = []
        for i in range(len(boc)):
            res = [_Resnet(sd, f"down_blocks.{i}.resnets.{j}.", dev) for j in range(cfg["layers_per_block"])]
            att = [_Transformer(sd, f"down_blocks.{i}.attentions.{j}.", dev, cfg["heads"]) for j in range(cfg["layers_per_block"])] if cfg["attn_blocks"][i] else None
            ds = None
            if i < len(boc) - 1:
                ds = (pack.pack_conv(g(f"down_blocks.{i}.downsamplers.0.conv.weight")), pack.f32(g(f"down_blocks.{i}.downsamplers.0.conv.bias")))
            self.down.append((res, att, ds))
        self.mid = (_Resnet(sd, "mid_block.resnets.0.", dev), _Transformer(sd, "mid_block.attentions.0.", dev, cfg["heads"]), _Resnet(sd, "mid_block.resnets.1.", dev))
        attn_up = tuple(reversed(cfg["attn_blocks"]))
        self.up = []
        for i, chans in enumerate(unet_up_channels(cfg)):
            res = [_Resnet(sd, f"up_blocks.{i}.resnets.{j}.", dev) for j in range(len(chans))]
            att = [_Transformer(sd, f"up_blocks.{i}.attentions.{j}.", dev, cfg["heads"]) for j in range(len(chans))] if attn_up[i] else None
            us = None
            if i < len(boc) - 1:
                us = (pack.pack_conv(g(f"up_blocks.{i}.upsamplers.0.conv.weight")), pack.f32(g(f"up_blocks.{i}.upsamplers.0.conv.bias")))
            self.up.append((res, att, us))
        self.now, self.nob = pack.f32(g("conv_norm_out.weight")), pack.f32(g("conv_norm_out.bias"))
        self.cow, self.cob = pack.pack_conv(g("conv_out.weight")), pack.f32(g("conv_out.bias"))
        # the 22 time_emb_proj linears all read silu(temb): stack them into one [sum Cout, 4C] GEMM (one launch instead of 22)
        self._resnets = [r for res, _, _ in self.down for r in res] + [self.mid[0], self.mid[2]] + [r for res, _, _ in self.up for r in res]
        self.tpw = torch.cat([r.tw for r in self._resnets], 0).contiguous()
        self.tpb = torch.cat([r.tb for r in self._resnets], 0).contiguous()
        off = 0
        for r in self._resnets:
            r.t_off, r.t_n = off, r.tw.shape[0]
            off += r.t_n
            del r.tw, r.tb
        self._graph = None

    def load_attn_procs(self, state_dict):
        """`pipe.unet.load_attn_procs(model_id, weight_name="pytorch_custom_diffusion_weights.bin")` (sd.py:57): keys
        `<block>.attentions.<j>.transformer_blocks.0.attn2.processor.to_{k,v}_custom_diffusion.weight`.  Returns the number of
        cross-attention layers that were replaced; invalidates a captured graph."""
        n = 0
        dev = self.cow.device
        named = []
        for i, (_, att, _) in enumerate(self.down):
            named += [(f"down_blocks.{i}.attentions.{j}.", a) for j, a in enumerate(att or [])]
        named.append(("mid_block.attentions.0.", self.mid[1]))
        for i, (_, att, _) in enumerate(self.up):
            named += [(f"up_blocks.{i}.attentions.{j}.", a) for j, a in enumerate(att or [])]
        for name, blk in named:
            n += int(blk.load_custom_diffusion(state_dict, name + "transformer_blocks.0.attn2.processor.", dev))
        self._graph = None
        return n

    def transformers(self):
        out = [a for _, att, _ in self.down for a in (att or [])] + [self.mid[1]] + [a for _, att, _ in self.up for a in (att or [])]
        return out

    def context_kv(self, ctx):
        """Cross-attention K / V^T of all 16 transformer blocks for one text embedding [B, 77, D] (32 launches that depend on the prompt
        only): computed once per prompt and passed to forward(..., ctx_kv=...) / cached by graphed()."""
        ctx = ctx.to(torch.float16).contiguous()
        return {id(a): a.context_kv(ctx) for a in self.transformers()}

    def forward(self, x, t, ctx, ctx_kv=None):
        """x [B, h, w, 8] half (4 latent channels + zero padding), t [B] float32 on device, ctx [B, 77, D] half
        -> eps [B, h, w, 4] half.  ctx_kv: context_kv(ctx) (then ctx itself is not read)."""
        kvs = ctx_kv if ctx_kv is not None else {}
        cfg = self.cfg
        G, eps = cfg["groups"], cfg["eps"]
        temb = ops.timestep_embedding(t, cfg["block_out_channels"][0])
        temb = ops.linear(ops.linear(temb, self.t1w, bias=self.t1b, act=ops.ACT_SILU), self.t2w, bias=self.t2b)
        temb_act = ops.silu(temb)                                                      # every resnet applies SiLU before time_emb_proj
        tp = ops.linear(temb_act, self.tpw, bias=self.tpb, out32=True)                 # [B, sum Cout] float32
        pool = ops.SumsPool(4 * len(self._resnets) + 34, x.shape[0], G, x.device)        # one zero-fill for every GroupNorm of the pass (a few slices go unused)
        tbs = {id(r): tp[:, r.t_off:r.t_off + r.t_n] for r in self._resnets}          # per-block [B, Cout] strided views (bias_rows operand)
        hs0 = pool.take()
        h, ok = ops.conv2d(x, self.ciw, self.cib, 3, gn=(hs0, G, x.shape[1] * x.shape[2]))
        hs = (hs0, ok)                                                             # statistics of h for the norm that reads it next
        skips = [h]
        for res, att, ds in self.down:
            for j, r in enumerate(res):
                h, hs = r(h, tbs[id(r)], G, eps, pool, x_sums=hs, out_gn=True)
                if att is not None:
                    h, hs = att[j](h, ctx, G, pool, x_sums=hs, out_gn=True, kv=kvs.get(id(att[j])))
                skips.append(h)
            if ds is not None:
                so = pool.take()
                h, ok = ops.conv2d(h, ds[0], ds[1], 3, stride=2, pad=1, gn=(so, G, (h.shape[1] // 2) * (h.shape[2] // 2)))
                hs = (so, ok)
                skips.append(h)
        h, hs = self.mid[0](h, tbs[id(self.mid[0])], G, eps, pool, x_sums=hs, out_gn=True)
        h, hs = self.mid[1](h, ctx, G, pool, x_sums=hs, out_gn=True, kv=kvs.get(id(self.mid[1])))
        h, hs = self.mid[2](h, tbs[id(self.mid[2])], G, eps, pool, x_sums=hs, out_gn=False)
        n_up = len(self.up)
        for bi, (res, att, us) in enumerate(self.up):
            for j, r in enumerate(res):
                # the block input is a channel concat: the concat kernel accumulates its norm's statistics (round 6; a k_gn_stats launch before);
                # the output feeds a norm only through attention
                sc_ = pool.take()
                hc, okc = ops.concat_channels(h, skips.pop(), gn=(sc_, G, h.shape[1] * h.shape[2]))
                h, hs = r(hc, tbs[id(r)], G, eps, pool, x_sums=(sc_, okc), out_gn=att is not None)
                if att is not None:
                    last = bi == n_up - 1 and j == len(res) - 1                     # -> conv_norm_out
                    h, hs = att[j](h, ctx, G, pool, x_sums=hs, out_gn=last, kv=kvs.get(id(att[j])))
            if us is not None:
                h = ops.conv2d(h, us[0], us[1], 3, ups=2)
        if hs is not None:
            h, _ = ops.groupnorm(h, self.now, self.nob, G, eps, True, sums=hs[0], sums_ready=hs[1])
        else:
            h, _ = ops.groupnorm(h, self.now, self.nob, G, eps, True, pool)
        return ops.conv2d(h, self.cow, self.cob, 3)

    __call__ = forward

    def _graph_entry(self, shape, ctx, x=None, t=None):
        if self._graph is None:
            self._graph = {}
        key = (tuple(shape), ctx.data_ptr(), ctx._version, tuple(ctx.shape))
        ent = self._graph.get(key)
        if ent is None:
            if len(self._graph) >= 4:                                                  # small LRU: drop the oldest context
                self._graph.pop(next(iter(self._graph)))
            sx = x.clone() if x is not None else torch.zeros(shape, dtype=torch.float16, device=ctx.device)
            st = t.clone() if t is not None else torch.full((shape[0],), 500.0, dtype=torch.float32, device=ctx.device)
            with torch.no_grad():
                kv = self.context_kv(ctx)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.forward(sx, st, None, ctx_kv=kv)                                  # warm-up: workspace growth, function attributes
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.forward(sx, st, None, ctx_kv=kv)
            ent = (graph, sx, st, (kv, ctx), out)                                   # holding ctx keeps its storage (the cache key) from being recycled
            self._graph[key] = ent
        return ent

    def graph_inputs(self, shape, ctx, create=True):
        """the static input buffers (x [2V, h, w, 8] half, t [2V] float32) of the graph for `ctx`: a caller that writes them itself (the SDS
        step: cnerf_sd_add_noise / cnerf_set_floats straight into them) replays with graphed(None, None, ctx, shape=shape).
        create=False: None when that graph has not been captured yet."""
        if not create and (self._graph is None or (tuple(shape), ctx.data_ptr(), ctx._version, tuple(ctx.shape)) not in self._graph):
            return None
        ent = self._graph_entry(shape, ctx)
        return ent[1], ent[2]

    def graphed(self, x, t, ctx, shape=None):
        """Same as forward() but replayed from a HIP graph captured on first use (static shapes; x and t are copied into the graph's
        buffers — unless both are None: the caller filled graph_inputs() — and the returned tensor is the graph's static output buffer,
        overwritten by the next call).  One graph per text embedding (keyed by the tensor's storage and version; the editing loop alternates
        between two prompts): its cross-attention K / V^T are computed once, outside the graph.  ~550 launches per forward would otherwise
        be host-bound."""
        graph, sx, st, _, out = self._graph_entry(shape if x is None else x.shape, ctx, x, t)
        if x is not None:
            sx.copy_(x); st.copy_(t)
        graph.replay()
        return out
