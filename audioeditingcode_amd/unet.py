"""U-Net forward compiled to an op tape.

One config-driven builder covers the three wrappers' U-Nets:
  * diffusers UNet2DConditionModel as the reference drives it in PipelineWrapper.unet_forward
    (/root/reference/code/models.py:160-393): AudioLDM-1 (CLAP FiLM through class_embedding,
    attn2 degenerates to self-attention because encoder_hidden_states is None) and TANGO
    (T5 cross-attention with an additive -10000 key mask);
  * AudioLDM2UNet2DConditionModel as driven by AudioLDM2Wrapper.unet_forward (models.py:691-899):
    three transformers per attention site -- [self,self], [self,cross->GPT-2 states],
    [self,cross->T5 states + mask].
Hooks of the reference forward (h-space tap/replace/add models.py:840-847, skip replace/zero
:854-866) are exposed through `mid_index` (the tape is cut after the mid block) and `skips`.

Everything here is host-side graph construction; arithmetic happens in libaed.so.
Activations are channels-last; the NCHW<->NHWC change happens only at the wrapper boundary.
"""
import torch

from . import _lib as L
from .tape import Tape
from .weights import ctx_dims_per_block, _per_block


# graph-level fusions of DESIGN.md section 2 (module constants; tools/unet_profile.py flips them for A/B runs)
TWO_SOURCE = True           # up-block concats read in place by GroupNorm and the shortcut conv
MERGE_FF2_PROJ = True       # FF2 o proj_out as one two-source GEMM with host-folded weights
FOLD_XATTN = True           # cross-attention over short constant keys as two skinny GEMMs (latency regime)


def geglu_pack_index(dff):
    """Row order of a GEGLU projection [2*dff, K] (value rows, then gate rows: diffusers GEGLU.chunk(2)) after packing
    for the fused epilogue: block q = rows [value 32q..32q+31 | gate 32q..32q+31]."""
    q = torch.arange(dff // 32)[:, None]
    j = torch.arange(32)[None, :]
    return torch.cat([32 * q + j, dff + 32 * q + j], 1).reshape(-1)


class PackedUNetWeights:
    """Device-resident, engine-layout copy of a diffusers-named U-Net state dict (shared by every
    UNetEngine built for the same model: batch shapes differ, weights do not)."""

    def __init__(self, sd, device):
        self.device = torch.device(device)
        self.wd = {}
        self._pack(sd)

    def _dev(self, t):
        return t.contiguous().to(self.device, torch.float32)

    def nbytes(self):
        return sum(v.numel() * 4 for v in self.wd.values())

    def _fold_ln(self, name, w, gamma, beta, bias):
        """Fold a LayerNorm's affine part into the following Linear (fp64 on the host, stored fp32)."""
        w64, g64, b64 = w.double(), gamma.double(), beta.double()
        wf = w64 * g64[None, :]
        t = w64 @ b64
        if bias is not None:
            t = t + bias.double()
        self.wd[name + ".weight"] = self._dev(wf.float())
        self.wd[name + ".rowsum"] = self._dev(wf.float().double().sum(1).float())
        self.wd[name + ".t"] = self._dev(t.float())

    def ensure_folded_cross_attention(self, base, H):
        """Per-head operands of the folded cross-attention of module `base` (= "...attn2"; see
        UNetEngine._folded_cross_attention), folded in fp64 on the host, built on first use.  For head h of width D:
          xq[h]  [C, D]   = (gamma o Wq_h)^T        -> G'[b,(h,j),:] = k_h[b,j] . xq[h]^T  (scores = LN-folded x . G'^T)
          xs[h]  [2, D]   = (Wq_h gamma, Wq_h beta) -> (sum_c G', G . beta): the two LayerNorm-fold vectors of that GEMM
          xo[h]  [C, D]   = Wo[:, hD:(h+1)D]        -> VO[b,(h,j),:] = v_h[b,j] . xo[h]^T  (out = P . VO + bo)"""
        if base + ".xq" in self.wd:
            return
        nrm = base.rsplit(".", 1)[0] + ".norm2"
        wq = self._sd[base + ".to_q.weight"].double()
        wo = self._sd[base + ".to_out.0.weight"].double()
        g, bta = self._sd[nrm + ".weight"].double(), self._sd[nrm + ".bias"].double()
        C = wq.shape[0]
        D = C // H
        wqh = wq.reshape(H, D, C)                                        # [h, d, c] = Wq[hD+d, c]
        self.wd[base + ".xq"] = self._dev((wqh * g[None, None, :]).permute(0, 2, 1).float())       # [H, C, D]
        self.wd[base + ".xs"] = self._dev(torch.stack([(wqh * g).sum(2), (wqh * bta).sum(2)], 1).float())  # [H, 2, D]
        self.wd[base + ".xo"] = self._dev(wo.reshape(C, H, D).permute(1, 0, 2).float())           # [H, C, D]

    def _pack(self, sd):
        """Re-lay weights: conv [O,I,kh,kw] -> [O, kh*kw*I]; fuse q/k/v; concatenate temb projections."""
        wd = self.wd
        self._sd = sd
        temb_w, temb_b, self.temb_off = [], [], {}
        off = 0
        for k, v in sd.items():
            if k.endswith("time_emb_proj.weight"):
                base = k[: -len(".weight")]
                self.temb_off[base] = off
                off += v.shape[0]
                temb_w.append(v)
                temb_b.append(sd[base + ".bias"])
        self.temb_total = off
        wd["temb_all.weight"] = self._dev(torch.cat(temb_w, 0))
        wd["temb_all.bias"] = self._dev(torch.cat(temb_b, 0))
        for k, v in sd.items():
            if "time_emb_proj" in k:
                continue
            if v.dim() == 4:
                v = v.permute(0, 2, 3, 1).reshape(v.shape[0], -1)
            if k.endswith(".to_q.weight"):
                base = k[: -len(".to_q.weight")]
                wq, wk, wv = v, sd[base + ".to_k.weight"], sd[base + ".to_v.weight"]
                # the LayerNorm in front of this attention (norm1 -> attn1, norm2 -> attn2) is folded into the
                # projection: LN(x).W^T = rstd*(x.(W*gamma)^T - mean*rowsum(W*gamma)) + W.beta
                blk, which = base.rsplit(".", 1)
                nrm = blk + (".norm1" if which == "attn1" else ".norm2")
                if wk.shape[1] == wq.shape[1]:
                    self._fold_ln(base + ".qkv_ln", torch.cat([wq, wk, wv], 0), sd[nrm + ".weight"], sd[nrm + ".bias"], None)
                self._fold_ln(base + ".q_ln", wq, sd[nrm + ".weight"], sd[nrm + ".bias"], None)
                wd[base + ".kv.weight"] = self._dev(torch.cat([wk, wv], 0))
                continue
            if k.endswith(".ff.net.0.proj.weight"):
                # FF1: LayerNorm (norm3) folded in, GEGLU fused into the epilogue -- rows packed [32 value | 32 gate] per
                # 32 features, so one wavefront's adjacent 32-column sub-tiles hold value and gate of the same features
                blk = k[: -len(".ff.net.0.proj.weight")]
                dff = v.shape[0] // 2
                if dff % 32:
                    raise NotImplementedError(f"{k}: feed-forward width {dff} is not a multiple of 32")
                perm = geglu_pack_index(dff)
                b1 = sd[blk + ".ff.net.0.proj.bias"]
                self._fold_ln(blk + ".ff1g_ln", v[perm], sd[blk + ".norm3.weight"], sd[blk + ".norm3.bias"], b1[perm])
                continue
            if k.endswith(".ff.net.0.proj.bias") or (".transformer_blocks." in k and k.rsplit(".", 2)[-2].startswith("norm")):
                continue            # consumed by the folds above (FF1 bias, LayerNorm affine parameters)
            if k.endswith(".ff.net.2.weight"):
                # FF2 followed by the site's proj_out is one linear map of [f | t2] (exact algebra, folded in fp64):
                #   proj_out(ff2(f) + t2) = f.(Wp.W2)^T + t2.Wp^T + (Wp.b2 + bp)
                # -> one GEMM with K = 4C + C over a two-source A instead of two dependent launches (same FLOPs)
                blk = k[: -len(".ff.net.2.weight")]
                site = blk[: blk.index(".transformer_blocks")]
                wp = sd[site + ".proj_out.weight"]
                if wp.dim() == 4:
                    wp = wp.reshape(wp.shape[0], -1)
                w2, b2, bp = v.double(), sd[blk + ".ff.net.2.bias"].double(), sd[site + ".proj_out.bias"].double()
                wp = wp.double()
                if w2.shape[1] % 64 == 0:
                    wd[blk + ".ff2_proj.weight"] = self._dev(torch.cat([wp @ w2, wp], 1).float())
                    wd[blk + ".ff2_proj.bias"] = self._dev((wp @ b2 + bp).float())
            if k.endswith(".to_k.weight") or k.endswith(".to_v.weight"):
                continue
            wd[k] = self._dev(v)


class UNetEngine:
    def __init__(self, cfg, weights, device, batch, H, W, ctx_len0=0, ctx_len1=0, use_ehs=True,
                 timesteps_dev=None, state_dev=None, two_source=None, share=1):
        self.cfg = cfg
        # share = S > 1 (round 5, "CFG row sharing"): the caller promises that batch rows S*p .. S*p + S - 1 carry the SAME sample and
        # timestep and differ only in their text context -- the [uncond | prompt ...] rows of one clip (inversion_utils.py:82-93 runs
        # the U-Net once per row on `xt` / `xt.expand(...)`).  Everything upstream of the first module that reads the context --
        # conv_in, the context-free down block(s), the first resnet and the double-self-attention transformer of the first attention
        # site (AudioLDM2: cross_attention_dim [None, 768, 1024]) -- is then computed ONCE per group of S rows at batch B / S and
        # its outputs (the stream and the skip tensors) are expanded to the full batch by strided copies: 12.3 of a sample
        # forward's 172.4 GF, 3.6 % of a CFG pair.  Not for class-conditioned models (AudioLDM-1: the FiLM embedding enters every
        # resnet).  `Tape.flops` keeps the algorithmic count of the full batch.
        self.S = int(share) if (share and share > 1 and batch % int(share) == 0 and
                                cfg.get("class_embed_type") is None) else 1
        self.cB = batch             # batch of the module being laid out (batch / S inside the shared prefix)
        # Every LayerNorm is folded into the GEMM that consumes it and the GEGLU gate rides in the FF1 epilogue (the
        # standalone kernels were retired in ABI v4); up-block concats are read in place (no copy launches).
        self.two_source = TWO_SOURCE if two_source is None else two_source
        # FF2 and the site's proj_out as one GEMM over [f | t2] with host-folded weights (one dependent launch less)
        self.merge_ff2_proj = MERGE_FF2_PROJ and self.two_source
        # cross-attention over the (short, per-prompt constant) text keys folded into two skinny GEMMs (latency regime)
        self.fold_xattn = FOLD_XATTN
        # GroupNorm(+SiLU) inside the conv A-loader is implemented and parity-tested but OFF by default: measured
        # on MI355X it is a wash at U-Net batch 2 (11.53 vs 11.51 ms/forward) and 5 % slower at batch 32
        # (77.8 vs 73.8 ms): the loader's 9x-per-tap SiLU recompute costs more than the saved launch + round trip.
        self.device = torch.device(device)
        self.B, self.H, self.W = batch, H, W
        self.use_ehs = use_ehs
        self.L0, self.L1 = ctx_len0, ctx_len1
        self.timesteps_dev, self.state_dev = timesteps_dev, state_dev
        self.tape = Tape(device)
        self.ctx_tape = Tape(device)
        self._tmp = {}
        # ops laid out inside the shared prefix execute 1 / S of the reference formulation's work: `flops` (ALGORITHMIC, the
        # reference's count for the full batch) is scaled back after the build, `exec_flops` stays what the kernels execute
        self._prefix_idx = []
        _add = self.tape._add

        def counted_add(*a, **k):
            idx = _add(*a, **k)
            if self.cB != self.B:
                self._prefix_idx.append(idx)
            return idx
        self.tape._add = counted_add
        if not isinstance(weights, PackedUNetWeights):
            weights = PackedUNetWeights(weights, device)
        self.weights = weights
        self.wd, self.temb_off, self.temb_total = weights.wd, weights.temb_off, weights.temb_total
        self._build()
        del self.tape._add                          # back to the class method
        for idx in self._prefix_idx:
            self.tape.meta[idx]["flops"] *= self.S

    def tmp(self, tag, *shape):
        key = (tag, tuple(shape))
        if key not in self._tmp:
            self._tmp[key] = self.tape.alloc(*shape)
        return self._tmp[key]

    # ------------------------------------------------------------------ modules
    def _resnet(self, p, x, Cin, Cout, H, W, dest, groups, eps, x2=None, C1=0):
        """ResnetBlock2D.  x2/C1: the block input is the channel concat (x[..., :C1] | x2) of an up block
        (models.py:349-357), read in place by GroupNorm and the shortcut conv -- never materialised."""
        tp, wd, B = self.tape, self.wd, self.cB
        h = self.tmp("res_h", B, H, W, Cout)
        off = self.temb_off[p + ".time_emb_proj"]
        a = self.tmp("gn_a", B, H, W, Cin)
        tp.groupnorm(x, wd[p + ".norm1.weight"], wd[p + ".norm1.bias"], a, B=B, HW=H * W, C=Cin, G=groups,
                     eps=eps, act=L.ACT_SILU, x2=x2, C1=C1, name=p + ".norm1")
        tp.conv(a, wd[p + ".conv1.weight"], wd[p + ".conv1.bias"], h, B=B, IH=H, IW=W, Cin=Cin, OH=H, OW=W, N=Cout,
                KH=3, KW=3, pad_h=1, pad_w=1, rowvec=self.temb_all[:, off:off + Cout],
                ld_rv=self.temb_total * (self.B // B),      # inside the shared prefix row p stands for full rows S*p ..: their temb
                name=p + ".conv1")
        a2 = self.tmp("gn_a2", B, H, W, Cout)
        tp.groupnorm(h, wd[p + ".norm2.weight"], wd[p + ".norm2.bias"], a2, B=B, HW=H * W, C=Cout, G=groups,
                     eps=eps, act=L.ACT_SILU, name=p + ".norm2")
        res = x
        if (p + ".conv_shortcut.weight") in wd:
            res = self.tmp("res_sc", B, H, W, Cout)
            tp.conv(x, wd[p + ".conv_shortcut.weight"], wd[p + ".conv_shortcut.bias"], res, B=B, IH=H, IW=W, Cin=Cin,
                    OH=H, OW=W, N=Cout, x2=x2, C1=C1, name=p + ".conv_shortcut")
        else:
            assert x2 is None, "a concatenated input always changes the channel count"
        tp.conv(a2, wd[p + ".conv2.weight"], wd[p + ".conv2.bias"], dest, B=B, IH=H, IW=W, Cin=Cout, OH=H, OW=W,
                N=Cout, KH=3, KW=3, pad_h=1, pad_w=1, res=res, name=p + ".conv2")
        return dest

    def _ln_linear(self, x, pfx, out, M, K, N, name):
        """Linear on LayerNorm(x): raw x + gamma-folded weights + row statistics gathered inside the GEMM (no LayerNorm
        launch, no normalised copy in HBM)."""
        wd = self.wd
        self.tape.linear(x, wd[pfx + ".weight"], wd[pfx + ".t"], out, M=M, K=K, N=N, ln_rowsum=wd[pfx + ".rowsum"],
                         name=name + "+ln")

    def _attn(self, p, x, C, N, heads, out, kv=None, Lk=0, bias=None):
        """attention sub-layer: LayerNorm-folded input projections + fused attention.  x: [B*N, C] (raw, un-normalised)."""
        tp, B = self.tape, self.cB
        M = B * N
        D = C // heads
        if kv is None:
            qkv = self.tmp("t_qkv", M, 3 * C)
            self._ln_linear(x, p + ".qkv_ln", qkv, M, C, 3 * C, p + ".qkv")
            tp.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], out, B=B, H=heads, Nq=N, Nk=N, D=D, ldq=3 * C, ldk=3 * C,
                         ldv=3 * C, ldo=C, bsq=N * 3 * C, bsk=N * 3 * C, bsv=N * 3 * C, bso=N * C,
                         scale=D ** -0.5, name=p + ".sdpa")
        else:
            q = self.tmp("t_q", M, C)
            self._ln_linear(x, p + ".q_ln", q, M, C, C, p + ".q")
            tp.attention(q, kv, kv[:, C:], out, B=B, H=heads, Nq=N, Nk=Lk, D=D, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C,
                         bsq=N * C, bsk=Lk * 2 * C, bsv=Lk * 2 * C, bso=N * C, scale=D ** -0.5, bias=bias,
                         ld_bias=Lk if bias is not None else 0, name=p + ".sdpa_x")
        return out

    def _transformer(self, p, x, C, H, W, heads, kind, dest, groups):
        tp, wd, B = self.tape, self.wd, self.cB
        N = H * W
        M = B * N
        t0 = self.tmp("t_0", M, C)
        n = self.tmp("t_n", M, C)
        tp.groupnorm(x, wd[p + ".norm.weight"], wd[p + ".norm.bias"], n, B=B, HW=N, C=C, G=groups, eps=1e-6,
                     name=p + ".norm")
        tp.linear(n, wd[p + ".proj_in.weight"], wd[p + ".proj_in.bias"], t0, M=M, K=C, N=C, name=p + ".proj_in")
        b = p + ".transformer_blocks.0"
        o = self.tmp("t_o", M, C)
        self._attn(b + ".attn1", t0, C, N, heads, o)
        t1 = self.tmp("t_1", M, C)
        tp.linear(o, wd[b + ".attn1.to_out.0.weight"], wd[b + ".attn1.to_out.0.bias"], t1, M=M, K=C, N=C, res=t0,
                  name=b + ".attn1.to_out")
        if kind == "self2":
            self._attn(b + ".attn2", t1, C, N, heads, o)
        else:
            which = 0 if kind == "cross0" else 1
            Lk = self.L0 if which == 0 else self.L1
            ctx = self.ehs0 if which == 0 else self.ehs1
            cdim = ctx.shape[-1]
            kv = self.ctx_tape.alloc(B * Lk, 2 * C)
            self.ctx_tape.linear(ctx.view(B * Lk, cdim), wd[b + ".attn2.kv.weight"], None, kv, M=B * Lk, K=cdim,
                                 N=2 * C, name=b + ".attn2.kv")
            bias = self.bias0 if which == 0 else self.bias1
            if self._fold_xattn_ok(C, N, heads, Lk):
                return self._folded_cross_attention(p, b, t1, kv, Lk, bias, C, N, heads, x, dest)
            self._attn(b + ".attn2", t1, C, N, heads, o, kv=kv, Lk=Lk, bias=bias)
        t2 = self.tmp("t_2", M, C)
        tp.linear(o, wd[b + ".attn2.to_out.0.weight"], wd[b + ".attn2.to_out.0.bias"], t2, M=M, K=C, N=C, res=t1,
                  name=b + ".attn2.to_out")
        return self._ff_and_out(p, b, t2, C, M, x, dest)

    def _fold_xattn_ok(self, C, N, heads, Lk):
        """Folded cross-attention: latency regime only (lin_gemm kernels), key count a power of two <= 32 (one softmax
        group per head inside a 32-column tile), 64-row aligned batch items."""
        return (self.fold_xattn and Lk in (8, 16, 32) and N % 64 == 0 and (heads * Lk) % 32 == 0 and
                self.cB * N <= 4096 and C % 32 == 0)

    def _folded_cross_attention(self, p, b, t1, kv, Lk, kbias, C, N, heads, x, dest):
        """Cross-attention over a SHORT, per-prompt-constant key set as two skinny GEMMs (exact algebra, no attention
        launch, ~C/(heads*Lk) x fewer FLOPs than q-projection + to_out):
            scores[m,(h,j)] = LN(t1)[m] . Wq_h^T k_h[j]      = LN-folded  t1[m] . G'[b]^T
            t2[m]           = sum_(h,j) softmax_j(scores)[m,(h,j)] . (v_h[j] Wo_h^T) + bo + t1[m]
        G' = gamma o (k_h Wq_h) and VO = v_h Wo_h^T depend only on the prompt: they are built once per set_conditioning
        on the context tape (one small launch per block, AED_OP_XATTN_FOLD) and indexed per batch item by the lin_gemm kernels."""
        tp, ct, wd, B = self.tape, self.ctx_tape, self.wd, self.cB
        base = b + ".attn2"
        self.weights.ensure_folded_cross_attention(base, heads)
        HL, D, M = heads * Lk, C // heads, B * N
        G = ct.alloc(B, HL, C)
        gs = ct.alloc(B, HL, 2)
        VOt = ct.alloc(B, C, HL)
        ct.xattn_fold(kv, wd[base + ".xq"], wd[base + ".xs"], wd[base + ".xo"], G, gs, VOt, B=B, Lk=Lk, H=heads, C=C, D=D,
                      name=base + ".fold")
        P = self.tmp("t_p", M, HL)
        gsf = gs.view(-1)
        tp.conv(t1, G, gsf[1:], P, B=B, IH=N, IW=1, Cin=C, OH=N, OW=1, N=HL, ln_rowsum=gsf, w_bs=HL * C, vec_ld=2,
                vec_bs=HL * 2, sm_group=Lk, sm_scale=D ** -0.5, kbias=kbias, name=base + ".scores+softmax",
                alg_flops=2 * M * C * C + 4 * M * Lk * C)       # what it replaces: q projection + QK^T + PV
        t2 = self.tmp("t_2", M, C)
        tp.conv(P, VOt, wd[base + ".to_out.0.bias"], t2, B=B, IH=N, IW=1, Cin=HL, OH=N, OW=1, N=C, res=t1, w_bs=C * HL,
                name=base + ".PV+to_out", alg_flops=2 * M * C * C)                  # what it replaces: to_out
        return self._ff_and_out(p, b, t2, C, M, x, dest)

    def _ff_and_out(self, p, b, t2, C, M, x, dest):
        tp, wd = self.tape, self.wd
        f = self.tmp("t_f", M, 4 * C)
        # FF1 with the LayerNorm fold and the GEGLU gate in its epilogue: the [M, 8C] projection never reaches HBM
        # (the packed rows exist whenever 4C is a multiple of 32 -- always, GroupNorm already needs 32 | C)
        tp.linear(t2, wd[b + ".ff1g_ln.weight"], wd[b + ".ff1g_ln.t"], f, M=M, K=C, N=8 * C,
                  ln_rowsum=wd[b + ".ff1g_ln.rowsum"], geglu=1, name=b + ".ff1+ln+geglu")
        if self.merge_ff2_proj and (b + ".ff2_proj.weight") in wd:
            tp.linear(f, wd[b + ".ff2_proj.weight"], wd[b + ".ff2_proj.bias"], dest, M=M, K=5 * C, N=C, x2=t2,
                      C1=4 * C, res=x, name=b + ".ff2+proj_out")
            return dest
        t3 = self.tmp("t_3", M, C)
        tp.linear(f, wd[b + ".ff.net.2.weight"], wd[b + ".ff.net.2.bias"], t3, M=M, K=4 * C, N=C, res=t2,
                  name=b + ".ff2")
        tp.linear(t3, wd[p + ".proj_out.weight"], wd[p + ".proj_out.bias"], dest, M=M, K=C, N=C, res=x,
                  name=p + ".proj_out")
        return dest

    def _site(self, prefix, k0, x, C, H, W, heads, dims, groups):
        """One attention site = len(dims) Transformer2D modules (3 for AudioLDM2)."""
        for j, cdim in enumerate(dims):
            if cdim is None or not self.use_ehs:
                kind = "self2"
            elif self.multi and j > 1:
                kind = "cross1"
            else:
                kind = "cross0"
            if kind != "self2" and self.cB != self.B:
                x = self._expand(x, H, W, C)                 # the first module that reads the text context: leave the shared prefix
            dest = self.tape.alloc(self.cB, H, W, C)
            x = self._transformer(f"{prefix}.attentions.{k0 + j}", x, C, H, W, heads, kind, dest, groups)
        return x

    def _expand(self, x, H, W, C, leave=True):
        """[B/S, H, W, C] -> [B, H, W, C] with full row S*p + k = shared row p (S strided copies); `leave`: the builder continues at
        the full batch."""
        S, Bp = self.S, self.B // self.S
        full = self.tape.alloc(self.B, H, W, C)
        n = H * W * C
        for k in range(S):
            self.tape.copy2d(x, full[k:], rows=Bp, cols=n, ld_src=n, ld_dst=S * n, name="cfg_share.expand")
        if leave:
            self.cB = self.B
        return full

    # ------------------------------------------------------------------ graph
    def _build(self):
        cfg, tp, wd, B, H, W = self.cfg, self.tape, self.wd, self.B, self.H, self.W
        boc = cfg["block_out_channels"]
        nb = len(boc)
        lpb = cfg.get("layers_per_block", 2)
        groups = cfg.get("norm_num_groups", 32)
        eps = cfg.get("norm_eps", 1e-5)
        heads_pb = _per_block(cfg.get("num_attention_heads") or cfg.get("attention_head_dim", 8), nb)
        ctx_pb, self.multi = ctx_dims_per_block(cfg)
        cin, cout = cfg["in_channels"], cfg["out_channels"]
        ted = boc[0] * 4
        has_class = cfg.get("class_embed_type") is not None
        concat = bool(cfg.get("class_embeddings_concat"))
        emb_dim = 2 * ted if (has_class and concat) else ted

        # ---- persistent inputs
        self.x_in = tp.alloc(B, H, W, cin, zero=True)
        self.ehs0 = self.ehs1 = self.bias0 = self.bias1 = self.class_labels = None
        if self.use_ehs:
            d0 = [c for blk in ctx_pb for c in blk if c is not None]
            if self.multi:
                dims0 = sorted({blk[1] for blk in ctx_pb if blk[1] is not None})
                dims1 = sorted({blk[2] for blk in ctx_pb if len(blk) > 2 and blk[2] is not None})
                self.ehs0 = tp.alloc(B, self.L0, dims0[0], zero=True)
                self.ehs1 = tp.alloc(B, self.L1, dims1[0], zero=True)
                self.bias1 = tp.alloc(B, self.L1, zero=True)
            else:
                self.ehs0 = tp.alloc(B, self.L0, d0[0], zero=True)
                self.bias0 = tp.alloc(B, self.L0, zero=True)
        if has_class:
            self.class_labels = tp.alloc(B, cfg["projection_class_embeddings_input_dim"], zero=True)

        # ---- time / class embedding, fused time_emb_proj for every resnet
        t_emb = tp.alloc(B, boc[0])
        self.time_op = len(tp.ops)
        tp.time_embed(t_emb, B=B, dim=boc[0], flip=cfg.get("flip_sin_to_cos", True), shift=cfg.get("freq_shift", 0),
                      timesteps=self.timesteps_dev, state=self.state_dev)
        e1 = tp.alloc(B, ted)
        tp.linear(t_emb, wd["time_embedding.linear_1.weight"], wd["time_embedding.linear_1.bias"], e1, M=B,
                  K=boc[0], N=ted, out_act=L.ACT_SILU, name="time_embedding.linear_1")
        self.emb = tp.alloc(B, emb_dim)
        if has_class and not concat:
            cemb = self.ctx_tape.alloc(B, ted)
            self.ctx_tape.linear(self.class_labels, wd["class_embedding.weight"], wd["class_embedding.bias"], cemb,
                                 M=B, K=self.class_labels.shape[1], N=ted, name="class_embedding")
            tp.linear(e1, wd["time_embedding.linear_2.weight"], wd["time_embedding.linear_2.bias"], self.emb, M=B,
                      K=ted, N=ted, res=cemb, name="time_embedding.linear_2")
        else:
            tp.linear(e1, wd["time_embedding.linear_2.weight"], wd["time_embedding.linear_2.bias"], self.emb, M=B,
                      K=ted, N=ted, name="time_embedding.linear_2")
            if has_class:
                self.ctx_tape.linear(self.class_labels, wd["class_embedding.weight"], wd["class_embedding.bias"],
                                     self.emb[:, ted:], M=B, K=self.class_labels.shape[1], N=ted,
                                     name="class_embedding")
        self.temb_all = tp.alloc(B, self.temb_total)
        tp.linear(self.emb, wd["temb_all.weight"], wd["temb_all.bias"], self.temb_all, M=B, K=emb_dim,
                  N=self.temb_total, in_act=L.ACT_SILU, name="time_emb_proj(all resnets)")

        # ---- conv_in + down.  With CFG row sharing the builder starts at batch B / S (row p reads x_in row S*p: the S rows of a
        # group hold the same sample) and returns to the full batch at the first module that reads the text context (_site).
        if self.S > 1 and not self.use_ehs:
            self.S = 1                              # no context at all: nothing distinguishes the rows, the caller's batch stands
        self.cB = Bc = B // self.S
        h = tp.alloc(Bc, H, W, boc[0])
        tp.conv(self.x_in, wd["conv_in.weight"], wd["conv_in.bias"], h, B=Bc, IH=H, IW=W, Cin=cin, OH=H, OW=W,
                N=boc[0], KH=3, KW=3, pad_h=1, pad_w=1, a_bs=self.S * H * W * cin, name="conv_in")

        def skip_of(t, c_, h_, w_):                 # a skip tensor is consumed at the full batch by the up blocks
            return (t if self.cB == B else self._expand(t, h_, w_, c_, leave=False), c_, h_, w_)
        skips = [skip_of(h, boc[0], H, W)]
        ch, hh, ww = boc[0], H, W
        for i, bt in enumerate(cfg["down_block_types"]):
            co = boc[i]
            for j in range(lpb):
                d = tp.alloc(self.cB, hh, ww, co)
                h = self._resnet(f"down_blocks.{i}.resnets.{j}", h, ch, co, hh, ww, d, groups, eps)
                ch = co
                if "CrossAttn" in bt:
                    h = self._site(f"down_blocks.{i}", j * len(ctx_pb[i]), h, co, hh, ww, heads_pb[i], ctx_pb[i],
                                   groups)
                skips.append(skip_of(h, ch, hh, ww))
            if i < nb - 1:
                oh, ow = (hh + 2 - 3) // 2 + 1, (ww + 2 - 3) // 2 + 1
                d = tp.alloc(self.cB, oh, ow, co)
                p = f"down_blocks.{i}.downsamplers.0.conv"
                tp.conv(h, wd[p + ".weight"], wd[p + ".bias"], d, B=self.cB, IH=hh, IW=ww, Cin=co, OH=oh, OW=ow, N=co,
                        KH=3, KW=3, stride=2, pad_h=1, pad_w=1, name=p)
                h, hh, ww = d, oh, ow
                skips.append(skip_of(h, ch, hh, ww))
        if self.cB != B:                            # no module read the context on the way down (never the case for the three families)
            h = self._expand(h, hh, ww, ch)

        # ---- mid
        d = tp.alloc(B, hh, ww, ch)
        h = self._resnet("mid_block.resnets.0", h, ch, ch, hh, ww, d, groups, eps)
        h = self._site("mid_block", 0, h, ch, hh, ww, heads_pb[-1], ctx_pb[-1], groups)
        self.h_space = tp.alloc(B, hh, ww, ch)
        self._resnet("mid_block.resnets.1", h, ch, ch, hh, ww, self.h_space, groups, eps)
        h = self.h_space
        self.mid_index = len(tp.ops)
        self.skips = [s[0] for s in skips]

        # ---- up
        for i, bt in enumerate(cfg["up_block_types"]):
            lvl = nb - 1 - i
            co = boc[lvl]
            for j in range(lpb + 1):
                sk, sc, sh_, sw_ = skips.pop()
                assert (sh_, sw_) == (hh, ww), "skip / feature-map size mismatch"
                d = tp.alloc(B, hh, ww, co)
                if self.two_source and ch % 64 == 0 and (ch + sc) % 32 == 0 and sc % 4 == 0:
                    # torch.cat([h, skip], dim=1) is never materialised: GroupNorm and the shortcut conv read both
                    h = self._resnet(f"up_blocks.{i}.resnets.{j}", h, ch + sc, co, hh, ww, d, groups, eps, x2=sk, C1=ch)
                else:
                    cat = self.tmp("cat", B, hh, ww, ch + sc)
                    tp.copy2d(h, cat, rows=B * hh * ww, cols=ch, ld_src=h.stride(-2), ld_dst=ch + sc, name="cat.h")
                    tp.copy2d(sk, cat[..., ch:], rows=B * hh * ww, cols=sc, ld_src=sk.stride(-2), ld_dst=ch + sc,
                              name="cat.skip")
                    h = self._resnet(f"up_blocks.{i}.resnets.{j}", cat, ch + sc, co, hh, ww, d, groups, eps)
                ch = co
                if "CrossAttn" in bt:
                    h = self._site(f"up_blocks.{i}", j * len(ctx_pb[lvl]), h, co, hh, ww, heads_pb[lvl], ctx_pb[lvl],
                                   groups)
            if i < nb - 1:
                # nearest resize to the NEXT skip's size (= 2x, or 2x-1 when that level was odd: the reference's
                # forward_upsample_size path, models.py:186-188 / :361-366), fused into the conv's A-loader
                p = f"up_blocks.{i}.upsamplers.0.conv"
                th, tw = skips[-1][2], skips[-1][3]
                assert th in (2 * hh, 2 * hh - 1) and tw in (2 * ww, 2 * ww - 1)
                d = tp.alloc(B, th, tw, co)
                tp.conv(h, wd[p + ".weight"], wd[p + ".bias"], d, B=B, IH=hh, IW=ww, Cin=co, OH=th, OW=tw,
                        N=co, KH=3, KW=3, pad_h=1, pad_w=1, up=1, name=p)
                h, hh, ww = d, th, tw

        # ---- out
        self.eps = tp.alloc(B, hh, ww, cout)
        a = self.tmp("gn_a", B, hh, ww, ch)
        tp.groupnorm(h, wd["conv_norm_out.weight"], wd["conv_norm_out.bias"], a, B=B, HW=hh * ww, C=ch, G=groups,
                     eps=eps, act=L.ACT_SILU, name="conv_norm_out")
        tp.conv(a, wd["conv_out.weight"], wd["conv_out.bias"], self.eps, B=B, IH=hh, IW=ww, Cin=ch, OH=hh, OW=ww,
                N=cout, KH=3, KW=3, pad_h=1, pad_w=1, name="conv_out")
        tp.finalize()
        self.ctx_tape.finalize()

    # ------------------------------------------------------------------ use
    @torch.inference_mode()
    def set_conditioning(self, ehs0=None, ehs1=None, bias0=None, bias1=None, class_labels=None):
        """Copy conditioning (already laid out [B, L, dim]) into the engine and run the per-prompt
        precompute tape (cross-attention K/V projections, class embedding) -- once per prompt set."""
        for dst, src in ((self.ehs0, ehs0), (self.ehs1, ehs1), (self.bias0, bias0), (self.bias1, bias1),
                         (self.class_labels, class_labels)):
            if dst is not None and src is not None:
                dst.copy_(src.to(dst.device, torch.float32).reshape(dst.shape))
        self.ctx_tape.run()

    def set_timestep(self, t):
        """Immediate-timestep mode (no device table): patch the time-embedding op."""
        self.tape.ops[self.time_op].i[4] = int(t)
        arr = self.tape.finalize()
        arr[self.time_op].i[4] = int(t)

    @torch.inference_mode()
    def forward(self, first_half_only=False, second_half_only=False):
        if second_half_only:
            self.tape.run(self.mid_index, None)
        elif first_half_only:
            self.tape.run(0, self.mid_index)
        else:
            self.tape.run()
        return self.eps
