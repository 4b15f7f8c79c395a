"""LISAForCausalLM.model_forward for MI355X — the training / validation forward of the fine-tune loop.

`LisaTrainable.forward(**batch_dict)` takes the dict built by the reference's collate_fn
(2Haff/utils/dataset.py:152-169) and returns what `LISAForCausalLM.forward(**kwargs)` -> `model_forward` returns
(2Haff/model/LISA.py:170-430): the dict {loss, ce_loss, taxonomy_ce_loss, mask_bce_loss, mask_dice_loss, mask_loss},
or with inference=True {pred_masks_left, pred_masks_right, pred_taxonomies, gt_masks_left, gt_masks_right,
gt_taxonomies}. Trainable set = train_ds.py:192-244: LoRA (r, alpha, dropout) on q_proj/v_proj of every Llama layer +
embed_tokens, lm_head, text_hidden_fcs, mask_decoder_left/right. SAM encoder, CLIP tower, projector and the Llama
base weights are frozen; base weights keep a resident transposed copy for the dX products (288 GB of HBM: +13.5 GB
for 7B is cheaper than re-transposing, and no activation checkpointing is needed either).
Every op is an autograd.Function over HIP kernels (autograd.py).
"""
import math
from collections import OrderedDict

import torch

from . import autograd as A
from . import ops
from .lisa import IMAGE_TOKEN_INDEX, N_IMG_PAD, LisaMI355

V = "model.visual_model"
ACT_GELU, ACT_RELU = 1, 3


def _pad8(n):
    return (n + 7) // 8 * 8


class LisaTrainable:
    def __init__(self, cfg, state_dict, dtype=torch.bfloat16, device="cuda:0", lora_r=8, lora_alpha=16, lora_dropout=0.05,
                 ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0, seed=0, lora_init_b_zero=True):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.base = LisaMI355(cfg, state_dict, dtype=dtype, device=device, fp32_tail=False)  # training is bf16 end to end, as the reference's
        self.lora_r, self.lora_scale, self.lora_dropout = lora_r, lora_alpha / lora_r, lora_dropout
        # peft draws one dropout mask per adapted Linear: q_proj's and v_proj's adapters see independently dropped inputs (the
        # reference's semantics; default since round 5). False: ONE mask per layer for both adapters (rounds 3-4: same marginal
        # distribution, one mask launch / one rank product / one dx pass less per layer: +1.6 % samples/s)
        self.independent_lora_dropout = True
        self.w_ce, self.w_dice, self.w_bce = ce_loss_weight, dice_loss_weight, bce_loss_weight
        self.training = True
        self.overlap_sam = True   # False: the SAM encoder on the caller's stream, in front of everything else (A/B)
        sd, dev = state_dict, self.device
        P = self.params = OrderedDict()

        def add(name, t, keep_f32=False):
            t = t.detach().to(dev, torch.float32 if keep_f32 else dtype).clone().contiguous().requires_grad_(True)
            P[name] = t
            return t
        # full fine-tune tensors (train_ds.py:233-244)
        add("model.embed_tokens.weight", sd["model.embed_tokens.weight"])
        add("lm_head.weight", sd["lm_head.weight"])
        for k in ("model.text_hidden_fcs.0.0", "model.text_hidden_fcs.0.2"):
            add(k + ".weight", sd[k + ".weight"])
            add(k + ".bias", sd[k + ".bias"], keep_f32=True)
        for k, t in sd.items():
            if k.startswith(V + ".mask_decoder_left.") or k.startswith(V + ".mask_decoder_right."):
                is_vec = t.dim() == 1  # biases and norm gains/offsets are consumed as fp32 vectors by the kernels
                add(k, t, keep_f32=is_vec)
        # LoRA adapters (peft: A ~ kaiming_uniform(a=sqrt(5)), B = 0)
        g = torch.Generator(device="cpu").manual_seed(seed)
        H = cfg.llm.hidden
        for i in range(cfg.llm.layers):
            for n in ("q_proj", "v_proj"):
                k = f"model.layers.{i}.self_attn.{n}"
                bound = 1.0 / math.sqrt(H)
                a = (torch.rand((lora_r, H), generator=g) * 2 - 1) * bound
                b = torch.zeros((H, lora_r)) if lora_init_b_zero else (torch.rand((H, lora_r), generator=g) * 2 - 1) * 0.05
                add(k + ".lora_A", a)
                add(k + ".lora_B", b)
        # frozen Llama base: resident transposed copies for dX = dY . W
        self.wt = []
        for L in self.base.llm.layers:
            self.wt.append({n: A.transpose(L[n])[0] for n in ("wqkv", "wo", "wgu", "wd")})

    # -- parameter plumbing ------------------------------------------------------------------------------------------
    def parameters(self):
        return list(self.params.values())

    def named_parameters(self):
        return list(self.params.items())

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def zero_grad(self):
        for p in self.params.values():
            p.grad = None

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.params.items())

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, v in sd.items():
                self.params[k].copy_(v.to(self.params[k].dtype))

    # -- Llama with LoRA ---------------------------------------------------------------------------------------------
    def _llm(self, x, B, T):
        """x [B*T, H] embeddings -> post-norm hidden [B*T, H] (LlamaModel.forward with peft LoRA on q/v)."""
        l = self.cfg.llm
        llm = self.base.llm
        H, nh, hd = l.hidden, l.heads, llm.hd
        cs = llm._cos_sin(T)
        P = self.params
        for i, L in enumerate(llm.layers):
            wt = self.wt[i]
            # (x, norm(x)) as one node: the adjoint adds the residual branch's gradient of x inside the norm kernel
            if A.FUSED_RESID_NORM:
                x, h = A.resid_rmsnorm(x, L["n1"], l.rms_eps)
            else:
                h = A.rmsnorm(x, L["n1"], l.rms_eps)
            pre = f"model.layers.{i}.self_attn."
            drop = self.lora_dropout if self.training else 0.0
            if A.FUSED_LORA_QKV and A.lora_qkv_rope_supported(h, L["wqkv"], P[pre + "q_proj.lora_A"], nh):
                # one node: q|k|v product, both rank-r updates, RoPE (csrc/lora.hip); the dropout mask as 0 / 1 values from one
                # Bernoulli launch, its 1/(1-p) folded into the adapter scale
                keep = None
                if drop > 0:   # peft: one lora_dropout module per adapted Linear (train_ds.py:218-230): two independent masks
                    n_masks = 2 if self.independent_lora_dropout else 1   # (one Bernoulli launch either way)
                    masks = torch.empty((n_masks,) + tuple(h.shape), dtype=h.dtype, device=h.device).bernoulli_(1.0 - drop)
                    keep = (masks[0], masks[1]) if n_masks == 2 else masks[0]
                q, k, v = A.lora_qkv_rope(h, L["wqkv"], wt["wqkv"], P[pre + "q_proj.lora_A"], P[pre + "q_proj.lora_B"],
                                          P[pre + "v_proj.lora_A"], P[pre + "v_proj.lora_B"], cs, T, nh,
                                          self.lora_scale / (1.0 - drop), keep)
            else:
                qkv = A.linear(h, L["wqkv"], None, None, wt["wqkv"])
                hl = hv = h
                if drop > 0:
                    draw = lambda: (torch.rand(h.shape, device=h.device) >= drop).to(h.dtype) / (1 - drop)   # noqa: E731
                    hl = DropoutMul.apply(h, draw())
                    hv = DropoutMul.apply(h, draw()) if self.independent_lora_dropout else hl
                dq = A.linear(A.linear(hl, P[pre + "q_proj.lora_A"]), _pad_k(P[pre + "q_proj.lora_B"]))
                dv = A.linear(A.linear(hv, P[pre + "v_proj.lora_A"]), _pad_k(P[pre + "v_proj.lora_B"]))
                q = A.add(qkv[:, :H], A.scale(dq, self.lora_scale))
                k = qkv[:, H:2 * H]
                v = A.add(qkv[:, 2 * H:], A.scale(dv, self.lora_scale))
                q = A.rope(q, cs, T, nh, hd)
                k = A.rope(k, cs, T, nh, hd)
            a = A.attention(q.view(B, T, H), k.view(B, T, H), v.view(B, T, H), nh, hd ** -0.5, True)
            x = A.linear(a.view(B * T, H), L["wo"], None, x, wt["wo"])
            if A.FUSED_RESID_NORM:
                x, h = A.resid_rmsnorm(x, L["n2"], l.rms_eps)
            else:
                h = A.rmsnorm(x, L["n2"], l.rms_eps)
            gu = A.linear(h, L["wgu"], None, None, wt["wgu"])
            x = A.linear(A.swiglu(gu), L["wd"], None, x, wt["wd"])
        return A.rmsnorm(x, llm.norm, l.rms_eps)

    # -- one mask decoder (MaskDecoder.predict_masks, mask_decoder.py:122-170; TwoWayTransformer, transformer.py) -------
    def _attn(self, pfx, q_in, k_in, v_in, Pn, nq, nk, heads=8):
        P = self.params
        qp = A.linear(q_in, P[pfx + ".q_proj.weight"], P[pfx + ".q_proj.bias"])
        kp = A.linear(k_in, P[pfx + ".k_proj.weight"], P[pfx + ".k_proj.bias"])
        vp = A.linear(v_in, P[pfx + ".v_proj.weight"], P[pfx + ".v_proj.bias"])
        C = qp.shape[1]
        d = C // heads
        o = A.attention(qp.view(Pn, nq, C), kp.view(Pn, nk, C), vp.view(Pn, nk, C), heads, 1.0 / math.sqrt(d), False)
        return o.view(Pn * nq, C), (P[pfx + ".out_proj.weight"], P[pfx + ".out_proj.bias"])

    def _ln(self, name, x, eps=1e-5):
        return A.layernorm(x, self.params[name + ".weight"], self.params[name + ".bias"], eps)

    def _mlp3(self, name, x):
        P = self.params
        x = A.act(A.linear(x, P[f"{name}.layers.0.weight"], P[f"{name}.layers.0.bias"]), ACT_RELU)
        x = A.act(A.linear(x, P[f"{name}.layers.1.weight"], P[f"{name}.layers.1.bias"]), ACT_RELU)
        return A.linear(x, P[f"{name}.layers.2.weight"], P[f"{name}.layers.2.bias"])

    def _decoder(self, side, src, text, taxonomy_on):
        """src [Pn, N, C] (constant: frozen image embedding + no_mask_embed), text [Pn, C] (differentiable)."""
        D = f"{V}.mask_decoder_{side}"
        P = self.params
        Pn, N, C = src.shape
        g = self.cfg.sam.grid
        nt = 6
        key_pe = self.base.sam_decoder.key_pe
        out_tok = torch.cat([P[D + ".iou_token.weight"], P[D + ".mask_tokens.weight"]], dim=0)
        tokens = torch.cat([out_tok.unsqueeze(0).expand(Pn, -1, -1), text.view(Pn, 1, C)], dim=1).reshape(Pn * nt, C)
        queries, keys = tokens, src.reshape(Pn * N, C)
        T_ = D + ".transformer"
        for li in range(2):
            L = f"{T_}.layers.{li}"
            if li == 0:
                a, (wo, bo) = self._attn(L + ".self_attn", queries, queries, queries, Pn, nt, nt)
                queries = A.linear(a, wo, bo)
            else:
                q = A.add(queries, tokens)
                a, (wo, bo) = self._attn(L + ".self_attn", q, q, queries, Pn, nt, nt)
                queries = A.linear(a, wo, bo, queries)
            queries = self._ln(L + ".norm1", queries)
            q = A.add(queries, tokens)
            k = A.add_const(keys, key_pe, N)
            a, (wo, bo) = self._attn(L + ".cross_attn_token_to_image", q, k, keys, Pn, nt, N)
            queries = self._ln(L + ".norm2", A.linear(a, wo, bo, queries))
            h = A.act(A.linear(queries, P[L + ".mlp.lin1.weight"], P[L + ".mlp.lin1.bias"]), ACT_RELU)
            queries = self._ln(L + ".norm3", A.linear(h, P[L + ".mlp.lin2.weight"], P[L + ".mlp.lin2.bias"], queries))
            q = A.add(queries, tokens)
            a, (wo, bo) = self._attn(L + ".cross_attn_image_to_token", k, q, queries, Pn, N, nt)
            keys = self._ln(L + ".norm4", A.linear(a, wo, bo, keys))
        q = A.add(queries, tokens)
        k = A.add_const(keys, key_pe, N)
        a, (wo, bo) = self._attn(T_ + ".final_attn_token_to_image", q, k, keys, Pn, nt, N)
        queries = self._ln(T_ + ".norm_final_attn", A.linear(a, wo, bo, queries))
        hs = queries.view(Pn, nt, C)
        # output_upscaling (mask_decoder.py:54-64): both k=s=2 transposed convs are per-pixel GEMMs
        w1 = P[D + ".output_upscaling.0.weight"].permute(2, 3, 1, 0).reshape(C, C)          # [(dy,dx,co), ci]
        b1 = P[D + ".output_upscaling.0.bias"].repeat(4)
        u = A.linear(keys, w1, b1).view(Pn * N * 4, C // 4)
        u = A.act(A.layernorm(u, P[D + ".output_upscaling.1.weight"], P[D + ".output_upscaling.1.bias"], 1e-6), ACT_GELU)
        w2 = P[D + ".output_upscaling.3.weight"].permute(2, 3, 1, 0).reshape(4 * (C // 8), C // 4)  # [(dy2,dx2,c2), co]
        b2 = P[D + ".output_upscaling.3.bias"].repeat(4)
        u = A.act(A.linear(u, w2, b2), ACT_GELU).view(Pn, N * 16, C // 8)
        hyper0 = self._mlp3(D + ".output_hypernetworks_mlps.0", hs[:, 1]).view(Pn, 1, C // 8)
        m = A.bmm_nt(hyper0, u)                                   # [Pn, 1, N*16], pixel order (y, x, dy, dx, dy2, dx2)
        m = A.cast(m, torch.float32).view(Pn, g, g, 2, 2, 2, 2).permute(0, 1, 3, 5, 2, 4, 6).reshape(Pn, 4 * g, 4 * g)
        tax_logits = None
        if taxonomy_on:
            tax_logits = A.cast(self._mlp3(D + ".taxonomy_embed", hs[:, 1:5].reshape(Pn, 4 * C)), torch.float32)
        return m, tax_logits

    # -- model_forward -----------------------------------------------------------------------------------------------
    def forward(self, images, images_clip, input_ids, labels, attention_masks, offset, masks_list_left, masks_list_right,
                taxonomies_list, label_list, resize_list, inference=False, **kwargs):
        cfg, dev = self.cfg, self.device
        base = self.base
        # Host-side bookkeeping FIRST, from host copies of the small integer inputs (one early read of input_ids / offset /
        # taxonomies when they live on the device): image-token positions, [SEG] rows, prompts per frame, loss weights. Read
        # back mid-forward (int(argmax), nonzero, tolist, .cpu()) each of them drained the launch queue — behind the Llama
        # forward the ~700 small launches of the two mask decoders then went out one by one with the GPU waiting on the host.
        ids_host, off_host = input_ids.detach().cpu(), [int(v) for v in offset.detach().cpu().tolist()]
        tax_host = taxonomies_list.detach().float().cpu()
        img_pos = [int((ids_host[b] == IMAGE_TOKEN_INDEX).int().argmax()) for b in range(ids_host.shape[0])]
        seg_host = ids_host[:, 1:] == cfg.seg_token_idx
        seg_host = torch.cat([torch.zeros((ids_host.shape[0], N_IMG_PAD), dtype=torch.bool), seg_host,
                              torch.zeros((ids_host.shape[0], 1), dtype=torch.bool)], dim=1)
        b_idx_h, t_idx_h = seg_host.nonzero(as_tuple=True)
        counts_h = seg_host.int().sum(-1)
        seg_off = [int(v) for v in torch.cat([torch.zeros(1, dtype=torch.long), counts_h.cumsum(-1)], 0)[off_host].tolist()]
        lab_host = labels.detach().cpu()
        bsz = len(off_host) - 1
        lab = []
        for b in range(ids_host.shape[0]):   # llava_arch.py:185-208 on the labels: image rows are -100; then shift by one
            p = img_pos[b]
            lb = torch.cat([lab_host[b, :p], torch.full((N_IMG_PAD + 1,), -100, dtype=lab_host.dtype), lab_host[b, p + 1:]])
            lab.append(torch.cat([lb[1:], torch.full((1,), -100, dtype=lab_host.dtype)]))
        lab = torch.stack(lab).reshape(-1)
        n_valid = int((lab >= 0).sum())
        # ... and every host -> device copy of the step up front as well (a pageable copy waits for the stream it is ordered on)
        input_ids, lab_dev = input_ids.to(dev), lab.to(dev)
        b_idx, t_idx = b_idx_h.to(dev), t_idx_h.to(dev)
        frame_idx = torch.tensor([i for i in range(bsz) for _ in range(seg_off[i + 1] - seg_off[i])], dtype=torch.long).to(dev)
        gt_l = torch.stack([t.to(dev) for t in masks_list_left], 0).float()
        gt_r = torch.stack([t.to(dev) for t in masks_list_right], 0).float()
        gt_tax = taxonomies_list.to(dev).float()
        # The frozen SAM encoder feeds only the mask decoders: it runs on the model's side stream beside the CLIP tower and
        # the Llama forward, whose M = conversations x tokens products leave CUs idle (2808 x 4096 outputs = 176 tiles of
        # 256 x 256 on 256 CUs), and is joined in front of the decoders.
        cur = torch.cuda.current_stream(dev)
        sam_stream = base._sam_stream if (self.overlap_sam and dev.type == "cuda") else cur
        with torch.no_grad():
            images = images.to(dev)
            if sam_stream is not cur:
                sam_stream.wait_stream(cur)
            with torch.cuda.stream(sam_stream):
                emb = base.get_visual_embs(images)                                # frozen SAM encoder (LISA.py:191)
            n_conv = input_ids.shape[0]
            reps = [off_host[i + 1] - off_host[i] for i in range(len(off_host) - 1)]
            clip_rep = torch.cat([images_clip[i:i + 1].expand(r, -1, -1, -1) for i, r in enumerate(reps)], 0)
            img = base.encode_images(clip_rep)                                     # frozen CLIP + projector
        assert emb.shape[0] == bsz
        L = input_ids.shape[1]
        T = L + N_IMG_PAD
        # splice (llava_arch.py:185-208): [embed(ids[:p]) ; image features ; embed(ids[p+1:])]
        tok = A.embed(self.params["model.embed_tokens.weight"], input_ids)       # sentinel rows are dropped below
        rows = []
        for b in range(n_conv):
            p = img_pos[b]
            rows.append(torch.cat([tok[b, :p], img[b], tok[b, p + 1:]], dim=0))
        x = torch.stack(rows, 0).reshape(n_conv * T, cfg.llm.hidden)
        hidden = self._llm(x, n_conv, T)
        out = {}
        if not inference:
            logits = A.linear(hidden, self.params["lm_head.weight"])
            ce = A.cross_entropy(logits, lab_dev, n_valid)
        # [SEG] rows (LISA.py:195-207) and text_hidden_fcs on those rows only
        sel = hidden.view(n_conv, T, -1)[b_idx, t_idx]
        P = self.params
        h = A.act(A.linear(sel, P["model.text_hidden_fcs.0.0.weight"], P["model.text_hidden_fcs.0.0.bias"]), ACT_RELU)
        pred = A.linear(h, P["model.text_hidden_fcs.0.2.weight"], P["model.text_hidden_fcs.0.2.bias"])
        Pn = pred.shape[0]
        N, C = emb.shape[1], emb.shape[2]
        if sam_stream is not cur:
            cur.wait_stream(sam_stream)
            emb.record_stream(cur)
        with torch.no_grad():
            src = emb.index_select(0, frame_idx).reshape(Pn * N, C)
            src = ops.add_bcast(src, base.sam_decoder.no_mask, mod=1).view(Pn, N, C)
        lo_l, tax_logits = self._decoder("left", src, pred, True)
        lo_r, _ = self._decoder("right", src, pred, False)
        S = cfg.sam.img_size
        pl, pr = [], []
        for i in range(bsz):
            a, b = seg_off[i], seg_off[i + 1]
            for lo, dst, side in ((lo_l, pl, "left"), (lo_r, pr, "right")):
                up = A.resize_bilinear(lo[a:b], lo.shape[-2:], (S, S))
                dst.append(A.resize_bilinear(up, resize_list[i], tuple(label_list[i][side].shape)))
        tax_loss_rows, tax_probs = A.taxonomy_ce(tax_logits, gt_tax[frame_idx])
        if inference:
            return {"pred_masks_left": torch.stack(pl, 0), "pred_masks_right": torch.stack(pr, 0),
                    "pred_taxonomies": torch.stack([tax_probs[seg_off[i]:seg_off[i + 1]] for i in range(bsz)]),
                    "gt_masks_left": gt_l, "gt_masks_right": gt_r, "gt_taxonomies": gt_tax}
        # losses (LISA.py:346-430)
        w_l = (tax_host[:, 0] + tax_host[:, 2] + tax_host[:, 3]).tolist()
        w_r = (tax_host[:, 1] + tax_host[:, 2] + tax_host[:, 3]).tolist()
        num_masks = 0
        bce_l = bce_r = dice_l = dice_r = 0.0
        for i in range(bsz):
            n = gt_l[i].shape[0]
            hw = gt_l[i][0].numel()
            ll = A.mask_losses(pl[i].reshape(n, hw), gt_l[i].reshape(n, hw), [w_l[i]] * n)
            lr = A.mask_losses(pr[i].reshape(n, hw), gt_r[i].reshape(n, hw), [w_r[i]] * n)
            # sigmoid_ce_loss / dice_loss: sum over masks / (num_masks + 1e-8) * num_masks  (LISA.py:394-410)
            bce_l = bce_l + ll[:, 0].sum() / (n + 1e-8) * n
            dice_l = dice_l + ll[:, 1].sum() / (n + 1e-8) * n
            bce_r = bce_r + lr[:, 0].sum() / (n + 1e-8) * n
            dice_r = dice_r + lr[:, 1].sum() / (n + 1e-8) * n
            num_masks += n
        tax_ce = tax_loss_rows.sum() / bsz
        mask_bce = self.w_bce * bce_l / (num_masks + 1e-8) + self.w_bce * bce_r / (num_masks + 1e-8)
        mask_dice = self.w_dice * dice_l / (num_masks + 1e-8) + self.w_dice * dice_r / (num_masks + 1e-8)
        ce = ce * self.w_ce
        mask_loss = mask_bce + mask_dice
        return {"loss": ce + mask_loss + tax_ce, "ce_loss": ce, "taxonomy_ce_loss": tax_ce, "mask_bce_loss": mask_bce,
                "mask_dice_loss": mask_dice, "mask_loss": mask_loss}

    __call__ = forward


def _pad_k(w):
    """LoRA B is [H, r]; the GEMM wants K % 8 == 0 — r is 8 by default, pad otherwise (zeros)."""
    r = w.shape[1]
    if r % 8 == 0:
        return w
    return torch.nn.functional.pad(w, (0, _pad8(r) - r))


class DropoutMul(torch.autograd.Function):
    """x * keep_mask (mask values 0 or 1/(1-p)) — peft's lora_dropout on the adapter input."""

    @staticmethod
    def forward(ctx, x, keep):
        ctx.save_for_backward(keep)
        return _mul(x, keep)

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        return _mul(dy, keep), None


def _mul(a, b):
    from .lib import check, load_library
    lib = load_library()
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    check(lib.haff_mul(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), A._dt(a), A._s()), "haff_mul")
    return out
