"""LISAForCausalLM for MI355X — the drop-in boundary of the 2Haff hot path.

`LisaMI355.evaluate(images_clip, images, input_ids, resize_list, original_size_list, max_new_tokens=32,
tokenizer=None)` has the signature and return tuple of the reference's `LISAForCausalLM.evaluate`
(2Haff/model/LISA.py:432-534): (output_ids, pred_masks_left, pred_masks_right, taxonomies); `predict` is
the alias BASELINE.json's north_star names. All numerics run in the HIP kernels of csrc/ through the C-ABI;
torch only owns HBM allocations, views and the stream.

Differences from the reference that do NOT change results:
  * KV-cached greedy decode (reference: use_cache=False, LISA.py:115, recomputes CLIP + prefix every token);
  * text_hidden_fcs runs only on the [SEG]-selected rows (reference: on all T positions, then masks, :467-473);
  * frames / prompts are batched through the SAM encoder and decoders (reference: python loops, :157-168,494-532).
Extension for synthetic benchmarking: `forced_answer` overrides the appended tokens (argmax still computed).
"""
import contextlib
import json
import os

import torch

from . import ops, overlap
from .llava import ClipTowerHip, LlamaHip, _f32
from .sam import SamEncoderHip, SamPromptDecoderHip

IMAGE_TOKEN_INDEX = -200
N_IMG_PAD = 255  # LISA.py:461


class LisaMI355:
    def __init__(self, cfg, state_dict, dtype=torch.bfloat16, device="cuda:0", sam_chunk=8, fp32_tail=True, fp32_stream=False,
                 neck_f32=False):
        if not torch.cuda.is_available():
            raise RuntimeError("LisaMI355 needs an MI355X (HIP device); there is no CPU fallback for the hot path")
        from .lib import load_library
        load_library()  # fail loudly if the HIP library is absent
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.seg_token_idx = cfg.seg_token_idx
        self.sam_chunk = sam_chunk
        self._sam_stream = torch.cuda.Stream(device=self.device)
        # How the encoder's stream shares the CUs with the decode steps (overlap.py): "auto" = the work-model plan per
        # evaluate() call, None = every launch on all CUs, or an explicit list of workgroup caps per encoder chunk;
        # sam_waits_for_prefill "auto" | True | False (late mode: the encoder starts behind the prefill on the GPU too).
        self.sam_chunk_caps = "auto"
        self.sam_waits_for_prefill = "auto"
        self.calibrate_overlap = True         # False: the plan uses overlap.NOMINAL (round 5's fitted constants) whatever the device
        self.last_rates = None
        self.expected_new_tokens = 8          # what the plan assumes a reply takes ("Sure, ... [SEG] ." templates) when max_new_tokens is larger
        self._plan = (None, False, None)      # (caps, wait, chunk) of the evaluate() call in flight (no caps outside one)
        self.last_plan = (None, False, None)  # ... of the last evaluate() call (what bench.py reports)
        self.last_decode_chain = False
        # The SAM encoder runs on its own HIP stream beside the language model and joins before the mask decoders
        # (+4-5 % frames/s); False serialises everything on the caller's stream (per-kernel measurements). Results are
        # bit-identical either way (tests/test_fullsize_gpu.py; the history of that check: DESIGN.md section 10a).
        self.overlap_streams = True
        self.sam_beside_decode = None   # None: by batch size (<= 4 frames: encoder enqueued behind the prefill); True / False force
        # KV-cached decode steps are launch-bound at small batch (32 layers x 9 launches per token): each step is
        # captured once per (batch, position) into a hipGraph and replayed. The KV cache is persistent per
        # (batch, capacity) so the captured pointers stay valid across evaluate() calls.
        self.decode_graphs = True
        # Decode steps of <= 8 rows as ONE chained launch per step (LlamaHip.decode_chain, csrc/decode_chain.hip). "auto": below
        # overlap.MIN_FRAMES (4) frames per call (round 6, same box: decode step 3.25 -> 3.11 ms at one row, one frame 41.3 -> 40.6 ms,
        # three frames 57.8 -> 56.7). From 4 frames on the plan caps the encoder to 128..192 CUs beside the decode steps, and there a
        # chained step's 80 000 workgroups keep refilling every slot of every CU while the encoder's one-per-CU launches wait — 4 / 8
        # frames lose 20..25 % — so the five short launches per layer stay. The choice depends on the batch alone, not on the schedule
        # (streams, caps): every schedule of a batch gives the same bits. True: chained at any batch <= 8 (A/B); False: never.
        # generate() called directly follows LlamaHip.decode_chain.
        self.decode_chain = "auto"
        # evaluate(): the last Llama layer of the prefill runs o_proj / MLP / final norm on the rows that are read only (LlamaHip.forward,
        # keep_rows; round 6). False: every row (A/B; generate() called directly always returns every row)
        self.prune_last_layer = True
        self._ingest = None
        self._graphs = {}
        # one memory pool shared by every captured step. Replays never overlap: every graph is replayed on the caller's stream,
        # one after another — the split-K workspaces baked into them (ops._workspace) may be the same buffer
        self._graph_pool = None
        self._caches = {}
        sd, dev = state_dict, self.device
        assert cfg.clip.n_patches == N_IMG_PAD + 1, "the reference hard-codes 256 image tokens (LISA.py:461)"
        # fp32 decoder tail (throughput mode): image embeddings leave the neck in fp32 and text_hidden_fcs, the prompt
        # encoder, both two-way mask decoders, the hypernetwork / IoU / taxonomy MLPs and the upscaler run on the
        # f32-input matrix cores (csrc/gemm_f32.hip) — 7 GFLOP of the 10 TFLOP per frame (hard part 3).
        # The ViT-H / CLIP / Llama stacks stay bf16 MFMA. False = the all-bf16 path of round 1.
        self.fp32_tail = fp32_tail and dtype == torch.bfloat16
        tail = torch.float32 if self.fp32_tail else dtype
        self.tail_dtype = tail
        self.sam_encoder = SamEncoderHip(sd, cfg.sam, dtype, dev)
        self.sam_encoder.emb_f32 = self.fp32_tail
        self.sam_decoder = SamPromptDecoderHip(sd, cfg.sam, tail, dev)
        self.clip = ClipTowerHip(sd, cfg.clip, dtype, dev)
        self.llm = LlamaHip(sd, cfg.llm, dtype, dev)
        # fp32 residual streams in the bf16 mode: True / "sam" / "llm" — the ViT-H and / or Llama hidden-state
        # stream kept in fp32 between the bf16 MFMA products (2.8x closer to the reference on the image embedding at depth 32)
        self.sam_encoder.fp32_stream = fp32_stream in (True, "sam", "both")
        self.llm.fp32_stream = fp32_stream in (True, "llm", "both")
        self.sam_encoder.neck_f32 = bool(neck_f32) and dtype == torch.bfloat16    # the ViT-H neck on the f32-input MFMA path (sam.py)
        self.w_proj = sd["model.mm_projector.weight"].to(dev, dtype).contiguous()
        self.b_proj = _f32(sd["model.mm_projector.bias"], dev)
        self.fc0 = (sd["model.text_hidden_fcs.0.0.weight"].to(dev, tail).contiguous(), _f32(sd["model.text_hidden_fcs.0.0.bias"], dev))
        self.fc2 = (sd["model.text_hidden_fcs.0.2.weight"].to(dev, tail).contiguous(), _f32(sd["model.text_hidden_fcs.0.2.bias"], dev))

    # ---- a4/a5: CLIP tower + projector -------------------------------------------------------------------
    def encode_images(self, images_clip):
        B = images_clip.shape[0]
        images_clip = images_clip.to(self.device)

        def fn(x):
            return (self.clip.project(self.clip.hidden(x), B, self.w_proj, self.b_proj),)
        if not self.decode_graphs or B > 4:
            return fn(images_clip)[0]
        return self._replay(("clip", B, images_clip.dtype, tuple(images_clip.shape[1:])), fn, [images_clip.contiguous()])[0]

    # ---- a6-a8: splice + greedy generate -----------------------------------------------------------------
    @torch.no_grad()
    def generate(self, images_clip, input_ids, max_new_tokens=32, forced_answer=None, attention_mask=None, after_prefill=None,
                 needed_hidden_only=False):
        """Greedy KV-cached decode (LISA.py:443-450). Rows may have different prompt lengths: input_ids is right-padded
        (utils/dataset.py:90-93) and attention_mask [B, L] (bool, True on real tokens; None = every row is full length)
        marks the real prefix of each row, as collate_fn builds it (:144-150). Row b's generated tokens are appended right
        after ITS last real token, so output_ids [B, L + N] is left-aligned per row (pad_token_id behind the end) and the
        hidden states [B, T + N - 1, H] line up with it position by position — the [SEG] rule of LISA.py:457-465 then
        applies per row unchanged. Equal to B independent batch-1 runs."""
        cfg = self.cfg
        input_ids = input_ids.to(self.device)
        B, L = input_ids.shape
        is_img = input_ids == IMAGE_TOKEN_INDEX
        assert bool((is_img.sum(1) == 1).all()), "exactly one <image> sentinel per row (LISA.py:458 hack)"
        img_pos = is_img.int().argmax(1).to(torch.int32)
        if attention_mask is None:
            lens = torch.full((B,), L, dtype=torch.int64, device=self.device)
        else:
            am = attention_mask.to(self.device).bool()
            lens = am.sum(1)
            assert bool((am == (torch.arange(L, device=self.device)[None, :] < lens[:, None])).all()), "right padding only"
            assert bool((img_pos.long() < lens).all())
        img = self.encode_images(images_clip)
        n_img = img.shape[1]
        # the sentinel slot itself is never dereferenced by the splice kernel
        x = ops.embed_splice(input_ids.clamp_min(IMAGE_TOKEN_INDEX).contiguous(), img_pos, self.llm.embed, img.contiguous())
        T = L + n_img - 1
        t_rows = lens + (n_img - 1)                      # real positions per row after the splice
        cache = self._persistent_cache(B, T + max_new_tokens)
        st = cache["book"]
        keep = None
        if needed_hidden_only and self.prune_last_layer:
            # evaluate() reads two kinds of prefill rows only: each row's last real position (the first token's logits) and the
            # positions in front of a [SEG] that is part of the PROMPT (LISA.py:457-465 gathers the state that precedes the token);
            # the last Llama layer skips o_proj / MLP / norm for every other row (LlamaHip.forward, keep_rows)
            b_last = torch.arange(B, device=self.device) * T + (t_rows - 1)
            pb, pt = ((input_ids[:, 1:] == self.seg_token_idx) &
                      (torch.arange(1, L, device=self.device)[None, :] < lens[:, None])).nonzero(as_tuple=True)
            # id position j + 1 holds [SEG] -> seg_embeddings reads hidden row j + 255 (the reference's fixed 255-row shift, wherever
            # the <image> sentinel sits: LISA.py:459-463)
            keep = torch.unique(torch.cat([b_last, pb * T + pt + (n_img - 1)]))
        prefill = self.llm.forward(x, cache, keep_rows=keep)   # causal: a row's real positions never see its padding
        if after_prefill is not None:                    # work the caller wants enqueued (elsewhere) behind the prefill
            after_prefill()
        hidden, out_ids = st["hidden"], st["out_ids"]    # persistent [B, tmax, H] / [B, tmax]: the decode graph writes into them
        hidden[:, :T] = prefill
        hidden[:, T:].zero_()
        out_ids.fill_(cfg.pad_token_id)
        out_ids[:, :L] = torch.where(torch.arange(L, device=self.device)[None, :] < lens[:, None], input_ids,
                                     torch.full_like(input_ids, cfg.pad_token_id))
        if max_new_tokens <= 0:
            return out_ids[:, :L].clone(), hidden[:, :T - 1].clone()
        st["t_rows"].copy_(t_rows)
        st["lens"].copy_(lens)
        st["steps"].zero_()
        st["finished"].zero_()
        if forced_answer is not None:
            assert forced_answer.shape[0] == B and forced_answer.shape[1] >= max_new_tokens
            st["forced"][:, :max_new_tokens] = forced_answer[:, :max_new_tokens].to(self.device)
        st["use_forced"].fill_(0 if forced_answer is None else 1)
        rows = torch.arange(B, device=self.device)
        logits = self.llm.next_token_logits(prefill[rows, t_rows - 1].contiguous())
        # token 0 (from the prefill logits), then one decode step per further token: embedding of the token just written ->
        # layers at each row's own position -> logits -> argmax -> bookkeeping of the next token, all device-side; the host
        # only reads the rows' finished flags between steps (generate()'s stopping rule)
        ops.decode_book(ops.argmax_rows(logits), st, None, cfg.pad_token_id, cfg.eos_token_id)
        n_done = 1
        while n_done < max_new_tokens and not bool(st["finished"].all()):
            self._decode_book_step(cache)
            n_done += 1
        ch = cache.get("chain")
        if ch is not None and n_done > 1 and not ops.decode_chain_status(ch["sync"], len(self.llm.layers)):
            # a bounded wait of the chained decode launch ran out (csrc/decode_chain.hip): the tokens above are garbage — fail loudly
            ch["sync"].zero_()
            raise RuntimeError("chained decode launch: a stage's arrival counter never filled (sticky error word set); "
                               "set LlamaHip.decode_chain = False to take the five-launch layer")
        return out_ids[:, :L + n_done].clone(), hidden[:, :T + n_done - 1].clone()

    def _persistent_cache(self, B, tmax):
        key = (B, tmax)
        c = self._caches.get(key)
        if c is None:
            if len(self._caches) >= 4:   # a few shapes at most stay resident (7B, B=64, 299 positions: 10 GB)
                old = next(iter(self._caches))
                del self._caches[old]
                self._graphs = {k: v for k, v in self._graphs.items() if k[-2:] != old}
            c = self._caches[key] = self.llm.new_cache(B, tmax)
            dev, i64, i32 = self.device, torch.int64, torch.int32
            # generate()'s per-row state, persistent so that the decode hipGraph can address it (ops.decode_book)
            c["book"] = {"forced": torch.zeros((B, tmax), dtype=i64, device=dev), "use_forced": torch.zeros((1,), dtype=i32, device=dev),
                         "steps": torch.zeros((B,), dtype=i32, device=dev), "finished": torch.zeros((B,), dtype=torch.uint8, device=dev),
                         "out_ids": torch.zeros((B, tmax), dtype=i64, device=dev), "lens": torch.zeros((B,), dtype=i64, device=dev),
                         "t_rows": torch.zeros((B,), dtype=i32, device=dev), "tok": torch.zeros((B,), dtype=i64, device=dev),
                         "pos": c["pos"], "nk": c["nk"],
                         "hidden": torch.zeros((B, tmax, self.cfg.llm.hidden), dtype=self.dtype, device=dev)}
        c["len"] = 0
        return c

    def _decode_book_eager(self, cache):
        st = cache["book"]
        B = st["tok"].shape[0]
        x1 = self.llm.embed.index_select(0, st["tok"]).view(B, 1, -1)
        h1 = self.llm.decode_rows(x1, cache)
        logits = self.llm.next_token_logits(h1[:, -1])
        ops.decode_book(ops.argmax_rows(logits), st, h1.reshape(B, -1), self.cfg.pad_token_id, self.cfg.eos_token_id)

    def _decode_book_step(self, cache):
        """One greedy step of generate(): the token written last (cache["book"]["tok"]) is decoded at each row's position and
        the next token is chosen and filed — ONE hipGraph replay per step (per (batch, capacity): the per-row state lives in
        device memory), nothing else on the stream. The first step of a new shape runs eagerly (it is the warm-up of the
        lazily built state inside the ops) and is followed by the capture, which executes nothing."""
        if not self.decode_graphs:
            return self._decode_book_eager(cache)
        key = ("book", self.llm.decode_chain, cache["book"]["tok"].shape[0], cache["tmax"])   # (the last two: what _persistent_cache evicts by)
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 16:
                self._graphs.clear()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            self._decode_book_eager(cache)
            torch.cuda.current_stream(self.device).synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
                self._decode_book_eager(cache)
            self._graphs[key] = g
            return
        g.replay()

    def _decode_step_eager(self, nxt, cache):
        B = nxt.shape[0]
        x1 = self.llm.embed.index_select(0, nxt).view(B, 1, -1)
        h1 = self.llm.decode_rows(x1, cache)
        logits = self.llm.next_token_logits(h1[:, -1])
        return h1, ops.argmax_rows(logits)

    def _decode_step(self, nxt, cache):
        """One greedy step: embed(nxt) -> the layers against the KV cache at each row's own position -> logits -> argmax,
        then every row's position advances by one. Returns (hidden [B,1,H], next ids [B]). The per-row positions live in
        device memory (cache["pos"], cache["nk"]), so ONE hipGraph per (batch, cache capacity) serves every step."""
        if not self.decode_graphs:
            out = self._decode_step_eager(nxt, cache)
        else:
            B = nxt.shape[0]
            key = (self.llm.decode_chain, B, cache["tmax"])
            ent = self._graphs.get(key)
            if ent is None:
                if len(self._graphs) >= 16:
                    self._graphs.clear()
                if self._graph_pool is None:
                    self._graph_pool = torch.cuda.graph_pool_handle()
                static_in = nxt.clone()
                # warm-up outside the capture (lazy one-time work inside the ops); it rewrites this step's K/V slots only
                self._decode_step_eager(static_in, cache)
                torch.cuda.current_stream(self.device).synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
                    h1, out = self._decode_step_eager(static_in, cache)
                ent = self._graphs[key] = (g, static_in, h1, out)
            g, static_in, h1, out = ent
            static_in.copy_(nxt)
            g.replay()
            # outputs are the graph's static tensors: hand out copies so the next replay cannot overwrite them
            out = (h1.clone(), out.clone())
        cache["pos"].add_(1)
        cache["nk"].add_(1)
        return out

    # ---- a10: SAM image encoder --------------------------------------------------------------------------
    @contextlib.contextmanager
    def _chunk_cap(self, n, frames):
        """Encoder chunk n of a `frames`-frame step: the persistent GEMM launches enqueued inside take `sam_chunk_caps[n]`
        workgroups instead of one per CU (ops.gemm_stream_cap on the stream they are enqueued on), leaving CUs to the other stream's kernels. None: no cap;
        "auto": overlap.plan, decided per evaluate() call."""
        caps = self._plan[0]
        cap = caps[min(n, len(caps) - 1)] if caps and self.overlap_streams else 256
        if cap == 256:
            yield
            return
        old = ops.gemm_stream_cap(cap)
        try:
            yield
        finally:
            ops.gemm_stream_cap(old)

    def _chunk(self, frames):
        """Frames per encoder launch sequence: `sam_chunk`, or the plan's choice for sam_chunk="auto"."""
        c = self._plan[2] if self.sam_chunk == "auto" else self.sam_chunk
        return int(c) if c else max(1, min(int(frames), 8))

    def _encoder_passes(self, F, rows_of):
        """The ViT over F frames in passes of `_chunk(F)` frames; rows_of(i, n) -> the patch rows of frames i .. i+n. Every pass
        writes its slice of ONE embedding tensor (torch.cat of four 67 MB pieces was 0.7 ms at the end of the encoder)."""
        enc = self.sam_encoder
        ch = self._chunk(F)
        emb = torch.empty((F, enc.cfg.grid ** 2, enc.cfg.out_chans), dtype=torch.float32 if enc.emb_f32 else enc.dtype, device=self.device) \
            if F > ch else None
        for n, i in enumerate(range(0, F, ch)):
            k = min(ch, F - i)
            with self._chunk_cap(n, F):
                y = enc.forward_rows(rows_of(i, k), k, out=None if emb is None else emb[i:i + k])
        return y if emb is None else emb

    @torch.no_grad()
    def get_visual_embs(self, images):
        """LISA.py:157-168 without the per-image python loop; chunked to bound activation memory."""
        enc = self.sam_encoder
        return self._encoder_passes(images.shape[0], lambda i, k: enc.patch_rows_from_nchw(images[i:i + k].to(self.device, enc.dtype)))

    @torch.no_grad()
    def get_visual_embs_u8(self, frames, mean, std):
        enc = self.sam_encoder
        return self._encoder_passes(frames.shape[0], lambda i, k: enc.patch_rows_from_u8(frames[i:i + k], mean, std))

    @torch.no_grad()
    def get_visual_embs_frames(self, frames, mean, std):
        """A list of uint8 HWC frames of DIFFERENT sizes (a directory of images): each goes through its own Pillow-exact
        resize + fused normalise/pad/patchify (cheap, HBM-bound), the patch rows are concatenated and the ViT runs batched."""
        ing, enc = self.frame_ingest(), self.sam_encoder
        outs = []
        ch = self._chunk(len(frames))
        for n, i in enumerate(range(0, len(frames), ch)):
            rows = []
            for fr in frames[i:i + ch]:
                u8, _ = ing.sam_frames(fr.to(self.device)[None], self.cfg.sam.img_size)
                rows.append(enc.patch_rows_from_u8(u8, mean, std))
            with self._chunk_cap(n, len(frames)):
                outs.append(enc.forward_rows(torch.cat(rows, 0) if len(rows) > 1 else rows[0], len(rows)))
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]

    # ---- a9: [SEG] gather + text_hidden_fcs ---------------------------------------------------------------
    def seg_embeddings(self, output_ids, hidden):
        mask = output_ids[:, 1:] == self.seg_token_idx
        mask = torch.cat([torch.zeros((mask.shape[0], N_IMG_PAD), dtype=torch.bool, device=mask.device), mask], dim=1)
        assert mask.shape[1] == hidden.shape[1], (mask.shape, hidden.shape)
        counts = mask.int().sum(-1)
        b_idx, t_idx = mask.nonzero(as_tuple=True)
        if b_idx.numel() == 0:
            return torch.empty((0, self.cfg.out_dim), dtype=self.tail_dtype, device=self.device), b_idx, counts
        rows = hidden[b_idx, t_idx].to(self.tail_dtype).contiguous()
        h = ops.linear(rows, self.fc0[0], bias=self.fc0[1], act=ops.ACT_RELU)
        return ops.linear(h, self.fc2[0], bias=self.fc2[1]), b_idx, counts

    def frame_ingest(self):
        """Device-side host preprocessing (rows a1/a2): Pillow-exact resizes + CLIP normalisation on uint8 frames."""
        if self._ingest is None:
            from .preprocess import FrameIngest
            self._ingest = FrameIngest(self.device)
        return self._ingest

    # ---- the boundary --------------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, images_clip, images, input_ids, resize_list, original_size_list, max_new_tokens=32,
                 tokenizer=None, forced_answer=None, frames_u8=None, attention_mask=None):
        # The SAM encoder (MFMA-bound, ~60 % of the FLOPs) does not depend on the language model: it runs on its own
        # HIP stream so its big GEMMs fill the CUs while the HBM/latency-bound greedy decode steps of the LLM trickle
        # through on the caller's stream. Joined before the mask decoders.
        cur = torch.cuda.current_stream(self.device)
        side = self._sam_stream if self.overlap_streams else cur
        frame_list = isinstance(frames_u8, (list, tuple))   # frames of different sizes: one [H,W,3] uint8 tensor each
        if frames_u8 is not None:
            if not frame_list:
                frames_u8 = frames_u8.to(self.device)
            if images_clip is None:   # a2 on the device: CLIPImageProcessor.preprocess of the same uint8 frames
                ing = self.frame_ingest()
                if frame_list:
                    images_clip = torch.cat([ing.clip_pixels(f.to(self.device)[None], self.cfg.clip.image, self.dtype)
                                             for f in frames_u8], 0)
                else:
                    images_clip = ing.clip_pixels(frames_u8, self.cfg.clip.image, self.dtype)
        inputs_ready = torch.cuda.Event()
        inputs_ready.record(cur)
        sam_out = []

        def launch_sam():
            side.wait_event(inputs_ready)
            if late and wait:   # really BEHIND the prefill on the GPU too, not only in the host's enqueue order
                ev = torch.cuda.Event()
                ev.record(cur)
                side.wait_event(ev)
            with torch.cuda.stream(side):
                if frames_u8 is not None:
                    from .preprocess import SAM_MEAN, SAM_STD
                    # a1 on the device: ResizeLongestSide (identity when the long side is img_size), then the fused
                    # normalise + pad + patchify of haff_patchify_u8
                    if frame_list:
                        sam_out.append(self.get_visual_embs_frames(frames_u8, SAM_MEAN, SAM_STD))
                    else:
                        sam_u8, _ = self.frame_ingest().sam_frames(frames_u8, self.cfg.sam.img_size)
                        sam_out.append(self.get_visual_embs_u8(sam_u8, SAM_MEAN, SAM_STD))
                else:
                    sam_out.append(self.get_visual_embs(images))

        # Where the encoder is enqueued. Throughput batches: first, so its GEMMs run beside the CLIP tower and the prefill.
        # A few frames (latency): behind the prefill, so it runs beside the HBM-bound decode steps, which leave the matrix
        # cores idle, instead of time-slicing them with the prefill (and the host enqueues it while the prefill runs).
        # (round 5, bench.py --sam-beside-decode on / off at 8 and 16 frames: behind the prefill is +0.4 ... +1.8 % for 7B and 13B there
        # too; at 64 frames the encoder is 60 % of the step and has to start first)
        late = self.overlap_streams and self.sam_beside_decode is not False and \
            (self.sam_beside_decode is True or input_ids.shape[0] <= 16)
        # Which kernels decode: a function of the BATCH alone, never of the schedule (every schedule of one batch must give the same
        # bits: tests/test_lisa_gpu.py::test_stream_schedules_do_not_change_results, tools/two_stream_soak.py). "auto": the chained
        # launch below overlap.MIN_FRAMES frames — where no encoder pass is ever capped beside the decode steps — five launches per
        # layer from there on (the plan caps the encoder beside them; a chained step would starve it: the comment at decode_chain).
        n_rows = input_ids.shape[0]
        chain_now = bool(self.llm.decode_chain) and n_rows <= self.llm.carry_rms_max_rows and \
            (self.decode_chain is True or (self.decode_chain == "auto" and n_rows < overlap.MIN_FRAMES))
        if chain_now and self.decode_chain == "auto" and self.sam_beside_decode is None:
            # chained decode steps: the encoder FIRST (beside CLIP + prefill), the steps alone behind it — a chained step is one 3 ms
            # launch whose workgroups hold every CU, the encoder's launches cannot slip in between (same box, one frame: encoder
            # first + chain 40.6 ms, behind the prefill + five launches 41.3, behind the prefill + chain 41.8-42.1)
            late = False
        # ... and which CUs its GEMM launches leave to the decode steps (overlap.py)
        n_frames = input_ids.shape[0]
        chunk = overlap.auto_chunk(n_frames, late) if self.sam_chunk == "auto" else self.sam_chunk
        caps, wait = None, False
        if self.overlap_streams:
            if self.sam_chunk_caps == "auto":
                # the work model's rates follow THIS device (two timed probes, once per process: overlap.calibrate)
                rates = overlap.calibrate(self.device) if self.calibrate_overlap and n_frames >= overlap.MIN_FRAMES else overlap.NOMINAL
                self.last_rates = rates
                caps, wait = overlap.plan(self.cfg, n_frames, chunk, input_ids.shape[1],
                                          min(max_new_tokens, self.expected_new_tokens), late, rates)
            elif self.sam_chunk_caps:
                caps, wait = list(self.sam_chunk_caps), late
            if self.sam_waits_for_prefill != "auto":
                wait = bool(self.sam_waits_for_prefill) and late
        self._plan = self.last_plan = (caps, wait, chunk)
        chain_prev = self.llm.decode_chain
        self.llm.decode_chain = chain_prev if chain_now else False
        self.last_decode_chain = chain_now
        if not late:
            launch_sam()
        try:
            output_ids, hidden = self.generate(images_clip, input_ids, max_new_tokens, forced_answer, attention_mask,
                                               after_prefill=launch_sam if late else None, needed_hidden_only=True)
        finally:
            self.llm.decode_chain = chain_prev
        emb = sam_out[0]
        self._plan = (None, False, chunk)     # the encoder is enqueued: direct get_visual_embs* calls take no caps
        pred, frame_idx, counts = self.seg_embeddings(output_ids, hidden)
        cur.wait_stream(side)
        emb.record_stream(cur)
        B = output_ids.shape[0]
        pred_masks_left, pred_masks_right, taxonomies = [], [], []
        if pred.shape[0] > 0:
            lo_l, lo_r, tax = self._decoder_tail(emb, frame_idx, pred)
        offs = torch.cat([torch.zeros(1, dtype=torch.long), counts.cpu().long().cumsum(0)]).tolist()
        # frames of one size (a batch of equally sized frames: the throughput case): ONE postprocess launch per hand over all
        # prompts, sliced per frame below, instead of two launches per frame
        same = pred.shape[0] > 0 and all(tuple(r) == tuple(resize_list[0]) for r in resize_list) and \
            all(tuple(o) == tuple(original_size_list[0]) for o in original_size_list)
        if same:
            post_l = self.sam_decoder.postprocess(lo_l.contiguous(), resize_list[0], original_size_list[0])
            post_r = self.sam_decoder.postprocess(lo_r.contiguous(), resize_list[0], original_size_list[0])
        for i in range(B):
            a, b = offs[i], offs[i + 1]
            if same and b > a:
                pred_masks_left.append(post_l[a:b])
                pred_masks_right.append(post_r[a:b])
                taxonomies.append(tax[a:b])
                continue
            if b == a:
                h0, w0 = original_size_list[i]
                pred_masks_left.append(torch.empty((0, h0, w0), dtype=torch.float32, device=self.device))
                pred_masks_right.append(torch.empty((0, h0, w0), dtype=torch.float32, device=self.device))
                taxonomies.append(torch.empty((0, 4), dtype=torch.float32, device=self.device))
                continue
            pred_masks_left.append(self.sam_decoder.postprocess(lo_l[a:b].contiguous(), resize_list[i], original_size_list[i]))
            pred_masks_right.append(self.sam_decoder.postprocess(lo_r[a:b].contiguous(), resize_list[i], original_size_list[i]))
            taxonomies.append(tax[a:b])
        return output_ids, pred_masks_left, pred_masks_right, taxonomies

    predict = evaluate

    def _replay(self, key, fn, inputs):
        """fn(*static inputs) -> tuple of tensors, as ONE hipGraph per key (a few frames: these stages are ~150-190 small
        launches each and the host, not the GPU, sets their pace). First use runs eagerly (warm-up and result), then captures."""
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= 16:
                self._graphs.clear()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            static = [t.clone() for t in inputs]
            out = fn(*static)
            torch.cuda.current_stream(self.device).synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
                outs = fn(*static)
            self._graphs[key] = (g, static, outs)
            return out
        g, static, outs = ent
        for dst, src in zip(static, inputs):
            dst.copy_(src)
        g.replay()
        return tuple(o.clone() for o in outs)

    def _decoder_tail(self, emb, frame_idx, pred):
        """Prompt encoder + both two-way mask decoders -> (low-res left, low-res right, taxonomy)."""
        fn = lambda e, i, t: tuple(self.sam_decoder.decode(e, i, t)[:3])   # noqa: E731
        if not self.decode_graphs or pred.shape[0] > 8 or emb.shape[0] > 8:
            return fn(emb, frame_idx, pred)
        return self._replay(("tail", emb.shape[0], pred.shape[0], emb.dtype), fn, [emb, frame_idx, pred])

    def eval(self):
        return self

    @classmethod
    def from_state_dict(cls, cfg, state_dict, dtype=torch.bfloat16, device="cuda:0", **kw):
        return cls(cfg, state_dict, dtype=dtype, device=device, **kw)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, vision_tower=None, seg_token_idx=None,
                        torch_dtype=torch.bfloat16, sam_checkpoint=None, device="cuda:0", **kw):
        """The constructor the reference's CLIs use (inference.py:158-168, chat.py:104-114):
        `LISAForCausalLM.from_pretrained(version, vision_tower=, seg_token_idx=, torch_dtype=)` followed by
        `initialize_vision_modules` — a merged checkpoint directory (merge_lora_weights_and_save_hf_model.py:149-155 layout:
        sharded weights + index + config.json, `vision_tower` keys excluded) plus the CLIP tower from its own directory
        (clip_encoder.py:21-29). Local directories only: there is no hub access in this setting."""
        from . import checkpoint
        cfg = checkpoint.config_from_dir(pretrained_model_name_or_path)
        if seg_token_idx is not None:
            cfg.seg_token_idx = int(seg_token_idx)
        sd = checkpoint.load_state_dict(pretrained_model_name_or_path, vision_tower, sam_checkpoint, cfg=cfg)
        # The reference takes [SEG] / <im_start> / <im_end> from the TOKENIZER (inference.py:115-131, train_ds.py:142-149); here, in
        # order of authority: the directory's added_tokens.json; else the three rows behind the sentencepiece vocabulary when the
        # embedding has exactly base + 3 rows; else (directories without tokenizer files: this repo's synthetic checkpoints) the
        # last three rows when the embedding has config.json's vocab_size or vocab_size + 3 rows. Anything else — a plain base
        # without the added rows, a vocabulary padded to a multiple of 64 — is refused instead of silently reading ordinary or
        # padding rows as [SEG]. An explicit seg_token_idx is kept as given.
        rows = sd["model.embed_tokens.weight"].shape[0]
        ids = checkpoint.resolve_added_tokens(pretrained_model_name_or_path, rows, layout_asserted=seg_token_idx is not None)
        cfg.llm.vocab = rows
        cfg.im_start_idx, cfg.im_end_idx = int(ids["<im_start>"]), int(ids["<im_end>"])
        if seg_token_idx is None:
            cfg.seg_token_idx = int(ids["[SEG]"])
        return cls(cfg, sd, dtype=torch_dtype, device=device, **kw)
