#!/usr/bin/env python3
"""LoRA fine-tune loop for MI355X — the train_ds.py contract of the reference (2Haff/train_ds.py:125-622).

Same flags (the ones that drive the loop), same trainable set, same optimiser schedule (AdamW lr/betas, WarmupDecayLR
100 warm-up steps then linear decay, gradient clipping 1.0, gradient accumulation), same meter names / log format
(`Epoch: [e][ i/500] Time ... Loss ...`, train_ds.py:500-523, temp_log.txt:16445), best-IoU-only checkpoint with
auto-resume (train_ds.py:395-412,470-486), validation IoU / IoCM (train_ds.py:625-796).

MI355X-first differences: DeepSpeed ZeRO-2 is replaced by plain data parallelism — one process per GPU, gradients of
the trainable set (294 M params for 7B) accumulate locally over the micro-steps and are averaged ONCE per optimizer
step with bucketed RCCL all-reduces over xGMI; optimizer state is replicated (3.5 GB fp32 for 7B in 288 GB).
The 2HANDS loaders (h5 / HF datasets, cv2 contour masks) are replaced by a seeded synthetic sample generator that
emits the same 12-tuple per sample and the same collate_fn batch dict (utils/dataset.py:47-60,152-169).

  python -m torch.distributed.run --nproc-per-node 8 2handedafforder_amd/train_ds.py --synthetic 7b --epochs 1 ...
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import haff  # noqa: F401
    from haff import checkpoint, config as hcfg, dist as hdist, prompt as hprompt, train_ops as T
    from haff.aff_dataset import AffRecordsDataset
    from haff.train_model import LisaTrainable
else:
    from . import checkpoint, config as hcfg, dist as hdist, prompt as hprompt, train_ops as T
    from .aff_dataset import AffRecordsDataset
    from .train_model import LisaTrainable


def parse_args(args):
    p = argparse.ArgumentParser(description="LISA Model Training")
    p.add_argument("--local_rank", default=0, type=int, help="node rank")
    p.add_argument("--version", default="liuhaotian/llava-v1.5-13b")
    p.add_argument("--vis_save_path", default="./vis_output", type=str)
    p.add_argument("--precision", default="bf16", type=str, choices=["fp32", "bf16", "fp16"])
    p.add_argument("--image_size", default=1024, type=int)
    p.add_argument("--model_max_length", default=575, type=int)
    p.add_argument("--lora_r", default=8, type=int)
    p.add_argument("--vision-tower", default="openai/clip-vit-large-patch14", type=str)
    p.add_argument("--dataset_dir", default="./dataset", type=str)
    p.add_argument("--log_base_dir", default="./runs", type=str)
    p.add_argument("--exp_name", default="lisa", type=str)
    p.add_argument("--epochs", default=10, type=int)
    p.add_argument("--steps_per_epoch", default=500, type=int)
    p.add_argument("--batch_size", default=2, type=int, help="batch size per device per step")
    p.add_argument("--grad_accumulation_steps", default=10, type=int)
    p.add_argument("--val_batch_size", default=1, type=int)
    p.add_argument("--workers", default=4, type=int)
    p.add_argument("--lr", default=0.001, type=float)
    p.add_argument("--ce_loss_weight", default=1.0, type=float)
    p.add_argument("--dice_loss_weight", default=0.5, type=float)
    p.add_argument("--bce_loss_weight", default=2.0, type=float)
    p.add_argument("--lora_alpha", default=16, type=int)
    p.add_argument("--lora_dropout", default=0.05, type=float)
    p.add_argument("--lora_target_modules", default="q_proj,v_proj", type=str)
    p.add_argument("--beta1", default=0.9, type=float)
    p.add_argument("--beta2", default=0.95, type=float)
    p.add_argument("--no_eval", action="store_true", default=False)
    p.add_argument("--benchmark_dir", default="", type=str,
                   help="validation folders <root>/<video>/<frame>/{inpainting.png, aff_left.png, aff_right.png, annotation.json} "
                        "(AffDatasetVal, train_ds.py:121,330-336 of the reference); empty: validate on held-out training records")
    p.add_argument("--eval_only", action="store_true", default=False)
    p.add_argument("--vision_pretrained", default="PATH_TO_SAM_ViT-H", type=str)
    p.add_argument("--out_dim", default=256, type=int)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--print_freq", default=1, type=int)
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--train_mask_decoder", action="store_true", default=True)
    p.add_argument("--use_mm_start_end", action="store_true", default=True)
    p.add_argument("--auto_resume", action="store_true", default=True)
    p.add_argument("--conv_type", default="llava_v1", type=str, choices=["llava_v1", "llava_llama_2"])
    # MI355X / offline extras
    p.add_argument("--synthetic", default=None, choices=["tiny", "mid", "7b", "13b"],
                   help="random-init model of this geometry + byte tokenizer + synthetic samples (no checkpoints needed); "
                        "without it --version / --vision-tower / --vision_pretrained / --dataset_dir are REAL local paths")
    p.add_argument("--sam_records", default=None, type=str,
                   help="a torch-saved list of 2HANDS records (narration, inpainted/image, taxonomy, masks) when the HF "
                        "`datasets` hub / the h5 layout under --dataset_dir is not available")
    p.add_argument("--val_samples", default=4, type=int)
    p.add_argument("--mask_hw", default=None, type=int, nargs=2, help="ground-truth mask size (default: image size)")
    p.add_argument("--seed", default=0, type=int)
    return p.parse_args(args)


# ---------------------------------------------------------------------------------------------------------------------
# meters (utils/utils.py:52-150) — same names/format, all-reduced in ONE 2*len(meters)-float collective
# ---------------------------------------------------------------------------------------------------------------------
class AverageMeter:
    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)


def all_reduce_meters(meters, device):
    import torch.distributed as dist
    if not dist.is_initialized():
        return
    t = torch.tensor([v for m in meters for v in (m.sum, m.count)], dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    vals = t.tolist()
    for i, m in enumerate(meters):
        m.sum, m.count = vals[2 * i], vals[2 * i + 1]
        m.avg = m.sum / (m.count + 1e-5)


def progress_line(epoch, step, total, meters):
    nd = len(str(total))
    head = "Epoch: [{}][{:>{w}d}/{}]".format(epoch, step, total, w=nd)
    return "\t".join([head] + [str(m) for m in meters])


# ---------------------------------------------------------------------------------------------------------------------
# synthetic 2HANDS-shaped samples + collate (utils/aff_dataset.py:267-280, utils/dataset.py:30-169)
# ---------------------------------------------------------------------------------------------------------------------
QUESTION = "Where would you interact with the object to perform action {}? Please output segmentation mask."
ANSWER = "It is [SEG]."


class SyntheticAffDataset:
    """Seeded stand-in for AffDataset: (image_path, image, image_clip, conversations, masks_left, masks_right,
    taxonomy, label, resize, questions, sampled_classes, inference)."""

    def __init__(self, cfg, n, seed, mask_hw=None, inference=False):
        self.cfg, self.n, self.seed, self.inference = cfg, n, seed, inference
        self.hw = tuple(mask_hw) if mask_hw else (cfg.sam.img_size, cfg.sam.img_size)

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        cfg = self.cfg
        g = torch.Generator().manual_seed(self.seed * 100003 + idx)
        S = cfg.sam.img_size
        image = torch.randn((3, S, S), generator=g)
        image_clip = torch.randn((3, cfg.clip.image, cfg.clip.image), generator=g)
        h, w = self.hw
        yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")

        def blob():
            cy, cx = torch.rand(2, generator=g).tolist()
            r = 0.1 + 0.2 * torch.rand(1, generator=g).item()
            return (((yy / h - cy) ** 2 + (xx / w - cx) ** 2) < r * r).float()[None]
        tax_idx = int(torch.randint(0, 4, (1,), generator=g))
        taxonomy = [0.0] * 4
        taxonomy[tax_idx] = 1.0
        action = "action%d" % int(torch.randint(0, 50, (1,), generator=g))
        conv = hprompt.default_conversation()
        conv.append_message(conv.roles[0], hprompt.DEFAULT_IMAGE_TOKEN + "\n" + QUESTION.format(action))
        conv.append_message(conv.roles[1], ANSWER)
        label = {"left": torch.zeros(h, w), "right": torch.zeros(h, w)}
        return ("synthetic/%06d.png" % idx, image, image_clip, [conv.get_prompt()], blob(), blob(), taxonomy, label,
                (S, S), [QUESTION.format(action)], [action], self.inference)


def collate_fn(batch, tokenizer, model_max_length=575, use_mm_start_end=True, conv_type="llava_v1"):
    """utils/dataset.py:30-169: pad with pad_token, mask the instruction spans with -100; conv_type picks the template whose sep2
    splits the rounds and the separator that ends a round's instruction (:97-101: " ASSISTANT: " or "[/INST] ")."""
    images, clips, convs, ml, mr, labels_l, resizes, tax, offs = [], [], [], [], [], [], [], [], [0]
    paths, questions, classes = [], [], []
    for (path, image, image_clip, conversations, m_left, m_right, taxonomy, label, resize, _q, _c, inference) in batch:
        paths.append(path)
        questions.append(_q)
        classes.append(_c)
        images.append(image)
        clips.append(image_clip)
        convs.extend(conversations)
        ml.append(m_left.float())
        mr.append(m_right.float())
        labels_l.append(label)
        resizes.append(resize)
        tax.append(torch.tensor(taxonomy))
        offs.append(offs[-1] + len(conversations))
    if use_mm_start_end:
        convs = [c.replace(hprompt.DEFAULT_IMAGE_TOKEN, hprompt.image_placeholder(True)) for c in convs]
    ids = [hprompt.tokenizer_image_token(c, tokenizer, return_tensors="pt") for c in convs]
    input_ids = torch.nn.utils.rnn.pad_sequence(ids, batch_first=True, padding_value=tokenizer.pad_token_id)
    attention_masks = input_ids.ne(tokenizer.pad_token_id)
    targets = input_ids.clone()
    conv = hprompt.get_conv(conv_type)
    sep = hprompt.label_separator(conv_type, conv)
    for conversation, target in zip(convs, targets):
        cur = 1
        target[:cur] = -100
        for rou in conversation.split(conv.sep2):
            if rou == "":
                break
            parts = rou.split(sep)
            assert len(parts) == 2, (len(parts), rou)
            parts[0] += sep
            round_len = len(hprompt.tokenizer_image_token(rou, tokenizer))
            instruction_len = len(hprompt.tokenizer_image_token(parts[0], tokenizer)) - 2
            target[cur:cur + instruction_len] = -100
            cur += round_len
        target[cur:] = -100
    if not batch[0][-1]:
        trunc = model_max_length - 255
        input_ids, targets, attention_masks = input_ids[:, :trunc], targets[:, :trunc], attention_masks[:, :trunc]
    return {"images": torch.stack(images, 0), "images_clip": torch.stack(clips, 0), "input_ids": input_ids,
            "labels": targets, "attention_masks": attention_masks, "masks_list_left": ml, "masks_list_right": mr,
            "label_list": labels_l, "resize_list": resizes, "offset": torch.LongTensor(offs),
            "inference": batch[0][-1], "conversation_list": convs, "taxonomies_list": torch.stack(tax, 0),
            "image_paths": paths, "questions_list": questions, "sampled_classes_list": classes}


# ---------------------------------------------------------------------------------------------------------------------
# validation metrics (train_ds.py:761-796)
# ---------------------------------------------------------------------------------------------------------------------
def calculate_iou(a, b):
    inter = np.logical_and(a, b).sum()
    union = np.logical_or(a, b).sum()
    return inter / union if union != 0 else 0.0


def calculate_iocm(benchmark_mask, comparison_mask):
    inter = np.logical_and(benchmark_mask, comparison_mask).sum()
    area = comparison_mask.sum()
    return inter / area if area != 0 else 0.0


@torch.no_grad()
def validate(model, dataset, tokenizer, args, rank, world, device):
    """train_ds.py:625-758: teacher-forced forward(inference=True), masks > 0, taxonomy-gated union of L/R vs GT union."""
    model.eval()
    iou_m, iocm_m = AverageMeter("IoU"), AverageMeter("IoCM")
    lo, hi = hdist.shard_bounds(len(dataset), rank, world)
    for idx in range(lo, hi):
        batch = collate_fn([dataset[idx]], tokenizer, args.model_max_length, conv_type=args.conv_type)
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        out = model(**batch)
        t = int(out["pred_taxonomies"][0][0].argmax())
        left = (out["pred_masks_left"][0][0] > 0).cpu().numpy()
        right = (out["pred_masks_right"][0][0] > 0).cpu().numpy()
        if t == 1:
            left[:] = False
        if t == 0:
            right[:] = False
        pred = np.logical_or(left, right)
        gt = np.logical_or(out["gt_masks_left"][0][0].cpu().numpy() > 0, out["gt_masks_right"][0][0].cpu().numpy() > 0)
        iou_m.update(calculate_iou(pred, gt))
        iocm_m.update(calculate_iocm(gt, pred))
    all_reduce_meters([iou_m, iocm_m], device)
    model.train()
    return iou_m.avg, iocm_m.avg


# ---------------------------------------------------------------------------------------------------------------------
def main(argv):
    args = parse_args(argv)
    hprompt.set_default_conversation(args.conv_type)      # train_ds.py:188-190
    rank, world, local_rank = hdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("train_ds.py needs MI355X devices: the fine-tune path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32
    if args.synthetic:
        cfg = {"tiny": hcfg.tiny, "mid": hcfg.mid, "7b": hcfg.haff_7b, "13b": hcfg.haff_13b}[args.synthetic]()
        tokenizer = checkpoint.ByteTokenizer(cfg)
        tokenizer.model_max_length = args.model_max_length
        sd = checkpoint.synthetic_state_dict(cfg, 1234, device, dtype)   # identical on every rank (same seed)
    else:
        # the reference's model construction (train_ds.py:135-181): tokenizer + added tokens from --version, the base
        # checkpoint, the CLIP tower from --vision-tower, SAM from --vision_pretrained, text_hidden_fcs / taxonomy head /
        # the three new vocabulary rows freshly initialised (checkpoint.complete_for_training, same seed on every rank)
        for flag, path in (("--version", args.version), ("--vision-tower", args.vision_tower),
                           ("--vision_pretrained", args.vision_pretrained)):
            if not os.path.exists(path):
                raise SystemExit(f"{flag} {path!r} is not a local path (no hub access here); pass a directory/file, "
                                 f"or --synthetic <geometry> for a random-init plumbing run")
        cfg = checkpoint.config_from_dir(args.version)
        cfg.out_dim = args.out_dim
        tokenizer = checkpoint.SentencePieceTokenizer(os.path.join(args.version, "tokenizer.model"))
        tokenizer.model_max_length = args.model_max_length
        cfg.seg_token_idx = tokenizer("[SEG]", add_special_tokens=False).input_ids[0]
        cfg.im_start_idx = tokenizer("<im_start>", add_special_tokens=False).input_ids[0]
        cfg.im_end_idx = tokenizer("<im_end>", add_special_tokens=False).input_ids[0]
        cfg.bos_token_id, cfg.eos_token_id, cfg.pad_token_id = tokenizer.bos_token_id, tokenizer.eos_token_id, tokenizer.pad_token_id
        # the vocabulary of the model being built is the tokenizer's (train_ds.py:231-233: resize_token_embeddings(len(tokenizer))),
        # whatever config.json's vocab_size says about added tokens (a base that already carries 32003 rows stays 32003)
        cfg.llm.vocab = len(tokenizer)
        sd = checkpoint.load_state_dict(args.version, args.vision_tower, args.vision_pretrained, for_training=True,
                                        seed=args.seed, cfg=cfg)
    model = LisaTrainable(cfg, sd, dtype=dtype, device=device, lora_r=args.lora_r, lora_alpha=args.lora_alpha,
                          lora_dropout=args.lora_dropout, ce_loss_weight=args.ce_loss_weight,
                          dice_loss_weight=args.dice_loss_weight, bce_loss_weight=args.bce_loss_weight, seed=args.seed)
    del sd
    n_lora = sum(p.numel() for k, p in model.named_parameters() if "lora_" in k)
    n_train = sum(p.numel() for p in model.parameters())
    if rank == 0:
        print(f"trainable params: {n_train:,d} (LoRA {n_lora:,d}) | world_size {world} | micro-batch {args.batch_size} "
              f"x accum {args.grad_accumulation_steps}")
    reducer = T.GradBucketReducer(model.named_parameters())   # p.grad become views into flat per-dtype buckets
    opt = T.BucketAdamW(reducer, model.named_parameters())    # fp32 master / moments per bucket, one fused launch each
    states = opt.states
    ckpt_dir = os.path.join(args.log_base_dir, args.exp_name, "ckpt_model")
    global_step, best_score, start_epoch = 0, 0.0, args.start_epoch
    resume = args.resume or (ckpt_dir if args.auto_resume and os.path.exists(os.path.join(ckpt_dir, "latest.pt")) else "")
    if resume:
        blob = torch.load(os.path.join(resume, "latest.pt"), map_location=device, weights_only=False)
        model.load_state_dict(blob["params"])
        for k, st in blob["optim"].items():
            states[k].master.copy_(st["master"]); states[k].m.copy_(st["m"]); states[k].v.copy_(st["v"]); states[k].step = st["step"]
        opt.refresh_lp()
        global_step, best_score = blob["global_step"], blob["best_score"]
        start_epoch = global_step // args.steps_per_epoch
        if rank == 0:
            print(f"resume training from {resume}, start from epoch {start_epoch}")
    if args.synthetic:
        train_ds = SyntheticAffDataset(cfg, 10 ** 9, args.seed + 1000 * rank, args.mask_hw)
        val_ds = SyntheticAffDataset(cfg, args.val_samples, 777, args.mask_hw, inference=True)
    else:
        # AffRecordsDataset = utils/aff_dataset.py:48-346 on the records' HF layout
        if args.sam_records:
            records = torch.load(args.sam_records, weights_only=False)
            train_ds = AffRecordsDataset(records, cfg, seed=args.seed + 1000 * rank)
            val_ds = AffRecordsDataset(records[:max(args.val_samples, 1)], cfg, samples_per_epoch=args.val_samples, inference=True, seed=777)
        elif os.path.isdir(args.dataset_dir):
            train_ds = AffRecordsDataset.from_local(args.dataset_dir, cfg, seed=args.seed + 1000 * rank)
            val_ds = AffRecordsDataset.from_local(args.dataset_dir, cfg, samples_per_epoch=args.val_samples, inference=True, seed=777)
        else:
            train_ds = AffRecordsDataset.from_hf(args.dataset_dir, cfg, seed=args.seed + 1000 * rank)
            val_ds = AffRecordsDataset.from_hf(args.dataset_dir, cfg, samples_per_epoch=args.val_samples, inference=True, seed=777)
    if not args.synthetic and args.benchmark_dir and os.path.isdir(args.benchmark_dir) and not args.no_eval:
        AffValDataset = sys.modules[AffRecordsDataset.__module__].AffValDataset   # the reference's validation set (train_ds.py:330-336)
        val_ds = AffValDataset(args.benchmark_dir, cfg, seed=777)
        if rank == 0:
            print(f"Training with {len(train_ds)} examples and validating with {len(val_ds)} examples.")
    total_steps = args.epochs * args.steps_per_epoch
    if args.eval_only:
        iou, iocm = validate(model, val_ds, tokenizer, args, rank, world, device)
        if rank == 0:
            print(f"IoU: {iou:.4f}, IoCM: {iocm:.4f}")
        return
    sample_idx = global_step * args.grad_accumulation_steps * args.batch_size
    for epoch in range(start_epoch, args.epochs):
        meters = [AverageMeter("Time", ":6.3f"), AverageMeter("Loss", ":.4f"), AverageMeter("CeLoss", ":.4f"),
                  AverageMeter("MaskLoss", ":.4f"), AverageMeter("MaskBCELoss", ":.4f"), AverageMeter("MaskDICELoss", ":.4f"),
                  AverageMeter("TaxonomyCELoss", ":.4f")]
        keys = [None, "loss", "ce_loss", "mask_loss", "mask_bce_loss", "mask_dice_loss", "taxonomy_ce_loss"]
        model.train()
        end = time.time()
        loss_acc = torch.zeros((len(keys) - 1,), dtype=torch.float32, device=device)   # meters stay on the device
        n_acc = 0
        for step in range(args.steps_per_epoch):
            reducer.zero()
            for micro in range(args.grad_accumulation_steps):
                batch = collate_fn([train_ds[sample_idx + j] for j in range(args.batch_size)], tokenizer, args.model_max_length,
                                   conv_type=args.conv_type)
                sample_idx += args.batch_size
                batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
                out = model(**batch)
                # gradients accumulate into the bucket views across the micro-steps; on the last one each bucket's
                # all-reduce (RCCL) is issued as soon as its last gradient lands, under the rest of backward
                reducer.begin(sync=micro == args.grad_accumulation_steps - 1)
                out["loss"].backward()
                loss_acc += torch.stack([out[k].detach().float().reshape(()) for k in keys[1:]])   # no host sync
                n_acc += 1
            reducer.finish()
            grads = reducer.grads()
            gscale = 1.0 / args.grad_accumulation_steps
            # gradient_clipping: 1.0 — the coefficient min(1, 1 / (norm + 1e-6)) is computed and consumed on the device (the
            # optimizer launches queue up behind backward; no host read of the norm)
            clip = T.clip_coef_device(T.grad_norm(grads) * gscale, 1.0)
            lr = T.warmup_decay_lr(global_step, total_steps, args.lr)
            opt.step(lr=lr, betas=(args.beta1, args.beta2), eps=1e-8, wd=0.0, gscale=gscale, gscale_dev=clip)
            global_step += 1
            meters[0].update(time.time() - end)
            end = time.time()
            if global_step % args.print_freq == 0:
                for m, v in zip(meters[1:], (loss_acc / max(n_acc, 1)).tolist()):   # ONE device->host read per log line
                    m.update(v, n_acc * args.batch_size)
                loss_acc.zero_()
                n_acc = 0
                all_reduce_meters(meters, device)
                if rank == 0:
                    print(progress_line(epoch, step + 1, args.steps_per_epoch, meters[:6]), flush=True)
                for m in meters:
                    m.reset()
        if not args.no_eval:
            iou, iocm = validate(model, val_ds, tokenizer, args, rank, world, device)
            if rank == 0:
                print(f"IoU: {iou:.4f}, IoCM: {iocm:.4f}")
            is_best = iou > best_score
            best_score = max(iou, best_score)
        else:
            is_best = True
        if is_best:
            if world > 1:
                torch.distributed.barrier()
            if rank == 0:  # parameters and optimizer state are replicated: rank 0 writes the only copy
                os.makedirs(ckpt_dir, exist_ok=True)
                torch.save({"params": model.state_dict(), "global_step": global_step, "best_score": best_score, "epoch": epoch,
                            "optim": {k: {"master": s.master, "m": s.m, "v": s.v, "step": s.step} for k, s in states.items()}},
                           os.path.join(ckpt_dir, "latest.pt"))
                print(f"saved checkpoint to {ckpt_dir} (global_step{global_step})")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
