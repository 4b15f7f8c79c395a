"""The reference's CLI entrypoints on the MI355X path (synthetic tiny model, byte tokenizer): same flags, same
directory contract and output file names as 2Haff/inference.py and 2Haff/chat.py."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _png(path, h, w, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(path)


def test_inference_cli_writes_threshold_masks(dev, tmp_path, monkeypatch):
    import haff  # noqa: F401
    from haff import inference, lisa
    bench = tmp_path / "bench" / "kitchen" / "clip0"
    bench.mkdir(parents=True)
    _png(bench / "inpainting.png", 150, 224, 0)
    (bench / "annotation.json").write_text(json.dumps({"narration": "open drawer"}))
    (tmp_path / "bench" / "kitchen" / "empty").mkdir()
    # random-init models never emit [SEG]; force one so the mask branch of the CLI runs
    orig = lisa.LisaMI355.evaluate

    def forced(self, *a, **kw):
        import torch
        kw["forced_answer"] = torch.tensor([[5, self.cfg.seg_token_idx, self.cfg.eos_token_id]])
        kw["max_new_tokens"] = 3
        return orig(self, *a, **kw)
    monkeypatch.setattr(lisa.LisaMI355, "evaluate", forced)
    out = tmp_path / "vis"
    inference.main(["--synthetic", "tiny", "--benchmark-dir", str(tmp_path / "bench"), "--vis_save_path", str(out),
                    "--image_size", "224"])
    from PIL import Image
    written = 0
    for th in (0.1, 0.2, 0.3, 0.5, 0.7):
        for side in ("left", "right"):
            p = f"{out}{th}/kitchen/clip0/aff_{side}.png"
            if os.path.exists(p):
                written += 1
                arr = np.asarray(Image.open(p))
                assert arr.shape == (150, 224) and set(np.unique(arr)) <= {0, 255}
    assert written in (5, 10)  # taxonomy gating drops at most one hand


def test_chat_cli_roundtrip(dev, tmp_path, monkeypatch, capsys):
    import haff  # noqa: F401
    from haff import chat, lisa
    img = tmp_path / "mug.png"
    _png(img, 224, 200, 1)
    orig = lisa.LisaMI355.evaluate

    def forced(self, *a, **kw):
        import torch
        kw["forced_answer"] = torch.tensor([[7, self.cfg.seg_token_idx, self.cfg.eos_token_id]])
        kw["max_new_tokens"] = 3
        return orig(self, *a, **kw)
    monkeypatch.setattr(lisa.LisaMI355, "evaluate", forced)
    answers = iter(["Where would you hold the mug?", str(img)])
    chat.main(["--synthetic", "tiny", "--vis_save_path", str(tmp_path / "vis"), "--image_size", "224"],
              input_fn=lambda _: next(answers), max_turns=1)
    text = capsys.readouterr().out
    assert "text_output:" in text and "[SEG]" in text
    for name in ("mug_mask_left0.jpg", "mug_mask_right0.jpg", "mug_masked_img_0.jpg"):
        assert os.path.exists(tmp_path / "vis" / name)


def test_output_gating_is_bit_exact(dev):
    """a15 (byte work, bit-exact bar): the uint8 planes the CLIs write — inference.py:276-334 (five sigmoid thresholds,
    0/255, a gated-out hand writes nothing) and chat.py:226-253 (mask > 0, *100, a gated-out hand is all zeros) — from
    the HIP gating kernel equal the reference rule restated in the oracle, on real evaluate() logits of three shapes
    (one with H0*W0 not a multiple of 4) with every taxonomy argmax, plus logits sitting on / one ulp around each
    threshold."""
    import torch
    import haff  # noqa: F401
    from haff import chat, config as hcfg, inference, postprocess as P, weights as hw
    from haff.lisa import LisaMI355
    from oracle import lisa_oracle as O
    cfg = hcfg.tiny()
    sd = hw.round_to_bf16_(hw.make_state_dict(cfg, 11))
    model = LisaMI355(cfg, sd, dtype=torch.bfloat16, device=dev)
    rng = np.random.default_rng(5)
    S = cfg.sam.img_size
    ids = torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx, 9, 8, 7, 6]])
    forced = torch.tensor([[5, cfg.seg_token_idx, cfg.eos_token_id]])
    for orig in ((S, S), (150, 224), (111, 97)):
        images = torch.from_numpy(rng.standard_normal((1, 3, S, S), dtype=np.float32))
        clip = torch.from_numpy(rng.standard_normal((1, 3, 224, 224), dtype=np.float32))
        _, left, right, tax = model.evaluate(clip.to(dev), images.to(dev), ids.to(dev), [(S, S)], [orig], max_new_tokens=3,
                                             forced_answer=forced)
        # scale the logits so every sigmoid threshold cuts through the mask, and plant boundary values
        left = [left[0] * 8.0]
        right = [right[0] * 8.0 + 0.3]
        flat = left[0].view(-1)
        k = 0
        for th in P.THRESHOLDS + (0.5,):
            xs = torch.tensor(P.sigmoid_logit_threshold(th), dtype=torch.float32)
            for v in (xs, torch.nextafter(xs, torch.tensor(float("inf"))), torch.nextafter(xs, torch.tensor(float("-inf")))):
                flat[k] = v.item()
                k += 1
        flat[k:k + 3] = torch.tensor([0.0, -0.0, 1e-45])
        for t_class in range(4):
            t = torch.full((1, 4), 0.1, device=dev)
            t[0, t_class] = 0.7
            got = inference.output_planes(left, right, [t])
            ref = O.inference_output_planes([m.cpu() for m in left], [m.cpu() for m in right], [t.cpu()])
            assert set(got) == set(ref), (orig, t_class)
            for key in ref:
                assert got[key].dtype == np.uint8 and np.array_equal(got[key], ref[key]), (orig, t_class, key)
            gl, gr, _ = chat.render_outputs(np.zeros(orig + (3,), np.uint8), left[0][0], right[0][0], t)
            rl, rr = O.chat_output_planes(left[0].cpu(), right[0].cpu(), t.cpu())
            assert np.array_equal(gl, rl) and np.array_equal(gr, rr), (orig, t_class)
        # two [SEG] prompts in one answer: chat.py takes the argmax of the FLATTENED [2, 4] tensor (an index >= 4 blanks neither
        # hand; index 1 / 0 of the first prompt decide otherwise) — the first row alone would gate differently
        for flat_arg in (0, 1, 5, 4):
            t2 = torch.full((2, 4), 0.1, device=dev)
            t2.view(-1)[flat_arg] = 0.9
            t2[0, 1 if flat_arg >= 4 else 3] += 0.05     # the first row's own argmax points elsewhere
            gl, gr, _ = chat.render_outputs(np.zeros(orig + (3,), np.uint8), left[0][0], right[0][0], t2)
            rl, rr = O.chat_output_planes(left[0].cpu(), right[0].cpu(), t2.cpu())
            assert np.array_equal(gl, rl) and np.array_equal(gr, rr), (orig, "flat", flat_arg)
    # device-side taxonomy gate of the kernel itself (ties -> first maximum, like torch.argmax)
    x = torch.randn((64, 65), device=dev)
    for tax, blank, open_ in (([0.4, 0.4, 0.1, 0.1], 0, False), ([0.4, 0.4, 0.1, 0.1], 1, True), ([0.1, 0.2, 0.35, 0.35], 2, False)):
        planes = P.ops.gate_threshold_masks(x, [0.0, 0.5], 255, torch.tensor(tax, device=dev), blank)
        exp = torch.stack([(x > 0), (x > 0.5)]).to(torch.uint8) * (255 if open_ else 0)
        assert torch.equal(planes, exp)


def test_inference_cli_batches_a_directory(dev, tmp_path, monkeypatch):
    """inference.py over a benchmark directory with frames of different sizes and narrations of different lengths:
    --batch-size 3 (one ragged evaluate() call) writes the same PNG bytes as the reference's frame-by-frame loop
    (--batch-size 1). fp32 mode: every output element is accumulated in a fixed k order, so batching is bit-invariant."""
    import torch
    import haff  # noqa: F401
    from haff import inference, lisa
    for i, (hw, text) in enumerate((((150, 224), "open drawer"), ((224, 224), "pour water from the kettle into the cup"),
                                    ((200, 120), "cut"))):
        d = tmp_path / "bench" / "kitchen" / f"clip{i}"
        d.mkdir(parents=True)
        _png(d / "inpainting.png", hw[0], hw[1], i)
        (d / "annotation.json").write_text(json.dumps({"narration": text}))
    orig = lisa.LisaMI355.evaluate

    def forced(self, *a, **kw):
        B = a[2].shape[0]
        kw["forced_answer"] = torch.tensor([[5, self.cfg.seg_token_idx, self.cfg.eos_token_id]]).expand(B, -1)
        kw["max_new_tokens"] = 3
        return orig(self, *a, **kw)
    monkeypatch.setattr(lisa.LisaMI355, "evaluate", forced)
    outs = {}
    for bs in (1, 3):
        out = tmp_path / f"vis_b{bs}_"
        inference.main(["--synthetic", "tiny", "--benchmark-dir", str(tmp_path / "bench"), "--vis_save_path", str(out),
                        "--image_size", "224", "--precision", "fp32", "--batch-size", str(bs)])
        files = {}
        for th in (0.1, 0.2, 0.3, 0.5, 0.7):
            for i in range(3):
                for side in ("left", "right"):
                    p = f"{out}{th}/kitchen/clip{i}/aff_{side}.png"
                    if os.path.exists(p):
                        files[(th, i, side)] = open(p, "rb").read()
        outs[bs] = files
    assert len(outs[1]) >= 15 and set(outs[1]) == set(outs[3])
    diff = [k for k in outs[1] if outs[1][k] != outs[3][k]]
    assert not diff, f"batched CLI wrote different files for {diff}"


def test_inference_cli_on_the_reference_benchmark_sample_and_rescoring(dev, tmp_path, monkeypatch):
    """inference.py --benchmark-dir over the leaf folders the reference ships (tests/golden/actaffordance_sample, copied from
    ActAffordance/data_zipped/masks by oracle/make_actaffordance_sample.py: 256 x 256 inpainting.png, 855 x 855 masks), then the
    written aff_{left,right}.png planes scored by evaluation.py the way calculate_iou.py:117-337 scores a prediction tree:
    predictions are 256 x 256 (the frame's size), the benchmark masks 855 x 855 -> the resize path; --map sweeps the five
    threshold folders. Random-init weights: the numbers only have to be well-formed."""
    import haff  # noqa: F401
    from haff import evaluation, inference, lisa
    sample = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "actaffordance_sample")
    orig = lisa.LisaMI355.evaluate

    def forced(self, *a, **kw):
        import torch
        B = a[2].shape[0]
        kw["forced_answer"] = torch.tensor([[5, self.cfg.seg_token_idx, self.cfg.eos_token_id]]).expand(B, -1)
        kw["max_new_tokens"] = 3
        return orig(self, *a, **kw)
    monkeypatch.setattr(lisa.LisaMI355, "evaluate", forced)
    out = tmp_path / "pred" / "th"
    inference.main(["--synthetic", "tiny", "--benchmark-dir", sample, "--vis_save_path", str(out), "--image_size", "224",
                    "--batch-size", "3"])
    from PIL import Image
    th_dirs = sorted(os.listdir(tmp_path / "pred"))
    assert th_dirs == ["th0.1", "th0.2", "th0.3", "th0.5", "th0.7"]
    n_files = 0
    for th in th_dirs:
        for sub, leaf in (("P14_05", "0003558"), ("P14_05", "0001413"), ("P14_05", "0002976"), ("8f91bc0d-9ce7-4b31-aba7-dd59791917df", "00000029")):
            d = tmp_path / "pred" / th / sub / leaf
            assert d.is_dir(), d
            for f in os.listdir(d):
                arr = np.asarray(Image.open(d / f))
                assert f in ("aff_left.png", "aff_right.png") and arr.shape == (256, 256) and set(np.unique(arr)) <= {0, 255}
                n_files += 1
    assert n_files >= 20                                               # at least one hand per frame and threshold
    res = evaluation.evaluate_folders(sample, str(tmp_path / "pred"), calc_map=True, verbose=False)
    assert [r["threshold"] for r in res["per_threshold"]] == th_dirs
    for r in res["per_threshold"]:
        assert r["count"] == 4 and 0.0 <= r["iou"] <= 1.0 and 0.0 <= r["iocm"] <= 1.0 and 0.0 <= r["directed_hd"] <= r["hd"] <= 2 ** 0.5 * 855
    areas = []                                                         # a higher threshold can only shrink a predicted region
    for th in th_dirs:
        tot = 0
        for d, _, fs in os.walk(tmp_path / "pred" / th):
            tot += sum(int((np.asarray(Image.open(os.path.join(d, f))) > 0).sum()) for f in fs)
        areas.append(tot)
    assert areas == sorted(areas, reverse=True)
    assert 0.0 <= res["mean_average_precision"] <= 1.0 and res["best"]["iocm"] == max(r["iocm"] for r in res["per_threshold"])
