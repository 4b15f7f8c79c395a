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
