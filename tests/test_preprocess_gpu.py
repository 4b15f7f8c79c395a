"""Device-side frame ingest (rows a1/a2/f2) against the reference's own host path: Pillow (bit-exact, integer work) and
transformers.CLIPImageProcessor (floating point, <= 1e-6), through the C-ABI kernels of csrc/frame_ingest.hip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 100 * np.sin(yy / 17.0)[..., None] * np.cos(xx[..., None] / 23.0 + np.arange(3))
    return np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("filt", ["bilinear", "bicubic"])
@pytest.mark.parametrize("hw,out", [((300, 400), (224, 298)), ((480, 360), (298, 224)), ((97, 131), (224, 302)),
                                      ((768, 1024), (480, 640)), ((150, 224), (686, 1024)), ((1024, 1024), (224, 224))])
def test_resample_matches_pillow_bit_for_bit(dev, filt, hw, out):
    from PIL import Image
    import haff  # noqa: F401
    from haff.preprocess import FrameIngest
    ing = FrameIngest(dev)
    frames = np.stack([_img(hw[0], hw[1], s) for s in (0, 1, 2)])
    got = ing.resize(torch.from_numpy(frames).to(dev), out, filt).cpu().numpy()
    pil = Image.BILINEAR if filt == "bilinear" else Image.BICUBIC
    for b in range(3):
        ref = np.asarray(Image.fromarray(frames[b]).resize((out[1], out[0]), pil))
        assert np.array_equal(got[b], ref), (filt, hw, out, b)


@pytest.mark.parametrize("hw", [(300, 400), (480, 360), (224, 224), (1024, 1024), (97, 131)])
def test_clip_pixels_match_clip_image_processor(dev, hw):
    tr = pytest.importorskip("transformers")
    import haff  # noqa: F401
    from haff.preprocess import FrameIngest
    proc = tr.CLIPImageProcessor()
    frames = np.stack([_img(hw[0], hw[1], s) for s in (3, 4)])
    got = FrameIngest(dev).clip_pixels(torch.from_numpy(frames).to(dev), 224, torch.float32).cpu().numpy()
    for b in range(2):
        ref = proc.preprocess(frames[b], return_tensors="pt")["pixel_values"][0].numpy()
        assert np.abs(got[b] - ref).max() <= 1e-6


def test_evaluate_from_uint8_frames_equals_host_preprocessing(dev):
    """evaluate(frames_u8=...) on a non-square frame (device resize of both towers' inputs) == evaluate() on tensors built
    by the reference's host recipe: Pillow resize + inference.preprocess for SAM, CLIPImageProcessor for CLIP."""
    tr = pytest.importorskip("transformers")
    from PIL import Image
    import haff  # noqa: F401
    from haff import config as hcfg, preprocess as P, weights as hw
    from haff.lisa import LisaMI355
    cfg = hcfg.tiny()
    sd = hw.round_to_bf16_(hw.make_state_dict(cfg, 4))
    model = LisaMI355(cfg, sd, dtype=torch.float32, device=dev)
    S = cfg.sam.img_size
    frame = _img(300, 400, 9)
    nh, nw = P.get_preprocess_shape(300, 400, S)
    resized = np.array(Image.fromarray(frame).resize((nw, nh), Image.BILINEAR))
    images = P.sam_preprocess(torch.from_numpy(resized), S)[None]
    clip = tr.CLIPImageProcessor().preprocess(frame, return_tensors="pt")["pixel_values"]
    ids = torch.tensor([[cfg.bos_token_id, cfg.im_start_idx, -200, cfg.im_end_idx, 9, 8, 7, 6]])
    forced = torch.tensor([[5, cfg.seg_token_idx, cfg.eos_token_id]])
    a = model.evaluate(clip.to(dev), images.to(dev), ids.to(dev), [(nh, nw)], [(300, 400)], max_new_tokens=3, forced_answer=forced)
    b = model.evaluate(None, None, ids.to(dev), [(nh, nw)], [(300, 400)], max_new_tokens=3, forced_answer=forced,
                       frames_u8=torch.from_numpy(frame)[None])
    for x, y in zip(a[1] + a[2] + a[3], b[1] + b[2] + b[3]):
        assert x.shape == y.shape and (x - y).abs().max().item() <= 2e-5 * max(1.0, x.abs().max().item())
