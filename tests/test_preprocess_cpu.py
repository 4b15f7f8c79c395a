"""Host side of the device frame ingest (rows a1/a2): the integer coefficient tables restating Pillow's antialiased
resampling are pinned against Pillow itself (the reference resizes through it: transforms.py:27-34 via torchvision,
CLIPImageProcessor via transformers), and the CLIP normalisation LUT + resize-shape rule against
transformers.CLIPImageProcessor. Integer work: bit-exact."""
import numpy as np
import pytest
import torch

import haff  # noqa: F401
from haff import preprocess as P

PIL = pytest.importorskip("PIL.Image")


def resample_u8_host(img, out_hw, filt):
    """Apply the tables the way the HIP kernels do (and Pillow does): horizontal pass, uint8 intermediate, vertical pass."""
    h, w, _ = img.shape
    oh, ow = out_hw
    x = img.astype(np.int64)
    if ow != w:
        bounds, co = P.pil_resample_tables(w, ow, filt)
        out = np.empty((h, ow, 3), np.int64)
        for xx in range(ow):
            x0, n = bounds[xx]
            out[:, xx] = (1 << 21) + np.tensordot(x[:, x0:x0 + n], co[xx, :n].astype(np.int64), axes=([1], [0]))
        x = np.clip(out >> 22, 0, 255)
    if oh != h:
        bounds, co = P.pil_resample_tables(h, oh, filt)
        out = np.empty((oh, x.shape[1], 3), np.int64)
        for yy in range(oh):
            y0, n = bounds[yy]
            out[yy] = (1 << 21) + np.tensordot(co[yy, :n].astype(np.int64), x[y0:y0 + n], axes=([0], [0]))
        x = np.clip(out >> 22, 0, 255)
    return x.astype(np.uint8)


def _img(h, w, seed, smooth=False):
    rng = np.random.default_rng(seed)
    if not smooth:
        return rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 100 * np.sin(yy / 17.0)[..., None] * np.cos(xx[..., None] / 23.0 + np.arange(3))
    return np.clip(base + rng.normal(0, 8, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("filt,pil_filter", [("bilinear", PIL.BILINEAR), ("bicubic", PIL.BICUBIC)])
@pytest.mark.parametrize("hw,out", [((300, 400), (224, 298)), ((480, 360), (298, 224)), ((97, 131), (224, 302)),
                                      ((768, 1024), (480, 640)), ((150, 224), (686, 1024)), ((64, 64), (64, 31))])
def test_tables_reproduce_pillow_bit_for_bit(filt, pil_filter, hw, out):
    for seed, smooth in ((0, False), (1, True)):
        img = _img(hw[0], hw[1], seed, smooth)
        ref = np.asarray(PIL.fromarray(img).resize((out[1], out[0]), pil_filter))
        got = resample_u8_host(img, out, filt)
        assert np.array_equal(got, ref), (filt, hw, out, int(np.abs(got.astype(int) - ref.astype(int)).max()))


def test_clip_shape_rule_and_normalisation_match_transformers():
    tr = pytest.importorskip("transformers")
    proc = tr.CLIPImageProcessor()       # the public defaults of openai/clip-vit-large-patch14
    assert tuple(proc.image_mean) == P.CLIP_MEAN and tuple(proc.image_std) == P.CLIP_STD
    lut = P.clip_normalize_lut()
    for h, w in ((300, 400), (480, 360), (224, 224), (1024, 1024), (97, 131)):
        img = _img(h, w, h + w, smooth=True)
        ref = proc.preprocess(img, return_tensors="pt")["pixel_values"][0].numpy()
        nh, nw = P.clip_resize_shape(h, w, 224)
        r = resample_u8_host(img, (nh, nw), "bicubic")
        top, left = (nh - 224) // 2, (nw - 224) // 2
        crop = r[top:top + 224, left:left + 224]
        got = np.stack([lut[c][crop[..., c]] for c in range(3)])
        assert got.shape == ref.shape == (3, 224, 224)
        assert np.abs(got - ref).max() <= 1e-6, (h, w, np.abs(got - ref).max())


def test_sam_resize_shape_rule():
    assert P.get_preprocess_shape(480, 640, 1024) == (768, 1024)
    assert P.get_preprocess_shape(150, 224, 1024) == (686, 1024)
