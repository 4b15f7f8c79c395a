#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only: reads /root/reference). Copies a few LEAF FOLDERS of the reference's own ActAffordance
benchmark archives — data files, not source — into tests/golden/actaffordance_sample/ so that the loaders and the scorer of rows
f3 / f4 are exercised on samples the reference itself ships:

    ActAffordance/data_zipped/masks/P14_05.tar.gz                                  (EPIC-style ids, annotation with obj_* keys)
    ActAffordance/data_zipped/masks/8f91bc0d-9ce7-4b31-aba7-dd59791917df.tar.gz    (Ego4D-style ids, annotation = narration + taxonomy)

Per leaf: annotation.json, inpainting.png (256 x 256 RGB — a SYNTHETIC stand-in, see KEEP below), aff_left|right.png (855 x 855 grey-level affordance maps, > 0 = inside),
obj_left|right.png (855 x 855, 0 / 255). bench_frame_overlay.png and frame.png (0.2 MB each, visualisation only) are left out.
Walked by 2Haff/utils/aff_dataset.py:457-544 and ActAffordance/scripts/evaluation/calculate_iou.py:117-337."""
import os
import shutil
import tarfile

REF = "/root/reference/ActAffordance/data_zipped/masks"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "actaffordance_sample")
PICK = {
    "P14_05.tar.gz": ["P14_05/0003558",      # right hand only          taxonomy [0, 1, 0, 0]
                      "P14_05/0001413",      # left hand only           taxonomy [1, 0, 0, 0]
                      "P14_05/0002976"],     # both objects annotated, only aff_right present, taxonomy [0, 0, 0, 1]
    "8f91bc0d-9ce7-4b31-aba7-dd59791917df.tar.gz": ["8f91bc0d-9ce7-4b31-aba7-dd59791917df/00000029"],   # both hands
}
# inpainting.png is NOT copied: the benchmark's RGB frames derive from EPIC-KITCHENS / Ego4D video, whose licences restrict
# redistribution, and the reference ships ActAffordance/ without a licence statement (only 2Haff/ carries one: Apache-2.0).
# A seeded synthetic 256 x 256 RGB image stands in for it (the loaders and the scorer only need the file, its size and mode);
# the annotation files and the mask PNGs are the reference authors' own annotation products (PROVENANCE.md is written beside them).
KEEP = ("annotation.json", "aff_left.png", "aff_right.png", "obj_left.png", "obj_right.png")
PROVENANCE = """# tests/golden/actaffordance_sample — provenance

Written by `oracle/make_actaffordance_sample.py` in the build container from
`/root/reference/ActAffordance/data_zipped/masks/{P14_05,8f91bc0d-9ce7-4b31-aba7-dd59791917df}.tar.gz` (pearl-robot-lab/2HandedAfforder).

* Copied as they are: `annotation.json`, `aff_left.png`, `aff_right.png`, `obj_left.png`, `obj_right.png` of four leaf folders —
  the benchmark authors' annotation products (test fixtures: data, no source text).
* NOT copied: `inpainting.png`, `frame.png`, `bench_frame_overlay.png` — RGB frames derived from EPIC-KITCHENS / Ego4D video. Those
  datasets' licences restrict redistribution and the reference repository states no licence for `ActAffordance/` (its `2Haff/`
  directory is Apache-2.0), so `inpainting.png` here is a SYNTHETIC 256 x 256 RGB stand-in (seeded noise over a gradient) of the
  same size and mode. Nothing in the tests depends on the frame's content.
"""


def synthetic_frame(seed, size=256):
    import numpy as np
    rng = np.random.default_rng(seed)
    yy = np.linspace(0, 255, size)[:, None, None]
    xx = np.linspace(255, 0, size)[None, :, None]
    img = 0.5 * rng.integers(0, 256, size=(size, size, 3)) + 0.25 * yy + 0.25 * xx
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def main():
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    for arc, leaves in PICK.items():
        with tarfile.open(os.path.join(REF, arc)) as tf:
            for m in tf.getmembers():
                leaf, name = os.path.dirname(m.name), os.path.basename(m.name)
                if m.isfile() and leaf in leaves and name in KEEP:
                    dst = os.path.join(OUT, leaf, name)
                    os.makedirs(os.path.dirname(dst), exist_ok=True)
                    with tf.extractfile(m) as src, open(dst, "wb") as f:
                        shutil.copyfileobj(src, f)
                    os.chmod(dst, 0o644)
    from PIL import Image
    for i, leaf in enumerate(sorted(l for ls in PICK.values() for l in ls)):
        Image.fromarray(synthetic_frame(1000 + i)).save(os.path.join(OUT, leaf, "inpainting.png"))
    with open(os.path.join(OUT, "PROVENANCE.md"), "w") as f:
        f.write(PROVENANCE)
    n = sum(len(fs) for _, _, fs in os.walk(OUT))
    size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(OUT) for f in fs)
    print(f"wrote {n} files, {size / 1024:.0f} KiB under {OUT}")


if __name__ == "__main__":
    main()
