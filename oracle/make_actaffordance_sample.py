#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only: reads /root/reference). Copies a few LEAF FOLDERS of the reference's own ActAffordance
benchmark archives — data files, not source — into tests/golden/actaffordance_sample/ so that the loaders and the scorer of rows
f3 / f4 are exercised on samples the reference itself ships:

    ActAffordance/data_zipped/masks/P14_05.tar.gz                                  (EPIC-style ids, annotation with obj_* keys)
    ActAffordance/data_zipped/masks/8f91bc0d-9ce7-4b31-aba7-dd59791917df.tar.gz    (Ego4D-style ids, annotation = narration + taxonomy)

Per leaf: annotation.json, inpainting.png (256 x 256 RGB), aff_left|right.png (855 x 855 grey-level affordance maps, > 0 = inside),
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
KEEP = ("annotation.json", "inpainting.png", "aff_left.png", "aff_right.png", "obj_left.png", "obj_right.png")


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
    n = sum(len(fs) for _, _, fs in os.walk(OUT))
    size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(OUT) for f in fs)
    print(f"wrote {n} files, {size / 1024:.0f} KiB under {OUT}")


if __name__ == "__main__":
    main()
