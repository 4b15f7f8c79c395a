"""The two-stream plan of LisaMI355.evaluate (overlap.py) is host logic: checked here against the schedules that measured best on
MI355X (profiles/r5_overlap_*), and for the properties every plan must have."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import haff  # noqa: F401,E402
from haff import config as hcfg, overlap  # noqa: E402


def test_plan_reproduces_the_measured_optima():
    c7, c13 = hcfg.haff_7b(), hcfg.haff_13b()
    # late mode (<= 16 frames): the whole encoder beside the decode steps, behind the prefill
    assert overlap.plan(c7, 4, 4, 36, 8, True) == ([128], True)
    assert overlap.plan(c7, 8, 8, 36, 8, True) == ([160], True)
    assert overlap.plan(c7, 16, 16, 36, 8, True) == ([192], True)
    assert overlap.plan(c13, 8, 8, 36, 8, True) == ([128], True)
    # throughput batches: the encoder starts first, its second half meets the decode steps
    assert overlap.plan(c7, 64, 16, 36, 8, False) == ([256, 256, 224, 224], False)
    assert overlap.plan(c7, 32, 8, 36, 8, False) == ([256, 256, 192, 192], False)
    # 32-frame passes at 64 frames: the encoder is nearly through when the prefill ends (measured: a cap only costs there)
    assert overlap.plan(c7, 64, 32, 36, 8, False) == (None, False)
    # one frame: nothing to gain (measured), nothing planned
    assert overlap.plan(c7, 1, 1, 36, 8, True) == (None, False)
    assert overlap.plan(c7, 64, 16, 36, 1, False) == (None, False)   # no decode steps
    assert overlap.auto_chunk(64, False) == 16 and overlap.auto_chunk(32, False) == 8 and overlap.auto_chunk(8, True) == 8
    assert overlap.auto_chunk(16, True) == 16 and overlap.auto_chunk(1, True) == 1


def test_plan_moves_smoothly_with_the_rates():
    """calibrate() scales the rates by what the device's probes read (boxes of this pool differ by ~5 %): a uniformly faster or
    slower box keeps every plan (only ratios enter); a box whose matrix cores / HBM differ by up to 10 % from nominal moves a cap by
    at most one step of CAPS, never the structure (which passes are capped, whether the encoder waits)."""
    from dataclasses import replace
    c7 = hcfg.haff_7b()
    N = overlap.NOMINAL
    cases = [(64, 16, False), (32, 8, False), (16, 16, True), (8, 8, True), (4, 4, True)]
    for f in (0.9, 0.95, 1.05, 1.1):
        uni = replace(N, enc=N.enc * f, llm=N.llm * f, stream_bw_per_cu=tuple(v * f for v in N.stream_bw_per_cu))
        for frames, chunk, late in cases:
            assert overlap.plan(c7, frames, chunk, 36, 8, late, uni) == overlap.plan(c7, frames, chunk, 36, 8, late)
        for skew in (replace(N, enc=N.enc * f, llm=N.llm * f), replace(N, stream_bw_per_cu=tuple(v * f for v in N.stream_bw_per_cu))):
            for frames, chunk, late in cases:
                a, wa = overlap.plan(c7, frames, chunk, 36, 8, late)
                b, wb = overlap.plan(c7, frames, chunk, 36, 8, late, skew)
                assert wa == wb and len(a) == len(b)
                steps = (256,) + tuple(reversed(overlap.CAPS))
                assert all(abs(steps.index(x) - steps.index(y)) <= 1 for x, y in zip(a, b)), (f, frames, a, b)


def test_plans_are_well_formed():
    for cfg in (hcfg.haff_7b(), hcfg.haff_13b(), hcfg.tiny()):
        for frames in (1, 2, 3, 4, 5, 8, 13, 16, 17, 24, 32, 48, 64, 100):
            for late in (True, False):
                for chunk in (1, 4, 8, 16, 32, overlap.auto_chunk(frames, late)):
                    for new_tokens in (1, 2, 8, 32):
                        caps, wait = overlap.plan(cfg, frames, chunk, 36, new_tokens, late)
                        assert wait in (True, False) and (not wait or late)
                        if caps is None:
                            assert not wait
                            continue
                        assert len(caps) == (frames + chunk - 1) // chunk
                        assert all(c == 256 or c in overlap.CAPS for c in caps) and all(c % 8 == 0 for c in caps)
                        assert caps == sorted(caps, reverse=True)      # full chip first, the capped passes at the end
