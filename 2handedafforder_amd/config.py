"""Model geometry of the 2Haff stack (what the reference hard-wires in build_sam.py / the HF checkpoint configs).

  SamCfg   <- build_sam.py:15-23,59-117 (ViT-H: dim 1280, depth 32, heads 16, window 14, global 7/15/23/31)
  ClipCfg  <- openai/clip-vit-large-patch14 (inference.py:34-36) with LLaVA's mm_vision_select_layer = -2
  LlamaCfg <- the LLaVA/Llama-2 checkpoint config (7B: 32x4096, 13B: 40x5120; vocab 32000 + [SEG], <im_start>,
              <im_end> added in that order, train_ds.py:142-149, temp_log.txt:23)
"""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class SamCfg:
    img_size: int = 1024
    patch: int = 16
    embed_dim: int = 1280
    depth: int = 32
    heads: int = 16
    mlp_ratio: int = 4
    window: int = 14
    global_idx: Tuple[int, ...] = (7, 15, 23, 31)
    out_chans: int = 256

    @property
    def grid(self):
        return self.img_size // self.patch


@dataclass
class ClipCfg:
    image: int = 224
    patch: int = 14
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    mlp: int = 4096
    select_layer: int = -2
    eps: float = 1e-5

    @property
    def n_patches(self):
        return (self.image // self.patch) ** 2


@dataclass
class LlamaCfg:
    hidden: int = 4096
    layers: int = 32
    heads: int = 32
    ffn: int = 11008
    vocab: int = 32003
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0


@dataclass
class LisaCfg:
    name: str = "2HandedAfforder-7B"
    sam: SamCfg = field(default_factory=SamCfg)
    clip: ClipCfg = field(default_factory=ClipCfg)
    llm: LlamaCfg = field(default_factory=LlamaCfg)
    out_dim: int = 256
    seg_token_idx: int = 32000
    im_start_idx: int = 32001
    im_end_idx: int = 32002
    bos_token_id: int = 1
    eos_token_id: int = 2
    pad_token_id: int = 0  # pad := unk (inference.py:122)


def haff_7b():
    return LisaCfg()


def haff_13b():
    return LisaCfg(name="2HandedAfforder-13B", llm=LlamaCfg(hidden=5120, layers=40, heads=40, ffn=13824))


def tiny():
    """BASELINE.json configs[0]: ViT-Tiny SAM + 3-layer CLIP + 2-layer LM + the standard 256-d decoders.
    Sizes keep the reference's hard-coded rules valid (256 CLIP patches => the literal 255 of LISA.py:461)."""
    return LisaCfg(
        name="tiny-LISA",
        sam=SamCfg(img_size=224, patch=16, embed_dim=64, depth=4, heads=2, window=7, global_idx=(1, 3)),
        clip=ClipCfg(image=224, patch=14, hidden=64, layers=3, heads=4, mlp=128),
        llm=LlamaCfg(hidden=64, layers=2, heads=4, ffn=192, vocab=323),
        seg_token_idx=320, im_start_idx=321, im_end_idx=322)


def mid():
    """A mid-size geometry for GPU parity tests: exercises window padding (grid 20 -> 21 with window 7),
    head_dim 80 (padded to 96 in the MFMA kernels), d=128 Llama heads, multi-tile GEMMs."""
    return LisaCfg(
        name="mid-LISA",
        sam=SamCfg(img_size=320, patch=16, embed_dim=160, depth=4, heads=2, window=7, global_idx=(1, 3)),
        clip=ClipCfg(image=224, patch=14, hidden=128, layers=3, heads=2, mlp=256),
        llm=LlamaCfg(hidden=256, layers=2, heads=2, ffn=512, vocab=323),
        seg_token_idx=320, im_start_idx=321, im_end_idx=322)
