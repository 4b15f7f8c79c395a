"""ctypes binding of libhaff_hip.so (the C-ABI declared in include/haff_hip.h).

The product path has NO fallback: if the HIP library is missing or a symbol is absent this module raises.
`build_library()` compiles it in-tree with hipcc for gfx950 (works without a GPU: hipcc cross-compiles).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HAFF_LIB_PATH") or os.path.join(_HERE, "lib", "libhaff_hip.so")   # override: A/B builds in tools/
CSRC_DIR = os.path.join(_HERE, "csrc")

c_void_p, c_long, c_int, c_float = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float

# name -> argtypes ; every entry point returns int (0 ok, <0 error) — mirrors include/haff_hip.h
_PROTOS = {
    "haff_gemm_bf16": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p,
                       c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_gemm_bf16_cfg": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p,
                           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_gemm_bf16_ws": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p,
                          c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_long, c_void_p],
    "haff_gemm_bf16_rms": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_int,
                           c_int, c_int, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "haff_gemm_bf16_gather": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p,
                              c_long, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_gemm_bf16_ln": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p,
                          c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_gemm_bf16_heads": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_int, c_int, c_long, c_long, c_void_p],
    "haff_gemm_stream_cap": [c_void_p, c_int],
    "haff_decode_chain_bf16": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_float, c_void_p, c_int, c_void_p],
    "haff_decode_chain_supported": [c_int, c_int, c_int, c_int, c_int],
    "haff_decode_chain_status": [c_void_p, c_int, c_void_p],
    "haff_decode_chain_sync_words": [c_int, c_int],
    "haff_row_stats": [c_void_p, c_long, c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p],
    "haff_row_stats_finalize": [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p],
    "haff_gemm_bf16_qkv_rope": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_gemm_bf16_rowstats": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p,
                                c_long, c_int, c_int, c_int, c_void_p, c_void_p],
    "haff_gemm_bf16_rowstats32": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p,
                                  c_int, c_int, c_int, c_void_p, c_void_p],
    "haff_gemm_f32": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p,
                      c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_attention_bf16": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                            c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                            c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_int,
                            c_void_p],
    "haff_attention_f32": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                           c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                           c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_int,
                           c_void_p],
    "haff_attention_decode_rows_bf16": [c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                        c_void_p, c_long, c_long, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p],
    "haff_attention_decode_rows_f32": [c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                       c_void_p, c_long, c_long, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p],
    "haff_decode_attention_rope_rows_bf16": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                             c_float, c_void_p, c_void_p],
    "haff_rope_cache_rows": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                             c_int, c_int, c_void_p],
    "haff_relpos_tables": [c_void_p, c_long, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_relpos_tables_bf16": [c_void_p, c_long, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_int, c_int, c_void_p],
    "haff_window_attention_bf16": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                   c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                   c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_int, c_long,
                                   c_void_p],
    "haff_attention_lse_bf16": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                c_void_p, c_long, c_long, c_long, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int,
                                c_void_p, c_void_p],
    "haff_attention_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_long, c_long, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p],
    "haff_gemm_tn_workspace_elems": [c_long, c_int, c_int],
    "haff_gemm_tn_bf16": [c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_int, c_void_p, c_long, c_void_p, c_int, c_void_p],
    "haff_lora_qkv_rope_fwd": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_long, c_long, c_int, c_int, c_int, c_float, c_void_p],
    "haff_lora_qkv_rope_bwd": [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_int, c_void_p],
    "haff_lora_dx": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_long, c_int, c_float, c_void_p],
    "haff_lora_dx2": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int, c_long, c_int, c_float,
                      c_void_p],
    "haff_lora_tn_workspace_elems": [c_long, c_int, c_int],
    "haff_lora_tn": [c_void_p, c_long, c_int, c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_void_p, c_long, c_int, c_int,
                     c_int, c_float, c_void_p],
    "haff_global_attention_bf16": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                   c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long,
                                   c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "haff_layernorm": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                       c_int, c_void_p],
    "haff_rmsnorm": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_int, c_int, c_float, c_int, c_void_p],
    "haff_patchify_nchw": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                           c_void_p],
    "haff_patchify_u8": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                         c_int, c_void_p],
    "haff_im2col3x3": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_embed_splice": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                          c_void_p],
    "haff_rope_cache": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                        c_int, c_int, c_void_p],
    "haff_argmax_rows": [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p],
    "haff_decode_book": [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p,
                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, c_long, c_long, c_int, c_void_p],
    "haff_add_bcast": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p],
    "haff_softmax_rows": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "haff_upscale_mask": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                          c_float, c_int, c_void_p],
    "haff_resize_bilinear": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_resample_u8": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "haff_clip_normalize_u8": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p],
    "haff_threshold_masks": [c_void_p, c_void_p, c_long, c_float, c_void_p],
    "haff_gate_threshold_masks": [c_void_p, c_void_p, c_long, c_long, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    # ---- training path (csrc/train.hip + batched GEMMs) ----
    "haff_gemm_bf16_batched": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long,
                               c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_gemm_f32_batched": [c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long, c_long, c_void_p, c_long, c_long,
                              c_long, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_transpose": [c_void_p, c_long, c_long, c_long, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_act_fwd": [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "haff_act_bwd": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "haff_swiglu_fwd": [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "haff_swiglu_bwd": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "haff_axpby": [c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_int, c_void_p],
    "haff_scale_dev": [c_void_p, c_void_p, c_long, c_long, c_void_p, c_long, c_int, c_void_p],
    "haff_reduce_partials": [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_int, c_void_p],
    "haff_sumsq_partials": [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p],
    "haff_mask_loss_stats_partials": [c_void_p, c_void_p, c_void_p, c_int, c_long, c_float, c_void_p, c_void_p],
    "haff_colsum_parts": [c_long],
    "haff_colsum_partials": [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "haff_scatter_add_rows_sorted": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "haff_mul": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p],
    "haff_norm_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p],
    "haff_norm_bwd_add": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p],
    "haff_colsum": [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "haff_softmax_fwd": [c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_int, c_float, c_int, c_int, c_int, c_void_p],
    "haff_softmax_bwd": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_float, c_int, c_void_p],
    "haff_rope": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_cross_entropy": [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_int, c_void_p],
    "haff_mask_loss_stats": [c_void_p, c_void_p, c_void_p, c_int, c_long, c_float, c_void_p],
    "haff_mask_loss_grad_dev": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_float, c_void_p, c_void_p],
    "haff_mask_loss_grad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_float, c_float, c_float, c_void_p],
    "haff_resize_bilinear_bwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "haff_scatter_add_rows": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "haff_taxonomy_ce": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "haff_sumsq": [c_void_p, c_void_p, c_long, c_int, c_void_p],
    "haff_adamw_step_dev": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float, c_float,
                            c_int, c_float, c_void_p, c_int, c_int, c_void_p],
    "haff_adamw_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float, c_float,
                        c_int, c_float, c_int, c_int, c_void_p],
}

EXPORTED_SYMBOLS = tuple(sorted(_PROTOS))

_lib = None


class HaffLibraryError(RuntimeError):
    pass


def build_library(verbose=False):
    """Compile every HIP source for gfx950 into lib/libhaff_hip.so (in-tree, so it travels with the repo)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-8000:])
    if res.returncode != 0:
        raise HaffLibraryError("hipcc build of libhaff_hip.so failed")
    return LIB_PATH


def source_hash():
    """sha256 (16 hex digits) over the HIP sources the library is built from: what profiles/pmc_gemm_traffic.json is stamped with,
    so bench.py never reports PMC traffic collected on another tree as this run's."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC_DIR)):
        if name.endswith((".hip", ".h", ".inc")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(CSRC_DIR, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def load_library():
    """dlopen the library and attach prototypes. Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HaffLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU or eager fallback for the hot path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _PROTOS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HaffLibraryError(f"symbol {name} missing from {LIB_PATH}") from e
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise HaffLibraryError(f"{what} failed with code {rc} "
                               "(-1 bad argument, -2 unsupported shape, -3 launch error)")
