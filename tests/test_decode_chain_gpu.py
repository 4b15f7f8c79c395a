"""The <= 8-row KV-cached decode step as ONE launch (csrc/decode_chain.hip, LlamaHip.decode_chain) against the five-launch layer it
replaces (haff_gemm_bf16_rms x 4 + haff_decode_attention_rope_rows_bf16 per transformers LlamaDecoderLayer, llava_llama.py:93-102):

  * the chained launch against THE SAME KERNEL launched once per (layer, stage) (`decode_chain = "stages"`: every wait already
    satisfied by stream order): BIT-IDENTICAL hidden states AND KV-cache contents over several steps at full 7B / 13B width — a
    stale read across a stage boundary shows up as a different bit;
  * against the five-launch layer of the library (same statements, but hipcc contracts the softmax / residual FMAs of the two
    translation units differently): equal within a few bf16 ulps of the output scale, layer 0's appended cache rows bit for bit;
  * ragged rows (every row at its own position), repeated launches equal bit for bit, the sticky timeout word stays 0;
  * through a hipGraph (the form generate() replays) the chained step equals its eager self.
Against the CPU oracle the chain is covered by test_configs_gpu.py::test_decode_rows_carrying_rmsnorm_matches_oracle (it is the
default path of LlamaHip.decode_rows at <= 8 rows)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(width, layers, dev, seed=33):
    import haff  # noqa: F401
    from haff import config as hcfg, weights as hw
    from haff.llava import LlamaHip
    cfg = hcfg.haff_7b() if width == "7b" else hcfg.haff_13b()
    cfg.llm.layers = layers
    shapes = {k: v for k, v in hw.llm_shapes(cfg).items() if k.startswith("model.layers.") or k == "model.norm.weight"}
    shapes["model.embed_tokens.weight"] = (8, cfg.llm.hidden)
    shapes["lm_head.weight"] = (8, cfg.llm.hidden)
    sd = hw.round_to_bf16_(hw.make_state_dict(cfg, seed, shapes))
    g = torch.Generator().manual_seed(5)
    for k in sd:   # norm weights away from 1 so that the fold matters
        if k.endswith("layernorm.weight") or k == "model.norm.weight":
            sd[k] = (1.0 + 0.5 * torch.randn(sd[k].shape, generator=g)).to(torch.bfloat16).float()
    return cfg, LlamaHip(sd, cfg.llm, torch.bfloat16, dev)


def _run(llm, xd, T, n_steps, chain, pos_rows=None):
    """prefill of T positions, then n_steps cached steps; returns (hidden [B, n_steps, H], K cache, V cache of the last layer)."""
    B = xd.shape[0]
    llm.decode_chain = chain
    cache = llm.new_cache(B, T + n_steps + 8)
    llm.forward(xd[:, :T].clone(), cache)   # (forward works in place on a contiguous input: B = 1 slices are)
    out = []
    for s_ in range(n_steps):
        if pos_rows is None:
            cache["pos"].fill_(T + s_)
            cache["nk"].fill_(T + s_ + 1)
        else:
            cache["pos"].copy_(pos_rows + s_)
            cache["nk"].copy_(pos_rows + s_ + 1)
        out.append(llm.decode_rows(xd[:, T + s_:T + s_ + 1].clone(), cache).clone())
    torch.cuda.synchronize()
    return torch.cat(out, 1), [k.clone() for k in cache["k"]], [v.clone() for v in cache["v"]], cache


@pytest.mark.parametrize("width,B,layers,T", [("7b", 8, 4, 70), ("7b", 1, 4, 291), ("7b", 5, 2, 130), ("13b", 8, 3, 64), ("13b", 3, 3, 200)])
def test_chain_is_bit_identical_to_its_own_kernel_launched_stage_by_stage(dev, width, B, layers, T):
    cfg, llm = _model(width, layers, dev)
    n_steps = 5
    x = torch.randn((B, T + n_steps, cfg.llm.hidden), generator=torch.Generator().manual_seed(2)).to(dev, torch.bfloat16)
    h_c, k_c, v_c, cache = _run(llm, x, T, n_steps, True)
    assert "chain" in cache, "the chained launch was not taken"
    h_s, k_s, v_s, cache_s = _run(llm, x, T, n_steps, "stages")
    from haff import ops
    assert ops.decode_chain_status(cache["chain"]["sync"], layers) and ops.decode_chain_status(cache_s["chain"]["sync"], layers)
    assert torch.isfinite(h_c.float()).all()
    assert torch.equal(h_c, h_s), f"max |diff| {(h_c.float() - h_s.float()).abs().max().item():.3e}"
    for a, b in zip(k_c + v_c, k_s + v_s):
        assert torch.equal(a[:, :T + n_steps], b[:, :T + n_steps])
    for _ in range(3):   # (other placements of the same workgroups: the result may not depend on who ran where)
        assert torch.equal(h_c, _run(llm, x, T, n_steps, True)[0])


@pytest.mark.parametrize("width,B", [("7b", 1), ("7b", 3), ("13b", 4), ("7b", 8), ("13b", 6)])
def test_chain_matches_the_five_launch_layer(dev, width, B):
    cfg, llm = _model(width, 2, dev)
    T, n_steps = 291 if B <= 4 else 96, 3
    x = torch.randn((B, T + n_steps, cfg.llm.hidden), generator=torch.Generator().manual_seed(3)).to(dev, torch.bfloat16)
    h_c, k_c, v_c, cache = _run(llm, x, T, n_steps, True)
    h_p, k_p, v_p, cache_p = _run(llm, x, T, n_steps, False)
    assert "chain" in cache and "chain" not in cache_p
    scale = h_p.float().abs().max().item()
    err = (h_c.float() - h_p.float()).abs().max().item() / scale
    print(f"chain vs five launches, {width} B={B}: {err:.3e} of the scale")
    assert err <= 1.5e-2
    # layer 0's appended cache rows come out of the q|k|v product + RoPE alone: equal bit for bit
    assert torch.equal(k_c[0][:, :T + n_steps], k_p[0][:, :T + n_steps]) and torch.equal(v_c[0][:, :T + n_steps], v_p[0][:, :T + n_steps])


def test_chain_with_ragged_positions_and_long_caches(dev):
    """Every row at its own position (right-padded prompts of different lengths), one of them past the 128 keys whose rows are
    requested before the wait and past a second trip: bit-identical to the stage-by-stage launches, within ulps of the library's."""
    cfg, llm = _model("7b", 2, dev)
    B, T, n_steps = 6, 400, 3
    x = torch.randn((B, T + n_steps, cfg.llm.hidden), generator=torch.Generator().manual_seed(4)).to(dev, torch.bfloat16)
    pos = torch.tensor([400, 1, 17, 129, 255, 320], dtype=torch.int32, device=dev)
    h_c = _run(llm, x, T, n_steps, True, pos)[0]
    h_s = _run(llm, x, T, n_steps, "stages", pos)[0]
    h_p = _run(llm, x, T, n_steps, False, pos)[0]
    assert torch.equal(h_c, h_s)
    assert (h_c.float() - h_p.float()).abs().max().item() <= 1.5e-2 * h_p.float().abs().max().item()


def test_chain_inside_a_hip_graph_equals_eager(dev):
    cfg, llm = _model("7b", 3, dev)
    B, T = 8, 64
    x = torch.randn((B, T + 1, cfg.llm.hidden), generator=torch.Generator().manual_seed(6)).to(dev, torch.bfloat16)
    llm.decode_chain = True
    cache = llm.new_cache(B, T + 8)
    llm.forward(x[:, :T].clone(), cache)
    cache["pos"].fill_(T)
    cache["nk"].fill_(T + 1)
    static_in = x[:, T:T + 1].clone()
    eager = llm.decode_rows(static_in.clone(), cache).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = llm.decode_rows(static_in.clone(), cache)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    from haff import ops
    assert ops.decode_chain_status(cache["chain"]["sync"], 3)


def test_chain_rejects_what_it_cannot_run(dev):
    from haff import ops
    assert ops.decode_chain_supported(8, 4096, 11008, 32, 32) and ops.decode_chain_supported(1, 5120, 13824, 40, 40)
    assert not ops.decode_chain_supported(9, 4096, 11008, 32, 32)       # more rows than the norm-carrying products gather
    assert not ops.decode_chain_supported(8, 4096, 11008, 64, 32)       # head dim != 128
    assert not ops.decode_chain_supported(8, 4096, 11136, 32, 32)       # ffn % 256
    assert not ops.decode_chain_supported(8, 4096, 11008, 32, 49)       # more layers than the kernel arguments hold
