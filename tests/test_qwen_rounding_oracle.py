"""Pins oracle/qwen25vl_engine_rounding.py (the rounding-matched restatement of the Qwen2.5-VL row in the HIP engine's
layouts) against oracle/qwen25vl_oracle.py (pinned to the HF modules by test_qwen_oracle_golden.py): with every rounding
switched off the two must agree to fp32 accuracy -- on grids with and without partial attention windows, with 32-, 80- and
128-lane heads, grouped-query heads and ragged (right-padded) batches."""
import os

import numpy as np
import pytest
import torch

from oracle.qwen25vl_engine_rounding import QwenEngineRounded, _identity, text_tap_names, vision_tap_names
from oracle.qwen25vl_oracle import QwenOracle
from t2v_metrics_amd.qwen import get_qwen_config
from t2v_metrics_amd.qwen.layout import text_layout, vision_layout
from t2v_metrics_amd.qwen.weights import make_seeded_qwen_weights


def _case(golden_dir, name, fixture):
    z = np.load(os.path.join(golden_dir, fixture + ".npz"))
    cfg = get_qwen_config(name)
    w = make_seeded_qwen_weights(cfg, seed=int(z["seed"]), dtype=torch.bfloat16, lm_head_gain=float(z["gain"]))
    grids = [tuple(int(x) for x in g) for g in z["grids"]]
    return cfg, w, grids, torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]), torch.from_numpy(z["pixel_values"])


def synthetic_case(name, grids, seed=5, n_text=(6, 9), gain=4.0):
    """A seeded batch for a configuration without an HF fixture (one video per sample, right-padded ids)."""
    cfg = get_qwen_config(name)
    w = make_seeded_qwen_weights(cfg, seed=seed, dtype=torch.bfloat16, lm_head_gain=gain)
    g = torch.Generator().manual_seed(seed)
    px = torch.randn(sum(a * b * c for a, b, c in grids), cfg.vision.patch_dim, generator=g).to(torch.bfloat16).float()
    rows = []
    for i, gr in enumerate(grids):
        n_merged = gr[0] * gr[1] * gr[2] // cfg.vision.merge_unit
        pre = torch.randint(10, cfg.text.vocab, (n_text[0] + i,), generator=g)
        post = torch.randint(10, cfg.text.vocab, (n_text[1] + 2 * i,), generator=g)
        rows.append(torch.cat([pre, torch.tensor([cfg.vision_start_token_id]), torch.full((n_merged,), cfg.video_token_id),
                               torch.tensor([cfg.vision_end_token_id]), post]))
    L = max(len(r) for r in rows)
    ids = torch.stack([torch.nn.functional.pad(r, (0, L - len(r))) for r in rows])
    mask = torch.stack([torch.nn.functional.pad(torch.ones(len(r), dtype=torch.long), (0, L - len(r))) for r in rows])
    return cfg, w, list(grids), ids, mask, px


C80_GRIDS = [(2, 12, 20), (2, 16, 16)]        # partial windows in the first, whole ones in the second


def _run(o, cfg, grids, ids, mask, px, record=None):
    merged, off = [], 0
    for g in grids:
        n = g[0] * g[1] * g[2]
        merged.append(o.vision_tower(px[off: off + n], vision_layout(cfg, [g])))
        off += n
    merged = torch.cat(merged)
    return merged, o.text_logits(merged, ids, text_layout(cfg, ids, mask, grids))


@pytest.mark.parametrize("name,fixture", [("qwen-tiny", "qwen_tiny"), ("qwen-small", "qwen_small"), ("qwen-tiny", "qwen_tiny_ragged")])
def test_roundings_off_reproduces_the_fp32_oracle(golden_dir, name, fixture):
    cfg, w, grids, ids, mask, px = _case(golden_dir, name, fixture)
    with torch.no_grad():
        merged, logits = _run(QwenEngineRounded(cfg, w, round_fn=_identity), cfg, grids, ids, mask, px)
    ref = QwenOracle(cfg, w).forward(ids, mask, px, grids, return_stages=True)
    assert (merged - ref["merged"]).abs().max().item() <= 2e-5 * max(1.0, ref["merged"].abs().max().item())
    assert (logits - ref["logits"]).abs().max().item() <= 5e-5 * max(1.0, ref["logits"].abs().max().item())


def test_roundings_off_reproduces_the_fp32_oracle_compact_heads():
    """The 7B tower's head geometry (80-lane heads kept compact in the qkv product and the attention output)."""
    from oracle.qwen25vl_engine_rounding import vision_heads_compact
    cfg, w, grids, ids, mask, px = synthetic_case("qwen-small-c80", C80_GRIDS)
    assert vision_heads_compact(cfg) and vision_heads_compact(get_qwen_config("qwen2.5-vl-7b")) and not vision_heads_compact(get_qwen_config("qwen-small"))
    with torch.no_grad():
        merged, logits = _run(QwenEngineRounded(cfg, w, round_fn=_identity), cfg, grids, ids, mask, px)
    ref = QwenOracle(cfg, w).forward(ids, mask, px, grids, return_stages=True)
    assert (merged - ref["merged"]).abs().max().item() <= 2e-5 * max(1.0, ref["merged"].abs().max().item())
    assert (logits - ref["logits"]).abs().max().item() <= 5e-5 * max(1.0, ref["logits"].abs().max().item())


@pytest.mark.parametrize("name,fixture", [("qwen-tiny", "qwen_tiny_ragged"), ("qwen-small", "qwen_small")])
def test_rounded_pass_sits_at_the_bf16_noise_floor(golden_dir, name, fixture):
    cfg, w, grids, ids, mask, px = _case(golden_dir, name, fixture)
    with torch.no_grad():
        _, logits = _run(QwenEngineRounded(cfg, w), cfg, grids, ids, mask, px)
    ref = QwenOracle(cfg, w).forward(ids, mask, px, grids)
    lp, ref_lp = torch.log_softmax(logits, -1), torch.log_softmax(ref, -1)
    top = ref_lp.argmax(-1)[:, None]
    d = (lp.gather(-1, top) - ref_lp.gather(-1, top)).abs().max().item()
    assert 0 < d <= 5e-2, d


def test_stage_locked_mode_is_exact_on_its_own_record(golden_dir):
    """A free-running pass recorded and replayed as the 'engine' must report zero differences everywhere, with every tap
    name the GPU tests register present -- the plumbing of the stage-locked check, including the layer subset."""
    cfg, w, grids, ids, mask, px = _case(golden_dir, "qwen-tiny", "qwen_tiny_ragged")
    o = QwenEngineRounded(cfg, w)
    g = grids[0]
    n = g[0] * g[1] * g[2]
    lay = vision_layout(cfg, [g])
    o.record = {}
    with torch.no_grad():
        o.vision_tower(px[:n], lay)
    rec = dict(o.record)
    o.record = None
    assert set(vision_tap_names(cfg)) <= set(rec)
    # the engine's ff rows are wider than mlp (zero padding): emulate that
    for i in range(cfg.vision.depth):
        rec[f"vis.{i}.ff"] = torch.nn.functional.pad(rec[f"vis.{i}.ff"], (0, 32))
    rep = o.vision_locked(rec, px[:n], lay)
    assert set(vision_tap_names(cfg)) <= set(rep)
    assert all(r["frac_diff"] == 0.0 and r["pad_nonzero"] == 0 for r in rep.values()), {k: r for k, r in rep.items() if r["frac_diff"]}
    sub = {k: v for k, v in rec.items() if k in set(vision_tap_names(cfg, [1, 3])) | {"vis.merged"}}
    rep = o.vision_locked(sub, px[:n], lay, layers=[1, 3])
    assert "vis.1.attn" in rep and "vis.0.attn" not in rep and all(r["frac_diff"] == 0.0 for r in rep.values())
    # a planted defect is seen at its launch (and at the launch that consumes the changed tensor), nowhere else
    bad = dict(rec)
    bad["vis.2.attn"] = rec["vis.2.attn"] * 1.02
    rep = o.vision_locked(bad, px[:n], lay)
    assert rep["vis.2.attn"]["max_own_ulps"] > 1.5 and rep["vis.2.d_attn"]["frac_diff"] > 0.5
    assert all(r["frac_diff"] == 0.0 for k, r in rep.items() if k not in ("vis.2.attn", "vis.2.d_attn"))
    # text stack
    merged = torch.cat([QwenEngineRounded(cfg, w).vision_tower(px[o_: o_ + gg[0] * gg[1] * gg[2]], vision_layout(cfg, [gg]))
                        for o_, gg in zip(np.cumsum([0] + [a * b * c for a, b, c in grids[:-1]]).tolist(), grids)])
    tl = text_layout(cfg, ids, mask, grids)
    o.record = {}
    with torch.no_grad():
        o.text_logits(merged, ids, tl)
    rec = dict(o.record)
    o.record = None
    assert set(text_tap_names(cfg)) <= set(rec)
    rep = o.text_locked(rec, merged, ids, tl)
    assert set(text_tap_names(cfg)) | {"txt.logits"} <= set(rep) and all(r["frac_diff"] == 0.0 for r in rep.values())


def test_decode_step_with_roundings_off_equals_a_forward_over_the_longer_sequence():
    """text_decode against the KV cache of a (recorded) prefill, roundings off, must reproduce the fp32 oracle run over prompt + the
    new tokens at HF's decode positions -- pins the cache indexing, the grouped-query one-row attention and the position rule."""
    from t2v_metrics_amd.qwen.layout import decode_tables
    from tests.test_qwen_host import OracleQwenEngine
    cfg, w, grids, ids, mask, px = synthetic_case("qwen-tiny", [(2, 8, 8), (2, 8, 12)], seed=9, n_text=(5, 4))
    o = QwenEngineRounded(cfg, w, round_fn=_identity)
    ref = OracleQwenEngine(cfg, w)
    with torch.no_grad():
        merged = torch.cat([o.vision_tower(px[a: a + n], vision_layout(cfg, [g]))
                            for a, n, g in [(0, 128, grids[0]), (128, 192, grids[1])]])
        lay = text_layout(cfg, ids, mask, grids)
        o.record = {}
        o.text_logits(merged, ids, lay)
        rec, o.record = o.record, None
    B, L = ids.shape
    steps, Lmax = 2, L + 2
    t_ = cfg.text
    kc = [torch.zeros(B, t_.kv_heads, Lmax, 128) for _ in range(t_.layers)]
    vc = [torch.zeros(B, t_.kv_heads, Lmax, 128) for _ in range(t_.layers)]
    for i in range(t_.layers):
        kc[i][:, :, :L] = rec[f"txt.{i}.k"]
        vc[i][:, :, :L] = rec[f"txt.{i}.v"]
    _, r_state = ref.prefill(merged, ids, mask, grids, steps + 1)
    length, pos = lay["seq_len"].long().clone(), lay["next_pos"].clone()
    forced = torch.randint(10, t_.vocab, (steps, B), generator=torch.Generator().manual_seed(1))
    for t in range(steps):
        cos, sin = decode_tables(cfg, pos)
        with torch.no_grad():
            lg = o.text_decode(forced[t], kc, vc, length, cos, sin)
        want = ref.decode(r_state, forced[t])
        assert (lg - want).abs().max().item() <= 5e-5 * max(1.0, want.abs().max().item()), t
        length, pos = length + 1, pos + 1


# ------------------------------------------------------------------------------------------------------------------------------
# Round 6: the range-safe fp16 forms.  The proof (csrc/qwen_decode.hip "Bind-time range proof", restated in torch by
# oracle/qwen25vl_engine_rounding.py::range_bounds) bounds every 16-bit activation site from the weights alone; here it is held against
# what the forward pass actually produces, on ordinary and on heavy-tailed weights.
# ------------------------------------------------------------------------------------------------------------------------------
def heavy_tailed(w, cfg, seed=3):
    """A checkpoint with the outliers real language models carry, exaggerated: norm weights with channels x 10^3, a down_proj / proj row
    x 10^3, large q|k|v biases, a few weight columns x 10^2 -- sites whose proven bound exceeds the fp16 range and need a scale."""
    g = torch.Generator().manual_seed(seed)
    w = {k: v.clone().float() for k, v in w.items()}

    def bump(name, rows, factor, dim=0):
        t = w[name]
        idx = torch.randperm(t.shape[dim], generator=g)[:rows]
        if dim == 0:
            t[idx] *= factor
        else:
            t[:, idx] *= factor

    v, t = cfg.vision, cfg.text
    for i in range(v.depth):
        p = f"model.visual.blocks.{i}."
        bump(p + "norm1.weight", 3, 1e3)
        bump(p + "mlp.down_proj.weight", 2, 1e3)
        bump(p + "attn.qkv.bias", 4, 3e3)
    bump("model.visual.merger.ln_q.weight", 2, 1e3)
    for i in range(t.layers):
        p = f"model.language_model.layers.{i}."
        bump(p + "post_attention_layernorm.weight", 3, 1e3)
        bump(p + "input_layernorm.weight", 2, 3e2)
        bump(p + "mlp.down_proj.weight", 2, 1e3)
        bump(p + "self_attn.o_proj.weight", 2, 1e2, dim=1)
        bump(p + "self_attn.k_proj.bias", 4, 2e3)
    return {k: x.to(torch.bfloat16) for k, x in w.items()}


def _observed_site_maxima(cfg, w, grids, ids, mask, px):
    """max |T| of every 16-bit site of a free-running pass with NO rounding (the tensors the proof bounds), by (stack, layer, kind)."""
    from oracle.qwen25vl_engine_rounding import _SITE_OF_TAP
    o = QwenEngineRounded(cfg, w, round_fn=_identity)
    o.record = {}
    with torch.no_grad():
        _run(o, cfg, grids, ids, mask, px)
    seen = {}
    for name, x in o.record.items():
        parts = name.split(".")
        kind = _SITE_OF_TAP.get(parts[-1])
        if kind is None:
            continue
        key = (parts[0], int(parts[1]) if len(parts) == 3 else -1, kind)
        seen[key] = max(seen.get(key, 0.0), float(x.abs().max()))
    return seen


@pytest.mark.parametrize("heavy", [False, True])
def test_range_proof_bounds_hold_on_real_activations(heavy):
    """Every site's proven bound is >= what a pass produces there (and not absurdly loose on ordinary weights: the sum-fed sites are the
    loose ones, which is why they sit behind scales)."""
    from oracle.qwen25vl_engine_rounding import range_bounds, sigma_of_bound
    cfg, w, grids, ids, mask, px = synthetic_case("qwen-small-c80", C80_GRIDS)
    if heavy:
        w = heavy_tailed(w, cfg)
    bounds = range_bounds(cfg, w)
    seen = _observed_site_maxima(cfg, w, grids, ids, mask, px)
    assert set(seen) == set(bounds)
    for key, b in bounds.items():
        assert np.isfinite(b) and seen[key] <= b * (1 + 1e-5), (key, seen[key], b)
        # under its scale the site's largest stored value stays below half of the fp16 maximum
        assert seen[key] * sigma_of_bound(b) <= 32768.0
    if heavy:
        assert sum(sigma_of_bound(b) < 1.0 for b in bounds.values()) >= 8      # the outliers do push sites behind scales
        assert max(seen.values()) > 65504.0                                     # ... and an unscaled fp16 tensor WOULD have overflowed


@pytest.mark.parametrize("heavy", [False, True])
def test_fp16_sites_oracle_is_closer_to_fp32_than_the_bf16_one(heavy):
    """The free-running oracle with the fp16 forms' roundings (fp16 behind each site's scale) beside the bf16 one, on what the 16-bit prefill
    hands to the head -- the merged vision tokens and the language model's final fp32 stream (the head itself is the precise tail's business):
    finite everywhere and several times closer to the unrounded pass, with and without outliers."""
    from oracle.qwen25vl_engine_rounding import range_bounds, sigma_of_bound
    cfg, w, grids, ids, mask, px = synthetic_case("qwen-small-c80", C80_GRIDS)
    if heavy:
        w = heavy_tailed(w, cfg)
    sites = {k: sigma_of_bound(b) for k, b in range_bounds(cfg, w).items()}

    def run(**kw):
        o = QwenEngineRounded(cfg, w, **kw)
        o.record = {}
        with torch.no_grad():
            merged, _ = _run(o, cfg, grids, ids, mask, px)
        return merged, o.record["txt.h_out"]

    m0, h0 = run(round_fn=_identity)
    m16, h16 = run(sites=sites)
    mb, hb = run()
    assert torch.isfinite(h16).all() and torch.isfinite(m16).all()
    valid = mask.reshape(-1).bool()
    rel = lambda a, b: float((a - b).norm() / b.norm())     # noqa: E731
    e16, eb = rel(h16[valid], h0[valid]), rel(hb[valid], h0[valid])
    # ordinary weights: 3 more significant bits everywhere.  Heavy-tailed ones: the worst-case bounds of the sum-fed sites sit 2^30 and more above
    # what a pass produces, small values reach the fp16 subnormals -- still finite and still closer than bf16, by less
    gain = 0.7 if heavy else 0.3
    assert e16 < gain * eb, (e16, eb)
    assert rel(m16, m0) < gain * rel(mb, m0), (rel(m16, m0), rel(mb, m0))
