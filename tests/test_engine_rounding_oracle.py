"""CPU pins of the rounding-matched oracle (oracle/clip_t5_engine_rounding.py), which is what the -m gpu tests hold
the HIP path to at 1e-3: its structure (tiled online softmax with the deferred running max, reassociated
cross-attention, fp32 stream with bf16 deltas) must be the SAME FUNCTION as the HF-pinned fp32 oracle -- identical to
fp32 accuracy when the roundings are switched off -- and with the roundings on it must sit at the bf16 noise floor."""

import numpy as np
import pytest
import torch

from oracle.clip_t5_engine_rounding import EngineRoundedOracle, bf16_round, tiled_attention
from oracle.clip_t5_oracle import Oracle
from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import make_seeded_weights


def _case(name, seed=2):
    cfg = get_config(name)
    w = make_seeded_weights(cfg, seed=1, device="cpu", lm_head_gain=2.0)
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn(3, 3, cfg.vision.image, cfg.vision.image, generator=g).to(torch.bfloat16)
    ids = torch.tensor([[11, 12, -200, 13, 14, 1, 7, 8, 1], [21, -200, 22, 1, 0, 0, 0, 0, 0], [5, 6, 7, -200, 9, 1, 0, 0, 0]])
    labels = torch.tensor([[40, 1, 9], [41, 1, -100], [42, 7, 1]])
    return cfg, w, pix, torch.tensor([0, 2, 1]), ids, labels


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_without_roundings_it_is_the_fp32_oracle(name):
    cfg, w, pix, idx, ids, labels = _case(name)
    ref = Oracle(cfg, w).forward(pix.float(), idx, ids, labels, return_stages=True)
    out = EngineRoundedOracle(cfg, w, round_fn=lambda x: x.float()).forward(pix.float(), idx, ids, labels, return_stages=True)
    m = ref["enc_mask"][..., None].float()
    for k, tol in (("vit_feats", 2e-5), ("proj", 2e-5), ("enc_out", 2e-5), ("dec_out", 2e-5), ("logits", 2e-5), ("label_logprobs", 2e-5)):
        a, b = ref[k], out[k]
        if k == "enc_out":
            a, b = a * m, b * m
        assert (a - b).abs().max().item() <= tol * max(1.0, a.abs().max().item()), k


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_with_roundings_it_sits_at_the_bf16_floor_and_dispatches_from_oracle(name):
    cfg, w, pix, idx, ids, labels = _case(name)
    emu = Oracle(cfg, w, emulate="engine")
    assert isinstance(emu, EngineRoundedOracle)
    ref = Oracle(cfg, w).forward(pix.float(), idx, ids, labels)["label_logprobs"]
    out = emu.forward(pix.float(), idx, ids, labels, return_stages=True)
    d = (out["label_logprobs"] - ref).abs().max().item()
    assert 1e-5 < d < 8e-2, d                    # roundings are really applied, and only roundings
    assert torch.equal(out["proj"], bf16_round(out["proj"]))     # the feature tensor of the C ABI is bf16: exactly representable
    # the encoder's output is an fp16 tensor under option dec_fp16 (the default), a bf16 one without it
    assert emu.dec_fp16 and torch.equal(out["enc_out"], out["enc_out"].half().float()) and not torch.equal(out["enc_out"], bf16_round(out["enc_out"]))
    no_x16 = EngineRoundedOracle(cfg, w, dec_fp16=False).forward(pix.float(), idx, ids, labels, return_stages=True)
    assert torch.equal(no_x16["enc_out"], bf16_round(no_x16["enc_out"]))
    assert emu.vit_fp16 and torch.equal(out["vit_feats"], out["vit_feats"].half().float())     # the tower ships in fp16 (option vit_fp16 = 1)
    bf_tower = EngineRoundedOracle(cfg, w, vit_fp16=False).forward(pix.float(), idx, ids, labels, return_stages=True)
    assert torch.equal(bf_tower["vit_feats"], bf16_round(bf_tower["vit_feats"]))               # rounds 1-3's tower: bf16
    # the decoder's final norm output is a SPLIT-bf16 tensor since round 4 (hi + lo planes): representable as such, not in one plane
    from oracle.clip_t5_engine_rounding import split_bf16_round
    assert torch.equal(out["dec_out"], split_bf16_round(out["dec_out"])) and not torch.equal(out["dec_out"], bf16_round(out["dec_out"]))
    legacy = EngineRoundedOracle(cfg, w, dec_precise=False).forward(pix.float(), idx, ids, labels, return_stages=True)
    assert torch.equal(legacy["dec_out"], bf16_round(legacy["dec_out"]))     # rounds 1-3's decoder: one bf16 plane
    with pytest.raises(ValueError):
        Oracle(cfg, w, emulate="nonsense")


@pytest.mark.parametrize("S,klen,bias", [(70, None, False), (150, [150, 97, 64, 1], True), (608, [608, 601], True)])
def test_tiled_attention_is_softmax_attention(S, klen, bias):
    """Unrounded, the 64-key-tile online softmax with the deferred running max equals softmax(QK^T*scale + bias) V."""
    g = torch.Generator().manual_seed(S)
    B = len(klen) if klen else 2
    H = 3
    q, k, v = (bf16_round(torch.randn(B, H, S, 64, generator=g) * (2.0 if i == 0 else 1.0)) for i in range(3))
    table = torch.randn(H, 2 * S - 1, generator=g) * 2 if bias else None
    kl = torch.tensor(klen) if klen else None
    out = tiled_attention(q, k, v, 0.125 if not bias else 1.0, table, kl, round_fn=lambda x: x.float())
    s = (q.double() @ k.double().transpose(-1, -2)) * (0.125 if not bias else 1.0)
    if bias:
        rel = torch.arange(S)[None, :] - torch.arange(S)[:, None] + (S - 1)
        s = s + table[:, rel][None].double()
    if klen:
        s = s.masked_fill((torch.arange(S)[None, :] >= kl[:, None])[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v.double()).transpose(1, 2).reshape(B, S, H * 64).float()
    assert (out - ref).abs().max().item() <= 2e-5
    # with P rounded to bf16 the result moves by bf16 noise only, and stays a convex combination of V rows
    out_r = tiled_attention(q, k, v, 0.125 if not bias else 1.0, table, kl)
    assert 0 < (out_r - ref).abs().max().item() <= 0.05
    assert out_r.abs().max().item() <= v.abs().max().item() * 1.01


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_stage_locked_run_on_its_own_record_reports_zero(name):
    """Plumbing of the stage-locked mode (tap names, shapes, masks): fed with the intermediates of its own free-running
    pass it must find every op result identical, cover every tap of tap_shapes(), and return the same log-probs."""
    cfg, w, pix, idx, ids, labels = _case(name)
    emu = EngineRoundedOracle(cfg, w)
    emu.record = {}
    free = emu.forward(pix.float(), idx, ids, labels)
    rec, emu.record = emu.record, None
    hi = {n: rec.pop(n) for n in list(rec) if n.endswith("#hi")}
    assert sorted(hi) == sorted(f"dec.{i}.xn1#hi" for i in range(cfg.t5.dec_layers))
    shapes = emu.tap_shapes(pix.shape[0], ids.shape[0], ids.shape[1], labels.shape[1])
    assert set(shapes) | {"proj", "enc_out", "dec_out", "logits"} == set(rec)
    for n, (shape, dt) in shapes.items():
        assert rec[n].numel() == int(np.prod(shape)), n
    # hand the taps over the way the engine does: flat 2-D buffers in the engine's dtypes
    def as_engine(n):                     # split tensors: the engine hands over two bf16 planes; here their exact sum
        if n not in shapes:
            return rec[n]
        shape, dt = shapes[n]
        return rec[n].reshape(shape) if dt == "split" else rec[n].reshape(shape).to(dt)
    taps = {n: as_engine(n) for n in rec}
    taps.update(hi)                                      # the hi planes the score path reads ("<name>#hi", as taps_to_values hands them over)
    report, lp = emu.forward_locked(taps, pix.float(), idx, ids, labels)
    assert set(report) == set(rec)
    assert all(r["max_abs"] == 0.0 for r in report.values()), {n: r for n, r in report.items() if r["max_abs"] > 0}
    assert torch.equal(lp, free["label_logprobs"])
    # and it localises a planted defect: the op that produced the tensor and the one op that consumes it, nothing else
    taps["enc.1.attn"] = taps["enc.1.attn"].clone()
    taps["enc.1.attn"][0, :8] += 0.5
    report, _ = emu.forward_locked(taps, pix.float(), idx, ids, labels)
    off = [n for n, r in report.items() if r["max_abs"] > 0]
    assert off == ["enc.1.attn", "enc.1.d_attn"], off


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_fp16_tower_mode_is_the_same_function_with_finer_roundings(name):
    """`vit_fp16=True` models the engine's option of that name: the tower's and the projector's tensors are IEEE fp16 (11 significant
    bits), their linear weights fp16 copies, the feature tensor still bf16.  Roundings off = the fp32 oracle; roundings on = closer to it
    than the bf16 tower; the tower's taps are fp16 tensors; a stage-locked run on its own record reports zero, with fp16 ulps."""
    cfg, w, pix, idx, ids, labels = _case(name)
    o = Oracle(cfg, w)
    ref = o.forward(pix.float(), idx, ids, labels, return_stages=True)
    off = EngineRoundedOracle(cfg, w, round_fn=lambda x: x.float(), vit_fp16=True).forward(pix.float(), idx, ids, labels, return_stages=True)
    for k in ("vit_feats", "proj", "label_logprobs"):
        assert (ref[k] - off[k]).abs().max().item() <= 5e-5 * max(1.0, ref[k].abs().max().item()), k   # fp16 weight copies: exact above 2^-14
    bf = EngineRoundedOracle(cfg, w, vit_fp16=False).forward(pix.float(), idx, ids, labels, return_stages=True)
    emu = Oracle(cfg, w, emulate="engine")                               # the default IS the fp16 tower
    assert emu.vit_fp16
    emu.record = {}
    hf = emu.forward(pix.float(), idx, ids, labels, return_stages=True)
    rec, emu.record = emu.record, None
    e_bf, e_hf = (bf["vit_feats"] - ref["vit_feats"]).abs().mean().item(), (hf["vit_feats"] - ref["vit_feats"]).abs().mean().item()
    assert e_hf < 0.3 * e_bf, (e_hf, e_bf)                               # the tower's own output: three more bits
    assert torch.equal(hf["vit_feats"], hf["vit_feats"].half().float()) and not torch.equal(hf["vit_feats"], bf16_round(hf["vit_feats"]))
    assert torch.equal(hf["proj"], bf16_round(hf["proj"]))                # the feature tensor of the C ABI stays bf16
    assert (hf["proj"] - ref["proj"]).abs().mean().item() < (bf["proj"] - ref["proj"]).abs().mean().item()
    # taps: the tower's 16-bit tensors are fp16, everything else what it was
    shapes = emu.tap_shapes(pix.shape[0], ids.shape[0], ids.shape[1], labels.shape[1])
    base = EngineRoundedOracle(cfg, w, vit_fp16=False, enc_fp16=False, dec_fp16=False).tap_shapes(pix.shape[0], ids.shape[0], ids.shape[1], labels.shape[1])
    assert set(shapes) == set(base)
    for n in shapes:
        tower16 = n.startswith("vit.") and n not in ("vit.patch_out", "vit.h0")
        enc16 = n.startswith("enc.") and n.split(".")[-1] in ("xn0", "q", "k", "v", "attn", "xn1")      # option enc_fp16 (default): the attention side
        dec16 = n.startswith("dec.") and n.split(".")[-1] in ("cq", "cqk", "cprobs")                    # option dec_fp16 (default): the cross score path
        assert shapes[n][0] == base[n][0] and shapes[n][1] == (torch.float16 if (tower16 or enc16 or dec16) else base[n][1]), n
    hi = {n: rec.pop(n) for n in list(rec) if n.endswith("#hi")}
    taps = {n: (rec[n].reshape(shapes[n][0]) if n in shapes and shapes[n][1] == "split" else rec[n].reshape(shapes[n][0]).to(shapes[n][1]) if n in shapes else rec[n])
            for n in rec}
    taps.update(hi)
    report, lp = emu.forward_locked(taps, pix.float(), idx, ids, labels)
    assert all(r["max_abs"] == 0.0 for r in report.values()), {n: r for n, r in report.items() if r["max_abs"] > 0}
    assert torch.equal(lp, hf["label_logprobs"])
    # the stage-locked run on its own record carries enc_out as fp32 values of an fp16 tensor: the count here is of taps DECLARED fp16
    assert sum(r["mant_bits"] == 10 for r in report.values()) == 9 * cfg.vision.layers_run + 2 + 6 * cfg.t5.layers + 3 * cfg.t5.dec_layers
    # one fp16 ulp planted in a tower tensor is seen as one ulp (with bf16 ulps it would read as 1/8 and pass any bound)
    t = taps["vit.0.mid"].clone()
    k = int(t.abs().float().argmax())
    t.view(-1)[k] = (t.view(-1)[k].view(torch.int16) + 1).view(torch.float16)            # the next fp16 value away from zero
    taps["vit.0.mid"] = t
    report, _ = emu.forward_locked(taps, pix.float(), idx, ids, labels)
    assert 0.99 <= report["vit.0.mid"]["max_own_ulps"] <= 1.01, report["vit.0.mid"]


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_fp16_encoder_attention_side_mode(name):
    """`enc_fp16=True` (the engine's option of that name, default since round 5): the encoder's norm outputs, q / k / v, probabilities and
    attention output are IEEE fp16 and q / k / v / o / wi read fp16 weight copies; sub-layer outputs, the gated product and the encoder's
    output stay bf16.  Roundings off = the fp32 oracle; the encoder's output is closer to fp32 than with the bf16 attention side."""
    cfg, w, pix, idx, ids, labels = _case(name)
    ref = Oracle(cfg, w).forward(pix.float(), idx, ids, labels, return_stages=True)
    off = EngineRoundedOracle(cfg, w, round_fn=lambda x: x.float()).forward(pix.float(), idx, ids, labels, return_stages=True)
    assert (ref["label_logprobs"] - off["label_logprobs"]).abs().max().item() <= 5e-5
    e = {}
    for flag in (False, True):
        o = EngineRoundedOracle(cfg, w, enc_fp16=flag, dec_fp16=False, classes=tuple(c for c in EngineRoundedOracle.CLASSES if c.startswith("enc.")))
        assert o.enc_fp16 is flag and (set(o.ENC_FP16_CLASSES) <= o.half_extra) is flag
        o.record = {}
        out = o.forward(pix.float(), idx, ids, labels, return_stages=True)
        e[flag] = (out["enc_out"] - ref["enc_out"]).abs().mean().item()
        rec = o.record
        x = rec["enc.0.xn0"]
        assert torch.equal(x, x.half().float()) is flag or torch.equal(x, bf16_round(x))              # fp16 operand / bf16 operand
        for nm in ("d_attn", "ff", "d_ff"):                                                           # stay bf16 either way
            assert torch.equal(rec[f"enc.0.{nm}"], bf16_round(rec[f"enc.0.{nm}"])), nm
        assert torch.equal(out["enc_out"], bf16_round(out["enc_out"]))
    assert e[True] < 0.8 * e[False], e


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_fp16_cross_attention_score_path_mode(name):
    """`dec_fp16=True` (the engine's option, default since round 5; precise decoder only): the encoder's output, the cross q (from the full
    split norm output), q.Wk and the probabilities are IEEE fp16 tensors; everything else of the precise decoder is what it was.  Roundings
    off = the fp32 oracle; with the decoder's classes alone rounding, the label log-probs are closer to fp32 than with the bf16 score path."""
    cfg, w, pix, idx, ids, labels = _case(name)
    ref = Oracle(cfg, w).forward(pix.float(), idx, ids, labels, return_stages=True)
    off = EngineRoundedOracle(cfg, w, round_fn=lambda x: x.float()).forward(pix.float(), idx, ids, labels, return_stages=True)
    assert (ref["label_logprobs"] - off["label_logprobs"]).abs().max().item() <= 5e-5
    assert not EngineRoundedOracle(cfg, w, dec_precise=False).dec_fp16                      # the option exists for the precise decoder only
    dec = tuple(c for c in EngineRoundedOracle.CLASSES if c.startswith("dec.")) + ("enc.out",)
    e = {}
    for flag in (False, True):
        o = EngineRoundedOracle(cfg, w, dec_fp16=flag, classes=dec)
        o.record = {}
        out = o.forward(pix.float(), idx, ids, labels, return_stages=True)
        rec = o.record
        for nm in ("cq", "cqk", "cprobs"):
            x = rec[f"dec.0.{nm}"]
            assert torch.equal(x, x.half().float()) if flag else torch.equal(x, bf16_round(x)), nm
        assert (torch.equal(out["enc_out"], out["enc_out"].half().float()) and not torch.equal(out["enc_out"], bf16_round(out["enc_out"]))) is flag
        e[flag] = (out["label_logprobs"] - ref["label_logprobs"]).abs().mean().item()
    assert e[True] < e[False], e
