"""End-to-end parity (-m gpu): the HIP scoring pass, through the C ABI, against the CPU oracle
(oracle/clip_t5_oracle.py, pinned to HF by tests/test_oracle_golden.py) on the same seeded weights and inputs.

Tolerances.  BASELINE.json's north_star asks for |delta log P("Yes")| <= 1e-3 against the reference CPU path.
The reference CPU path is itself bf16 end to end (mm_utils.py:228) and, on the committed HF fixtures, sits
0.02-0.1 away from fp32 arithmetic on the same bf16 weights (tests/golden/e2e_*.npz: `logprobs_hf_bf16` vs
`logprobs_fp32`), because every matmul operand is rounded to 8 mantissa bits.  The HIP path has the same
unavoidable operand rounding (bf16 MFMA) but keeps the residual stream, softmax and all reductions in fp32, so:
  * test_fixture_parity_vs_reference_noise_floor: HIP error vs fp32 truth must be <= max(1e-3, the reference's
    own bf16 error vs the same truth), max and mean, on every fixture;
  * test_low_sensitivity_regime_meets_1e3: where the head is not sharply peaked the literal 1e-3 holds;
  * test_end_to_end_vs_oracle: stage-by-stage bounds against the CPU oracle (bf16 storage: 2^-8 relative per
    rounding, a few roundings deep) and LOGPROB_TOL_BF16 * max(1, lm_head gain) on the label log-probs."""
import json
import os

import numpy as np
import pytest
import torch

from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import make_seeded_weights

pytestmark = pytest.mark.gpu

LOGPROB_TOL = 1e-3          # north_star
LOGPROB_TOL_BF16 = 4.0e-3   # 16-bit-operand bound per unit of logit scale on the 128/256-wide test configurations: 2 x the largest measured with the round-5 defaults (0.7e-3 .. 2.0e-3 per unit over the nine cases; round 4: 1.5e-2)
                            # 3.5e-3 .. 5e-3 at gain 1 with the precise decoder (round 3: 2.5e-2 against 4e-3 .. 1e-2)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(cfg, B, n_img, L, T, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    v, t = cfg.vision, cfg.t5
    pix = torch.randn(n_img, 3, v.image, v.image, generator=g).to(torch.bfloat16)
    ids = torch.randint(3, t.vocab, (B, L), generator=g)
    sent = torch.randint(0, L - 1, (B,), generator=g)
    for b in range(B):
        n = L if (not ragged or b == 0) else int(torch.randint(max(2, L // 2), L + 1, (1,), generator=g))
        sp = min(int(sent[b]), n - 2)
        ids[b, sp] = -200
        ids[b, n - 1] = t.eos_id
        ids[b, n:] = 0
    labels = torch.randint(3, t.vocab, (B, T), generator=g)
    labels[:, -1] = t.eos_id
    if ragged and T > 2 and B > 1:
        labels[1, -1] = -100
        labels[1, -2] = t.eos_id
    img_index = torch.randint(0, n_img, (B,), generator=g)
    return pix, img_index, ids, labels


def _record(name, payload):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_e2e.jsonl"), "a") as f:
        f.write(json.dumps({"case": name, **payload}) + "\n")


@pytest.mark.parametrize("name,B,n_img,L,T,gain", [("tiny", 3, 2, 9, 3, 1.0), ("tiny", 5, 3, 12, 2, 4.0),
                                                    ("small", 4, 2, 20, 2, 1.0), ("small", 2, 2, 33, 2, 3.0)])
def test_end_to_end_vs_oracle(name, B, n_img, L, T, gain):
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.engine import VqsEngine

    cfg = get_config(name)
    w = make_seeded_weights(cfg, seed=11, device="cpu", lm_head_gain=gain)
    pix, img_index, ids, labels = _inputs(cfg, B, n_img, L, T, seed=100 + B)
    ref = Oracle(cfg, w).forward(pix.float(), img_index, ids, labels, return_stages=True)

    eng = VqsEngine(cfg, w, device="cuda:0")
    feats = eng.encode_images(pix.cuda())
    lp, sc = eng.score(feats, img_index, ids, labels)
    torch.cuda.synchronize()
    assert int(eng.stage("flags")[0]) == 0

    stats = {}

    def cmp(tag, out, r, atol, rtol, mask=None):
        o = out.detach().float().cpu()
        e = (o - r).abs()
        if mask is not None:
            e = e * mask
        stats[tag] = float(e.max())
        lim = atol + rtol * r.abs()
        bad = e > lim
        assert not torch.isnan(o).any(), f"{tag}: NaN"
        assert not bad.any(), f"{tag}: max err {float(e.max()):.4g} (ref absmax {float(r.abs().max()):.4g}), {int(bad.sum())} bad"

    hid = eng.stage("vit_hidden")
    cmp("vit_feats", hid[:, 1:], ref["vit_feats"], 0.15, 0.03)
    cmp("proj", feats, ref["proj"], 0.15, 0.03)
    m = ref["enc_mask"][..., None].float()
    cmp("enc_out", eng.stage("enc_out"), ref["enc_out"], 0.08, 0.03, m)
    valid = (labels != -100)
    cmp("logits", eng.stage("logits"), ref["logits"], 0.05 * gain, 0.02, valid[..., None].float())
    assert torch.equal(eng.stage("enc_len").cpu().long(), ref["enc_mask"].sum(-1))
    dlp = (lp.cpu() - ref["label_logprobs"]).abs()
    stats["label_logprob"] = float(dlp.max())
    stats["score"] = float((sc.cpu() - ref["scores"]).abs().max())
    stats["ref_logprob_range"] = [float(ref["label_logprobs"].min()), float(ref["label_logprobs"].max())]
    _record(f"{name}-B{B}-L{L}-T{T}-gain{gain}", stats)
    tol = LOGPROB_TOL_BF16 * max(1.0, gain)
    assert dlp.max().item() <= tol, f"|dlogP|={dlp.max().item():.3e} > {tol} ({stats})"
    rel_sc = ((sc.cpu() - ref["scores"]).abs() / ref["scores"].clamp(min=1e-30)).max().item()
    assert rel_sc <= 2 * tol, f"relative score error {rel_sc:.3e}"
    eng.close()


@pytest.mark.parametrize("fixture", ["e2e_tiny_g1", "e2e_tiny_g4", "e2e_small_g1", "e2e_small_g4"])
def test_fixture_parity_vs_reference_noise_floor(golden_dir, fixture):
    """Committed HF-module fixtures (oracle/make_golden.py::golden_e2e): the HIP path must be at least as close to
    fp32 arithmetic as the reference's own bf16 CPU path, and within 1e-3 wherever that path is."""
    from t2v_metrics_amd.engine import VqsEngine
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    cfg = get_config(fixture.split("_")[1])
    w = make_seeded_weights(cfg, seed=int(g["seed"]), device="cpu", lm_head_gain=float(g["gain"]))
    eng = VqsEngine(cfg, w, device="cuda:0")
    feats = eng.encode_images(torch.from_numpy(g["pixels"]).to(torch.bfloat16).cuda())
    lp, sc = eng.score(feats, torch.from_numpy(g["img_index"]), torch.from_numpy(g["ids"]), torch.from_numpy(g["labels"]))
    torch.cuda.synchronize()
    truth = torch.from_numpy(g["logprobs_fp32"])
    e_hip = (lp.cpu() - truth).abs()
    e_ref = (torch.from_numpy(g["logprobs_hf_bf16"]) - truth).abs()
    _record(fixture, {"hip_max": float(e_hip.max()), "hip_mean": float(e_hip.mean()), "hf_bf16_max": float(e_ref.max()),
                      "hf_bf16_mean": float(e_ref.mean()), "logp_range": [float(truth.min()), float(truth.max())]})
    assert e_hip.max().item() <= max(LOGPROB_TOL, e_ref.max().item()), (e_hip.max().item(), e_ref.max().item())
    assert e_hip.mean().item() <= max(LOGPROB_TOL, e_ref.mean().item()), (e_hip.mean().item(), e_ref.mean().item())
    s_truth = torch.from_numpy(g["scores_fp32"])
    assert ((sc.cpu() - s_truth).abs() / s_truth).max().item() <= max(LOGPROB_TOL, 1.5 * e_ref.max().item())
    eng.close()


@pytest.mark.parametrize("fixture", ["generate_tiny_g8", "generate_small_g8"])
def test_greedy_generate_matches_hf_fixture(golden_dir, fixture):
    """vqs_generate against HF T5ForConditionalGeneration.generate(do_sample=False) (oracle/make_golden.py::golden_generate).
    Token ids must be identical up to the first step whose top-1/top-2 logit gap in the fp32 reference is below
    GEN_MARGIN (a bf16 path may legitimately flip a near-tie; everything after such a step is conditioned differently).
    Also: scoring the generated prefix teacher-forced reproduces the same arg-max path (generate == repeated score)."""
    GEN_MARGIN = 0.25
    from t2v_metrics_amd.engine import VqsEngine
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    cfg = get_config(fixture.split("_")[1])
    w = make_seeded_weights(cfg, seed=int(g["seed"]), device="cpu", lm_head_gain=float(g["gain"]))
    eng = VqsEngine(cfg, w, device="cuda:0")
    feats = eng.encode_images(torch.from_numpy(g["pixels"]).to(torch.bfloat16).cuda())
    ids, idx = torch.from_numpy(g["ids"]), torch.from_numpy(g["img_index"])
    max_new = int(g["max_new"])
    toks = eng.generate(feats, idx, ids, max_new).cpu().long()
    ref, margins = torch.from_numpy(g["tokens"]).long(), torch.from_numpy(g["margins"])
    compared = 0
    for b in range(ref.shape[0]):
        for t in range(max_new):
            if margins[b, t] < GEN_MARGIN:
                break
            assert toks[b, t] == ref[b, t], (b, t, toks[b].tolist(), ref[b].tolist(), margins[b].tolist())
            compared += 1
    assert compared >= ref.numel() // 2, f"fixture too ambiguous: only {compared} of {ref.numel()} steps compared"
    # teacher-forced scoring of the generated sequence: every generated token is the arg-max of its own step
    lp, _ = eng.score(feats, idx, ids, toks.to(torch.int32))
    torch.cuda.synchronize()
    logits = eng.stage("logits").float().cpu()              # [B, T, vocab] of the teacher-forced pass
    # generate runs the bf16 decoder incrementally, score the precise decoder over all T rows at once: the two may order a near-tie
    # differently (as either may against fp32), so a step counts only where the teacher-forced top-1 / top-2 gap reaches GEN_MARGIN
    top2 = logits.topk(2, -1).values
    decided = (top2[..., 0] - top2[..., 1]) >= GEN_MARGIN
    assert bool(((logits.argmax(-1) == toks) | ~decided).all()) and int(decided.sum()) >= toks.numel() // 2, (logits.argmax(-1).tolist(), toks.tolist())
    #  what the margin filter lets through is recorded, so that a decoder regression cannot hide behind it unnoticed
    _record(fixture, {"steps_compared": compared, "steps_total": int(ref.numel()), "teacher_forced_steps_decided": int(decided.sum()),
                      "teacher_forced_steps_total": int(toks.numel()), "teacher_forced_argmax_equal_on_decided": int(((logits.argmax(-1) == toks) & decided).sum())})
    eng.close()


@pytest.mark.parametrize("name,B,L,max_new,seed,gain", [("tiny", 3, 10, 40, 31, 8.0), ("small", 3, 21, 24, 32, 16.0)])
def test_long_greedy_generate_over_the_kv_cache_matches_the_oracle(name, B, L, max_new, seed, gain):
    """vqs_generate decodes incrementally (one new row per step, self-attention over a K/V cache in the workspace) and is
    no longer capped at 16 tokens: against the oracle's greedy loop (which recomputes every row every step and is pinned
    token for token to HF generate on the g8 fixtures), ids must be identical up to the first step whose fp32 top-1/top-2
    margin is below GEN_MARGIN; and the first 16 tokens must be what teacher-forced scoring of them arg-maxes to."""
    GEN_MARGIN = 0.25
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config(name)
    w = make_seeded_weights(cfg, seed=seed, device="cpu", lm_head_gain=gain)
    pix, img_index, ids, _ = _inputs(cfg, B, 2, L, 2, seed=200 + B)
    ref, margins = Oracle(cfg, w).generate(pix.float(), img_index, ids, max_new, return_margins=True)
    eng = VqsEngine(cfg, w, device="cuda:0")
    feats = eng.encode_images(pix.cuda())
    toks = eng.generate(feats, img_index, ids, max_new).cpu().long()
    assert toks.shape == (B, max_new)
    compared, longest = 0, 0
    for b in range(B):
        n = 0
        for t in range(max_new):
            if margins[b, t] < GEN_MARGIN:
                break
            assert toks[b, t] == ref[b, t], (b, t, toks[b].tolist(), ref[b].tolist(), margins[b].tolist())
            n += 1
        compared += n
        longest = max(longest, n)
    assert longest >= 20, f"longest compared run {longest}: the test must reach well past the old 16-token cap"
    lp, _ = eng.score(feats, img_index, ids, toks[:, :16].to(torch.int32))
    torch.cuda.synchronize()
    tf = eng.stage("logits").float().cpu().argmax(-1)
    agree = (tf == toks[:, :16]) | (margins[:, :16] < GEN_MARGIN)
    assert bool(agree.all()), (tf.tolist(), toks[:, :16].tolist())
    _record(f"generate-long-{name}", {"steps_compared": compared, "steps_total": B * max_new})
    eng.close()


def test_low_sensitivity_regime_meets_1e3():
    """With an unpeaked head (lm_head gain 0.02: logits ~ N(0, 0.02^2)) the literal north_star tolerance holds."""
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=13, device="cpu", lm_head_gain=0.02)
    pix, img_index, ids, labels = _inputs(cfg, 6, 3, 18, 2, seed=77)
    ref = Oracle(cfg, w).forward(pix.float(), img_index, ids, labels)
    eng = VqsEngine(cfg, w, device="cuda:0")
    lp, sc = eng.score(eng.encode_images(pix.cuda()), img_index, ids, labels)
    torch.cuda.synchronize()
    d = (lp.cpu() - ref["label_logprobs"]).abs().max().item()
    _record("low-sensitivity-small-gain0.02", {"label_logprob": d})
    assert d <= LOGPROB_TOL, d
    eng.close()


def test_reassociated_cross_attention_matches_direct_form():
    """Option cross_mode=0 projects K|V of the encoder output per decoder layer (what HF executes); the default
    reassociates ((q Wk) E^T, (P E) Wv^T).  Same function: both must agree with the oracle and with each other."""
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=17, device="cpu", lm_head_gain=2.0)
    pix, img_index, ids, labels = _inputs(cfg, 7, 3, 21, 3, seed=5)
    ref = Oracle(cfg, w).forward(pix.float(), img_index, ids, labels)["label_logprobs"]
    out = {}
    for mode in ("0", "1"):
        eng = VqsEngine(cfg, w, device="cuda:0", options={"cross_mode": int(mode)})
        lp, _ = eng.score(eng.encode_images(pix.cuda()), img_index, ids, labels)
        torch.cuda.synchronize()
        out[mode] = lp.cpu()
        eng.close()
    d01 = (out["0"] - out["1"]).abs().max().item()
    e0, e1 = (out["0"] - ref).abs().max().item(), (out["1"] - ref).abs().max().item()
    _record("cross-mode", {"direct_vs_oracle": e0, "reassoc_vs_oracle": e1, "direct_vs_reassoc": d01})
    # the direct form runs the bf16 decoder of rounds 1-3 (the precise decoder exists for the reassociated form only): its bound is the
    # round-3 one (2.5e-2 per unit of head gain; measured 2.1e-2 here at gain 2), the default form's the round-4 one (measured 5.7e-3)
    legacy = 2 * 2.5e-2
    assert e0 <= legacy and e1 <= 2 * LOGPROB_TOL_BF16 and d01 <= legacy, (e0, e1, d01)


def test_decoder_split_k_matches_unsplit():
    """The decoder's skinny nn.Linear GEMMs run split-K (fp32 partial slices + a fixed-order reduction) by default;
    option splitk=0 runs them as single GEMMs.  Same function: both agree with the oracle and with each other, and the
    split path is deterministic run to run (no atomics)."""
    import dataclasses
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.engine import VqsEngine
    base = get_config("small")
    cfg = dataclasses.replace(base, name="splitk-test",
                              t5=dataclasses.replace(base.t5, d_model=512, heads=8, d_ff=1024, layers=2, dec_layers=2))
    w = make_seeded_weights(cfg, seed=19, device="cpu", lm_head_gain=2.0)
    pix, img_index, ids, labels = _inputs(cfg, 7, 3, 21, 3, seed=6)
    ref = Oracle(cfg, w).forward(pix.float(), img_index, ids, labels)["label_logprobs"]
    out = {}
    for mode in ("0", "1", "1b"):
        eng = VqsEngine(cfg, w, device="cuda:0", options={"splitk": int(mode[0])})
        lp, _ = eng.score(eng.encode_images(pix.cuda()), img_index, ids, labels)
        torch.cuda.synchronize()
        out[mode] = lp.cpu()
        eng.close()
    assert torch.equal(out["1"], out["1b"])
    d01 = (out["0"] - out["1"]).abs().max().item()
    e0, e1 = (out["0"] - ref).abs().max().item(), (out["1"] - ref).abs().max().item()
    _record("split-k", {"unsplit_vs_oracle": e0, "split_vs_oracle": e1, "unsplit_vs_split": d01})
    assert e0 <= 2 * LOGPROB_TOL_BF16 and e1 <= 2 * LOGPROB_TOL_BF16 and d01 <= 2 * LOGPROB_TOL_BF16, (e0, e1, d01)


def test_precise_decoder_is_closer_to_fp32_than_the_bf16_decoder():
    """Round 4: the scoring decoder holds its activations as split-bf16 / fp32 (option dec_precise, default 1) because the error
    attribution (profiles/r4_error_attribution.md) puts most of the path's |delta log P| into the decoder's bf16 roundings.
    dec_precise=0 is the bf16 decoder of rounds 1-3.  Same function: both agree with the fp32 oracle; over a batch of ragged
    pairs at a peaked head the precise decoder's error is smaller in the mean and stays under the bound the CPU attribution
    measured for this arithmetic; it is bitwise repeatable, and independent of split-K (fp32 partial sums only reorder)."""
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=29, device="cpu", lm_head_gain=4.0)
    pix, img_index, ids, labels = _inputs(cfg, 16, 5, 24, 2, seed=12)
    ref = Oracle(cfg, w).forward(pix.float(), img_index, ids, labels)["label_logprobs"]
    out = {}
    for mode, opts in (("bf16", {"dec_precise": 0}), ("precise", {}), ("precise-b", {}), ("precise-unsplit", {"splitk": 0})):
        eng = VqsEngine(cfg, w, device="cuda:0", options=opts)
        lp, _ = eng.score(eng.encode_images(pix.cuda()), img_index, ids, labels)
        torch.cuda.synchronize()
        out[mode] = lp.cpu()
        eng.close()
    assert torch.equal(out["precise"], out["precise-b"])
    err = {k: (v - ref).abs() for k, v in out.items()}
    _record("precise-decoder", {k: {"max": float(e.max()), "mean": float(e.mean())} for k, e in err.items()})
    assert float(err["bf16"].max()) <= 6e-2          # the bf16 decoder of rounds 1-3 at gain 4 (measured 3.4e-2)
    assert float(err["precise"].mean()) < float(err["bf16"].mean()), (float(err["precise"].mean()), float(err["bf16"].mean()))
    assert float(err["precise"].max()) <= 2.5e-2 and float(err["precise"].mean()) <= 8e-3       # CPU attribution at gain 4 (small): 8.7e-3 / 3.0e-3
    assert float((out["precise"] - out["precise-unsplit"]).abs().max()) <= 2e-3                  # only the bf16 score path can flip a rounding


def test_fp16_vision_tower_is_closer_to_fp32_than_the_bf16_tower():
    """Option vit_fp16 (default 1; 0 = the bf16 tower of rounds 1-3): the vision tower and the projector on IEEE fp16 operands -- three more significant bits than bf16 at the
    same MFMA rate and bytes (CLIP was trained in fp16; the T5 stack is not fp16-safe and stays bf16).  Same function: both agree with the
    fp32 oracle; the image features (what the tower hands to the T5 pass) are several times closer to the oracle's with fp16 operands, the
    label log-probs closer in the mean; bitwise repeatable; switching the option back restores the bf16 tower's bits on the same handle."""
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=31, device="cpu", lm_head_gain=4.0)
    pix, img_index, ids, labels = _inputs(cfg, 16, 5, 24, 2, seed=14)
    o = Oracle(cfg, w)
    ref = o.forward(pix.float(), img_index, ids, labels)
    feats_ref = o.projector(o.vision_features(pix.float()))
    eng = VqsEngine(cfg, w, device="cuda:0")
    out, feats = {}, {}
    try:
        for mode, val in (("bf16", 0), ("fp16", 1), ("fp16-b", 1), ("bf16-again", 0)):
            eng.set_option("vit_fp16", val)
            f = eng.encode_images(pix.cuda())
            lp, _ = eng.score(f, img_index, ids, labels)
            torch.cuda.synchronize()
            out[mode], feats[mode] = lp.cpu(), f.float().cpu()
    finally:
        eng.close()
    assert torch.equal(out["fp16"], out["fp16-b"]) and torch.equal(feats["fp16"], feats["fp16-b"])
    assert torch.equal(out["bf16"], out["bf16-again"]) and torch.equal(feats["bf16"], feats["bf16-again"])
    assert feats["fp16"].dtype == feats["bf16"].dtype and not torch.equal(feats["fp16"], feats["bf16"])
    ferr = {k: float((feats[k] - feats_ref.reshape(feats[k].shape)).abs().mean()) for k in ("bf16", "fp16")}
    err = {k: (out[k] - ref["label_logprobs"]).abs() for k in ("bf16", "fp16")}
    _record("fp16-vision-tower", {"features_mean_abs_err": ferr, "logp": {k: {"max": float(e.max()), "mean": float(e.mean())} for k, e in err.items()}})
    # both towers end in a bf16 feature tensor, so the fp16 tower's feature error is bounded below by that last rounding: well under the
    # bf16 tower's accumulated error, not eight times under
    assert ferr["fp16"] < 0.75 * ferr["bf16"], ferr
    assert float(err["fp16"].max()) <= 2.5e-2 and float(err["fp16"].mean()) <= float(err["bf16"].mean()) * 1.25 + 1e-4, err   # CPU emulation of this case: 3.6e-3 vs 4.8e-3


def test_fp16_encoder_attention_side_is_closer_to_fp32_than_the_bf16_encoder():
    """Option enc_fp16 (round 5, default 1; 0 = the bf16 encoder of rounds 1-4): the T5 encoder's norm outputs, q / k / v, probabilities and
    attention output in IEEE fp16, q|k|v / o / wi on fp16 weight copies; sub-layer outputs, the gated product, wo and the encoder's output
    stay bf16.  Same function: both agree with the fp32 oracle; the encoder's OUTPUT (what the decoder reads) is closer to the oracle's
    with the fp16 attention side; bitwise repeatable; switching back restores the bf16 encoder's bits on the same handle; a GEMM form
    without fp16 instantiations refuses the option by name."""
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.engine import VqsEngine, VqsError
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=37, device="cpu", lm_head_gain=4.0)
    pix, img_index, ids, labels = _inputs(cfg, 16, 5, 24, 2, seed=15)
    o = Oracle(cfg, w)
    ref = o.forward(pix.float(), img_index, ids, labels, return_stages=True)
    eng = VqsEngine(cfg, w, device="cuda:0")
    out, enc = {}, {}
    try:
        assert eng.get_option("enc_fp16") == 1
        feats = eng.encode_images(pix.cuda())
        for mode, val in (("bf16", 0), ("fp16", 1), ("fp16-b", 1), ("bf16-again", 0)):
            eng.set_option("enc_fp16", val)
            lp, _ = eng.score(feats, img_index, ids, labels)
            torch.cuda.synchronize()
            out[mode], enc[mode] = lp.cpu(), eng.stage("enc_out").float().cpu().clone()
        eng.set_option("enc_fp16", 1)
        eng.set_option("gemm_variant", 11)
        with pytest.raises(VqsError, match="fp16.*gemm_variant 3"):      # the first fp16 option the pass meets names itself and the way out
            eng.score(feats, img_index, ids, labels)
    finally:
        eng.close()
    assert torch.equal(out["fp16"], out["fp16-b"]) and torch.equal(enc["fp16"], enc["fp16-b"])
    assert torch.equal(out["bf16"], out["bf16-again"]) and torch.equal(enc["bf16"], enc["bf16-again"])
    assert not torch.equal(enc["fp16"], enc["bf16"])
    # the oracle's encoder output on the ENGINE's own image features would isolate the encoder; the features differ from the oracle's by the
    # tower's noise in both legs alike, so the comparison of the two legs against the oracle's encoder output is still like for like
    mask = ref["enc_mask"].reshape(-1)
    eerr = {k: float((enc[k].reshape(ref["enc_out"].shape[0] * ref["enc_out"].shape[1], -1) - ref["enc_out"].reshape(-1, ref["enc_out"].shape[-1]))[mask].abs().mean())
            for k in ("bf16", "fp16")}
    err = {k: (out[k] - ref["label_logprobs"]).abs() for k in ("bf16", "fp16")}
    _record("fp16-encoder-attention-side", {"enc_out_mean_abs_err": eerr, "logp": {k: {"max": float(e.max()), "mean": float(e.mean())} for k, e in err.items()}})
    assert eerr["fp16"] < 0.9 * eerr["bf16"], eerr                 # two of the encoder's bf16 classes remain (deltas, FFN product) + the bf16 output itself
    assert float(err["fp16"].max()) <= 2.5e-2 and float(err["fp16"].mean()) <= float(err["bf16"].mean()) * 1.25 + 1e-4, err


@pytest.mark.parametrize("stack,option", [("vit", "vit_fp16"), ("enc", "enc_fp16")])
def test_an_activation_beyond_the_fp16_range_never_reaches_a_drop_in_user(stack, option):
    """v_cvt_pk_f16_f32 stores +-inf for |x| >= 65 520 without a trace.  A checkpoint with a planted activation
    of ~1e5 (a) loses that fp16 option AT BIND TIME when it was on by default -- the range proof (engine.fp16_range_proof) cannot bound the site --
    and scores finitely on bf16 operands, with a warning, never an exception; (b) when the caller INSISTS on the option, the pass sets bit 1 of
    its status word and the scores are non-finite rather than silently wrong; (c) the model wrapper then re-scores the batch with the fp16
    options off and returns what the bf16 engine returns."""
    import warnings
    import t2v_metrics_amd as t2v
    from t2v_metrics_amd.engine import VqsEngine
    from tests.test_host_api import FakeTokenizer
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=41, device="cpu")
    if stack == "vit":
        # eight units of layer 0's FFN get a bias of 1e5: the FFN product (an fp16 tensor under the option) holds 1e5 there -- inf in fp16, an
        # ordinary number in bf16, and later LayerNorms bring the stream back to scale.  (Not a norm gain: q.k of 1e10 leaves the range in
        # which the bias-free softmax's single FMA 2^(s c - m) is meaningful in ANY operand type.)
        key = "vision.encoder.layers.0.mlp.fc1.bias"
        b = w[key].float().clone()
        b[:8] = 1.0e5
        w[key] = b.to(torch.bfloat16)
    else:
        key = "encoder.block.0.layer.0.layer_norm.weight"
        w[key] = (w[key].float() * 0 + 1.0e5).to(torch.bfloat16)  # the norm's output (an fp16 tensor under the option) is ~1e5 x a unit-variance row
    pix, img_index, ids, labels = _inputs(cfg, 4, 2, 12, 2, seed=16)
    # (a) defaults: the proof switches the option off at bind time
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        eng = VqsEngine(cfg, w, device="cuda:0")
    try:
        assert option in eng.fp16_auto_off and eng.get_option(option) == 0 and any(option in str(r.message) for r in rec)
        assert not eng.range_proof[option]["holds"] and eng.range_proof[option]["worst_bound"] > 65504.0
        lp, sc_bf16 = eng.score(eng.encode_images(pix.cuda()), img_index, ids, labels)
        torch.cuda.synchronize()
        assert int(eng.stage("flags")[0]) == 0 and bool(torch.isfinite(sc_bf16).all()) and bool(torch.isfinite(lp).all())
    finally:
        eng.close()
    # (b) the caller insists: honoured, and the overflow is reported, not hidden
    eng = VqsEngine(cfg, w, device="cuda:0", options={option: 1})
    try:
        assert eng.get_option(option) == 1 and option not in eng.fp16_auto_off
        lp, sc = eng.score(eng.encode_images(pix.cuda()), img_index, ids, labels)
        torch.cuda.synchronize()
        assert int(eng.stage("flags")[0]) & 2 and not bool(torch.isfinite(sc).all())
    finally:
        eng.close()
    # (c) the wrapper: one warning, the options go off, the batch is scored again -- the reference's behaviour (a score for every finite input)
    model = t2v.VQAScore(model="clip-flant5-xl", device="cuda:0", config=cfg, weights=w, tokenizer=FakeTokenizer(cfg.t5.vocab), image_workers="thread",
                         engine_options={option: 1}).model
    model.load_images = lambda paths: pix[: len(paths)].cuda()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        s1 = model.score_pairs(["a", "b"], [0, 1], ["one caption", "another caption"], ["Yes", "Yes"])
    assert bool(torch.isfinite(s1).all()) and any("re-scored" in str(r.message) for r in rec)
    assert model.engine.get_option(option) == 0
    with warnings.catch_warnings(record=True) as rec2:
        warnings.simplefilter("always")
        s2 = model.score_pairs(["a", "b"], [0, 1], ["one caption", "another caption"], ["Yes", "Yes"])
    assert torch.equal(s1, s2) and not any("re-scored" in str(r.message) for r in rec2)      # warned once; the scorer stays on bf16 operands
    # (d) generate(): a non-finite logit would make argmax emit token 0 silently -- vqs_generate raises status bit 1 and the wrapper runs
    # the call again on bf16 operands; the tokens are the bf16 scorer's
    gen_bf16 = model.generate_ids(["a", "b"], ["one caption", "another caption"], max_new_tokens=4)
    forced = t2v.VQAScore(model="clip-flant5-xl", device="cuda:0", config=cfg, weights=w, tokenizer=FakeTokenizer(cfg.t5.vocab), image_workers="thread",
                          engine_options={option: 1}).model
    forced.load_images = lambda paths: pix[: len(paths)].cuda()
    with warnings.catch_warnings(record=True) as rec3:
        warnings.simplefilter("always")
        gen = forced.generate_ids(["a", "b"], ["one caption", "another caption"], max_new_tokens=4)
    if stack == "enc":          # the generation path runs the encoder (and its fp16 attention side); the tower's overflow reaches it through the features
        assert any("generating" in str(r.message) for r in rec3)
    assert gen == gen_bf16 and forced.engine.get_option(option) == (0 if any("generating" in str(r.message) for r in rec3) else 1)


def test_fused_residual_rmsnorm_matches_separate_kernels():
    """Option fused_norm=1: the T5 encoder's o / wo GEMM epilogues update the residual stream and hand the next RMSNorm's
    operand + row sums of squares to the consuming GEMM (no norm kernel).  Default (0): separate add+norm kernels.
    Same function: both agree with the oracle and with each other; the fused path is bitwise repeatable."""
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=23, device="cpu", lm_head_gain=2.0)
    pix, img_index, ids, labels = _inputs(cfg, 9, 3, 24, 2, seed=8)
    ref = Oracle(cfg, w).forward(pix.float(), img_index, ids, labels)["label_logprobs"]
    out = {}
    for mode in ("0", "1", "1b"):
        # the fused epilogue runs the 8-wave bf16 kernels: the fp16 attention side (option enc_fp16, quad form only) is off for both legs
        eng = VqsEngine(cfg, w, device="cuda:0", options={"fused_norm": int(mode[0]), "enc_fp16": 0})
        lp, _ = eng.score(eng.encode_images(pix.cuda()), img_index, ids, labels)
        torch.cuda.synchronize()
        out[mode] = lp.cpu()
        eng.close()
    assert torch.equal(out["1"], out["1b"])
    d01 = (out["0"] - out["1"]).abs().max().item()
    e0, e1 = (out["0"] - ref).abs().max().item(), (out["1"] - ref).abs().max().item()
    _record("fused-norm", {"separate_vs_oracle": e0, "fused_vs_oracle": e1, "separate_vs_fused": d01})
    assert e0 <= 2 * LOGPROB_TOL_BF16 and e1 <= 2 * LOGPROB_TOL_BF16 and d01 <= 2 * LOGPROB_TOL_BF16, (e0, e1, d01)


def test_deferred_stream_store_is_bitwise_the_stored_form():
    """Default: the post-attention norm of a layer does not write the fp32 stream and the next pre-norm stores
    (hidden + attention delta) + mlp delta (22 instead of 24 bytes per element and layer).  Option norm_defer=0 stores in
    every norm.  The fp32 additions are the same in the same order, so features, log-probs and scores must be bitwise
    equal (vision tower, T5 encoder; the decoder always stores)."""
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=29, device="cpu", lm_head_gain=2.0)
    pix, img_index, ids, labels = _inputs(cfg, 6, 3, 19, 2, seed=10)
    out = {}
    for mode in ("0", "1"):
        eng = VqsEngine(cfg, w, device="cuda:0", options={"norm_defer": int(mode)})
        feats = eng.encode_images(pix.cuda())
        lp, sc = eng.score(feats, img_index, ids, labels)
        torch.cuda.synchronize()
        out[mode] = (feats.cpu().clone(), lp.cpu(), sc.cpu())
        eng.close()
    for a, b in zip(out["0"], out["1"]):
        assert torch.equal(a, b)


def test_engine_matches_hf_golden_fixture(golden_dir):
    """The committed HF-module fixture (tests/golden/hf_tiny.npz): vision hidden_states[-2] from the HIP tower."""
    from t2v_metrics_amd.engine import VqsEngine
    g = np.load(os.path.join(golden_dir, "hf_tiny.npz"))
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=int(g["seed"]), device="cpu")
    eng = VqsEngine(cfg, w, device="cuda:0")
    eng.encode_images(torch.from_numpy(g["pixels"]).to(torch.bfloat16).cuda())
    torch.cuda.synchronize()
    hid = eng.stage("vit_hidden").float().cpu()[:, 1:]      # patch rows (the CLS row of the last layer is not materialised)
    ref = torch.from_numpy(g["vit_hidden_m2"])[:, 1:]
    err = (hid - ref).abs().max().item()
    assert err <= 0.15 + 0.03 * ref.abs().max().item(), err
    eng.close()


def test_a_whole_pass_is_hip_graph_capturable_and_replays_bitwise():
    """vqs_encode_images + vqs_score allocate nothing, never synchronise and read no host memory, so a pass can be captured
    into a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed: same bits as the eager launches, also after the
    inputs are overwritten in place (the graph holds pointers, not values)."""
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=5, device="cpu")
    pix, idx, ids, labels = _inputs(cfg, 5, 3, 14, 2, seed=9)
    pix2, _, ids2, _ = _inputs(cfg, 5, 3, 14, 2, seed=10)
    eng = VqsEngine(cfg, w, device="cuda:0")
    d_pix, d_idx = pix.cuda(), idx.to("cuda", torch.int32)
    d_ids, d_lab = ids.to("cuda", torch.int32).contiguous(), labels.to("cuda", torch.int32).contiguous()

    def step():
        return eng.score(eng.encode_images(d_pix), d_idx, d_ids, d_lab)

    lp_a, sc_a = (t.clone() for t in step())
    d_pix.copy_(pix2.cuda()); d_ids.copy_(ids2.to("cuda", torch.int32))
    lp_b, sc_b = (t.clone() for t in step())
    assert not torch.equal(lp_a, lp_b)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        lp_g, sc_g = step()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(lp_g, lp_b) and torch.equal(sc_g, sc_b)
    d_pix.copy_(pix.cuda()); d_ids.copy_(ids.to("cuda", torch.int32))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(lp_g, lp_a) and torch.equal(sc_g, sc_a)
    eng.close()


def test_dedup_and_reuse_of_image_features():
    """M x N grids score each image once: pairs that share an image give identical results to separate calls."""
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=5, device="cpu")
    pix, _, ids, labels = _inputs(cfg, 4, 2, 8, 2, seed=7, ragged=False)
    eng = VqsEngine(cfg, w, device="cuda:0")
    feats = eng.encode_images(pix.cuda())
    idx = torch.tensor([0, 1, 0, 1])
    lp_all, _ = eng.score(feats, idx, ids, labels)
    lp_all = lp_all.clone()
    lp_01, _ = eng.score(feats, idx[:2], ids[:2], labels[:2])
    torch.cuda.synchronize()
    assert torch.equal(lp_all[:2].cpu(), lp_01.cpu())     # batch-size invariant, bit for bit
    eng.close()


def test_missing_weight_and_bad_prompt_are_reported():
    from t2v_metrics_amd.engine import VqsEngine, VqsError
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=5, device="cpu")
    w2 = dict(w)
    del w2["lm_head.weight"]
    eng = VqsEngine(cfg, w2, device="cuda:0")
    pix, idx, ids, labels = _inputs(cfg, 2, 1, 6, 2, seed=1, ragged=False)
    feats = eng.encode_images(pix.cuda())
    with pytest.raises(VqsError, match="lm_head.weight"):
        eng.score(feats, idx, ids, labels)
    eng.close()
    eng = VqsEngine(cfg, w, device="cuda:0")
    bad = ids.clone()
    bad[0][bad[0] == -200] = 7        # no sentinel
    eng.score(feats, idx, bad, labels)
    torch.cuda.synchronize()
    assert int(eng.stage("flags")[0]) == 1
    eng.close()


def test_drop_in_api_4x4_grid_on_gpu(tmp_path):
    """BASELINE.json configs[0] shape (4 images x 4 prompts through VQAScore.forward) on the HIP engine, against the
    same host pipeline driving the CPU oracle; plus the reference's own smoke assertions (test.py:106-144)."""
    import numpy as np
    from PIL import Image
    import t2v_metrics_amd as t2v
    from tests.test_host_api import FakeTokenizer, OracleEngine
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=29, device="cpu")
    rng = np.random.RandomState(4)
    paths = []
    for i, (h, wd) in enumerate([(256, 256), (300, 200), (120, 180), (64, 64)]):
        p = tmp_path / f"img{i}.png"
        Image.fromarray(rng.randint(0, 256, (h, wd, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))
    texts = ["someone talks on the phone angrily while another person sits happily",
             "someone talks on the phone happily while another person sits angrily",
             "a photo of an astronaut riding a horse", "a dog"]
    tok = FakeTokenizer(cfg.t5.vocab)
    hip = t2v.VQAScore(model="clip-flant5-xl", device="cuda", cache_dir=str(tmp_path / "c1"), config=cfg, weights=w, tokenizer=tok)
    ref = t2v.VQAScore(model="clip-flant5-xl", device="cpu", cache_dir=str(tmp_path / "c2"), config=cfg, tokenizer=tok,
                       engine=OracleEngine(cfg, w))
    s_hip = hip(images=paths, texts=texts)
    s_ref = ref(images=paths, texts=texts)
    assert s_hip.shape == (4, 4) and s_hip.device.type == "cuda" and s_hip.dtype == torch.float32
    assert ((s_hip >= 0) & (s_hip <= 1)).all()
    rel = ((s_hip.cpu() - s_ref.cpu()).abs() / s_ref.cpu()).max().item()
    _record("api-4x4", {"max_rel_score_err": rel})
    assert rel <= 2 * LOGPROB_TOL_BF16, rel          # d(score)/score = d(mean log-prob)
    one = hip(images=paths[0], texts=texts[0])
    assert one.shape == (1, 1) and abs(one.item() - s_hip[0, 0].item()) <= 1e-6
    bf = hip.batch_forward([{"images": paths[:2], "texts": texts[:2]}, {"images": paths[2:], "texts": texts[2:]}], batch_size=2)
    assert bf.shape == (2, 2, 2) and torch.allclose(bf[0].cpu(), s_hip[:2, :2].cpu(), atol=1e-6)
