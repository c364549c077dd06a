"""Full-size checks (-m gpu) at the real clip-flant5-xl / -xxl architectures (BASELINE.json configs[1] / configs[2],
the metric's model): the oracle is too slow for whole batches here, so parity rests on size-independent properties
plus one-pair three-way comparisons (HIP, rounding-matched CPU oracle, fp32 oracle) with a random and a PEAKED head.  The
literal 1e-3 of north_star is asserted stage-locked for every launch of a full-size pass in tests/test_gpu_stage_locked.py
(why not end to end: tests/test_gpu_parity_noise_floor.py)."""
import json
import os

import pytest
import torch

from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import make_seeded_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xl():
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("clip-flant5-xl")
    w = make_seeded_weights(cfg, seed=0, device="cuda:0")
    eng = VqsEngine(cfg, w, device="cuda:0")
    yield cfg, w, eng
    eng.close()


def _batch(cfg, B, n_img, L, seed):
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn(n_img, 3, cfg.vision.image, cfg.vision.image, generator=g).to(torch.bfloat16)
    ids = torch.randint(3, 32100, (B, L), generator=g)
    for b in range(B):
        n = L if b % 3 == 0 else int(torch.randint(L // 2, L + 1, (1,), generator=g))
        ids[b, int(torch.randint(0, n - 1, (1,), generator=g))] = -200
        ids[b, n - 1] = 1
        ids[b, n:] = 0
    labels = torch.tensor([[2163, 1]] * B)
    idx = torch.randint(0, n_img, (B,), generator=g)
    return pix, idx, ids, labels


def test_batch_composition_padding_and_image_order_invariance(xl):
    cfg, w, eng = xl
    pix, idx, ids, labels = _batch(cfg, 16, 4, 33, seed=3)
    feats = eng.encode_images(pix.cuda())
    lp16, sc16 = eng.score(feats, idx, ids, labels)
    lp16, sc16 = lp16.clone(), sc16.clone()
    assert torch.isfinite(lp16).all() and (lp16 <= 0).all() and ((sc16 > 0) & (sc16 <= 1)).all()
    # (a) a pair's result does not depend on what else is in the batch
    lp4, _ = eng.score(feats, idx[:4], ids[:4], labels[:4])
    assert torch.equal(lp4, lp16[:4])
    # (b) ... nor on where its image sits among the encoded images
    perm = torch.tensor([2, 0, 3, 1])
    feats_p = eng.encode_images(pix[perm].cuda())
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(4)
    lp_p, _ = eng.score(feats_p, inv[idx], ids, labels)
    assert torch.equal(lp_p, lp16)
    # (c) ... nor on extra right padding of the prompts (longer static encoder length, more masked keys)
    ids_pad = torch.cat([ids, torch.zeros(16, 7, dtype=ids.dtype)], dim=1)
    lp_pad, _ = eng.score(feats, idx, ids_pad, labels)
    assert (lp_pad - lp16).abs().max().item() <= 1e-4
    # (d) score = exp(mean label log-prob)
    assert torch.allclose(sc16, torch.exp(lp16.mean(-1)), rtol=1e-5, atol=0)


def _record(name, payload):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_e2e.jsonl"), "a") as f:
        f.write(json.dumps({"case": name, **payload}) + "\n")


def _label_logprobs(dec_out, head, labels, acc=torch.float32):
    """fp32 label log-probs of an lm_head [V, D] over a final decoder state [B, T, D] (the head is the only thing the regimes
    below change, and the decoder input -- shift_right(labels) -- does not depend on it: one oracle pass serves them all)."""
    logits = (dec_out.to(acc) @ head.to(acc).t()).float()
    return torch.log_softmax(logits, -1).gather(-1, labels.long().unsqueeze(-1)).squeeze(-1)


def _one_pair_three_way(cfg, w, eng, tag, seed, check_random=True):
    """HIP vs the fp32 oracle for one pair of the full-size model, with the rounding-matched oracle (the engine's arithmetic on
    the CPU, free-running) beside it as the calibrated noise level.  Four heads over the same pass:
      random-head     the seeded lm_head: log P ~ -10.4 (near-uniform over 32 128 tokens);
      planted-self    lm_head rows of the two labels := 12 x / |x|^2 with x the ENGINE's own final decoder state ("planted
                      direction"): P(label) 0.2-0.8, the regime a real checkpoint scores in -- and the
                      label logit is first-order insensitive to upstream error by construction;
      planted-oracle  the same with x taken from the FP32 ORACLE's decoder state: the direction is independent of the engine;
                      the state sits behind an RMSNorm, so an upstream relative error eps moves the logit by 12 eps^2 / 2;
      planted-tilted  direction at atan(3) ~ 72 degrees from the oracle's state (x + 3 |x| r, r a seeded unit vector orthogonal
                      to x), gain such that the fp32 logit is 12: first-order sensitive -- the logit moves by 12 * 3 * (the
                      state's relative error along r, ~ its norm-wise error / sqrt(D) for a direction drawn at random).  Held
                      to the calibrated noise level and recorded for DESIGN.md, like the random head.
    north_star's 1e-3 is asserted in the two planted regimes where it is well defined."""
    from oracle.clip_t5_oracle import Oracle
    pix, idx, ids, labels = _batch(cfg, 1, 1, 33, seed=seed)
    head = w["lm_head.weight"]
    saved = head[[2163, 1]].clone()
    # Both oracles are evaluated by torch ON THE DEVICE (the same code; the host cores need ~40 s per XXL pass and pair, the device a
    # second): test infrastructure after the engine's pass.  tests/test_oracle_golden.py and bench.py's cross-check pin the device
    # evaluation of this code to the host's (1.9e-6 on log-probs).
    dev = torch.device("cuda:0")
    with torch.device(dev):
        ref = Oracle(cfg, w, device=dev).forward(pix.float().to(dev), idx.to(dev), ids.to(dev), labels.to(dev), return_stages=True)
        emu = Oracle(cfg, w, emulate="engine", device=dev).forward(pix.float().to(dev), idx.to(dev), ids.to(dev), labels.to(dev), return_stages=True)
    ref = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in ref.items()}
    emu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in emu.items()}
    head_cpu = w["lm_head.weight"].float().cpu().clone()
    x_ref = ref["dec_out"].float()[0]                                            # [T, D]
    out = {}
    try:
        lp, sc = eng.score(eng.encode_images(pix.cuda()), idx, ids, labels)
        torch.cuda.synchronize()
        x_eng = eng.stage("dec_out").float()[0].cpu()
        g = torch.Generator().manual_seed(seed + 1000)
        regimes = {"random-head": None, "planted-self": x_eng, "planted-oracle": x_ref}
        tilt = []
        for t in range(2):
            r = torch.randn(x_ref.shape[1], generator=g)
            r = r - (r @ x_ref[t]) / (x_ref[t] @ x_ref[t]) * x_ref[t]
            tilt.append(x_ref[t] + 3.0 * x_ref[t].norm() * r / r.norm())
        regimes["planted-tilted"] = torch.stack(tilt)
        for regime, xdir in regimes.items():
            if regime == "random-head" and not check_random:
                continue
            hc = head_cpu.clone()
            if xdir is not None:
                for t, row in enumerate((2163, 1)):
                    v = 12.0 * xdir[t] / (xdir[t] @ x_ref[t])                    # fp32 logit of the label = 12 (planted-self: ~12)
                    head[row] = v.to(head.dtype).to(head.device)
                    hc[row] = v.to(head.dtype).float()
                lp, sc = eng.score(eng.encode_images(pix.cuda()), idx, ids, labels)     # lm_head is read in place
                torch.cuda.synchronize()
            lp_ref = _label_logprobs(ref["dec_out"].float(), hc, labels)
            lp_emu = _label_logprobs(emu["dec_out"].float(), hc, labels, acc=torch.float64)
            out[regime] = {"logp_hip": lp.cpu().tolist(), "logp_fp32": lp_ref.tolist(),
                           "dlogp_vs_fp32": (lp.cpu() - lp_ref).abs().max().item(),
                           "dlogp_vs_rounding_matched": (lp.cpu() - lp_emu).abs().max().item(),
                           "rounding_matched_vs_fp32": (lp_emu - lp_ref).abs().max().item()}
            head[[2163, 1]] = saved
    finally:
        head[[2163, 1]] = saved
    rel = ((x_eng - x_ref).norm(dim=-1) / x_ref.norm(dim=-1)).max().item()
    out["decoder_state_relative_error_vs_fp32"] = rel
    _record("fullsize/" + tag, out)
    for regime in ("planted-self", "planted-oracle", "planted-tilted"):
        assert out[regime]["logp_fp32"][0][0] > -3.0, out                        # the planted regimes really are peaked
    # north_star's literal tolerance |delta log P| <= 1e-3 in EVERY peaked regime (P(label) 0.2-0.8, where a real checkpoint's
    # "Yes" sits): direction = the decoder state (engine's or oracle's; measured 1e-5 .. 3e-5 with the precise decoder of round 4)
    # and the first-order sensitive tilted direction (measured 1.9e-4 / 2.1e-4; 3.9e-4 / 2.1e-3 with the bf16 decoder of round 3)
    assert out["planted-self"]["dlogp_vs_fp32"] <= 1e-4, out
    assert out["planted-oracle"]["dlogp_vs_fp32"] <= 1e-4, out
    assert out["planted-tilted"]["dlogp_vs_fp32"] <= 1e-3, out
    # the near-uniform seeded head (log P ~ -10.4: every one of 32 128 logits enters the normaliser): what is left is the bf16
    # operand noise of the vision tower and the encoder (profiles/r4_error_attribution.md: 1.1e-3 + 0.6e-3 at XL).  Measured
    # 1.0e-3 (XL) / 1.4e-3 (XXL); gate = 3 x that, and never more than 3 x what the same arithmetic shows on the CPU + 1e-3
    if "random-head" in out:
        o = out["random-head"]
        # round 5 (fp16 tower + fp16 encoder attention side): one pair is one draw of a distribution whose maximum over 256 bench pairs
        # this pair measures 2.3e-4 (XXL) / 3.3e-4 (XL); over 256 pairs of the bench batch max 5.2e-4 / 5.8e-4 (profiles/r5_call7_*): the gate is
        # north_star's 1e-3 itself (round 4: 4.5e-3, round 3: 2.5e-2)
        assert o["dlogp_vs_fp32"] <= min(1.0e-3, 3.0 * o["rounding_matched_vs_fp32"] + 1e-3), out
    assert rel <= 0.02, rel


def test_xl_one_pair_three_way_random_and_peaked_head(xl):
    cfg, w, eng = xl
    _one_pair_three_way(cfg, w, eng, "clip-flant5-xl", seed=9)


def test_xxl_one_pair_three_way_random_and_peaked_head():
    """The metric's model (BASELINE.json: CLIP-FlanT5-XXL): 11.5 B parameters, 23 GB of bf16 weights on the device; the
    oracles up-cast one tensor at a time on the host."""
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("clip-flant5-xxl")
    w = make_seeded_weights(cfg, seed=0, device="cuda:0")
    eng = VqsEngine(cfg, w, device="cuda:0")
    try:
        _one_pair_three_way(cfg, w, eng, "clip-flant5-xxl", seed=10)
    finally:
        eng.close()
        del w
        torch.cuda.empty_cache()
