"""CPU tests of the round-4 parity tooling: the per-class switches of the rounding-matched oracle (what tools/error_attribution.py
varies), the split-bf16 rounding, and the arithmetic of bench.py's |delta log P| table (parity_sample, parity_jobs) on an engine double."""
import pytest
import torch

from oracle.clip_t5_engine_rounding import EngineRoundedOracle, bf16_round, split_bf16_round
from oracle.clip_t5_oracle import Oracle
from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import make_seeded_weights


def _case(name="tiny", seed=3):
    cfg = get_config(name)
    w = make_seeded_weights(cfg, seed=seed, device="cpu", lm_head_gain=2.0)
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn(2, 3, cfg.vision.image, cfg.vision.image, generator=g).to(torch.bfloat16).float()
    ids = torch.tensor([[11, 12, -200, 13, 14, 1, 7, 1], [21, -200, 22, 1, 0, 0, 0, 0], [5, 6, 7, -200, 9, 1, 0, 0]])
    labels = torch.tensor([[40, 1], [41, 1], [42, 1]])
    return cfg, w, pix, torch.tensor([0, 1, 1]), ids, labels


def test_split_bf16_round_carries_sixteen_bits():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.logspace(-6, 6, 4096)
    y = split_bf16_round(x)
    hi = bf16_round(x)
    assert torch.equal(bf16_round(y - hi), y - hi)                          # y = hi + a bf16 value
    rel = ((y - x).abs() / x.abs()).max().item()
    assert rel <= 2.0 ** -16 and ((bf16_round(x) - x).abs() / x.abs()).max().item() > 2.0 ** -10
    assert torch.equal(split_bf16_round(hi), hi)                            # bf16 values pass through unchanged
    # bf16(hi + lo) is NOT always hi: lo is rounded itself and can put the sum exactly on a tie (why the oracle carries the hi plane)
    assert (bf16_round(y) != hi).any()


def test_rounding_classes_switch_individually_and_the_precise_decoder_is_closer():
    cfg, w, pix, idx, ids, labels = _case()
    ref = Oracle(cfg, w).forward(pix, idx, ids, labels)["label_logprobs"]
    run = lambda **kw: EngineRoundedOracle(cfg, w, acc=torch.float32, **kw).forward(pix, idx, ids, labels)["label_logprobs"]
    none = run(classes=())
    assert (none - ref).abs().max().item() <= 2e-5                          # every class off = the fp32 oracle
    errs = {}
    for c in EngineRoundedOracle.CLASSES:
        errs[c] = (run(classes=(c,), dec_precise=False) - none).abs().max().item()
        assert errs[c] > 0.0, c                                             # every class is a live rounding site
    assert abs((run(dec_precise=False) - run(classes=EngineRoundedOracle.CLASSES, dec_precise=False)).abs().max().item()) == 0.0
    dec = [c for c in EngineRoundedOracle.CLASSES if c.startswith("dec.")]
    legacy = (run(classes=dec, dec_precise=False) - none).abs().max().item()
    precise = (run(classes=dec, dec_precise=True) - none).abs().max().item()
    assert precise < 0.25 * legacy, (precise, legacy)                       # the decoder's own contribution shrinks by >= 4x
    with pytest.raises(ValueError):
        EngineRoundedOracle(cfg, w, classes=("dec.nope",))


def test_what_if_switches_of_the_attribution_tool():
    """`split_classes` / `half_classes` / `vit_fp16` (what tools/error_attribution.py's what-if runs vary): a stage modelled as split-bf16 or
    fp16 tensors sits between its bf16 model and exact; "vit.v" splits the value heads only; unknown names are refused; the engine's option
    (vit_fp16 = True, the default) is the all-fp16 tower with the projector's result rounded to fp16 and then to the bf16 feature tensor."""
    cfg, w, pix, idx, ids, labels = _case()
    feats = lambda **kw: EngineRoundedOracle(cfg, w, acc=torch.float32, dec_precise=True, **kw).vision_features(pix)
    exact = feats(classes=(), vit_fp16=False)
    vit = tuple(c for c in EngineRoundedOracle.CLASSES if c.startswith("vit."))
    err = lambda f: (f - exact).abs().mean().item()
    e_bf = err(feats(classes=vit, vit_fp16=False))
    e_half = err(feats(classes=vit, vit_fp16=False, half_classes=vit))
    e_split = err(feats(classes=vit, vit_fp16=False, split_classes=("vit.norm", "vit.v", "vit.attn", "vit.act", "vit.delta", "vit.feat")))
    e_split_no_v = err(feats(classes=vit, vit_fp16=False, split_classes=("vit.norm", "vit.attn", "vit.act", "vit.delta", "vit.feat")))
    assert 0.0 < e_half < 0.25 * e_bf and 0.0 < e_split < 0.5 * e_bf and e_split < e_split_no_v < e_bf, (e_bf, e_half, e_split, e_split_no_v)   # q, k, P stay bf16 in the split model
    assert torch.equal(feats(classes=vit, vit_fp16=False, half_classes=vit), feats(classes=vit, vit_fp16=True))      # the option = every tower class in fp16
    o = EngineRoundedOracle(cfg, w, acc=torch.float32)
    assert o.vit_fp16 and "proj.mid" in o.half_extra and "proj.out" not in o.half_extra
    proj = o.projector(o.vision_features(pix))
    assert torch.equal(proj, bf16_round(proj))                              # the feature tensor handed to the T5 pass stays bf16
    for bad in ({"split_classes": ("vit.nope",)}, {"half_classes": ("vit.v",)}, {"half_classes": ("enc.nope",)}):
        with pytest.raises(ValueError):
            EngineRoundedOracle(cfg, w, **bad)


def test_bench_parity_sample_table_on_an_engine_double():
    """parity_sample: truth = the oracle on the job's device, HIP logits = the engine's `logits` stage; head gains re-read both."""
    import bench
    cfg, w, pix, idx, ids, labels = _case()
    pix3 = pix[[0, 1, 1]]                                                    # one image per pair, as the bench batch has
    truth = Oracle(cfg, w).forward(pix3, torch.arange(3), ids, labels, return_stages=True)

    class Eng:
        scored = 0

        def encode_images(self, pixels):
            return pixels

        def score(self, feats, img_index, ids_, labels_):                   # parity_sample re-scores the job it tabulates
            self.scored += 1

        def stage(self, name):
            assert name == "logits" and self.scored
            return truth["logits"] + 1e-3 * torch.sign(truth["logits"])      # a known perturbation

    job = (pix3.to(torch.bfloat16), torch.arange(3), ids.int(), labels.int(), None)
    out, lp_truth = bench.parity_sample(cfg, w, Eng(), job, 16)
    assert out["pairs"] == 3 and set(out["gains"]) == {"1", "4"}
    assert torch.allclose(lp_truth, truth["label_logprobs"], atol=1e-6)
    g1, g4 = out["gains"]["1"], out["gains"]["4"]
    assert 0 < g1["max"] <= 2.5e-3 and g1["mean"] <= g1["max"] and len(g1["per_pair"]) == 3
    assert g4["max"] > g1["max"]                                             # the same logit error weighs more under a peaked head
    lr = Oracle.label_logprobs(truth["logits"] * 4.0, labels)
    assert abs(g4["logp_yes_range"][0] - float(lr[:, 0].min())) < 1e-3
    assert out["bound"] == bench.DLOGP_BOUND == 1e-3 and g1["pairs_over_bound"] == sum(x > 1e-3 for x in g1["per_pair"])
    assert out["encoder_len_range"][0] <= out["encoder_len_range"][1] and g1["yes_token_max"] <= g1["max"]
    # jobs of different lengths: the shortest and the longest are tabulated, half of the pairs each, and merged
    eng = Eng()
    short = (job[0], job[1], ids.int()[:, : ids.shape[1]], job[3], None)
    longer = (job[0], job[1], torch.cat([ids.int(), torch.zeros(3, 2, dtype=torch.int32)], 1), job[3], None)
    table, _ = bench.parity_jobs(cfg, w, eng, [short, longer, longer], 4)
    assert table["pairs"] == 4 and len(table["gains"]["1"]["per_job"]) == 2 and len(table["gains"]["1"]["per_pair"]) == 4
    one, _ = bench.parity_jobs(cfg, w, eng, [short, short], 2)
    assert one["pairs"] == 2 and "per_job" not in one["gains"]["1"]
