"""Pin the CPU oracle against outputs of the HF modules the reference executes
(fixtures from oracle/make_golden.py)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle.clip_t5_oracle import Oracle, relative_position_bucket, shift_right
from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import make_seeded_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_relpos_bucket_numpy_matches_hf(golden_dir):
    g = np.load(os.path.join(golden_dir, "relpos_buckets.npz"))
    rp = g["relative_position"]
    assert np.array_equal(relative_position_bucket(rp, True), g["bidirectional"])
    assert np.array_equal(relative_position_bucket(rp, False), g["causal"])


def test_relpos_bucket_c_matches_hf(golden_dir):
    so = os.path.join(ROOT, "oracle", "_build", "librelpos_oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    g = np.load(os.path.join(golden_dir, "relpos_buckets.npz"))
    rp = np.ascontiguousarray(g["relative_position"].astype(np.int32))
    out = np.zeros_like(rp)
    for bidir, key in ((1, "bidirectional"), (0, "causal")):
        lib.t5_relpos_bucket_many(rp.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(rp.size), ctypes.c_int32(bidir),
                                  ctypes.c_int32(32), ctypes.c_int32(128), out.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(out, g[key]), key


def test_shift_right():
    lab = torch.tensor([[2163, 1], [5, -100]])
    assert shift_right(lab).tolist() == [[0, 2163], [0, 5]]
    lab3 = torch.tensor([[7, 8, -100]])
    assert shift_right(lab3).tolist() == [[0, 7, 8]]


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_oracle_matches_hf_modules(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"hf_{name}.npz"))
    cfg = get_config(name)
    w = make_seeded_weights(cfg, seed=int(g["seed"]), device="cpu", dtype=torch.bfloat16)
    o = Oracle(cfg, w)
    with torch.no_grad():
        # vision tower: hidden_states[-2]
        h = o.vision_embeddings(torch.from_numpy(g["pixels"]))
        for i in range(cfg.vision.layers_run):
            h = o.vision_layer(h, i)
        ref = torch.from_numpy(g["vit_hidden_m2"])
        assert h.shape == ref.shape
        assert (h - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
        assert torch.equal(o.vision_features(torch.from_numpy(g["pixels"])), h[:, 1:])

        # T5 encoder / decoder / head / score
        emb = torch.from_numpy(g["emb"])
        mask = torch.from_numpy(g["mask"])
        labels = torch.from_numpy(g["labels"])
        enc = o.t5_encoder(emb, mask)
        enc_ref = torch.from_numpy(g["enc_out"])
        # padded query rows are don't-care
        assert ((enc - enc_ref).abs() * mask[..., None]).max().item() < 2e-4
        dec = o.t5_decoder(shift_right(labels), enc, mask)
        logits = o.lm_logits(dec)
        ref_logits = torch.from_numpy(g["logits"])
        valid = (labels != -100)
        assert ((logits - ref_logits).abs() * valid[..., None]).max().item() < 5e-4
        lp = o.label_logprobs(logits, labels)
        scores = o.scores_from_logprobs(lp, labels)
        assert np.allclose(scores.numpy(), g["scores"], rtol=1e-3, atol=1e-7)


def test_splice_layout():
    """Hand-built case for the (non-HF) splice glue: SURVEY.md §8a row a12."""
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=1, device="cpu")
    o = Oracle(cfg, w)
    P, D = cfg.vision.n_patches, cfg.t5.d_model
    proj = torch.arange(2 * P * D, dtype=torch.float32).reshape(2, P, D)
    ids = torch.tensor([[5, 6, -200, 7, 1], [9, -200, 1, 0, 0]])
    emb, mask, lens = o.splice(proj, torch.tensor([1, 0]), ids)
    S_e = ids.shape[1] - 1 + P
    assert emb.shape == (2, S_e, D) and lens.tolist() == [4 + P, 2 + P]
    shared = w["shared.weight"].float()
    assert torch.equal(emb[0, 0], shared[5]) and torch.equal(emb[0, 1], shared[6])
    assert torch.equal(emb[0, 2:2 + P], proj[1])
    assert torch.equal(emb[0, 2 + P], shared[7]) and torch.equal(emb[0, 3 + P], shared[1])
    assert torch.equal(emb[1, 0], shared[9]) and torch.equal(emb[1, 1:1 + P], proj[0])
    assert torch.equal(emb[1, 1 + P], shared[1])
    assert mask[0].all() and mask[1, :2 + P].all() and not mask[1, 2 + P:].any()
    assert emb[1, 2 + P:].abs().sum() == 0


def test_oracle_end_to_end_shapes_and_range():
    """The reference's own smoke assertions (test.py:106-117,134-144): shape and [0,1] range."""
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=2, device="cpu")
    o = Oracle(cfg, w)
    g = torch.Generator().manual_seed(0)
    pix = torch.randn(2, 3, cfg.vision.image, cfg.vision.image, generator=g)
    ids = torch.tensor([[11, 12, -200, 13, 1], [21, -200, 22, 23, 1], [31, -200, 1, 0, 0]])
    labels = torch.tensor([[40, 1], [40, 1], [41, 1]])
    out = o.forward(pix, torch.tensor([0, 1, 1]), ids, labels)
    assert out["scores"].shape == (3,) and out["label_logprobs"].shape == (3, 2)
    assert ((out["scores"] >= 0) & (out["scores"] <= 1)).all()


@pytest.mark.parametrize("name", ["generate_tiny_g8", "generate_small_g8"])
def test_oracle_greedy_generate_matches_hf(name, golden_dir):
    """Greedy search of HF T5ForConditionalGeneration.generate (fixture from oracle/make_golden.py::golden_generate):
    the oracle's token ids are identical, step for step."""
    import numpy as np
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.config import get_config
    from t2v_metrics_amd.weights import make_seeded_weights
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = get_config(name.split("_")[1])
    w = make_seeded_weights(cfg, seed=int(z["seed"]), device="cpu", dtype=torch.bfloat16, lm_head_gain=float(z["gain"]))
    o = Oracle(cfg, {k: v.float() for k, v in w.items()})
    toks, margins = o.generate(torch.from_numpy(z["pixels"]), torch.from_numpy(z["img_index"]), torch.from_numpy(z["ids"]),
                               int(z["max_new"]), return_margins=True)
    assert torch.equal(toks, torch.from_numpy(z["tokens"]).long())
    assert torch.allclose(margins, torch.from_numpy(z["margins"]), atol=2e-3)



@pytest.mark.parametrize("fixture", ["e2e_tiny_g1", "e2e_tiny_g4", "e2e_small_g1", "e2e_small_g4"])
def test_hf_reference_builder_reproduces_the_fixtures(golden_dir, fixture):
    """oracle/hf_reference.py (the reference's arithmetic assembled from the HF modules on the meta device; what
    bench.py's cpu_baseline times and tools/config1_cpu.py drives) against the committed fixtures, which
    oracle/make_golden.py generated from the same modules: fp32 and the reference's bf16 configuration."""
    import warnings
    warnings.filterwarnings("ignore")
    from oracle.hf_reference import HFReference
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    cfg = get_config(fixture.split("_")[1])
    w = make_seeded_weights(cfg, seed=int(g["seed"]), device="cpu", lm_head_gain=float(g["gain"]))
    pix = torch.from_numpy(g["pixels"]).to(torch.bfloat16)
    idx, ids, labels = (torch.from_numpy(g[k]) for k in ("img_index", "ids", "labels"))
    r32 = HFReference(cfg, w, torch.float32).forward(pix, idx, ids, labels)
    r16 = HFReference(cfg, w, torch.bfloat16).forward(pix, idx, ids, labels)
    assert (r32["label_logprobs"] - torch.from_numpy(g["logprobs_fp32"])).abs().max().item() <= 1e-5
    assert (r32["scores"] - torch.from_numpy(g["scores_fp32"])).abs().max().item() <= 1e-6
    assert (r16["label_logprobs"] - torch.from_numpy(g["logprobs_hf_bf16"])).abs().max().item() <= 1e-5
