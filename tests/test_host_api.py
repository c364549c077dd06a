"""Host-side logic of the drop-in boundary, on CPU: prompt recipe, tokenisation with the image sentinel,
preprocessing against HF, the M x N / dataset loops, batching and image de-duplication, error conventions.
The device engine is replaced by test doubles here (a recording fake, and the CPU oracle as a checker)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import t2v_metrics_amd as t2v
from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.constants import IMAGE_TOKEN_INDEX, SYSTEM_MSG
from t2v_metrics_amd.models.vqascore_models import clip_t5_model as cm
from t2v_metrics_amd.models.vqascore_models.mm_utils import expand2square, t5_tokenizer_image_token
from t2v_metrics_amd.preprocess import OPENAI_CLIP_MEAN, clip_preprocess


class FakeTokenizer:
    """HF protocol: tokenizer(text).input_ids; whitespace words -> stable ids in [3, vocab), trailing </s> = 1."""

    def __init__(self, vocab=512):
        self.vocab = vocab

    def __call__(self, text):
        import zlib

        class R:
            pass

        r = R()
        r.input_ids = [3 + zlib.crc32(w.encode()) % (self.vocab - 3) for w in text.split()] + [1]
        return r

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(f"<{i}>" for i in ids if not (skip_special_tokens and i in (0, 1)))


class RecordingEngine:
    """Engine double: remembers what it was asked to do and returns a deterministic function of its inputs."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.encode_calls, self.score_calls = [], []

    def encode_images(self, pixels):
        self.encode_calls.append(tuple(pixels.shape))
        # feature "= mean pixel" so that scores depend on the image
        return pixels.float().mean(dim=(1, 2, 3)).reshape(-1, 1, 1).expand(-1, self.cfg.vision.n_patches, self.cfg.t5.d_model)

    def score(self, feats, img_index, input_ids, labels):
        self.score_calls.append((tuple(input_ids.shape), tuple(labels.shape), img_index.tolist()))
        img_term = feats[img_index.long(), 0, 0].float()
        txt_term = (input_ids.clamp(min=0).float().sum(-1) % 97) / 97.0
        sc = torch.sigmoid(img_term + txt_term)
        return torch.log(sc)[:, None].expand(-1, labels.shape[1]).contiguous(), sc


class OracleEngine:
    """The CPU oracle behind the engine interface (checker only; lives in tests/)."""

    def __init__(self, cfg, weights):
        from oracle.clip_t5_oracle import Oracle
        self.o = Oracle(cfg, weights)

    def encode_images(self, pixels):
        with torch.no_grad():
            return self.o.projector(self.o.vision_features(pixels.float()))

    def score(self, feats, img_index, input_ids, labels):
        o = self.o
        with torch.no_grad():
            emb, mask, _ = o.splice(feats, img_index, input_ids.long())
            enc = o.t5_encoder(emb, mask)
            from oracle.clip_t5_oracle import shift_right
            dec = o.t5_decoder(shift_right(labels.long()), enc, mask)
            lp = o.label_logprobs(o.lm_logits(dec), labels.long())
            return lp, o.scores_from_logprobs(lp, labels.long())

    def generate(self, feats, img_index, input_ids, max_new_tokens):
        o = self.o
        with torch.no_grad():
            emb, mask, _ = o.splice(feats, img_index, input_ids.long())
            enc = o.t5_encoder(emb, mask)
            dec_ids = torch.zeros(input_ids.shape[0], 1, dtype=torch.long)
            for _ in range(max_new_tokens):
                nxt = o.lm_logits(o.t5_decoder(dec_ids, enc, mask))[:, -1].argmax(-1, keepdim=True)
                dec_ids = torch.cat([dec_ids, nxt], 1)
            return dec_ids[:, 1:].to(torch.int32)


@pytest.fixture()
def images(tmp_path):
    rng = np.random.RandomState(0)
    paths = []
    for i, (h, w) in enumerate([(60, 60), (40, 70), (80, 50)]):
        p = tmp_path / f"im{i}.png"
        Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))
    arr = rng.randint(0, 256, (30, 30, 3), dtype=np.uint8)
    np.save(tmp_path / "bgr.npy", arr)
    paths.append(str(tmp_path / "bgr.npy"))
    return paths


def make_scorer(tmp_path, engine=None, **kw):
    cfg = get_config("tiny")
    engine = engine or RecordingEngine(cfg)
    s = t2v.VQAScore(model="clip-flant5-xl", device="cpu", cache_dir=str(tmp_path / "cache"), config=cfg, engine=engine,
                     tokenizer=FakeTokenizer(cfg.t5.vocab), **kw)
    return s, engine


# ----------------------------------------------------------------------------------------- prompt recipe
def test_registry_and_error_conventions(tmp_path):
    assert t2v.list_all_models() == ["clip-flant5-xxl", "clip-flant5-xl", "qwen2.5-vl-7b"]
    with pytest.raises(AssertionError):                       # score.py:28
        t2v.VQAScore(model="no-such-model", device="cpu", cache_dir=str(tmp_path))
    with pytest.raises(NotImplementedError):                  # __init__.py:30-33
        t2v.get_score_model(model="no-such-model")
    s, _ = make_scorer(tmp_path)
    with pytest.raises(AssertionError):                       # forward asserts len(images) == len(texts)
        s.model.forward(["a.png"], ["x", "y"])
    with pytest.raises(NotImplementedError):
        s(images=["clip.mp4"], texts=["x"])
    assert s.model.video_mode == "concat" and s.model.allows_image


def test_question_format_and_templates():
    q = cm.format_question(cm.default_question_template.format("a dog"))
    assert q == SYSTEM_MSG + ' USER: <image>\nDoes this figure show "a dog"? Please answer yes or no. ASSISTANT: '
    assert cm.default_answer_template == "Yes"


def test_t5_tokenizer_image_token_places_one_sentinel_between_chunks():
    tok = FakeTokenizer()
    ids = t5_tokenizer_image_token("hello world <image> bye", tok)
    a, b = tok("hello world ").input_ids, tok(" bye").input_ids
    assert ids == a + [IMAGE_TOKEN_INDEX] + b          # each chunk keeps its own trailing </s>
    assert ids.count(IMAGE_TOKEN_INDEX) == 1 and a[-1] == 1 and b[-1] == 1
    assert t5_tokenizer_image_token("no image here", tok) == tok("no image here").input_ids
    assert torch.equal(t5_tokenizer_image_token("x <image>", tok, return_tensors="pt"),
                       torch.tensor(tok("x ").input_ids + [IMAGE_TOKEN_INDEX] + tok("").input_ids))


def test_tokenize_pads_and_validates(tmp_path):
    s, _ = make_scorer(tmp_path)
    ids, lab = s.model.tokenize(["one two three <image>".replace(" <image>", ""), "one"], ["Yes", "Yes it is"])
    assert ids.dtype == torch.int32 and lab.dtype == torch.int32
    assert (ids == IMAGE_TOKEN_INDEX).sum(-1).tolist() == [1, 1]
    assert ids[1, -1] == 0 and ids[0, -1] == 1          # right padded with the pad id
    assert lab.tolist()[0][-1] == -100 and lab.tolist()[1][-1] == 1


# ----------------------------------------------------------------------------------------- images
def test_expand2square():
    im = Image.new("RGB", (10, 4), (1, 2, 3))
    sq = expand2square(im, (9, 9, 9))
    assert sq.size == (10, 10)
    a = np.asarray(sq)
    assert (a[3:7] == (1, 2, 3)).all() and (a[:3] == 9).all() and (a[7:] == 9).all()
    tall = expand2square(Image.new("RGB", (4, 9), (1, 2, 3)), (9, 9, 9))
    assert tall.size == (9, 9) and (np.asarray(tall)[:, 2:6] == (1, 2, 3)).all()
    assert expand2square(Image.new("RGB", (5, 5)), (0, 0, 0)).size == (5, 5)


def test_clip_preprocess_matches_hf_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "clip_preprocess.npz"))
    size = int(g["size"])
    i = 0
    while f"img{i}" in g:
        out = clip_preprocess(Image.fromarray(g[f"img{i}"]), size, pad_to_square=False)
        assert out.shape == (3, size, size)
        assert np.abs(out - g[f"px{i}"]).max() < 2e-6, i
        i += 1
    assert i == 5


def test_load_images_pads_with_clip_mean_and_reads_bgr_npy(tmp_path, images):
    s, _ = make_scorer(tmp_path)
    px = s.model.load_images(images)
    S = s.model.cfg.vision.image
    assert px.shape == (4, 3, S, S) and px.dtype == torch.bfloat16
    # image 1 is 40x70 (h x w): padded rows are the CLIP mean colour -> ~0 after normalisation
    top = px[1, :, 0, :].float().abs().max().item()
    assert top < 0.02, top
    arr = np.load(images[3])
    ref = clip_preprocess(Image.fromarray(arr[:, :, ::-1].copy()), S)
    assert np.abs(px[3].float().numpy() - ref).max() < 0.02          # bf16 rounding only
    assert tuple(int(x * 255) for x in OPENAI_CLIP_MEAN) == (122, 116, 104)


# ----------------------------------------------------------------------------------------- the loops
def test_forward_grid_dedups_images_and_batches_pairs(tmp_path, images):
    s, eng = make_scorer(tmp_path, max_pairs=4)
    texts = ["a cat", "a dog on a mat", "two birds"]
    out = s(images=images[:3] + [images[0]], texts=texts)
    assert out.shape == (4, 3) and out.dtype == torch.float32
    assert eng.encode_calls == [(3, 3, 56, 56)]                 # 4 image slots, 3 distinct files, encoded once
    assert [c[0][0] for c in eng.score_calls] == [4, 4, 4]      # 12 pairs in chunks of max_pairs
    assert torch.equal(out[0], out[3])                          # same file -> same row
    # pairs are scored in order of prompt length (stable; length bucketing), results come back row-major: row i = image i
    assert eng.score_calls[0][2] == [0, 0, 1, 1]                # the two 2-word prompts of images 0 and 1 first
    assert [c[0][1] for c in eng.score_calls] == sorted(c[0][1] for c in eng.score_calls)     # batch width grows
    assert eng.score_calls[0][0][1] < eng.score_calls[-1][0][1]   # short prompts are not padded to the longest one
    single = s(images=images[1], texts=texts[2])
    assert single.shape == (1, 1) and torch.allclose(single[0, 0], out[1, 2])


def test_scores_do_not_depend_on_arrival_order_or_batching(tmp_path, images):
    """Length bucketing re-orders the pairs internally; every pair's score must come back in its own slot whatever the
    order of the inputs, the batch size or the image chunking."""
    texts = ["a", "b c d e f g", "h i", "j k l m", "n o p q r s t u", "v w x"]
    s1, _ = make_scorer(tmp_path, max_pairs=256)
    ref = s1(images=images[:3], texts=texts)
    for mp, mi in ((4, 256), (5, 2), (1, 1)):
        s2, eng = make_scorer(tmp_path, max_pairs=mp)
        s2.model.max_images = mi
        assert torch.allclose(s2(images=images[:3], texts=texts), ref)
        perm = [4, 0, 5, 2, 1, 3]
        out = s2(images=[images[2], images[0], images[1]], texts=[texts[k] for k in perm])
        assert torch.allclose(out, ref[[2, 0, 1]][:, perm])


def test_forward_kwargs_reach_the_model(tmp_path, images):
    s, eng = make_scorer(tmp_path)
    a = s(images=images[:1], texts=["x y"])
    b = s(images=images[:1], texts=["x y"], question_template="Is this {}?", answer_template="{}")
    assert not torch.equal(a, b)
    assert eng.score_calls[-1][1][1] == 3                       # answer "{}" -> "x y" -> 2 words + </s>


def test_model_forward_matches_reference_row_semantics(tmp_path, images):
    """score.py:105-106: scores[i] = model.forward([image]*N, texts).  Row i of the grid equals that call."""
    s, _ = make_scorer(tmp_path)
    texts = ["alpha", "beta gamma"]
    grid = s(images=images[:2], texts=texts)
    for i in range(2):
        row = s.model.forward([images[i]] * len(texts), texts)
        assert isinstance(row, torch.Tensor) and row.device.type == "cpu"
        assert torch.allclose(row, grid[i])


def test_batch_forward_shape_and_values(tmp_path, images):
    s, eng = make_scorer(tmp_path)
    dataset = [{"images": [images[0], images[1]], "texts": ["t one", "t two", "t three"]},
               {"images": [images[2], images[0]], "texts": ["u one", "u two", "u three"]},
               {"images": [images[1], images[1]], "texts": ["v one", "v two", "v three"]}]
    out = s.batch_forward(dataset, batch_size=2)
    assert out.shape == (3, 2, 3)
    for k, d in enumerate(dataset):
        assert torch.allclose(out[k], s(images=d["images"], texts=d["texts"]))
    assert torch.equal(out[2, 0], out[2, 1])
    with pytest.raises(AssertionError):
        s.batch_forward([dataset[0], {"images": [images[0]], "texts": ["a", "b", "c"]}])


def test_full_host_pipeline_against_oracle(tmp_path, images):
    """Prompt -> ids -> splice -> score through the real host code with the CPU oracle as the engine equals
    calling the oracle by hand on hand-built inputs."""
    from oracle.clip_t5_oracle import Oracle
    from t2v_metrics_amd.weights import make_seeded_weights
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=3, device="cpu")
    s, _ = make_scorer(tmp_path, engine=OracleEngine(cfg, w))
    texts = ["red car", "a blue bicycle leaning on a wall"]
    out = s(images=images[:2], texts=texts)
    tok = FakeTokenizer(cfg.t5.vocab)
    px = s.model.load_images(images[:2]).float()
    for i in range(2):
        for j, t in enumerate(texts):
            ids = t5_tokenizer_image_token(cm.format_question(cm.default_question_template.format(t)), tok)
            lab = tok("Yes").input_ids
            ref = Oracle(cfg, w).forward(px[i: i + 1], torch.tensor([0]), torch.tensor([ids]), torch.tensor([lab]))
            assert abs(ref["scores"][0].item() - out[i, j].item()) < 1e-5 * max(1.0, out[i, j].item())
    assert ((out >= 0) & (out <= 1)).all()                     # the reference's own smoke assertion (test.py:110-112)


def test_threaded_image_pipeline_matches_serial(tmp_path, images):
    """Thread-pool decode/preprocess and chunk prefetch give the same tensors and scores as the serial path."""
    s1, e1 = make_scorer(tmp_path, num_workers=1, max_images=2)
    s8, e8 = make_scorer(tmp_path, num_workers=8, max_images=2)
    a = s1.model.load_images(images)
    b = s8.model.load_images(images)
    assert torch.equal(a, b)
    texts = ["x", "y z"]
    assert torch.equal(s1(images=images, texts=texts), s8(images=images, texts=texts))
    assert e8.encode_calls == [(2, 3, 56, 56), (2, 3, 56, 56)] == e1.encode_calls      # 4 images in chunks of max_images


def test_generate_api(tmp_path, images):
    """model.generate(images, texts) (/root/reference/V_3.0_README.md:316-325): one decoded string per pair, cut at EOS,
    same answer for the same (image, prompt) wherever it sits in the batch."""
    from t2v_metrics_amd.weights import make_seeded_weights
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=9, device="cpu", dtype=torch.bfloat16, lm_head_gain=8.0)
    s, _ = make_scorer(tmp_path, engine=OracleEngine(cfg, {k: v.float() for k, v in w.items()}))
    texts = ["Please describe this image:", "Is there a dog ?", "Please describe this image:"]
    out = s.model.generate(images=[images[0], images[1], images[0]], texts=texts, max_new_tokens=4)
    assert isinstance(out, list) and len(out) == 3 and all(isinstance(x, str) for x in out)
    assert out[0] == out[2]
    ids = s.model.generate_ids([images[0]], [texts[0]], max_new_tokens=4)
    assert 1 <= len(ids[0]) <= 4 and (1 not in ids[0][:-1])
    with pytest.raises(ValueError):
        s.model.generate(images=[images[0]], texts=[texts[0]], max_new_tokens=513)
    with pytest.raises(AssertionError):
        s.model.generate(images=images[:2], texts=texts[:1])


def test_empty_inputs_follow_the_reference(tmp_path, images):
    """score.py:104-106: an empty image or text list gives an empty [M, N] tensor, not an error."""
    s, eng = make_scorer(tmp_path)
    assert tuple(s(images=[], texts=[]).shape) == (0, 0)
    assert tuple(s(images=images[:2], texts=[]).shape) == (2, 0)
    assert tuple(s(images=[], texts=["a", "b"]).shape) == (0, 2)
    assert tuple(s.model.forward([], []).shape) == (0,)
    assert eng.score_calls == []



def test_rank_affinity_plan_follows_the_gpus_numa_node():
    """bench.py / sharding.set_rank_affinity: a rank's host cores are an equal slice of the NUMA node its GPU hangs off (VERDICT r3
    item 8); unknown topology = an equal split of the allowed cores; a restricted affinity mask is honoured."""
    from t2v_metrics_amd.sharding import _parse_cpulist, plan_rank_affinity
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    node_cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}
    numa = [0, 0, 0, 0, 1, 1, 1, 1]
    got = [plan_rank_affinity(numa, node_cpus, set(range(128)), r) for r in range(8)]
    assert [len(g) for g in got] == [16] * 8 and sorted(c for g in got for c in g) == list(range(128))
    assert all(set(got[r]) <= set(node_cpus[numa[r]]) for r in range(8))
    assert plan_rank_affinity([1, 0], node_cpus, set(range(128)), 0) == list(range(64, 128))      # GPU 0 on node 1
    assert plan_rank_affinity([-1, -1], {}, set(range(8)), 1) == [4, 5, 6, 7]                     # unknown topology
    assert plan_rank_affinity(numa, node_cpus, set(range(0, 128, 2)), 5) == list(range(80, 96, 2)) # restricted mask: 32 allowed cores on node 1, 4 ranks
    assert plan_rank_affinity([0], node_cpus, {3, 4}, 0) == [3, 4]                                # one rank: untouched


def test_weights_that_do_not_fit_fp16_are_named_before_the_fp16_tower_reads_them():
    """engine.fp16_unsafe_weights: the tensors vqs_bind_weights copies to fp16 for option vit_fp16 (the tower's and the projector's linear
    weights) must be finite and below 65 520 in magnitude; VqsEngine.bind raises with their names otherwise."""
    from t2v_metrics_amd.engine import FP16_MAX_FINITE_ROUNDED, fp16_unsafe_weights
    from t2v_metrics_amd.weights import make_seeded_weights
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=3, device="cpu")
    assert fp16_unsafe_weights(w) == []
    assert torch.tensor(FP16_MAX_FINITE_ROUNDED).half().isinf() and torch.isfinite(torch.tensor(65504.0).half())
    k1, k2 = "vision.encoder.layers.1.mlp.fc1.weight", "mm_projector.2.weight"
    w[k1] = w[k1].clone(); w[k1][0, 0] = 7.0e4
    w[k2] = w[k2].clone(); w[k2][1, 1] = float("nan")
    t5 = "encoder.block.0.layer.1.DenseReluDense.wo.weight"
    w[t5] = w[t5].clone(); w[t5][0, 0] = 1.0e6                                   # the T5 stacks stay bf16: not this check's business
    w["vision.encoder.layers.0.layer_norm1.weight"] = w["vision.encoder.layers.0.layer_norm1.weight"] * 1e5   # norm parameters stay bf16 too
    assert sorted(fp16_unsafe_weights(w)) == sorted([k1, k2])
    # the engine's bind-time check and the option guard, on an engine object without a device (the handle's options live in a dict)
    import ctypes
    import types
    import warnings
    from t2v_metrics_amd.engine import FP16_OPTIONS, VqsEngine, VqsError

    def fake_engine(weights, explicit=()):
        e = object.__new__(VqsEngine)
        e.cfg, e.weights, e._h, e._options, e._explicit, e.fp16_auto_off = cfg, weights, None, {}, set(explicit), {}
        state = {k: 1 for k in FP16_OPTIONS}
        e.get_option = lambda name: state.get(name, 0)
        e.lib = types.SimpleNamespace(vqs_set_option=lambda h, n, v: (state.__setitem__(n.decode(), int(v)), 0)[1],
                                      vqs_last_error=lambda h: b"")
        return e, state

    # (a) the caller asked for the fp16 tower: refused, naming the tensors
    e, _ = fake_engine(w, explicit={"vit_fp16"})
    with pytest.raises(VqsError, match="vit_fp16 = 1 refused.*exceed the fp16 range"):
        e._check_fp16_weights()
    # (b) the option is on by default: switched off with one warning, the reference's bf16 runs -- no exception for a drop-in user
    e, state = fake_engine(w)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        e._check_fp16_weights()
    assert state["vit_fp16"] == 0 and "vit_fp16" in e.fp16_auto_off and any("vit_fp16" in str(r.message) for r in rec)
    with pytest.raises(VqsError, match="vit_fp16 = 1 refused"):
        e.set_option("vit_fp16", 1)
    # (c) clean weights: nothing switched off, the range proof holds for every option and the projector needs no scale at this size
    e, state = fake_engine(make_seeded_weights(cfg, seed=3, device="cpu"))
    e._check_fp16_weights()
    assert e._fp16_unsafe == [] and e.fp16_auto_off == {} and all(state[k] == 1 for k in FP16_OPTIONS)
    assert all(e.range_proof[k]["holds"] for k in FP16_OPTIONS) and state.get("proj_fs_shift") == 0 and state.get("proj_mid_shift") == 0


def test_fp16_range_proof_bounds_hold_and_switch_defaults_off():
    """engine.fp16_range_proof: every bound follows from the weights alone and is >= what a pass produces at that site;
    a heavy-tailed checkpoint (norm weights x 10^3 on the encoder stream) loses enc_fp16 / dec_fp16 BY DEFAULT with a warning and keeps them
    when the caller insists; the projector's stream-fed sites get the shifts their bounds ask for."""
    import types
    import warnings
    from oracle.clip_t5_engine_rounding import EngineRoundedOracle
    from t2v_metrics_amd.engine import FP16_HEAD, FP16_OPTIONS, VqsEngine, fp16_range_proof, fp16_shift
    from t2v_metrics_amd.weights import make_seeded_weights
    cfg = get_config("small")
    w = make_seeded_weights(cfg, seed=1, device="cpu")
    proof = fp16_range_proof(cfg, w)
    assert all(proof[k]["holds"] for k in FP16_OPTIONS)
    # observed maxima of an unrounded pass against the proof's worst bounds, per option
    g = torch.Generator().manual_seed(2)
    pix = torch.randn(2, 3, cfg.vision.image, cfg.vision.image, generator=g).to(torch.bfloat16)
    ids = torch.tensor([[11, 12, -200, 13, 14, 1], [21, -200, 22, 1, 0, 0]])
    o = EngineRoundedOracle(cfg, w, round_fn=lambda x: x.float())
    o.record = {}
    o.forward(pix.float(), torch.tensor([0, 1]), ids, torch.tensor([[40, 1], [41, 1]]))
    seen = {"vit_fp16": 0.0, "proj_fp16": 0.0, "enc_fp16": 0.0, "dec_fp16": 0.0}
    for name, x in o.record.items():
        parts = name.split(".")
        m = float(x.abs().max())
        if parts[0] == "vit" and parts[-1] in ("feat_in", "pmid"):
            seen["proj_fp16"] = max(seen["proj_fp16"], m)
        elif parts[0] == "vit" and parts[-1] in ("xn0", "q", "k", "v", "attn", "d_attn", "xn1", "mid", "d_mlp"):
            seen["vit_fp16"] = max(seen["vit_fp16"], m)
        elif parts[0] == "enc" and parts[-1] in ("xn0", "q", "k", "v", "attn", "xn1"):
            seen["enc_fp16"] = max(seen["enc_fp16"], m)
        elif (parts[0] == "enc" and parts[-1] == "out") or (parts[0] == "dec" and parts[-1] in ("cq", "cqk")):
            seen["dec_fp16"] = max(seen["dec_fp16"], m)
    for k in FP16_OPTIONS:
        assert 0.0 < seen[k] <= proof[k]["worst_bound"] * (1 + 1e-5), (k, seen[k], proof[k])
    assert fp16_shift(FP16_HEAD) == 0 and fp16_shift(FP16_HEAD * 2 + 1) == 2 and fp16_shift(float("nan")) is None and fp16_shift(float("inf")) is None
    # heavy tail on the encoder stream
    w2 = dict(w)
    for k in ("encoder.block.1.layer.0.layer_norm.weight", "encoder.final_layer_norm.weight"):
        w2[k] = w[k].clone()
        w2[k][:3] *= 3.0e3
    proof2 = fp16_range_proof(cfg, w2)
    assert not proof2["enc_fp16"]["holds"] and not proof2["dec_fp16"]["holds"] and proof2["vit_fp16"]["holds"]

    def fake_engine(explicit=()):
        e = object.__new__(VqsEngine)
        e.cfg, e.weights, e._h, e._options, e._explicit, e.fp16_auto_off = cfg, w2, None, {}, set(explicit), {}
        state = {k: 1 for k in FP16_OPTIONS}
        e.get_option = lambda name: state.get(name, 0)
        e.lib = types.SimpleNamespace(vqs_set_option=lambda h, n, v: (state.__setitem__(n.decode(), int(v)), 0)[1], vqs_last_error=lambda h: b"")
        return e, state

    e, state = fake_engine()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        e._check_fp16_weights()
    assert state["enc_fp16"] == 0 and state["dec_fp16"] == 0 and state["vit_fp16"] == 1 and len(rec) == 2
    assert "no proof" in e.fp16_auto_off["enc_fp16"]
    e, state = fake_engine(explicit={"enc_fp16", "dec_fp16"})
    e._check_fp16_weights()
    assert state["enc_fp16"] == 1 and state["dec_fp16"] == 1 and e.fp16_auto_off == {}


def test_gpu_numa_lookup_formats_the_pci_address_from_torchs_integer_fields(tmp_path, monkeypatch):
    """torch's `pci_bus_id` is an int, not a BDF string -- the sysfs path must be built from domain / bus / device."""
    import types
    from t2v_metrics_amd import sharding
    props = [types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0x05, pci_device_id=0), types.SimpleNamespace(pci_domain_id=1, pci_bus_id=0xc5, pci_device_id=0),
             types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0x99, pci_device_id=0)]
    assert [sharding.pci_bdf(p) for p in props] == ["0000:05:00.0", "0001:c5:00.0", "0000:99:00.0"]
    for bdf, node in (("0000:05:00.0", 0), ("0001:c5:00.0", 1)):
        (tmp_path / bdf).mkdir()
        (tmp_path / bdf / "numa_node").write_text("%d\n" % node)
    monkeypatch.setattr(sharding.torch.cuda, "get_device_properties", lambda i: props[i])
    assert sharding.gpu_numa_nodes(3, sysfs=str(tmp_path)) == [0, 1, -1]        # the third device has no sysfs entry: unknown


def test_staging_event_is_recorded_on_the_models_device_and_stream(tmp_path, monkeypatch):
    """load_images runs on the image thread, whose current device is 0 whatever the main thread set.  The H2D copy of a
    reused pinned staging buffer and the event that guards the buffer's next use must sit on the MODEL's device and its current stream --
    an event recorded on device 0 while the copy runs on cuda:1 is complete at once and the buffer is overwritten under the DMA.  No second
    GPU here: torch.cuda's device / stream / event entry points are replaced by recorders and the call is made from a worker thread."""
    import threading
    from PIL import Image
    import t2v_metrics_amd as t2v
    from t2v_metrics_amd.config import get_config
    cfg = get_config("tiny")
    rng = np.random.RandomState(4)
    paths = []
    for i in range(3):
        p = tmp_path / f"s{i}.png"
        Image.fromarray(rng.randint(0, 256, (40, 40, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))

    class Eng(RecordingEngine):
        def normalize_u8(self, u8, mean, std):
            log.append(("normalize", getattr(tls, "device", None)))
            return u8.permute(0, 3, 1, 2).float()

    log, tls = [], threading.local()
    m = t2v.VQAScore(model="clip-flant5-xl", device="cpu", config=cfg, engine=Eng(cfg), tokenizer=FakeTokenizer(cfg.t5.vocab), image_workers="thread").model
    m.device = "cuda:1"                                        # a rank whose GPU is not device 0

    class FakeDeviceCtx:
        def __init__(self, dev):
            self.dev = str(dev)

        def __enter__(self):
            self.prev = getattr(tls, "device", "cuda:0")       # a fresh thread's current device is 0
            tls.device = self.dev

        def __exit__(self, *a):
            tls.device = self.prev

    class FakeEvent:
        def record(self, stream=None):
            log.append(("record", getattr(tls, "device", "cuda:0"), stream))

        def synchronize(self):
            log.append(("sync",))

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device", FakeDeviceCtx)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: ("stream-of", str(dev) if dev is not None else getattr(tls, "device", "cuda:0")))
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)
    real_to = torch.Tensor.to

    def fake_to(self, *a, **k):
        if a and str(a[0]).startswith("cuda"):
            log.append(("h2d", str(a[0]), getattr(tls, "device", "cuda:0")))
            return self
        return real_to(self, *a, **k)
    monkeypatch.setattr(torch.Tensor, "to", fake_to)
    th = threading.Thread(target=lambda: m.load_images(paths))
    th.start(); th.join()
    kinds = [e[0] for e in log]
    assert kinds == ["h2d", "record", "normalize"], log
    assert log[0][1:] == ("cuda:1", "cuda:1")                  # the copy is issued with cuda:1 current ...
    assert log[1][1] == "cuda:1" and log[1][2] == ("stream-of", "cuda:1")      # ... and the event is recorded on cuda:1's current stream
