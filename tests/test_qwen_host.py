"""Host side of the Qwen2.5-VL row (t2v_metrics_amd/models/vqascore_models/qwen25vl_model.py): preprocessing pinned to the
HF image processor, prompt/placeholder handling, batching by grid, scoring recipe -- on the CPU with the oracle behind the
engine interface (checker only)."""
import zlib

import numpy as np
import pytest
import torch

import t2v_metrics_amd as t2v
from t2v_metrics_amd.models.vqascore_models.qwen25vl_model import (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD, Qwen25VLModel, chat_prompt,
                                                                   patchify, smart_resize)
from t2v_metrics_amd.qwen import get_qwen_config
from t2v_metrics_amd.qwen.weights import make_seeded_qwen_weights

SPECIALS = {"<|im_start|>": 8, "<|im_end|>": 9, "<|vision_start|>": 6, "<|vision_end|>": 7, "<|video_pad|>": 5, "<|image_pad|>": 4}


class FakeQwenTokenizer:
    """HF protocol subset: encode(text, add_special_tokens=False); chat-template specials -> fixed ids, words -> hashed ids."""

    def __init__(self, vocab=512):
        self.vocab = vocab

    def encode(self, text, add_special_tokens=False):
        for s in SPECIALS:
            text = text.replace(s, f" {s} ")
        return [SPECIALS[w] if w in SPECIALS else 10 + zlib.crc32(w.encode()) % (self.vocab - 10) for w in text.split()]

    def decode(self, ids, skip_special_tokens=False):
        special = set(SPECIALS.values())
        return " ".join(f"t{int(i)}" for i in ids if not (skip_special_tokens and int(i) in special))


class OracleQwenEngine:
    def __init__(self, cfg, weights):
        from oracle.qwen25vl_oracle import QwenOracle
        self.o, self.cfg = QwenOracle(cfg, weights), cfg
        self.vision_calls = []

    def encode_vision(self, patches, grids):
        self.vision_calls.append(list(grids))
        with torch.no_grad():
            return self.o.vision_tower(patches.float(), grids)

    def score_logits(self, merged, input_ids, attention_mask, grids):
        from oracle.qwen25vl_oracle import mrope_position_ids
        o, c = self.o, self.cfg
        with torch.no_grad():
            emb = o._w("model.language_model.embed_tokens.weight")[input_ids]
            m = (input_ids == c.video_token_id) & attention_mask.bool()
            emb = emb.masked_scatter(m[..., None].expand_as(emb), merged.float())
            pos = mrope_position_ids(input_ids, attention_mask, c.image_token_id, c.video_token_id, [], grids,
                                     c.vision.spatial_merge, c.vision.tokens_per_second)
            hid = o.text_model(emb, pos, attention_mask)
            last = attention_mask.long().sum(-1) - 1
            return hid[torch.arange(hid.shape[0]), last] @ o._w("lm_head.weight").t()

    # generation with a "cache": the double re-runs the fp32 forward over prompt + tokens so far, the generated tokens at the
    # positions HF generate uses (the prompt's last position + 1 + step on each axis)
    def prefill(self, merged, input_ids, attention_mask, grids, max_new_tokens):
        state = {"merged": merged, "ids": input_ids.clone(), "mask": attention_mask.clone(), "grids": grids, "new": [], "steps": 0}
        self.prefills = getattr(self, "prefills", 0) + 1
        return self.score_logits(merged, input_ids, attention_mask, grids), state

    def decode(self, state, token_ids):
        from oracle.qwen25vl_oracle import mrope_position_ids
        o, c = self.o, self.cfg
        state["new"].append(token_ids.clone())
        state["steps"] += 1
        ids0, mask0 = state["ids"], state["mask"]
        B, L = ids0.shape
        t = len(state["new"])
        n = mask0.long().sum(-1)
        ids = torch.zeros(B, L + t, dtype=torch.long)
        mask = torch.zeros(B, L + t, dtype=torch.long)
        pos0 = mrope_position_ids(ids0, mask0, c.image_token_id, c.video_token_id, [], state["grids"], c.vision.spatial_merge,
                                  c.vision.tokens_per_second)
        pos = torch.zeros(3, B, L + t, dtype=torch.long)
        for b in range(B):
            nb = int(n[b])
            ids[b, :nb] = ids0[b, :nb]
            ids[b, nb: nb + t] = torch.stack([x[b] for x in state["new"]])
            mask[b, : nb + t] = 1
            pos[:, b, :nb] = pos0[:, b, :nb]
            pos[:, b, nb: nb + t] = (pos0[:, b, nb - 1] + 1)[:, None] + torch.arange(t)[None]
        with torch.no_grad():
            emb = o._w("model.language_model.embed_tokens.weight")[ids]
            m = (ids == c.video_token_id) & mask.bool()
            emb = emb.masked_scatter(m[..., None].expand_as(emb), state["merged"].float())
            hid = o.text_model(emb, pos, mask)
            return hid[torch.arange(B), mask.long().sum(-1) - 1] @ o._w("lm_head.weight").t()


def test_patchify_and_smart_resize_match_hf():
    assert smart_resize(336, 448) == (336, 448) and smart_resize(360, 640, max_pixels=360 * 420) == (280, 504)
    pil = pytest.importorskip("transformers.models.qwen2_vl.image_processing_pil_qwen2_vl")
    ip = pil.Qwen2VLImageProcessorPil()
    img = np.random.RandomState(0).randint(0, 256, (112, 168, 3), dtype=np.uint8)
    out = ip(images=[img], do_resize=False, return_tensors="pt")
    x = torch.from_numpy(img).permute(2, 0, 1).float()[None] / 255.0
    x = (x - torch.tensor(OPENAI_CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(OPENAI_CLIP_STD).view(1, 3, 1, 1)
    mine, g = patchify(x, 14, 2, 2)
    assert list(g) == out["image_grid_thw"][0].tolist()
    assert (out["pixel_values"] - mine).abs().max().item() <= 1e-6
    # a two-frame video of the same picture is the same patch matrix as the still image (HF duplicates the frame)
    assert torch.equal(patchify(torch.cat([x, x]), 14, 2, 2)[0], mine)


def test_forward_recipe_batching_and_errors(tmp_path):
    cfg = get_qwen_config("qwen-tiny")
    w = make_seeded_qwen_weights(cfg, seed=3, dtype=torch.bfloat16, lm_head_gain=4.0)
    eng = OracleQwenEngine(cfg, w)
    tok = FakeQwenTokenizer(cfg.text.vocab)
    rng = np.random.RandomState(1)
    paths = []
    for i, shape in enumerate([(4, 112, 112, 3), (4, 112, 112, 3), (112, 168, 3), (3, 112, 112, 3)]):
        p = tmp_path / f"v{i}.npy"
        np.save(p, rng.randint(0, 256, shape, dtype=np.uint8))
        paths.append(str(p))
    scorer = t2v.VQAScore(model="qwen2.5-vl-7b", device="cpu", config=cfg, engine=eng, tokenizer=tok)
    texts = ["a cat jumps", "a dog runs fast", "two birds", "a cat jumps"]
    s = scorer.model.forward(paths, texts)
    assert s.shape == (4,) and bool(((s > 0) & (s < 1)).all())
    # samples 0, 1 and 3 share the grid (2, 8, 8) (3 frames are padded to 4) -> one vision call for the three
    assert sorted(len(c) for c in eng.vision_calls) == [1, 3]
    # the recipe: softmax(last-position logits)[first token of the answer] on the chat-template prompt
    from oracle.qwen25vl_oracle import QwenOracle
    item = scorer.model.load_images([paths[2]])[0]
    patches, grid = scorer.model.preprocess(item)
    ids = scorer.model.build_ids('Does this figure show "two birds"? Please answer Yes or No.', "image", grid[1] * grid[2] // 4)
    assert chat_prompt("q", "<|image_pad|>").count("<|vision_start|>") == 1 and ids.count(cfg.video_token_id) == grid[1] * grid[2] // 4
    ids_t = torch.tensor([ids])
    logits = QwenOracle(cfg, w).forward(ids_t, torch.ones_like(ids_t), patches, [grid])
    yes = tok.encode("Yes")[0]
    assert abs(torch.softmax(logits, -1)[0, yes].item() - s[2].item()) <= 1e-5 * max(1.0, s[2].item()) + 1e-8
    # temperature and answer template are honoured
    s_t = scorer.model.forward(paths[:1], texts[:1], temperature=2.0, answer_template="No")
    assert s_t.shape == (1,) and abs(s_t[0].item() - s[0].item()) > 0
    # score() through the public API
    grid_scores = scorer(images=paths[:2], texts=texts[:2])
    assert grid_scores.shape == (2, 2)
    with pytest.raises(NotImplementedError):
        scorer.model.forward(["clip.mp4"], ["x"])
    odd = tmp_path / "odd.npy"                      # 84 x 140 px: 3 x 5 merged cells, partial 2 x 2-cell windows at the edges
    np.save(odd, rng.randint(0, 256, (2, 84, 140, 3), dtype=np.uint8))
    assert scorer.model.forward([str(odd)], ["x"]).shape == (1,)
    with pytest.raises(AssertionError):
        scorer.model.forward(paths[:2], texts[:1])


def test_published_checkpoint_key_layout_is_accepted(tmp_path):
    """The safetensors shards of Qwen/Qwen2.5-VL-*-Instruct use the legacy keys (visual.*, model.layers.*,
    model.embed_tokens.weight, model.norm.weight); HF renames them on load.  Both layouts must map onto the inventory."""
    from safetensors.torch import save_file
    from t2v_metrics_amd.qwen.weights import canonical_qwen_name, load_qwen_checkpoint, qwen_weight_specs
    cfg = get_qwen_config("qwen-tiny")
    w = make_seeded_qwen_weights(cfg, seed=5, dtype=torch.bfloat16)

    def legacy(k):
        if k.startswith("model.visual."):
            return k[len("model."):]
        if k.startswith("model.language_model."):
            return "model." + k[len("model.language_model."):]
        return k

    names = [n for n, _, _ in qwen_weight_specs(cfg)]
    assert {canonical_qwen_name(legacy(n)) for n in names} == set(names) == {canonical_qwen_name(n) for n in names}
    assert canonical_qwen_name("visual.blocks.0.attn.qkv.weight") == "model.visual.blocks.0.attn.qkv.weight"
    assert canonical_qwen_name("model.layers.3.mlp.up_proj.weight") == "model.language_model.layers.3.mlp.up_proj.weight"
    assert canonical_qwen_name("model.embed_tokens.weight") == "model.language_model.embed_tokens.weight"
    assert canonical_qwen_name("model.norm.weight") == "model.language_model.norm.weight"
    assert canonical_qwen_name("lm_head.weight") == "lm_head.weight"
    items = sorted(w.items())
    half = len(items) // 2
    save_file({legacy(k): v.contiguous() for k, v in items[:half]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({legacy(k): v.contiguous() for k, v in items[half:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    got = load_qwen_checkpoint(str(tmp_path))
    assert set(got) == set(w) and all(torch.equal(got[k], w[k]) for k in w)


def test_npy_frame_lists_are_not_capped_at_the_container_file_limit(tmp_path):
    """qwen2vl_model.py:141-144 sets max_pixels = 360*420 for video FILE paths only; a 4-D .npy becomes a frame list
    without max_pixels (:151-153), so a 448 x 672 clip keeps its grid (a capped one would be 308 x 476)."""
    cfg = get_qwen_config("qwen-tiny")
    m = Qwen25VLModel(model_name="qwen2.5-vl-7b", device="cpu", config=cfg, engine=object(), tokenizer=FakeQwenTokenizer(cfg.text.vocab))
    p = tmp_path / "big.npy"
    np.save(p, np.zeros((2, 448, 672, 3), dtype=np.uint8))
    item = m.load_images([str(p)])[0]
    assert item["type"] == "video"
    _, grid = m.preprocess(item)
    assert tuple(grid) == (1, 448 // 14, 672 // 14)
    assert smart_resize(448, 672, max_pixels=360 * 420) == (308, 476)          # what the cap would have produced


def test_multi_token_answers_processed_scores_and_eos_rule(tmp_path):
    """max_new_tokens > 1: greedy generation, answer tokens scored at the LAST positions of the generated scores, geometric
    mean (qwen2vl_model.py:265-289); repetition penalty applied to prompt + generated ids before the softmax (what HF
    generate's output_scores holds); a generation that is only a special token raises like the reference (:252-256)."""
    from oracle.qwen25vl_oracle import QwenOracle
    cfg = get_qwen_config("qwen-tiny")
    w = make_seeded_qwen_weights(cfg, seed=3, dtype=torch.bfloat16, lm_head_gain=4.0)
    tok = FakeQwenTokenizer(cfg.text.vocab)
    p = tmp_path / "img.npy"
    np.save(p, np.random.RandomState(2).randint(0, 256, (112, 112, 3), dtype=np.uint8))
    scorer = t2v.VQAScore(model="qwen2.5-vl-7b", device="cpu", config=cfg, engine=OracleQwenEngine(cfg, w), tokenizer=tok)
    m = scorer.model
    ans = "Yes indeed"
    a = tok.encode(ans)
    assert len(a) == 2
    s2 = m.forward([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=2)
    # by hand with the oracle
    item = m.load_images([str(p)])[0]
    patches, grid = m.preprocess(item)
    ids = m.build_ids(default_q("a red cube"), "image", grid[1] * grid[2] // 4)
    o = QwenOracle(cfg, w)

    def logits_of(seq):
        t = torch.tensor([seq])
        return o.forward(t, torch.ones_like(t), patches, [grid])[0]

    l0 = logits_of(ids)
    g0 = int(l0.argmax())
    l1 = logits_of(ids + [g0])
    want = (torch.softmax(l0, -1)[a[0]] * torch.softmax(l1, -1)[a[1]]).item() ** 0.5
    assert abs(s2[0].item() - want) <= 1e-5 * max(want, 1e-3) + 1e-9
    # a one-token budget truncates the answer to the tokens that were generated (:258-262)
    s1 = m.forward([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=1)
    assert abs(s1[0].item() - torch.softmax(l0, -1)[a[0]].item()) <= 1e-6
    # repetition penalty 1.3: the prompt's tokens are penalised before the softmax
    m.repetition_penalty = 1.3
    seen = torch.tensor(sorted(set(ids)))
    lp = l0.clone()
    lp[seen] = torch.where(lp[seen] < 0, lp[seen] * 1.3, lp[seen] / 1.3)
    in_prompt = tok.encode("Yes")[0]                       # "Yes" occurs in the question template
    assert in_prompt in ids
    sp = m.forward([str(p)], ["a red cube"], max_new_tokens=1)
    assert abs(sp[0].item() - torch.softmax(lp, -1)[in_prompt].item()) <= 1e-6 * max(1.0, sp[0].item()) + 1e-9
    m.repetition_penalty = 1.0
    # the generated token is the EOS: nothing left to score
    tok.eos_token_id = g0
    with pytest.raises(ValueError, match="No content tokens"):
        m.forward([str(p)], ["a red cube"], max_new_tokens=1)
    # HF generate STOPS at any id of generation_config.eos_token_id, not only at the tokenizer's eos (the special-token rule above
    # is the tokenizer's, as in the reference :239-244): with g0 among them a 2-token budget yields one score, and the answer is
    # truncated to it (:258-262) exactly as with a 1-token budget
    tok.eos_token_id = cfg.text.vocab - 1
    assert g0 != tok.eos_token_id
    assert abs(m.forward([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=2)[0].item() - s2[0].item()) <= 1e-7
    m._gen_eos_ids = [cfg.text.vocab - 2, g0]
    stopped = m.forward([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=2)
    assert abs(stopped[0].item() - s1[0].item()) <= 1e-7 and abs(s1[0].item() - s2[0].item()) > 1e-6


def test_generation_config_eos_ids_and_penalty_are_read_from_the_checkpoint_dir(tmp_path):
    import json
    from t2v_metrics_amd.models.vqascore_models.qwen25vl_model import read_generation_config
    assert read_generation_config(str(tmp_path)) == (1.0, [])
    (tmp_path / "generation_config.json").write_text(json.dumps({"eos_token_id": [151645, 151643], "repetition_penalty": 1.05, "do_sample": True}))
    assert read_generation_config(str(tmp_path)) == (1.05, [151645, 151643])
    (tmp_path / "generation_config.json").write_text(json.dumps({"eos_token_id": 151645}))
    assert read_generation_config(str(tmp_path)) == (1.0, [151645])


def default_q(text):
    from t2v_metrics_amd.models.vqascore_models.qwen25vl_model import default_question_template
    return default_question_template.format(text)


def test_generation_is_one_prefill_plus_cached_steps_and_generate_returns_text(tmp_path):
    """max_new_tokens > 1 and generate() (qwen2vl_model.py:222-230, :495-563) run ONE prefill per batch and one cached position per
    further token; greedy tokens equal the ones a prefill over prompt + tokens-so-far picks; generate() decodes them without the
    specials; sampling (temperature > 0) draws from the nucleus only."""
    cfg = get_qwen_config("qwen-tiny")
    w = make_seeded_qwen_weights(cfg, seed=3, dtype=torch.bfloat16, lm_head_gain=4.0)
    tok = FakeQwenTokenizer(cfg.text.vocab)
    paths = []
    for i, shape in enumerate([(112, 112, 3), (112, 112, 3), (2, 112, 168, 3)]):
        p = tmp_path / f"x{i}.npy"
        np.save(p, np.random.RandomState(i).randint(0, 256, shape, dtype=np.uint8))
        paths.append(str(p))
    eng = OracleQwenEngine(cfg, w)
    m = t2v.VQAScore(model="qwen2.5-vl-7b", device="cpu", config=cfg, engine=eng, tokenizer=tok).model
    m._gen_eos_ids = []                                    # never stop early: every sample runs the 4 steps
    texts = ["what is shown", "describe the picture please", "what happens"]
    out = m.generate(paths, texts, max_new_tokens=4)
    assert eng.prefills == 2 and len(out) == 3            # two grids -> two batches, one prefill each
    assert all(len(o.split()) <= 4 and all(t.startswith("t") for t in o.split()) for o in out)
    # by hand: re-prefill greedy over prompt + generated tokens
    item = m.load_images([paths[2]])[0]
    patches, grid = m.preprocess(item)
    merged = eng.encode_vision(patches, [grid])
    row = m.build_ids(texts[2], "video", grid[0] * grid[1] * grid[2] // 4)
    got = []
    for _ in range(4):
        ids = torch.tensor([row + got])
        got.append(int(eng.score_logits(merged, ids, torch.ones_like(ids), [grid]).argmax(-1)))
    assert out[2] == tok.decode(got)
    # nucleus sampling: with top_p -> 0 only the most likely token survives, so sampling equals greedy
    torch.manual_seed(0)
    assert m.generate(paths[2:], texts[2:], max_new_tokens=3, temperature=0.7, top_p=1e-6) == [tok.decode(got[:3])]
    torch.manual_seed(0)
    sampled = m.generate(paths[2:], texts[2:], max_new_tokens=3, temperature=5.0, top_p=0.95)
    assert len(sampled[0].split()) <= 3
    # stop ids end a sample; its text excludes the special
    first = got[0]
    m._gen_eos_ids = [first]
    assert m.generate(paths[2:], texts[2:], max_new_tokens=4) == [tok.decode([first])]


def test_cached_generation_matches_hf_generate_fixture(golden_dir):
    """tests/golden/qwen_tiny_gen4.npz = HF generate(use_cache=True, do_sample=False, 4 tokens) per sample, made by
    oracle/make_golden.py::golden_qwen_generate.  The fp32 double's prefill + decode (the semantics vqs_qwen_prefill / vqs_qwen_decode
    are stage-locked to) must give HF's scores at every step on the ragged batch -- which pins the position rule of generated tokens
    (the prompt's last position + 1 + step on each axis, what HF generate does): sample 1 is a long narrow video whose temporal
    positions (up to 17) run far past its last text token (position 8), where "max prompt position + 1" -- the rope_deltas formula --
    would rotate the new token differently (it misses HF's scores by 1.4 there)."""
    import os
    from t2v_metrics_amd.qwen.layout import text_layout
    z = np.load(os.path.join(golden_dir, "qwen_tiny_gen4.npz"))
    cfg = get_qwen_config("qwen-tiny")
    w = make_seeded_qwen_weights(cfg, seed=int(z["seed"]), dtype=torch.bfloat16, lm_head_gain=float(z["gain"]))
    grids = [tuple(int(x) for x in g) for g in z["grids"]]
    ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    px = torch.from_numpy(z["pixel_values"])
    want, want_ids = torch.from_numpy(z["gen_scores"]), torch.from_numpy(z["gen_ids"])          # [B, steps, vocab], [B, steps]
    lay = text_layout(cfg, ids, mask, grids)
    n1 = int(mask[1].sum())                                      # the long video: the decode position is NOT "max prompt position + 1"
    assert lay["next_pos"][:, 1].tolist() == [9, 9, 9] and int(z["prompt_position_max"][1]) + 1 == 18
    assert lay["next_pos"][:, 1].tolist() == (lay["position_ids"][:, 1, n1 - 1] + 1).tolist()
    eng = OracleQwenEngine(cfg, w)
    merged, off = [], 0
    for g in grids:
        n = g[0] * g[1] * g[2]
        merged.append(eng.encode_vision(px[off: off + n], [g]))
        off += n
    merged = torch.cat(merged)
    steps = want.shape[1]
    logits, state = eng.prefill(merged, ids, mask, grids, steps)
    scale = max(1.0, want.abs().max().item())
    for t in range(steps):
        assert (logits - want[:, t]).abs().max().item() <= 5e-4 * scale, t
        assert logits.argmax(-1).tolist() == want_ids[:, t].tolist()
        if t + 1 < steps:
            logits = eng.decode(state, want_ids[:, t])


def test_logits_processing_equals_hf_processors():
    """The wrapper's repetition penalty and temperature + nucleus filter against HF's own processors on random scores."""
    lp = pytest.importorskip("transformers.generation.logits_process")
    g = torch.Generator().manual_seed(3)
    scores = torch.randn(5, 300, generator=g) * 3
    rows = [torch.randint(0, 300, (int(n),), generator=g).tolist() for n in (7, 40, 1, 120, 12)]
    cfg = get_qwen_config("qwen-tiny")
    m = Qwen25VLModel(model_name="qwen2.5-vl-7b", device="cpu", config=cfg, engine=object(), tokenizer=FakeQwenTokenizer(cfg.text.vocab))
    m.repetition_penalty = 1.05
    want = torch.stack([lp.RepetitionPenaltyLogitsProcessor(1.05)(torch.tensor([r]), scores[k: k + 1].clone())[0] for k, r in enumerate(rows)])
    assert torch.equal(m._processed_scores(scores, rows), want)
    for temperature, top_p in ((0.7, 0.9), (1.3, 0.5), (1.0, 0.999), (2.0, 1e-6)):
        ref = lp.TopPLogitsWarper(top_p)(None, lp.TemperatureLogitsWarper(temperature)(None, scores.clone()))
        got = Qwen25VLModel._warp(scores, temperature, top_p)
        assert torch.equal(torch.isinf(got), torch.isinf(ref)) and torch.allclose(got[~torch.isinf(got)], ref[~torch.isinf(ref)])
    assert int((~torch.isinf(Qwen25VLModel._warp(scores, 2.0, 1e-6))).sum()) == scores.shape[0]      # only the top token survives
    # the chain HF itself builds for the reference's call generate(do_sample=True, temperature=, top_p=) with the library's
    # default generation config: temperature -> top-k 50 -> top-p (the top-k warper was missing here)
    from transformers import GenerationConfig
    from transformers.generation.utils import GenerationMixin
    for temperature, top_p in ((0.7, 0.9), (3.0, 0.95), (5.0, 0.999)):
        gc = GenerationConfig(do_sample=True, temperature=temperature, top_p=top_p)
        if gc.top_k is None:          # transformers 5.x applies the library defaults at generate() time (_prepare_generation_config)
            gc.update(**GenerationConfig._get_default_generation_params(), defaults_only=True)
        assert gc.top_k == 50
        chain = GenerationMixin._get_logits_processor(GenerationMixin.__new__(GenerationMixin), gc, input_ids_seq_length=1,
                                                      encoder_input_ids=None, prefix_allowed_tokens_fn=None, logits_processor=[])
        assert [type(c).__name__ for c in chain] == ["TemperatureLogitsWarper", "TopKLogitsWarper", "TopPLogitsWarper"]
        ref = chain(torch.zeros(scores.shape[0], 1, dtype=torch.long), scores.clone())
        got = Qwen25VLModel._warp(scores, temperature, top_p, 50)
        assert torch.equal(torch.isinf(got), torch.isinf(ref)) and torch.allclose(got[~torch.isinf(got)], ref[~torch.isinf(ref)])
        assert int((~torch.isinf(got)).sum(-1).max()) <= 50
    assert m.top_k == 50                                                    # no generation_config.json: HF's default


def test_generation_config_top_k_is_read(tmp_path):
    import json
    from t2v_metrics_amd.models.vqascore_models.qwen25vl_model import read_top_k
    assert read_top_k(str(tmp_path)) == 50
    (tmp_path / "generation_config.json").write_text(json.dumps({"top_k": 1, "top_p": 0.001, "repetition_penalty": 1.05}))
    assert read_top_k(str(tmp_path)) == 1


def test_grid_costs_one_tower_pass_per_medium_and_video_paths_reach_the_model(tmp_path):
    """An M x N grid runs the vision tower M times (the reference: M * N, score.py:104-106): Score.forward goes through
    forward_grid, identical paths are one medium, and a medium's pairs never straddle two tower calls even when M * N exceeds
    max_batch.  Container paths are passed to a video_mode == 'direct' model as the reference does (score.py:69-101); the model
    itself refuses what it cannot decode (qwen2vl_model.py:135-158 needs decord)."""
    cfg = get_qwen_config("qwen-tiny")
    w = make_seeded_qwen_weights(cfg, seed=3, dtype=torch.bfloat16, lm_head_gain=4.0)
    eng = OracleQwenEngine(cfg, w)
    tok = FakeQwenTokenizer(cfg.text.vocab)
    rng = np.random.RandomState(5)
    paths = []
    for i, shape in enumerate([(4, 112, 112, 3), (4, 112, 112, 3), (112, 168, 3)]):
        p = tmp_path / f"m{i}.npy"
        np.save(p, rng.randint(0, 256, shape, dtype=np.uint8))
        paths.append(str(p))
    texts = ["a cat jumps", "a dog runs fast", "two birds", "a red car", "rain"]
    scorer = t2v.VQAScore(model="qwen2.5-vl-7b", device="cpu", config=cfg, engine=eng, tokenizer=tok, max_batch=4)
    grid = scorer(images=paths, texts=texts)                       # 3 x 5 = 15 pairs, max_batch 4
    assert grid.shape == (3, 5)
    media_encoded = sum(len(c) for c in eng.vision_calls)
    assert media_encoded == 3, eng.vision_calls                    # one tower pass per medium, not 15
    # the grid equals pair-by-pair scoring (the reference's row loop)
    eng.vision_calls.clear()
    for i in (0, 2):
        row = scorer.model.forward([paths[i]] * len(texts), texts)
        assert torch.allclose(grid[i].cpu(), row, rtol=1e-5, atol=1e-7)
    assert sum(len(c) for c in eng.vision_calls) == 2              # [path] * N is ONE medium as well
    # video container paths: passed through to the direct-mode model, which refuses them itself
    assert scorer.model.video_mode == "direct"
    with pytest.raises(NotImplementedError, match="decord"):
        scorer(images=["clip.mp4"], texts=["x"])


def test_forward_with_trace_follows_the_reference_rules(tmp_path):
    """forward_with_trace (qwen2vl_model.py:303-493): scores equal forward() for score_position='end'; 'start' scores the first
    generated positions; the trace fields and the special-token / short-generation adjustments follow the reference's arithmetic,
    restated literally here on the double's own generation."""
    cfg = get_qwen_config("qwen-tiny")
    w = make_seeded_qwen_weights(cfg, seed=3, dtype=torch.bfloat16, lm_head_gain=4.0)
    tok = FakeQwenTokenizer(cfg.text.vocab)
    p = tmp_path / "img.npy"
    np.save(p, np.random.RandomState(2).randint(0, 256, (112, 112, 3), dtype=np.uint8))
    m = t2v.VQAScore(model="qwen2.5-vl-7b", device="cpu", config=cfg, engine=OracleQwenEngine(cfg, w), tokenizer=tok).model
    m._gen_eos_ids = []
    ans = "Yes indeed"
    a = tok.encode(ans)
    for pos_mode in ("end", "start"):
        s, tr = m.forward_with_trace([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=4, temperature=0.8, score_position=pos_mode)
        t = tr[0]
        assert t["generated_length"] == 4 and t["score_position"] == pos_mode and len(t["token_details"]) == 2
        assert t["scored_indices"] == ([2, 3] if pos_mode == "end" else [0, 1]) and t["score_start_idx"] == t["scored_indices"][0]
        probs = [d["probability"] for d in t["token_details"]]
        assert abs(t["probability"] - (probs[0] * probs[1]) ** 0.5) <= 1e-7 and abs(float(s[0]) - t["probability"]) <= 1e-7
        for d, want_id in zip(t["token_details"], a):
            assert d["expected_token_id"] == want_id and d["expected_token_text"] == tok.decode([want_id])
            alts = d["top_alternatives"]
            assert len(alts) == 5 and all(alts[i]["probability"] >= alts[i + 1]["probability"] for i in range(4))
            assert all(x["token_text"] == tok.decode([x["token_id"]]) for x in alts)
            hit = [x for x in alts if x["token_id"] == want_id]
            assert not hit or abs(hit[0]["probability"] - d["probability"]) <= 1e-6
        assert t["generated_text"].split() == [f"t{i}" for i in _generated_ids(m, p, 4)]
    # 'end' equals forward()
    s_end, _ = m.forward_with_trace([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=4, temperature=0.8)
    assert abs(float(s_end[0]) - float(m.forward([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=4, temperature=0.8)[0])) <= 1e-7
    # generation ends in a special token: one position earlier; fewer generated tokens than answer tokens: n is cut
    gen = _generated_ids(m, p, 4)
    tok.eos_token_id = gen[1]
    m._gen_eos_ids = [gen[1]]
    s_sp, tr = m.forward_with_trace([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=4)
    assert tr[0]["generated_length"] == 2 and tr[0]["scored_indices"] == [0] and len(tr[0]["token_details"]) == 1     # n = min(2, 2 - 1), offset 1
    s_st, tr = m.forward_with_trace([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=4, score_position="start")
    assert tr[0]["scored_indices"] == [0, 1]                                                                          # 'start' ignores the special rule
    tok.eos_token_id = gen[0]
    m._gen_eos_ids = [gen[0]]
    with pytest.raises(ValueError, match="No tokens available"):
        m.forward_with_trace([str(p)], ["a red cube"], answer_template=ans, max_new_tokens=4)
    with pytest.raises(AssertionError):
        m.forward_with_trace([str(p)], ["a red cube"], score_position="middle")


def _generated_ids(m, path, n):
    """The model's greedy ids for the default question on `path`, stop ids disabled."""
    keep = m._gen_eos_ids
    m._gen_eos_ids = []
    try:
        from t2v_metrics_amd.models.vqascore_models.qwen25vl_model import default_question_template
        return m._generate_scores([str(path)], ["a red cube"], None, default_question_template, "Yes", n)[0][1]
    finally:
        m._gen_eos_ids = keep


def test_batch_forward_takes_a_videos_dataset_for_a_direct_mode_model(tmp_path):
    """The reference keys a dataset's media under "videos" or "images" (score.py:124-128); Qwen2.5-VL (video_mode "direct") scores either
    through batch_forward, an image-only model refuses a video dataset."""
    cfg = get_qwen_config("qwen-tiny")
    w = make_seeded_qwen_weights(cfg, seed=3, dtype=torch.bfloat16, lm_head_gain=4.0)
    eng = OracleQwenEngine(cfg, w)
    rng = np.random.RandomState(2)
    paths = []
    for i in range(3):
        p = tmp_path / f"v{i}.npy"
        np.save(p, rng.randint(0, 256, (4, 112, 112, 3), dtype=np.uint8))
        paths.append(str(p))
    scorer = t2v.VQAScore(model="qwen2.5-vl-7b", device="cpu", config=cfg, engine=eng, tokenizer=FakeQwenTokenizer(cfg.text.vocab))
    ds_v = [{"videos": [paths[k]], "texts": ["a cat jumps", "a dog runs"]} for k in range(3)]
    ds_i = [{"images": [paths[k]], "texts": ["a cat jumps", "a dog runs"]} for k in range(3)]
    sv, si = scorer.batch_forward(ds_v, batch_size=2), scorer.batch_forward(ds_i, batch_size=2)
    assert sv.shape == (3, 1, 2) and torch.equal(sv, si)
    from tests.test_host_api import FakeTokenizer, RecordingEngine
    from t2v_metrics_amd.config import get_config
    ccfg = get_config("tiny")
    img_only = t2v.VQAScore(model="clip-flant5-xl", device="cpu", config=ccfg, engine=RecordingEngine(ccfg), tokenizer=FakeTokenizer(ccfg.t5.vocab))
    with pytest.raises(NotImplementedError, match="video-native"):
        img_only.batch_forward(ds_v)


def test_decode_workspace_covers_the_partials_of_every_linear_at_other_model_sizes():
    """the decode step's split-K scratch was sized by a hand-written bound that missed gate|up with two K slices (Qwen2.5-VL-3B:
    hidden 2048, mlp 11008 -- every decode step would have returned VQS_ERR_WORKSPACE).  carve_decode now sizes it with the slice rule
    decode_linear launches with; here the rule is restated and the host-side workspace query (no GPU needed) checked against it."""
    import ctypes
    from t2v_metrics_amd.engine import load_library
    from t2v_metrics_amd.qwen.engine import VqsQwenConfig, _SIGS
    lib = load_library()
    for name in ("vqs_qwen_create", "vqs_qwen_destroy", "vqs_qwen_decode_workspace_bytes"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _SIGS[name]

    def slices(N, K):
        nblk, nsl = (N + 127) // 128, K // 64
        if nblk >= 192 or K % 64:
            return 1
        best = 1
        for s in range(2, 17):
            if nsl % s == 0:
                best = s
                if nblk * s >= 192:
                    break
        return best

    for hidden, heads, kv, mlp in ((2048, 16, 2, 11008), (1024, 8, 2, 4096), (3584, 28, 4, 18944)):     # 3B, a 1024-wide model, 7B
        c = VqsQwenConfig(32, 1280, 16, 3420, 1176, 4, hidden, 0x80808080, 1e-6, 151936, hidden, 4, heads, kv, mlp, 1e-6)
        h = ctypes.c_void_p()
        assert lib.vqs_qwen_create(ctypes.byref(c), ctypes.byref(h)) == 0
        try:
            QN, IQ = (heads + 2 * kv) * 128, heads * 128
            gu = 2 * (-(-mlp // 32) * 32)
            ffld = -(-mlp // 64) * 64
            widest = max(slices(QN, hidden) * QN, slices(hidden, IQ) * hidden, slices(gu, hidden) * gu, slices(hidden, ffld) * hidden)
            for B in (1, 4, 64):
                assert lib.vqs_qwen_decode_workspace_bytes(h, B) >= B * widest * 4, (hidden, mlp, B)
        finally:
            lib.vqs_qwen_destroy(h)


def test_non_finite_logits_under_the_fp16_forms_fall_back_to_bf16_once(tmp_path):
    """VERDICT r5 item 2, Qwen row: the wrapper never raises on finite inputs -- a non-finite logit while the engine's fp16 forms are active
    switches option fp16 off (one warning) and the call is run again; scores equal the bf16 run's, later calls stay on bf16 without a warning."""
    import warnings
    import t2v_metrics_amd as t2v
    cfg = get_qwen_config("qwen-tiny")
    w = make_seeded_qwen_weights(cfg, seed=3, dtype=torch.bfloat16, lm_head_gain=4.0)

    class Overflowing(OracleQwenEngine):
        fp16_active = True
        switched = []

        def set_option(self, name, value):
            self.switched.append((name, value))
            if name == "fp16":
                self.fp16_active = bool(value)

        def score_logits(self, *a):
            lg = super().score_logits(*a)
            return lg * float("nan") if self.fp16_active else lg

    rng = np.random.RandomState(5)
    p = tmp_path / "v.npy"
    np.save(p, rng.randint(0, 256, (2, 56, 56, 3), dtype=np.uint8))
    tok = FakeQwenTokenizer(cfg.text.vocab)
    eng = Overflowing(cfg, w)
    m = t2v.VQAScore(model="qwen2.5-vl-7b", device="cpu", config=cfg, engine=eng, tokenizer=tok).model
    ref = t2v.VQAScore(model="qwen2.5-vl-7b", device="cpu", config=cfg, engine=OracleQwenEngine(cfg, w), tokenizer=tok).model
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        s = m.forward([str(p)], ["a thing happens"])
    assert torch.isfinite(s).all() and torch.equal(s, ref.forward([str(p)], ["a thing happens"]))
    assert eng.switched == [("fp16", 0)] and sum("fp16" in str(r.message) for r in rec) == 1
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m.forward([str(p)], ["a thing happens"])
    assert not rec and eng.switched == [("fp16", 0)]
