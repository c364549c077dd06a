"""The Qwen2.5-VL wrapper's scoring tail against the reference's OWN ``Qwen2VLModel.forward`` / ``forward_with_trace``
(/root/reference/t2v_metrics/models/vqascore_models/qwen2vl_model.py:165-301, 303-493), run here from where it lies.

The reference methods are driven without a checkpoint: an instance made with ``object.__new__`` gets a stub processor (chat
template = this package's ``chat_prompt``, tokenizer = the same fake tokenizer our wrapper is given) and a stub
``model.generate`` that is plain greedy decoding over a SCRIPTED language model -- logits are a seeded function of (prompt text
ids, tokens generated so far).  Our wrapper gets the same scripted model behind its engine interface.  What is compared is what
the two wrappers do with identical generations: which score rows they read (last-n, one earlier after a trailing special
token, first-n for ``score_position="start"``), truncation to what was generated, temperature, the geometric mean, the trace
fields, and the errors.

Skips when /root/reference is absent (the GPU box)."""
import types
import zlib

import numpy as np
import pytest
import torch

import t2v_metrics_amd as t2v
from t2v_metrics_amd.models.vqascore_models.qwen25vl_model import chat_prompt
from t2v_metrics_amd.qwen import get_qwen_config
from tests.reference_loader import reference_module
from tests.test_qwen_host import SPECIALS, FakeQwenTokenizer

CFG = get_qwen_config("qwen-tiny")
PADS = (SPECIALS["<|image_pad|>"], SPECIALS["<|video_pad|>"])
assert (CFG.image_token_id, CFG.video_token_id) == PADS


class Tokenizer(FakeQwenTokenizer):
    eos_token_id = None
    bos_token_id = None
    pad_token_id = None


def scripted_logits(prompt_ids, generated, gain=6.0):
    """The scripted LM: a [vocab] fp32 row that depends on the prompt's text tokens (vision placeholders dropped: their count is
    the wrappers' business, not the script's) and on what has been generated so far."""
    key = ",".join(str(int(i)) for i in prompt_ids if int(i) not in PADS) + "|" + ",".join(str(int(g)) for g in generated)
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    return gain * torch.randn(CFG.text.vocab, generator=g)


class ScriptedEngine:
    """The scripted LM behind this package's engine interface (vision rows are ignored by the script)."""

    cfg = CFG

    def encode_vision(self, patches, grids):
        n = sum(t * h * w for t, h, w in grids) // CFG.vision.merge_unit
        return torch.zeros(n, CFG.text.hidden)

    def _rows(self, ids, mask):
        return [ids[k][mask[k].bool()].tolist() for k in range(ids.shape[0])]

    def score_logits(self, merged, ids, mask, grids):
        return torch.stack([scripted_logits(r, ()) for r in self._rows(ids, mask)])

    def prefill(self, merged, ids, mask, grids, max_new_tokens):
        rows = self._rows(ids, mask)
        return torch.stack([scripted_logits(r, ()) for r in rows]), {"rows": rows, "gen": [[] for _ in rows]}

    def decode(self, state, token_ids):
        for g, t in zip(state["gen"], token_ids.tolist()):
            g.append(int(t))
        return torch.stack([scripted_logits(r, g) for r, g in zip(state["rows"], state["gen"])])


class Inputs(dict):
    """What the HF processor returns, as far as the reference touches it: mapping + attribute access + ``.to``."""

    __getattr__ = dict.__getitem__

    def to(self, device):
        return self


class StubProcessor:
    def __init__(self, tok):
        self.tokenizer = tok

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True):
        assert tokenize is False and add_generation_prompt is True and len(messages) == 1 and messages[0]["role"] == "user"
        medium, text = messages[0]["content"]
        return chat_prompt(text["text"], "<|video_pad|>" if medium["type"] == "video" else "<|image_pad|>")

    def batch_decode(self, rows, skip_special_tokens=True, clean_up_tokenization_spaces=False):
        return [self.tokenizer.decode(r, skip_special_tokens=skip_special_tokens) for r in rows]

    def __call__(self, text, images=None, videos=None, padding=True, return_tensors="pt", do_resize=False, **kw):
        assert len(text) == 1 and (images is None) != (videos is None) and do_resize is False
        return Inputs(input_ids=torch.tensor([self.tokenizer.encode(text[0], add_special_tokens=False)]))


class StubGenerator:
    """HF ``generate(do_sample=False, output_scores=True, return_dict_in_generate=True)`` over the scripted LM: greedy, stops after
    emitting one of ``stop_ids`` (generation_config.eos_token_id), ``scores[t]`` = the [1, vocab] row token t was picked from."""

    def __init__(self, stop_ids):
        self.stop_ids = list(stop_ids)
        self.calls = 0

    def generate(self, input_ids, max_new_tokens, do_sample, temperature=1.0, output_scores=False, return_dict_in_generate=False):
        assert temperature == 1.0 and do_sample is False and output_scores == return_dict_in_generate
        self.calls += 1
        prompt, gen, scores = input_ids[0].tolist(), [], []
        for _ in range(max_new_tokens):
            row = scripted_logits(prompt, gen)
            scores.append(row[None])
            gen.append(int(row.argmax()))
            if gen[-1] in self.stop_ids:
                break
        if not return_dict_in_generate:                       # the free-form generate() call: just the sequences
            return torch.tensor([prompt + gen])
        return types.SimpleNamespace(sequences=torch.tensor([prompt + gen]), scores=tuple(scores))


def _pair(tmp_path, stop_ids=(), **special):
    """(our wrapper, the reference's wrapper) over the same scripted LM, tokenizer and stop ids."""
    ref = reference_module("models.vqascore_models.qwen2vl_model", optional=("qwen_vl_utils", "decord", "cv2"))

    def process_vision_info(messages, return_video_kwargs=False):
        medium = messages[0]["content"][0]
        if medium["type"] == "video":
            return None, [medium["video"]], {"fps": [2.0]}
        return [medium["image"]], None, {}
    ref.process_vision_info = process_vision_info

    tok = Tokenizer(CFG.text.vocab)
    for k, v in special.items():
        setattr(tok, k, v)
    theirs = object.__new__(ref.Qwen2VLModel)                 # no checkpoint: __init__ / load_model never run
    theirs.model_name, theirs.device, theirs.model_info = "qwen2.5-vl-7b", "cpu", {"fps": 8.0}
    theirs.processor, theirs.model = StubProcessor(tok), StubGenerator(stop_ids)
    ours = t2v.VQAScore(model="qwen2.5-vl-7b", device="cpu", config=CFG, engine=ScriptedEngine(), tokenizer=tok).model
    ours._gen_eos_ids = list(stop_ids)
    return ours, theirs


@pytest.fixture()
def media(tmp_path):
    rs = np.random.RandomState(4)
    paths = []
    for k, shape in enumerate([(56, 56, 3), (84, 56, 3), (4, 56, 56, 3)]):        # two stills, one 4-frame clip
        p = tmp_path / f"m{k}.npy"
        np.save(p, rs.randint(0, 256, shape, dtype=np.uint8))
        paths.append(str(p))
    return paths


TEXTS = ["a red cube", "two dogs on a beach", "someone opens a door"]


def _first_tokens(texts, n):
    """What the scripted LM generates greedily for each default-template prompt: the test picks its special ids from these."""
    tok = Tokenizer(CFG.text.vocab)
    out = []
    for t, ph in zip(texts, ("<|image_pad|>", "<|image_pad|>", "<|video_pad|>")):
        ids = tok.encode(chat_prompt(f'Does this figure show "{t}"? Please answer Yes or No.', ph))
        gen = []
        for _ in range(n):
            gen.append(int(scripted_logits(ids, gen).argmax()))
        out.append(gen)
    return out


@pytest.mark.parametrize("answer", ["Yes", "Yes indeed", "Yes it certainly is"])
@pytest.mark.parametrize("budget", [1, 2, 3, 5])
def test_forward_reads_the_same_score_rows_as_the_reference(tmp_path, media, answer, budget, capsys):
    ours, theirs = _pair(tmp_path)
    for temperature in (1.0, 0.5, 2.5):
        a = ours.forward(media, TEXTS, answer_template=answer, max_new_tokens=budget, temperature=temperature)
        b = theirs.forward(media, TEXTS, answer_template=answer, max_new_tokens=budget, temperature=temperature)
        assert a.shape == b.shape == (3,) and a.dtype == b.dtype == torch.float32
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-12), (a, b)
    capsys.readouterr()                                        # the reference prints a warning when it truncates the answer


def test_trailing_special_token_shifts_the_window_in_both(tmp_path, media):
    """A generation that ENDS in eos / bos / pad is scored one row earlier (:239-256); one that is ONLY the special token has nothing
    left and raises the same error in both."""
    g = _first_tokens(TEXTS, 3)
    # sample 0 stops at its 3rd token, sample 1 at its 2nd (eos), sample 2 never
    assert len({g[0][2], g[1][1]}) == 2 and g[0][2] not in g[0][:2] + g[2] and g[1][1] not in g[1][:1] + g[2] + g[0][:2]
    ours, theirs = _pair(tmp_path, stop_ids=(g[0][2], g[1][1]), eos_token_id=g[1][1], pad_token_id=g[0][2])
    for answer in ("Yes", "Yes indeed", "Yes it certainly is"):
        a = ours.forward(media, TEXTS, answer_template=answer, max_new_tokens=4)
        b = theirs.forward(media, TEXTS, answer_template=answer, max_new_tokens=4)
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-12), (answer, a, b)
    # the stop id is what HF generate stops at, the special-token rule is the TOKENIZER's: a stop id that is not a tokenizer
    # special is scored like any other last token
    ours, theirs = _pair(tmp_path, stop_ids=(g[0][2], g[1][1]), eos_token_id=g[1][1])
    a = ours.forward(media, TEXTS, answer_template="Yes indeed", max_new_tokens=4)
    b = theirs.forward(media, TEXTS, answer_template="Yes indeed", max_new_tokens=4)
    assert torch.allclose(a, b, rtol=2e-6, atol=1e-12)
    # only a special token generated: nothing to score
    ours, theirs = _pair(tmp_path, stop_ids=(g[0][0],), bos_token_id=g[0][0])
    for m in (ours, theirs):
        with pytest.raises(ValueError, match="No content tokens to score after removing special tokens"):
            m.forward(media[:1], TEXTS[:1], max_new_tokens=3)


def _same_trace(a, b):
    assert set(a) == set(b), (set(a) ^ set(b))
    for k in a:
        if k == "probability":
            assert a[k] == pytest.approx(b[k], rel=2e-6, abs=1e-12)
        elif k == "token_details":
            assert len(a[k]) == len(b[k])
            for da, db in zip(a[k], b[k]):
                assert set(da) == set(db)
                assert (da["position"], da["expected_token_id"], da["expected_token_text"]) == \
                       (db["position"], db["expected_token_id"], db["expected_token_text"])
                assert da["probability"] == pytest.approx(db["probability"], rel=2e-6, abs=1e-12)
                assert [x["token_id"] for x in da["top_alternatives"]] == [x["token_id"] for x in db["top_alternatives"]]
                assert [x["token_text"] for x in da["top_alternatives"]] == [x["token_text"] for x in db["top_alternatives"]]
                for xa, xb in zip(da["top_alternatives"], db["top_alternatives"]):
                    assert set(xa) == set(xb) and xa["probability"] == pytest.approx(xb["probability"], rel=2e-6, abs=1e-12)
        else:
            assert a[k] == b[k], k


@pytest.mark.parametrize("score_position", ["end", "start"])
def test_forward_with_trace_is_the_references(tmp_path, media, score_position, capsys):
    g = _first_tokens(TEXTS, 3)
    for stops, special in (((), {}), ((g[0][2], g[1][1]), {"eos_token_id": g[1][1], "pad_token_id": g[0][2]})):
        ours, theirs = _pair(tmp_path, stop_ids=stops, **special)
        for answer, budget, temperature in (("Yes", 1, 1.0), ("Yes indeed", 4, 0.7), ("Yes it certainly is", 2, 1.0),
                                            ("Yes it certainly is", 6, 1.9)):
            pa, ta = ours.forward_with_trace(media, TEXTS, answer_template=answer, max_new_tokens=budget, temperature=temperature,
                                             score_position=score_position)
            pb, tb = theirs.forward_with_trace(media, TEXTS, answer_template=answer, max_new_tokens=budget, temperature=temperature,
                                               score_position=score_position)
            assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-12) and len(ta) == len(tb) == 3
            for x, y in zip(ta, tb):
                _same_trace(x, y)
    capsys.readouterr()
    for m in _pair(tmp_path):
        with pytest.raises(AssertionError, match="score_position must be"):
            m.forward_with_trace(media[:1], TEXTS[:1], score_position="middle")
        with pytest.raises(AssertionError, match="must match"):
            m.forward(media, TEXTS[:2])


def test_one_generation_per_pair_in_the_reference_one_tower_pass_per_medium_here(tmp_path, media):
    ours, theirs = _pair(tmp_path)
    seen = []
    enc = ours.engine.encode_vision
    ours.engine.encode_vision = lambda patches, grids: (seen.append(list(grids)), enc(patches, grids))[1]
    images, texts = [media[0]] * 3 + [media[1]], TEXTS + ["a fourth caption"]
    a, b = ours.forward(images, texts), theirs.forward(images, texts)
    assert torch.allclose(a, b, rtol=2e-6, atol=1e-12)
    assert theirs.model.calls == 4 and sum(len(g) for g in seen) == 2          # 4 generate() calls there; 2 media encoded here


def test_greedy_generate_returns_the_references_strings(tmp_path, media):
    """generate() (:495-563) with temperature 0: the text is the whole user turn, greedy until a stop id or the budget, decoded without
    special tokens and stripped."""
    g = _first_tokens(TEXTS, 3)
    for stops, special in (((), {}), ((g[0][2], g[1][1]), {"eos_token_id": g[1][1]})):
        ours, theirs = _pair(tmp_path, stop_ids=stops, **special)
        questions = [f'Does this figure show "{t}"? Please answer Yes or No.' for t in TEXTS]     # so that _first_tokens' script applies
        for budget in (1, 3, 7):
            a = ours.generate(media, questions, max_new_tokens=budget)
            b = theirs.generate(media, questions, max_new_tokens=budget)
            assert a == b and len(a) == 3 and all(isinstance(x, str) for x in a)
        if stops:                                                                                  # stopped early where the script says so
            full = ours.generate(media, questions, max_new_tokens=7)
            assert len(full[0].split()) <= 3 and len(full[1].split()) <= 2 and len(full[2].split()) == 7
    for m in _pair(tmp_path):
        with pytest.raises(AssertionError, match="must match"):
            m.generate(media, TEXTS[:1])
