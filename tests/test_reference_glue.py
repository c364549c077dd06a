"""The host glue against the reference's OWN code, where that code survives in /root/reference (this container only).

What the reference still holds of the CLIP-FlanT5 path (the model wrapper itself was dropped in v3.1) is
``constants.py``, ``score.py`` (``Score.forward`` / ``batch_forward``) and the helpers in
``models/vqascore_models/mm_utils.py`` (``expand2square``, ``t5_tokenizer_image_token``).  These tests import those files
as they lie -- without executing the package ``__init__`` files, which pull in every model family -- and run them next to
this package's restatements on the same inputs: same ids, same pixels, same [M, N] / [n, n_vis, n_txt] tensors.

/root/reference does not exist on the GPU box: every test here skips there, and nothing under ``-m gpu`` depends on it.
"""
import types
import zlib

import numpy as np
import pytest
import torch
from PIL import Image

import t2v_metrics_amd.constants as our_constants
from t2v_metrics_amd.models.vqascore_models.mm_utils import expand2square, t5_tokenizer_image_token
from t2v_metrics_amd.score import Score as OurScore

from tests.reference_loader import reference_module


def _reference():
    """``(constants, mm_utils, score)`` modules of the reference, loaded from where they lie."""
    return (reference_module("constants"), reference_module("models.vqascore_models.mm_utils", optional=("cv2",)),
            reference_module("score", optional=("cv2",)))


class WordTokenizer:
    """HF call protocol with the T5 convention (no BOS, trailing </s> = 1 on every call, also on the empty string)."""

    bos_token_id = None

    def __call__(self, text):
        return types.SimpleNamespace(input_ids=[3 + zlib.crc32(w.encode()) % 32000 for w in text.split()] + [1])


class PieceTokenizer(WordTokenizer):
    """Same protocol over a real sentencepiece unigram model (trained here on the prompts, T5's special ids)."""

    def __init__(self, tmp_path, corpus):
        spm = pytest.importorskip("sentencepiece")
        src = tmp_path / "corpus.txt"
        src.write_text("\n".join(corpus * 20))
        spm.SentencePieceTrainer.train(input=str(src), model_prefix=str(tmp_path / "sp"), vocab_size=96, model_type="unigram",
                                       pad_id=0, eos_id=1, unk_id=2, bos_id=-1, hard_vocab_limit=False, minloglevel=2)
        self.sp = spm.SentencePieceProcessor(model_file=str(tmp_path / "sp.model"))

    def __call__(self, text):
        return types.SimpleNamespace(input_ids=self.sp.encode(text) + [1])


PROMPTS = [
    our_constants.SYSTEM_MSG + ' USER: <image>\nDoes this figure show "a dog chasing a red ball"? Please answer yes or no. ASSISTANT: ',
    "<image>", "", "no image token at all", "<image><image>", "tail <image>", "<image> head", "a <image> b <image> c",
    "  spaces   around  <image>   kept by split  ", "<image>\n<image>\n", "<Image> is not the token", "x<image>y",
]


def test_constants_are_the_references():
    ref, _, _ = _reference()
    for name in ("HF_CACHE_DIR", "CONTEXT_LEN", "SYSTEM_MSG", "IGNORE_INDEX", "IMAGE_TOKEN_INDEX", "DEFAULT_IMAGE_TOKEN"):
        assert getattr(our_constants, name) == getattr(ref, name), name


def test_t5_tokenizer_image_token_is_the_references(tmp_path):
    _, ref, _ = _reference()
    for tok in (WordTokenizer(), PieceTokenizer(tmp_path, [p for p in PROMPTS if p.strip()])):
        for prompt in PROMPTS:
            want = ref.t5_tokenizer_image_token(prompt, tok)
            assert t5_tokenizer_image_token(prompt, tok) == want, prompt
            assert want.count(ref.IMAGE_TOKEN_INDEX) == prompt.count("<image>")
            got = t5_tokenizer_image_token(prompt, tok, return_tensors="pt")
            assert got.dtype == torch.long and torch.equal(got, ref.t5_tokenizer_image_token(prompt, tok, return_tensors="pt"))
        with pytest.raises(ValueError, match="Unsupported tensor type"):
            t5_tokenizer_image_token("x", tok, return_tensors="np")
        with pytest.raises(ValueError, match="Unsupported tensor type"):
            ref.t5_tokenizer_image_token("x", tok, return_tensors="np")
        other = t5_tokenizer_image_token("a <image> b", tok, image_token_index=-7)
        assert other == ref.t5_tokenizer_image_token("a <image> b", tok, image_token_index=-7) and other.count(-7) == 1


@pytest.mark.parametrize("mode,fill", [("RGB", (124, 116, 104)), ("RGB", (0, 0, 0)), ("L", 77), ("RGBA", (1, 2, 3, 4))])
def test_expand2square_is_the_references(mode, fill):
    _, ref, _ = _reference()
    rng = np.random.default_rng(5)
    for w, h in ((10, 4), (4, 10), (7, 7), (1, 9), (9, 1), (33, 32), (32, 33), (640, 427), (427, 640)):
        bands = len(Image.new(mode, (1, 1)).getbands())
        px = rng.integers(0, 256, (h, w) if bands == 1 else (h, w, bands), dtype=np.uint8)
        im = Image.fromarray(px, mode)
        ours, theirs = expand2square(im, fill), ref.expand2square(im, fill)
        assert ours.size == theirs.size == (max(w, h),) * 2 and ours.mode == theirs.mode == mode
        assert np.array_equal(np.asarray(ours), np.asarray(theirs))
        if w == h:
            assert ours is im and theirs is im               # the square case hands the SAME object back, in both


class PairModel:
    """A score model both Score classes can drive: a deterministic function of the (image, text) strings; keeps the calls."""

    video_mode = "concat"
    allows_image = True

    def __init__(self):
        self.calls = []

    def forward(self, images, texts, **kwargs):
        assert len(images) == len(texts)
        self.calls.append((list(images), list(texts), dict(kwargs)))
        gain = kwargs.get("gain", 1.0)
        return torch.tensor([gain * (zlib.crc32((i + "|" + t).encode()) % 10007) / 10007.0 for i, t in zip(images, texts)])


def _scorers():
    _, _, ref_score = _reference()

    def make(base):
        class S(base):
            def list_all_models(self):
                return ["pair-model"]

            def prepare_scoremodel(self, model, device, cache_dir, **kwargs):
                return PairModel()
        return S("pair-model", device="cpu")
    return make(OurScore), make(ref_score.Score)


def test_score_forward_grid_is_the_references():
    ours, theirs = _scorers()
    images = [f"img_{k}.png" for k in range(5)]
    texts = [f"caption number {k}" for k in range(3)]
    for im, tx in ((images, texts), (images[0], texts), (images, texts[1]), (images[2], texts[2]), (images[:1] * 3, texts)):
        a, b = ours(images=im, texts=tx), theirs(images=im, texts=tx)
        assert a.shape == b.shape and a.dtype == b.dtype == torch.float32 and torch.equal(a, b)
    a, b = ours(images=images, texts=texts, gain=0.5), theirs(images=images, texts=texts, gain=0.5)      # kwargs reach the model
    assert torch.equal(a, b) and ours.model.calls[-1][2] == theirs.model.calls[-1][2] == {"gain": 0.5}
    # what the model is asked, call by call, is the reference's too (score.py:104-106: one call per image, the image repeated)
    assert ours.model.calls == theirs.model.calls


def test_score_forward_video_gate_is_the_references(capsys):
    ours, theirs = _scorers()
    for s in (ours, theirs):
        s.model.video_mode = "neither"
    assert ours(images=["clip.MP4"], texts=["x"]) is None and theirs(images=["clip.MP4"], texts=["x"]) is None
    said = capsys.readouterr().out.strip().splitlines()
    assert len(said) == 2 and said[0] == said[1] and "video_mode" in said[0]
    for s in (ours, theirs):
        s.model.video_mode = "direct"                        # video-native: the path reaches the model untouched
    assert torch.equal(ours(images=["clip.mp4", "b.mov"], texts=["x", "y"]), theirs(images=["clip.mp4", "b.mov"], texts=["x", "y"]))
    assert ours.model.calls == theirs.model.calls and ours.model.calls[0][0] == ["clip.mp4", "clip.mp4"]


@pytest.mark.parametrize("n_samples,n_vis,n_txt,batch_size", [(7, 2, 3, 4), (4, 1, 1, 16), (5, 3, 1, 1), (9, 1, 4, 3)])
def test_batch_forward_is_the_references(n_samples, n_vis, n_txt, batch_size):
    ours, theirs = _scorers()
    dataset = [{"images": [f"s{k}_v{v}.png" for v in range(n_vis)], "texts": [f"s{k} text {t}" for t in range(n_txt)]}
               for k in range(n_samples)]
    a = ours.batch_forward(dataset, batch_size=batch_size, gain=2.0)
    b = theirs.batch_forward(dataset, batch_size=batch_size, gain=2.0)
    assert a.shape == b.shape == (n_samples, n_vis, n_txt) and a.dtype == b.dtype and torch.equal(a, b)
    # the same set of (image, text) pairs was asked of the model, each exactly once -- in fewer, larger calls
    flat = lambda calls: sorted((i, t) for im, tx, _ in calls for i, t in zip(im, tx))
    assert flat(ours.model.calls) == flat(theirs.model.calls) and len(flat(ours.model.calls)) == n_samples * n_vis * n_txt
    assert len(ours.model.calls) == -(-n_samples // batch_size) and len(theirs.model.calls) == n_samples * n_vis * n_txt
    assert all(kw == {"gain": 2.0} for _, _, kw in ours.model.calls + theirs.model.calls)


def test_batch_forward_rejects_ragged_samples_like_the_reference():
    ours, theirs = _scorers()
    dataset = [{"images": ["a.png", "b.png"], "texts": ["x"]}, {"images": ["c.png"], "texts": ["y"]}]
    with pytest.raises(AssertionError, match="Expected 2 visuals"):
        ours.batch_forward(dataset, batch_size=1)
    with pytest.raises(Exception):                           # the reference trips in DataLoader collation or on its own assert
        theirs.batch_forward(dataset, batch_size=2)


def test_image_loader_is_the_references(tmp_path):
    """models/model.py:10-14: a .npy file holds BGR pixels (cv2 layout) and is flipped to RGB, everything else goes through PIL."""
    from t2v_metrics_amd.models.model import image_loader
    ref = reference_module("models.model")
    rs = np.random.RandomState(8)
    px = rs.randint(0, 256, (9, 13, 3), dtype=np.uint8)
    np.save(tmp_path / "a.npy", px)
    Image.fromarray(px).save(tmp_path / "a.png")
    Image.fromarray(px[..., 0]).save(tmp_path / "grey.png")                      # mode L -> RGB
    Image.fromarray(np.dstack([px, px[..., :1]]), "RGBA").save(tmp_path / "alpha.png")
    Image.fromarray(px).save(tmp_path / "a.jpg", quality=90)
    for name in ("a.npy", "a.png", "grey.png", "alpha.png", "a.jpg"):
        ours, theirs = image_loader(str(tmp_path / name)), ref.image_loader(str(tmp_path / name))
        assert ours.mode == theirs.mode == "RGB" and ours.size == theirs.size
        assert np.array_equal(np.asarray(ours), np.asarray(theirs)), name
    assert np.array_equal(np.asarray(image_loader(str(tmp_path / "a.npy"))), px[..., ::-1])
