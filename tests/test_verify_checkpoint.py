"""tools/verify_checkpoint.py is known to RUN: a synthetic checkpoint directory in the v3.0 key layout (safetensors shards, the
CLIP tower in its own directory), a SentencePiece model trained here with T5's special ids, four image files -- the tool must load them the
way the reference's loader does (mm_utils.py:198-241), push the 4 x 4 grid through the fp32 and the bf16 HF legs, apply the reference's
smoke-test assertions (test.py:110-112, 138-139), print BASELINE.md section 3's table and, on this GPU-less box, report the HIP leg as NOT RUN
instead of falling back to anything."""
import io
import os
import types

import numpy as np
import pytest
import torch

from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import make_seeded_weights


class _SPTokenizer:
    def __init__(self, tmp_path, corpus):
        spm = pytest.importorskip("sentencepiece")
        src = tmp_path / "corpus.txt"
        src.write_text("\n".join(corpus * 20))
        spm.SentencePieceTrainer.train(input=str(src), model_prefix=str(tmp_path / "sp"), vocab_size=120, model_type="unigram",
                                       pad_id=0, eos_id=1, unk_id=2, bos_id=-1, hard_vocab_limit=False, minloglevel=2)
        self.sp = spm.SentencePieceProcessor(model_file=str(tmp_path / "sp.model"))

    def __call__(self, text):
        return types.SimpleNamespace(input_ids=self.sp.encode(text) + [1])


def test_verify_checkpoint_runs_on_a_synthetic_checkpoint(tmp_path):
    import sys
    from PIL import Image
    from safetensors.torch import save_file
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import verify_checkpoint as vc
    from t2v_metrics_amd.models.vqascore_models.clip_t5_model import default_answer_template, default_question_template, format_question
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=5, device="cpu", lm_head_gain=2.0)
    ck, tower = tmp_path / "clip-flant5-tiny", tmp_path / "clip-vit"
    ck.mkdir(); tower.mkdir()
    main = {("encoder.mm_projector." + k[len("mm_projector."):] if k.startswith("mm_projector.") else k): v.float().contiguous()
            for k, v in w.items() if not k.startswith("vision.")}                       # fp32 on disk: cast to bf16 on load (mm_utils.py:228)
    items = sorted(main.items())
    save_file(dict(items[: len(items) // 2]), str(ck / "model-00001-of-00002.safetensors"))
    save_file(dict(items[len(items) // 2:]), str(ck / "model-00002-of-00002.safetensors"))
    save_file({"vision_model." + k[len("vision."):]: v.contiguous() for k, v in w.items() if k.startswith("vision.")}, str(tower / "model.safetensors"))
    texts = list(vc.CAPTIONS)
    tok = _SPTokenizer(tmp_path, [format_question(default_question_template.format(t)).replace("<image>", " ") for t in texts] + [default_answer_template])
    rng = np.random.RandomState(2)
    images = []
    for i, (h, wd) in enumerate([(64, 64), (40, 90), (120, 50), (336, 336)]):
        p = tmp_path / f"im{i}.{'png' if i % 2 else 'jpg'}"
        Image.fromarray(rng.randint(0, 256, (h, wd, 3), dtype=np.uint8)).save(p)
        images.append(str(p))
    buf = io.StringIO()
    rep = vc.run(str(ck), str(tower), "clip-flant5-xl", images, texts, device="cpu", config=cfg, tokenizer=tok, out=buf)
    text = buf.getvalue()
    assert set(rep["legs"]) == {"(i)", "(ii)", "(iii)"} and "not_run" in rep["legs"]["(iii)"] and "no CPU route" in rep["legs"]["(iii)"]["not_run"]
    assert rep["verdict"].startswith("INCOMPLETE") and "(iii) NOT RUN" in text and "## |delta log P(Yes)|" in text
    for leg in ("(i)", "(ii)"):
        sc = torch.tensor(rep["legs"][leg]["scores"])
        assert sc.shape == (4, 4) and bool(((sc >= 0) & (sc <= 1)).all())
    d = rep["dlogp"]["(ii) vs (i)"]
    assert 0 < d["mean"] <= d["max"] < 0.5                                     # the reference's own bf16 path against fp32 on the same weights
    assert rep["answer_template_ids"][-1] == 1 and len(rep["answer_template_ids"]) >= 2      # "<answer> </s>"
    # the fp32 leg IS the oracle's arithmetic on the loaded weights: the same grid through oracle/clip_t5_oracle.py via the engine double of the host tests
    from tests.test_host_api import OracleEngine
    import t2v_metrics_amd as t2v
    from t2v_metrics_amd.weights import load_checkpoint_weights, read_checkpoint_dir
    w2 = load_checkpoint_weights(cfg, read_checkpoint_dir(str(ck)), "cpu", vision_state_dict=read_checkpoint_dir(str(tower)))
    assert all(torch.equal(w2[k], w[k]) for k in w)
    s = t2v.VQAScore(model="clip-flant5-xl", device="cpu", config=cfg, engine=OracleEngine(cfg, w2), tokenizer=tok, cache_dir=str(tmp_path))
    grid = s(images=images, texts=texts)
    assert torch.allclose(grid.double(), torch.tensor(rep["legs"]["(i)"]["scores"]).double(), atol=2e-5, rtol=1e-4)
