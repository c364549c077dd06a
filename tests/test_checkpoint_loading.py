"""Checkpoint -> inventory mapping on synthetic HF directories (no real checkpoint is reachable offline): the v3.0 key
prefixes of the zhiqiulin/clip-flant5-* state dicts, safetensors and pytorch_model-*.bin shards, a CLIP tower that
lives in its own directory (the reference loads it separately, mm_utils.py:236-237), and the failure modes."""
import os

import pytest
import torch

from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import (canonical_name, load_checkpoint_weights, make_seeded_weights, read_checkpoint_dir,
                                     weight_specs)


def _v30_key(name: str) -> str:
    """inventory name -> key as the v3.0 CLIP-FlanT5 state dict spells it ([RECALLED] layout, weights.py)."""
    if name.startswith("vision."):
        return "encoder.vision_tower.vision_tower.vision_model." + name[len("vision."):]
    if name.startswith("mm_projector."):
        return "encoder.mm_projector." + name[len("mm_projector."):]
    return name


def test_prefix_mapping_covers_the_inventory():
    cfg = get_config("tiny")
    names = [n for n, _, _ in weight_specs(cfg)]
    assert [canonical_name(_v30_key(n)) for n in names] == names
    assert canonical_name("vision_model.encoder.layers.0.mlp.fc1.weight") == "vision.encoder.layers.0.mlp.fc1.weight"
    assert canonical_name("vision_tower.vision_model.pre_layrnorm.bias") == "vision.pre_layrnorm.bias"
    assert canonical_name("lm_head.weight") == "lm_head.weight"


@pytest.mark.parametrize("fmt", ["safetensors", "bin", "mixed"])
def test_sharded_directory_round_trips(tmp_path, fmt):
    from safetensors.torch import save_file
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=4, device="cpu")
    items = [(_v30_key(k), v.contiguous()) for k, v in sorted(w.items())]
    a, b = dict(items[: len(items) // 2]), dict(items[len(items) // 2:])
    if fmt in ("safetensors", "mixed"):
        save_file(a, str(tmp_path / "model-00001-of-00002.safetensors"))
    else:
        torch.save(a, str(tmp_path / "pytorch_model-00001-of-00002.bin"))
    if fmt == "safetensors":
        save_file(b, str(tmp_path / "model-00002-of-00002.safetensors"))
    else:
        torch.save(b, str(tmp_path / "pytorch_model-00002-of-00002.bin"))
    (tmp_path / "config.json").write_text("{}")              # other files are ignored
    got = load_checkpoint_weights(cfg, read_checkpoint_dir(str(tmp_path)), "cpu")
    assert set(got) == set(w)
    assert all(torch.equal(got[k], w[k]) and got[k].dtype == torch.bfloat16 and got[k].is_contiguous() for k in w)


def test_separate_vision_tower_directory_and_casting(tmp_path):
    """T5 + projector in the model directory (fp32 on disk -> cast to bf16 as mm_utils.py:228 does), CLIP tower in
    openai/clip-vit-large-patch14-336 layout (vision_model.*) in its own directory."""
    from safetensors.torch import save_file
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=6, device="cpu")
    main = {_v30_key(k): v.float().contiguous() for k, v in w.items() if not k.startswith("vision.")}
    tower = {"vision_model." + k[len("vision."):]: v.contiguous() for k, v in w.items() if k.startswith("vision.")}
    tower["text_model.embeddings.token_embedding.weight"] = torch.zeros(4, 4)       # CLIP's text half is ignored
    tower["vision_model.post_layernorm.weight"] = torch.ones(cfg.vision.hidden)     # unused by the path, ignored
    os.makedirs(tmp_path / "m"); os.makedirs(tmp_path / "v")
    save_file(main, str(tmp_path / "m" / "model.safetensors"))
    torch.save(tower, str(tmp_path / "v" / "pytorch_model.bin"))
    sd, vt = read_checkpoint_dir(str(tmp_path / "m")), read_checkpoint_dir(str(tmp_path / "v"))
    with pytest.raises(KeyError, match="vision"):
        load_checkpoint_weights(cfg, sd, "cpu")
    got = load_checkpoint_weights(cfg, sd, "cpu", vision_state_dict=vt)
    assert set(got) == set(w) and all(torch.equal(got[k], w[k]) for k in w)
    # through the model wrapper: checkpoint=<dir>, vision_tower=<dir>; stop at the engine (no GPU here)
    from t2v_metrics_amd.models.vqascore_models.clip_t5_model import CLIPT5Model
    m = CLIPT5Model.__new__(CLIPT5Model)
    m._checkpoint, m._vision_tower_dir, m.cache_dir, m.model_name = str(tmp_path / "m"), str(tmp_path / "v"), str(tmp_path), "clip-flant5-xl"
    got2 = load_checkpoint_weights(cfg, m._read_checkpoint(), "cpu", vision_state_dict=m._read_vision_tower())
    assert all(torch.equal(got2[k], w[k]) for k in w)
    m._vision_tower_dir = str(tmp_path / "nope")
    with pytest.raises(FileNotFoundError):
        m._read_vision_tower()


def test_failure_modes(tmp_path):
    cfg = get_config("tiny")
    w = make_seeded_weights(cfg, seed=4, device="cpu")
    with pytest.raises(FileNotFoundError):
        read_checkpoint_dir(str(tmp_path))                     # empty directory
    sd = {_v30_key(k): v for k, v in w.items()}
    del sd["lm_head.weight"]
    with pytest.raises(KeyError, match="lm_head.weight"):
        load_checkpoint_weights(cfg, sd, "cpu")
    sd = {_v30_key(k): v for k, v in w.items()}
    sd["shared.weight"] = sd["shared.weight"][:-1]
    with pytest.raises(ValueError, match="shared.weight"):
        load_checkpoint_weights(cfg, sd, "cpu")
