"""Model registry (interface of /root/reference/t2v_metrics/models/vqascore_models/__init__.py:14-54).

Only the model family of the MI355X hot path is registered; the reference's other families (remote APIs,
Qwen/Gemma/PaliGemma wrappers) are out of scope and are deliberately NOT silently redirected
to another backend.  Wrappers are imported lazily so that importing the package needs nothing but torch."""
from ...constants import HF_CACHE_DIR
from .clip_t5_model import CLIP_T5_MODELS, CLIPT5Model

ALL_VQA_MODELS = [
    CLIP_T5_MODELS,
]


def list_all_vqascore_models():
    return [model for models in ALL_VQA_MODELS for model in models]


def get_vqascore_model(model_name, device='cuda', cache_dir=HF_CACHE_DIR, **kwargs):
    assert model_name in list_all_vqascore_models()
    if model_name in CLIP_T5_MODELS:
        return CLIPT5Model(model_name, device=device, cache_dir=cache_dir, **kwargs)
    else:
        raise NotImplementedError()
