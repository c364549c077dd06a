"""Model registry (interface of /root/reference/t2v_metrics/models/vqascore_models/__init__.py:14-54).

Registered: the model family of the MI355X hot path (CLIP-FlanT5) and the next model row, Qwen2.5-VL-7B (SURVEY.md §8f
rank 2).  The reference's other families (remote APIs, Gemma/PaliGemma/InternVL wrappers, other Qwen generations) are
out of scope and are deliberately NOT silently redirected to another backend.  Wrappers are imported lazily so that importing the package needs nothing but torch."""
from ...constants import HF_CACHE_DIR
from .clip_t5_model import CLIP_T5_MODELS, CLIPT5Model
from .qwen25vl_model import QWEN25_VL_MODELS, Qwen25VLModel

ALL_VQA_MODELS = [
    CLIP_T5_MODELS,
    QWEN25_VL_MODELS,
]


def list_all_vqascore_models():
    return [model for models in ALL_VQA_MODELS for model in models]


def get_vqascore_model(model_name, device='cuda', cache_dir=HF_CACHE_DIR, **kwargs):
    assert model_name in list_all_vqascore_models()
    if model_name in CLIP_T5_MODELS:
        return CLIPT5Model(model_name, device=device, cache_dir=cache_dir, **kwargs)
    elif model_name in QWEN25_VL_MODELS:
        return Qwen25VLModel(model_name, device=device, cache_dir=cache_dir, **kwargs)
    else:
        raise NotImplementedError()
