"""t2v_metrics_amd: MI355X-native VQAScore (CLIP-FlanT5) behind the t2v_metrics scoring API.

Drop-in surface (same names as /root/reference/t2v_metrics/__init__.py:23-33):
    import t2v_metrics_amd as t2v_metrics
    scorer = t2v_metrics.VQAScore(model='clip-flant5-xxl')
    scores = scorer(images=[...], texts=[...])          # fp32 [M, N]

Unlike the reference, importing the package does not require ffmpeg (the import-time gate at
/root/reference/t2v_metrics/__init__.py:10-20 only serves the video path, which is out of scope here).
"""
from .constants import HF_CACHE_DIR
from .vqascore import VQAScore, list_all_vqascore_models


def list_all_models():
    return list_all_vqascore_models()


def get_score_model(model='clip-flant5-xxl', device='cuda', cache_dir=HF_CACHE_DIR, **kwargs):
    if model in list_all_vqascore_models():
        return VQAScore(model, device=device, cache_dir=cache_dir, **kwargs)
    else:
        raise NotImplementedError()
