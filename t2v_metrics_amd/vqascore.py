"""``VQAScore`` -- the user-facing scorer class.

Interface of /root/reference/t2v_metrics/vqascore.py:9-23: a ``Score`` whose plugin comes from the VQAScore model registry.
``VQAScore(model='clip-flant5-xxl', device='cuda', cache_dir=..., **kw)``; extra keyword arguments reach the plugin's
constructor unchanged (this build uses them for ``weights=``, ``tokenizer=``, ``checkpoint=``, ``config=``, ``engine=``).
"""
from typing import List

from . import constants
from .models import vqascore_models as _registry
from .models.vqascore_models import list_all_vqascore_models  # noqa: F401  (re-exported: t2v_metrics_amd/__init__.py)
from .score import Score


class VQAScore(Score):
    """P(answer | image, question) scorers: CLIP-FlanT5 (XL / XXL) and Qwen2.5-VL-7B on the MI355X engine."""

    def prepare_scoremodel(self, model='clip-flant5-xxl', device='cuda', cache_dir=constants.HF_CACHE_DIR, **kwargs):
        return _registry.get_vqascore_model(model, device=device, cache_dir=cache_dir, **kwargs)

    def list_all_models(self) -> List[str]:
        return _registry.list_all_vqascore_models()
