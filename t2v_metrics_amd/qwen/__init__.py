"""Qwen2.5-VL VQAScore (BASELINE.json configs[4]): architecture constants (config.py), weight inventory and
checkpoint loading (weights.py), the integer layout work of a pass -- window permutation, rotary tables, placeholder slots (layout.py) --
and the ctypes binding of the HIP engine behind include/vqs_qwen.h (engine.py: vision tower, prefill with the precise tail, KV-cached
decode; bf16 forms or the range-safe fp16 forms).  The product path: ``VQAScore(model="qwen2.5-vl-7b")`` ->
models/vqascore_models/qwen25vl_model.py::Qwen25VLModel -> QwenEngine.  The CPU oracle (oracle/qwen25vl_oracle.py) shares config and layout."""
from .config import Qwen25VLConfig, QwenTextConfig, QwenVisionConfig, get_qwen_config  # noqa: F401
