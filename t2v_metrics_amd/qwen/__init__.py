"""Qwen2.5-VL VQAScore (BASELINE.json configs[4]) -- groundwork: architecture constants, weight
inventory and the layout helpers shared by the CPU oracle (oracle/qwen25vl_oracle.py) and the future HIP path.
There is NO product path for this model yet: nothing here is reachable from ``VQAScore``."""
from .config import Qwen25VLConfig, QwenTextConfig, QwenVisionConfig, get_qwen_config  # noqa: F401
