"""Constants of the scoring API (same names and values as /root/reference/t2v_metrics/constants.py:1-8)."""
HF_CACHE_DIR = "./hf_cache/"

# CLIP-FlanT5 prompt recipe
CONTEXT_LEN = 2048
SYSTEM_MSG = ("A chat between a curious user and an artificial intelligence assistant. "
              "The assistant gives helpful, detailed, and polite answers to the user's questions.")
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
