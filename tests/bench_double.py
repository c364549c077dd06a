"""Test double for bench.py's harness self-test (tests/test_bench_harness.py): stands in for VqsEngine on a box without
GPUs so that the launcher / sharding / gather / JSON logic of bench.py can be exercised.  Scores are a deterministic
function of the prompt ids only -- NOT a model; bench.py labels any line produced with it as not-a-measurement."""
import torch


class DoubleEngine:
    def __init__(self, cfg):
        self.cfg = cfg

    def encode_images(self, pixels):
        return pixels.reshape(pixels.shape[0], -1)[:, :4].float()

    def score(self, feats, img_index, ids, labels):
        h = (ids.long().clamp(min=0) * torch.arange(1, ids.shape[1] + 1)).sum(1) % 9973
        sc = (h.float() + 1.0) / 9974.0
        lp = torch.log(sc)[:, None].expand(-1, labels.shape[1]).contiguous()
        return lp, sc

    def profile(self, on):
        pass

    def profile_read(self, reset=True):
        return 0, 0.0, 0.0

    def profile_bytes(self):
        return 0.0

    def profile_report(self):
        return ""
