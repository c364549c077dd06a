"""N > 1 path on CPU: world_size-2 gloo processes shard a dataset, score with an engine double, all_gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from t2v_metrics_amd import sharding


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 100, 12500 * 8 + 3):
        for ws in (1, 2, 3, 8):
            blocks = [sharding.shard_range(n, r, ws) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, ws, port, tmp, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        # raw gather
        n = 11
        lo, hi = sharding.shard_range(n)
        local = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3) + rank * 0.0
        full = sharding.gather_rows(local, n)
        assert torch.equal(full, torch.arange(n, dtype=torch.float32)[:, None].repeat(1, 3))
        # the dataset loop under sharding
        from tests.test_host_api import FakeTokenizer, RecordingEngine
        import t2v_metrics_amd as t2v
        from t2v_metrics_amd.config import get_config
        cfg = get_config("tiny")
        eng = RecordingEngine(cfg)
        s = t2v.VQAScore(model="clip-flant5-xl", device="cpu", cache_dir=os.path.join(tmp, f"c{rank}"), config=cfg,
                         engine=eng, tokenizer=FakeTokenizer(cfg.t5.vocab), distributed=True)
        imgs = sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(".png"))
        dataset = [{"images": [imgs[k % len(imgs)]], "texts": [f"caption number {k}", f"other {k}"]} for k in range(7)]
        out = s.batch_forward(dataset, batch_size=3)
        assert out.shape == (7, 1, 2)
        n_scored = sum(c[0][0] for c in eng.score_calls)
        lo, hi = sharding.shard_range(7)
        assert n_scored == (hi - lo) * 2              # each rank scored only its own block
        if rank == 0:
            np.save(out_path, out.numpy())
        # the M x N grid call under sharding (by IMAGE -- each rank encodes only its own images, one gather of rows)
        for m_img, tag in ((3, "grid3"), (1, "grid1")):           # 1 image on 2 ranks: rank 1 owns nothing and still joins the gather
            eng.encode_calls.clear(); eng.score_calls.clear()
            texts = [f"text {j}" for j in range(4)]
            grid = s(images=imgs[:m_img], texts=texts)
            assert grid.shape == (m_img, 4)
            lo, hi = sharding.shard_range(m_img)
            assert sum(c[0] for c in eng.encode_calls) == hi - lo, (eng.encode_calls, lo, hi)      # this rank's engine saw only its images
            assert sum(c[0][0] for c in eng.score_calls) == (hi - lo) * 4
            np.save(out_path.replace(".npy", f"_{tag}_r{rank}.npy"), grid.numpy())
        whole = s(images=imgs, texts=["a", "b"], shard=False)     # opt-out: the whole grid on the calling rank, no collective
        assert whole.shape == (3, 2) and sum(c[0] for c in eng.encode_calls) >= 3
        # the sharded calls are collectives -- ranks that disagree on the inputs get a ValueError (all of them), not a deadlock
        # or rows mixed from different inputs
        import pytest as _pytest
        with _pytest.raises(ValueError, match="do not agree on the inputs"):
            s(images=imgs[:2], texts=[f"rank {rank} asks something else"])
        with _pytest.raises(ValueError, match="do not agree on the inputs"):
            s.batch_forward(dataset[: 5 + rank], batch_size=3)
        # ... and the DEFAULT scorer (distributed=False: the reference's semantics) never enters a collective: a rank-0-only call returns
        plain = t2v.VQAScore(model="clip-flant5-xl", device="cpu", cache_dir=os.path.join(tmp, f"p{rank}"), config=cfg,
                             engine=RecordingEngine(cfg), tokenizer=FakeTokenizer(cfg.t5.vocab))
        if rank == 0:
            assert plain(images=imgs[:2], texts=["only rank 0 calls"]).shape == (2, 1)
            assert plain.batch_forward(dataset[:3], batch_size=2).shape == (3, 1, 2)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_batch_forward_sharded_over_two_gloo_ranks(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(1)
    for i in range(3):
        Image.fromarray(rng.randint(0, 256, (60, 60, 3), dtype=np.uint8)).save(tmp_path / f"im{i}.png")
    out_path = str(tmp_path / "sharded.npy")
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), out_path), nprocs=2, join=True)
    sharded = np.load(out_path)
    # single-process reference
    from tests.test_host_api import FakeTokenizer, RecordingEngine
    import t2v_metrics_amd as t2v
    from t2v_metrics_amd.config import get_config
    cfg = get_config("tiny")
    s = t2v.VQAScore(model="clip-flant5-xl", device="cpu", cache_dir=str(tmp_path / "c"), config=cfg,
                     engine=RecordingEngine(cfg), tokenizer=FakeTokenizer(cfg.t5.vocab))
    imgs = sorted(str(tmp_path / f) for f in os.listdir(tmp_path) if f.endswith(".png"))
    dataset = [{"images": [imgs[k % len(imgs)]], "texts": [f"caption number {k}", f"other {k}"]} for k in range(7)]
    single = s.batch_forward(dataset, batch_size=3).numpy()
    assert np.array_equal(sharded, single)
    for m_img, tag in ((3, "grid3"), (1, "grid1")):
        want = s(images=imgs[:m_img], texts=[f"text {j}" for j in range(4)]).numpy()
        for r in range(2):                                        # every rank holds the whole gathered grid, equal to the single-process one
            assert np.array_equal(np.load(out_path.replace(".npy", f"_{tag}_r{r}.npy")), want), (tag, r)
