"""Worker-process image pool (t2v_metrics_amd/imgpool.py): bytes identical to the in-process path, errors
surface in the parent, the model wires it in by default and falls back to threads for a custom loader."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from t2v_metrics_amd._imgprep import clip_preprocess_u8, image_loader
from t2v_metrics_amd.imgpool import ImagePoolError, ImageProcessPool


def _files(tmp_path, n=9):
    rng = np.random.RandomState(3)
    paths = []
    for i in range(n):
        h, w = [(64, 64), (50, 90), (120, 40), (336, 336), (400, 300)][i % 5]
        arr = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 4 == 3:
            p = tmp_path / f"im{i}.npy"
            np.save(p, arr)                                   # OpenCV-style BGR array (reference model.py:10-14)
        else:
            p = tmp_path / f"im{i}.{'png' if i % 2 else 'jpg'}"
            Image.fromarray(arr).save(p)
        paths.append(str(p))
    return paths


def _raises_of(fn):
    try:
        fn()
    except Exception as e:                       # noqa: BLE001
        return e
    raise AssertionError("did not raise")


def test_pool_bytes_equal_the_in_process_path_and_errors_surface(tmp_path):
    paths = _files(tmp_path)
    pool = ImageProcessPool(3)
    try:
        for size, pad in ((56, True), (112, False)):
            got = np.array(pool.load_u8(paths, size, pad))
            want = np.stack([clip_preprocess_u8(image_loader(p), size, pad) for p in paths])
            assert got.shape == want.shape and np.array_equal(got, want)
        assert pool.load_u8([], 56, True).shape == (0, 56, 56, 3)
        # a bad path raises what the reference's in-process loader raises (PIL's Image.open, models/model.py:10-14), not a pool-specific type
        with pytest.raises(FileNotFoundError, match="nope.png") as ei:
            pool.load_u8(paths[:2] + [str(tmp_path / "nope.png")], 56, True)
        assert not isinstance(ei.value, ImagePoolError) and ei.value.errno == 2
        junk = tmp_path / "junk.png"
        junk.write_bytes(b"not an image")
        from PIL import UnidentifiedImageError
        with pytest.raises(UnidentifiedImageError):
            pool.load_u8([str(junk)], 56, True)
        with pytest.raises(type(_raises_of(lambda: image_loader(str(junk))))):
            pool.load_u8([str(junk)], 56, True)
        # a worker whose process died is replaced, not handed the next request
        victim = pool._workers[0]
        victim.proc.kill()
        victim.proc.wait()
        for _ in range(3):                       # the dead worker is somewhere on the idle queue: at most one batch reports it
            try:
                got = np.array(pool.load_u8(paths, 56, True))
                break
            except ImagePoolError:
                continue
        assert np.array_equal(got, np.stack([clip_preprocess_u8(image_loader(p), 56, True) for p in paths]))
        assert victim not in pool._workers and len(pool._workers) == 3 and all(w.proc.poll() is None for w in pool._workers)
        assert np.array_equal(np.array(pool.load_u8(paths[:2], 56, True)), np.stack([clip_preprocess_u8(image_loader(p), 56, True) for p in paths[:2]]))
        shm = pool._shm_path
        assert os.path.exists(shm)
    finally:
        pool.close()
    assert not os.path.exists(shm) and all(w.proc.poll() is not None for w in pool._workers) or not pool._workers
    with pytest.raises(ImagePoolError):
        pool.load_u8(paths[:1], 56, True)


def test_model_uses_worker_processes_by_default_and_threads_for_a_custom_loader(tmp_path):
    import t2v_metrics_amd as t2v
    from t2v_metrics_amd.config import get_config
    from tests.test_host_api import RecordingEngine as FakeEngine, FakeTokenizer
    cfg = get_config("tiny")
    paths = _files(tmp_path, 6)
    out = {}
    for mode in ("process", "thread"):
        m = t2v.VQAScore(model="clip-flant5-xl", device="cpu", config=cfg, engine=FakeEngine(cfg), tokenizer=FakeTokenizer(cfg.t5.vocab),
                         num_workers=2, image_workers=mode).model
        assert m._use_process_pool() == (mode == "process")
        k, u8 = m._load_images_host_u8(paths)
        out[mode] = u8.clone()
        assert (m._proc_pool is not None) == (mode == "process")
        m.image_loader = lambda p: image_loader(p)          # a user hook: the workers cannot see it -> threads
        assert not m._use_process_pool()
        if m._proc_pool is not None:
            m._proc_pool.close()
    assert torch.equal(out["process"], out["thread"])
    S = cfg.vision.image
    assert out["process"].shape == (6, S, S, 3) and torch.equal(out["process"][0], torch.from_numpy(clip_preprocess_u8(image_loader(paths[0]), S, True)))


def test_a_single_image_is_decoded_on_the_calling_thread_with_the_pools_bytes(tmp_path):
    """The reference's per-pair loops (score.py:143-153) hand the model one image per call: no worker process is started or waited for, and the staged
    bytes are the pooled path's."""
    import t2v_metrics_amd as t2v
    from t2v_metrics_amd.config import get_config
    from tests.test_host_api import RecordingEngine as FakeEngine, FakeTokenizer
    cfg = get_config("tiny")
    paths = _files(tmp_path, 3)
    m = t2v.VQAScore(model="clip-flant5-xl", device="cpu", config=cfg, engine=FakeEngine(cfg), tokenizer=FakeTokenizer(cfg.t5.vocab), num_workers=2).model
    _, one = m._load_images_host_u8(paths[1:2])
    one = one.clone()
    assert m._proc_pool is None, "one image must not start the worker processes"
    _, three = m._load_images_host_u8(paths)
    assert m._proc_pool is not None and torch.equal(three[1], one[0])
    with pytest.raises(FileNotFoundError):
        m._load_images_host_u8([str(tmp_path / "missing.png")])
    m._proc_pool.close()

