"""GenAI-Bench evaluation driver: metrics against scipy / a brute-force restatement of the
reference's pair loop, dataset wrapper and score cache on a synthetic on-disk dataset."""
import json
import os

import numpy as np
import pytest
import torch

from t2v_metrics_amd.genai_bench import (GENAI_MODELS, GenAIBenchImage, calc_pearson, kendall_tau_b,
                                         pairwise_acc_with_tie_optimization, run_genai_image_eval)


def brute_force_acc23(gold, metric):
    """tau_optimization.py:203-298 for one row, literally: flip pairs to metric ties in order of |difference|."""
    n = len(gold)
    pairs = [(abs(metric[i] - metric[j]), gold[i], gold[j], metric[i], metric[j]) for i in range(n) for j in range(i + 1, n)]

    def correct(h1, h2, m1, m2, tie):
        if tie or m1 == m2:
            return h1 == h2
        return h1 != h2 and ((h1 > h2) == (m1 > m2))

    P = len(pairs)
    thresholds = [0.0] + sorted(set(p[0] for p in pairs))
    best, best_t = -1.0, None
    for t in sorted(set(thresholds)):
        acc = sum(correct(h1, h2, m1, m2, d <= t) for d, h1, h2, m1, m2 in pairs) / P
        if acc > best:
            best, best_t = acc, t
    return best, best_t


def test_metrics_match_scipy_and_the_reference_pair_loop():
    import scipy.stats
    rng = np.random.RandomState(0)
    for n in (7, 40):
        gold = rng.randint(1, 6, size=n).astype(float)                    # human ratings tie a lot
        metric = np.round(rng.rand(n) + 0.15 * gold, 2)
        assert abs(calc_pearson(gold, metric) - 100 * scipy.stats.pearsonr(gold, metric)[0]) < 1e-9
        assert abs(kendall_tau_b(gold, metric) - scipy.stats.kendalltau(gold, metric, variant="b")[0]) < 1e-12
        acc, thr = pairwise_acc_with_tie_optimization(gold, metric)
        ref_acc, ref_thr = brute_force_acc23(gold.tolist(), metric.tolist())
        assert abs(acc - ref_acc) < 1e-12 and abs(thr - ref_thr) < 1e-12
    assert np.isnan(kendall_tau_b([1, 1, 1], [0.1, 0.2, 0.3]))
    with pytest.raises(ValueError):
        pairwise_acc_with_tie_optimization([1, 2], [1, 2], sample_rate=0.0)


def _reference_tau_optimization():
    """The reference's own module where it is on disk (this container; not the GPU box): /root/reference/tau_optimization.py."""
    import importlib.util
    path = "/root/reference/tau_optimization.py"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    spec = importlib.util.spec_from_file_location("ref_tau_optimization", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pairwise_accuracy_equals_the_references_tau_optimization():
    """The checker is the reference itself (tau_optimization.tau_optimization with TauSufficientStats.acc_23, as dataset.py:163-166
    calls it): ungrouped vectors and the grouped N x M case (dataset.py:159-161), with heavy ties in both score sets."""
    from t2v_metrics_amd.genai_bench import calc_metric
    ref = _reference_tau_optimization()
    rng = np.random.RandomState(7)
    for trial in range(25):
        if trial % 2 == 0:
            shape = (int(rng.randint(2, 30)),)
        else:
            shape = (int(rng.randint(1, 9)), int(rng.randint(2, 8)))
        gold = rng.randint(1, 6, size=shape).astype(float)
        metric = np.round(rng.rand(*shape) + 0.2 * gold, 1 if trial % 3 == 0 else 3)
        r = ref.tau_optimization(metric, gold, ref.TauSufficientStats.acc_23)
        acc, thr = calc_metric(gold, metric, "pairwise_acc_with_tie_optimization")
        assert abs(acc - r.best_tau) < 1e-12, (trial, shape, acc, r.best_tau)
        # the threshold: the reference's running float sums can order two mathematically tied thresholds either way (here the
        # counts are integers and the smallest wins), so it has to match only where the reference's maximum is unique
        taus, ths = np.array(r.taus), np.array(r.thresholds)
        if (taus > r.best_tau - 1e-9).sum() == 1:
            assert abs(thr - r.best_threshold) < 1e-12, (trial, shape, thr, r.best_threshold)
        else:
            assert abs(taus[np.argmin(np.abs(ths - thr))] - r.best_tau) < 1e-9 and np.abs(ths - thr).min() < 1e-12
    # rows without a pair (one system) do not count as rows
    acc, _ = calc_metric(np.array([[1.0], [2.0]]), np.array([[0.1], [0.2]]))
    assert np.isnan(acc)


def _make_dataset(root, n_prompts=5):
    d = os.path.join(root, "GenAI-Image-527")
    os.makedirs(d)
    rng = np.random.RandomState(1)
    meta = {f"{i:05d}": {"prompt": f"prompt number {i}", "models": {m: rng.randint(1, 6, size=3).tolist() for m in GENAI_MODELS}}
            for i in range(n_prompts)}
    json.dump(meta, open(os.path.join(d, "genai_image.json"), "w"))
    json.dump({"counting": [0, 1, 2], "spatial": [2, 3, 4]}, open(os.path.join(d, "genai_skills.json"), "w"))
    return d


class FakeScorer:
    def __init__(self):
        self.calls = 0

    def batch_forward(self, dataset, batch_size=16, **kw):
        self.calls += 1
        g = torch.Generator().manual_seed(5)
        return torch.rand(len(dataset), 1, 1, generator=g)


def test_dataset_wrapper_and_cached_driver(tmp_path):
    _make_dataset(str(tmp_path))
    ds = GenAIBenchImage(root_dir=str(tmp_path), num_prompts=527)
    assert len(ds) == 5 * len(GENAI_MODELS)
    it = ds[7]
    assert it["texts"] == ["prompt number 2"] and it["images"][0].endswith(os.path.join("SDXL_Turbo", "00002.jpeg"))
    scorer = FakeScorer()
    res = run_genai_image_eval(scorer, ds, str(tmp_path / "results"), "fake-model", num_prompts=527)
    assert set(res["alignment"]) == {"pearson", "kendall_b", "pairwise_acc"} and set(res["per_skill"]) == {"counting", "spatial"}
    assert os.path.exists(tmp_path / "results" / "fake-model_527_prompts.pt")
    res2 = run_genai_image_eval(scorer, ds, str(tmp_path / "results"), "fake-model", num_prompts=527)
    assert scorer.calls == 1 and res2["alignment"] == res["alignment"]                       # second run read the cache
    with pytest.raises(FileNotFoundError):
        GenAIBenchImage(root_dir=str(tmp_path / "nowhere"), num_prompts=1600)
    with pytest.raises(RuntimeError, match="no network"):
        GenAIBenchImage(root_dir=str(tmp_path / "nowhere"), num_prompts=1600, download=True)


def _reference_dataset_module():
    """/root/reference/dataset.py as it lies (its ``calc_metric`` imports ``tau_optimization`` by its top-level name; ``cv2`` is
    imported at the top for the video datasets and never called here)."""
    import importlib.util
    import sys
    import types
    path = "/root/reference/dataset.py"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    sys.modules.setdefault("tau_optimization", _reference_tau_optimization())
    if "cv2" not in sys.modules:
        try:
            import cv2  # noqa: F401
        except ImportError:
            sys.modules["cv2"] = types.ModuleType("cv2")
    spec = importlib.util.spec_from_file_location("ref_dataset", path)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError as e:
        pytest.skip(f"the reference's dataset.py does not import here: {e}")
    return mod


@pytest.mark.parametrize("variant", ["pairwise_acc_with_tie_optimization", "tau_with_tie_optimization", "pairwise_acc_ignore_tie",
                                     "tau_b", "tau_c"])
def test_calc_metric_equals_the_references_for_every_variant(variant):
    """dataset.py:151-188 run next to ours: 1-D (one group) and 2-D (grouped by item) inputs, ratings that tie a lot, metric scores
    with and without exact ties."""
    from t2v_metrics_amd.genai_bench import calc_metric
    ref = _reference_dataset_module()
    rng = np.random.RandomState(11)
    cases = []
    for n in (5, 17, 60):
        gold = rng.randint(1, 6, size=n).astype(float)
        cases.append((gold, np.round(rng.rand(n) + 0.2 * gold, 2)))           # metric ties present
        cases.append((gold, rng.rand(n) + 0.1 * gold))                       # none
    for rows, cols in ((4, 6), (25, 6), (3, 12)):
        gold = rng.randint(1, 6, size=(rows, cols)).astype(float)
        gold[:, 0] += 0.5                                                    # no row is constant in gold
        cases.append((gold, np.round(rng.rand(rows, cols) + 0.2 * gold, 1)))
        cases.append((gold, rng.rand(rows, cols)))
    for gold, metric in cases:
        want = ref.calc_metric(gold, metric, variant=variant)
        got = calc_metric(gold, metric, variant=variant)
        if isinstance(want, tuple):
            assert isinstance(got, tuple) and len(got) == 2
            assert got[0] == pytest.approx(float(want[0]), rel=1e-12, abs=1e-12), (variant, gold.shape)
            if variant == "pairwise_acc_ignore_tie":
                assert got[1] == want[1] == 0.0
            else:
                # thresholds: only where the reference's maximum is unique under its running float sums (see the acc23 test above)
                taus = np.asarray(sys_modules_tau().tau_optimization(
                    metric if metric.ndim == 2 else metric[None], gold if gold.ndim == 2 else gold[None],
                    getattr(sys_modules_tau().TauSufficientStats, "acc_23" if variant.startswith("pairwise") else "tau_23")).taus)
                if (np.abs(taus - taus.max()) < 1e-12).sum() == 1:
                    assert got[1] == pytest.approx(float(want[1]), rel=1e-12, abs=1e-15)
        else:
            assert got == pytest.approx(float(want), rel=1e-12, abs=1e-12), (variant, gold.shape)
    with pytest.raises(ValueError):
        calc_metric([1.0, 2.0], [1.0, 2.0], variant="tau_a")


def sys_modules_tau():
    import sys
    return sys.modules["tau_optimization"]


def test_dataset_wrapper_equals_the_references_class_on_the_same_directory(tmp_path, capsys):
    """GenAIBench_Image (dataset.py:1225-1391) instantiated on the synthetic directory next to ours: same length, items, order,
    correlation tables (whole set and per skill)."""
    ref = _reference_dataset_module()
    _make_dataset(str(tmp_path))
    ours = GenAIBenchImage(root_dir=str(tmp_path), num_prompts=527)
    theirs = ref.GenAIBench_Image(root_dir=str(tmp_path), num_prompts=527, download=False)
    assert len(ours) == len(theirs)
    for k in range(len(ours)):
        assert ours[k] == theirs[k]
    scores = torch.rand(len(ours), 1, 1, generator=torch.Generator().manual_seed(9))
    a, b = ours.evaluate_scores(scores), theirs.evaluate_scores(scores)
    assert set(a) == set(b) == {"alignment"}

    def same(x, y):
        assert set(x) == set(y) == {"pearson", "kendall_b", "pairwise_acc"}
        assert x["pearson"] == pytest.approx(float(y["pearson"]), rel=1e-12)
        assert x["kendall_b"] == pytest.approx(float(y["kendall_b"]), rel=1e-12)
        assert x["pairwise_acc"][0] == pytest.approx(float(y["pairwise_acc"][0]), rel=1e-12)
    same(a["alignment"], b["alignment"])
    pa, pb = ours.evaluate_scores_per_skill(scores), theirs.evaluate_scores_per_skill(scores)
    assert list(pa) == list(pb)
    for tag in pa:
        same(pa[tag]["alignment"], pb[tag]["alignment"])
    capsys.readouterr()
