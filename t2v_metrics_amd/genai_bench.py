"""GenAI-Bench (image) evaluation driver -- SURVEY.md §8f rank 3: the dataset wrapper, the correlation metrics and the
score cache the reference uses to produce its published quality table, wired to this package's ``batch_forward``.

Restates, in vectorised numpy:
  * ``GenAIBench_Image``            /root/reference/dataset.py:1225-1391 (item layout, human-rating averaging, per-skill tables)
  * ``calc_pearson`` / tau-b / pairwise accuracy with tie optimisation   dataset.py:14-188 and tau_optimization.py:136-298
    (Deutsch et al. 2023, "Ties Matter", arXiv:2305.14324: sort all pairs by |metric difference|, turn them into metric ties
    one threshold at a time, keep the threshold with the best accuracy)
  * the driver with its ``<result_dir>/<model>_<n>_prompts.pt`` cache   /root/reference/genai_image_eval.py:109-168
No network here: the dataset must already be on disk (``download=True`` raises instead of calling wget).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

GENAI_MODELS = ['DALLE_3', 'SDXL_Turbo', 'DeepFloyd_I_XL_v1', 'Midjourney_6', 'SDXL_2_1', 'SDXL_Base']   # dataset.py:1235


# ------------------------------------------------------------------------------------------------ metrics
def calc_pearson(metric1_scores, metric2_scores) -> float:
    """100 x Pearson r (dataset.py:14-16)."""
    return float(100 * np.corrcoef(np.asarray(metric1_scores, dtype=np.float64), np.asarray(metric2_scores, dtype=np.float64))[0, 1])


def _pair_arrays(gold: np.ndarray, metric: np.ndarray):
    i, j = np.triu_indices(len(gold), k=1)
    dg, dm = gold[i] - gold[j], metric[i] - metric[j]
    return dg, dm


def kendall_tau_b(gold_scores, metric_scores) -> float:
    """Kendall tau-b (KendallVariants variant 'b', dataset.py:69-148): (C - D) / sqrt((P - T_x)(P - T_y))."""
    g, m = np.asarray(gold_scores, dtype=np.float64), np.asarray(metric_scores, dtype=np.float64)
    dg, dm = _pair_arrays(g, m)
    con = int(((dg > 0) & (dm > 0) | (dg < 0) & (dm < 0)).sum())
    dis = int(((dg > 0) & (dm < 0) | (dg < 0) & (dm > 0)).sum())
    tot = dg.size
    xtie, ytie = int((dm == 0).sum()), int((dg == 0).sum())
    if xtie == tot or ytie == tot:
        return float("nan")
    return float((con - dis) / np.sqrt(tot - xtie) / np.sqrt(tot - ytie))


def tau_with_tie_optimization(gold_scores, metric_scores, sample_rate: float = 1.0,
                              rng: Optional[np.random.RandomState] = None) -> Tuple[float, float]:
    """tau23 = (concordant + tied-in-both - discordant - tied-in-one) / pairs under the same threshold search (calc_metric variant
    "tau_with_tie_optimization", dataset.py:171-174; TauSufficientStats.tau_23, tau_optimization.py:60-67)."""
    return _tie_search(gold_scores, metric_scores, sample_rate, rng, True)


def pairwise_acc_with_tie_optimization(gold_scores, metric_scores, sample_rate: float = 1.0,
                                       rng: Optional[np.random.RandomState] = None) -> Tuple[float, float]:
    """acc23 = (concordant + tied-in-both) / pairs, maximised over the tie threshold (calc_metric's default variant,
    dataset.py:163-166; TauSufficientStats.acc_23, tau_optimization.py:69-70)."""
    return _tie_search(gold_scores, metric_scores, sample_rate, rng, False)


def _tie_search(gold_scores, metric_scores, sample_rate: float = 1.0, rng: Optional[np.random.RandomState] = None,
                signed: bool = False) -> Tuple[float, float]:
    """The threshold search of tau_optimization.py:203-298 for a statistic that is a sum over pairs: acc23 (a pair scores 1 or 0) or,
    with `signed`, tau23 (+1 or -1).  The tie threshold epsilon on the metric differences is chosen to maximise the statistic.
    1-D inputs: one group.  2-D inputs [N items, M systems] (dataset.py:159-161 "group by item"): pairs are formed INSIDE each
    row, ONE global threshold is searched, and the statistic is the mean over rows of the row's accuracy -- each pair weighs
    1 / (rows x pairs of its row), which is what tau_optimization's per-row running sums add up to.
    Candidate thresholds: 0 and every distinct |metric difference|; a pair with |dm| <= epsilon counts as correct iff the gold
    scores tie.  Returns (best value, best threshold), the smallest threshold on ties of the maximum (np.nanargmax).
    sample_rate < 1 subsamples pairs (approximate; the reference draws its sample pair by pair from np.random)."""
    if sample_rate <= 0 or sample_rate > 1:
        raise ValueError(f"`sample_rate` must be in the range (0, 1]. Found {sample_rate}")
    g, m = np.asarray(gold_scores, dtype=np.float64), np.asarray(metric_scores, dtype=np.float64)
    if g.shape != m.shape:
        raise ValueError('Human and metric scores must have the same shape.')
    if g.ndim == 1:
        g, m = g[None], m[None]
    assert g.ndim == 2
    i, j = np.triu_indices(g.shape[1], k=1)
    dg, dm = (g[:, i] - g[:, j]), (m[:, i] - m[:, j])                       # [rows, pairs per row]
    keep = np.ones(dg.shape, dtype=bool)
    if sample_rate < 1.0:
        keep = (rng or np.random).random_sample(dg.shape) <= sample_rate
    per_row = keep.sum(1)
    rows = int((per_row > 0).sum())
    if rows == 0:
        return float("nan"), 0.0
    if sample_rate == 1.0:
        # every row has the same number of pairs: integer counts, exact ties between thresholds (smallest threshold wins, as the
        # reference's nanargmax does on its list), one division at the end
        wgt, scale = np.ones(dg.shape, dtype=np.int64), 1.0 / (rows * int(per_row[0]))
    else:
        wgt, scale = np.where(per_row > 0, 1.0 / np.maximum(per_row, 1), 0.0)[:, None] / rows * keep, 1.0   # a pair's share of the row-averaged accuracy
    dg, dm, wgt = dg[keep], dm[keep], wgt[keep]
    adm = np.abs(dm)
    order = np.argsort(adm, kind="stable")
    adm, dg, dm, wgt = adm[order], dg[order], dm[order], wgt[order]
    gold_tie = (dg == 0) * wgt
    conc = (((dg > 0) & (dm > 0)) | ((dg < 0) & (dm < 0))) * wgt
    if signed:      # tau23: a pair that is not counted +1 counts -1 (discordant, or tied in exactly one of the two score lists)
        gold_tie, conc = 2 * gold_tie - wgt, 2 * conc - wgt
    # accuracy when the first k pairs (smallest differences) are metric ties: gold ties among them + concordant among the rest
    tie_prefix = np.concatenate([[0], np.cumsum(gold_tie)])
    conc_suffix = np.concatenate([np.cumsum(conc[::-1])[::-1], [0]])
    acc = tie_prefix + conc_suffix
    # threshold d <-> k = number of pairs with |dm| <= d
    thresholds = np.concatenate([[0.0], adm])
    ks = np.searchsorted(adm, thresholds, side="right")
    vals = acc[ks]
    uniq, first = np.unique(thresholds, return_index=True)
    best = int(np.argmax(vals[first]))
    return float(vals[first][best] * scale), float(uniq[best])


def pairwise_acc_ignore_tie(gold_scores, metric_scores) -> Tuple[float, float]:
    """concordant / (pairs - pairs tied only in the gold scores) with no threshold introduced (calc_metric variant
    "pairwise_acc_ignore_tie", dataset.py:167-170 = ``result.taus[0], result.thresholds[0]``; TauSufficientStats.acc_ignore_tie,
    tau_optimization.py:72-76), mean over rows.  A row whose every pair is tied only in the gold scores has no defined value (the
    reference stops in a debugger there): ValueError."""
    g, m = np.asarray(gold_scores, dtype=np.float64), np.asarray(metric_scores, dtype=np.float64)
    if g.ndim == 1:
        g, m = g[None], m[None]
    i, j = np.triu_indices(g.shape[1], k=1)
    dg, dm = g[:, i] - g[:, j], m[:, i] - m[:, j]
    con = (((dg > 0) & (dm > 0)) | ((dg < 0) & (dm < 0))).sum(1)
    den = dg.shape[1] - ((dg == 0) & (dm != 0)).sum(1)
    if (den == 0).any():
        raise ValueError("a row in which every pair is tied in the gold scores only has no accuracy")
    return float(np.mean(con / den)), 0.0


def kendall_tau_c(gold_scores, metric_scores) -> float:
    """Kendall tau-c (KendallVariants variant 'c', dataset.py:136-138): 2 (C - D) / (n^2 (m - 1) / m), m = the smaller number of
    distinct values of the two lists."""
    g, m = np.asarray(gold_scores, dtype=np.float64), np.asarray(metric_scores, dtype=np.float64)
    dg, dm = _pair_arrays(g, m)
    con = int(((dg > 0) & (dm > 0) | (dg < 0) & (dm < 0)).sum())
    dis = int(((dg > 0) & (dm < 0) | (dg < 0) & (dm > 0)).sum())
    tot = dg.size
    if int((dm == 0).sum()) == tot or int((dg == 0).sum()) == tot:
        return float("nan")
    classes = min(len(set(m.tolist())), len(set(g.tolist())))
    return float(2 * (con - dis) / (g.size ** 2 * (classes - 1) / classes))


def calc_metric(gold_scores, metric_scores, variant: str = "pairwise_acc_with_tie_optimization", sample_rate: float = 1.0):
    """dataset.py:151-188, all five variants: "pairwise_acc_with_tie_optimization" / "tau_with_tie_optimization" -> (best value, best
    threshold); "pairwise_acc_ignore_tie" -> (value, 0.0); "tau_b" / "tau_c" -> mean over rows, NaN rows skipped.  1-D = one group,
    2-D = grouped by item (last dimension = systems)."""
    g, m = np.asarray(gold_scores), np.asarray(metric_scores)
    assert g.shape == m.shape
    if variant == "pairwise_acc_with_tie_optimization":
        return pairwise_acc_with_tie_optimization(g, m, sample_rate)
    if variant == "tau_with_tie_optimization":
        return tau_with_tie_optimization(g, m, sample_rate)
    if variant == "pairwise_acc_ignore_tie":
        return pairwise_acc_ignore_tie(g, m)
    if variant in ("tau_b", "tau_c"):
        if g.ndim == 1:
            g, m = g[None], m[None]
        fn = kendall_tau_b if variant == "tau_b" else kendall_tau_c
        return float(np.nanmean(np.array([fn(a, b) for a, b in zip(g, m)])))
    raise ValueError(f"unsupported variant {variant!r}")


# ------------------------------------------------------------------------------------------------ dataset
class GenAIBenchImage:
    """GenAI-Bench with 527 / 1600 prompts x 6 generators (dataset.py:1225-1391).  Items are
    ``{"images": [path], "texts": [prompt]}`` as ``Score.batch_forward`` expects."""

    def __init__(self, root_dir: str = "./", download: bool = False, num_prompts: int = 1600):
        assert num_prompts in [527, 1600], "Invalid 'num_prompts' value. It must be one of [527, 1600]"
        self.root_dir = os.path.join(root_dir, f'GenAI-Image-{num_prompts}')
        self.models = list(GENAI_MODELS)
        meta = os.path.join(self.root_dir, "genai_image.json")
        if not os.path.exists(meta):
            if download:
                raise RuntimeError("no network in this environment: fetch GenAI-Bench "
                                   "(huggingface.co/datasets/BaiqiL/GenAI-Bench-1600 or zhiqiulin/GenAI-Bench-527) into " + self.root_dir)
            raise FileNotFoundError(meta)
        self.dataset = json.load(open(meta, 'r'))
        self.images: List[Dict] = []
        self.prompt_to_images: Dict[str, List[int]] = {}
        for model in self.models:
            for prompt_idx in self.dataset:
                if model not in self.dataset[prompt_idx]['models']:
                    continue
                self.images.append({'prompt_idx': prompt_idx, 'prompt': self.dataset[prompt_idx]['prompt'], 'model': model,
                                    'image': os.path.join(self.root_dir, model, f"{prompt_idx}.jpeg"),
                                    'human_alignment': self.dataset[prompt_idx]['models'][model]})
                self.prompt_to_images.setdefault(prompt_idx, []).append(len(self.images) - 1)

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx):
        item = self.images[idx]
        return {"images": [item['image']], "texts": [str(item['prompt'])]}

    def _our_and_human(self, scores) -> Tuple[List[float], List[float]]:
        s = scores.mean(axis=1) if hasattr(scores, "mean") else np.asarray(scores).mean(axis=1)
        ours = [float(s[idx][0]) for idx in range(len(self.images))]
        human = [float(np.array(self.images[idx]['human_alignment']).mean()) for idx in range(len(self.images))]
        return ours, human

    @staticmethod
    def correlation(our_scores, human_scores) -> Dict:
        return {'pearson': calc_pearson(human_scores, our_scores), 'kendall_b': kendall_tau_b(human_scores, our_scores),
                'pairwise_acc': pairwise_acc_with_tie_optimization(human_scores, our_scores)}

    def evaluate_scores(self, scores) -> Dict:
        """scores [n_items, n_images, n_texts] as returned by batch_forward (dataset.py:1335-1343)."""
        ours, human = self._our_and_human(scores)
        return {'alignment': self.correlation(ours, human)}

    def evaluate_scores_per_skill(self, scores) -> Dict:
        """Per-skill correlations from genai_skills.json (dataset.py:1345-1389)."""
        ours, human = self._our_and_human(scores)
        tags = json.load(open(os.path.join(self.root_dir, "genai_skills.json")))
        out = {}
        for tag, prompts in tags.items():
            idxs = [i for p in prompts for i in self.prompt_to_images.get(f"{int(p):05d}", [])]
            out[tag] = {'alignment': self.correlation([ours[i] for i in idxs], [human[i] for i in idxs])}
        return out


# ------------------------------------------------------------------------------------------------ driver
def run_genai_image_eval(score_func, dataset: GenAIBenchImage, result_dir: str, model_name: str, batch_size: int = 16,
                         num_prompts: Optional[int] = None, **kwargs) -> Dict:
    """genai_image_eval.py:109-168: scores are computed once with ``score_func.batch_forward`` and cached as
    ``<result_dir>/<model>_<n>_prompts.pt``; later runs only recompute the metrics."""
    os.makedirs(result_dir, exist_ok=True)
    n = num_prompts if num_prompts is not None else len(dataset.dataset)
    result_path = os.path.join(result_dir, f"{model_name}_{n}_prompts.pt")
    if os.path.exists(result_path):
        scores = torch.load(result_path)
    else:
        scores = score_func.batch_forward(dataset, batch_size=batch_size, **kwargs).cpu()
        torch.save(scores, result_path)
    results = dataset.evaluate_scores(scores)
    skills_path = os.path.join(dataset.root_dir, "genai_skills.json")
    if os.path.exists(skills_path):
        per_skill = dataset.evaluate_scores_per_skill(scores)
        with open(os.path.join(result_dir, f"{model_name}_{n}_per_skill.json"), 'w') as f:
            json.dump(per_skill, f)
        results['per_skill'] = per_skill
    return results
