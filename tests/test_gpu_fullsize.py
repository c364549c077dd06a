"""Full-size checks (-m gpu) at the real clip-flant5-xl architecture (BASELINE.json configs[1] shapes): the oracle is
too slow for whole batches here, so parity rests on size-independent properties plus a one-pair oracle comparison."""
import pytest
import torch

from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import make_seeded_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xl():
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("clip-flant5-xl")
    w = make_seeded_weights(cfg, seed=0, device="cuda:0")
    eng = VqsEngine(cfg, w, device="cuda:0")
    yield cfg, w, eng
    eng.close()


def _batch(cfg, B, n_img, L, seed):
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn(n_img, 3, cfg.vision.image, cfg.vision.image, generator=g).to(torch.bfloat16)
    ids = torch.randint(3, 32100, (B, L), generator=g)
    for b in range(B):
        n = L if b % 3 == 0 else int(torch.randint(L // 2, L + 1, (1,), generator=g))
        ids[b, int(torch.randint(0, n - 1, (1,), generator=g))] = -200
        ids[b, n - 1] = 1
        ids[b, n:] = 0
    labels = torch.tensor([[2163, 1]] * B)
    idx = torch.randint(0, n_img, (B,), generator=g)
    return pix, idx, ids, labels


def test_batch_composition_padding_and_image_order_invariance(xl):
    cfg, w, eng = xl
    pix, idx, ids, labels = _batch(cfg, 16, 4, 33, seed=3)
    feats = eng.encode_images(pix.cuda())
    lp16, sc16 = eng.score(feats, idx, ids, labels)
    lp16, sc16 = lp16.clone(), sc16.clone()
    assert torch.isfinite(lp16).all() and (lp16 <= 0).all() and ((sc16 > 0) & (sc16 <= 1)).all()
    # (a) a pair's result does not depend on what else is in the batch
    lp4, _ = eng.score(feats, idx[:4], ids[:4], labels[:4])
    assert torch.equal(lp4, lp16[:4])
    # (b) ... nor on where its image sits among the encoded images
    perm = torch.tensor([2, 0, 3, 1])
    feats_p = eng.encode_images(pix[perm].cuda())
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(4)
    lp_p, _ = eng.score(feats_p, inv[idx], ids, labels)
    assert torch.equal(lp_p, lp16)
    # (c) ... nor on extra right padding of the prompts (longer static encoder length, more masked keys)
    ids_pad = torch.cat([ids, torch.zeros(16, 7, dtype=ids.dtype)], dim=1)
    lp_pad, _ = eng.score(feats, idx, ids_pad, labels)
    assert (lp_pad - lp16).abs().max().item() <= 1e-4
    # (d) score = exp(mean label log-prob)
    assert torch.allclose(sc16, torch.exp(lp16.mean(-1)), rtol=1e-5, atol=0)


def test_one_pair_against_the_cpu_oracle_at_full_size(xl):
    """fp32 oracle on the host for ONE pair of the full-size model (~15-30 s); bound = bf16 operand noise."""
    from oracle.clip_t5_oracle import Oracle
    cfg, w, eng = xl
    pix, idx, ids, labels = _batch(cfg, 1, 1, 33, seed=9)
    lp, sc = eng.score(eng.encode_images(pix.cuda()), idx, ids, labels)
    torch.cuda.synchronize()
    ref = Oracle(cfg, {k: v.cpu() for k, v in w.items()}).forward(pix.float(), idx, ids, labels)
    d = (lp.cpu() - ref["label_logprobs"]).abs().max().item()
    assert d <= 2.5e-2, d
