"""Parity on the configuration that is BENCHMARKED (-m gpu): clip-flant5-xxl, bench.synth_batch, 256 pairs per pass -- M = 155 648
rows per encoder GEMM, the library's own tile order for that working set ((4, 2) for wi / qkv / wo), the A-panel L2 touch of wo,
~190 output tiles per persistent workgroup, two M-tiles in the decoder's split-K GEMMs.  No smaller test reaches those launch
configurations.

(a) The reference scores every (image, text) cell independently (/root/reference/t2v_metrics/score.py:104-106), so a pair's
    label log-probs in the 256-pair pass must be BIT-EQUAL to its log-probs in a 4-pair pass of the same pairs: vision tower,
    encoder, decoder (split-K slicing is a function of the weight's shape only) and score head.
(b) Stage-locked: every launch output of the 256-pair pass, restricted to the rows of two sampled pairs (vqs_debug_tap_window),
    against the rounding-matched oracle evaluated on the engine's own inputs (<= 1 bf16 ulp, per element).
"""
import pytest
import torch

import bench
from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import make_seeded_weights

pytestmark = pytest.mark.gpu
B = 256


@pytest.fixture(scope="module")
def xxl_bench():
    from t2v_metrics_amd.engine import VqsEngine
    cfg = get_config("clip-flant5-xxl")
    dev = torch.device("cuda", 0)
    w = make_seeded_weights(cfg, seed=0, device=dev)
    eng = VqsEngine(cfg, w, device=dev)
    pix, idx, ids, labels = bench.synth_batch(cfg, B, seed=1234, device=dev)       # the bench default's first batch
    lp, sc = eng.score(eng.encode_images(pix), idx, ids, labels)
    torch.cuda.synchronize()
    yield cfg, w, eng, (pix, idx, ids, labels), lp.clone(), sc.clone()
    eng.close()
    del w
    torch.cuda.empty_cache()


def test_bench_batch_is_the_benchmarked_launch_configuration(xxl_bench):
    """Guards the premise of this file: the pass really runs the launch configurations the bench line is measured on."""
    import ctypes
    import numpy as np
    from t2v_metrics_amd import engine
    cfg, w, eng, (pix, idx, ids, labels), lp, sc = xxl_bench
    S = ids.shape[1] - 1 + cfg.vision.n_patches
    assert (B * S, S) == (155648, 608)
    lib = engine.load_library()
    out = np.zeros(4 * 608 * 80, dtype=np.int32)
    for N, K in ((20480, 4096), (12288, 4096), (4096, 10240)):                     # wi, qkv, wo: working set > 180 MB -> (4, 2)
        rc = lib.vqs_debug_tile_order(B * S, N, K, 1, 0, 0, 256, out.ctypes.data_as(ctypes.c_void_p))
        assert rc == (4 | 2 << 8), (N, K, rc)
    assert torch.isfinite(lp).all() and (lp <= 0).all() and ((sc > 0) & (sc <= 1)).all()
    assert torch.allclose(sc, torch.exp(lp.mean(-1)), rtol=1e-5, atol=0)


@pytest.mark.parametrize("sel", [[3, 77, 130, 201], [64, 127, 128, 255]])
def test_a_pairs_logprobs_in_the_256_batch_are_bit_equal_to_a_4_pair_pass(xxl_bench, sel):
    cfg, w, eng, (pix, idx, ids, labels), lp256, sc256 = xxl_bench
    s = torch.tensor(sel, device=pix.device)
    feats4 = eng.encode_images(pix[s].contiguous())                                 # 4 images: M = 2 308 rows in the tower's GEMMs
    lp4, sc4 = eng.score(feats4, torch.arange(4, dtype=idx.dtype, device=idx.device), ids[s].contiguous(), labels[s].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(lp4, lp256[s]), (lp4 - lp256[s]).abs().max().item()
    assert torch.equal(sc4, sc256[s])


def test_repeating_the_256_pass_is_bitwise_stable(xxl_bench):
    cfg, w, eng, (pix, idx, ids, labels), lp256, sc256 = xxl_bench
    lp, sc = eng.score(eng.encode_images(pix), idx, ids, labels)
    torch.cuda.synchronize()
    assert torch.equal(lp, lp256) and torch.equal(sc, sc256)


def test_stage_locked_rows_of_two_sampled_pairs_inside_the_256_batch(xxl_bench):
    """793 launch outputs of the B = 256 pass, rows of pairs 130 and 131 (mid-batch: M-tile 308 of 608, second half of the
    workgroups' tile lists), each against the oracle on the engine's own inputs."""
    from tests.test_gpu_stage_locked import run_stage_locked
    cfg, w, eng, (pix, idx, ids, labels), lp256, sc256 = xxl_bench
    report, lp = run_stage_locked(cfg, w, eng, pix, idx.cpu(), ids, labels, "clip-flant5-xxl/bench-batch-256/pairs-130-131", window=(130, 2),
                                  oracle_device="cuda")
    assert torch.equal(lp, lp256[130:132].cpu())                                    # the tapped pass is the same pass
