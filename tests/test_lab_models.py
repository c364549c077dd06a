"""Host models of the hand-scheduled GEMM forms (tools/lab/model_ring_layout.py): the LDS images of the wide form (in the
library, gemm.hip gemm_bf16_wide) and of the ring form (lab build, csrc/lab/gemm_ring.inc) replayed lane by lane with the
kernels' address formulas, the accumulator -> AGPR tuple map, and the ring form's vmcnt protocol replayed over hundreds of
tile / slice / partial-tile schedules.  Arithmetic only: no GPU, no library."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("model_ring_layout", os.path.join(ROOT, "tools", "lab", "model_ring_layout.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_lds_images_and_accumulator_map():
    m = _model()
    assert "1 024 fragment reads correct" in m.check_ring() and "conflict-free" in m.check_ring()
    assert "2 048 fragment reads correct" in m.check_wide()
    assert "16 tuples" in m.check_acc_map()


def test_ring_vmcnt_protocol_covers_the_slice_read_next():
    assert "432 schedules" in _model().check_ring_protocol(4)
    assert "432 schedules" in _model().check_ring_protocol(3)


def test_the_models_formulas_are_the_kernels():
    """The model restates the kernels' address arithmetic; pin the source expressions it restates so that a change of one
    side without the other fails here."""
    wide = open(os.path.join(ROOT, "t2v_metrics_amd", "csrc", "gemm.hip")).read()
    ring = open(os.path.join(ROOT, "t2v_metrics_amd", "csrc", "lab", "gemm_ring.inc")).read()
    for needle in ["const int sw = ((w & 1) << 2) + (lane >> 4);", "koff[ks] = (((ks * 2 + (lane >> 5)) ^ swr) << 4);",
                   "const int a_row = (wr * 128 + (lane & 31)) * 128;", "return ((((j) & 3) >> 1) * 4 + ((j) >> 2)) * 2 + ((j) & 1);"]:
        assert needle in wide, needle
    for needle in ["const uint32_t gch_b = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);", "const int fsw = (lane >> 2) & 3;",
                   "koff[ks] = (((ks * 2 + (lane >> 5)) ^ fsw) << 4);", "const int a_row = (wr * 128 + (lane & 31)) * 64;",
                   "const int r0 = w * 16 + (lane >> 2);", "sync_extra = full ? RING_AHEAD : 0;", "constexpr int YOUNGER = 8 * (RING_AHEAD - 1);"]:
        assert needle in ring, needle
    assert len(re.findall(r"stream_piece\(D\)", ring)) == 1 and "RK_B(15, 7)" in ring
