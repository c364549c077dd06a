"""Property tests (hypothesis) of the integer / host logic around the hot path: they run on the CPU and need no GPU.
Each property is the size-independent statement the corresponding fixture test checks at a few points."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytest.importorskip("hypothesis")
from hypothesis import assume, given, settings, strategies as st  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ sharding (score.py:104-106 cells are independent)
@settings(max_examples=200, deadline=None)
@given(n=st.integers(0, 5000), ws=st.integers(1, 64))
def test_shard_range_is_a_balanced_contiguous_partition(n, ws):
    from t2v_metrics_amd.sharding import shard_range
    blocks = [shard_range(n, r, ws) for r in range(ws)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    sizes = []
    for (lo, hi), nxt in zip(blocks, blocks[1:] + [(n, n)]):
        assert 0 <= lo <= hi <= n and hi == nxt[0]
        sizes.append(hi - lo)
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


# ------------------------------------------------------------------ T5 relative-position buckets (HF modeling_t5.py:224-269)
def _c_bucket_lib():
    so = os.path.join(ROOT, "oracle", "_build", "librelpos_oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(so)


@settings(max_examples=60, deadline=None)
@given(rel=st.lists(st.integers(-6000, 6000), min_size=1, max_size=64), bidir=st.booleans(),
       nb=st.sampled_from([8, 16, 32, 64]), md=st.sampled_from([16, 64, 128, 256]))
def test_relpos_bucket_three_restatements_agree_with_hf(rel, bidir, nb, md):
    """numpy oracle == C oracle == the library's host function == HF's T5Attention._relative_position_bucket, for any
    distance, direction, bucket count and max distance (the fixture pins 1 401 points of the (32, 128) case)."""
    from oracle.clip_t5_oracle import relative_position_bucket
    from t2v_metrics_amd import engine
    # the formula divides by log(max_distance / max_exact): configurations with max_distance <= max_exact are degenerate
    # in HF itself (0/0 cast to an integer) and no T5 uses them (T5 / Flan-T5: 32 buckets, max distance 128)
    max_exact = (nb // 2 if bidir else nb) // 2
    assume(md > max_exact)
    rp = np.asarray(rel, dtype=np.int32)
    ref_np = relative_position_bucket(rp, bidir, nb, md).astype(np.int64)
    out = np.empty(rp.size, dtype=np.int32)
    _c_bucket_lib().t5_relpos_bucket_many(rp.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(rp.size), ctypes.c_int32(int(bidir)),
                                          ctypes.c_int32(nb), ctypes.c_int32(md), out.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(out.astype(np.int64), ref_np)
    lib_out = np.asarray([engine.relpos_bucket(int(r), bidir, nb, md) for r in rel], dtype=np.int64)
    assert np.array_equal(lib_out, ref_np)
    try:
        from transformers.models.t5.modeling_t5 import T5Attention
    except Exception:            # transformers is part of the image; keep the test meaningful without it
        return
    hf = T5Attention._relative_position_bucket(torch.from_numpy(rp.astype(np.int64)), bidirectional=bidir, num_buckets=nb,
                                               max_distance=md).numpy()
    assert np.array_equal(hf, ref_np)
    assert ref_np.min() >= 0 and ref_np.max() < nb


# ------------------------------------------------------------------ Qwen2.5-VL host layout (HF vision_utils.py:81-188)
@settings(max_examples=80, deadline=None)
@given(t=st.integers(1, 3), gh=st.integers(1, 20), gw=st.integers(1, 20))
def test_window_slots_place_every_cell_exactly_once(t, gh, gw):
    from t2v_metrics_amd.qwen.layout import window_slots
    merge, window, patch = 2, 112, 14
    ws = window // merge // patch
    slots, valid = window_slots(t, gh * merge, gw * merge, merge, window, patch)
    assert slots.numel() == valid.numel() * ws * ws
    present = slots[slots >= 0]
    assert torch.equal(torch.sort(present).values, torch.arange(t * gh * gw))        # a permutation of the cells
    assert int(valid.sum()) == t * gh * gw and int(valid.min()) >= 1 and int(valid.max()) <= ws * ws
    win = slots.reshape(-1, ws * ws)
    for row, v in zip(win, valid):                 # present cells first, in row-major (increasing) order, then padding
        v = int(v)
        assert bool((row[:v] >= 0).all()) and bool((row[v:] < 0).all())
        assert torch.equal(row[:v], torch.sort(row[:v]).values)
    # every window lies inside one frame and one ws x ws block of cells
    frame = present.new_tensor([int(c) // (gh * gw) for c in win[:, 0]])
    for row, v, f in zip(win, valid, frame):
        cells = row[: int(v)]
        assert bool((cells // (gh * gw) == f).all())
        r, c = (cells % (gh * gw)) // gw, (cells % (gh * gw)) % gw
        assert int(r.max() - r.min()) < ws and int(c.max() - c.min()) < ws


@settings(max_examples=40, deadline=None)
@given(t=st.integers(1, 2), gh=st.integers(1, 12), gw=st.integers(1, 12), nvid=st.integers(1, 3))
def test_vision_layout_maps_are_mutually_inverse(t, gh, gw, nvid):
    from t2v_metrics_amd.qwen.config import get_qwen_config
    from t2v_metrics_amd.qwen.layout import vision_layout
    cfg = get_qwen_config("qwen-tiny")
    m = cfg.vision.spatial_merge
    lay = vision_layout(cfg, [(t, gh * m, gw * m)] * nvid)
    N, Np, unit = lay["N"], lay["Np"], cfg.vision.merge_unit
    row_map, inv_row = lay["row_map"].long(), lay["inv_row"].long()
    assert N == nvid * t * gh * gw * unit and row_map.numel() == Np and Np % lay["win_len"] == 0
    assert torch.equal(row_map[inv_row], torch.arange(N))                 # windowed slot of patch i holds patch i
    real = row_map >= 0
    assert int(real.sum()) == N and int(lay["win_valid"].sum()) == N
    assert torch.equal(inv_row[row_map[real]], torch.nonzero(real)[:, 0])
    cell_inv = lay["cell_inv"].long()
    assert torch.equal(torch.sort(cell_inv).values, torch.unique(cell_inv)) and cell_inv.numel() == N // unit
    # the 4 patches of a merge cell stay consecutive in the windowed layout (the merger concatenates them)
    first = row_map.reshape(-1, unit)[:, 0]
    assert bool(((row_map.reshape(-1, unit) - first[:, None])[first >= 0] == torch.arange(unit)).all())
    # padded slots carry a zero rotation (cos 1, sin 0), real ones the frame table's entry of their patch
    assert torch.equal(lay["cos_w"][~real], torch.ones_like(lay["cos_w"][~real]))
    assert torch.equal(lay["cos_w"][real], lay["cos_f"][row_map[real]])


@settings(max_examples=300, deadline=None)
@given(h=st.integers(16, 5000), w=st.integers(16, 5000))
def test_smart_resize_properties_and_hf_equality(h, w):
    from t2v_metrics_amd.models.vqascore_models.qwen25vl_model import smart_resize
    if max(h, w) / min(h, w) > 200:
        with pytest.raises(ValueError):
            smart_resize(h, w)
        return
    factor, lo, hi = 28, 56 * 56, 14 * 14 * 4 * 1280
    hb, wb = smart_resize(h, w, factor, lo, hi)
    assert hb % factor == 0 and wb % factor == 0 and hb >= factor and wb >= factor
    assert hb * wb <= hi
    if h * w >= lo:
        assert hb * wb >= lo or min(hb, wb) == factor
    try:
        from transformers.models.qwen2_vl.image_processing_qwen2_vl import smart_resize as hf_smart_resize
    except Exception:
        return
    assert (hb, wb) == tuple(hf_smart_resize(h, w, factor=factor, min_pixels=lo, max_pixels=hi))


# ------------------------------------------------------------------ decoder inputs (HF modeling_t5.py:618-640)
@settings(max_examples=100, deadline=None)
@given(rows=st.lists(st.lists(st.integers(-1, 50), min_size=1, max_size=6), min_size=1, max_size=4))
def test_shift_right_matches_the_hf_rule(rows):
    from oracle.clip_t5_oracle import shift_right
    T = max(len(r) for r in rows)
    lab = torch.tensor([[(-100 if v < 0 else v) for v in r] + [-100] * (T - len(r)) for r in rows])
    out = shift_right(lab)
    assert out.shape == lab.shape and bool((out[:, 0] == 0).all())
    exp = lab[:, :-1].clone()
    exp[exp == -100] = 0
    assert torch.equal(out[:, 1:], exp)


# ------------------------------------------------------------------ head-major scatter of the QKV GEMM epilogue (HF modeling_t5.py:311-323)
@settings(max_examples=300, deadline=None)
@given(row0=st.integers(0, 200_000), S=st.integers(8, 1300), hx=st.integers(1, 64), hdim=st.sampled_from([64, 128]),
       n=st.integers(1, 64))
def test_heads_epilogue_stepped_offsets_equal_the_division(row0, S, hx, hdim, n):
    """The GEMM's head-major epilogue splits a lane's first row into (sample, position) once and then steps 8 rows at a
    time (vqs_kernels.h: heads_off_first / heads_off_step8, the same inline functions the kernel compiles); the host hook
    must reproduce  ((row // S) * hx * S + row % S) * hdim  for every row of the sequence."""
    from t2v_metrics_amd import engine
    lib = engine.load_library()
    out = np.empty(n, dtype=np.int64)
    assert lib.vqs_debug_heads_rows(row0, S, hx, hdim, n, out.ctypes.data_as(ctypes.c_void_p)) == 0
    rows = row0 + 8 * np.arange(n, dtype=np.int64)
    assert np.array_equal(out, ((rows // S) * hx * S + rows % S) * hdim)


def test_heads_epilogue_hook_rejects_short_sequences():
    from t2v_metrics_amd import engine
    lib = engine.load_library()
    out = np.empty(4, dtype=np.int64)
    assert lib.vqs_debug_heads_rows(0, 7, 2, 64, 4, out.ctypes.data_as(ctypes.c_void_p)) != 0      # S < 8 runs variant 0 instead


# ------------------------------------------------------------------ workgroup -> tile order of the GEMM kernels (gemm.hip / vqs_kernels.h tile_of_slot)
def _legacy_tile_map(M, N, batch):
    """The map of rounds 1-2, restated: each XCD (slot % 8) owns a contiguous run of the tile list, walked in groups of
    8 M-tiles x all N-tiles, M fastest."""
    tm, tn = -(-M // 256), -(-N // 256)
    nwg = tm * tn * batch
    q, r = nwg >> 3, nwg & 7
    out = {}
    for pid in range(nwg):
        xcd, local = pid & 7, pid >> 3
        t = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + local
        bz, t = divmod(t, tm * tn)
        group, x = divmod(t, 8 * tn)
        gsz = min(tm - group * 8, 8)
        out[pid] = ((group * 8 + x % gsz) * 256, (x // gsz) * 256, bz)
    return out


@settings(max_examples=120, deadline=None)
@given(tm=st.integers(1, 90), tn=st.integers(1, 40), batch=st.integers(1, 3), gm=st.sampled_from([0, 1, 2, 3, 4, 8, 16, 64]),
       ns=st.sampled_from([0, 1, 2, 4, 5]), ragged=st.booleans())
def test_gemm_tile_order_is_a_permutation(tm, tn, batch, gm, ns, ragged):
    """Whatever (gm, ns) a caller asks for, the persistent walk visits every tile of every batch entry exactly once; (0, 0)
    and (8, 1) are the map of rounds 1-2; an illegal ns (remainders, batched, too few tiles) falls back to 1."""
    from t2v_metrics_amd import engine
    lib = engine.load_library()
    M, N = tm * 256 - (17 if ragged else 0), tn * 256 - (8 if ragged else 0)
    nwg = tm * tn * batch
    out = np.empty((nwg, 4), dtype=np.int32)
    rc = lib.vqs_debug_tile_order(M, N, 64, batch, gm, ns, 256, out.ctypes.data_as(ctypes.c_void_p))
    assert rc > 0
    rgm, rns = rc & 0xff, rc >> 8
    assert rgm == (gm if gm else 8)                    # K = 64: tiny working set, the library's own choice is (8, 1) too
    legal = ns > 1 and batch == 1 and tn % ns == 0 and (tm * (tn // ns)) % 8 == 0 and tm * tn >= 8 * 256
    assert rns == (ns if legal else 1)
    assert sorted(out[:, 0].tolist()) == list(range(nwg))                      # every slot once
    tiles = {(int(m0), int(n0), int(bz)) for _, m0, n0, bz in out}
    assert tiles == {(i * 256, j * 256, b) for i in range(tm) for j in range(tn) for b in range(batch)}
    if rgm == 8 and rns == 1:
        legacy = _legacy_tile_map(M, N, batch)
        assert all(legacy[int(pid)] == (int(m0), int(n0), int(bz)) for pid, m0, n0, bz in out)


def test_gemm_tile_order_column_ranges_are_walked_one_after_the_other():
    """ns = 2 on the T5-XXL wi shape: every XCD finishes its share of the first half of N before any tile of the second."""
    from t2v_metrics_amd import engine
    lib = engine.load_library()
    M, N = 155648, 20480
    nwg = (M // 256) * (N // 256)
    out = np.empty((nwg, 4), dtype=np.int32)
    assert lib.vqs_debug_tile_order(M, N, 4096, 1, 4, 2, 256, out.ctypes.data_as(ctypes.c_void_p)) == (4 | 2 << 8)
    order = out[np.argsort(out[:, 0], kind="stable")]                           # by slot = by time within an XCD
    for xcd in range(8):
        mine = order[order[:, 0] % 8 == xcd]
        half = (mine[:, 2] >= N // 2).astype(int)
        assert (np.diff(half) >= 0).all() and half.sum() * 2 == len(mine)
        first = mine[:32]                                                        # the XCD's first 32 concurrent tiles: 4 M x 8 N
        assert len(set(first[:, 1])) == 4 and len(set(first[:, 2])) == 8


@pytest.mark.parametrize("what,M,N,K,expect", [
    ("t5-xxl wi", 155648, 20480, 4096, (4, 2)), ("t5-xxl wo", 155648, 4096, 10240, (4, 2)), ("t5-xxl qkv", 155648, 12288, 4096, (4, 2)),
    ("t5-xxl o", 155648, 4096, 4096, (8, 1)), ("t5-xl wi", 155648, 10240, 2048, (8, 1)), ("t5-xl wo", 155648, 2048, 5120, (4, 1)),
    ("t5-xl qkv", 155648, 6144, 2048, (8, 1)), ("vit fc1", 147712, 4096, 1024, (8, 1)), ("vit fc2", 147712, 1024, 4096, (8, 1)),
    ("genai bucket wo, S_e 648", 256 * 648, 4096, 10240, (4, 2)), ("dec wi (512 rows: small launch)", 512, 20480, 4096, (8, 1))])
def test_gemm_tile_order_library_choice_by_shape(what, M, N, K, expect):
    """The library's own tile order (gm = ns = 0): groups of 8 M-tiles while 8 XCDs x (8 M-tiles x K) of A plus W stay under
    180 MB of the 256 MB Infinity Cache, else groups of 4, and two column ranges when that still leaves > 240 MB --
    the measured optimum of every path shape (profiles/r2_call25_tile_order_*.jsonl)."""
    from t2v_metrics_amd import engine
    lib = engine.load_library()
    nwg = -(-M // 256) * -(-N // 256)
    out = np.empty((nwg, 4), dtype=np.int32)
    rc = lib.vqs_debug_tile_order(M, N, K, 1, 0, 0, 256, out.ctypes.data_as(ctypes.c_void_p))
    assert (rc & 0xff, rc >> 8) == expect, what


# ---------------------------------------------------------------------------------------------- GEMM form resolution
def _form(M, N, K, epi, batch=1, variant=3, lda=None, ldw=None, S=0, inner=0, inner_kv=0):
    from t2v_metrics_amd import engine
    return engine.load_library().vqs_debug_gemm_form(M, N, K, lda or K, ldw or K, epi, batch, variant, S, inner, inner_kv)


@given(st.integers(1, 400000), st.sampled_from([(20480, 4096, 5), (4096, 10240, 0), (12288, 4096, 6), (4096, 1024, 1), (1024, 4096, 0),
                                                (4096, 1024, 2), (3072, 1024, 6), (32128, 4096, 3), (2048, 64, 0)]))
def test_gemm_form_depends_on_the_weight_never_on_the_row_count(M, shape):
    """A pair's bits must not depend on the batch it is scored in (reference contract: independent cells, score.py:104-106):
    the kernel family of a call site is a function of the epilogue and the weight's shape -- for every M the same."""
    N, K, epi = shape
    S, inner = (608, N // 3) if epi == 6 else (0, 0)
    # 3 (persistent 8-wave) and 12 (stream form, round 4: few rows, many columns) produce the same bits -- one family; what must
    # not move with M is membership in the quad form (10), whose summation order differs
    fam = lambda f: 3 if f == 12 else (10 if f == 13 else f)      # 13 (round 6): the quad family's few-row launch shape, gemm_slim.inc -- same bits as 10
    f = fam(_form(M, N, K, epi, S=S, inner=inner))
    assert f == fam(_form(7, N, K, epi, S=S, inner=inner)) == fam(_form(155648, N, K, epi, S=S, inner=inner))
    assert f == (10 if epi in (0, 1, 2, 5, 6) and K >= 128 else 3)


def test_stream_form_takes_skinny_launches_with_many_columns_only():
    """gemm_stream.inc: <= 128 rows per batch entry, >= 4 (entry, 128-column) items (rounds 4-5: 192; round 6 lets the decoder's linears of a
    B <= 32 call stream their weights too), plain fp32 / bf16 results that are not quad call sites; everything else keeps its kernel."""
    assert _form(128, 608, 4096, 3, batch=256) == 12          # cross scores at the bench batch: 5 x 256 items
    assert _form(128, 4096, 640, 0, batch=256) == 12          # cross P.E (bf16 result, batched: never a quad launch)
    assert _form(128, 608, 4096, 3, batch=16) == 12           # 80 items (the persistent kernel until round 6)
    assert _form(4, 4096, 1024, 3, batch=4) == 12 and _form(4, 12288, 4096, 3) == 12      # a decoder linear of ONE pair: 4 K-slices x 32 blocks / 96 blocks
    assert _form(128, 608, 4096, 3, batch=1) == 12            # one pair's cross scores: 5 items
    assert _form(64, 384, 4096, 3) == 3                       # 3 items: not worth a launch shape of its own
    assert _form(129, 608, 4096, 3, batch=256) == 3           # more rows than the form holds
    assert _form(64, 152064, 3584, 3) == 12                   # a decode step's lm_head
    assert _form(64, 32768, 4096, 0) == 13 and _form(64, 32768, 4096, 0, variant=10) == 10      # a bf16-result nn.Linear is a quad call site for EVERY M (13: its few-row launch shape, same bits)
    assert _form(64, 32768, 4096, 0, variant=11) == 12        # under the 8-wave rule the same launch may stream (same bits as 3)
    assert _form(128, 32768, 4096, 5) == 13 and _form(128, 32768, 4096, 3, variant=0) == 0


def test_slim_form_takes_the_few_row_launches_of_quad_call_sites_only():
    """gemm_slim.inc: a quad call site whose 256 x 256 tiles would leave most of the chip idle (one or two pairs per call) runs 128 x 128 tiles instead --
    a function of (M, N, K) and the epilogue; variant 10 = the quad kernel whatever the shape; nothing changes at the bench batch."""
    assert _form(608, 4096, 4096, 0) == 13 and _form(608, 4096, 10240, 0) == 13          # T5-XXL encoder o / wo of one pair: 48 quad tiles
    assert _form(608, 12288, 4096, 6, S=608, inner=4096) == 10                            # q|k|v: 144 quad tiles against 480 slim tiles in two rounds (measured: 83 vs 94 us)
    assert _form(608, 20480, 4096, 5) == 10                                               # gated wi: 240 quad tiles fill the chip
    assert _form(577, 1024, 4096, 0) == 13 and _form(577, 4096, 1024, 1) == 13            # the tower's fc2 / fc1 of one image
    assert _form(1216, 4096, 4096, 0) == 10 and _form(1154, 1024, 4096, 0) == 13          # two pairs: 320 slim tiles = two rounds, the quad launch; the tower's fc2 of two images: 80
    assert _form(155648, 4096, 4096, 0) == 10 and _form(155648, 12288, 4096, 6, S=608, inner=4096) == 10   # the bench batch
    assert _form(608, 4096, 4096, 0, variant=10) == 10                                    # the quad kernel on request
    assert _form(608, 4000, 4096, 0) == 10 and _form(608, 4096, 128, 0) == 10             # N not a multiple of 128 / a short K: not worth it
    assert _form(4, 4096, 1024, 3, batch=4) == 12                                         # fp32-result decoder linears: the stream form as before


def test_gemm_operands_of_4_gib_never_reach_a_32_bit_kernel():
    """the 8-wave kernels address a batch entry's operands with 32-bit byte offsets; the XXL wo GEMM's A operand is
    3.19 GB at 256 pairs and S = 608, 4.3 GB at 345 pairs.  The quad form has no such limit (tile-relative offsets, 64-bit
    descriptor base); what still runs on the 8-wave kernels falls back to the one-tile-per-workgroup kernel or is refused."""
    big_m = 400000                                          # x 10 240 x 2 B = 8.2 GB
    assert _form(big_m, 4096, 10240, 0) == 10               # bf16 result: quad form, any size
    assert _form(big_m, 4096, 10240, 0, variant=11) == 0    # the 8-wave rule: falls back to 64-bit pointers (same bits)
    assert _form(big_m, 4096, 10240, 3) == 0                # fp32 result
    assert _form(big_m, 4096, 10240, 3, batch=2) == -1      # batched entries exist in the 32-bit kernels only: refused, not wrapped
    assert _form(big_m, 4096, 10240, 7) == -1               # fused residual + RMSNorm epilogue likewise
    assert _form(155648, 4096, 10240, 3) == 3 and _form(155648, 4096, 10240, 3, batch=2) == 3     # 3.19 GB: fits
    assert _form(209715, 4096, 10240, 3) == 3 and _form(209716, 4096, 10240, 3) == 0              # the boundary: M * lda * 2 < 2^32
    assert _form(512, 32128, 4096, 3, lda=4096, ldw=4096 * 17) == 0                                  # a strided W operand counts with its stride
