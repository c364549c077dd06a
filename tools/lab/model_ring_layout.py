#!/usr/bin/env python3
"""Host model of the ring GEMM form's LDS image (csrc/lab/gemm_ring.inc): replays the LDS-DMA piece map and the fragment read
addresses lane by lane with the formulas of the kernel and checks that (i) every 16-B chunk of a 256-row x 32-k slice half lands
exactly once, (ii) every fragment read returns the (row, k-chunk) the MFMA operand map wants, (iii) each 16-lane group of a
ds_read_b128 touches 16 distinct 16-B slots of the 256-B bank row (conflict-free).  Also replays the wide form's 128-B-row image
(gemm.hip gemm_bf16_wide) the same way.  No GPU, no library: arithmetic only."""
import itertools


def check_ring():
    lds = {}                                                     # byte offset (16-B aligned) -> (row, logical chunk)
    for w, i, lane in itertools.product(range(4), range(4), range(64)):
        q = 4 * i + w                                            # piece: rows 16 q .. 16 q + 15
        row = w * 16 + (lane >> 2) + i * 64                      # r0 + i * 64 of stream_tile()
        assert row == 16 * q + (lane >> 2)
        chunk = (lane & 3) ^ ((lane >> 4) & 3)                   # gch_b >> 4: the logical chunk this lane fetches
        off = q * 1024 + lane * 16                               # lane-linear LDS destination of the piece
        assert off not in lds
        lds[off] = (row, chunk)
    assert len(lds) == 256 * 4 and set(lds.values()) == {(r, c) for r in range(256) for c in range(4)}
    for wr, m, ks, lane in itertools.product(range(2), range(4), range(2), range(64)):
        fsw = (lane >> 2) & 3
        koff = ((ks * 2 + (lane >> 5)) ^ fsw) << 4
        a_row = (wr * 128 + (lane & 31)) * 64
        off = a_row + m * 2048 + koff
        want = (wr * 128 + m * 32 + (lane & 31), ks * 2 + (lane >> 5))      # row of the wave tile, 8-wide k chunk of the slice
        assert lds[off] == want, (wr, m, ks, lane, lds[off], want)
    for m, ks, grp in itertools.product(range(4), range(2), range(4)):        # bank slots per 16-lane group
        slots = set()
        for lane in range(16 * grp, 16 * grp + 16):
            fsw = (lane >> 2) & 3
            off = (lane & 31) * 64 + m * 2048 + (((ks * 2 + (lane >> 5)) ^ fsw) << 4)
            slots.add((off % 256) // 16)
        assert len(slots) == 16, (m, ks, grp, sorted(slots))
    return "ring image: 1 024 chunks placed once, 1 024 fragment reads correct, 32 read groups conflict-free"


def check_wide():
    lds = {}
    for w, i, lane in itertools.product(range(4), range(8), range(64)):
        q = 4 * i + w                                            # piece: rows 8 q .. 8 q + 7 (128-B rows)
        row = (i * 4 + w) * 8 + (lane >> 3)
        sw = ((w & 1) << 2) + (lane >> 4)
        assert sw == (row >> 1) & 7
        chunk = (lane & 7) ^ sw
        off = q * 1024 + lane * 16
        assert off not in lds
        lds[off] = (row, chunk)
    assert len(lds) == 256 * 8 and set(lds.values()) == {(r, c) for r in range(256) for c in range(8)}
    for wr, m, ks, lane in itertools.product(range(2), range(4), range(4), range(64)):
        swr = (lane >> 1) & 7
        koff = ((ks * 2 + (lane >> 5)) ^ swr) << 4
        off = (wr * 128 + (lane & 31)) * 128 + m * 4096 + koff
        want = (wr * 128 + m * 32 + (lane & 31), ks * 2 + (lane >> 5))
        assert lds[off] == want, (wr, m, ks, lane, lds[off], want)
    return "wide image: 2 048 chunks placed once, 2 048 fragment reads correct"


def check_acc_map():
    """MFMA j of a k-step (row block j >> 2, column block j & 3) -> AGPR tuple ((half * 4 + m) * 2 + n); the epilogue reads tuples
    0..7 as c[m][n] of half 0 and 8..15 of half 1."""
    seen = {}
    for j in range(16):
        m, n4 = j >> 2, j & 3
        idx = (((j & 3) >> 1) * 4 + (j >> 2)) * 2 + (j & 1)
        half, n = n4 >> 1, n4 & 1
        assert idx == (half * 4 + m) * 2 + n
        seen[idx] = (half, m, n)
    assert sorted(seen) == list(range(16))
    for half in range(2):
        for k, (m, n) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1)]):
            assert seen[8 * half + k] == (half, m, n)
    return "accumulator map: 16 tuples, epilogue order matches"


if __name__ == "__main__":
    for f in (check_ring, check_wide, check_acc_map):
        print(f())


def check_ring_protocol(slots=4):
    """Replay of the ring form's VMEM issue order and slice-boundary waits for one wave (control flow of gemm_bf16_ring, ring of
    `slots` slices, LDS-DMA slots - 1 slices ahead): at every boundary the `s_waitcnt vmcnt(N)` used must be <= the number of VMEM
    operations issued AFTER the last LDS-DMA piece of the slice that is read next (VMEM retires in order: then that slice has
    landed), for every slices-per-tile count, number of tiles per workgroup, full / partial tile pattern and epilogue store
    count; and a piece must only be written to the slot of a slice whose reads are over."""
    import random
    rng = random.Random(0)
    ahead = slots - 1
    checked = 0
    for ns in (2, 3, 4, 5, 8, 64):
        for ntiles in (1, 2, 3, 5):
            for nst in (16, 32, 64):
                for trial in range(6):
                    full = [rng.random() < 0.7 for _ in range(ntiles)]
                    total = ns * ntiles
                    log = []                                     # VMEM ops in issue order: ("dma", slice) x 8 or ("st", tile)
                    s_g = 0                                      # stream slices issued
                    def issue():
                        nonlocal s_g
                        log.extend([("dma", s_g)] * 8)
                        s_g += 1
                    for _ in range(ahead):
                        if s_g < total:
                            issue()
                    c_g, counted, sync_extra = 0, False, 0
                    def need(slice_idx, n_imm):
                        last = max(i for i, op in enumerate(log) if op == ("dma", slice_idx))
                        younger = len(log) - 1 - last
                        assert n_imm <= younger or n_imm == 0, (slots, ns, ntiles, nst, slice_idx, n_imm, younger)
                    def boundary_wait():
                        nonlocal sync_extra
                        if s_g >= total:                         # !s_valid
                            sync_extra = 0
                            return 0
                        if sync_extra > 0:
                            sync_extra -= 1
                            return min(8 * (ahead - 1) + nst, 63)
                        return 8 * (ahead - 1)
                    for t in range(ntiles):
                        if counted:
                            n = boundary_wait()
                        else:
                            n, sync_extra = 0, 0
                        need(c_g, n)
                        for k in range(ns):
                            if s_g < total:                      # k-step 0 of slice c_g issues stream slice c_g + ahead into the slot of slice c_g - 1
                                assert s_g == c_g + ahead and (s_g % slots) == ((c_g - 1) % slots)
                                issue()
                            if k + 1 < ns:
                                need(c_g + 1, boundary_wait())
                            c_g += 1
                        stores = nst if full[t] else rng.randrange(0, nst)
                        log.extend([("st", t)] * stores)
                        counted = full[t]
                        sync_extra = ahead if full[t] else 0
                    assert c_g == total and s_g == total
                    checked += 1
    return "ring protocol (%d slots): %d schedules replayed, every boundary wait covers the slice read next" % (slots, checked)


if __name__ == "__main__":
    print(check_ring_protocol(4))
    print(check_ring_protocol(3))
