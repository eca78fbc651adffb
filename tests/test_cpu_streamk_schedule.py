# coding: utf-8
"""The stream-K schedule of the 256 x 256 tap-GEMM (csrc/conv_gemm_pp2.hip, SK form) restated on the host.
  * The P workgroups form 8 groups of q = P / 8 (one per XCD: dv3_xcd_remap gives XCD x the pid range [x q, (x + 1) q),
    slot s = blockIdx / 8); a group owns a whole number of tiles (no tile is cut between two groups) and deals its
    (tile, chunk) units out evenly.
  * Inside a group the unit ranges are numbered AGAINST the dispatch order (round 5): slot s takes index q - 1 - s.
  * A workgroup walks its range in ascending order; a tile's pieces are summed by the workgroup that holds its chunk 0,
    which waits for the FIRST segment of the workgroups with the next higher indices -- i.e. of workgroups with a LOWER
    block index, dispatched earlier, which wait for nothing.  No cycle, no need for the whole grid to be resident
    (rounds 3-4 waited for workgroups dispatched later), and a sum order that is a function of the shape.
(The kernel itself is compared with the tile-per-workgroup form on the GPU: tests/test_gpu_kernels.py.)"""
import itertools

import pytest


def eligible(tiles, S, P):
    """conv_gemm_pp2.hip: dv3_conv_gemm_pp2_dispatch (grp_ok and the unit count)"""
    q = P // 8
    return P % 8 == 0 and q > 0 and (q & (q - 1)) == 0 and tiles >= 8 and tiles * S >= 2 * P and (S & (S - 1)) == 0


def schedule(tiles, S, P):
    """-> {block index: (index w the kernel works with, [segments (tile, c0, c1, role, waits) in execution order])};
    waits = the indices whose pieces a finishing segment adds (kernel: the w2 loop)"""
    q = P // 8
    tg, tr = divmod(tiles, 8)
    out = {}
    for blk in range(P):
        xcd, slot = blk % 8, blk // 8                   # dv3_xcd_remap (P a multiple of 8): pid = xcd * q + slot
        w = xcd * q + (q - 1 - slot)                    # the index the kernel works with
        grp, ls = divmod(w, q)
        big = grp < tr
        base, rem = divmod((tg + (1 if big else 0)) * S, q)
        g0 = (grp * tg + min(grp, tr)) * S
        start = lambda sl: g0 + sl * base + min(sl, rem)
        u, end = start(ls), start(ls + 1)
        segs = []
        while u < end:
            tile = u // S
            c0 = u - tile * S
            c1 = min(S, c0 + end - u)
            waits = []
            if c0 == 0 and c1 < S:
                sl2 = ls + 1
                while sl2 < q and start(sl2) < (tile + 1) * S:
                    waits.append(grp * q + sl2)
                    sl2 += 1
            segs.append((tile, c0, c1, "partial" if c0 else "finish", waits))
            u += c1 - c0
        out[blk] = (w, segs)
    return out


CASES = [(152, 16, 256), (76, 32, 256), (808, 16, 256), (102, 8, 256), (404, 8, 256), (512, 8, 256), (78, 8, 256),
         (256, 16, 256), (257, 2, 256), (9, 16, 64), (8, 32, 8), (33, 8, 16), (64, 1, 16), (1000, 4, 128)]


@pytest.mark.parametrize("tiles,S,P", CASES + [(t, s, p) for t, s, p in itertools.product((8, 9, 33, 100), (1, 2, 8), (8, 16, 64))
                                               if t * s >= 2 * p])
def test_stream_k_schedule_covers_every_unit_once_and_cannot_deadlock(tiles, S, P):
    assert eligible(tiles, S, P)
    sch = schedule(tiles, S, P)
    q = P // 8
    blk_of = {w: blk for blk, (w, _) in sch.items()}
    assert sorted(blk_of) == list(range(P))
    done = {}
    for blk, (w, segs) in sch.items():
        # at most one hand-over per workgroup (one workspace slot), and it is the workgroup's FIRST segment
        assert sum(1 for s in segs if s[3] == "partial") <= 1 and all(s[3] == "finish" for s in segs[1:])
        for (tile, c0, c1, role, waits) in segs:
            assert 0 <= c0 < c1 <= S and tile < tiles
            for c in range(c0, c1):
                assert (tile, c) not in done
                done[(tile, c)] = w
    assert len(done) == tiles * S
    for blk, (w, segs) in sch.items():
        for (tile, c0, c1, role, waits) in segs:
            assert len(set(done[(tile, c)] // q for c in range(S))) == 1            # a tile never leaves its XCD group
            if role != "finish":
                continue
            # the pieces it adds are exactly the rest of the tile, in index order; each is the FIRST segment of a workgroup
            # of the same group with a LOWER block index (dispatched earlier), which itself waits for nothing
            rest = sorted(set(done[(tile, c)] for c in range(c1, S)))
            assert rest == waits
            for w2 in waits:
                assert w2 // q == w // q and blk_of[w2] < blk and blk - blk_of[w2] == 8 * (w2 - w)
                first = sch[blk_of[w2]][1][0]
                assert first[0] == tile and first[3] == "partial" and first[4] == []
    # balance: inside a group the shares differ by at most one unit; between groups by a tile's worth spread over q
    loads = {w: sum(c1 - c0 for (_, c0, c1, _, _) in segs) for _, (w, segs) in sch.items()}
    for g in range(8):
        lg = [loads[w] for w in range(g * q, (g + 1) * q)]
        assert max(lg) - min(lg) <= 1
    assert max(loads.values()) - min(loads.values()) <= -(-S // q) + 1


def test_stream_k_needs_whole_xcd_groups():
    assert not eligible(152, 16, 250) and not eligible(152, 16, 24 * 8) and not eligible(5, 16, 256) and eligible(152, 16, 256)
