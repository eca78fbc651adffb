# coding: utf-8
"""The stream-K schedule of the 256 x 256 tap-GEMM (csrc/conv_gemm_pp2.hip, SK form) restated on the host: every
(tile, chunk) unit is computed exactly once, a tile's parts are summed by the workgroup that holds its chunk 0, and a
workgroup only ever waits for the FIRST segment of a workgroup with a higher index -- which waits for nothing -- so the
launch cannot deadlock whatever order the hardware starts the workgroups in, and the sum order is a function of the shape.
(The kernel itself is compared with the tile-per-workgroup form on the GPU: tests/test_gpu_kernels.py.)"""
import itertools

import pytest


def schedule(tiles, S, P):
    """-> per workgroup w: list of segments (tile, c0, c1, role) in execution order, role in {"partial", "finish"};
    for every finishing segment with c1 < S the list of workgroups whose partials it adds (kernel: the w2 loop)."""
    U = tiles * S
    base, rem = divmod(U, P)
    start = lambda w: w * base + min(w, rem)
    out = []
    for w in range(P):
        u, end = start(w), start(w) + base + (1 if w < rem else 0)
        segs = []
        while u < end:
            tile = u // S
            c0 = u - tile * S
            c1 = min(S, c0 + end - u)
            waits = []
            if c0 == 0 and c1 < S:
                w2 = w + 1
                while w2 < P and start(w2) < (tile + 1) * S:
                    waits.append(w2)
                    w2 += 1
            segs.append((tile, c0, c1, "partial" if c0 else "finish", waits))
            u += c1 - c0
        out.append(segs)
    return out


@pytest.mark.parametrize("tiles,S,P", [(152, 16, 256), (76, 32, 256), (808, 16, 256), (102, 8, 256), (404, 8, 256), (512, 8, 256),
                                       (78, 8, 256), (5, 4, 7), (3, 16, 8), (256, 16, 256), (257, 2, 256)] +
                         [(t, s, p) for t, s, p in itertools.product((1, 2, 9, 33), (1, 2, 8), (1, 3, 16)) if t * s >= 2 * p])
def test_stream_k_schedule_covers_every_unit_once_and_cannot_deadlock(tiles, S, P):
    sch = schedule(tiles, S, P)
    done = {}
    for w, segs in enumerate(sch):
        assert sum(1 for s in segs if s[3] == "partial") <= 1 and (not segs or all(s[3] == "finish" for s in segs[1:]))
        for (tile, c0, c1, role, waits) in segs:
            assert 0 <= c0 < c1 <= S and tile < tiles
            for c in range(c0, c1):
                assert (tile, c) not in done
                done[(tile, c)] = w
    assert len(done) == tiles * S
    for w, segs in enumerate(sch):
        for (tile, c0, c1, role, waits) in segs:
            if role != "finish":
                continue
            # the parts it adds are exactly the rest of the tile, each the FIRST segment of a later workgroup, in order
            rest = sorted(set(done[(tile, c)] for c in range(c1, S)))
            assert rest == waits and all(w2 > w for w2 in waits)
            for w2 in waits:
                first = sch[w2][0]
                assert first[0] == tile and first[3] == "partial" and first[4] == []
    # balance: the shares differ by at most one unit
    loads = [sum(c1 - c0 for (_, c0, c1, _, _) in segs) for segs in sch]
    assert max(loads) - min(loads) <= 1
