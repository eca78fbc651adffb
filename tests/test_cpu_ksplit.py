# coding: utf-8
"""The k-split form of the 128 x 64 split tile (csrc/conv_gemm_bf16x3.hip, template KS = 2; round 5) restated on the host:
how the input-channel chunks of a tile are dealt to the two wave groups of a workgroup, how many barrier rounds each
executes, and the dispatcher's rule.  (The kernel itself is compared with the one-group loop and the oracle on the GPU:
tests/test_gpu_kernels.py::test_k_split_form_of_the_128x64_tile.)"""
import pytest


def split(cin, taps):
    """-> [(first chunk, chunks, k-steps)] of group 0 and group 1, and the barrier rounds of the common loop"""
    n = (cin + 31) // 32
    first = (n + 1) // 2
    groups = [(0, first, first * taps), (first, n - first, (n - first) * taps)]
    return groups, first * taps


@pytest.mark.parametrize("cin", [33, 40, 64, 72, 96, 160, 256, 512, 513, 1024])
@pytest.mark.parametrize("taps", [1, 3, 5])
def test_every_chunk_belongs_to_one_group_and_both_groups_run_the_same_barriers(cin, taps):
    groups, rounds = split(cin, taps)
    n = (cin + 31) // 32
    covered = [c for (c0, nc, _) in groups for c in range(c0, c0 + nc)]
    assert covered == list(range(n))                         # each chunk once, in order: sum = (first half) + (second half)
    assert groups[0][1] - groups[1][1] in (0, 1) and groups[1][1] >= 1
    # the loop's trip count is the first group's; the second idles (barrier only) for at most one chunk's worth of steps
    assert rounds == groups[0][2] and rounds - groups[1][2] in (0, taps)
    # a partial last chunk (cin % 32 != 0) always lands in the second group
    if cin % 32:
        assert groups[1][0] + groups[1][1] == n


def rule(B, T, M, cin, taps, gated=False, mode=1, max_blocks=256, min_steps=8):
    """dv3_conv_gemm_bf16x3_dispatch: `ks_ok`"""
    m_tiles = -(-(M // 2) // 64) if gated else -(-M // 128)
    nb2 = m_tiles * -(-(B * T) // 64)
    nch = (cin + 31) // 32
    ksteps = nch * taps
    if nch < 2 or mode == 0:
        return False
    if mode == 2:
        return True
    return ksteps >= min_steps and nb2 <= (max_blocks if ksteps >= 16 else max_blocks * 5 // 8)


def test_rule_takes_the_small_grids_of_batch_16_and_leaves_batch_64_alone():
    # profiles/r05_k_split.txt: B = 16 -- the encoder's input gradients (152 tiles, 96 steps), the decoder's three-tap layers
    # (102 / 204 tiles), 1 x 1 layers of K = 256 on at most 160 tiles
    assert rule(16, 150, 512, 1024, 3) and rule(16, 201, 256, 512, 3) and rule(16, 201, 512, 256, 3, gated=True)
    assert rule(16, 150, 256, 256, 1) and rule(16, 201, 256, 256, 1) and not rule(16, 201, 512, 256, 1)      # 204 tiles, 8 steps
    assert not rule(16, 150, 1024, 512, 3, gated=True)       # 304 tiles: slower there (1.19 x)
    assert not rule(16, 201, 256, 80, 1)                     # 3 k-steps: nothing to split
    # B = 64: the grids are past one tile per CU, and the 201-tile M = 80 layers have 8 steps only
    assert not rule(64, 150, 512, 1024, 3) and not rule(64, 201, 256, 512, 3) and not rule(64, 201, 80, 256, 1)
    assert not rule(64, 201, 256, 256, 1)
    # forced (dv3_debug_set(44, 2)) needs two chunks; off is off
    assert rule(64, 804, 512, 64, 3, mode=2) and not rule(64, 804, 512, 32, 3, mode=2) and not rule(16, 150, 512, 1024, 3, mode=0)
