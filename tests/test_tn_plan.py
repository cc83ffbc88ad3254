"""Host logic of the per-pass wgrad launch (csrc/gemm_f32.hip::sln_tn_multi_plan) - runs without a GPU:
every (problem, output tile, row chunk) appears exactly once, the table is a multiple of 8 workgroups long, all tiles of one
(problem, chunk) sit on ONE XCD (workgroup b runs on XCD b % 8) in consecutive slots, the XCDs' loads are balanced, the
deterministic mode plans one chunk per problem, and a launch that would not fit the table gets longer chunks."""
import ctypes as C

import numpy as np

from conftest import pkg


def _plan(shapes):
    lib = pkg("_lib").lib()
    R, N, K = (np.ascontiguousarray([s[i] for s in shapes], np.int32) for i in range(3))
    rpb = np.zeros(len(shapes), np.int32)
    items = np.full((4096, 3), -7, np.int32)
    nb = lib.sln_debug_tn_plan(R.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), K.ctypes.data_as(C.c_void_p), len(shapes),
                               rpb.ctypes.data_as(C.c_void_p), items.ctypes.data_as(C.c_void_p), 4096)
    assert nb >= 0, nb
    return rpb, items[:nb]


def _check(shapes, rpb, items):
    assert items.shape[0] % 8 == 0
    seen = {}
    for b, (p, t, c) in enumerate(items):
        if p < 0:
            continue
        assert (p, t, c) not in seen, "duplicate work item"
        seen[(p, t, c)] = b
    want = 0
    per_xcd = np.zeros(8)
    for p, (R, N, K) in enumerate(shapes):
        assert rpb[p] % 32 == 0 and rpb[p] > 0
        tiles, chunks = -(-N // 64) * -(-K // 64), -(-R // rpb[p])
        want += tiles * chunks
        for c in range(chunks):
            slots = [seen[(p, t, c)] for t in range(tiles)]              # KeyError = a missing item
            assert len({s % 8 for s in slots}) == 1, "tiles of one (problem, chunk) on several XCDs"
            assert slots == list(range(slots[0], slots[0] + 8 * tiles, 8)), "tiles of a group are consecutive on their XCD"
            per_xcd[slots[0] % 8] += tiles * min(rpb[p], R - c * rpb[p])
    assert len(seen) == want
    return per_xcd


def test_default_model_pass_is_covered_once_and_balanced():
    T, O, H, D = 4096, 2048, 256, 128                                    # one pass of the 64-graph step: 5 layers + heads
    shapes = []
    for _ in range(5):
        shapes += [(T, 2 * H + D, H), (O, H, H), (O, D, H)]
    shapes += [(O, H, 128), (O, 128, H), (O, 48, 128), (O, 48, 128), (O, 16, 128), (O, 16, 128)]
    rpb, items = _plan(shapes)
    load = _check(shapes, rpb, items)
    assert load.max() <= 1.15 * load.mean(), load                       # longest-first dealing keeps the XCDs within 15 %
    assert all(r in (768, 704, 1024, 2048) or r % 32 == 0 for r in rpb)
    assert rpb[0] < T and -(-T // rpb[0]) >= 4                            # row chunks of ~768 rows


def test_ragged_shapes_and_tails():
    shapes = [(13, 8, 36), (100, 24, 256), (4097, 640, 256), (33, 70, 130), (2048, 6, 256)]
    rpb, items = _plan(shapes)
    _check(shapes, rpb, items)


def test_a_launch_that_overflows_the_table_gets_longer_chunks():
    shapes = [(262144, 640, 256)] * 10 + [(131072, 256, 256)] * 10      # 4 096 graphs per step
    rpb, items = _plan(shapes)
    assert items.shape[0] <= 4096
    _check(shapes, rpb, items)
    assert rpb[0] > 768


def test_deterministic_mode_plans_one_chunk_per_problem():
    lib = pkg("_lib").lib()
    shapes = [(4096, 640, 256), (2048, 256, 256), (4096, 256, 384)]
    try:
        lib.sln_set_deterministic(1)
        rpb, items = _plan(shapes)
    finally:
        lib.sln_set_deterministic(0)
    _check(shapes, rpb, items)
    assert all(rpb[p] >= shapes[p][0] for p in range(3))
    assert {c for p, t, c in items if p >= 0} == {0}
