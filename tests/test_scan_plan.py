"""Host logic of the persistent ADC scan (scan_variant 6, cvt_amd/csrc/adc_scan_h.hip): the item table must cover every
(query group, row) exactly once, segment indices must follow the rows (the merge's tie rule), and every workgroup must get
its items in round order.  Pure host code: runs without a GPU (cvtmi_opq_scan_plan)."""
import ctypes as C

import numpy as np
import pytest

import cvt_amd


def plan(n_rows, nq, splits=0, cus=256):
    lib = cvt_amd.lib()
    lib.cvtmi_opq_scan_plan.restype = C.c_int64
    lib.cvtmi_opq_scan_plan.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
    g, r, s = C.c_int(), C.c_int(), C.c_int()
    n = lib.cvtmi_opq_scan_plan(n_rows, nq, splits, cus, None, 0, C.byref(g), C.byref(r), C.byref(s))
    assert n >= 0 and n == g.value * r.value
    items = np.zeros((max(n, 1), 6), np.int64)
    assert lib.cvtmi_opq_scan_plan(n_rows, nq, splits, cus, items.ctypes.data, n, C.byref(g), C.byref(r), C.byref(s)) == n
    return items[:n].reshape(r.value, g.value, 6), g.value, r.value, s.value


@pytest.mark.parametrize("n_rows,nq,splits", [
    (1_000_000, 10_000, 0), (1_000_000, 1000, 0), (1_000_000, 8, 0), (1_000_000, 1, 0), (1_000_000, 4500, 0), (20_013, 9, 0), (7, 3, 0),
    (7, 3, 2), (20_013, 77, 3), (20_013, 77, 8), (1_000_000, 64, 16), (1 << 30, 10_000, 0), (128_000_000, 2048, 0), (300_000_000, 16, 0),
    (2049, 8, 0), (70_005, 530, 0), (5_999_999, 100, 0), (6_000_001, 100, 0)])
def test_item_table_covers_every_row_once(n_rows, nq, splits):
    items, grid, rounds, stride = plan(n_rows, nq, splits)
    groups = (nq + 7) // 8
    assert 1 <= grid <= 512
    used = items[items[:, :, 4] > 0]
    assert np.all(used[:, 0] < groups) and np.all(used[:, 2] <= (1 << 28) - 4096)
    u = items[:, :, 4] > 0                     # unused entries only after a workgroup's used ones
    assert rounds == 1 or np.all(u[:-1] | ~u[1:])
    sample = range(groups) if groups <= 200 else list(range(0, groups, max(1, groups // 150))) + [groups - 1]
    for g in sample:
        seg = used[used[:, 0] == g]
        seg = seg[np.argsort(seg[:, 3])]
        assert np.all(seg[:, 5] * 64 < np.maximum(seg[:, 2], 1))   # the walk starts inside the segment
        assert len(seg) >= 1 and np.array_equal(seg[:, 3], np.arange(len(seg))) and np.all(seg[:, 4] == len(seg)) and len(seg) <= stride
        row = 0
        for _, r0, rows, _, _, _ in seg:          # ascending, gap-free, ending at n_rows
            assert r0 * 64 == row or rows == 0
            row += rows
        assert row == n_rows
    assert stride == max(1, used[:, 4].max())
    assert int(used[:, 2].sum()) == groups * n_rows


def test_row_time_layout():
    """Shares of at least one group: whole groups start at the share's row time, cut groups give their first rows to the share that
    starts with them -- so that concurrently busy workgroups are at the same rows."""
    cvt_amd.set_tuning("scanh_balance", 1)
    try:
        items, grid, rounds, stride = plan(1_000_000, 10_000)
    finally:
        cvt_amd.set_tuning("scanh_balance", 0)
    tg = (1_000_000 + 2047) // 2048
    for w in range(0, grid, 37):
        t = 0                                                        # tiles this workgroup has been through
        mine = [it for it in items[:, w] if it[4] > 0]
        for j, (g, r0, rows, sidx, nseg, c0) in enumerate(mine):
            if rows == 1_000_000:
                assert c0 == (t % tg) * 32
            elif j == 0 and nseg == 2:
                assert r0 == 0 and c0 == 0 and sidx == 0             # begins with the group's first rows
            else:
                assert j == len(mine) - 1 and r0 * 64 + rows == 1_000_000 and sidx == nseg - 1
            t += (rows + 2047) // 2048


def test_last_round_can_be_cut_finer():
    """scanh_tail = 1 at SIFT-1M x 10 000 queries: two whole rounds of 512 query groups, the 226 groups of the last round in two row
    splits each -- every workgroup gets the same rows (2.5 M).  (Off by default: it measured no gain.)"""
    items, grid, rounds, stride = plan(1_000_000, 10_000)
    assert (grid, rounds, stride) == (512, 3, 1) and np.sum(items[2, :, 4] > 0) == 226
    cvt_amd.set_tuning("scanh_tail", 1)
    try:
        items, grid, rounds, stride = plan(1_000_000, 10_000)
    finally:
        cvt_amd.set_tuning("scanh_tail", 0)
    assert (grid, rounds, stride) == (512, 3, 2)
    assert np.all(items[:2, :, 2] == 1_000_000) and np.all(items[:2, :, 4] == 1)
    last = items[2]
    assert np.sum(last[:, 4] == 2) == 452 and np.all(last[last[:, 4] == 2][:, 0] >= 1024)


def test_balanced_shares_are_equal():
    cvt_amd.set_tuning("scanh_balance", 1)
    try:
        items, grid, rounds, stride = plan(1_000_000, 10_000)
    finally:
        cvt_amd.set_tuning("scanh_balance", 0)
    per_wg = (items[:, :, 2] * (items[:, :, 4] > 0)).sum(axis=0)
    assert grid == 512 and stride in (2, 3)
    assert per_wg.max() - per_wg.min() <= 2 * 4 * 2048 + 2048        # boundaries snap by at most 4 tiles
    seg = items[items[:, :, 4] > 0][:, 2]
    assert seg.min() >= 4 * 2048 - 2048                              # no sliver segments


@pytest.mark.parametrize("nq,want", [(600, 6), (1000, 4), (1300, 3), (1500, 4), (1800, 2), (3000, 2), (3500, 1), (4000, 1), (10_000, 1)])
def test_planner_follows_the_measured_split_counts(nq, want):
    """The planner prices a split count by the makespan of the persistent grid (items round-robin over 512 workgroups, two per CU, the
    one that is left running 1.65 x faster, an item = 80 K + 80 K / S + rows / S row-equivalents).  At 1 M rows it has to land on the
    split counts tools/sweep_scan_h.py measured fastest on an MI355X: e.g. 1000 queries 4 splits (0.43 ms; 2 splits
    0.70), 3000 queries 2 (1.04 ms; whole groups 1.24), whole groups from 3500 queries on."""
    _, grid, rounds, stride = plan(1_000_000, nq)
    assert stride == want, (nq, stride, grid, rounds)


def dispatch(n_rows, nq, M=16, k=100, D=128, K=256):
    lib = cvt_amd.lib()
    out = (C.c_int * 7)()
    assert lib.cvtmi_opq_describe_dispatch(D, M, K, C.c_int64(n_rows), C.c_int64(nq), k, out) == 0
    return dict(small=out[0], variant=out[1], qtile=out[2], splits=out[3], groups_a=out[4], splits_b=out[5], padded_m=out[6])


def test_dispatch_rules_of_round_5():
    """The OPQ search's choice of scan form (cvtmi_opq_describe_dispatch: host logic, 256 CUs assumed without a device) at the cells
    the round-5 sweeps fixed (profiles/r05_scan_dispatch_sweep.txt, r05_opq_m_sweep.txt) and at the ones that must not move."""
    # the headline: whole query groups through adc_scan16q, no tail splits at 10 000 queries, the tail rule at 4256
    d = dispatch(1_000_000, 10_000)
    assert (d["small"], d["variant"], d["qtile"], d["splits"], d["splits_b"], d["padded_m"]) == (0, 3, 8, 1, 0, 0)
    d = dispatch(1_000_000, 4256)
    assert d["variant"] == 3 and d["splits"] == 1 and d["groups_a"] == 512 and d["splits_b"] > 1
    # the SIFT-1B-shaped shard keeps adc_scan16q
    assert dispatch(1 << 30, 1000)["variant"] == 3 and dispatch(128_000_000, 1000)["variant"] == 3
    # small batches: the small-batch form while rows x query groups <= 48 M ...
    assert dispatch(1_000_000, 1)["small"] == 1 and dispatch(1_000_000, 128)["small"] == 1 and dispatch(2_000_000, 128)["small"] == 1
    assert dispatch(10_000_000, 32)["small"] == 1 and dispatch(30_000_000, 8)["small"] == 1
    # ... and not beyond (30 M rows x 128 queries took 2.6 ms there against 1.3)
    for n, nq in ((10_000_000, 64), (30_000_000, 16), (30_000_000, 128), (100_000_000, 1), (100_000_000, 128)):
        assert dispatch(n, nq)["small"] == 0, (n, nq)
    # the persistent grid: from 33 queries on a cache-resident matrix, up to 512 queries while it fits the Infinity Cache, and for one to
    # three queries on tables too large for the small-batch form
    assert dispatch(1_000_000, 1000)["variant"] == 6 and dispatch(3_000_000, 3000)["variant"] == 6 and dispatch(1_000_000, 4096)["variant"] == 3
    assert dispatch(10_000_000, 64)["variant"] == 6 and dispatch(10_000_000, 512)["variant"] == 6 and dispatch(10_000_000, 1000)["variant"] == 3
    assert dispatch(30_000_000, 256)["variant"] == 3
    assert dispatch(100_000_000, 1)["variant"] == 6 and dispatch(100_000_000, 3)["variant"] == 6 and dispatch(100_000_000, 4)["variant"] == 3
    # M < 16: the M = 16 kernels over padded rows, at every batch size; never the forms that build their tables from the codebooks
    for M in (8, 4, 12, 1):
        for nq in (1, 3, 64, 1000, 10_000):
            d = dispatch(1_000_000, nq, M=M, D=M * 8)
            assert d["padded_m"] == M and d["variant"] == 3 and d["small"] == 0 and d["qtile"] == 8, (M, nq, d)
    assert dispatch(1_000_000, 10_000, M=16)["padded_m"] == 0
    # k > 128: one query per workgroup, the large selection buffer
    d = dispatch(1_000_000, 100, k=200)
    assert d["variant"] == 0 and d["qtile"] == 1 and d["small"] == 0


def flat_dispatch(metric, D, n_rows, nq, k=10):
    lib = cvt_amd.lib()
    out = (C.c_int * 4)()
    assert lib.cvtmi_flat_describe_dispatch(metric, D, C.c_int64(n_rows), C.c_int64(nq), k, out) == 0
    return dict(f32_stream=out[0], f32_filter=out[1], u8_filter=out[2], u8_stream=out[3])


def test_flat_dispatch_rules_of_round_5():
    """The flat search's routes (cvtmi_flat_describe_dispatch) at the cells the round-5 sweeps fixed (profiles/r05_u8_dispatch_sweep.txt,
    r05_flat_small_tables.txt) and at the structural bounds that must stay."""
    IP, L2F, L2U8 = 0, 1, 2
    # uint8, C3: one stream up to 96 queries (128 on tables under a GB); beyond, the threshold filter of round 6 (u8_filter == 2; it replaced the sample +
    # filter pipeline, u8_filter == 1, wherever its kernels exist: 32 .. 512-d in steps of 32, >= 65 536 rows)
    assert flat_dispatch(L2U8, 512, 10_000_000, 1)["u8_stream"] == 1 and flat_dispatch(L2U8, 512, 10_000_000, 96)["u8_stream"] == 1
    assert flat_dispatch(L2U8, 512, 1_000_000, 128)["u8_stream"] == 1
    for nq in (97, 129, 256, 512, 1000, 4096):
        assert flat_dispatch(L2U8, 512, 10_000_000, nq) == dict(f32_stream=0, f32_filter=0, u8_filter=2, u8_stream=0), nq
    assert flat_dispatch(L2U8, 512, 2_000_000, 256)["u8_filter"] == 2 and flat_dispatch(L2U8, 512, 1_048_576, 129)["u8_filter"] == 2
    assert flat_dispatch(L2U8, 128, 1_048_576, 512)["u8_filter"] == 2 and flat_dispatch(L2U8, 128, 4_000_000, 512)["u8_filter"] == 2
    assert flat_dispatch(L2U8, 512, 10_000_000, 1000, k=100)["u8_filter"] == 2
    # tables from 65 536 rows on (round 6, late): the same batch bound; widths without a streaming kernel (64 / 96 / 192 / 384-d) from two queries on
    assert flat_dispatch(L2U8, 512, 65_536, 129)["u8_filter"] == 2 and flat_dispatch(L2U8, 512, 65_536, 128)["u8_stream"] == 1
    assert flat_dispatch(L2U8, 512, 65_535, 1000)["u8_filter"] == 0 and flat_dispatch(L2U8, 512, 65_535, 1000)["u8_stream"] == 1
    assert flat_dispatch(L2U8, 96, 1_000_000, 2)["u8_filter"] == 2 and flat_dispatch(L2U8, 96, 1_000_000, 1)["u8_filter"] == 0
    assert flat_dispatch(L2U8, 384, 65_536, 8, k=100)["u8_filter"] == 2 and flat_dispatch(L2U8, 320, 1_000_000, 8)["u8_filter"] == 2
    assert flat_dispatch(L2U8, 32, 1_000_000, 2)["u8_filter"] == 2 and flat_dispatch(L2U8, 480, 70_000, 2, k=129)["u8_filter"] == 2
    assert flat_dispatch(L2U8, 100, 1_000_000, 1000)["u8_filter"] == 0 and flat_dispatch(L2U8, 544, 1_000_000, 1000)["u8_filter"] == 0   # (not 32 .. 512 in steps of 32)
    # k = 65 .. 128 from 97 queries on; k = 129 .. 2048 at every batch size (the exact kernels behind took one query per workgroup)
    assert flat_dispatch(L2U8, 128, 10_000_000, 97, k=100)["u8_filter"] == 2 and flat_dispatch(L2U8, 128, 10_000_000, 96, k=100)["u8_stream"] == 1
    assert flat_dispatch(L2U8, 128, 10_000_000, 97, k=64)["u8_filter"] == 2 and flat_dispatch(L2U8, 128, 10_000_000, 96, k=64)["u8_stream"] == 1   # (a GB of rows and more: 97 at any k)
    assert flat_dispatch(L2U8, 128, 2_000_000, 97, k=64)["u8_stream"] == 1
    assert flat_dispatch(L2U8, 512, 2_000_000, 1, k=129)["u8_filter"] == 2 and flat_dispatch(L2U8, 512, 2_000_000, 1000, k=2048)["u8_filter"] == 2
    # ... on tables from 65 536 rows while the sample can fill 1.25 k slots
    assert flat_dispatch(L2U8, 512, 100_000, 1000, k=129)["u8_filter"] == 2 and flat_dispatch(L2U8, 512, 100_000, 100, k=128)["u8_stream"] == 1
    assert flat_dispatch(L2U8, 128, 65_536, 10, k=1000)["u8_filter"] == 2 and flat_dispatch(L2U8, 128, 65_536, 10, k=2048)["u8_filter"] == 0
    assert flat_dispatch(L2U8, 512, 65_535, 1000, k=129) == dict(f32_stream=0, f32_filter=0, u8_filter=0, u8_stream=0)
    # "flat_u8_tfilter" 0 brings the rules of round 5 back: the sample + filter pipeline from rows x width x queries >= 1.3e11 and 524 288 rows on
    try:
        cvt_amd.set_tuning("flat_u8_tfilter", 0)
        for nq in (129, 256, 512, 1000, 4096):
            assert flat_dispatch(L2U8, 512, 10_000_000, nq)["u8_filter"] == 1, nq
        assert flat_dispatch(L2U8, 512, 2_000_000, 256)["u8_filter"] == 1 and flat_dispatch(L2U8, 512, 1_048_576, 129)["u8_filter"] == 0
        assert flat_dispatch(L2U8, 128, 1_048_576, 512)["u8_filter"] == 0 and flat_dispatch(L2U8, 128, 4_000_000, 512)["u8_filter"] == 1
        assert flat_dispatch(L2U8, 512, 400_000, 1000)["u8_filter"] == 0 and flat_dispatch(L2U8, 512, 400_000, 1000)["u8_stream"] == 1
        assert flat_dispatch(L2U8, 512, 10_000_000, 1000, k=100)["u8_filter"] == 0          # k > 64: passes through the stream
        assert flat_dispatch(L2U8, 512, 2_000_000, 1000, k=129) == dict(f32_stream=0, f32_filter=0, u8_filter=0, u8_stream=0)
    finally:
        cvt_amd.set_tuning("flat_u8_tfilter", 1)
    # small tables: the stream from its structural bound of 4096 rows
    assert flat_dispatch(L2U8, 512, 4096, 100)["u8_stream"] == 1 and flat_dispatch(L2U8, 128, 65_536, 16)["u8_stream"] == 1
    assert flat_dispatch(L2U8, 512, 4095, 100) == dict(f32_stream=0, f32_filter=0, u8_filter=0, u8_stream=0)
    assert flat_dispatch(L2U8, 96, 1_000_000, 16)["u8_stream"] == 0                      # widths the matrix-core kernels do not take
    # fp32: the stream at its widths from 32 768 rows (structural: lowered, it returned a wrong list), exact kernels elsewhere
    # (round 6: batches of 65 queries or more -- 16 at the widths the stream kernels do not take -- over 262 144 rows or more take the threshold
    #  filter, route 2)
    for D in (32, 64, 96, 128, 192, 256):
        assert flat_dispatch(L2F, D, 32_768, 1, k=100)["f32_stream"] == 1
        assert flat_dispatch(IP, D, 1_000_000, 1000, k=100)["f32_stream"] == 2
    for D in (32, 64, 96, 128, 192, 256):
        assert flat_dispatch(L2F, D, 1_000_000, 64, k=100)["f32_stream"] == 1 and flat_dispatch(L2F, D, 1_000_000, 97, k=100)["f32_stream"] == 2
        assert flat_dispatch(IP, D, 262_143, 1000, k=100)["f32_stream"] == 1 and flat_dispatch(IP, D, 262_144, 1000, k=128)["f32_stream"] == 2
    assert flat_dispatch(L2F, 64, 1_000_000, 65, k=100)["f32_stream"] == 2 and flat_dispatch(L2F, 128, 1_000_000, 96, k=100)["f32_stream"] == 1
    assert flat_dispatch(L2F, 128, 32_767, 1, k=100)["f32_stream"] == 0
    for D in (102, 2052, 4096):
        assert flat_dispatch(L2F, D, 1_000_000, 100, k=100)["f32_stream"] == 0
    for D in (100, 160, 320, 384, 512, 768, 900, 1024, 1536, 2048):   # round 6: widths only the threshold filter takes (from 16 queries over 262 144 rows on)
        # (from one query on up to 512-d, and wider with k <= 32; wider rows with more neighbours from 16 queries)
        assert flat_dispatch(L2F, D, 1_000_000, 100, k=100)["f32_stream"] == 2 and flat_dispatch(IP, D, 1_000_000, 15, k=100)["f32_stream"] == (2 if D <= 512 else 0)
        assert flat_dispatch(IP, D, 1_000_000, 1, k=32)["f32_stream"] == 2 and flat_dispatch(IP, D, 1_000_000, 16, k=100)["f32_stream"] == 2
        # (late round 6: these widths from 65 536 rows on -- nothing but the exact kernels is there; rows wider than 512-d with k > 32 from 129 queries)
        assert flat_dispatch(L2F, D, 65_535, 1000, k=100)["f32_stream"] == 0 and flat_dispatch(L2F, D, 65_536, 1000, k=100)["f32_stream"] == 2
        assert flat_dispatch(L2F, D, 100_000, 100, k=100)["f32_stream"] == (2 if D <= 512 else 0) and flat_dispatch(L2F, D, 100_000, 100, k=32)["f32_stream"] == 2
    # (round 6: k = 129 .. 2048 take the threshold filter at every batch size; the stream kernels stop at 128)
    assert flat_dispatch(L2F, 128, 1_000_000, 100, k=129)["f32_stream"] == 2 and flat_dispatch(L2F, 128, 1_000_000, 1, k=2048)["f32_stream"] == 2
    # ... on tables from 65 536 rows and 48 k (the sample still fills its slots); k <= 128 leaves such tables to the stream / exact kernels
    assert flat_dispatch(L2F, 128, 100_000, 100, k=129)["f32_stream"] == 2 and flat_dispatch(L2F, 128, 100_000, 100, k=128)["f32_stream"] == 1
    assert flat_dispatch(L2F, 128, 100_000, 100, k=2048)["f32_stream"] == 2 and flat_dispatch(L2F, 128, 98_000, 100, k=2048)["f32_stream"] == 0
    assert flat_dispatch(L2F, 128, 65_535, 100, k=129)["f32_stream"] == 0
    assert flat_dispatch(IP, 1024, 200_000, 1, k=129)["f32_stream"] == 0 and flat_dispatch(IP, 1024, 200_000, 16, k=129)["f32_stream"] == 2
