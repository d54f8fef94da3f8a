"""CPU: the static schedule of the fused MLP launch (ani_kernels_mlpf.hip:fused_schedule, through ani_debug_fused_schedule).

Which workgroup runs which tiles is decided on the host by multifit (the smallest makespan for which first-fit-decreasing
packs the tiles into the CUs).  Properties: every item exactly once; a workgroup's list in descending cost; makespan never
above list scheduling's (longest first, next free workgroup -- what drawing tiles from a counter does) and within multifit's 13/11
of a lower bound (average load, largest item, ceil(n / bins) of the cheapest), give or take one item; the benchmark case -- 521 + 261 tiles of cost
440 and 296 on 256 CUs -- packs into three of the costly tiles' time where the counter needs 3.35."""
import ctypes as C
import heapq

import numpy as np
import pytest

from lammps_ani_amd import ani_hip


def _schedule(count, cost, bins):
    lib = ani_hip.lib()
    lib.ani_debug_fused_schedule.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    count = np.ascontiguousarray(count, dtype=np.int32)
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    items = np.full(int(count.sum()), -1, dtype=np.int32)
    off = np.zeros(bins + 1, dtype=np.int32)
    mk = C.c_double()
    rc = lib.ani_debug_fused_schedule(len(count), count.ctypes.data, cost.ctypes.data, bins, items.ctypes.data, off.ctypes.data,
                                      C.addressof(mk))
    assert rc == 0
    return items, off, mk.value


def _list_scheduling(count, cost, bins):
    loads = [0.0] * bins
    heapq.heapify(loads)
    for j in np.argsort(-np.asarray(cost)):
        for _ in range(int(count[j])):
            heapq.heappush(loads, heapq.heappop(loads) + float(cost[j]))
    return max(loads)


@pytest.mark.parametrize("count,cost,bins", [([521, 261], [440.0, 296.0], 256), ([66, 33], [440.0, 296.0], 256), ([1], [5.0], 8),
                                             ([424, 216], [440.0, 296.0], 256), ([300, 200, 100, 7], [284.0, 284.0, 284.0, 91.0], 256),
                                             ([5000, 3000], [440.0, 296.0], 256), ([3, 2, 9], [1.0, 7.5, 2.25], 4)])
def test_schedule_is_a_partition_and_beats_the_counter(count, cost, bins):
    items, off, mk = _schedule(count, cost, bins)
    n = int(np.sum(count))
    assert sorted(items.tolist()) == list(range(n))
    assert off[0] == 0 and off[-1] == n and np.all(np.diff(off) >= 0)
    type_of = np.repeat(np.arange(len(count)), count)
    c = np.asarray(cost)[type_of]
    loads = [c[items[off[b]:off[b + 1]]].sum() for b in range(bins)]
    assert abs(max(loads) - mk) < 1e-9
    for b in range(bins):   # costliest first inside a workgroup
        cb = c[items[off[b]:off[b + 1]]]
        assert np.all(np.diff(cb) <= 1e-12)
    # lower bounds: the average load, the largest item, and the cheapest way to give some workgroup ceil(n / bins) items
    cmin = min(co for co, k in zip(cost, count) if k > 0)
    lower = max(float(np.dot(count, cost)) / bins, max(co for co, k in zip(cost, count) if k > 0), cmin * -(-n // bins))
    assert mk <= _list_scheduling(count, cost, bins) + 1e-9
    assert lower - 1e-9 <= mk <= 13.0 / 11.0 * lower + max(cost)


def test_benchmark_case_packs_into_three_rounds():
    _, _, mk = _schedule([521, 261], [440.0, 296.0], 256)
    assert abs(mk - 3 * 440.0) < 1e-6
    assert _list_scheduling([521, 261], [440.0, 296.0], 256) > 3.3 * 440.0


def _schedule_halves(count, cost, bins, mode, ratio=0.75):
    lib = ani_hip.lib()
    lib.ani_debug_fused_schedule_halves.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 5
    count = np.ascontiguousarray(count, dtype=np.int32)
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    n = int(count.sum())
    items = np.full(2 * max(n, 1), -1, dtype=np.int32)
    off = np.zeros(bins + 1, dtype=np.int32)
    split = np.zeros(len(count), dtype=np.int32)
    nout = C.c_int()
    mk = C.c_double()
    rc = lib.ani_debug_fused_schedule_halves(len(count), count.ctypes.data, cost.ctypes.data, ratio, bins, mode, split.ctypes.data,
                                             items.ctypes.data, off.ctypes.data, C.addressof(nout), C.addressof(mk))
    assert rc == 0
    return items[: nout.value], off, split, mk.value


@pytest.mark.parametrize("count,cost,bins", [([521, 261], [440.0, 296.0], 256), ([261, 131], [440.0, 296.0], 256), ([66, 33], [440.0, 296.0], 256),
                                             ([424, 216], [440.0, 296.0], 256), ([1], [5.0], 8), ([300, 200, 100, 7], [284.0, 284.0, 284.0, 91.0], 256),
                                             ([5000, 3000], [440.0, 296.0], 256), ([3, 2, 9], [1.0, 7.5, 2.25], 4)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_schedule_with_half_items_covers_every_item_once(count, cost, bins, mode):
    """Every item runs exactly once: whole, or as its two halves (ids n + 2 i + h); the halves are the LAST split[j] items of a type;
    the searched split never makes the schedule longer than whole items only, and shortens the 50 001-atom case (261 + 131 tiles: five
    costly tiles beyond one round) by more than a tenth."""
    ratio = 0.75
    items, off, split, mk = _schedule_halves(count, cost, bins, mode, ratio)
    n = int(np.sum(count))
    first = np.concatenate([[0], np.cumsum(count)])
    whole = sorted(int(i) for i in items if i < n)
    halves = sorted(int(i) - n for i in items if i >= n)
    cut = sorted(halves[0::2])
    assert halves == sorted([2 * (i // 2) + h for i in halves[0::2] for h in (0, 1)])      # both halves of every cut item
    assert [i // 2 for i in cut] == [i for j in range(len(count)) for i in range(first[j + 1] - split[j], first[j + 1])]
    assert sorted(whole + [i // 2 for i in cut]) == list(range(n))
    assert off[0] == 0 and off[-1] == len(items) and np.all(np.diff(off) >= 0)
    if mode == 0:
        assert split.sum() == 0
    if mode == 2:
        assert np.array_equal(split, np.asarray(count))
    # loads
    type_of = np.repeat(np.arange(len(count)), count)
    c = np.asarray(cost, dtype=float)

    def item_cost(i):
        return c[type_of[i]] if i < n else ratio * c[type_of[(i - n) // 2]]
    loads = [sum(item_cost(int(i)) for i in items[off[b]:off[b + 1]]) for b in range(bins)]
    assert max(loads) <= mk * (1 + 1e-9) + 1e-9
    _, _, _, mk0 = _schedule_halves(count, cost, bins, 0, ratio)
    if mode == 1:
        assert mk <= mk0 * (1 + 1e-9)
        if list(count) == [261, 131]:
            assert mk < 0.9 * mk0 and split.sum() > 0
