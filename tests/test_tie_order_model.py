"""The formulation of libstdc++'s introsort that ``cifseeds_tie_kernel`` (openpifpaf_amd/csrc/cifseeds.hip) executes,
as a numpy model, against ``std::sort`` itself (the oracle's, the comparator and element type of cif_seeds.cpp:94).

The kernel does not walk a Hoare partition with two pointers: the k-th element the left scan stops at (``!(x > pivot)``) is
exchanged with the k-th the right scan stops at (``!(pivot > x)``) while it lies to its left; only segments that hold a
score occurring twice are followed; and the final insertion sort is a stable placement inside every segment of at most
16 elements.  This file pins those three statements on the CPU; tests/test_gpu_ties.py pins the kernel."""
import numpy as np
import pytest


@pytest.fixture(scope='module')
def port():
    from oracle import port as p
    return p


def heapsort_segment(a, v, first, last):
    """``std::__partial_sort(first, last, last, comp)`` -- what ``__introsort_loop`` does with a segment that reaches its depth
    limit: ``__make_heap`` + ``__sort_heap`` (bits/stl_heap.h: ``__adjust_heap``, ``__push_heap``, ``__pop_heap``), comp(x, y) =
    x.v > y.v.  In place on a[first:last] (original positions); scores v."""
    comp = lambda x, y: v[x] > v[y]

    def push_heap(hole, top, value):
        parent = (hole - 1) // 2
        while hole > top and comp(a[first + parent], value):
            a[first + hole] = a[first + parent]
            hole = parent
            parent = (hole - 1) // 2
        a[first + hole] = value

    def adjust_heap(hole, length, value):
        top = hole
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if comp(a[first + child], a[first + child - 1]):
                child -= 1
            a[first + hole] = a[first + child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            a[first + hole] = a[first + child - 1]
            hole = child - 1
        push_heap(hole, top, value)

    length = last - first
    if length >= 2:                                       # __make_heap
        parent = (length - 2) // 2
        while True:
            adjust_heap(parent, length, a[first + parent])
            if parent == 0:
                break
            parent -= 1
    end = last                                            # __sort_heap
    while end - first > 1:
        end -= 1
        value = a[end]
        a[end] = a[first]
        adjust_heap(0, end - first, value)


def introsort_model(v, follow_only_tied=True, heapsort=False):
    """-> perm (original positions in final order); where std::sort switches to heapsort (a segment at the depth limit): None,
    or with ``heapsort`` the segment is heap-sorted like ``std::__partial_sort`` does it (round 6: the kernel does)."""
    v = np.asarray(v, dtype=np.float32)
    n = len(v)
    a = np.arange(n)                                      # a[k]: original position of the element at k
    if n == 0:
        return a
    vals, counts = np.unique(v, return_counts=True)
    tied_value = set(vals[counts > 1].tolist())
    tied = np.array([x in tied_value for x in v.tolist()])
    marks = {0}
    todo = [(0, n, 2 * (int(n).bit_length() - 1))] if n > 16 else []
    while todo:
        first, last, depth = todo.pop()
        if follow_only_tied and not tied[a[first:last]].any():
            continue                                      # nobody asks where these end up: each has a rank of its own
        if depth == 0:
            if not heapsort:
                return None
            heapsort_segment(a, v, first, last)           # sorted now: every element of it is a segment of its own
            marks.update(range(first, last + 1))
            continue
        x = lambda k: v[a[k]]
        A, B, C = first + 1, first + (last - first) // 2, last - 1     # __move_median_to_first
        if x(A) > x(B):
            t = B if x(B) > x(C) else (C if x(A) > x(C) else A)
        elif x(A) > x(C):
            t = A
        elif x(B) > x(C):
            t = C
        else:
            t = B
        a[first], a[t] = a[t], a[first]
        pivot = v[a[first]]
        seg = v[a[first + 1:last]]
        left = first + 1 + np.nonzero(~(seg > pivot))[0]              # stops of the scan from the left, in order
        right = (first + 1 + np.nonzero(~(pivot > seg))[0])[::-1]     # stops of the scan from the right, in order
        m = 0
        while m < len(left) and m < len(right) and left[m] < right[m]:
            m += 1
        for k in range(m):
            a[left[k]], a[right[k]] = a[right[k]], a[left[k]]
        if m == 0:
            cut = int(left[0])
        else:
            cut = int(right[m - 1])
            if m < len(left) and left[m] < cut:
                cut = int(left[m])
        marks.add(cut)
        if cut - first > 16:
            todo.append((first, cut, depth - 1))
        if last - cut > 16:
            todo.append((cut, last, depth - 1))
    # final insertion sort: a tied element goes behind the larger and the equal-and-earlier elements of the segment
    # between the mark at or before it and the next mark; everything else sits at its rank, wherever the loop left it
    out = np.full(n, -1, dtype=np.int64)
    bounds = sorted(marks | {n})
    order = np.argsort(-v, kind='stable')
    rank_of = np.empty(n, dtype=np.int64)
    rank_of[order] = np.arange(n)
    for k in range(n):
        e = a[k]
        if not tied[e]:
            out[rank_of[e]] = e                           # a score of its own: its rank is its place
            continue
        import bisect
        at = bisect.bisect_right(bounds, k)
        ls, le = bounds[at - 1], bounds[at]
        assert le - ls <= 16, (ls, le)
        seg = v[a[ls:le]]
        pos = ls + int((seg > v[e]).sum()) + int((seg[:k - ls] == v[e]).sum())
        out[pos] = e
    assert (out >= 0).all()
    return out


def cases():
    rng = np.random.default_rng(0)
    for trial in range(120):
        n = int(rng.choice([1, 2, 5, 16, 17, 18, 33, 64, 65, 100, 257, 1000, 3000]))
        kind = trial % 6
        if kind == 0:
            v = rng.integers(0, 4, n) / 4
        elif kind == 1:
            v = rng.integers(0, 50, n) / 50
        elif kind == 2:
            v = np.round(rng.random(n) * 256) / 256
        elif kind == 3:
            v = np.full(n, 0.5)
        elif kind == 4:                                   # a few equal pairs among distinct scores (float32 fields)
            v = rng.random(n)
            for _ in range(min(3, n // 2)):
                i, j = rng.integers(0, n, 2)
                v[i] = v[j]
        else:
            v = rng.random(n)
        v = v.astype(np.float32)
        if trial % 7 == 0:
            v = np.sort(v)
        if trial % 11 == 0:
            v = np.sort(v)[::-1].copy()
        yield trial, v


def test_stop_pairing_and_segment_placement_equal_std_sort(port):
    checked = 0
    for trial, v in cases():
        want = port.sorted_seed_order(v)
        for follow_only_tied in (False, True):
            got = introsort_model(v, follow_only_tied)
            assert got is not None, 'depth limit reached in trial %d (not expected for these inputs)' % trial
            assert np.array_equal(got, want), 'trial %d (n %d, follow_only_tied %s): first difference at rank %d' % (
                trial, len(v), follow_only_tied, int(np.argmax(got != want)))
            checked += 1
    assert checked == 240


def test_std_sort_is_not_stable_here(port):
    """The reason for all this: among equal scores std::sort's order is neither the input order nor its reverse."""
    v = (np.arange(200) % 4).astype(np.float32)
    perm = port.sorted_seed_order(v)
    stable = np.argsort(-v, kind='stable')
    assert not np.array_equal(perm, stable)
    assert sorted(perm.tolist()) == list(range(200)) and (np.diff(v[perm]) <= 0).all()
