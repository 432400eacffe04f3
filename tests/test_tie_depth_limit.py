"""Introsort's depth limit and the tie pass.

``std::sort`` (cif_seeds.cpp:94) is libstdc++'s introsort: quicksort with a median-of-three pivot that falls back to
heapsort (``std::__partial_sort``) for a segment that is still longer than 16 elements after ``2 * floor(log2 n)``
partition levels.  The device pass that reproduces the reference's order of EQUAL scores (csrc/cifseeds_tie.hpp) follows
the partitions but not the heapsort.  That matters only where the heapsort would have to order equal scores: a segment
WITHOUT equal scores ends up sorted whatever sorts it.  So

* a sequence that drives introsort to its depth limit, with equal scores only among the elements the early partitions
  split off: the pass reproduces ``std::sort`` exactly (``seed_ties`` = 1);
* the same sequence with two equal scores inside the segment that reaches the limit: the image keeps the cell-index order
  of equal scores and says so (``seed_ties`` = -1, the documented deviation of INTEGRATION.md section 4).

The sequence is McIlroy's adversary ("A killer adversary for quicksort", 1999) run against a scalar restatement of
libstdc++'s ``__introsort_loop`` / ``__move_median_to_first`` / ``__unguarded_partition`` (bits/stl_algo.h); the CPU tests pin
that restatement and the numpy model of tests/test_tie_order_model.py against ``std::sort`` itself (the oracle's), the GPU
test pins the kernel."""
import numpy as np
import pytest

from test_tie_order_model import introsort_model


@pytest.fixture(scope='module')
def port():
    from oracle import port as p
    return p


# ---- libstdc++'s introsort on a list of items with a comparator (comp(a, b): a goes before b)
def _move_median_to_first(a, comp, result, ia, ib, ic):
    if comp(a[ia], a[ib]):
        if comp(a[ib], a[ic]):
            t = ib
        elif comp(a[ia], a[ic]):
            t = ic
        else:
            t = ia
    elif comp(a[ia], a[ic]):
        t = ia
    elif comp(a[ib], a[ic]):
        t = ic
    else:
        t = ib
    a[result], a[t] = a[t], a[result]


def _unguarded_partition(a, comp, first, last, pivot):
    while True:
        while comp(a[first], a[pivot]):
            first += 1
        last -= 1
        while comp(a[pivot], a[last]):
            last -= 1
        if not first < last:
            return first
        a[first], a[last] = a[last], a[first]
        first += 1


def quicksort_loop(a, comp, first, last, depth_limit=None):
    """``__introsort_loop`` without (depth_limit None) or with its depth limit; -> the segments (first, last) that reached the
    limit (where std::sort switches to heapsort); segments of at most 16 elements are left as they are."""
    hit = []
    todo = [(first, last, depth_limit)]
    while todo:
        first, last, depth = todo.pop()
        while last - first > 16:
            if depth is not None:
                if depth == 0:
                    hit.append((first, last))
                    break
                depth -= 1
            mid = first + (last - first) // 2
            _move_median_to_first(a, comp, first, first + 1, mid, last - 1)
            cut = _unguarded_partition(a, comp, first + 1, last, first)
            todo.append((cut, last, depth))
            last = cut
    return hit


def killer(n):
    """-> ranks (0 = goes first) of a sequence of n distinct elements on which the quicksort above takes ~n / 2 levels."""
    gas = n
    val = [gas] * n
    state = {'solid': 0, 'candidate': 0}

    def comp(x, y):
        if val[x] == gas and val[y] == gas:
            if x == state['candidate']:
                val[x] = state['solid']
            else:
                val[y] = state['solid']
            state['solid'] += 1
        if val[x] == gas:
            state['candidate'] = x
        elif val[y] == gas:
            state['candidate'] = y
        return val[x] < val[y]
    items = list(range(n))
    quicksort_loop(items, comp, 0, n)
    for i in range(n):                                    # whatever stayed gas: any order
        if val[i] == gas:
            val[i] = state['solid']
            state['solid'] += 1
    assert sorted(val) == list(range(n))
    return np.asarray(val)


def killer_scores(n, tie):
    """Seed scores in raster order: descending comparator (cif_seeds.cpp:94: a.v > b.v), so rank 0 = the largest score.
    ``tie`` = 'shallow': two equal scores among the elements the first partitions split off; 'deep': inside the segment that
    reaches the depth limit.  -> (float32 scores, positions of the two equal ones)"""
    rank = killer(n)
    v = (0.95 - 0.7 * rank / n).astype(np.float32)
    assert len(np.unique(v)) == n
    # where does the depth limit strike?
    a = list(range(n))
    hit = quicksort_loop(a, lambda x, y: v[x] > v[y], 0, n, 2 * (int(n).bit_length() - 1))
    assert hit, 'the sequence does not reach the depth limit'
    deep = set()
    for f, l in hit:
        deep.update(a[f:l])
    assert len(deep) > n // 2, (len(deep), n)
    if tie == 'deep':
        i, j = sorted(deep)[len(deep) // 3], sorted(deep)[2 * len(deep) // 3]
    else:
        shallow = sorted(set(range(n)) - deep)
        assert len(shallow) >= 8
        i, j = shallow[1], shallow[len(shallow) // 2]
    v[j] = v[i]
    return v, (i, j)


def killer_scores_many(n, groups=6):
    """Like ``killer_scores(n, 'deep')`` with ``groups`` more pairs and triples of equal scores spread over the segment that
    reaches the limit (spread evenly: a handful of changed scores leaves the adversary's pivots what they were)."""
    v, _ = killer_scores(n, 'deep')
    a = list(range(n))
    hit = quicksort_loop(a, lambda x, y: v[x] > v[y], 0, n, 2 * (int(n).bit_length() - 1))
    deep = sorted(set(e for f, l in hit for e in a[f:l]))
    L = len(deep)
    for g in range(groups):
        i, j, k = (deep[(3 * g + t) * L // (3 * groups + 2)] for t in (1, 2, 3))
        v[j] = v[i]
        if g % 2:
            v[k] = v[i]
    return v


def test_the_scalar_restatement_sorts_like_std_sort(port):
    rng = np.random.default_rng(5)
    for n in (17, 100, 1000, 3000):
        v = (np.round(rng.random(n) * 64) / 64).astype(np.float32)        # plenty of equal scores
        a = list(range(n))
        hit = quicksort_loop(a, lambda x, y: v[x] > v[y], 0, n, 2 * (int(n).bit_length() - 1))
        assert not hit
        # __final_insertion_sort: stable inside what the loop left; emulate with a stable sort of every run between cuts --
        # instead compare through the model, which is pinned against std::sort element by element
        want = port.sorted_seed_order(v)
        got = introsort_model(v, follow_only_tied=False)
        assert got is not None and np.array_equal(got, want)


def test_killer_reaches_the_depth_limit_and_the_model_says_what_matters(port):
    """A tie the early partitions split off: the model (and the kernel, below) reproduce std::sort although a segment without
    equal scores reaches the depth limit; a tie inside that segment: the model gives up (None), like the kernel (-1)."""
    for n in (1500, 5000):
        v, (i, j) = killer_scores(n, 'shallow')
        assert introsort_model(v, follow_only_tied=False) is None             # the limit IS reached ...
        got = introsort_model(v, follow_only_tied=True)                        # ... by a segment nobody has to order
        assert got is not None and np.array_equal(got, port.sorted_seed_order(v))
        v, (i, j) = killer_scores(n, 'deep')
        assert introsort_model(v, follow_only_tied=True) is None
        # round 6: the segment at the limit is heap-sorted like std::__partial_sort does it (__make_heap + __sort_heap):
        # std::sort's order, the equal pair included -- with every segment followed, and with only the tied ones
        perm = port.sorted_seed_order(v)
        assert (np.diff(v[perm]) <= 0).all()
        for only_tied in (False, True):
            got = introsort_model(v, follow_only_tied=only_tied, heapsort=True)
            assert got is not None and np.array_equal(got, perm), (n, only_tied)
    # heapsort with SEVERAL groups of equal scores inside the segment at the limit
    for n in (1500, 5000):
        v = killer_scores_many(n)
        assert introsort_model(v, follow_only_tied=True) is None, 'the sequence no longer reaches the limit with its ties'
        got = introsort_model(v, follow_only_tied=True, heapsort=True)
        assert got is not None and np.array_equal(got, port.sorted_seed_order(v)), n


def _field_from_scores(v, H=None):
    """One CIF field whose cells, in raster order, carry confidence v[k] (inactive behind them): with the no-rescore
    ablation (cif_seeds.cpp:52-53) the seed scores ARE these confidences."""
    n = len(v)
    W = 81
    H = (n + W - 1) // W
    cif = np.zeros((2, 5, H, W), dtype=np.float32)       # (a second, empty field: the skeleton needs two joints)
    conf = np.zeros(H * W, dtype=np.float32)
    conf[:n] = v
    cif[0, 1] = conf.reshape(H, W)
    jj, ii = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    cif[:, 2], cif[:, 3], cif[:, 4] = ii, jj, 1.0
    return cif


@pytest.mark.gpu
def test_kernel_flags_only_a_tied_segment_at_the_depth_limit(port):
    torch = pytest.importorskip('torch')
    from openpifpaf_amd import native
    assert torch.cuda.is_available()
    skel0 = np.asarray([[0, 1]], dtype=np.int64)
    native.CifSeeds.set_ablation_no_rescore(True)
    port_params = port.default_params(ablation_cifseeds_no_rescore=1)
    try:
        # one LDS block / the split first sort, both inside the LDS arrays of the pass; 9000: the arrays in global memory
        for n in (1500, 5000, 9000):
            for tie, want_state in ((('shallow', 1), ('deep', 1), ('many', 1)) if n < 9000 else (('many', 1),)):      # round 6: the heapsort branch is reproduced
                v, (i, j) = killer_scores(n, tie) if tie != 'many' else (killer_scores_many(n), (0, 0))
                cif = _field_from_scores(v)
                caf = np.zeros((1, 8, cif.shape[2], cif.shape[3]), dtype=np.float32)
                dec = native.CifCaf(2, torch.from_numpy(skel0))
                dec.call_batch(torch.from_numpy(cif[None]).cuda(), 8, torch.from_numpy(caf[None]).cuda(), 8)
                state = int(dec.workspace_view('seed_ties', torch.int32)[0].cpu())
                assert state == want_state, (n, tie, state)
                n_seeds = int(dec.workspace_view('seed_count', torch.int32)[0].cpu())
                assert n_seeds == n
                got_v = dec.workspace_view('seed_vxys', torch.float32)[:4 * n].view(n, 4).cpu().numpy()
                hr = port.cifhr_accumulate(cif, 8)
                if want_state == 1:                  # exactly std::sort's order
                    want_f, want_v = port.cifseeds(cif, 8, hr, params=port_params)
                    assert np.array_equal(got_v, want_v), (n, tie)
                else:                                # the documented fallback: equal scores by cell index (raster position)
                    port.set_seed_tie_rule(1)
                    try:
                        want_f, want_v = port.cifseeds(cif, 8, hr, params=port_params)
                    finally:
                        port.set_seed_tie_rule(0)
                    assert np.array_equal(got_v, want_v), (n, tie)
                    k = int(np.nonzero(got_v[:, 0] == v[i])[0][0])
                    assert got_v[k, 0] == got_v[k + 1, 0] and (got_v[k, 2] * 81 + got_v[k, 1]) < (got_v[k + 1, 2] * 81 + got_v[k + 1, 1])
    finally:
        native.CifSeeds.set_ablation_no_rescore(False)
