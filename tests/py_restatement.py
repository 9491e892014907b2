"""TEST INFRASTRUCTURE: a second, independent restatement of the reference's HNSW build and search — plain Python,
written from the Rust source (/root/reference/instant-distance/src/lib.rs and types.rs), NOT from oracle/idist_oracle.c.

Purpose: the reference cannot be built here (no rustc), so the C oracle is pinned only behaviourally by the reference's
own known answers.  Two restatements made separately from the same source and agreeing array for array on the graph they
build is the strongest extra evidence available; tests/test_oracle_restatement.py compares them.  The
`select_heuristic(None)` splice depends on the probe order of std's `slice::binary_search_by` (its comparator is
reversed): restated here from the std source as of Rust 1.82+ (branch-free form) — best effort, like
the oracle's.
`Heuristic::extend_candidates` follows the code with its locks ignored (upstream it deadlocks: lib.rs:649 read-locks the
node write-locked at :438), i.e. `node.set(i, pid)` (:516) is visible to later reads of the new node's row.

Distances come in as functions / tables (the arithmetic is pinned elsewhere: tests/test_oracle_golden.py).
Sizes of a few dozen points only: everything is a Python loop.
"""
import bisect
import heapq

import numpy as np

M = 32                      # lib.rs:787
INVALID = 0xFFFFFFFF        # types.rs: INVALID = PointId(u32::MAX)


def layer_sizes(n, ml):
    """lib.rs:238-250: f32 multiply, truncation; returns [(size, cumulative)] top layer first."""
    sizes, num = [], n
    while True:
        nxt = int(np.float32(num) * np.float32(ml))
        if nxt < M:
            break
        sizes.append((num - nxt, num))
        num = nxt
    sizes.append((num, num))
    sizes.reverse()
    return sizes


def valid_prefix(row):
    """NearestIter (types.rs:158-192): stops at the first INVALID slot."""
    out = []
    for pid in row:
        if pid == INVALID:
            break
        out.append(int(pid))
    return out


class Search:
    """lib.rs:560-574; Candidate order (distance, pid): types.rs:229-234."""

    def __init__(self):
        self.visited = set()
        self.candidates = []        # min-heap of (distance, pid)  (BinaryHeap<Reverse<Candidate>>)
        self.nearest = []           # sorted, nearest first
        self.working = []
        self.discarded = []
        self.ef = 1

    def reset(self):                # :740-755
        self.visited.clear()
        self.candidates = []
        self.nearest = []
        self.working = []
        self.discarded = []

    def push(self, pid, dist_to):   # :704-720
        if pid in self.visited:
            return
        self.visited.add(pid)
        new = (dist_to(pid), pid)
        idx = bisect.bisect_left(self.nearest, new)       # binary_search: Err(idx), keys are distinct
        if idx >= self.ef:
            return
        self.nearest.insert(idx, new)
        heapq.heappush(self.candidates, new)

    def search(self, dist_to, row_of, links):             # :598-614
        while self.candidates:
            cand = heapq.heappop(self.candidates)
            if self.nearest and cand[0] > self.nearest[-1][0]:
                break
            for pid in row_of(cand[1])[:links]:
                self.push(pid, dist_to)
            del self.nearest[self.ef:]

    def cull(self):                 # :729-737
        self.candidates = list(self.nearest)
        heapq.heapify(self.candidates)
        self.visited = {pid for _, pid in self.nearest}

    def select_heuristic(self, dist_to, pair_dist, row_of, extend, keep_pruned):   # :636-698
        self.working = []
        for cand in self.nearest:
            self.working.append(cand)
            if extend:                                     # :648-659
                for hop in row_of(cand[1]):
                    if hop in self.visited:
                        continue
                    self.visited.add(hop)
                    self.working.append((dist_to(hop), hop))
        if extend:
            self.working.sort()                            # :662-664
        self.nearest = []
        self.discarded = []
        for cand in self.working:                          # :668-685
            if len(self.nearest) >= 2 * M:
                break
            is_nearest = not any(pair_dist(cand[1], res[1]) < cand[0] for res in self.nearest)
            (self.nearest if is_nearest else self.discarded).append(cand)
        self.working = []                                  # the Drain is dropped
        if keep_pruned:                                    # :687-695
            for cand in self.discarded:
                if len(self.nearest) >= 2 * M:
                    break
                self.nearest.append(cand)
            self.discarded = []
        return self.nearest

    def add_neighbor_heuristic(self, new, current, dist_to, pair_dist, row_of, extend, keep_pruned):   # :616-631
        self.reset()
        self.push(new, dist_to)
        for pid in current:
            self.push(pid, dist_to)
        return self.select_heuristic(dist_to, pair_dist, row_of, extend, keep_pruned)


def rewrite(row, pids):
    """ZeroNode::rewrite, types.rs:88-98."""
    it = iter(pids)
    for slot in range(len(row)):
        nxt = next(it, None)
        if nxt is not None:
            row[slot] = nxt
        elif row[slot] != INVALID:
            row[slot] = INVALID
        else:
            break


def binary_search_by(length, f):
    """std::slice::binary_search_by (Rust >= 1.82): f(index) in {-1 Less, 0 Equal, +1 Greater}; returns the Ok/Err index."""
    size, base = length, 0
    if size == 0:
        return 0
    while size > 1:
        half = size // 2
        mid = base + half
        base = base if f(mid) > 0 else mid
        size -= half
    c = f(base)
    return base if c == 0 else base + (1 if c < 0 else 0)


def zero_insert(row, idx, pid):
    """ZeroNode::insert, types.rs:100-113."""
    if idx >= len(row):
        return
    if row[idx] != INVALID:
        row[idx + 1:] = row[idx:-1].copy()                 # copy_within(idx..end, idx + 1), end = len - 1
    row[idx] = pid


def build(D, n, ml, ef_construction, extend=False, keep_pruned=True, heuristic=True):
    """Hnsw::new with one thread (lib.rs:209-345; Construction::insert :437-528) on points given by their pairwise
    distance table D (n x n, already in PointId order).  Returns (zero [n][64], layers [[len][32]])."""
    zero = np.full((n, 2 * M), INVALID, dtype=np.uint32)
    if n == 0:
        return zero, []
    sizes = layer_sizes(n, ml)
    num_layers = len(sizes)
    top = num_layers - 1
    ranges = []
    for i, (size, cumulative) in enumerate(sizes):         # :275-281
        start = cumulative - size
        ranges.append((num_layers - i - 1, max(start, 1), cumulative))
    layers = [None] * top
    search, insertion = Search(), Search()                 # the pool hands out one pair, :439
    pair = lambda a, b: float(D[a][b])                     # noqa: E731
    row_zero = lambda pid: valid_prefix(zero[pid])         # noqa: E731

    def insert(new, layer):                                # :437-528
        insertion.ef = ef_construction
        dist_new = lambda pid: float(D[new][pid])          # noqa: E731
        search.reset()
        search.push(0, dist_new)
        num = 2 * M if layer == 0 else M
        for cur in range(top, -1, -1):                     # self.top.descend()
            search.ef = ef_construction if cur <= layer else 1
            if cur > layer:
                upper = layers[cur - 1]
                search.search(dist_new, lambda pid: valid_prefix(upper[pid]), num)
                search.cull()
            else:
                search.search(dist_new, row_zero, num)
                break
        if heuristic:
            found = search.select_heuristic(dist_new, pair, row_zero, extend, keep_pruned)
        else:
            found = search.nearest[:2 * M]                 # select_simple, :466-469, :758-760
        for i, (distance, pid) in enumerate(found):        # :481-516
            if heuristic:
                dist_old = lambda x, pid=pid: float(D[pid][x])   # noqa: E731
                res = insertion.add_neighbor_heuristic(new, row_zero(pid), dist_old, pair, row_zero, extend, keep_pruned)
                rewrite(zero[pid], [p for _, p in res])
            else:                                          # :497-515: the comparator is target.cmp(element), :510
                def f(slot, pid=pid, distance=distance):
                    third = int(zero[pid][slot])
                    if third == INVALID:
                        return 1                           # Ordering::Greater, :505-508
                    other = float(D[pid][third])
                    return (distance > other) - (distance < other)
                zero_insert(zero[pid], binary_search_by(2 * M, f), new)
            zero[new][i] = pid                             # node.set(i, pid)

    for layer, lo, hi in ranges:                           # :304-329
        for pid in range(lo, hi):
            insert(pid, layer)
        if layer != 0:
            layers[layer - 1] = zero[:hi, :M].copy()       # UpperNode::from_zero, types.rs:66-70
    return zero, layers


def search_index(zero, layers, dist_to, ef_search):
    """Hnsw::search, lib.rs:352-383: [(distance, pid)] nearest first."""
    s = Search()
    if len(zero) == 0:
        return []
    s.push(0, dist_to)
    for cur in range(len(layers), -1, -1):
        if cur == 0:
            s.ef = ef_search
            s.search(dist_to, lambda pid: valid_prefix(zero[pid]), 2 * M)
        else:
            s.ef = 1
            upper = layers[cur - 1]
            s.search(dist_to, lambda pid: valid_prefix(upper[pid]), M)
            s.cull()
    return s.nearest
