"""CPU restatement of k_chunk's bottom-up build of the in-chunk nodes (csrc/lbvh.hip, "the climb"), checked against the oracle's
Karras tree (collision.cl:81-121 as restated in oracle/collision_oracle.c): for every internal node whose range lies inside one
chunk of 256 sorted leaves the climb must name the same other end and the same split, and it must name no other node.
Used by tests/test_host_logic.py::test_climb_names_the_in_chunk_nodes_of_the_karras_tree; `python tests/analysis/climb_prototype.py`
runs a longer sweep."""
import numpy as np

C = 256


def clz32(x):
    return 32 - int(x).bit_length()


def climb(codes, order_rng):
    """-> {node: (other_end, split)} for the nodes the climb names; leaves of a chunk arrive in a random order."""
    n = len(codes)

    def adj(q):   # delta(q, q + 1): common prefix of the keys (code, position); -1 outside
        if q < 0 or q + 1 >= n:
            return -1
        a, b = int(codes[q]), int(codes[q + 1])
        return clz32(a ^ b) if a != b else 32 + clz32(q ^ (q + 1))

    named = {}
    for c0 in range(0, n, C):
        m = min(C, n - c0)
        meet = {}
        for t in order_rng.permutation(m):
            l = r = int(t)
            is_leaf, split = True, None
            while True:
                dl, dr = adj(c0 + l - 1), adj(c0 + r)
                right = dr > dl                       # the parent continues to the right: this range is its LEFT child, Karras' node r
                if not is_leaf:
                    named[c0 + (r if right else l)] = (c0 + (l if right else r), c0 + split)
                if dl < 0 and dr < 0:
                    break                             # the root
                g = r if right else l - 1             # the parent splits behind position g
                if g < 0 or g >= C - 1:
                    break                             # the sibling lies outside the chunk
                if g not in meet:
                    meet[g] = l if right else r       # first: leave the far end, stop
                    break
                other = meet[g]
                if right:
                    r = other
                else:
                    l = other
                split, is_leaf = g, False
    return named


def karras_ranges(nodes, n):
    """-> {node: (first, last, split)} from the oracle's Node array"""
    out = {}
    for i in range(n - 1):
        a = int(nodes[i]["data"][0])
        split = a if a < n - 1 else a - (n - 1)
        while a < n - 1:
            a = int(nodes[a]["data"][0])
        out[i] = (a - (n - 1), int(nodes[i]["right_edge"]), split)
    return out


def check(oracle, codes, rng):
    n = len(codes)
    nodes = oracle.build_bvh(codes, np.arange(n, dtype=np.uint32))
    named = climb(codes, rng)
    for i, (first, last, split) in karras_ranges(nodes, n).items():
        if first // C == last // C:
            assert named.get(i) == (last if i == first else first, split), (i, first, last, split, named.get(i))
        else:
            assert i not in named, i
    return len(named)


def scenes(rng, trials, nmax):
    for trial in range(trials):
        n = int(rng.integers(2, nmax))
        kind = trial % 3
        if kind == 0:
            yield np.sort(rng.integers(0, 1 << 30, n, dtype=np.uint32))
        elif kind == 1:
            yield np.sort(rng.integers(0, 50, n).astype(np.uint32))                     # heaps of equal codes
        else:
            yield np.sort((rng.integers(0, 8, n).astype(np.uint32) << 27) | rng.integers(0, 4, n).astype(np.uint32))


if __name__ == "__main__":
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import oracle
    oracle.build()
    rng = np.random.default_rng(5)
    for codes in scenes(rng, 60, 6000):
        print(len(codes), "leaves:", check(oracle, codes, rng), "nodes named, all as the oracle's")
