"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement of the reference's Chu-Liu-Edmonds head decoding
(reference: cube/networks/utils.py:400-577), in NumPy fp64, written iteratively
(explicit contraction records instead of recursion) — the same structure the HIP
kernel `cube_mst_decode` uses, so each step can be compared.

Pinned by: tests/golden/g4_mst.npz (heads produced by the reference's own
`ChuLiuEdmondsDecoder.decode` in the build container, see tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np


class NoTreeError(AssertionError):
    """The reference asserts `best_tree is not None` (utils.py:556-562) and re-raises."""


def _find_cycle(tree):
    """The reference runs Tarjan's SCC over "node -> its dependents" (utils.py:404-443) and
    contracts `cycles.pop()`, the LAST cycle found (utils.py:456).  Each node has exactly one
    head, so a cycle can only be entered from the outer loop (ascending node id) at its
    smallest node: cycles are found in order of their minimum node and the last one is the
    cycle whose minimum node is largest.  Returns a bool mask or None."""
    m = len(tree)
    state = np.zeros(m, dtype=np.int64)  # 0 = unseen, else walk id
    best_min, best = -1, None
    for s in range(m):
        if state[s]:
            continue
        k = s
        while not state[k]:
            state[k] = s + 1
            k = tree[k]
        if state[k] == s + 1:  # closed a new loop at k
            nodes = [k]
            j = tree[k]
            while j != k:
                nodes.append(j)
                j = tree[j]
            if len(nodes) > 1 and min(nodes) > best_min:
                best_min = min(nodes)
                best = np.zeros(m, dtype=bool)
                best[nodes] = True
    return best


def chuliu_edmonds(scores):
    """reference: utils.py:445-518.  `scores` (m x m, fp64; row = dependent, column = head) is
    modified in place exactly like the reference's top-level call; deeper levels work on copies."""
    records = []
    cur = scores
    while True:
        m = cur.shape[0]
        cur *= (1 - np.eye(m))
        cur[0] = 0
        cur[0, 0] = 1
        tree = np.argmax(cur, axis=1)
        cyc = _find_cycle(tree)
        if cyc is None:
            break
        C = np.where(cyc)[0]
        N = np.where(~cyc)[0]
        cyc_scores = cur[C, tree[C]]
        total = 1.0
        for v in cyc_scores:  # ascending node order, sequential product
            total = total * v
        with np.errstate(divide="ignore", invalid="ignore"):
            head_sc = cur[C][:, N] / cyc_scores[:, None] * total  # (c x n)
        dep_sc = cur[N][:, C]  # (n x c)
        best_c = np.argmax(head_sc, axis=0)
        best_dep = np.argmax(dep_sc, axis=1)
        sub = np.zeros((len(N) + 1, len(N) + 1))
        sub[:-1, :-1] = cur[N][:, N]
        sub[-1, :-1] = head_sc[best_c, np.arange(len(N))]
        sub[:-1, -1] = dep_sc[np.arange(len(N)), best_dep]
        records.append((C, N, tree[C].copy(), best_c, best_dep))
        cur = sub
    # expand (utils.py:491-518), deepest level first
    for C, N, tree_c, best_c, best_dep in reversed(records):
        cycle_head = tree[-1]
        inner = tree[:-1]
        new_tree = -np.ones(len(C) + len(N), dtype=np.int64)
        keep = inner < len(inner)
        new_tree[N[keep]] = N[inner[keep]]
        new_tree[N[~keep]] = C[best_dep[~keep]]
        new_tree[C] = tree_c
        new_tree[C[best_c[cycle_head]]] = N[cycle_head]
        tree = new_tree
    return tree


def root_candidates(scores, tree):
    """reference: utils.py:525-534 (`scores` already carries the in-place edits of the first
    chuliu_edmonds call: zero diagonal, row 0 = e_0)."""
    roots = np.where(tree[1:] == 0)[0] + 1
    if len(roots) == 1:
        return roots, True
    if len(roots) == 0:
        roots = np.where(scores[1:, 0] >= 1 / len(scores))[0] + 1
    if len(roots) == 0:
        roots = np.array([np.argmax(scores[1:, 0]) + 1])
    return roots, False


def rooted_tree(scores, root):
    """One candidate of the single-root repair (utils.py:537-552): -> (tree, log-score)."""
    root_score = scores[root, 0]
    s = np.array(scores)
    s[1:, 0] = 0
    s[root] = 0
    s[root, 0] = 1
    tree = chuliu_edmonds(s)
    probs = s[np.arange(len(s)), tree]
    with np.errstate(divide="ignore"):
        sc = np.log(probs).sum() + np.log(root_score) if probs.all() else -np.inf
    return tree, sc


def chuliu_edmonds_one_root(scores):
    """reference: utils.py:520-563."""
    scores = scores.astype(np.float64)
    tree = chuliu_edmonds(scores)
    roots, single = root_candidates(scores, tree)
    if single:
        return tree
    best_score, best_tree = -np.inf, None
    for r in roots:
        t, sc = rooted_tree(scores, r)
        if sc > best_score:
            best_score, best_tree = sc, t
    if best_tree is None:
        raise NoTreeError("no single-rooted tree with non-zero probability")
    return best_tree


def decode(score, lens):
    """reference: ChuLiuEdmondsDecoder.decode, utils.py:565-577.
    score [B, L, T] probabilities (row i = word i+1, column j = head j, 0 = root),
    lens [B] -> list of int64 arrays (heads of words 1..len)."""
    out = []
    for b in range(score.shape[0]):
        n = int(lens[b]) + 1
        s = np.zeros((n, n))
        s[1:, :] = score[b, : n - 1, :n]
        s *= (1 - np.eye(n))
        out.append(chuliu_edmonds_one_root(s)[1:])
    return out
