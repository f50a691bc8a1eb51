"""How good is relmc_case_load's elimination order?  (developer tool, host only, round 3)

Model of the update phase of the sparse 2x2-block LDL' (one bus per pivot): a task starts when its operands are complete, the updates of one
target block run one per pass in pivot order (read-modify-write chains), unlimited lanes.  Simulated annealing over the elimination order
(reference bus last) minimises  max(critical path, tasks / lanes) + 0.6 (tree height - 1)  = update passes + back-substitution passes.

Result (DESIGN_HISTORY.md H3, round 3): RTS-24 11 / 8 (critical path / tree height) under the shipped level-then-fill rule, nothing better found;
RTS-96 19 / 12 shipped, 17 / 11 at best -- a nested-dissection order has nothing to offer, the shipped trees are as shallow as these graphs allow.
    python scripts/order_search.py
"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from powersystemsreliabilityassessment_amd import case24, case96


def graph(c):
    adj = [set() for _ in range(c.nb)]
    for f, t in zip(c.br_from, c.br_to):
        adj[int(f)].add(int(t)); adj[int(t)].add(int(f))
    return adj


def shipped_rule(adj, ref):
    """level first, then fill, then degree, then bus number: case_symbolic's key (csrc/relmc_schedule.hip)"""
    n = len(adj); A = [set(s) for s in adj]; gone = [False] * n; level = [-1] * n; order = []
    for step in range(n):
        best = None
        for b in range(n):
            if gone[b] or (b == ref and step < n - 1): continue
            lev = max([level[x] + 1 for x in A[b] if gone[x]], default=0)
            live = [x for x in A[b] if not gone[x]]
            fill = sum(1 for i, x in enumerate(live) for y in live[i + 1:] if y not in A[x])
            key = ((lev * 1000 + fill) * 10000 + len(live) * 100 + b)
            if best is None or key < best[0]: best = (key, b, lev)
        _, b, lev = best
        level[b] = lev
        live = [x for x in A[b] if not gone[x]]
        for x in live:
            for y in live:
                if x != y: A[x].add(y)
        gone[b] = True; order.append(b)
    return order


def model(adj, order, lanes):
    """(critical path of the update phase, update tasks, tree height, tasks / lanes)"""
    pos = {v: k for k, v in enumerate(order)}
    A = [set(s) for s in adj]
    finD, finK, finY, level = {}, {}, {}, {}
    relD = {v: [] for v in order}; relK = {}; relY = {v: [] for v in order}
    tasks = 0

    def chain(rels):
        e = 0
        for r in rels: e = max(e, r) + 1
        return e
    for v in order:
        hi = sorted([u for u in A[v] if pos[u] > pos[v]], key=lambda u: pos[u])
        level[v] = 1 + max([level[u] for u in A[v] if pos[u] < pos[v]], default=-1)
        finD[v] = chain(relD[v]); finY[v] = chain(relY[v])
        for a in hi: finK[(a, v)] = chain(relK.get((a, v), []))
        for ia, a in enumerate(hi):
            for b in hi[:ia + 1]:
                r = max(finD[v], finK[(a, v)], finK[(b, v)])
                (relD[a] if a == b else relK.setdefault((a, b), [])).append(r)
                tasks += 1
            relY[a].append(max(finD[v], finK[(a, v)], finY[v])); tasks += 1
        for u in hi:
            for w in hi:
                if u != w: A[u].add(w)
    cp = max(list(finD.values()) + list(finY.values()) + list(finK.values()) + [0])
    return cp, tasks, max(level.values()) + 1, -(-tasks // lanes)


def anneal(adj, order0, lanes, iters, seed):
    rnd = random.Random(seed); n = len(adj)

    def cost(o):
        cp, tasks, h, cap = model(adj, o, lanes)
        return max(cp, cap) + 0.6 * (h - 1) + 0.002 * tasks, (cp, tasks, h, cap)
    cur = list(order0); cc, ci = cost(cur); bc, bi, bo = cc, ci, list(cur)
    T = 0.5
    for _ in range(iters):
        i, j = rnd.randrange(n - 1), rnd.randrange(n - 1)          # the reference bus stays last
        if i == j: continue
        new = list(cur)
        if rnd.random() < 0.5: new[i], new[j] = new[j], new[i]
        else: new.insert(j, new.pop(i))
        nc, ni = cost(new)
        if nc <= cc or rnd.random() < np.exp((cc - nc) / T):
            cur, cc, ci = new, nc, ni
            if nc < bc: bc, bi, bo = nc, ni, list(new)
        T = max(0.02, T * 0.9997)
    return bi, bo


if __name__ == "__main__":
    for name, c, lanes in (("RTS-24", case24.rts24(), 16), ("RTS-96", case96.rts96(), 64)):
        adj = graph(c); o = shipped_rule(adj, int(c.ref_bus))
        print(name, "shipped rule: (critical path, update tasks, tree height, tasks / lanes) =", model(adj, o, lanes))
        for seed in (1, 2, 3):
            bi, bo = anneal(adj, o, lanes, 20000 if c.nb < 40 else 40000, seed)
            print(name, "annealed, seed", seed, "->", bi, "RELMC_ORDER=" + ",".join(str(v) for v in bo), flush=True)
