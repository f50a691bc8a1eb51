"""Would type-pure update passes pay?  (host-only model, round 3)

A full-form update pass costs 10 LDS instructions because its lanes may hold any task.  A pass of right-hand-side tasks only would cost 7
(no second rows), one of diagonal-target tasks only 8 (Wb is Wa).  This script rebuilds the dependency graph of the shipped update phase
from relmc_debug_symbolic's pass program (full-form equivalents of every task), and list-schedules it again with passes that are either
pure (R / G) or mixed, choosing per pass whichever candidate set maximises (tasks done) / (instructions spent), then applies the same
half / quarter forms to sparsely filled passes.  Output: passes and LDS instructions of the update phase, shipped against typed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# The shipped librelmc.so reads no RELMC_* ablation variable: RELMC_NO_QUARTER exists only in the -DRELMC_DEV_SWITCHES build
# (make -C powersystemsreliabilityassessment_amd/csrc ablate/librelmc_dev.so).  Without it the dump holds half- and quarter-form passes
# and this model would read them as full-form tasks: refuse instead of printing a wrong number.
DEV = os.path.join(ROOT, "powersystemsreliabilityassessment_amd", "csrc", "ablate", "librelmc_dev.so")
if not os.environ.get("RELMC_LIB_PATH"):
    if not os.path.exists(DEV):
        sys.exit(f"typed_pass_model.py needs the dev-switch build: {DEV} is missing (make -C powersystemsreliabilityassessment_amd/csrc ablate/librelmc_dev.so)")
    os.environ["RELMC_LIB_PATH"] = DEV
os.environ["RELMC_NO_QUARTER"] = "1"        # the dump then holds every update task in full form
import numpy as np
from powersystemsreliabilityassessment_amd import case24, case96
from tests import schedule_interp as si


def tasks_of(case):
    s = si.symbolic(case, 0, case.elim_order)
    T = []
    for p in range(s.npass_upd):
        for r in range(s.rw):
            d = [int(x) for x in s.tasks[p, r]]
            if d[0] == 0xffff: continue
            vec = bool(d[0] & 0x8000); t = d[0] & 0x7fff
            kind = "R" if vec else ("G" if d[1] == d[2] else "O")
            rd = {("b", d[3] // 4), ("b", d[1] // 4) if not vec else ("y", d[1]), ("b", d[2] // 4), ("y", t) if vec else ("b", t // 4)}
            wr = ("y", t) if vec else ("b", t // 4)
            T.append(dict(kind=kind, rd=rd, wr=wr, p=p))
    return s, T


def deps(T):
    lastw, readers = {}, {}
    strict = [set() for _ in T]; weak = [set() for _ in T]
    for i, t in enumerate(T):
        for u in t["rd"]:
            if u in lastw: strict[i].add(lastw[u])
        w = t["wr"]
        if w in lastw: strict[i].add(lastw[w])
        for q in readers.get(w, ()):
            if q != i: weak[i].add(q)
        for u in t["rd"]: readers.setdefault(u, []).append(i)
        lastw[w] = i; readers[w] = []
    return strict, weak


STRICT = True


def schedule(T, rw, typed):
    strict, weak = deps(T)
    n = len(T); succ = [[] for _ in T]
    for i in range(n):
        for q in strict[i]: succ[q].append(i)
    depth = [0] * n
    for i in range(n - 1, -1, -1):
        for j in succ[i]: depth[i] = max(depth[i], depth[j] + 1)
    po = [-1] * n; passes = []
    while min(po) < 0:
        cur = len(passes)
        ready = [i for i in range(n) if po[i] < 0 and all(0 <= po[q] < cur for q in strict[i])]
        ready.sort(key=lambda i: (-depth[i], i))
        def pick(cands, cap):
            chosen = []; inp = set()
            for i in cands:
                if len(chosen) >= cap: break
                if all(po[q] >= 0 or q in inp for q in weak[i]): chosen.append(i); inp.add(i)
            return chosen
        options = [("M", pick(ready, rw), 10)]
        if typed:
            options.append(("R", pick([i for i in ready if T[i]["kind"] == "R"], rw), 7))
            options.append(("G", pick([i for i in ready if T[i]["kind"] == "G"], rw), 8))
        # a pure pass must not starve the critical path: only when it holds every ready task of maximal depth
        dmax = max(depth[i] for i in ready)
        crit = {i for i in ready if depth[i] == dmax}
        best = None
        for name, ch, cost in options:
            if not ch: continue
            if name != "M" and STRICT and not crit <= set(ch): continue
            score = len(ch) / cost
            if best is None or score > best[0]: best = (score, name, ch, cost)
        _, name, ch, cost = best
        for i in ch: po[i] = cur
        passes.append((name, ch))
    total = 0; forms = []
    for name, ch in passes:
        rows = sum(1 if T[i]["kind"] == "R" else 2 for i in ch); elems = sum(2 if T[i]["kind"] == "R" else 4 for i in ch)
        if elems <= rw: c, f = 6, "q"
        elif rows <= rw: c, f = 7, "h"
        else: c, f = {"M": 10, "R": 7, "G": 8}[name], name
        total += c; forms.append("%s%d" % (f, len(ch)))
    return len(passes), total, forms


for name, case in (("RTS-24", case24.rts24()), ("RTS-96", case96.rts96())):
    s, T = tasks_of(case)
    print(name, "update tasks", len(T), {k: sum(1 for t in T if t["kind"] == k) for k in "RGO"})
    for typed, strict in ((False, True), (True, True), (True, False)):
        STRICT = strict
        np_, lds, forms = schedule(T, s.rw, typed)
        print("   %s: %d passes, %d LDS instructions, cost (instructions + 4 per pass) %d   %s" % (
            "mixed passes only                         " if not typed else ("pure passes that do not starve the chain  " if strict else "pure passes wherever they score better    "),
            np_, lds, lds + 4 * np_, " ".join(forms)))
