#!/usr/bin/env python3
"""Per-class instruction budget of relmc_eval_kernel<0, tile> (VERDICT r5 'do this' 3): what the wavefront ISSUES per interior-point iteration,
by class, from the compiler's own assembly weighted by how often each basic block runs -- and the same totals as the hardware counted them.

    python scripts/isa_budget.py [24|96] [--pmc profiles/<name>/pmc_classes[_rts96].json] [--tree]

Static side (no GPU): hipcc -S of relmc_core.hip, the kernel's control-flow graph (basic blocks, dominators, natural loops), and a frequency for
every block from the loop nest:
  * the window / scenario-group loops and the interior-point loop from the launch's geometry and the measured mean of the per-wavefront maximum
    iteration count (the loop runs until the slowest of the wavefront's rows has converged);
  * the five pass loops of the static LDL' schedule (they are the inner loops that prefetch a task descriptor with a global load) from the
    schedule's own pass counts (relmc_debug_symbolic: host only);
  * the loop that clears the fill-only blocks from nzero; compiler-made "waterfall" loops (v_readfirstlane + s_and_saveexec: one trip per
    distinct value among the lanes) one trip per scenario row of the wavefront;
  * everything else inside a loop runs once per trip of that loop (a conditional region counts as taken: an upper bound that the reconciliation
    below tests), except the blocks that only run when a row has just converged / failed, which are given their own small frequency.
Dynamic side: rocprofv3 --pmc totals of one 1e6-scenario launch (scripts/pmc_classes.sh) for the classes the hardware can count.  The two are
printed side by side; the classes the hardware cannot separate (v_cndmask, v_mov, DPP, v_readlane, fp64 compare / max) come from the static side only.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "powersystemsreliabilityassessment_amd", "csrc")


def classify(op, text):
    """Instruction class of one asm line."""
    dpp = (" row_" in text) or ("dpp" in op) or ("quad_perm" in text) or ("row_bcast" in text) or ("wave_" in text)
    if op.startswith(("v_fma_f64", "v_fmac_f64")): return "fp64 fma"
    if op.startswith("v_mul_f64"): return "fp64 mul"
    if op.startswith("v_add_f64"): return "fp64 add"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return "fp64 rcp (trans)"
    if op.startswith(("v_max_f64", "v_min_f64")): return "fp64 max/min"
    if op.startswith(("v_cmp_", "v_cmpx_")) and "_f64" in op: return "fp64 compare"
    if op.startswith(("v_div_", "v_ldexp_f64", "v_frexp", "v_trig", "v_fract_f64", "v_floor_f64", "v_ceil_f64", "v_rndne_f64")): return "fp64 other"
    if op.startswith("v_cvt_"): return "convert"
    if op.startswith("v_cndmask"): return "select (v_cndmask)"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "lane <-> scalar (v_readlane ...)"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")): return "move, DPP" if dpp else "move"
    if op.startswith(("v_cmp_", "v_cmpx_")): return "integer compare"
    if op.startswith(("v_mad_u64", "v_mad_i64", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co")): return "integer 64 / carry (address)"
    if op.startswith("v_"): return "integer 32 / logic" + (", DPP" if dpp else "")
    if op.startswith("ds_read") or op.startswith("ds_load"): return "LDS load"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "LDS store"
    if op.startswith("ds_"): return "LDS other"
    if op.startswith("scratch_load"): return "scratch load"
    if op.startswith("scratch_store"): return "scratch store"
    if op.startswith(("global_load", "flat_load", "buffer_load")): return "global load"
    if op.startswith(("global_store", "flat_store", "buffer_store", "global_atomic", "flat_atomic")): return "global store / atomic"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")): return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache")): return "scalar memory"
    if op.startswith(("s_nop", "s_sleep", "s_setprio", "s_barrier", "s_sethalt", "s_sendmsg", "s_getreg", "s_setreg", "s_inst_prefetch", "s_waitcnt_")): return "s_misc"
    if op.startswith("s_"): return "scalar ALU"
    return "other"


VALU = lambda c: c.startswith(("fp64", "convert", "select", "lane", "move", "integer"))


def compile_asm(out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, "relmc_core.hip")],
                          stderr=subprocess.DEVNULL)


def kernel_body(path, tile):
    lines = open(path).read().split("\n")
    key = "_ZN5relmc17relmc_eval_kernelILi0ENS_5TileTILi%d" % (16 if tile == 24 else 64)
    start = next(i for i, l in enumerate(lines) if l.startswith(key) and re.match(r"^\w+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start + 1:end]


class Block:
    def __init__(self, idx, label):
        self.idx, self.label, self.ins, self.succ, self.pred = idx, label, [], [], []


def build_cfg(body):
    blocks, cur = [], Block(0, "<entry>")
    blocks.append(cur)
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            nb = Block(len(blocks), m.group(1)); blocks.append(nb); cur = nb
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        cur.ins.append((op, re.sub(r"\s*;.*", "", s)))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            nb = Block(len(blocks), None); blocks.append(nb); cur = nb
    by_label = {b.label: b for b in blocks if b.label}
    for i, b in enumerate(blocks):
        last = b.ins[-1] if b.ins else None
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        if last and last[0].startswith("s_branch"):
            b.succ = [by_label[last[1].split()[-1]]]
        elif last and last[0].startswith("s_cbranch"):
            b.succ = [by_label[last[1].split()[-1]]] + ([nxt] if nxt else [])
        elif last and last[0].startswith(("s_endpgm", "s_setpc")):
            b.succ = []
        else:
            b.succ = [nxt] if nxt else []
    for b in blocks:
        for s in b.succ:
            s.pred.append(b)
    return blocks


def dominators(blocks):
    n = len(blocks)
    order, seen = [], set()
    stack = [(blocks[0], iter(blocks[0].succ))]
    seen.add(0)
    while stack:
        b, it = stack[-1]
        adv = False
        for s in it:
            if s.idx not in seen:
                seen.add(s.idx); stack.append((s, iter(s.succ))); adv = True; break
        if not adv:
            order.append(b); stack.pop()
    rpo = order[::-1]
    pos = {b.idx: i for i, b in enumerate(rpo)}
    idom = {rpo[0].idx: rpo[0].idx}

    def inter(a, b):
        while a != b:
            while pos[a] > pos[b]: a = idom[a]
            while pos[b] > pos[a]: b = idom[b]
        return a
    changed = True
    while changed:
        changed = False
        for b in rpo[1:]:
            ps = [p.idx for p in b.pred if p.idx in idom]
            if not ps: continue
            new = ps[0]
            for p in ps[1:]: new = inter(new, p)
            if idom.get(b.idx) != new:
                idom[b.idx] = new; changed = True
    def dom(a, b):          # a dominates b
        while True:
            if a == b: return True
            if b == idom.get(b, b): return False
            b = idom[b]
    return idom, dom, pos


def loop_forest(blocks, reach):
    """Loop nesting forest by recursive strongly-connected components (works on irreducible flow: the compiler rotates the pass loops so that
    they are entered both at the header and in the body).  Returns [(header idx, frozenset of block idx, depth, parent header or None)]."""
    sys.setrecursionlimit(10000)
    out = []

    def sccs(nodes, removed):
        index, low, on, st, res, counter = {}, {}, set(), [], [], [0]
        def succ(v):
            return [w.idx for w in blocks[v].succ if w.idx in nodes and (v, w.idx) not in removed]
        for root in sorted(nodes):
            if root in index: continue
            work = [(root, iter(succ(root)))]
            index[root] = low[root] = counter[0]; counter[0] += 1; st.append(root); on.add(root)
            while work:
                v, it = work[-1]
                adv = False
                for w in it:
                    if w not in index:
                        index[w] = low[w] = counter[0]; counter[0] += 1; st.append(w); on.add(w)
                        work.append((w, iter(succ(w)))); adv = True; break
                    elif w in on:
                        low[v] = min(low[v], index[w])
                if adv: continue
                work.pop()
                if work: low[work[-1][0]] = min(low[work[-1][0]], low[v])
                if low[v] == index[v]:
                    comp = set()
                    while True:
                        w = st.pop(); on.discard(w); comp.add(w)
                        if w == v: break
                    res.append(comp)
        return res

    def rec(nodes, removed, depth, parent):
        for comp in sccs(nodes, removed):
            if len(comp) == 1:
                v = next(iter(comp))
                if not any(w.idx == v and (v, v) not in removed for w in blocks[v].succ): continue
            entries = sorted(v for v in comp if any(p.idx not in comp for p in blocks[v].pred) or v == 0)
            h = entries[0] if entries else min(comp)
            out.append((h, frozenset(comp), depth, parent))
            rem2 = set(removed) | {(u, h) for u in comp if any(w.idx == h for w in blocks[u].succ)}
            rec(comp, rem2, depth + 1, h)
    rec(set(reach), set(), 0, None)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--") and a.isdigit()]
    tile = int(args[0]) if args else 24
    # launch geometry and iteration statistics of the 1e6-scenario launch the counters were taken on (bench.py's step)
    P = dict(n=1_000_000, waves=256 * 2 * 4 if tile == 24 else 256 * 8, rows=4 if tile == 24 else 1, rw=16 if tile == 24 else 64)
    if "--params" in sys.argv: P.update(json.loads(sys.argv[sys.argv.index("--params") + 1]))
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "core.s")
        compile_asm(s)
        body = kernel_body(s, tile)
    blocks = build_cfg(body)
    idom, dom, pos = dominators(blocks)
    reach = set(pos)
    forest = loop_forest(blocks, reach)
    loops = collections.OrderedDict((h, set(body)) for h, body, d, p in forest)
    hdrs = list(loops)
    parent = {h: p for h, body, d, p in forest}
    depth = {h: d for h, body, d, p in forest}
    own = {h: set(loops[h]) for h in hdrs}
    for h in hdrs:
        for g in hdrs:
            if parent[g] == h: own[h] -= loops[g]
    feat = {}
    for h in hdrs:
        own_ops = [op for i in own[h] for op, _ in blocks[i].ins]
        feat[h] = dict(n=sum(len(blocks[i].ins) for i in loops[h]), own=len(own_ops), gload=sum(o.startswith("global_load") for o in own_ops),
                       rfl=sum(o.startswith("v_readfirstlane") for o in own_ops), ds=sum(o.startswith("ds_") for o in own_ops),
                       setprio=sum(o.startswith("s_setprio") for o in own_ops), rcp=sum(o.startswith("v_rcp_f64") for o in own_ops),
                       philox=sum(o.startswith(("v_mad_u64_u32", "v_mul_hi_u32")) for o in own_ops))

    # ---- what each loop is, and how often it runs ------------------------------------------------------------------------------------------
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import schedule_interp
    from powersystemsreliabilityassessment_amd import case24, case96
    case = case24.rts24() if tile == 24 else case96.rts96()
    S = schedule_interp.symbolic(case, order=getattr(case, "elim_order", None))
    npuf = S.npass_upd - S.npass_updh - S.npass_updq
    trips_pass = [npuf, S.npass_updh, S.npass_updq, S.npass_inv, S.npass_bwd]
    if tile == 96:            # the wide tile carries the back substitution twice: all passes in half form (taken when the schedule allows it) or in full form
        all_half = S.bwd_half != 0
        trips_pass = [npuf, S.npass_updh, S.npass_updq, S.npass_inv, S.npass_bwd if all_half else 0, 0 if all_half else S.npass_bwd]
    children = lambda h: sorted([g for g in hdrs if parent[g] == h], key=lambda g: min(loops[g]))
    top = [h for h in hdrs if parent[h] is None]
    # the interior-point loop = the smallest loop that holds the s_setprio of the priority balancing AND the pass loops
    cands = [h for h in hdrs if sum(feat[g]["setprio"] for g in hdrs if loops[g] <= loops[h]) > 0 and sum(feat[g]["gload"] > 0 and feat[g]["ds"] >= 5 for g in hdrs if loops[g] < loops[h]) >= 3]
    ipm = min(cands, key=lambda h: len(loops[h]))
    passes = sorted([g for g in hdrs if loops[g] < loops[ipm] and feat[g]["ds"] >= 5 and (feat[g]["gload"] > 0 or any(feat[c]["gload"] for c in children(g)))
                     and not any(loops[g] < loops[x] < loops[ipm] and feat[x]["ds"] >= 5 for x in hdrs)], key=lambda g: min(loops[g]))
    zero = [g for g in hdrs if loops[g] < loops[ipm] and feat[g]["ds"] and not feat[g]["gload"] and feat[g]["n"] < 30 and g not in passes and not any(loops[g] < loops[x] for x in passes)]
    group_loop = parent[ipm]
    while group_loop is not None and feat[group_loop]["own"] < 100: group_loop = parent[group_loop]
    n, waves, rows = P["n"], P["waves"], P["rows"]
    groups = n / rows / waves                                 # scenario groups per wavefront
    T = P.get("trips")                                        # interior-point loop trips per group (evaluations: the last one only tests convergence)
    # per-wavefront frequency of a loop's ONE trip
    freq_loop = {}
    notes = {}
    def assign(h, f, what):
        freq_loop[h] = f; notes[h] = what
    for h in top:
        if loops[h] >= loops[ipm]:
            continue
        assign(h, None, "LDS copy of the case tables")
    W = next(h for h in top if loops[h] >= loops[ipm])
    nblk = (case.ncomp + 3) // 4
    def walk(h, f_parent):
        """f_parent = how often the code around loop h runs per wavefront; sets this loop's per-wavefront trip frequency, recurses"""
        T = P["trips"]
        in_group = group_loop is not None and loops[h] < loops[group_loop]
        if h == ipm:
            f = f_parent * T; what = f"interior-point loop: {T:.2f} trips per scenario group"
        elif h in passes:
            k = passes.index(h); f = f_parent * trips_pass[k]; what = (["update passes (full form)", "update passes (half form)", "update passes (quarter form)", "pivot inversion passes", "back-substitution passes"] + ["back-substitution passes (full form)"])[k] + f": {trips_pass[k]} per Newton step"
        elif h in zero:
            t = -(-S.nzero // P["rw"]); f = f_parent * t; what = f"clears the fill-only blocks: {t} trips"
        elif h == W:
            f = groups / 16 if tile == 24 else 1.0; what = "window loop (64 scenarios sampled per trip)" if tile == 24 else "kernel body"
        elif h == group_loop:
            f = f_parent * 16 if tile == 24 else groups; what = "scenario-group loop"
        elif loops[h] > loops[ipm]:
            f = f_parent; what = "(loop-forest artefact of an irreducible entry: same frequency as its parent)"
        elif loops[h] < loops[ipm]:
            f = f_parent; what = "(pass loop's second entry)"
        elif not in_group and feat[h]["philox"] >= 8:
            f = f_parent * nblk; what = f"Philox sampling: {nblk} blocks of 4 components"
        elif not in_group and children(h):
            f = f_parent; what = "(loop-forest artefact of an irreducible entry: same frequency as its parent)"
        elif not in_group and feat[h]["ds"] >= 3 and 40 < feat[h]["n"] < 90 and tile == 24:
            f = f_parent * case.ng; what = f"capacity of the units in service (the window's ordering key): {case.ng} units"
        elif not in_group:
            f = f_parent; what = "short loop of the sampling prologue: one trip"
        else:
            f = f_parent * P.get("p_lineout", 0.2 if tile == 24 else 0.15) * 3.0
            what = "topology work that runs only when a line is out in the wavefront: modelled at P(line out in the wavefront) x 3 trips"
        assign(h, f, what)
        for c in children(h):
            walk(c, f * ((T - 1) / T if h == ipm and c in passes else 1.0))          # the Newton step runs T - 1 times: the loop's last trip only tests convergence
    # copy loops at the top: once per wavefront, trip counts from the table sizes (negligible either way)
    for h in top:
        if h != W: assign(h, 8.0, "copy of the case tables into LDS")
    return_data = dict(blocks=blocks, loops=loops, parent=parent, depth=depth, own=own, feat=feat, ipm=ipm, passes=passes, zero=zero, W=W, group_loop=group_loop,
                       walk=walk, pos=pos, freq_loop=freq_loop, notes=notes, S=S, case=case, P=P, children=children, reach=reach, trips_pass=trips_pass)
    return return_data


def budget(D, T):
    """Per-wavefront dynamic instruction counts by class for interior-point trips T per group -> Counter(class -> count), plus the same split
    into {setup per launch / window / group, interior-point loop}."""
    D["P"]["trips"] = T
    blocks, loops, own, parent = D["blocks"], D["loops"], D["own"], D["parent"]
    D["freq_loop"].clear()
    for h in [h for h in loops if parent[h] is None and h != D["W"]]:
        D["freq_loop"][h] = 8.0; D["notes"][h] = "copy of the case tables into LDS"
    D["walk"](D["W"], 1.0)
    fb = {}
    innermost = {}
    for h, body in loops.items():
        for i in own[h]: innermost[i] = h
    for b in blocks:
        if b.idx not in D["reach"]: continue
        h = innermost.get(b.idx)
        fb[b.idx] = 1.0 if h is None else D["freq_loop"][h]
    # second half of the interior-point loop (Newton step, step lengths, update): everything from the first task-descriptor prefetch on runs T - 1 times
    ipm = D["ipm"]
    pos = D["pos"]
    marks = [pos[i] for i in own[ipm] if any(op.startswith("global_load_dwordx2") for op, _ in blocks[i].ins)]
    second = min(marks) if marks else None
    if second is not None:
        for i in own[ipm]:
            if pos[i] > second: fb[i] *= (T - 1.0) / T
    # the unrolled incidence-list gathers of the assembly (two bus slots x {line list, injection list}, two entries per step, `if (e >= longest) break`):
    # four runs of four wave-uniform `s_cbranch_vccnz` exits in the loop's own code; step k runs while 2k < the longest list of the slot
    runs, cur = [], []
    for b in blocks:
        if b.idx in own[ipm] and b.ins and b.ins[-1][0] == "s_cbranch_vccnz" and any(op.startswith("ds_read") for op, _ in b.ins + blocks[b.idx + 1].ins):
            if cur and b.idx - cur[-1] > 1: runs.append(cur); cur = []
            cur.append(b.idx)
        elif cur and b.idx - cur[-1] > 1:
            runs.append(cur); cur = []
    if cur: runs.append(cur)
    runs = [r for r in runs if len(r) == 4]
    S = D["S"]
    D["gather_runs"] = len(runs)
    if len(runs) == 4:
        longest = [S.maxdeg[0], S.maxinj[0], S.maxdeg[1], S.maxinj[1]]
        for r, ln in zip(runs, longest):
            for k in range(4):                          # the code of step k sits behind exit k: in the block that ends with exit k + 1 (the last step: the fall-through block)
                if not (2 * k < ln):
                    fb[r[k] + 1] = 0.0
    total = collections.Counter(); region = collections.defaultdict(collections.Counter)
    for b in blocks:
        if b.idx not in fb: continue
        f = fb[b.idx]
        h = innermost.get(b.idx)
        reg = "outside the interior-point loop"
        x = h
        while x is not None:
            if x == ipm: reg = "interior-point loop"; break
            x = parent[x]
        for op, text in b.ins:
            c = classify(op, text)
            total[c] += f; region[reg][c] += f
    return total, region, fb


def report(tile, pmc_path, D, out_md=None):
    P = D["P"]
    waves = P["waves"]
    pmc = json.load(open(pmc_path)) if pmc_path else None
    groups_total = P["n"] / P["rows"]
    # T from the hardware's LDS-instruction count (one measured number fixes the one free parameter; every other class is then a prediction)
    def lds_total(T):
        tot, _, _ = budget(D, T)
        return waves * sum(v for k, v in tot.items() if k.startswith("LDS"))
    if pmc:
        lo, hi = 5.0, 30.0
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            if lds_total(mid) < pmc["SQ_INSTS_LDS"]: lo = mid
            else: hi = mid
        T = 0.5 * (lo + hi)
    else:
        T = P.get("trips", 13.7)
    tot, region, fb = budget(D, T)
    L = []
    pr = L.append
    pr(f"# Instruction budget of relmc_eval_kernel<0, Tile{tile}> -- one launch of {P['n']} scenarios ({waves} wavefronts, {P['rows']} scenario row(s) each)")
    pr("")
    pr(f"Static side: the compiler's assembly, basic-block frequencies from the loop nest (scripts/isa_budget.py).  Interior-point loop trips per scenario group: "
       f"**{T:.2f}**" + (f" (fitted to the hardware's SQ_INSTS_LDS; the sampled iteration counts give {pmc['mean_max_iters']:.2f} + 1 for consecutive groups of {P['rows']}, "
                         f"{pmc['mean_iters']:.2f} + 1 per scenario)" if pmc else " (assumed)") + ".")
    pr(f"Static schedule of the case: {D['trips_pass'][0]} full + {D['trips_pass'][1]} half + {D['trips_pass'][2]} quarter update passes, {D['trips_pass'][3]} inversion, {D['trips_pass'][4]} back-substitution passes per Newton step.")
    pr("")
    valu = {k: v for k, v in tot.items() if VALU(k)}
    nv = sum(valu.values())
    pr("## VALU issue by class (wave-instructions of the launch)")
    pr("")
    pr("| class | static x frequency | share of VALU | of which inside the interior-point loop |")
    pr("|---|---:|---:|---:|")
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1]):
        pr(f"| {k} | {waves * v:.4g} | {100 * v / nv:.1f} % | {100 * region['interior-point loop'][k] / max(v, 1e-30):.0f} % |")
    pr(f"| **all VALU** | **{waves * nv:.4g}** | 100 % | {100 * sum(v for k, v in region['interior-point loop'].items() if VALU(k)) / nv:.0f} % |")
    pr("")
    other = {k: v for k, v in tot.items() if not VALU(k)}
    pr("## Everything else")
    pr("")
    pr("| class | static x frequency |")
    pr("|---|---:|")
    for k, v in sorted(other.items(), key=lambda kv: -kv[1]):
        pr(f"| {k} | {waves * v:.4g} |")
    pr("")
    if pmc:
        g = lambda *ks: waves * sum(v for k, v in tot.items() if any(k.startswith(x) for x in ks))
        rows = [("SQ_INSTS_VALU", g("fp64", "convert", "select", "lane", "move", "integer")),
                ("SQ_INSTS_VALU_FMA_F64", g("fp64 fma")), ("SQ_INSTS_VALU_MUL_F64", g("fp64 mul")), ("SQ_INSTS_VALU_ADD_F64", g("fp64 add")),
                ("SQ_INSTS_VALU_TRANS_F64", g("fp64 rcp")), ("SQ_INSTS_VALU_INT64", g("integer 64")), ("SQ_INSTS_LDS", g("LDS")), ("SQ_INSTS_LDS_LOAD", g("LDS load")),
                ("SQ_INSTS_LDS_STORE", g("LDS store")), ("SQ_INSTS_SALU", g("scalar ALU")), ("SQ_INSTS_BRANCH", g("branch")),
                ("SQ_INSTS_VMEM_RD", g("global load", "scratch load")), ("SQ_INSTS_VMEM_WR", g("global store", "scratch store"))]
        pr("## Reconciliation with the hardware's counters (rocprofv3 --pmc, scripts/pmc_classes.sh, same launch)")
        pr("")
        pr("| counter | measured | static x frequency | ratio |")
        pr("|---|---:|---:|---:|")
        for name, pred in rows:
            if name in pmc: pr(f"| {name} | {pmc[name]:.4g} | {pred:.4g} | {pred / pmc[name]:.3f} |")
        meas_int32 = pmc.get("SQ_INSTS_VALU_INT32")
        if meas_int32:
            pr(f"| SQ_INSTS_VALU_INT32 | {meas_int32:.4g} | (integer 32 / logic + integer compare: {g('integer 32', 'integer compare'):.4g}; with v_cndmask: {g('integer 32', 'integer compare', 'select'):.4g}) | |")
        rest = pmc["SQ_INSTS_VALU"] - sum(pmc.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"))
        pr(f"| VALU not in any class counter | {rest:.4g} | (moves + DPP moves + lane<->scalar + fp64 max/min + fp64 compare + select: {g('move', 'lane', 'fp64 max', 'fp64 compare', 'select'):.4g}) | |")
        pr("")
    text = "\n".join(L)
    print(text)
    if out_md:
        open(out_md, "w").write(text + "\n")
    return tot, region, T


def per_trip(tile, D, it_lo, it_hi, out_md=None):
    """ONE trip of the interior-point loop (evaluation + Newton step + update) per wavefront: static count of the loop's blocks x the pass loops' trips,
    against the hardware's count = difference of two launches whose rows all stop after max_it = k and k + 1 Newton steps (scripts/pmc_classes.sh <k>)."""
    a, b = json.load(open(it_lo)), json.load(open(it_hi))
    P = D["P"]
    groups_total = P["n"] / P["rows"]
    _, _, fb = budget(D, 1e9)                        # T -> infinity: (T - 1) / T = 1, frequencies relative to the loop's own trip follow below
    blocks, loops, own, parent, ipm = D["blocks"], D["loops"], D["own"], D["parent"], D["ipm"]
    f_ipm = D["freq_loop"][ipm]
    tot = collections.Counter()
    for i in loops[ipm]:
        f = fb[i] / f_ipm
        for op, text in blocks[i].ins:
            tot[classify(op, text)] += f
    L = []; pr = L.append
    pr(f"# One trip of the interior-point loop of relmc_eval_kernel<0, Tile{tile}>, per wavefront ({P['rows']} scenario row(s))")
    pr("")
    pr(f"Static: the loop's basic blocks x the static schedule's pass counts ({D['trips_pass'][0]} full + {D['trips_pass'][1]} half + {D['trips_pass'][2]} quarter update passes, "
       f"{D['trips_pass'][3]} inversion, {D['trips_pass'][4]} back-substitution); the unrolled incidence-list gathers run the steps the case's longest lists ask for "
       f"(lines {D['S'].maxdeg[0]} / {D['S'].maxdeg[1]}, injections {D['S'].maxinj[0]} / {D['S'].maxinj[1]} per bus slot: {D['gather_runs']} of 4 gather runs found in the assembly); every other conditional region counted as taken.  Measured: rocprofv3 --pmc counters of two "
       f"{P['n']}-scenario launches in which every row stops after {int(a['mean_max_iters'])} and {int(b['mean_max_iters'])} Newton steps (mpoption(max_it), no second attempts), "
       f"difference / {groups_total:.0f} wavefront-groups.")
    pr("")
    valu = {k: v for k, v in tot.items() if VALU(k)}
    nv = sum(valu.values())
    pr("| class | instructions per trip (static) | share of VALU issue | algorithmic? |")
    pr("|---|---:|---:|---|")
    alg = {"fp64 fma": "yes", "fp64 mul": "yes", "fp64 add": "yes", "fp64 rcp (trans)": "yes (pivots, 1/z: one v_rcp_f64 + 2 Newton steps each)",
           "fp64 max/min": "yes (norms of the termination tests, ratio tests; run at the fp64 rate)", "fp64 compare": "yes (ratio tests, bounds)",
           "fp64 other": "yes", "move, DPP": "yes: the row all-reduces (4 row_ror steps per reduction; their add / max partners are counted above)",
           "move": "no: register copies (64-bit pairs around the LDS gathers and the unrolled slots)", "select (v_cndmask)": "no: slot predicates (in service / boxed / owner) as selects",
           "integer 32 / logic": "no: LDS addresses (descriptor fields -> byte offsets), slot flag tests, loop counters", "integer compare": "no: slot predicates, descriptor tests",
           "lane <-> scalar (v_readlane ...)": "no: SGPR spills parked in VGPR lanes (v_readlane / v_writelane), wave-uniform values of the wide tile",
           "integer 64 / carry (address)": "no: global addresses of the descriptor prefetch", "convert": "no", "integer 32 / logic, DPP": "no"}
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1]):
        pr(f"| {k} | {v:.0f} | {100 * v / nv:.1f} % | {alg.get(k, '')} |")
    pr(f"| **all VALU** | **{nv:.0f}** | 100 % | |")
    for k, v in sorted(((k, v) for k, v in tot.items() if not VALU(k)), key=lambda kv: -kv[1]):
        pr(f"| {k} | {v:.0f} | | |")
    pr("")
    g = lambda *ks: sum(v for k, v in tot.items() if any(k.startswith(x) for x in ks))
    d = lambda name: (b[name] - a[name]) / groups_total
    rows = [("SQ_INSTS_VALU", g("fp64", "convert", "select", "lane", "move", "integer")), ("SQ_INSTS_VALU_FMA_F64", g("fp64 fma")), ("SQ_INSTS_VALU_MUL_F64", g("fp64 mul")),
            ("SQ_INSTS_VALU_ADD_F64", g("fp64 add")), ("SQ_INSTS_VALU_TRANS_F64", g("fp64 rcp")), ("SQ_INSTS_LDS", g("LDS")), ("SQ_INSTS_LDS_LOAD", g("LDS load")),
            ("SQ_INSTS_LDS_STORE", g("LDS store")), ("SQ_INSTS_SALU", g("scalar ALU")), ("SQ_INSTS_BRANCH", g("branch")), ("SQ_INSTS_VMEM_RD", g("global load", "scratch load")),
            ("SQ_INSTS_VMEM_WR", g("global store", "scratch store"))]
    pr("| hardware counter | measured per trip | static per trip | static / measured |")
    pr("|---|---:|---:|---:|")
    for name, pred in rows:
        m = d(name)
        pr(f"| {name} | {m:.1f} | {pred:.1f} | {pred / m if m else float('nan'):.3f} |")
    i32, i64 = d("SQ_INSTS_VALU_INT32"), d("SQ_INSTS_VALU_INT64")
    pr(f"| SQ_INSTS_VALU_INT32 | {i32:.1f} | integer 32 / logic {g('integer 32'):.0f}, integer compare {g('integer compare'):.0f}, v_cndmask {g('select'):.0f} | |")
    pr(f"| SQ_INSTS_VALU_INT64 | {i64:.1f} | integer 64 {g('integer 64'):.0f} (the hardware's class is wider than the mnemonics sorted here: 64-bit moves / v_lshl_add_u64 of the descriptor addresses) | |")
    rest = d("SQ_INSTS_VALU") - sum(d(k) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"))
    pr(f"| VALU in no class counter | {rest:.1f} | moves {g('move'):.0f} (of them DPP {g('move, DPP'):.0f}), lane<->scalar {g('lane'):.0f}, fp64 max/min {g('fp64 max'):.0f}, fp64 compare {g('fp64 compare'):.0f} | |")
    pr("")
    nonalg = g("move") - g("move, DPP") + g("select") + g("integer") + g("lane") + g("convert")
    pr(f"Not algorithmic (copies, selects, integer / address work, lane<->scalar traffic): {nonalg:.0f} of {nv:.0f} VALU instructions per trip = {100 * nonalg / nv:.1f} % (static); "
       f"measured upper bound: VALU - fp64 add/mul/fma - rcp = {d('SQ_INSTS_VALU') - d('SQ_INSTS_VALU_FMA_F64') - d('SQ_INSTS_VALU_MUL_F64') - d('SQ_INSTS_VALU_ADD_F64') - d('SQ_INSTS_VALU_TRANS_F64'):.0f} "
       f"of {d('SQ_INSTS_VALU'):.0f} = {100 * (1 - (d('SQ_INSTS_VALU_FMA_F64') + d('SQ_INSTS_VALU_MUL_F64') + d('SQ_INSTS_VALU_ADD_F64') + d('SQ_INSTS_VALU_TRANS_F64')) / d('SQ_INSTS_VALU')):.1f} % "
       f"(includes the algorithmic fp64 max / compare and the row reductions' DPP moves).")
    text = "\n".join(L)
    print(text)
    if out_md: open(out_md, "w").write(text + "\n")
    return tot


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a.isdigit()]
    tile = int(args[0]) if args else 24
    pmc_path = sys.argv[sys.argv.index("--pmc") + 1] if "--pmc" in sys.argv else None
    out_md = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    D = main()
    if "--tree" in sys.argv:
        budget(D, 13.7)
        for h in sorted(D["loops"], key=lambda h: min(D["loops"][h])):
            print("  " * D["depth"][h] + f"{D['blocks'][h].label}: {D['feat'][h]['own']} own instr., {D['freq_loop'].get(h)} trips per wavefront -- {D['notes'].get(h)}")
    if "--trip" in sys.argv:
        k = sys.argv.index("--trip")
        per_trip(tile, D, sys.argv[k + 1], sys.argv[k + 2], out_md)
    else:
        report(tile, pmc_path, D, out_md)
