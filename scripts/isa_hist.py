"""Instruction-class histogram of asm line ranges:  python scripts/isa_hist.py k0.s 1874:3816 3816:3960 ...
(ranges of the kernel's .s between the s_memtime markers of a -DRELMC_PHASE_TIMING build; developer tool)"""
import re, sys
from collections import Counter
def cls(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64")): return "fp64:" + op[2:5]
    if op.startswith("v_rcp_f64"): return "fp64:rcp"
    if op.startswith(("v_mov", "v_accvgpr")): return "mov"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp"): return "vcmp"
    if op.startswith("v_"): return "valu-other"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"
lines = open(sys.argv[1]).read().split("\n")
for rg in sys.argv[2:]:
    a, b = map(int, rg.split(":"))
    c = Counter(); dpp = 0
    for ln in lines[a:b]:
        m = re.match(r"\s+([a-z_0-9]+)", ln)
        if not m: continue
        c[cls(m.group(1))] += 1
        if "row_" in ln or "dpp" in ln: dpp += 1
    tot_v = sum(v for k, v in c.items() if k.startswith(("fp64", "mov", "lane", "cndmask", "vcmp", "valu")))
    print(rg, "VALU", tot_v, "dpp", dpp, dict(sorted(c.items(), key=lambda kv: -kv[1])))
