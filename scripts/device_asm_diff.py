"""Which device functions of a unit changed between two source states?  Compiles the unit's gfx950 assembly for a git revision (default HEAD,
via `git stash`-free `git show` into a temp tree) and for the working tree, strips comments / directives and compares function by function:
    python scripts/device_asm_diff.py [unit.hip] [rev]
Used in round 5 to show that a change to relmc_finalize_kernel left all 14 relmc_eval_kernel instantiations byte-identical."""
import hashlib, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "powersystemsreliabilityassessment_amd", "csrc")
unit = sys.argv[1] if len(sys.argv) > 1 else "relmc_core.hip"
rev = sys.argv[2] if len(sys.argv) > 2 else "HEAD"


def asm(csrc, out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, os.path.join(csrc, unit)],
                          stderr=subprocess.DEVNULL)


def funcs(path):
    out, cur, buf = {}, None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, buf = m.group(1), []
        elif cur is not None:
            if ln.startswith(".Lfunc_end"):
                out[cur] = buf; cur = None
            elif not ln.strip().startswith((";", ".")):
                buf.append(re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r";.*", "", ln)).strip())      # block labels carry the function's ordinal in the unit
    return out


with tempfile.TemporaryDirectory() as tmp:
    old = os.path.join(tmp, "old"); os.makedirs(os.path.join(old, "powersystemsreliabilityassessment_amd"))
    subprocess.check_call(f"git -C {ROOT} archive {rev} powersystemsreliabilityassessment_amd/csrc include | tar -x -C {old}", shell=True)
    asm(os.path.join(old, "powersystemsreliabilityassessment_amd", "csrc"), os.path.join(tmp, "old.s"))
    asm(CSRC, os.path.join(tmp, "new.s"))
    a, b = funcs(os.path.join(tmp, "old.s")), funcs(os.path.join(tmp, "new.s"))
    sub = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()[:110]
    for k in sorted(set(a) | set(b)):
        same = hashlib.sha256("\n".join(a.get(k, [])).encode()).digest() == hashlib.sha256("\n".join(b.get(k, [])).encode()).digest()
        print(("same " if same else "DIFF ") + sub(k), len(a.get(k, [])), "->", len(b.get(k, [])), "instructions")
