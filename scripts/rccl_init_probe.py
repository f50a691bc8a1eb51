"""How often does RCCL's communicator bootstrap stall on this box, and where?  Spawns N one-rank inits through the library
(relmc_comm_unique_id + relmc_comm_init) as child processes with NCCL_DEBUG=INFO and a deadline; prints the durations and the
tail of a stalled child's debug output.   python scripts/rccl_init_probe.py [N=30] [deadline=25] [ENV=VALUE ...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, ctypes as C
sys.path.insert(0, %r)
from powersystemsreliabilityassessment_amd import api, case24
e = api.Engine(case24.rts24())
e.comm_set_timeout(0)
t = time.time()
uid = (C.c_uint8 * 128)()
assert e.L.relmc_comm_unique_id(uid) == 0
rc = e.L.relmc_comm_init(e._h, 1, 0, uid)
sys.stdout.write("\nINIT rc %%d %%.2f s\n" %% (rc, time.time() - t)); sys.stdout.flush()
""" % ROOT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
deadline = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
env = dict(os.environ, NCCL_DEBUG="INFO")
for kv in sys.argv[3:]:
    k, v = kv.split("=", 1); env[k] = v
dur, stalls = [], 0
for i in range(n):
    t = time.time()
    p = subprocess.Popen([sys.executable, "-c", CHILD], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    try:
        out, _ = p.communicate(timeout=deadline)
        line = [l for l in out.splitlines() if l.startswith("INIT")]
        dur.append(time.time() - t)
        print(i, line[-1] if line else "no INIT line (rc %d): %r" % (p.returncode, out[-300:]), "wall %.1f s" % dur[-1], flush=True)
    except subprocess.TimeoutExpired:
        p.kill(); out, _ = p.communicate()
        stalls += 1
        print(i, "STALLED after %.0f s; last lines of its output:" % deadline, flush=True)
        for l in out.splitlines()[-12:]:
            print("     |", l[:220], flush=True)
print("runs %d, stalls %d, wall of the others: min %.1f max %.1f s; extra env %s" % (n, stalls, min(dur) if dur else 0, max(dur) if dur else 0, sys.argv[3:]))
