import os, sys, subprocess
def run(lib, w):
    env=dict(os.environ, RELMC_FIRST_HALF=w)
    out=subprocess.run([sys.executable,"scripts/variant_check.py",lib],env=env,capture_output=True,text=True).stdout.strip()
    print(lib, w, out.split("fixture")[0][-12:], out[-60:], flush=True)
for l in ("librelmc_pc13","librelmc_pc15","librelmc_pc17"): run(l, "0")
env=dict(os.environ, RELMC_FIRST_HALF="0"); 
src=open("scripts/wave_tail.py").read().replace("librelmc_pt.so","librelmc_prio_pt.so")
print(subprocess.run([sys.executable,"-c",src],env=env,capture_output=True,text=True).stdout)
