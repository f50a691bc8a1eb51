"""PCIe-inclusive rate of the host-buffer entry points (never bench.py's `value`): 1e6 states from and to pageable host memory.
  python scripts/pcie_rate.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from powersystemsreliabilityassessment_amd import api, _abi
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng = api.Engine()
t = time.perf_counter(); st = eng.mc_sampling(None, n, seed=1); t_samp = time.perf_counter() - t
t = time.perf_counter(); st = eng.mc_sampling(None, n, seed=1); t_samp = min(t_samp, time.perf_counter() - t)
st = np.ascontiguousarray(st)
o = api.mpoption()
dns = np.zeros(n); nodal = np.zeros((n, eng.case.nb)); status = np.zeros(n, dtype=np.int32); iters = np.zeros(n, dtype=np.int32)
def run(with_nodal, with_info):
    t = time.perf_counter()
    rc = eng.L.relmc_mc_simulation(eng._h, st.ctypes.data_as(_abi.c_uint8_p), n, C.byref(o), dns.ctypes.data_as(_abi.c_double_p),
                                   nodal.ctypes.data_as(_abi.c_double_p) if with_nodal else None,
                                   status.ctypes.data_as(_abi.c_int32_p) if with_info else None, iters.ctypes.data_as(_abi.c_int32_p) if with_info else None)
    assert rc == 0
    return (time.perf_counter() - t) * 1e3
run(True, True)
for name, a, b in (("dns + nodal + status + iters", True, True), ("dns + nodal", True, False), ("dns only", False, False)):
    ts = [run(a, b) for _ in range(5)]
    print(f"relmc_mc_simulation, {n} states host -> host, outputs {name}: min {min(ts):.2f} ms, median {sorted(ts)[2]:.2f} ms "
          f"({n / min(ts) / 1e3:.1f} M states/s), kernel share {eng.last_kernel_ms():.2f} ms")
print(f"relmc_mc_sampling, {n} states to the host: {t_samp * 1e3:.2f} ms")
acc = eng.nsq_accumulate(1, 0, n)
assert abs(dns.sum() - acc.sum_dns) < 1e-6 * acc.sum_dns and int((dns > 1e-4).sum()) == acc.n_fail
print("checked: sum(dns) and the loss count equal the fused path's accumulators")
