"""Two ranks of relmc_comm_* on the ONE GPU of the box: RCCL refuses (ncclCommInitRank: invalid usage), which is why the N > 1 RCCL path can
only run on the driver's multi-GPU node; the library reports the refusal as RELMC_ERR_HIP with RCCL's message and stays usable."""
import os, sys, time, ctypes as C, multiprocessing as mp
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
def worker(rank, path):
    from powersystemsreliabilityassessment_amd import api, _abi
    eng = api.Engine()
    uid = (C.c_uint8 * 128)()
    if rank == 0:
        assert eng.L.relmc_comm_unique_id(uid) == 0
        open(path, "wb").write(bytes(uid)); os.rename(path, path + ".ok")
    else:
        while not os.path.exists(path + ".ok"): time.sleep(0.05)
        uid = (C.c_uint8 * 128).from_buffer_copy(open(path + ".ok", "rb").read())
    rc = eng.L.relmc_comm_init(eng._h, 2, rank, uid)
    print("rank", rank, "init rc", rc, eng.L.relmc_last_error(eng._h), flush=True)
    if rc == 0:
        acc = eng.nsq_accumulate(1, rank * 50000, 50000)
        out = _abi.Acc.from_buffer_copy(bytes(acc))
        rc = eng.L.relmc_comm_allreduce_acc(eng._h, C.byref(out))
        print("rank", rank, "allreduce rc", rc, "n", out.n, "n_fail", out.n_fail, flush=True)
if __name__ == "__main__":
    mp.set_start_method("spawn")
    path = "/tmp/relmc_uid_%d" % os.getpid()
    ps = [mp.Process(target=worker, args=(r, path)) for r in range(2)]
    [p.start() for p in ps]
    for p in ps:
        p.join(120)
        if p.is_alive(): p.kill(); print("killed a hung rank")
