cd $GRAFT_REPO_ROOT
P=29533
run() { timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus $1 --steps 2 --warmup 1 --backend gloo --share-device --no-cpu-baseline --dump-acc /tmp/acc_$2.json "${@:3}" > gpurun_out/$2.out 2> gpurun_out/$2.err; echo "$2 rc $? $(date +%s)"; }
run 2 q2 --workload seq --years 8 --comm host --no-time-to-cov
run 2 x2 --batch 20000 --comm native --no-time-to-cov
grep -c "communicator init failed" gpurun_out/x2.err; grep -i "watchdog\|waited\|error\|address" gpurun_out/x2.err | head -10 | cut -c1-300
run 2 f2 --batch 50000 --no-time-to-cov
grep -i "falling back\|waited\|watchdog" gpurun_out/f2.err | head -5 | cut -c1-300; head -c 300 gpurun_out/f2.out
