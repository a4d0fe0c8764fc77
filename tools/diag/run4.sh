cd /root/repo
mkdir -p gpurun_out
export UAPS_BENCH_BACKEND=gloo UAPS_BENCH_DEVICE=0
timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --steps 3 --warmup 2 --analysis-steps 1 --exact-steps 1 --no-cpu-baseline > gpurun_out/run4.out 2> gpurun_out/run4.err
echo rc=$?
grep -v "Warning\|warn" gpurun_out/run4.err | tail -40
cut -c1-400 gpurun_out/run4.out | tail -3
