cd /root/repo
mkdir -p gpurun_out
timeout 900 python bench.py --net resnet50_uaps --size 640 --classes 2 --batch 8 --steps 5 --warmup 2 > gpurun_out/run1_bench.json 2> gpurun_out/run1_bench.err
tail -c 300 gpurun_out/run1_bench.err
