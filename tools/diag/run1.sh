cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_strided.py tests/test_res_uaps.py -x -q -m gpu 2>&1 | tail -8
timeout 900 python bench.py --net resnet50_uaps --size 640 --classes 2 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/run1_bench.json 2> gpurun_out/run1_bench.err
tail -c 300 gpurun_out/run1_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/run1_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
