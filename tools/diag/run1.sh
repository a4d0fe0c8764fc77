cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_strided.py tests/test_res_uaps.py -x -q -m gpu 2>&1 | tail -15
