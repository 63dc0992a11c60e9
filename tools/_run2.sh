R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_streamed_gpu.py tests/test_c5_gpu.py -x -q -m gpu -s 2>&1 | grep -E "passed|failed|error|assert|C5|recall" | tail -12
timeout 300 python tools/st_ab.py 1000000 bf4 2>&1 | tail -1
timeout 600 python tools/serial_ranks.py --n 8000000 --worlds 1 --fits 2 --out gpurun_out/sr_c5_new.json 2>&1 | tail -1 | cut -c1-500
