cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_property.py tests/test_device_model_gpu.py -x -q -m gpu -k "wasserstein or digits" 2>&1 | tail -3
timeout 300 python tools/c4_time.py 2>&1 | tail -1
