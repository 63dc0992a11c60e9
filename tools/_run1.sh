cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_streamed_gpu.py -x -q -m gpu -k "rccl or nccl" 2>&1 | tail -25
