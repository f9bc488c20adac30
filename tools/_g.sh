cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_sparse_kernels_gpu.py -x -q -m gpu 2>&1 | tail -5
cd tools && timeout 60 python spmv_time.py 2>&1 | tail -5
