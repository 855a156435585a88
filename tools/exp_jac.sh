timeout -k 10 120 python tools/jacobi_probe.py 2>&1 | grep -v amdgpu
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k jacobi 2>&1 | tail -3
