python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tests/kernel_resources.py > gpurun_out/r05_kernel_resources.txt 2>&1; tail -30 gpurun_out/r05_kernel_resources.txt
tests/profile_c5.sh r05_c5 2>&1 | tail -70
