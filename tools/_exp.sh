cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{
for i in 1 2 3; do python tools/time_decoder.py f16x2 2; done
python tools/time_decoder.py f16x2 4
python tools/time_decoder.py f16x2 1
python tools/time_decoder.py f16 8
python -m pytest tests/test_gpu_nets.py tests/test_gpu_config3.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -5
} > gpurun_out/r05/exp4.log 2>&1
