mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_checkpoint_gpu.py tests/test_dp_gpu.py tests/test_label_net_gpu.py tests/test_normal_net_gpu.py tests/test_variants_gpu.py -m gpu -q -s > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r2i/pytest.log
grep -n "full-size step" gpurun_out/r2i/pytest.log | cut -c1-1200
bash tools/profile_step.sh r2i_nlspn --nlspn 2>&1 | tail -45
