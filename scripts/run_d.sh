python -m pytest tests/test_gpu_configs.py tests/test_gpu_kernels.py tests/test_bench_launch.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|FAILED" | cut -c1-300 | head -30
echo "--- runner loop, default waits"; python scripts/runner_loop.py 10 2>/dev/null | tail -1 | cut -c1-600
echo "--- runner loop, HSA_ENABLE_INTERRUPT=0"; HSA_ENABLE_INTERRUPT=0 python scripts/runner_loop.py 10 2>/dev/null | tail -1 | cut -c1-600
