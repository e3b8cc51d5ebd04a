mkdir -p gpurun_out/tests
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests/gputests.log 2>&1
grep -E "passed|failed" gpurun_out/tests/gputests.log | tail -3
grep -B5 -A25 "Error\|FAILED" gpurun_out/tests/gputests.log | head -80
