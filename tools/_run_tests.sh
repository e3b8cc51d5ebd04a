mkdir -p gpurun_out/tests
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests/gputests.log 2>&1
tail -5 gpurun_out/tests/gputests.log
