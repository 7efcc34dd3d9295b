#!/bin/bash
# round 5, call 28 (the last GPU seconds of the round): the full GPU suite on the final library
set -u
O=gpurun_out/r5_28; mkdir -p $O
timeout 318 python -m pytest tests -m gpu -q -s --durations=8 > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
