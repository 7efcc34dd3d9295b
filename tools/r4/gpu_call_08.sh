#!/bin/bash
set -u
O=gpurun_out/r4_08; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log | cut -c1-250
