#!/bin/bash
# repeat the non-blocking shim cases and show how a failing one differs
for i in 1 2 3 4 5 6; do
  timeout -k 10 120 python -u -m pytest tests/test_gpu_cpp_shim.py -m gpu -q --timeout 100 --timeout-method thread -k "nonblocking" 2>&1 | grep -E "^E  .*AssertionError|^FAILED|passed|failed" | cut -c1-600
done
