#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -s 2>&1 | grep -E "impact|reach of the floor|after the first contact|one at a time|passed|failed|FAILED|Error|assert" | tail -120
