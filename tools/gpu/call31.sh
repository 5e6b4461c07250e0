#!/bin/bash
timeout 600 python -m pytest "tests/test_hip_parity.py::test_infadp_baseline_shapes_vs_reference" -q -m gpu -x 2>&1 | grep -E "assert|Error|error|rel|passed|failed" | head -30
