#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r04_call4; mkdir -p $out
timeout 600 tools/microbench/pk_hazard 2>&1 | tee $out/pk_hazard.txt
