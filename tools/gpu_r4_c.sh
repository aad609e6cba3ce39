#!/bin/bash
mkdir -p gpurun_out
timeout 900 tools/gemm_lab2 a > gpurun_out/c_gemm_lab2_abl.log 2>&1
echo "abl exit $?"; grep -E "^c5|^big|g3|64x64 w2x2 S=1 wgs=[0-9]+ " gpurun_out/c_gemm_lab2_abl.log
timeout 1500 tools/gemm_lab2 v > gpurun_out/c_gemm_lab2.log 2>&1
echo "lab exit $?"; grep -E "^[a-z]|BEST" gpurun_out/c_gemm_lab2.log
