#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_quality_0_1.py -x -q -m gpu > gpurun_out/r05_q01_tests.log 2>&1
tail -3 gpurun_out/r05_q01_tests.log
timeout 600 python tools/q01_probe.py > gpurun_out/r05_q01_probe.jsonl 2> gpurun_out/r05_q01_probe.err
cat gpurun_out/r05_q01_probe.jsonl; tail -3 gpurun_out/r05_q01_probe.err
