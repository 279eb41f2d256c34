#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ae; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_streams_gpu.py -q -x > $O/tests.log 2>&1; rc=$?; echo tests=$rc; tail -8 $O/tests.log
