#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aa; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_entrypoints_gpu.py -q -x > $O/tests.log 2>&1; rc=$?; echo tests=$rc; tail -3 $O/tests.log
