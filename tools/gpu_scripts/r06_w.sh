#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_learner_gpu.py tests/test_curriculum_gpu.py -q -x > $O/tests.log 2>&1; rc=$?; echo tests=$rc; tail -5 $O/tests.log
