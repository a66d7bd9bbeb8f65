#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/exp_store_roof.py
bash tools/gpu_tests.sh
