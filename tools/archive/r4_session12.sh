#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12) > gpurun_out/r4s12_pytest.txt
tail -6 gpurun_out/r4s12_pytest.txt
