cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_rccl_single.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -8
