#!/bin/bash
# final check of the round: the whole GPU suite + smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02as; mkdir -p $O
timeout 110 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^W2026\|^E2026" | tail -6 > $O/pytest.log
cat $O/pytest.log
