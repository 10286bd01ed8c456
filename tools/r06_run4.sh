#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/sweep_xwr_abl.sh 2>&1 | grep -v amdgpu.ids
