#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8
