#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 bench.py --no-cpu --no-train --no-stream --no-api --no-bf16 --no-profile 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity'))"
