#!/bin/bash
# column-tile sweep of the streaming exact-f32 GEMM with the BN-backward reduction in its epilogue (tools/bench_red.py): AMS_PW_FORCE = s,<RM>,<NT>
for f in default s,2,2 s,2,3 s,2,4 s,2,5 s,2,6; do
  if [ "$f" = default ]; then unset AMS_PW_FORCE; else export AMS_PW_FORCE=$f; fi
  echo "== force=$f"
  python3 tools/bench_red.py 2>&1 | grep -v amdgpu.ids
done
