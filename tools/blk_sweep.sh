python -m pytest tests -m gpu -q -p no:cacheprovider -k "whole_block" 2>&1 | tail -1
for t in 4x8 8x8; do echo S1 tile $t; for sh in "129 257 24 144 24 1 1" "65 129 32 192 32 1 1"; do AMS_BLK_TILE=$t python tools/block_one.py 32 $sh; done; done
for t in 4x8 2x8 4x4; do echo S2 tile $t; for sh in "257 513 16 96 24 2 0" "129 257 24 144 32 2 0" "65 129 32 192 64 2 0"; do AMS_BLK_TILE=$t python tools/block_one.py 32 $sh; done; done
