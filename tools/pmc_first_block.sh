#!/bin/bash
# SQ counters of the first-block kernel of the 32-frame step, walking form and one tile per block (AMS_FB_WALK=0)
bash tools/pmc_cmd.sh fb_walk first_block python3 tools/infer_loop.py 32 512 6 2 0 > /dev/null
export AMS_FB_WALK=0
bash tools/pmc_cmd.sh fb_tile first_block python3 tools/infer_loop.py 32 512 6 2 0 > /dev/null
cat gpurun_out/sq_fb_walk.txt gpurun_out/sq_fb_tile.txt
