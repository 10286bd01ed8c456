python tools/infer_loop.py 32 512 40 10 0
python tools/xwr_phases.py 32
export AMS_XWR_NWD=8
python tools/infer_loop.py 32 512 40 10 0
python tools/xwr_phases.py 32
