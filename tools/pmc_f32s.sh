export BENCH_RED_ONLY=2
bash tools/pmc_cmd.sh f32s "pw_gemm_f32_s" python3 tools/bench_red.py
cat gpurun_out/sq_f32s.txt
