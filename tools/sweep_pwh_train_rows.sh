for shape in "960 160" "576 96" "384 64" "160 960" "64 384"; do
  set -- $shape
  for cfg in "4,2 2,5" "12,2 2,5" "8,2 2,5" "4,2 2,4" "12,2 2,4" "8,2 2,4" "12,3 1,10"; do
    set -- $shape
    v=${cfg% *}; f=${cfg#* }
    export AMS_PWH_VARIANT=$v AMS_PWX_FORCE=$f
    echo -n "variant=$v tile=$f  "
    python3 tools/bench_kernel.py 17160 $1 $2 f16p 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
