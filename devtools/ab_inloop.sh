python devtools/dwconv_bench.py > gpurun_out/r2_dwconv_rows_ab.log 2>&1; cat gpurun_out/r2_dwconv_rows_ab.log | cut -c1-150
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "dwconv or block or upsample" 2>&1 | tail -3
b() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' '; echo; }
echo "default:"; b
echo "NT=1:"; SEA_UPSAMPLE_NT=1 b
echo "NT=0:"; SEA_UPSAMPLE_NT=0 b
echo "K2 tune4 (4w plain):"; SEA_K2_FORCE=0x40 b
echo "K2 tune2 (nt-load):"; SEA_K2_FORCE=0x20 b
echo "K2 tune5 (4w nt-store):"; SEA_K2_FORCE=0x50 b
echo "K2 untuned:"; SEA_K2_FORCE=0xf0 b
echo "cnxs:"; b --backbone ConvNeXt-S_CVST --classes 151
