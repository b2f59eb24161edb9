#!/bin/bash
# final evidence of round 3, part 1: tests with printed bounds, benches of the three model configs, PIR-AT, M8 micro-bench
#   gpurun -- 'bash devtools/collect_round3_final.sh'
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r3_smoke.log 2>&1; tail -1 $O/r3_smoke.log
python -m pytest tests -m gpu -q --durations=12 > $O/r3_pytest_gpu.log 2>&1; tail -3 $O/r3_pytest_gpu.log
python -m pytest tests/test_teacher_forced_gpu.py -q -s > $O/r3_teacher_forced_final.log 2>&1; tail -1 $O/r3_teacher_forced_final.log
python -m pytest tests/test_real_models_gpu.py tests/test_config1_parity.py -q -s > $O/r3_real_models_bounds_final.log 2>&1; tail -1 $O/r3_real_models_bounds_final.log
python -m pytest tests/test_miou_claim_gpu.py -q -s > $O/r3_miou_vs_reference.log 2>&1; tail -1 $O/r3_miou_vs_reference.log
python -m pytest tests/test_gemm_split_gpu.py -q -s > $O/r3_gemm_split_tests.log 2>&1; tail -1 $O/r3_gemm_split_tests.log
python bench.py --steps 20 --warmup 5 > $O/r3_final_bench.log 2>/dev/null; cut -c1-220 $O/r3_final_bench.log | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backbone ConvNeXt-S_CVST --classes 151 > $O/r3_bench_cnxs_c151.log 2>/dev/null; cut -c1-220 $O/r3_bench_cnxs_c151.log | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backbone vit_small_patch16_224 --classes 151 > $O/r3_bench_vits_c151.log 2>/dev/null; cut -c1-220 $O/r3_bench_vits_c151.log | tail -1
SEA_HIP_GRAPH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r3_final_bench_eager.log 2>/dev/null; cut -c1-220 $O/r3_final_bench_eager.log | tail -1
python devtools/gemm_split_bench.py > $O/r3_gemm_split_bench_final.log 2>&1; tail -1 $O/r3_gemm_split_bench_final.log
python devtools/pirat_bench.py > $O/r3_pirat_config4_fp32_vs_bf16.log 2>&1; tail -3 $O/r3_pirat_config4_fp32_vs_bf16.log
