#!/bin/bash
# final evidence of round 4: tests with printed values, benches of the three model configs, rocprofv3 summaries
#   gpurun --timeout 3000 -- 'bash devtools/collect_round4_final.sh'
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r4_smoke.log 2>&1; tail -1 $O/r4_smoke.log
python -m pytest tests -m gpu -q --durations=12 > $O/r4_pytest_gpu.log 2>&1; tail -3 $O/r4_pytest_gpu.log
python -m pytest tests/test_teacher_forced_gpu.py -q -s > $O/r4_teacher_forced.log 2>&1; tail -1 $O/r4_teacher_forced.log
python -m pytest tests/test_real_models_gpu.py tests/test_config1_parity.py -q -s > $O/r4_real_models_bounds.log 2>&1; tail -1 $O/r4_real_models_bounds.log
python -m pytest tests/test_miou_claim_gpu.py -q -s > $O/r4_miou_vs_reference.log 2>&1; tail -1 $O/r4_miou_vs_reference.log
python -m pytest tests/test_controller_exact_gpu.py -q -s > $O/r4_controller_exact.log 2>&1; tail -1 $O/r4_controller_exact.log
python -m pytest tests/test_full_protocol_gpu.py -q -s > $O/r4_full_protocol.log 2>&1; tail -1 $O/r4_full_protocol.log
python -m pytest tests/test_gemm_split_gpu.py -q -s > $O/r4_gemm_split_tests.log 2>&1; tail -1 $O/r4_gemm_split_tests.log
python bench.py --steps 20 --warmup 5 > $O/r4_final_bench.log 2>/dev/null; cut -c1-220 $O/r4_final_bench.log | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backbone ConvNeXt-S_CVST --classes 151 > $O/r4_bench_cnxs_c151.log 2>/dev/null; cut -c1-220 $O/r4_bench_cnxs_c151.log | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backbone vit_small_patch16_224 --classes 151 > $O/r4_bench_vits_c151.log 2>/dev/null; cut -c1-220 $O/r4_bench_vits_c151.log | tail -1
SEA_ATTN_TERMS_BWD=2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backbone vit_small_patch16_224 --classes 151 > $O/r4_bench_vits_c151_attn_bwd2.log 2>/dev/null; cut -c1-220 $O/r4_bench_vits_c151_attn_bwd2.log | tail -1
SEA_GEMM_TERMS_BWD=2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r4_bench_bwd_bf16x2.log 2>/dev/null; cut -c1-220 $O/r4_bench_bwd_bf16x2.log | tail -1
SEA_HIP_GRAPH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 > $O/r4_final_bench_eager.log 2>/dev/null; cut -c1-220 $O/r4_final_bench_eager.log | tail -1
bash devtools/prof_bench_steady.sh r4final/r4_grid > /dev/null 2>&1; head -3 $O/r4_grid_by_grid.txt
bash devtools/collect_profiles.sh r4 > $O/collect_profiles.log 2>&1; tail -4 $O/collect_profiles.log
cp gpurun_out/r4_* $O/ 2>/dev/null
ls $O | wc -l
