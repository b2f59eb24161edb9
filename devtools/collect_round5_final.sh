#!/bin/bash
# final evidence of round 5: tests with printed values, benches of the model configs, rocprofv3 summaries
#   gpurun --timeout 3000 -- 'bash devtools/collect_round5_final.sh'
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5final6
mkdir -p $O
# the rocprofv3 summaries first: collect_profiles.sh ends by copying profiles/r5_* (the fresh summaries AND every older committed
# r5 file) into gpurun_out/; copied into $O here, BEFORE the runs below write their logs, so that a fresh log replaces a
# stale file of the same name and never the other way round (the first collections of this round had it the other way round)
bash devtools/collect_profiles.sh r5 > $O/collect_profiles.log 2>&1; tail -4 $O/collect_profiles.log
cp gpurun_out/r5_* gpurun_out/k2_traffic.json $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r5_smoke.log 2>&1; tail -1 $O/r5_smoke.log
[ "$SKIP_SUITE" = 1 ] || { python -m pytest tests -m gpu -q --durations=12 > $O/r5_pytest_gpu.log 2>&1; tail -3 $O/r5_pytest_gpu.log; }
python -m pytest tests/test_gemm_split_gpu.py -q -s > $O/r5_gemm_split_tests.log 2>&1; tail -1 $O/r5_gemm_split_tests.log
python -m pytest tests/test_controller_exact_gpu.py -q -s > $O/r5_controller_exact.log 2>&1; tail -1 $O/r5_controller_exact.log
python -m pytest tests/test_full_protocol_gpu.py -q -s > $O/r5_full_protocol.log 2>&1; tail -1 $O/r5_full_protocol.log
python -m pytest tests/test_kernels_gpu.py -q -s -k attention > $O/r5_attention_tests.log 2>&1; tail -1 $O/r5_attention_tests.log
SEA_MIOU_FULL=1 python -m pytest tests/test_miou_claim_gpu.py -q -s > $O/r5_miou_vs_reference_full.log 2>&1; tail -1 $O/r5_miou_vs_reference_full.log
python -m pytest tests/test_stem_gpu.py -q -s > $O/r5_stem_tests.log 2>&1; tail -1 $O/r5_stem_tests.log
python devtools/stem_bench.py > $O/r5_stem_kernels.log 2>&1; tail -6 $O/r5_stem_kernels.log
python bench.py --steps 20 --warmup 5 > $O/r5_final_bench.log 2>/dev/null; cut -c1-220 $O/r5_final_bench.log | tail -1
SEA_GEMM_PIPE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --no-model-roofline > $O/r5_bench_pipe0.log 2>/dev/null; cut -c1-220 $O/r5_bench_pipe0.log | tail -1
SEA_FUSED_STEM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --no-model-roofline > $O/r5_bench_stem_library_path.log 2>/dev/null; cut -c1-220 $O/r5_bench_stem_library_path.log | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backbone ConvNeXt-S_CVST --classes 151 > $O/r5_bench_cnxs_c151.log 2>/dev/null; cut -c1-220 $O/r5_bench_cnxs_c151.log | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backbone vit_small_patch16_224 --classes 151 > $O/r5_bench_vits_c151.log 2>/dev/null; cut -c1-220 $O/r5_bench_vits_c151.log | tail -1
SEA_ATTN_TERMS=3 SEA_ATTN_TERMS_BWD=3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --backbone vit_small_patch16_224 --classes 151 > $O/r5_bench_vits_c151_attn_bf16x3.log 2>/dev/null; cut -c1-220 $O/r5_bench_vits_c151_attn_bf16x3.log | tail -1
python devtools/pirat_bench.py > $O/r5_pirat_config4_fp32_vs_bf16.log 2>&1; tail -2 $O/r5_pirat_config4_fp32_vs_bf16.log | cut -c1-400
bash devtools/prof_bench_steady.sh r5final6/r5_grid > /dev/null 2>&1; head -3 $O/r5_grid_by_grid.txt
python devtools/gemm_pipe_ab.py 22 > $O/r5_gemm_pipe_ab.log 2>&1; tail -1 $O/r5_gemm_pipe_ab.log
rm -f gpurun_out/gemm_pmc/summary.txt
bash devtools/gemm_split_pmc.sh 36 8192 512 512 22 > /dev/null 2>&1; bash devtools/gemm_split_pmc.sh 1 8192 384 1536 22 > /dev/null 2>&1
cp gpurun_out/gemm_pmc/summary.txt $O/r5_gemm_split_pmc.txt
ls $O | wc -l
