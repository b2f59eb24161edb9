#!/bin/bash
# final evidence of round 6: tests with printed values, benches of the model configs, rocprofv3 summaries, the round's A/B knobs
#   gpurun --timeout 3300 -- 'bash devtools/collect_round6_final.sh'
# SKIP_SUITE=1 leaves the whole-suite run out; SKIP_MIOU=1 the SEA_MIOU_FULL run (every committed reference part).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6final
mkdir -p $O
# the rocprofv3 summaries first (collect_profiles.sh ends by copying profiles/r6_* into gpurun_out/; copied into $O BEFORE the runs
# below write their logs, so that a fresh log replaces a stale file of the same name and never the other way round)
bash devtools/collect_profiles.sh r6 > $O/collect_profiles.log 2>&1; tail -4 $O/collect_profiles.log
cp gpurun_out/r6_* gpurun_out/k2_traffic.json $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r6_smoke.log 2>&1; tail -1 $O/r6_smoke.log
[ "$SKIP_SUITE" = 1 ] || { python -m pytest tests -m gpu -q --durations=12 > $O/r6_pytest_gpu.log 2>&1; tail -3 $O/r6_pytest_gpu.log; }
for t in gemm_split mlp_fused classifier l2 controller_exact full_protocol stem; do
  python -m pytest tests/test_${t}_gpu.py -q -s > $O/r6_${t}_tests.log 2>&1; echo "$t: $(tail -1 $O/r6_${t}_tests.log)"
done
python -m pytest tests/test_kernels_gpu.py -q -s -k attention > $O/r6_attention_tests.log 2>&1; tail -1 $O/r6_attention_tests.log
[ "$SKIP_MIOU" = 1 ] || { SEA_MIOU_FULL=1 python -m pytest tests/test_miou_claim_gpu.py -q -s > $O/r6_miou_vs_reference_full.log 2>&1; tail -1 $O/r6_miou_vs_reference_full.log; }
python bench.py --steps 20 --warmup 5 > $O/r6_final_bench.log 2>/dev/null; cut -c1-220 $O/r6_final_bench.log | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backbone ConvNeXt-S_CVST --classes 151 > $O/r6_bench_cnxs_c151.log 2>/dev/null; cut -c1-220 $O/r6_bench_cnxs_c151.log | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --backbone vit_small_patch16_224 --classes 151 > $O/r6_bench_vits_c151.log 2>/dev/null; cut -c1-220 $O/r6_bench_vits_c151.log | tail -1
python devtools/pirat_bench.py > $O/r6_pirat_config4_fp32_vs_bf16.log 2>&1; tail -2 $O/r6_pirat_config4_fp32_vs_bf16.log | cut -c1-400
# the round's switches, alternating on this lease (ms per step of a 40-step window each)
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-model-roofline --strict-steps 0 --sustain 0 "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; }
{
  echo "# UperNet-ConvNeXt-T C=21, B=8, 512x512: ms per step of a 40-step window, shipped build vs ONE switch flipped, alternating (two rounds)"
  for r in 1 2; do
    echo "round $r"
    echo "  shipped                                     $(b)"
    echo "  SEA_MLP_FUSED=0        (two GEMMs per MLP)    $(SEA_MLP_FUSED=0 b)"
    echo "  SEA_MLP_FUSE_LN=0      (LayerNorm separate)   $(SEA_MLP_FUSE_LN=0 b)"
    echo "  SEA_DWCONV_AB=16       (plain dwconv kernels) $(SEA_DWCONV_AB=16 b)"
    echo "  SEA_WINO_SPLIT_MIN_TILES=32 (PSP on library)  $(SEA_WINO_SPLIT_MIN_TILES=32 b)"
    echo "  SEA_CLASSIFIER=0       (library classifier)   $(SEA_CLASSIFIER=0 b)"
    echo "  SEA_FUSE_CLS_GATE=0    (separate gate pass)   $(SEA_FUSE_CLS_GATE=0 b)"
    echo "  SEA_WINO_IN_VEC4=0     (2 channels per lane)  $(SEA_WINO_IN_VEC4=0 b)"
    echo "  SEA_TAP_INNER=0        (general tap weights)  $(SEA_TAP_INNER=0 b)"
    echo "  SEA_GEMM_BIG_WAVES=4   (4 x 128x128 waves)    $(SEA_GEMM_BIG_WAVES=4 b)"
  done
  echo "# UperNet-ConvNeXt-S C=151 / Segmenter ViT-S C=151: K2u (fused final up-sampling + loss) on (shipped for C >= 96) vs off"
  echo "  cnxs shipped $(b --backbone ConvNeXt-S_CVST --classes 151)   --no-fuse-upsample $(b --backbone ConvNeXt-S_CVST --classes 151 --no-fuse-upsample)"
  echo "  vits shipped $(b --backbone vit_small_patch16_224 --classes 151)   --no-fuse-upsample $(b --backbone vit_small_patch16_224 --classes 151 --no-fuse-upsample)"
} > $O/r6_inloop_switches_ab.log 2>&1; cat $O/r6_inloop_switches_ab.log
bash devtools/prof_bench_steady.sh r6final/r6_grid > /dev/null 2>&1; head -3 $O/r6_grid_by_grid.txt
bash devtools/prof_bench_steady.sh r6final/r6_cnxs_c151_grid --backbone ConvNeXt-S_CVST --classes 151 > /dev/null 2>&1
bash devtools/prof_bench_steady.sh r6final/r6_vits_c151_grid --backbone vit_small_patch16_224 --classes 151 > /dev/null 2>&1
python devtools/k2u_bench.py > $O/r6_k2u_bench_final.log 2>&1; tail -13 $O/r6_k2u_bench_final.log
python devtools/dwconv_bench.py > $O/r6_dwconv_pipe_ab.log 2>&1; tail -2 $O/r6_dwconv_pipe_ab.log
python devtools/classifier_bench.py > $O/r6_classifier_bench.log 2>&1; tail -4 $O/r6_classifier_bench.log
python devtools/mlp_fused_bench.py > $O/r6_mlp_fused_bench_final.log 2>&1; tail -6 $O/r6_mlp_fused_bench_final.log
rm -f gpurun_out/gemm_pmc/summary.txt
bash devtools/gemm_split_pmc.sh 36 8192 512 512 22 > /dev/null 2>&1; bash devtools/gemm_split_pmc.sh 1 8192 384 1536 22 > /dev/null 2>&1
cp gpurun_out/gemm_pmc/summary.txt $O/r6_gemm_split_pmc.txt 2>/dev/null
ls $O | wc -l
