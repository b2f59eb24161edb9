#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (raw output in /tmp, summaries into gpurun_out/ -> profiles/).
#   gpurun -- 'bash devtools/collect_profiles.sh r2'
R=${1:-r2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
RAW=/tmp/sea_prof_$R
mkdir -p $OUT $RAW
export TMPDIR=/tmp
cd $REPO
CASES=devtools/profile_cases.py
python3 $CASES > $OUT/${R}_cases.txt 2>$RAW/cases.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/cases -- python3 $CASES > $RAW/cases_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/fetch -- python3 $CASES > $RAW/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/write -- python3 $CASES > $RAW/write.log 2>&1
SEA_PROFILE_REPS=2 timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
  --output-format csv -d $RAW/sq -- python3 $CASES > $RAW/sq.log 2>&1 || tail -3 $RAW/sq.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/bench -- python3 bench.py --steps 10 --warmup 2 --sustain 0 --no-cpu-baseline --no-model-roofline --strict-steps 0 > $RAW/prof_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/bench_s -- python3 bench.py --steps 10 --warmup 2 --sustain 0 --no-cpu-baseline --no-model-roofline --strict-steps 0 --backbone ConvNeXt-S_CVST --classes 151 > $RAW/prof_bench_cnxs.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/bench_v -- python3 bench.py --steps 10 --warmup 2 --sustain 0 --no-cpu-baseline --no-model-roofline --strict-steps 0 --backbone vit_small_patch16_224 --classes 151 > $RAW/prof_bench_vits.log 2>&1
python3 robust-segmentation_amd/tools/summarize_profile.py --round ${R}_cnxs_c151 --bench $RAW/bench_s --title "B=8, C=151, 512x512, UperNet-ConvNeXt-S, fp32" | tail -2
python3 robust-segmentation_amd/tools/summarize_profile.py --round ${R}_vits_c151 --bench $RAW/bench_v --title "B=8, C=151, 512x512, Segmenter ViT-S/16, fp32" | tail -2
SQ=""
ls $RAW/sq/*/*_counter_collection.csv >/dev/null 2>&1 && SQ="--sq $RAW/sq"
python3 robust-segmentation_amd/tools/summarize_profile.py --round $R --bench $RAW/bench --kernels $RAW/cases --fetch $RAW/fetch \
  --write $RAW/write --cases $OUT/${R}_cases.txt $SQ 2>&1 | tail -5
cp profiles/${R}_* profiles/k2_traffic.json $OUT/ 2>/dev/null
# (after the copy: the committed logs of the same names must not replace this run's)
cp $RAW/prof_bench.log $OUT/${R}_prof_bench.log; cp $RAW/prof_bench_cnxs.log $OUT/${R}_prof_bench_cnxs.log; cp $RAW/prof_bench_vits.log $OUT/${R}_prof_bench_vits.log
ls -la $OUT | tail -12
