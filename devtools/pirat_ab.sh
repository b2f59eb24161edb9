#!/bin/bash
# PIR-AT outer step (configs[3]) with ONE round-6 switch flipped, alternating on this lease:  gpurun -- 'bash devtools/pirat_ab.sh [fp32|bf16]'
MODE=${1:-fp32}
r() { PIRAT_MODE=$MODE python devtools/pirat_bench.py 2>/dev/null | grep -o '"ms_per_outer_step": [0-9.]*' | tr '\n' ' '; }
for round in 1 2; do
  echo "round $round ($MODE)"
  echo "  shipped                      $(r)"
  echo "  SEA_MLP_FUSED=0              $(SEA_MLP_FUSED=0 r)"
  echo "  SEA_MLP_FUSE_LN=0            $(SEA_MLP_FUSE_LN=0 r)"
  echo "  SEA_DWCONV_AB=16             $(SEA_DWCONV_AB=16 r)"
  echo "  SEA_WINO_SPLIT_MIN_TILES=32  $(SEA_WINO_SPLIT_MIN_TILES=32 r)"
  echo "  SEA_WINO_IN_VEC4=0           $(SEA_WINO_IN_VEC4=0 r)"
  echo "  SEA_FUSE_UPSAMPLE=0 (K2u off) $(SEA_FUSE_UPSAMPLE=0 r)"
done
