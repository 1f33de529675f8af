#!/bin/bash
# rocprofv3 kernel stats of tools/spatial_bench.py for the regular library and every variant given
# usage (on the GPU box): bash tools/prof_variants.sh name1 name2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ "$v" = base ]; then unset TECM_LIB; else export TECM_LIB=$R/tec-mollm_amd/tecmollm/variants/libtecmollm_hip_$v.so; fi
  rm -rf $R/gpurun_out/pv_$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pv_$v -o p -- python3 $R/tools/spatial_bench.py > $R/gpurun_out/pv_$v.log 2>&1
  echo "== $v"; grep "spatial fwd\|stamps" $R/gpurun_out/pv_$v.log
  python3 $R/tools/rocpd_stats.py $R/gpurun_out/pv_$v/p_results.db --top 3 | head -3
done
