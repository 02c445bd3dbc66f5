#!/bin/bash
# Run ON THE GPU BOX from the repo root (via gpurun): rocprofv3 evidence for the cfg5 kernels (HCC HEX8 H(126) + solid H(126)).
#   tools/make_profiles_cfg5.sh <tag>   -> gpurun_out/profiles_<tag>_cfg5/   (copy what should be judged into profiles/)
set -u
tag=${1:-r02}
out=gpurun_out/profiles_${tag}_cfg5
mkdir -p $out
export TMPDIR=/tmp
ARGS="tools/perf_table.py hcc126 solid126"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $ARGS > $out/perf_table.jsonl 2> $out/stats.err
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc$i -- python3 $ARGS > $out/pmc$i.jsonl 2> $out/pmc$i.err || echo "pmc pass $i failed"
done
{
  echo "# rocprofv3 evidence, cfg5 kernels ($out)"
  echo
  echo "Command profiled: \`python3 $ARGS\` (HCC HEX8 H(126) row gather + coloured, solid H(126) default + two-pass; 1 warm-up + 6 timed launches each)."
  echo
  echo "## kernel-trace --stats (top kernels)"
  echo '```'
  head -12 $out/stats/*/*_kernel_stats.csv | cut -c1-220
  echo '```'
  echo "## counters: one line per kernel (first dispatch of each), sums over the device"
  echo '```'
  for k in 1 2 3 4; do python3 tools/pmc_summary.py $out/pmc$k | grep -v "^==" | awk '{k=$2" "$3" "$4; if (!(k in seen)) {seen[k]=1; print}}' | cut -c1-330; done
  echo '```'
  echo "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count correction); FP64 flop = 64 * (2 * FMA + MUL + ADD)."
  echo "## un-profiled timings (HIP events)"
  echo '```'
  python3 $ARGS
  echo '```'
} > $out/REPORT.md
cat $out/REPORT.md
