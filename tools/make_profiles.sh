#!/bin/bash
# Run ON THE GPU BOX from the repo root (via gpurun): collects the rocprofv3 evidence bench.py's numbers rest on.
#   tools/make_profiles.sh <tag>      -> gpurun_out/profiles_<tag>/   (copy what should be judged into profiles/)
set -u
tag=${1:-r01}
out=gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
ARGS="bench.py --steps 10 --warmup 3 --cpu-baseline 0 --configs 0 --handback 0 --two-part 0 --state-check 0"
# 1. kernel trace + stats of the bench command itself (no counters in this run)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $ARGS > $out/bench_under_rocprof.json 2> $out/stats.err
# 2. counters, each set in its own run (no trace domains besides kernel-trace)
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc$i -- python3 $ARGS > $out/pmc$i.json 2> $out/pmc$i.err || { echo "FAILED: pmc pass $i ($ctrs)"; tail -5 $out/pmc$i.err; exit 1; }
done
# 3. plain bench run (un-profiled) for the headline line
python3 bench.py --cpu-baseline 0 --configs 0 --handback 0 --two-part 0 > $out/bench.json 2> $out/bench.err
python3 tools/profile_report.py $out > $out/REPORT.md
cat $out/REPORT.md
