#!/bin/bash
# Run ON THE GPU BOX from the repo root (via gpurun): rocprofv3 evidence for EVERY entry of bench.py's "configs" array
# (cfg2 PIHNA K(55), general-parameter PIHNA K(119), cfg3 RIPF K(94) shipped / all terms, cfg5 HCC H(126) all rates / shipped,
# cfg5 solid H(126)).  Kernel trace + stats in one run, then the counter sets in separate --pmc passes (FETCH_SIZE and
# WRITE_SIZE each alone: together they exceed the TCC counter slots -- "Request exceeds the capabilities of the hardware").
#   tools/make_profiles_configs.sh <tag>   -> gpurun_out/profiles_<tag>_configs/   (REPORT.md + pmc_configs.json: copy into profiles/)
set -u
tag=${1:-r03}
out=gpurun_out/profiles_${tag}_configs
mkdir -p $out
export TMPDIR=/tmp
ARGS="bench.py --configs-only 1"
fail=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $ARGS > $out/configs_under_rocprof.json 2> $out/stats.err || { echo "FAILED: kernel-trace pass (see $out/stats.err)"; fail=1; }
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  if [ $fail -ne 0 ]; then break; fi
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc$i -- python3 $ARGS > $out/pmc$i.json 2> $out/pmc$i.err || { echo "FAILED: pmc pass $i ($ctrs), see $out/pmc$i.err"; tail -5 $out/pmc$i.err; fail=1; }
done
if [ $fail -ne 0 ]; then echo "profile collection FAILED: no report written"; exit 1; fi
python3 $ARGS > $out/configs.json 2> $out/configs.err || { echo "FAILED: un-profiled run"; exit 1; }
python3 tools/configs_report.py $out > $out/REPORT.md || exit 1
cat $out/REPORT.md
