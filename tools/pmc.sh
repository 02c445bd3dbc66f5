#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> -- <python args...>   (run on the GPU box, from the repo root)
# Separate rocprofv3 passes (kernel-trace/stats only; counters in their own runs, as the guide prescribes).
set -u
out=gpurun_out/$1; shift; shift
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 "$@" > $out/stats.out 2> $out/stats.err
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc$i -- python3 "$@" > $out/pmc$i.out 2> $out/pmc$i.err || echo "pass $i failed"
done
ls -R $out | head -40
