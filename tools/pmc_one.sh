#!/bin/bash
# usage: tools/pmc_one.sh <outdir-under-gpurun_out> "<counters>" -- <python args...>   (on the GPU box, from the repo root)
# ONE rocprofv3 counter pass (kernel-trace only, as the guide prescribes) + the per-kernel summary.
set -u
out=gpurun_out/$1; ctrs=$2; shift; shift; shift
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 "$@" > $out/run.out 2> $out/run.err || { echo "FAILED: rocprofv3 pass ($ctrs)"; tail -5 $out/run.err; exit 1; }
python3 tools/pmc_summary.py $out | grep -v "^==" | awk '{k=$2" "$3" "$4; if (!(k in seen)) {seen[k]=1; print}}' | cut -c1-400
