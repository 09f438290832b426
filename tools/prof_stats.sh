#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof_stats.sh <name> -- <command...>
# runs rocprofv3 --kernel-trace --stats, copies the kernel-stats CSV to gpurun_out/${ROUND:-r04}/<name>_kernel_stats.csv, prints top rows
name="$1"; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- "$@" > /tmp/prof_$name.log 2>&1
mkdir -p gpurun_out/${ROUND:-r04}
cp /tmp/prof_$name/p_kernel_stats.csv gpurun_out/${ROUND:-r04}/${name}_kernel_stats.csv
python3 tools/show_stats.py gpurun_out/${ROUND:-r04}/${name}_kernel_stats.csv 18
