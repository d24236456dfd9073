#!/bin/bash
# usage: tools/pmc_run.sh <out_dir> <counter list> -- <command...>
# One rocprofv3 --pmc pass (counters only, with --kernel-trace for the kernel names; never combined with other trace
# domains).  Run from the repo root on the GPU box.
out=$1; shift
ctr=$1; shift
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$out" -o pmc --output-format csv -- "$@" > "$GRAFT_REPO_ROOT/$out.log" 2>&1
