#!/bin/bash
# kernel sequence of one multi-rank (1 rank) far-field build: rocprofv3 --kernel-trace of tools/multi_far_probe.py, the launches of
# the last build in order   -> gpurun_out/<tag>/trace_multi.txt
tag=${1:-r05c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tr -o p --output-format csv -- python $R/tools/multi_far_probe.py multi 4 > $O/probe_multi.log 2>&1
grep "ms per build" $O/probe_multi.log
python - <<PY
import csv, glob
f = glob.glob("$O/tr/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = open("$O/trace_multi.txt", "w")
for r in rows[-80:]:
    out.write("%s %8.2f us  %s\n" % (r["Start_Timestamp"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:90]))
out.close()
PY
tail -70 $O/trace_multi.txt | cut -c14-130
rm -rf $O/tr
