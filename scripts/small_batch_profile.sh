#!/bin/bash
# Per-kernel time of the 10-step DDIM chunk at small environment batches (rocprofv3 kernel trace of scripts/rollout_batch_probe.py <B>).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/small_batch; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in ${@:-1 32}; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$B -o p -- python $R/scripts/rollout_batch_probe.py $B > $O/b$B.log 2>&1
  tail -1 $O/b$B.log
  python - <<P
import csv
rows = list(csv.DictReader(open("$O/b$B/p_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:110]:110s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:7.2f} us  {float(r['Percentage']):5.1f}%")
P
  rm -f $O/b$B/p_kernel_trace.csv $O/b$B/p_agent_info.csv
done
