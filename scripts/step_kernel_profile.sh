#!/bin/bash
# Kernel profile of ONE steady-state step of `bench.py --mode $1` (train | agent): rocprofv3 kernel trace, the last complete step cut out at the
# denoiser's first kernel (sigma_embed) -> gpurun_out/step_<mode>_seq.txt (ordered: start us, duration us, gap to the previous kernel on the queue, name)
# and gpurun_out/step_<mode>_stats.txt (per kernel name: calls, total us, share of the step span).  Warm-up (MIOpen's algorithm search) is excluded.
MODE=${1:-train}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/step_prof_$MODE; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python $R/bench.py --mode $MODE --steps 4 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/step_${MODE}_bench.json 2>/dev/null
cd $R && MODE=$MODE python - <<'PY'
import collections, csv, glob, os
mode = os.environ["MODE"]
f = glob.glob(f"gpurun_out/step_prof_{mode}/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sigma_embed_kernel" in r["Kernel_Name"] and "bwd" not in r["Kernel_Name"]]
# agent mode: a step starts at the encoders, i.e. right after the previous step's last optimizer kernel; cut from sigma_embed to sigma_embed (one full period)
lo, hi = marks[-2], marks[-1]
t0 = int(rows[lo]["Start_Timestamp"]); span = (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3
prev_end = {}; agg = collections.OrderedDict()
with open(f"gpurun_out/step_{mode}_seq.txt", "w") as out:
    for r in rows[lo:hi]:
        q = r.get("Queue_Id", "0"); s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = max(e, prev_end.get(q, 0))
        out.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f} q{q} {r['Kernel_Name'][:160]}\n")
        a = agg.setdefault(r["Kernel_Name"][:160], [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    out.write(f"step span {span:.1f} us, {hi - lo} kernels\n")
tot = sum(v[1] for v in agg.values())
busy = 0; cur_s = cur_e = None                                # union of the kernel intervals = time at least one kernel is on the device
for r in rows[lo:hi]:
    s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    if cur_e is None or s > cur_e:
        busy += (cur_e - cur_s) if cur_e is not None else 0
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy = (busy + (cur_e - cur_s)) / 1e3
with open(f"gpurun_out/step_{mode}_stats.txt", "w") as out:
    out.write(f"one steady-state step of bench.py --mode {mode}: span {span:.1f} us, {hi - lo} kernels, kernel time {tot:.1f} us (queues overlap), device busy {busy:.1f} us = {100 * busy / span:.0f}% of the span\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write(f"{v[1]:10.1f} us {100 * v[1] / span:5.1f}% {v[0]:5d} x  {k}\n")
PY
rm -rf $O
