#!/bin/bash
# Ordered kernel sequence of ONE steady-state training step (rocprofv3 kernel trace of `bench.py --mode train`), written to gpurun_out/train_seq.txt:
# start offset (us) from the step's first kernel, duration (us), gap to the previous kernel's end on the same queue, kernel name.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/train_seq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python $R/bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $R && python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/train_seq/**/p_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last occurrence of the step's first library kernel marks the last step
first = [i for i, r in enumerate(rows) if "sigma_embed_kernel" in r["Kernel_Name"] and "bwd" not in r["Kernel_Name"]]
lo, hi = first[-2], first[-1]
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = {}
with open(os.path.join(os.path.dirname(os.path.dirname(f)) if False else "gpurun_out", "train_seq.txt"), "w") as out:
    for r in rows[lo:hi]:
        q = r.get("Queue_Id", "0"); s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = e
        out.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f} q{q} {r['Kernel_Name'][:140]}\n")
    out.write(f"step span {(int(rows[hi]['Start_Timestamp']) - t0) / 1e3:.1f} us, {hi - lo} kernels\n")
PY
rm -rf $O
