import os, sys, time, json, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench, torch
orig = bench.train_leg
def wrapped(*a, **k):
    pr = cProfile.Profile(); pr.enable()
    r = orig(*a, **k)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    sys.stderr.write(s.getvalue()[:6000])
    sys.stderr.write("TRAIN %s\n" % json.dumps({k_: r[k_] for k_ in ("train_ms_per_step", "train_ms_per_step_blocks")}))
    return r
bench.train_leg = wrapped
bench.main()
