rm -f gpurun_out/r05_j.txt
for v in "tr_adamw_ns=1,tr_adamw_flags=4" "tr_adamw_ns=2,tr_adamw_flags=4" "tr_adamw_ns=1"; do echo "== probe $v" >> gpurun_out/r05_j.txt; MODE_HIP_OPTS=$v python scripts/fused_adamw_probe.py 2>&1 | grep "fused\|alone" >> gpurun_out/r05_j.txt; done
for v in "tr_adamw_ns=1,tr_adamw_flags=4" "tr_adamw_ns=2,tr_adamw_flags=4" "tr_adamw_ns=1" "tr_adamw_ns=1,tr_adamw_flags=4" "tr_adamw_ns=2,tr_adamw_flags=4" "tr_adamw_ns=1"; do echo "== $v" >> gpurun_out/r05_j.txt; MODE_HIP_OPTS=$v MODE_FUSE_EXPERT_STEP=1 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['train_ms_per_step_blocks'], d['exposed_exchange_ms'])" >> gpurun_out/r05_j.txt; done
cat gpurun_out/r05_j.txt
