rm -f gpurun_out/r05_f.txt
for ns in 2 1 2 1; do echo "== tr_adamw_ns=$ns" >> gpurun_out/r05_f.txt; MODE_HIP_OPTS=tr_adamw_ns=$ns MODE_FUSE_EXPERT_STEP=1 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['train_ms_per_step_blocks'])" >> gpurun_out/r05_f.txt; done
MODE_HIP_OPTS=tr_adamw_ns=2 MODE_FUSE_EXPERT_STEP=1 bash scripts/step_kernel_profile.sh train
head -4 gpurun_out/step_train_stats.txt >> gpurun_out/r05_f.txt
cat gpurun_out/r05_f.txt
