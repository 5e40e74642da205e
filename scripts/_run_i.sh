rm -f gpurun_out/r05_i.txt
for ns in 2 1; do echo "== probe tr_adamw_ns=$ns" >> gpurun_out/r05_i.txt; MODE_HIP_OPTS=tr_adamw_ns=$ns python scripts/fused_adamw_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_i.txt; done
python -m pytest tests/test_gpu_train.py -q -k "fused" 2>&1 | tail -2 >> gpurun_out/r05_i.txt
MODE_HIP_OPTS=tr_adamw_ns=2 python -m pytest tests/test_gpu_train.py -q -k "fused" 2>&1 | tail -2 >> gpurun_out/r05_i.txt
for v in "2 1" "1 1" "1 0" "2 1" "1 1" "1 0"; do set -- $v; echo "== tr_adamw_ns=$1 fuse=$2" >> gpurun_out/r05_i.txt; MODE_HIP_OPTS=tr_adamw_ns=$1 MODE_FUSE_EXPERT_STEP=$2 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['train_ms_per_step_blocks'], d['exposed_exchange_ms'])" >> gpurun_out/r05_i.txt; done
cat gpurun_out/r05_i.txt
