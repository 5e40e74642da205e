python -m pytest tests/test_gpu_train.py -q -k "fused" 2>&1 | tail -6 > gpurun_out/r05_e.txt
MODE_FUSE_EXPERT_STEP=1 bash scripts/step_kernel_profile.sh train
head -6 gpurun_out/step_train_stats.txt >> gpurun_out/r05_e.txt
MODE_FUSE_EXPERT_STEP=1 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-220 >> gpurun_out/r05_e.txt
cat gpurun_out/r05_e.txt
