python scripts/_dbg_fused.py > gpurun_out/r05_d_dbg.txt 2>&1
python -m pytest tests/test_gpu_train.py -q -k "epilogue_gemm" 2>&1 | tail -15 >> gpurun_out/r05_d_dbg.txt
MODE_FUSE_EXPERT_STEP=1 bash scripts/step_kernel_profile.sh train
cp gpurun_out/step_train_stats.txt gpurun_out/r05_d_train_fused_stats.txt; cp gpurun_out/step_train_seq.txt gpurun_out/r05_d_train_fused_seq.txt
cat gpurun_out/r05_d_dbg.txt; head -12 gpurun_out/r05_d_train_fused_stats.txt
