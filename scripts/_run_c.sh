python -m pytest tests/test_gpu_train.py -x -q -k "fused" 2>&1 | tail -15 > gpurun_out/r05_c_tests.txt
python scripts/pp_l2touch_probe.py s2 s4 t2 > gpurun_out/r05_pp_l2touch.txt 2>&1
rm -f gpurun_out/r05_pp_l2touch_chain.txt
for i in 1 2; do for t in base s2 s4 t2; do if [ $t = base ]; then unset MODE_HIP_LIB; else export MODE_HIP_LIB=$PWD/mode_diffusion_policy_amd/libmode_hip_$t.so; fi; echo "== $t" >> gpurun_out/r05_pp_l2touch_chain.txt; python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"avg_launch_us\"])" >> gpurun_out/r05_pp_l2touch_chain.txt; done; done
unset MODE_HIP_LIB
for i in 1 2; do
for v in "MODE_FUSE_EXPERT_STEP=1 MODE_OPT_OVERLAP=0" "MODE_FUSE_EXPERT_STEP=1 MODE_OPT_OVERLAP=1" "MODE_FUSE_EXPERT_STEP=0 MODE_OPT_OVERLAP=1"; do echo "== $v" >> gpurun_out/r05_c_train.txt; env $v python bench.py --mode train --no-cpu-baseline 2>>gpurun_out/r05_c_train.err | tail -1 >> gpurun_out/r05_c_train.txt; done; done
cat gpurun_out/r05_c_tests.txt gpurun_out/r05_pp_l2touch.txt gpurun_out/r05_pp_l2touch_chain.txt; cut -c1-600 gpurun_out/r05_c_train.txt
