R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_f32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc$i -o p -- python $R/scripts/gemm_f32_probe.py > /dev/null 2>&1
done
cd $R && python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc_f32/pmc*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32_kernel" not in r["Kernel_Name"]: continue
        a = acc[r["Kernel_Name"][:70] + " grid" + r.get("Grid_Size", "")][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    v = {c: s / n for c, (s, n) in cs.items()}
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, "cycles", round(cyc), "mfma_busy", round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(cyc * 1024, 1), 3), "wait_any", round(v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 3),
          "wait_lds", round(v.get("SQ_WAIT_INST_LDS", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 3), "lds_conf", round(v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3),
          {c: round(x) for c, x in v.items() if c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES")})
PY
rm -rf $O
