"""Per-kernel averages of the rocprofv3 --pmc passes written by scripts/collect_profiles.sh (gpurun_out/prof_final/pmc*/p_counter_collection.csv)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_final"
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(f"{root}/pmc*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_bf16_kernel" not in k:
            continue
        a = acc[k.split("(")[0].replace("void mode::gemm_bf16_kernel", "")][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 1])[0]):
    v = {c: s / n for c, (s, n) in cs.items()}
    n = max(n for _, n in cs.values())
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    print(f"{k}  launches/pass~{n}")
    print(f"   GUI_ACTIVE/8 {cyc / 1e3:.1f}K  MFMA busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1e6:.2f}M = {100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(cyc * 1024, 1):.1f}%"
          f"  WAIT_ANY/WAVE_CYCLES {100 * v.get('SQ_WAIT_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):.0f}%")
    print(f"   TCC hit {100 * v.get('TCC_HIT_sum', 0) / max(v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0), 1):.1f}%  FETCH_SIZE {v.get('FETCH_SIZE', 0):.0f} KB (x2 = {v.get('FETCH_SIZE', 0) * 2 * 1.024 / 1e3:.1f} MB)"
          f"  WRITE_SIZE {v.get('WRITE_SIZE', 0):.0f} KB  bank conflicts {v.get('SQ_LDS_BANK_CONFLICT', 0) / 1e3:.0f}K / {v.get('SQ_LDS_IDX_ACTIVE', 0) / 1e6:.2f}M")
