"""Per-kernel averages of the rocprofv3 --pmc passes written by scripts/collect_profiles.sh (<root>/pmc*/p_counter_collection.csv).

Prints a table and writes a machine-readable summary (default profiles/r05_gemm_pmc.json) that bench.py reads `roofline.traffic` from:
HBM bytes per launch = FETCH_SIZE x 2 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both in KB.

    python scripts/pmc_summary.py gpurun_out/prof_final profiles/r05_gemm_pmc.json
"""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_final"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r05_gemm_pmc.json"
ROLES = [("gemm_pp_kernel<4, true, 3", "expert_up_projection"), ("gemm_pp_kernel<0, true, 3", "expert_down_projection"),
         ("gemm_bf16_kernel<128, 64, 2, 2, 2, 1,", "qkv_projection"), ("gemm_bf16_kernel<64, 64, 2, 2, 3, 5,", "c_proj_residual_ln2"),
         ("combine_norm_row_kernel", "combine_ln1"), ("attn_bf16_kernel", "attention"), ("qkv_attn_kernel", "qkv_projection+attention")]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(f"{root}/pmc*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        role = next((ro for sub, ro in ROLES if sub in k), None)
        if role is None:
            continue
        a = acc[(role, next(sub for sub, ro in ROLES if ro == role))][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
res = {"source": root, "note": "averages per launch over all sampled launches of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras`; "
                                "one rocprofv3 --pmc pass per counter group, never combined with trace domains", "kernels": {}}
for (role, sub), cs in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 1])[0]):
    v = {c: s / n for c, (s, n) in cs.items()}
    n = max(n for _, n in cs.values())
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    ent = {"kernel_substr": sub, "launches_sampled": n, "gui_active_cycles": round(cyc), "mfma_busy_frac": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(cyc * 1024, 1), 4),
           "wait_any_frac": round(v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 4),
           "tcc_hit_frac": round(v.get("TCC_HIT_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1), 4),
           "fetch_size_kb": round(v.get("FETCH_SIZE", 0)), "write_size_kb": round(v.get("WRITE_SIZE", 0)),
           "hbm_bytes_per_launch": round((v.get("FETCH_SIZE", 0) * 2 + v.get("WRITE_SIZE", 0)) * 1024) if "FETCH_SIZE" in v and "WRITE_SIZE" in v else None,
           "lds_bank_conflict_cycles": round(v.get("SQ_LDS_BANK_CONFLICT", 0)), "lds_active_cycles": round(v.get("SQ_LDS_IDX_ACTIVE", 0))}
    res["kernels"][role] = ent
    print(f"{role:26s} {sub}")
    print(f"   launches/pass~{n}  GUI_ACTIVE/8 {cyc / 1e3:.1f}K  MFMA busy {100 * ent['mfma_busy_frac']:.1f}%  WAIT_ANY/WAVE_CYCLES {100 * ent['wait_any_frac']:.0f}%  TCC hit {100 * ent['tcc_hit_frac']:.1f}%")
    print(f"   FETCH_SIZE {ent['fetch_size_kb']} KB (x2 = {ent['fetch_size_kb'] * 2 * 1.024 / 1e3:.1f} MB)  WRITE_SIZE {ent['write_size_kb']} KB  -> HBM bytes/launch {ent['hbm_bytes_per_launch']}"
          f"  LDS bank conflicts {ent['lds_bank_conflict_cycles'] / 1e3:.0f}K / {ent['lds_active_cycles'] / 1e6:.2f}M")
json.dump(res, open(out, "w"), indent=1)
print("wrote", out)
