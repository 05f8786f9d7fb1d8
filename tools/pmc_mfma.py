#!/usr/bin/env python3
"""Summarise one rocprofv3 SQ/GRBM counter pass over bench.py per kernel template.

usage: pmc_mfma.py <dir of the pass> <out.json>

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the counter adds the busy cycles of all 1024 matrix
pipes (256 CUs x 4 SIMDs; 16 cycles per v_mfma_f32_16x16x32_bf16), GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs
(MI355X_MICROARCH.md, "DVFS give-back").  Ratios are per kernel, dispatches summed."""
import collections, csv, glob, json, os, re, sys


def main():
    d, out = sys.argv[1:3]
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    assert f, f"no counter_collection.csv under {d}"      # (newest first: a merged gpurun_out/ keeps earlier calls' files)
    tot = collections.defaultdict(collections.Counter)
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
    rows = {}
    for k, c in tot.items():
        if c["GRBM_GUI_ACTIVE"] <= 0:
            continue
        pipes_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        rows[k] = {"launches": len(disp[k]), "mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / pipes_cycles,
                   "mfma_insts_per_launch": c["SQ_INSTS_MFMA"] / max(len(disp[k]), 1),
                   "gui_active_cycles_per_launch": c["GRBM_GUI_ACTIVE"] / 8.0 / max(len(disp[k]), 1),
                   "wait_any_frac": c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
                   "wait_inst_frac": c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
                   "active_inst_frac": c["SQ_ACTIVE_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0)}
    # the bench's dominant template: every gemm_kernel<.., A_KM=false, B_KM=false, ..> instantiation and the four-wave kernel
    # (gemm4_kernel<KIND, F16>: row-major x row-major by construction)
    dom = [k for k in rows if re.search(r"gemm_kernel<\d+, \d+, \d+, \d+, \d+, false, false, \d+(, (true|false))?>", k)
           or re.search(r"gemm4_kernel<\d+, (true|false)>", k)
           or "gemmfr_kernel<2>" in k]       # (+ the full-row kernel with the residual epilogue: the student's fc2 forward)
    assert dom, "no gemm_kernel<.., A_KM=false, B_KM=false, ..> dispatch found: the kernel-name pattern is stale"
    busy = sum(tot[k]["SQ_VALU_MFMA_BUSY_CYCLES"] for k in dom)
    act = sum(tot[k]["GRBM_GUI_ACTIVE"] for k in dom) / 8.0 * 1024.0
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                   # kernel_sources_hash(): which kernel sources this summary was taken on
    res = {"kernel_sources_hash": bench.kernel_sources_hash(), "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                     "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace over bench.py --steps 2 --warmup 1 "
                     "--teacher-lookahead 0 (one launch at a time)",
           "dominant_template": "gemm_kernel<*, A_row, B_row, *> + gemm4_kernel<*> + gemmfr_kernel<2> (the student's fc2 forward)", "mfma_busy": round(busy / max(act, 1.0), 4),
           "kernels": {k[:120]: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                       for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["gui_active_cycles_per_launch"] * kv[1]["launches"])[:24]}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
