#!/usr/bin/env python3
"""Summarise rocprofv3 FETCH_SIZE / WRITE_SIZE passes per kernel template.

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>

Units and corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads (16 B per lane, LDS-DMA included) at 64 bytes, so the read side is doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Infinity-Cache hits are counted, not excluded: the figure
is fabric-side traffic of the XCD L2s, an upper bound of the HBM traffic."""
import collections, csv, glob, json, os, re, sys


def per_kernel(d, counter):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    assert f, f"no counter_collection.csv under {d}"      # (newest first: a merged gpurun_out/ keeps earlier calls' files)
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    fdir, wdir, out = sys.argv[1:4]
    ft, fc = per_kernel(fdir, "FETCH_SIZE")
    wt, wc = per_kernel(wdir, "WRITE_SIZE")
    rows = {}
    for k in set(ft) | set(wt):
        rows[k] = {"launches": int(max(fc[k], wc[k])),
                   "fetch_bytes_per_launch": ft[k] / max(fc[k], 1) * 1024 * 2,      # KiB -> B, x2 (gfx950)
                   "write_bytes_per_launch": wt[k] / max(wc[k], 1) * 1024}
    # the bench's dominant template: every gemm_kernel<.., A_KM=false, B_KM=false, ..> instantiation and the four-wave kernel
    # (gemm4_kernel<KIND, F16>: row-major x row-major by construction)
    dom = [k for k in rows if re.search(r"gemm_kernel<\d+, \d+, \d+, \d+, \d+, false, false, \d+(, (true|false))?>", k)
           or re.search(r"gemm4_kernel<\d+, (true|false)>", k)
           or "gemmfr_kernel<2>" in k]       # (+ the full-row kernel with the residual epilogue: the student's fc2 forward)
    assert dom, "no gemm_kernel<.., A_KM=false, B_KM=false, ..> dispatch found: the kernel-name pattern is stale"
    n = sum(rows[k]["launches"] for k in dom)
    fetch = sum(rows[k]["fetch_bytes_per_launch"] * rows[k]["launches"] for k in dom) / max(n, 1)
    write = sum(rows[k]["write_bytes_per_launch"] * rows[k]["launches"] for k in dom) / max(n, 1)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                   # kernel_sources_hash(): which kernel sources this summary was taken on
    res = {"kernel_sources_hash": bench.kernel_sources_hash(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 "
                     "--teacher-lookahead 0; FETCH_SIZE x2 (gfx950 128-B requests counted as 64 B); Infinity-Cache hits included",
           "dominant_template": "gemm_kernel<*, A_row, B_row, *> + gemm4_kernel<*> + gemmfr_kernel<2> (the student's fc2 forward)", "launches": n,
           "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
           "traffic_bytes_per_launch": round(fetch + write),
           "kernels": {k[:120]: {kk: (round(vv) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                       for k, v in sorted(rows.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])[:24]}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
