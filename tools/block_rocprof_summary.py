"""Per-dispatch reading of a `rocprofv3 --kernel-trace` run of tools/block_prof.py.

usage: python tools/block_rocprof_summary.py <rocprofv3 output dir> <c2|c4> <profiles/block_rocprof.json> <per-dispatch csv out>

Splits the trace at the marker launches (uno::gelu_pad_fwd_kernel) into warm / timed x forward / backward, and for the two TIMED
segments writes
  * the per-dispatch rows (segment, call-relative order, kernel, start ns, duration ns) to the csv (committed under profiles/),
  * per kernel: launches, mean / median / min duration,
  * per call: sum of the mean kernel durations (what a --stats table gives) and the wall span of the segment / calls
    (first start to last end: includes the gaps between kernels, overlaps of the side-stream weight gradient counted once).
bench.py reads block_rocprof.json and reports roofline.frac_rocprof = algorithmic bytes / span-per-call / 8 TB/s next to the live
HIP-event figure."""
import csv
import glob
import json
import os
import statistics
import sys

BYTES = {"c2": (1478172672, 1504387072), "c4": (2 * 8 * 32 * 64 * 64 * 20 * 4 + 4 * 32 * 32 * 16 * 16 * 8 * 8,
                                                2 * 8 * 32 * 64 * 64 * 20 * 4 + 2 * 4 * 32 * 32 * 16 * 16 * 8 * 8)}


def main():
    d, which, out_json, out_csv = sys.argv[1:5]
    calls = int(sys.argv[5]) if len(sys.argv) > 5 else 100
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no kernel_trace.csv under {d}")
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    segs, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "gelu_pad_fwd_kernel" in name:
            cur = []
            segs.append(cur)
        elif cur is not None:
            cur.append(r)
    segs = [s for s in segs if s]
    if len(segs) != 4:
        sys.exit(f"expected 4 segments between markers, found {len(segs)}")
    table = {}
    with open(out_csv, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["segment", "index", "kernel", "start_ns_rel", "duration_ns"])
        for label, seg, nbytes in (("forward", segs[1], BYTES[which][0]), ("backward", segs[3], BYTES[which][1])):
            t0 = int(seg[0]["Start_Timestamp"])
            per = {}
            for i, r in enumerate(seg):
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                nm = r["Kernel_Name"].replace("void ", "").split("(")[0]
                wr.writerow([label, i, nm, s - t0, e - s])
                per.setdefault(nm, []).append(e - s)
            span = (max(int(r["End_Timestamp"]) for r in seg) - t0) / calls
            ksum = sum(sum(v) for v in per.values()) / calls
            table[label] = {
                "calls": calls, "span_us_per_call": span / 1e3, "kernel_sum_us_per_call": ksum / 1e3,
                "algorithmic_bytes": nbytes, "frac_span": nbytes / (span * 1e-9) / 8e12, "frac_kernel_sum": nbytes / (ksum * 1e-9) / 8e12,
                "kernels": {k: {"launches_per_call": len(v) / calls, "mean_us": statistics.mean(v) / 1e3,
                                "median_us": statistics.median(v) / 1e3, "min_us": min(v) / 1e3} for k, v in per.items()}}
    allj = {}
    if os.path.exists(out_json):
        try:
            allj = json.load(open(out_json))
        except Exception:
            allj = {}
    allj[which] = table
    json.dump(allj, open(out_json, "w"), indent=1)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
