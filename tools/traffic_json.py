"""profiles/hbm_traffic.json + the text summary from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh.
usage: python tools/traffic_json.py <pmc dir> <out json> <out txt>"""
import collections, csv, glob, json, re, sys
root, out_json, out_txt = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "uno::" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
method = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; read = 2*FETCH_SIZE*1024 (gfx950 "
          "correction), write = WRITE_SIZE*1024; mean over the launches of the run")
js, lines = {}, []
for k in sorted(acc):
    if "FETCH_SIZE" not in acc[k] or "WRITE_SIZE" not in acc[k]:
        continue
    fs = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
    ws = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
    e = {"bytes_per_launch": 2 * fs * 1024 + ws * 1024, "read_bytes": 2 * fs * 1024, "write_bytes": ws * 1024,
         "launches": len(acc[k]["FETCH_SIZE"]), "method": method}
    js[k] = e
    base = re.sub(r"<.*", "", k)          # the library's profile records name some kernels without template arguments
    if base != k and base not in js:
        # launch-weighted mean over the instantiations
        ks = [q for q in acc if re.sub(r"<.*", "", q) == base and "FETCH_SIZE" in acc[q] and "WRITE_SIZE" in acc[q]]
        n = sum(len(acc[q]["FETCH_SIZE"]) for q in ks)
        f2 = sum(sum(acc[q]["FETCH_SIZE"]) for q in ks) / n
        w2 = sum(sum(acc[q]["WRITE_SIZE"]) for q in ks) / sum(len(acc[q]["WRITE_SIZE"]) for q in ks)
        js[base] = {"bytes_per_launch": 2 * f2 * 1024 + w2 * 1024, "read_bytes": 2 * f2 * 1024, "write_bytes": w2 * 1024,
                    "launches": n, "method": method}
    lines.append(f"{k}\n   FETCH_SIZE {fs:16.0f} KiB (n={len(acc[k]['FETCH_SIZE'])})\n   WRITE_SIZE {ws:16.0f} KiB (n={len(acc[k]['WRITE_SIZE'])})")
json.dump(js, open(out_json, "w"), indent=1)
open(out_txt, "w").write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline\n"
                         "# mean counter value per launch, per kernel.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B\n"
                         "# requests as 64 B -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE taken as is.\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
