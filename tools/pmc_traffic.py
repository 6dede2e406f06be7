"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, as
MI355X_MICROARCH.md prescribes).  Usage:
  pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.csv> [<json out> <kernel substr> <algorithmic bytes> <pairs per step>]
The CSV lists mean KB per dispatch for every kernel; the JSON singles out the LARGEST-GRID dispatch family of the
named kernel (for the Winograd convolution: the conv1b launch).  gfx950 correction: FETCH_SIZE x2 (the counter reports half
of wide coalesced reads; verified on this box on an HBM-streaming kernel), WRITE_SIZE as reported."""
import collections
import csv
import json
import sys


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


def main(a):
    fe, wr = load(a[0], "FETCH_SIZE"), load(a[1], "WRITE_SIZE")
    rows = []
    for k in sorted(set(fe) | set(wr)):
        f = sum(fe.get(k, [0])) / max(len(fe.get(k, [0])), 1)
        w = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1)
        rows.append((k[0], k[1], len(fe.get(k, [])), f, w, 2 * f * 1024 + w * 1024))
    rows.sort(key=lambda r: -r[5] * max(r[2], 1))
    with open(a[2], "w") as f:
        f.write("kernel,grid_size,dispatches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_bytes_per_launch_corrected(2*FETCH+WRITE)\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%.1f,%.0f\n' % (r[0].replace('"', "'")[:120], r[1], r[2], r[3], r[4], r[5]))
    if len(a) >= 6:
        cand = [r for r in rows if a[4] in r[0]]
        best = max(cand, key=lambda r: r[1])
        json.dump({"kernel": best[0][:80], "grid_size": best[1], "dispatches": best[2], "FETCH_SIZE_KB": best[3], "WRITE_SIZE_KB": best[4],
                   "hbm_bytes_per_launch": best[5], "algorithmic_bytes_per_launch": int(a[5]), "pairs_per_step": int(a[6]) if len(a) > 6 else None,
                   "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section); "
                                 "WRITE_SIZE as reported; separate --pmc passes",
                   "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline",
                   "round": 2}, open(a[3], "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
