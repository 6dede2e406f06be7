"""Which host call is running while the GPU has nothing to do?  Reads rocprofv3's kernel trace and HIP API trace of one process
(rocprofv3 --kernel-trace --hip-trace --output-format csv), finds the intervals in which no kernel executes (longer than --min-us, between
the first and the last kernel of the busiest part of the run) and charges each to the HIP API calls that overlap it.
python tools/idle_gaps.py <dir with *_kernel_trace.csv / *_hip_api_trace.csv> [--min-us 300] [--anchor KERNEL_SUBSTRING] [--json out.json]"""
import argparse, collections, csv, glob, json, os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir"); ap.add_argument("--min-us", type=float, default=300.0); ap.add_argument("--anchor", default="")
    ap.add_argument("--skip-anchors", type=int, default=0, help="ignore everything before the N-th launch of the anchor kernel (warm-up run)")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    kf = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    hf = glob.glob(os.path.join(a.dir, "**", "*hip_api_trace.csv"), recursive=True)[0]
    ker = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kf))]
    ker.sort()
    t_begin = ker[0][0]
    if a.anchor:
        hits = [k for k in ker if a.anchor in k[2]]
        if len(hits) > a.skip_anchors:
            t_begin = hits[a.skip_anchors][1]
    ker = [k for k in ker if k[0] >= t_begin]
    api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(hf)) if int(r["End_Timestamp"]) >= t_begin]
    api.sort()
    gaps, busy_end = [], ker[0][1]
    for s, e, _ in ker[1:]:
        if s - busy_end > a.min_us * 1e3:
            gaps.append((busy_end, s))
        busy_end = max(busy_end, e)
    span = ker[-1][1] - ker[0][0]
    busy = span - sum(g[1] - g[0] for g in gaps)
    charge, n_by = collections.Counter(), collections.Counter()
    uncovered = 0
    j = 0
    for g0, g1 in gaps:
        cov = 0
        while j < len(api) and api[j][1] < g0:
            j += 1
        k = j
        while k < len(api) and api[k][0] < g1:
            ov = min(api[k][1], g1) - max(api[k][0], g0)
            if ov > 0:
                charge[api[k][2]] += ov; n_by[api[k][2]] += 1; cov += ov
            k += 1
        uncovered += max(0, (g1 - g0) - cov)
    out = {"span_ms": span / 1e6, "kernel_busy_ms_incl_short_gaps": busy / 1e6, "idle_ms_in_gaps": sum(g[1] - g[0] for g in gaps) / 1e6, "gaps": len(gaps),
           "min_gap_us": a.min_us, "largest_gaps_ms": sorted(((g[1] - g[0]) / 1e6 for g in gaps), reverse=True)[:8],
           "idle_ms_by_overlapping_hip_call": {f: round(v / 1e6, 2) for f, v in charge.most_common(10)},
           "calls_overlapping_gaps": {f: n_by[f] for f, _ in charge.most_common(10)},
           "idle_ms_with_no_hip_call_in_flight (python between calls)": round(uncovered / 1e6, 2)}
    print(json.dumps(out, indent=1))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
