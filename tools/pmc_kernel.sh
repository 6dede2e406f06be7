# PMC passes (separate runs, one counter group each: MI355X_MICROARCH.md HBM / rocprofv3 section) on ONE kernel launched by tools/pmc_driver.py
# usage: bash tools/pmc_kernel.sh <what> <kernel-name substring> <out.json> [split]      e.g.  bash tools/pmc_kernel.sh gemm gemm_split_d gpurun_out/pmc_gemm.json
WHAT=$1; KSUB=$2; OUT=$3; SPLIT=$4
D=/root/repo/gpurun_out/pmc_$WHAT; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d $D/p$i -o run --output-format csv -- python /root/repo/tools/pmc_driver.py $WHAT $SPLIT > $D/p$i.log 2>&1
done
cd /root/repo
python - "$D" "$KSUB" "$OUT" "$WHAT" <<'PY'
import csv, glob, json, collections, sys
D, ksub, out, what = sys.argv[1:5]
acc = collections.defaultdict(list); dur = []; name = None
for f in glob.glob(D + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); name = r["Kernel_Name"]
for f in glob.glob(D + "/p1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
d = {c: sum(v) / len(v) for c, v in acc.items()}
o = {"what": what, "kernel": (name or "")[:100], "launch_ms_under_pmc": sum(dur) / max(len(dur), 1), "counters": d}
# FETCH_SIZE: gfx950 reports 1/2 of the bytes of WIDE (16 B / lane) coalesced reads (MI355X_MICROARCH.md, HBM): doubled for the kernels that stream that way.  The direct
# convolution (dconv_*) fetches its patches 4 bytes per lane: calibrated on the layer's known byte count (input x halo 1.33 + residual: 3.7 GB for dconv_l1, counter 3.97 GB) -> as reported.
fscale = 1 if what.startswith("dconv") else 2
if "FETCH_SIZE" in d: o["hbm_read_GB_corrected(%dx)" % fscale] = fscale * d["FETCH_SIZE"] * 1024 / 1e9
if "WRITE_SIZE" in d: o["hbm_write_GB"] = d["WRITE_SIZE"] * 1024 / 1e9
if "TCC_HIT_sum" in d: o["l2_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1)
if "SQ_WAVE_CYCLES" in d:
    wc = d["SQ_WAVE_CYCLES"]
    o["wave_cycle_shares"] = {"parked_waitcnt_or_barrier": d.get("SQ_WAIT_ANY", 0) / wc, "blocked_at_issue": d.get("SQ_WAIT_INST_ANY", 0) / wc, "issuing": d.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                              "of_which_lds_issue_stall": d.get("SQ_WAIT_INST_LDS", 0) / wc}
if "SQ_INSTS_MFMA" in d and d["SQ_INSTS_MFMA"]:
    o["per_mfma"] = {"valu": d.get("SQ_INSTS_VALU", 0) / d["SQ_INSTS_MFMA"], "lds": d.get("SQ_INSTS_LDS", 0) / d["SQ_INSTS_MFMA"], "vmem_rd": d.get("SQ_INSTS_VMEM_RD", 0) / d["SQ_INSTS_MFMA"]}
if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
    o["mfma_pipe_busy_over_sq_busy_x4simd"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * d["SQ_BUSY_CYCLES"])
ALG = {"dconv_l1out2": (32 * 196 * 360 * 272 * 4 * 2, 16), "conv1b": (64 * 64 * 720 * 540 * 4 + 64 * 64 * 360 * 270 * 4, 32), "conv1ab": (64 * 1 * 720 * 540 * 4 + 64 * 64 * 360 * 270 * 4, 32), "loftr_l1out2": (32 * 196 * 360 * 272 * 4 * 2, 16)}
if what in ALG and "FETCH_SIZE" in d and "WRITE_SIZE" in d:       # the per-launch record bench.py quotes as roofline.traffic
    o["hbm_bytes_per_launch"] = fscale * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024
    o["algorithmic_bytes_per_launch"], o["pairs_per_step"] = ALG[what]
o["correction"] = ("FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section)" if fscale == 2 else "FETCH_SIZE as reported (4-byte-per-lane reads, calibrated on the layer's known byte count)") + "; WRITE_SIZE as reported; separate --pmc passes"
if "GRBM_GUI_ACTIVE" in d and o["launch_ms_under_pmc"]:
    o["effective_clock_GHz"] = d["GRBM_GUI_ACTIVE"] / 8 / (o["launch_ms_under_pmc"] * 1e6)        # (summed over the 8 XCDs)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d: o["mfma_pipe_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (d["GRBM_GUI_ACTIVE"] / 8)
json.dump(o, open(out, "w"), indent=1)
print(json.dumps(o)[:1500])
PY
find $D -name '*.csv' -size +1M -delete 2>/dev/null
