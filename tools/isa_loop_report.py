"""Compressed instruction trace of a HIP kernel's basic blocks, to see what hipcc's scheduler did to a software pipeline: per block, the
sequence of matrix instructions (M), vector-ALU runs (V xN), global / buffer loads (GL), LDS reads / writes (DSR / DSW), barriers and
every s_waitcnt with its counters.  What to look for (each found and fixed this way in round 3):
  * a prefetch whose loads were sunk BELOW the MFMAs they were meant to overlap (gemm_bf16x3: `GL .. W:vmcnt .. V(split) .. DSW .. M x48`);
  * `W:vmcnt(0)` inside a steady-state loop whose loads are supposed to stay in flight (conv_gemm_bf16: a load under a condition made
    the wait-count bookkeeping drain the queue every third step);
  * a select / mask on a freshly loaded value right after the load (attention: the wave waited for its loads before multiplying);
  * filter / operand requests spread between the MFMAs of a phase instead of issued at its top (wino_bf16x3: WB_FENCE).
No GPU needed: python tools/isa_loop_report.py map-free-reloc_amd/csrc/gemm_bf16x3.hip gemm_bf16x3_kernelILi0 [--min-mfma 8] [extra hipcc flags]"""
import re
import subprocess
import sys
import tempfile


def report(src, kernel_substr, min_mfma=1, extra=()):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-ffp-contract=fast", "-S", "--cuda-device-only", src, "-o", f.name] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        lines = open(f.name).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and kernel_substr in l]
    if not starts:
        sys.exit(f"no kernel symbol containing {kernel_substr!r}; candidates: " + ", ".join(sorted({l[:-1] for l in lines if re.match(r'^_Z\w*:$', l)}))[:2000])
    start = starts[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], ["entry", []]
    blocks.append(cur)
    for l in lines[start:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = [l.split(":")[0], []]
            blocks.append(cur)
        elif l.strip() and not l.strip().startswith((".", ";")):
            cur[1].append(l.strip())
    short = {"global_load_dwordx4": "GL4", "global_load_dwordx2": "GL2", "global_load_dword": "GL1", "buffer_load_dwordx4": "BL4", "ds_read_b128": "DSR",
             "ds_write_b128": "DSW", "ds_read2_b64": "DSR2", "ds_write_b64": "DSW64", "ds_write2_b64": "DSW2"}
    print(lines[start][:-1])
    for name, ins in blocks:
        nm = sum(1 for i in ins if i.startswith("v_mfma"))
        if nm < min_mfma:
            continue
        seq = []
        for i in ins:
            op = i.split()[0]
            if op.startswith("v_mfma"):
                seq.append("M")
            elif op == "s_waitcnt":
                seq.append("W:" + i.split(None, 1)[1].replace(" ", ""))
            elif op.startswith(("global_load", "buffer_load", "ds_read", "ds_write", "scratch_")):
                seq.append(short.get(op, op))
            elif op in ("s_barrier",) or op.startswith("s_cbranch"):
                seq.append(op)
            elif op.startswith("v_"):
                seq.append("V")
        out, prev, cnt = [], None, 0
        for x in seq + [None]:
            if x == prev:
                cnt += 1
                continue
            if prev is not None:
                out.append(f"{prev} x{cnt}" if cnt > 1 else prev)
            prev, cnt = x, 1
        nv = sum(1 for i in ins if i.startswith("v_") and not i.startswith("v_mfma"))
        print(f"{name}: {len(ins)} instructions, {nm} MFMA, {nv} VALU")
        print("    " + " | ".join(out))


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    a = sys.argv[3:]
    mm = 1
    if "--min-mfma" in a:
        k = a.index("--min-mfma"); mm = int(a[k + 1]); a = a[:k] + a[k + 2:]
    report(sys.argv[1], sys.argv[2], mm, a)
