"""Compact instruction-class stream of one kernel from `hipcc -S --cuda-device-only` output, to eyeball how the MFMAs, the VALU work, LDS
traffic and waits interleave (no GPU needed).  M = MFMA, v = VALU, x = v_exp / transcendental, r / w = ds_read / ds_write, L = global / buffer
load, D = LDS-DMA load, S = store, | = s_waitcnt, B = s_barrier, ^ = branch, . = other scalar.
usage: python tools/isa_stream.py file.s kernel_name_substring [width]"""
import sys, textwrap
s = open(sys.argv[1]).read()
key = sys.argv[2]
i = s.index(key + ":") if (key + ":") in s else s.index(key)
k = s[i:]
k = k[:k.index("s_endpgm")]
def cls(l):
    op = l.split()[0]
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "x"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "r"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "w"
    if op.startswith("buffer_load") and " lds" in l: return "D"
    if op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")): return "L"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")): return "S"
    if op.startswith("s_waitcnt"): return "|"
    if op.startswith("s_barrier"): return "B"
    if op.startswith(("s_cbranch", "s_branch")): return "^"
    if op.startswith("v_"): return "v"
    if op.startswith("s_"): return "."
    return "?"
lines = [l.strip() for l in k.split("\n")]
st = "".join(cls(l) for l in lines if l and not l.startswith((".", ";", "//")) and not l.endswith(":"))
print(len(st), "instructions;", {c: st.count(c) for c in "MvxrwLDS|B"})
print("\n".join(textwrap.wrap(st, int(sys.argv[3]) if len(sys.argv) > 3 else 160)))
