"""Per-kernel register / scratch / LDS figures of a compiled TU, from the .amdhsa metadata of `hipcc -S --cuda-device-only` output.
usage: python tools/isa_meta.py file.s [name filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    g = lambda k: re.search(k + r'\s+(\d+)', body).group(1)
    print(f"{name[:90]:90s} vgpr {g('.amdhsa_next_free_vgpr'):>4s} accum_off {g('.amdhsa_accum_offset'):>4s} scratch {g('.amdhsa_private_segment_fixed_size'):>5s} lds {g('.amdhsa_group_segment_fixed_size'):>6s}")
