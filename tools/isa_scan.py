"""Scan a hipcc -S listing: for one kernel (substring of the mangled name) list the control-flow / sync / spill instructions
with the running MFMA count, to check the shape of a hand-scheduled loop (where the barriers, vmcnt waits and scratch
accesses sit relative to the MFMA slots).   python tools/isa_scan.py file.s NAME_SUBSTRING [regex]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else r'scratch_|s_barrier|vmcnt\(|s_cbranch|^\.LBB|s_setprio|s_endpgm'
start = next(i for i, l in enumerate(s) if l.startswith('_Z') and key in l.split(':')[0])
end = next((i for i in range(start + 1, len(s)) if s[i].startswith('_Z') and s[i].rstrip().endswith(':') or '.Lfunc_end' in s[i]), len(s))
m = d = v = 0
for i in range(start, end):
    l = s[i]
    if 'v_mfma' in l: m += 1
    if re.search(r'\bds_read', l): d += 1
    if re.search(r'buffer_load.*lds', l): v += 1
    if re.search(pat, l):
        print(f"{i-start:6d} mfma={m:4d} dsr={d:4d} dma={v:3d}  {l.strip()[:100]}")
