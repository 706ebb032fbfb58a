#!/bin/bash
# kernel resource usage of one .hip file: tools/kres.sh viewcrafter_amd/csrc/gemm_pp.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kres.o 2>&1 \
 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur={'name':subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()}; rows.append(cur); continue
    if cur is None:
        if 'error' in l or 'warning' in l: print(l.rstrip())
        continue
    for key in ('VGPRs','AGPRs','SGPRs','ScratchSize [bytes/lane]','VGPR Spill','SGPR Spill','Occupancy [waves/SIMD]','LDS Size [bytes/block]'):
        m=re.search(re.escape(key)+r': (\d+)',l)
        if m: cur[key]=int(m.group(1))
for r in rows:
    n=re.sub(r'\(anonymous namespace\)::','',r['name']); n=re.sub(r'vcxgemm::GemmArgs.*','',n)
    print(f\"{n[:110]:110s} V={r.get('VGPRs')} A={r.get('AGPRs')} S={r.get('SGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} vspill={r.get('VGPR Spill')} occ={r.get('Occupancy [waves/SIMD]')}\")
"
