#!/bin/bash
# usage: regs.sh <file.hip> [extra flags]  -> per-kernel vgpr/scratch summary + instruction counts
f=$1; shift
b=$(basename $f .hip)
cd "$(dirname "$0")/../../normalizing_flows_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -fno-honor-nans -fno-slp-vectorize "$@" -c $f -o /tmp/$b.o -save-temps=obj 2>&1 | grep -E "error"
python3 - /tmp/$b-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import re,sys,collections
s=open(sys.argv[1]).read()
meta={}
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', s):
    meta[m.group(1)]=(m.group(2),m.group(3))
for name,(scr,vg) in meta.items():
    try:
        i=s.index('\n'+name+':'); j=s.index('s_endpgm',i)
    except ValueError: continue
    c=collections.Counter(re.findall(r'^\s+([a-z_0-9]+)', s[i:j], re.M))
    mf=sum(v for k,v in c.items() if 'mfma' in k)
    va=sum(v for k,v in c.items() if k.startswith('v_') and 'mfma' not in k)
    short=re.sub(r'_ZN2nf\d+','',name)[:60]
    print(f"{short:60s} vgpr {vg:>4s} scratch {scr:>4s} mfma {mf:4d} valu {va:5d} ds {sum(v for k,v in c.items() if k.startswith('ds_')):4d} nop {c.get('s_nop',0)}")
PY
