#!/bin/bash
# static instruction counts between the phase stamps of a kernel in a lab program: tools/bqlab/isacount.sh lab2 <kernel-substring>
here="$(cd "$(dirname "$0")" && pwd)"; repo="$(cd "$here/../.." && pwd)"
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I "$repo/include" -I "$repo/khairil_tum-facade_semantic_segmentation_amd/csrc" -I "$here" -S --cuda-device-only -o /tmp/isa/$1.s "$here/$1.hip" "${@:3}" 2>/dev/null
python3 - "$1" "$2" <<'PY'
import re,sys
txt=open('/tmp/isa/%s.s'%sys.argv[1]).read().split('\n')
out=[];p=False
for l in txt:
    if l.startswith('_Z') and sys.argv[2] in l and l.rstrip().endswith(':') or (l.startswith('_Z') and sys.argv[2] in l and ':' in l): p=True
    if p: out.append(l)
    if p and 's_endpgm' in l: break
sec=0;counts={}
for l in out:
    t=l.strip()
    if t.startswith('s_memrealtime'): sec+=1
    m=re.match(r'^(v_|s_|ds_|buffer_|global_)',t)
    if not m: continue
    kind={'v_':'valu','s_':'salu','ds_':'lds','buffer_':'vmem','global_':'vmem'}[m.group(1)]
    if t.startswith('s_waitcnt') or t.startswith('s_nop'): kind='wait'
    counts.setdefault(sec,{}).setdefault(kind,0); counts[sec][kind]+=1
for s in sorted(counts): print(s,counts[s])
print('lines',len(out))
PY
