#!/bin/bash
# spill / wait summary of the fused-chain kernels' ISA (neat_fused.hip):  bash scripts/probes/isa_fused.sh [extra -D flags]
R=$(cd "$(dirname "$0")/../.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/neat_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize "$@" -S --cuda-device-only $R/neat_amd/csrc/neat_fused.hip -o /tmp/fused.s 2>/dev/null
python - <<'PY'
import re
s = open('/tmp/fused.s').read()
for k in re.split(r'\n(?=_ZN\d+neat\w+:)', s):
    name = k.split(':')[0]
    if 'kernel' not in name or name.startswith('\t'):
        continue
    lines = [l.strip() for l in k.split('\n')]
    sl = sum(l.startswith('scratch_load') for l in lines)
    ss = sum(l.startswith('scratch_store') for l in lines)
    w0 = sum(l.startswith('s_waitcnt vmcnt(0)') for l in lines)
    sp = re.search(r'; ScratchSize: (\d+)', k)
    vg = re.search(r'; NumVgprs: (\d+)', k)
    print(name[4:70], 'scratch ld/st', sl, ss, 'bytes', sp and sp.group(1), 'vgpr', vg and vg.group(1), 'vmcnt(0)', w0, 'lines', len(lines))
PY
