#!/bin/bash
# usage: scripts/probes/isa_stats.sh <mangled-name-regex>   -- builds with -save-temps and prints register/spill stats
cd /root/repo/neat_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I. neat_api.hip -o /tmp/isa_stats_scratch.so -save-temps=obj 2>&1 | grep -E "error|warning: v" | head
python3 - "$1" <<'PY'
import re,sys
s=open('neat_api-hip-amdgcn-amd-amdhsa-gfx950.s').read()
pat=sys.argv[1]
for m in re.finditer(r'\.name:\s+(\S+)',s):
    if re.search(pat,m.group(1)) and not m.group(1).endswith('.kd'):
        blk=s[m.start():m.start()+900]
        print(m.group(1), dict(re.findall(r'\.(vgpr_count|sgpr_count|vgpr_spill_count|agpr_count|private_segment_fixed_size):\s+(\d+)',blk)))
PY
rm -f neat_api-hip-* neat_api-host-* neat_api.hip-hip-*
