# usage (build container): bash tools/kres.sh [extra hipcc flags] -- registers / scratch / occupancy of every step-kernel instance
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=12 -mllvm -amdgpu-sched-strategy=max-memory-clause "$@" \
  -c "$(dirname "$0")/../fleetrl_amd/csrc/fleet_kernels.hip" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; continue
    for k in ('VGPRs','SGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','LDS Size [bytes/block]'):
        m=re.search(re.escape(k)+r': (\d+)',line)
        if m and cur is not None and ' '+k in line: cur[k]=int(m.group(1))
    if cur and 'LDS Size [bytes/block]' in cur:
        n=cur['name']
        m=re.search(r'fleet_step_kernelILi(\d+)ELi(\d)ELb(\d)ELb(\d)',n)
        if m: print('step<G=%s,DEG=%s,MULTI=%s,WIDE=%s> vgpr=%d sgpr=%d scratch=%d occ=%d'%(m.group(1),m.group(2),m.group(3),m.group(4),cur.get('VGPRs',-1),cur.get('SGPRs',-1),cur.get('ScratchSize [bytes/lane]',-1),cur.get('Occupancy [waves/SIMD]',-1)))
        cur=None
"
