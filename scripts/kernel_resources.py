"""hipcc -Rpass-analysis=kernel-resource-usage of one .hip file as a table: python scripts/kernel_resources.py dasr_amd/csrc/conv.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Iinclude', '-c', src, '-o', '/tmp/_kr.o',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r'remark: [^:]*:\d+:\d+:\s+(.*?): (\d+)', line) or re.search(r'\s{4}([A-Za-z \[\]/]+): (\d+)', line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', r['name']).split('(')[0]
    if flt in n:
        print('%-70s VGPR %3d AGPR %3d spill %3d occ %2d scratch %d' % (n[:70], r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('VGPRs Spill', -1),
                                                                       r.get('Occupancy [waves/SIMD]', -1), r.get('ScratchSize [bytes/lane]', -1)))
