"""Cost of synchronising neighbouring workgroups through flags in global memory instead of a kernel boundary (dasr_probe_tile_sync,
include/dasr_hip_bench.h): the feasibility number for a persistent per-RDB kernel.
python scripts/micro_sync.py [blocks=512] [stages=400]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dasr_amd import _lib, engine
engine.ensure_runtime_ready()
L = _lib.bench_lib()   # libdasr_bench.so (python -m dasr_amd.build --bench)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 512
stages = int(sys.argv[2]) if len(sys.argv) > 2 else 400
us, to, stale = C.c_float(0), C.c_int32(0), C.c_int32(0)
print('blocks %d, stages %d' % (blocks, stages))
print('%-11s %-10s %-38s %10s %8s %8s' % ('tile', 'neighbours', 'scope', 'us/stage', 'timeout', 'stale'))
for words in (256, 8192):   # 1 KiB; 32 KiB = a 16x32-pixel x 32-channel bf16 tile
    for stride, where in ((8, 'same XCD'), (1, 'other XCD')):
        for scope, name in ((0, 'none (floor)'), (1, 'agent release/acquire fences'), (2, 'workgroup fences + sc1 flag/data loads')):
            for rep in range(2):   # second run: warm
                _lib.check(L.dasr_probe_tile_sync(blocks, stages, stride, scope, words, C.byref(us), C.byref(to), C.byref(stale), None), 'probe')
            print('%-11s %-10s %-38s %10.3f %8d %8d' % ('%d KiB' % (words * 4 // 1024), where, name, us.value, to.value, stale.value))
