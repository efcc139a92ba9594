"""What the store path sustains when every workgroup's epilogue stores at once (dasr_probe_store, include/dasr_hip_bench.h): bytes per clock per CU for 8 / 64 / 256
workgroups (1 / 8 / 32 per XCD), 32 KiB and 192 KiB bursts (the Cout-32 and conv5 epilogues of rdb_is_kernel), plain and sc1 stores.   python scripts/micro_store.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dasr_amd import _lib, engine
engine.ensure_runtime_ready()
L = _lib.bench_lib()
buf = torch.zeros(256 * 192 * 1024 * 41, dtype=torch.uint8, device='cuda')   # 2 GB: 40 regions of 48 MB for the roaming modes
cyc = torch.zeros(256, dtype=torch.int64, device='cuda')
print('%-8s %-6s %-20s %10s %12s' % ('blocks', 'KiB', 'store', 'cycles', 'B/clk/CU'))
for kb in (32, 192):
    for blocks in (8, 64, 128, 256):
        for mode, name in ((0, 'plain'), (1, 'sc1')) + (((2, 'plain/tile pattern'), (3, 'sc1/tile pattern'), (4, 'plain/roaming'), (5, 'sc1/roaming'), (7, 'sc1/tile/roaming')) if kb == 192 else ()):
            for rep in range(2):
                _lib.check(L.dasr_probe_store(buf.data_ptr(), blocks, kb, mode, 40, 200, cyc.data_ptr(), None), 'probe')
            c = cyc[:blocks].double().mean().item() / 40
            print('%-8d %-6d %-20s %10.0f %12.2f' % (blocks, kb, name, c, kb * 1024 / c))
