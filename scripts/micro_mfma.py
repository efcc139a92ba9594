"""MFMA-only micro-benchmark at several kernel lengths: python scripts/micro_mfma.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dasr_amd import _lib
L = _lib.bench_lib()   # libdasr_bench.so (python -m dasr_amd.build --bench)
torch.zeros(1, device='cuda')
pk = C.c_float(0)
for it in (9, 23, 90, 360, 3600, 20000, 90, 23, 9):
    r = []
    for _ in range(5):
        L.dasr_probe_mfma_peak(it, C.byref(pk), None)
        r.append(pk.value)
    us = [512 * 4 * it * 4 * 2.0 * 32 * 32 * 16 / (x * 1e12) * 1e6 for x in r]
    print('iters %6d (x4 MFMA per wave): TFLOP/s %s | us %s' % (it, ' '.join('%.0f' % x for x in r), ' '.join('%.1f' % x for x in us)))
# operand data dependence of the sustained rate (dasr_probe_mfma_data): random bf16 / random f16 / all-zero operands, same instruction stream
for mode, name in ((2, 'bf16 zeros '), (0, 'bf16 random'), (1, 'f16  random'), (2, 'bf16 zeros '), (0, 'bf16 random'), (1, 'f16  random')):
    r = []
    for _ in range(4):
        _lib.check(L.dasr_probe_mfma_data(19968, mode, C.byref(pk), None), 'probe')
        r.append(pk.value)
    print('MFMA-only, %s operands: TFLOP/s %s' % (name, ' '.join('%.0f' % x for x in r)))
