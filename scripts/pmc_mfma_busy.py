"""MFMA-pipe utilisation per kernel from two rocprofv3 --pmc passes (scripts/pmc_mfma_busy.sh: DASR_STREAMS=1, bench.py --steps 1 --warmup 1):
   busy % = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines)
The normalisation is checked on the MFMA-only probe kernel of the same run (must read ~100 %).
python scripts/pmc_mfma_busy.py gpurun_out > profiles/<tag>_pmc_mfma_busy.txt"""
import json
import sys

d = sys.argv[1]
busy = json.load(open(d + '/pmc_sq_SQ_VALU_MFMA_BUSY_CYCLES.json'))
sq = json.load(open(d + '/pmc_sq_SQ_BUSY_CYCLES.json'))
mops = json.load(open(d + '/pmc_sq_SQ_INSTS_VALU_MFMA_MOPS_BF16.json'))
print('# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE / --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 (two passes, --kernel-trace only)')
print('# command: DASR_STREAMS=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary   (batch-16 launches, one at a time)')
print('# per launch, summed over the chip: SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_32x32x16_bf16 per SIMD (1024 SIMDs);')
print('# SQ_BUSY_CYCLES counts per shader engine (32 of them).  MFMA busy % = (MFMA_BUSY / 1024) / (SQ_BUSY / 32)')
print('%-52s %9s %16s %14s %10s' % ('kernel', 'launches', 'MFMA busy cyc', 'SQ busy cyc', 'MFMA busy'))
tot_b = tot_s = 0.0
for k, v in busy.items():
    if k not in sq or v['avg_kb'] <= 0:
        continue
    b, s = v['avg_kb'] / 1024.0, sq[k]['avg_kb'] / 32.0
    print('%-52s %9d %16.0f %14.0f %9.1f %%' % (k[:52], v['launches'], b, s, 100.0 * b / s))
    if k.startswith('conv_glds_kernel') or k.startswith('conv_chain') or k in ('wgrad3_ld_kernel<false>', 'wgrad3_kernel<true, false, false, 0>'):   # bf16 trunk kernels (not the f16 HR tail)
        tot_b += b * v['launches']
        tot_s += s * sq[k]['launches']
print('RRDB trunk kernels (conv_chain*_kernel<*>, conv_glds_kernel<*>, wgrad3_ld_kernel<false>), time-weighted: %.1f %% MFMA busy' % (100.0 * tot_b / tot_s))
if len(sys.argv) > 3:   # python scripts/pmc_mfma_busy.py gpurun_out <tag> profiles/pmc_mfma_busy.json : the table bench.py reads
    out = {'source': 'profiles/%s_pmc_mfma_busy.txt: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / --pmc SQ_BUSY_CYCLES (separate passes, --kernel-trace only) over '
                     '`DASR_STREAMS=1 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary`; busy = (MFMA_BUSY/1024 SIMDs)/(SQ_BUSY/32 shader engines); '
                     'the MFMA-only probe kernel of the same run reads ~100 %%' % sys.argv[2],
           'kernels': {k: round((v['avg_kb'] / 1024.0) / (sq[k]['avg_kb'] / 32.0), 4) for k, v in busy.items() if k in sq and v['avg_kb'] > 0},
           'rrdb_trunk_time_weighted': round(tot_b / tot_s, 4)}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
