"""Copy the artefacts of one `scripts/gpu_round.sh` session from gpurun_out/ into profiles/ under a round tag, and rebuild
profiles/pmc_traffic.json (read by bench.py for roofline.traffic) from the two PMC pass summaries.
python scripts/collect_profiles.py r02c"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')
line = [l for l in open(os.path.join(G, 'bench.log')) if l.startswith('{"metric"')][-1]
open(os.path.join(P, tag + '_bench.json'), 'w').write(line)
shutil.copy(os.path.join(G, 'prof', tag + '_kernel_stats.csv'), os.path.join(P, tag + '_kernel_stats.csv'))
shutil.copy(os.path.join(G, 'parity_margins.log'), os.path.join(P, tag[:3] + '_parity_margins.log'))
txt = open(os.path.join(G, 'pmc_FETCH_SIZE_summary.txt')).read() + open(os.path.join(G, 'pmc_WRITE_SIZE_summary.txt')).read()
open(os.path.join(P, tag + '_pmc_traffic.txt'), 'w').write(txt)
f, w = (json.load(open(os.path.join(G, 'pmc_%s_summary.json' % c))) for c in ('FETCH_SIZE', 'WRITE_SIZE'))
out = {'source': 'profiles/%s_pmc_traffic.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over '
                 '`bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary`; per-launch averages in KB; FETCH_SIZE x2 on gfx950 '
                 '(MI355X_MICROARCH.md, HBM section)' % tag,
       'fetch_correction_gfx950': 2.0, 'kernels': {}}
for k, v in f.items():
    if k in w:
        out['kernels'][k] = {'fetch_size_kb_raw': round(v['avg_kb'], 1), 'write_size_kb': round(w[k]['avg_kb'], 1), 'launches': v['launches']}
# secondary workloads (configs[2] with VGG19-54 / LPIPS, configs[4] with VGG16 / LPIPS): kernel stats of scripts/prof_secondary.sh and the PMC passes
# of scripts/r04/call3.sh (gpurun_out/pmcs_<workload>_<COUNTER>_summary.json), keyed like bench.py's `secondary` entries
out['workloads'] = {}
for wl in ('dasr_vgg', 'dasr_lpips', 'dsn_vgg', 'dsn_lpips'):
    st = os.path.join(G, 'prof_sec', wl + '_kernel_stats.csv')
    if os.path.exists(st):
        shutil.copy(st, os.path.join(P, '%s_%s_kernel_stats.csv' % (tag, wl)))
    fs, wsz = (os.path.join(G, 'pmcs_%s_%s_summary.json' % (wl, c)) for c in ('FETCH_SIZE', 'WRITE_SIZE'))
    if os.path.exists(fs) and os.path.exists(wsz):
        f2, w2 = json.load(open(fs)), json.load(open(wsz))
        out['workloads'][wl] = {k: {'fetch_size_kb_raw': round(v['avg_kb'], 1), 'write_size_kb': round(w2[k]['avg_kb'], 1), 'launches': v['launches']}
                                for k, v in f2.items() if k in w2}
        txt2 = open(fs.replace('.json', '.txt')).read() + open(wsz.replace('.json', '.txt')).read()
        open(os.path.join(P, '%s_%s_pmc_traffic.txt' % (tag, wl)), 'w').write(txt2)
json.dump(out, open(os.path.join(P, 'pmc_traffic.json'), 'w'), indent=1)
print('collected', tag, len(out['kernels']), 'kernels;', {k: len(v) for k, v in out['workloads'].items()})
