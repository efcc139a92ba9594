"""Build libdasr_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a
plain C-ABI shared object (include/dasr_hip.h) loaded through ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdasr_hip.so')
SOURCES = ['conv.hip', 'wgrad.hip', 'misc.hip', 'gan.hip', 'lpips.hip', 'rccl.hip']


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f != 'bench_probes.hip'] + [os.path.join(HERE, '..', 'include', 'dasr_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


BENCH_LIB = os.path.join(HERE, 'libdasr_bench.so')


def build_bench(force=False, verbose=False):
    """libdasr_bench.so: the micro-benchmark probes (csrc/bench_probes.hip, include/dasr_hip_bench.h).  Separate from the product library on
    purpose: bench.py / scripts/micro_*.py load it next to libdasr_hip.so; nothing on the product path does."""
    src = os.path.join(CSRC, 'bench_probes.hip')
    deps = [src, os.path.join(HERE, '..', 'include', 'dasr_hip_bench.h')]
    if not force and os.path.exists(BENCH_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(BENCH_LIB) for d in deps):
        return BENCH_LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', src, '-o', BENCH_LIB]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if verbose or p.returncode != 0:
        sys.stderr.write(p.stdout.decode())
    if p.returncode != 0:
        raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    return BENCH_LIB


def build(force=False, verbose=False, trace=False, ablate=False):
    """trace=True: instrumented copy (libdasr_hip_trace.so, -DDASR_TRACE: per-workgroup s_memtime stamps in conv_kernel) for
    scripts/micro_conv.py; ablate=True: libdasr_hip_ablate.so (-DDASR_BENCH: the wrong-result ablation instantiations of the dense conv /
    weight-gradient kernels, timing experiments only).  Neither is loaded by the product path unless DASR_HIP_LIB points at it."""
    global LIB
    if trace:
        LIB = os.path.join(HERE, 'libdasr_hip_trace.so')
    if ablate:
        LIB = os.path.join(HERE, 'libdasr_hip_ablate.so')
    if not force and not trace and not ablate and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, 'build', src.replace('.hip', '_trace.o' if trace else ('_ablate.o' if ablate else '.o')))
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', path, '-o', obj] + (['-DDASR_TRACE'] if trace else []) + (['-DDASR_BENCH'] if ablate else [])
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if verbose or p.returncode != 0:
            sys.stderr.write(out.decode())
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl', '-pthread']
    subprocess.check_call(cmd)
    out, LIB = LIB, os.path.join(HERE, 'libdasr_hip.so')
    return out


if __name__ == '__main__':
    if '--bench' in sys.argv:
        print(build_bench(force='--force' in sys.argv, verbose=True))
    else:
        print(build(force='--force' in sys.argv, verbose=True, trace='--trace' in sys.argv, ablate='--ablate' in sys.argv))
