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
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'dasr_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, trace=False):
    """trace=True: instrumented copy (libdasr_hip_trace.so, -DDASR_TRACE: per-workgroup s_memtime stamps in conv_kernel) for
    scripts/micro_conv.py; never loaded by the product path unless DASR_HIP_LIB points at it."""
    global LIB
    if trace:
        LIB = os.path.join(HERE, 'libdasr_hip_trace.so')
    if not force and not trace and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, 'build', src.replace('.hip', '_trace.o' if trace else '.o'))
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', path, '-o', obj] + (['-DDASR_TRACE'] if trace else [])
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if verbose or p.returncode != 0:
            sys.stderr.write(out.decode())
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, trace='--trace' in sys.argv))
