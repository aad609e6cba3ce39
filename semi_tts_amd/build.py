"""Builds libsemitts_hip.so (gfx950 only) in-tree with hipcc.  `python -m semi_tts_amd.build`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libsemitts_hip.so')
SOURCES = ['skinny.hip', 'skinny_packed.hip', 'attention.hip', 'gemm.hip', 'rnn.hip', 'vq.hip', 'runtime.hip', 'decoder.hip', 'grad.hip', 'attention_bwd.hip', 'decoder_bwd.hip', 'loss.hip', 'optim.hip', 'norm.hip']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), 'include', 'semitts.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -> semi_tts_amd/lib/libsemitts_hip.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace('.hip', '.o'))
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-mllvm', '-amdgpu-kernarg-preload-count=16', '-c', os.path.join(CSRC, src), '-o', obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out))
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout)
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
