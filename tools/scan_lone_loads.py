#!/usr/bin/env python3
"""Scan the gfx950 ISA of every kernel of semi_tts_amd/csrc for loads the compiler serialised: a global / buffer load that is
followed -- with no other load in between -- by a full `s_waitcnt vmcnt(0)` within a few instructions.  That is what a select on a
loaded value (`cond ? p[i] : 0`, `if (!ok) v = 0`, `ok = ok && v != X`) compiles to: a branch around the load and a wait behind it;
N "independent" loads then cost N serial round trips.  Runs here (hipcc cross-compiles without a GPU):

    python tools/scan_lone_loads.py [min_count]

prints per kernel `lone loads / all loads`.  Epilogues and cold fall-back loops show up too: read the assembly before acting."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm', '-amdgpu-kernarg-preload-count=16', '-S', '--cuda-device-only']
LOAD = re.compile(r'\b(global_load|buffer_load|flat_load)')


def main():
    min_count = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    out = tempfile.mkdtemp(prefix='st_asm_')
    procs = []
    for src in sorted(glob.glob(os.path.join(ROOT, 'semi_tts_amd', 'csrc', '*.hip'))):
        dst = os.path.join(out, os.path.basename(src)[:-4] + '.s')
        procs.append((dst, subprocess.Popen(['hipcc'] + FLAGS + ['-o', dst, src], stderr=subprocess.DEVNULL)))
    for dst, p in procs:
        p.wait()
        lines = open(dst).read().split('\n')
        kern, lone, total = None, {}, {}
        for i, l in enumerate(lines):
            m = re.match(r'^(_Z\w+):', l)
            if m:
                kern = m.group(1)
            if kern and LOAD.search(l) and 'lds' not in l:
                total[kern] = total.get(kern, 0) + 1
                n_load, hit = 1, False
                for x in lines[i + 1:i + 14]:
                    x = x.strip()
                    if LOAD.search(x):
                        n_load += 1
                    if x.startswith('s_waitcnt vmcnt(0)'):
                        hit = True
                        break
                if hit and n_load == 1:
                    lone[kern] = lone.get(kern, 0) + 1
        for k, v in lone.items():
            if v >= min_count:
                name = subprocess.run(['c++filt', k], stdout=subprocess.PIPE, text=True).stdout.strip().replace('(anonymous namespace)::', '')
                print('%-18s %3d / %3d  %s' % (os.path.basename(dst), v, total[k], name[:120]))


if __name__ == '__main__':
    main()
