"""Build libtabcorr_hip.so (HIP for gfx950) in-tree with hipcc.

``python -m tabcorr_amd.build`` or ``tabcorr_amd.build.build()``.  The shared
library lands next to this file so that it travels with the source tree; the
loader (``tabcorr_amd/_lib.py``) never builds implicitly on a GPU box.
"""

import glob
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBRARY = os.path.join(HERE, 'libtabcorr_hip.so')
# launch.hip (prediction) and paircount.hip (tabulation) hold the device code; the other
# units are host-only C++
SOURCES = ['launch.hip', 'paircount.hip', 'table.cpp', 'interp.cpp', 'comm.cpp',
           'runtime.cpp', 'hostmath.cpp']
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fno-gpu-rdc',
         '-Wall', '-Wno-unused-function']


def hipcc():
    for candidate in [os.environ.get('HIPCC'), shutil.which('hipcc'),
                      '/opt/rocm/bin/hipcc']:
        if candidate and os.path.exists(candidate):
            return candidate
    raise RuntimeError('hipcc not found; set HIPCC')


def dependencies():
    files = [os.path.join(CSRC, f) for f in SOURCES]
    files += glob.glob(os.path.join(CSRC, '*.h'))
    files += glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    return files


def is_stale():
    if not os.path.exists(LIBRARY):
        return True
    built = os.path.getmtime(LIBRARY)
    return any(os.path.getmtime(f) > built for f in dependencies())


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources."""
    if not force and not is_stale():
        return LIBRARY
    compiler = hipcc()

    def run(command):
        if verbose:
            print(' '.join(command), flush=True)
        subprocess.run(command, check=True)

    with tempfile.TemporaryDirectory(prefix='tabcorr_amd_build_') as tmp:
        objects = [os.path.join(tmp, os.path.splitext(f)[0] + '.o')
                   for f in SOURCES]
        commands = [[compiler] + FLAGS + ['-c', os.path.join(CSRC, f), '-o', o]
                    for f, o in zip(SOURCES, objects)]
        with ThreadPoolExecutor(max_workers=min(4, len(commands))) as pool:
            list(pool.map(run, commands))
        # bind every HIP symbol at load time to the ROCm runtime this library
        # was linked against, even if another copy is loaded later
        run([compiler, '--offload-arch=gfx950', '-fno-gpu-rdc', '-shared',
             '-Wl,-z,now', '-Wl,-rpath,/opt/rocm/lib', '-o', LIBRARY] + objects)
    return LIBRARY


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(LIBRARY)
