"""Build libtabcorr_hip.so (HIP for gfx950) in-tree with hipcc.

``python -m tabcorr_amd.build`` or ``tabcorr_amd.build.build()``.  The shared
library lands next to this file so that it travels with the source tree; the
loader (``tabcorr_amd/_lib.py``) never builds implicitly on a GPU box.
"""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBRARY = os.path.join(HERE, 'libtabcorr_hip.so')
SOURCES = ['capi.hip', 'hostmath.cpp']
HEADERS = ['hostmath.h', 'kernels.hip.h',
           os.path.join('..', '..', 'include', 'tabcorr_amd.h')]


def hipcc():
    for candidate in [os.environ.get('HIPCC'), shutil.which('hipcc'),
                      '/opt/rocm/bin/hipcc']:
        if candidate and os.path.exists(candidate):
            return candidate
    raise RuntimeError('hipcc not found; set HIPCC')


def is_stale():
    if not os.path.exists(LIBRARY):
        return True
    built = os.path.getmtime(LIBRARY)
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(f) > built for f in files)


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources."""
    if not force and not is_stale():
        return LIBRARY
    command = [
        hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC',
        '-shared', '-fgpu-rdc' if False else '-fno-gpu-rdc',
        '-Wall', '-Wno-unused-function',
        # bind every HIP symbol at load time to the ROCm runtime this library
        # was linked against, even if another copy is loaded later
        '-Wl,-z,now', '-Wl,-rpath,/opt/rocm/lib',
        '-o', LIBRARY]
    command += [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(' '.join(command))
    subprocess.run(command, check=True)
    return LIBRARY


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(LIBRARY)
