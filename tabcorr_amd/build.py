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
# inst_*.hip hold the kernel instances of the prediction path (one unit per kernel family: they
# compile in parallel, the longest first), paircount.hip the tabulation kernels; launch.hip and
# the .cpp units are host code
SOURCES = ['inst_fused.hip', 'inst_fused32.hip', 'inst_fused16.hip', 'inst_fused40.hip',
           'inst_cross.hip', 'inst_quad.hip', 'inst_single.hip', 'launch.hip', 'paircount.hip',
           'table.cpp', 'interp.cpp', 'comm.cpp', 'runtime.cpp', 'hostmath.cpp']
# per-unit flags (inst_single.hip: see its header)
EXTRA_FLAGS = {'inst_single.hip': ['-ffp-contract=on']}
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fno-gpu-rdc',
         '-Wall', '-Wno-unused-function']


def hipcc():
    for candidate in [os.environ.get('HIPCC'), shutil.which('hipcc'),
                      '/opt/rocm/bin/hipcc']:
        if candidate and os.path.exists(candidate):
            return candidate
    raise RuntimeError('hipcc not found; set HIPCC')


def device_units():
    """The translation units that hold kernels of the prediction path."""
    return [f for f in SOURCES if f.startswith('inst_')]


def kernel_resource_remarks():
    """hipcc's kernel-resource-usage remarks (registers, spills, scratch, LDS per kernel) of
    every device unit of the prediction path, as one text (tools/kernel_resources.py parses
    it).  Compiles the units in parallel into a scratch directory."""
    compiler = hipcc()

    def remarks(unit):
        with tempfile.TemporaryDirectory(prefix='tabcorr_amd_remarks_') as tmp:
            out = subprocess.run(
                [compiler] + FLAGS + EXTRA_FLAGS.get(unit, []) +
                ['-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(CSRC, unit), '-o',
                 os.path.join(tmp, 'unit.o')], capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError('%s does not compile: %s' % (unit, out.stderr[-2000:]))
        return out.stderr

    with ThreadPoolExecutor(max_workers=8) as pool:
        return '\n'.join(pool.map(remarks, device_units()))


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
        commands = [[compiler] + FLAGS + EXTRA_FLAGS.get(f, []) +
                    ['-c', os.path.join(CSRC, f), '-o', o]
                    for f, o in zip(SOURCES, objects)]
        with ThreadPoolExecutor(max_workers=min(8, len(commands))) as pool:
            list(pool.map(run, commands))
        # bind every HIP symbol at load time to the ROCm runtime this library
        # was linked against, even if another copy is loaded later
        run([compiler, '--offload-arch=gfx950', '-fno-gpu-rdc', '-shared',
             '-Wl,-z,now', '-Wl,-rpath,/opt/rocm/lib', '-o', LIBRARY] + objects)
    return LIBRARY


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(LIBRARY)
