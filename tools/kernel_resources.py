#!/usr/bin/env python3
"""Per-kernel register / spill / LDS table from hipcc's -Rpass-analysis=kernel-resource-usage
remarks (stdin or a file): one line per kernel, sorted by name.

    python -c 'from tabcorr_amd import build; print(build.kernel_resource_remarks())' | tools/kernel_resources.py
(the remarks of every inst_*.hip unit; or pipe one unit's hipcc ... -Rpass-analysis=kernel-resource-usage output)
"""
import re
import subprocess
import sys


def parse(text):
    kernels = {}
    current = None
    for line in text.splitlines():
        m = re.search(r'remark: .*Function Name: (\S+)', line)
        if m:
            current = m.group(1)
            kernels[current] = {}
            continue
        m = re.search(r'remark: .*?\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/workgroup\])?: (\d+)', line)
        if m and current:
            kernels[current][m.group(1).strip()] = int(m.group(2))
    return kernels


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names),
                         capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
    kernels = parse(text)
    names = demangle(list(kernels))
    print('%-6s %-6s %-6s %-6s %-8s %-8s %s' % ('vgpr', 'agpr', 'sgpr', 'vspill', 'sspill',
                                                 'scratch', 'kernel'))
    for mangled in sorted(kernels, key=lambda k: names[k]):
        k = kernels[mangled]
        print('%-6d %-6d %-6d %-6d %-8d %-8d %s' % (
            k.get('VGPRs', -1), k.get('AGPRs', -1), k.get('TotalSGPRs', k.get('SGPRs', -1)),
            k.get('VGPRs Spill', k.get('VGPR Spill', -1)),
            k.get('SGPRs Spill', k.get('SGPR Spill', -1)),
            k.get('ScratchSize', -1), names[mangled].replace('void ', '')[:150]))


if __name__ == '__main__':
    main()
