#!/usr/bin/env python3
"""A/B of library builds in ONE process tree on ONE box: alternates the builds (ABAB...) and
prints ms_per_step of bench.py's headline (or of `--only-config TAG`) for each.

    python tools/ab_bench.py [--rounds 3] [--args "..."] name=path[,OPTION=VALUE...] ...

`path` = a libtabcorr_hip*.so ('' or 'tree' = the in-tree build); options are passed as
`--option name=value`."""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--rounds', type=int, default=3)
    parser.add_argument('--args', default='--detail 0 --cpu-seconds 0')
    parser.add_argument('builds', nargs='+')
    args = parser.parse_args()
    results = {}
    for _ in range(args.rounds):
        for build in args.builds:
            name, _, rest = build.partition('=')
            parts = rest.split(',') if rest else ['tree']
            path, options = parts[0], parts[1:]
            env = dict(os.environ)
            if path not in ('', 'tree'):
                env['TABCORR_AMD_LIBRARY'] = os.path.join(REPO, path)
            command = [sys.executable, os.path.join(REPO, 'bench.py')] + args.args.split()
            for option in options:
                command += ['--option', option]
            out = subprocess.run(command, env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(name, 'FAILED', out.stderr[-500:], flush=True)
                continue
            record = json.loads(out.stdout.strip().splitlines()[-1])
            if 'ms_per_step' in record:
                value = record['ms_per_step'] * 1e3
            else:       # --only-config: {name: record}
                value = list(record.values())[0]['us_per_step']
            results.setdefault(name, []).append(value)
            print('%-12s %.2f us' % (name, value), flush=True)
    for name, values in results.items():
        print('%-12s mean %.2f  min %.2f  (%s)' % (name, sum(values) / len(values), min(values),
                                                    ' '.join('%.2f' % v for v in values)))


if __name__ == '__main__':
    main()
