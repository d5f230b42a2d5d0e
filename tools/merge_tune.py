#!/usr/bin/env python
"""Add the keys of a tuning cache file (CENTERTRACK_TUNE_CACHE of a GPU run) to the pinned table
centertrack_amd/tune_table.json.  Existing keys are kept unless --overwrite (a pinned choice fixes fp32 summation orders).
    python tools/merge_tune.py gpurun_out/tune_new.json [--overwrite] [--only PREFIX]"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
TABLE = os.path.join(ROOT, 'centertrack_amd', 'tune_table.json')


def main(argv):
    path = argv[0]
    overwrite = '--overwrite' in argv
    only = argv[argv.index('--only') + 1] if '--only' in argv else ''
    with open(TABLE) as f:
        table = json.load(f)
    with open(path) as f:
        new = json.load(f)
    added = changed = 0
    for k, v in sorted(new.items()):
        if only and not k.startswith(only):
            continue
        if k not in table:
            table[k] = v
            added += 1
        elif overwrite and table[k] != v:
            table[k] = v
            changed += 1
    with open(TABLE, 'w') as f:
        json.dump({k: table[k] for k in sorted(table)}, f, indent=0)
    print('%d keys added, %d changed, %d total' % (added, changed, len(table)))


if __name__ == '__main__':
    main(sys.argv[1:])
