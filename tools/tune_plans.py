#!/usr/bin/env python
"""Build the launch plans of a list of (config, images per step) so that every launch shape they need is timed once and
lands in the cache file named by CENTERTRACK_TUNE_CACHE (keys the pinned table already holds are not re-timed);
`tools/merge_tune.py` then adds the new keys to centertrack_amd/tune_table.json.
    CENTERTRACK_TUNE_CACHE=gpurun_out/tune_new.json python tools/tune_plans.py [name:N ...]
default list: every plan the tests, bench.py and tools/sweep_configs.sh build (N = streams x 2 under flip_test)."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)

DEFAULT = ['mot17_512:1', 'mot17_512:8', 'mot17_512:16', 'mot17_512:32', 'nusc_800x448:1', 'nusc_800x448:4', 'nusc_800x448:8',
           'nusc_800x448:16', 'nusc_800x448:32', 'kitti_1280x384:8', 'kitti_1280x384:4', 'coco_512:4', 'coco_512:2',
           'mot17_544x960:1', 'mot17_544x960:8']


def main(items):
    assert os.environ.get('CENTERTRACK_TUNE_CACHE'), 'set CENTERTRACK_TUNE_CACHE to the file the new keys go to'
    import scenarios as S
    from centertrack_amd import autotune
    from centertrack_amd.model import DLASegHIP
    for it in items:
        name, n = it.split(':')
        cfg = S.CONFIGS[name]
        before = set(autotune._CACHE)
        t0 = time.time()
        model = DLASegHIP(S.HEAD_SETS[cfg['heads']]).to('cuda')
        model.get_plan(int(n), cfg['H'], cfg['W'], True, True, True)
        new = sorted(set(autotune._CACHE) - before - set(autotune._PINNED))
        print('%-18s N=%-3s %5.1f s, %d new keys %s' % (name, n, time.time() - t0, len(new), new[:4]))
        del model


if __name__ == '__main__':
    main(sys.argv[1:] or DEFAULT)
