#!/bin/bash
# round 6, call M: three A/B switches in one lease -- XCD-affine head order (heads_order = 2), write-through (sc1) split-K partial
# stores, the multi-chunk Winograd shape on three waves per SIMD -- on the headline frame and on coco_512 x 4; + FETCH_SIZE of the
# fused heads launch under both orders
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_m; mkdir -p $O
V=$R/centertrack_amd/build/variants
B1="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes"
B4="python bench.py --config coco_512 --streams 4 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-box-probes"
run() { # tag env...
  tag=$1; shift
  env "$@" $B1 > $O/b1_$tag.json 2> $O/b1_$tag.err
  env "$@" $B4 > $O/b4_$tag.json 2> $O/b4_$tag.err
}
for rep in 1 2; do
run base_$rep X=1
run heads2_$rep CENTERTRACK_TUNE=heads_order=2
run wsaux16_$rep CENTERTRACK_LIB=$V/libcentertrack_hip_wsaux16.so
run winom3_$rep CENTERTRACK_LIB=$V/libcentertrack_hip_winom3.so
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_m/b?_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print('%-40s fps %8.1f dev %.4f dcn %.4f conv %.4f' % (f.split('/')[-1], j['value'], j.get('device_ms_per_frame_batch'), j['roofline'].get('total_ms'), j.get('roofline_conv',{}).get('total_ms')))
    except Exception as e: print(f, 'ERR', e)
PY
BENCH2="python $R/bench.py --steps 1 --warmup 1 --frames-per-step 8 --no-cpu-baseline --no-roofline --no-resident --no-extra-configs --no-box-probes"
cd /tmp && export TMPDIR=/tmp
for o in 1 2; do
  rm -rf /tmp/pmcF$o
  CENTERTRACK_TUNE=heads_order=$o timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF$o -o p -- $BENCH2 > /dev/null 2>&1
  python $R/tools/pmc_stats.py $(ls /tmp/pmcF$o/*counter_collection.csv /tmp/pmcF$o/*/*counter_collection.csv 2>/dev/null | head -1) 12 > $O/fetch_heads_order$o.txt 2>&1
  grep "true>" $O/fetch_heads_order$o.txt | cut -c1-140
done
