#!/bin/bash
# division-free tile decode in the DCN MAIN / OFFSETS prologue (host-made magic multipliers, variant build -DCT_FASTDIV): parity of the DCN op
# tests with the variant, per-layer times and the DCN sequence of the headline and of coco x 4, base and variant alternating in one lease
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_bi; mkdir -p $O
V=$R/centertrack_amd/build/variants/libcentertrack_hip_fd.so
CENTERTRACK_LIB=$V timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -q -k "dcn or forward" 2>&1 | tail -2
for rep in 1 2; do for lib in base fd; do
L=""; [ $lib = fd ] && L=$V
for c in "mot17_512 1" "coco_512 4" "nusc_800x448 4"; do set -- $c
CENTERTRACK_LIB=$L python bench.py --config $1 --streams $2 --steps 8 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-4s %-14s x%d  fps %7.1f  device_ms %.4f  dcn_ms %.4f (frac %.4f)' % ('$lib', '$1', $2, j['value'], j.get('device_ms_per_frame_batch',0), j['roofline']['total_ms'], j['roofline']['frac']))" | tee -a $O/fastdiv.txt
done; done; done
