#!/bin/bash
# bench.py's kernel pass measured inside the frame (parts of the plan replayed in order with events between them): default line twice,
# standalone coco x 4 / nusc x 4 / headline, the --no-roofline path
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_bl; mkdir -p $O
show() { python -c "
import json,sys
j=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$2', j['value'], j['ms_per_step'], 'dcn', j['roofline']['frac'], j['roofline']['total_ms'], 'conv', j.get('roofline_conv',{}).get('frac'), [ (round(c['roofline']['frac'],3), round(c['roofline_conv']['frac'],3)) for c in j.get('configs',[]) if 'roofline' in c])"; }
for i in 1 2; do S=$(date +%s); python bench.py > $O/default_$i.json 2> $O/default_$i.err; echo "wall $(( $(date +%s) - S )) s"; show $O/default_$i.json default$i; done
for c in "coco_512 4" "nusc_800x448 4" "mot17_512 1"; do set -- $c; python bench.py --config $1 --streams $2 --steps 8 --warmup 3 --no-cpu-baseline --no-extra-configs --no-box-probes > $O/$1.json 2>/dev/null; show $O/$1.json $1; done
python bench.py --steps 5 --no-roofline --no-cpu-baseline --no-extra-configs --no-box-probes 2>/dev/null | tail -1 | cut -c1-200
