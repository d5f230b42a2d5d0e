# scratch command file for `gpurun -- 'bash tools/_gpu_call.sh'` (last content: the round's final validation)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -4
python -c "
import __graft_entry__ as g
g.build(); g.smoke()" 2>&1 | tail -1
python bench.py 2>&1 | tail -1
