cd $GRAFT_REPO_ROOT
for nt in "" "--dcn-nt"; do
python tools/kbench.py --no-conv --reps 10 --batch 8 --off-scale 0.4 $nt 2>&1 | cut -c1-250 | tail -9
done
python tools/kbench.py --no-conv --reps 20 --off-scale 0.4 --dcn-nt 2>&1 | cut -c1-250 | tail -9
