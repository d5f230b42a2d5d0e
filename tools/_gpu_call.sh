cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "dcn") 2>&1 | tail -3
python tools/kbench.py --no-conv --reps 20 --off-scale 0.4 > gpurun_out/kbench_dcn_b1_s.txt 2>&1; cut -c1-330 gpurun_out/kbench_dcn_b1_s.txt | tail -9
python tools/kbench.py --no-conv --reps 10 --batch 8 --off-scale 0.4 > gpurun_out/kbench_dcn_b8_s.txt 2>&1; cut -c1-330 gpurun_out/kbench_dcn_b8_s.txt | tail -9
python tools/kbench.py --no-conv --reps 10 --batch 8 --off-scale 1.5 --dcn-layers 64-64 > gpurun_out/kbench_dcn_b8_l.txt 2>&1; cut -c1-330 gpurun_out/kbench_dcn_b8_l.txt | tail -3
