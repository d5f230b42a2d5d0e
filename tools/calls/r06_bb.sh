#!/bin/bash
# capture guard: the regression test (guard on / off), then the GPU suite twice with it and twice without (CT_NO_CAPTURE_GUARD=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_bb; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_e2e.py -q -k "collection_of_an_old" 2>&1 | tail -6 | cut -c1-250
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -q > $O/suite_guard_$i.log 2>&1; echo "guard: $(tail -1 $O/suite_guard_$i.log)"; grep -E "^(FAILED|ERROR)" $O/suite_guard_$i.log | cut -c1-160; done
for i in 1 2; do CT_NO_CAPTURE_GUARD=1 timeout 1500 python -m pytest tests -m gpu -q --deselect "tests/test_hip_e2e.py::test_graph_capture_and_the_collection_of_an_old_detector" > $O/suite_noguard_$i.log 2>&1; echo "no guard: $(tail -1 $O/suite_noguard_$i.log)"; grep -E "^(FAILED|ERROR)" $O/suite_noguard_$i.log | cut -c1-120 | head -4; done
