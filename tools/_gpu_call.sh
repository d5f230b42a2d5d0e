cd $GRAFT_REPO_ROOT
D=$PWD/centertrack_amd/build/dbg
for dv in 32x64/1 4x32x64/1 F32x64/1; do
for b in 1 8; do
for v in s0 s3 final; do
L=$D/libct_$v.so; [ $v = final ] && L=$PWD/centertrack_amd/libcentertrack_hip.so
echo "== $dv B=$b $v"; CENTERTRACK_LIB=$L python tools/kbench.py --no-conv --batch $b --dvariant $dv 2>&1 | grep "^dcn" | cut -c1-60 | tr '\n' ';' ; echo
done; done; done
python tools/dcn_slots.py 2>&1 | tail -17
