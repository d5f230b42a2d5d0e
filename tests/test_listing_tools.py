"""CPU: the two listing tools that guard blind kernel work -- tools/isa_diff.py (did a change leave the shipped kernels'
instruction streams alone?) and tools/occupancy.py (registers / LDS / waves per SIMD from the code-object metadata)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
if ROOT not in sys.path:
    sys.path.insert(1, ROOT)

LISTING = '''
\t.type\t_Z1aPf,@function
_Z1aPf:                                 ; @_Z1aPf
; %%bb.0:
\ts_load_dwordx2 s[0:1], s[0:1], 0x0
.LBB0_%d:
\tv_add_f32_e32 v0, %s, v0               ; a comment
\ts_cbranch_scc1 .LBB0_%d
\ts_endpgm
.Lfunc_end0:
\t.size\t_Z1aPf, .Lfunc_end0-_Z1aPf
\t.type\t_Z1bPf,@function
_Z1bPf:                                 ; @_Z1bPf
\ts_endpgm
.Lfunc_end1:
amdhsa.kernels:
  - .agpr_count:     0
    .group_segment_fixed_size: 4096
    .name:           _Z1aPf
    .private_segment_fixed_size: 0
    .vgpr_count:     %d
    .vgpr_spill_count: 0
  - .agpr_count:     16
    .group_segment_fixed_size: 0
    .name:           _Z1bPf
    .private_segment_fixed_size: 24
    .vgpr_count:     200
    .vgpr_spill_count: 6
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
'''


def test_isa_diff_ignores_label_numbers_and_comments_but_not_instructions(tmp_path, capsys):
    import isa_diff
    a, b, c = tmp_path / 'a.s', tmp_path / 'b.s', tmp_path / 'c.s'
    a.write_text(LISTING % (3, 'v1', 3, 148))
    b.write_text(LISTING % (7, 'v1', 7, 148))           # labels renumbered only
    c.write_text(LISTING % (3, 'v2', 3, 148))           # another operand
    assert isa_diff.main(str(a), str(b)) == 0
    assert isa_diff.main(str(a), str(c)) == 1
    out = capsys.readouterr().out
    assert 'CHANGED _Z1aPf' in out and 'same    _Z1bPf' in out


def test_occupancy_reads_registers_spills_and_lds(tmp_path):
    import occupancy
    p = tmp_path / 'k.s'
    p.write_text(LISTING % (1, 'v1', 1, 148))
    rows = {r['name']: r for r in occupancy.kernels(str(p))}
    assert rows['_Z1aPf'] == dict(name='_Z1aPf', vgpr=148, spill=0, lds=4096, scratch=0)
    assert rows['_Z1bPf']['spill'] == 6 and rows['_Z1bPf']['scratch'] == 24 and rows['_Z1bPf']['vgpr'] == 200


def test_rounds_residency_by_registers_lds_and_workgroup_size():
    import rounds
    assert rounds.per_cu(148, 0, 29696, 256)[0] == 3          # the fused DCN kernel: registers
    assert rounds.per_cu(112, 16, 29696, 256)[0] == 4         # 128 in total (unified file): 4 per CU
    assert rounds.per_cu(124, 0, 40960, 256)[0] == 4          # capped stem: registers and LDS both allow 4
    assert rounds.per_cu(52, 0, 81920, 256)[0] == 2           # LDS-bound
    assert rounds.per_cu(164, 0, 0, 512)[0] == 1              # 8-wave workgroups at 3 waves per SIMD: one fits
    assert rounds.per_cu(96, 0, 0, 1024)[0] == 1              # 16-wave workgroups: 5 waves per SIMD, 4 needed per workgroup


def test_bench_flags_used_by_the_collection_scripts_exist():
    """tools/collect_profiles.sh / sweep_configs.sh / calls/*.sh drive bench.py with flags; a renamed flag would only show
    on the GPU box, minutes into a paid call.  Every `--flag` those scripts pass must parse, and the configurations the
    default line appends must exist."""
    import glob
    import importlib
    import re
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    import scenarios as S
    assert all(name in S.CONFIGS and streams >= 1 and steps >= 0 for name, streams, steps in bench.EXTRA_CONFIGS)
    flags = set()
    for path in glob.glob(os.path.join(root, 'tools', '*.sh')) + glob.glob(os.path.join(root, 'tools', 'calls', '*.sh')):
        text = open(path).read()
        uses_b = re.search(r'^\s*B="python (\$R/)?bench\.py', text, re.M) is not None
        for line in text.splitlines():
            direct = re.search(r'(?<![a-z])bench\.py', line) is not None
            via_b = uses_b and re.match(r'\s*(CENTERTRACK_\w+=\S+ )*\$B ', line) is not None
            if direct or via_b:
                flags.update(re.findall(r'(--[a-z][a-z0-9-]+)', line))
    flags -= {'--kernel-trace', '--stats', '--pmc', '--output-format', '--timeout'}
    known = {a.option_strings[0] for a in bench.build_parser()._actions if a.option_strings}
    missing = sorted(f for f in flags if f not in known)
    assert not missing, 'bench.py does not know %s' % missing


def test_box_state_classification_from_the_instruction_fetch_probe():
    """bench.py's ``box_calibration.state``: the recorded leases fall into their classes, anything else is not forced into one
    (profiles/r05_a_box_probes.jsonl calls 39-48: nine fast leases 38.8-40.1 us, the slow one 56.5 us)"""
    import json
    from tools import box_calib
    rows = [json.loads(l) for l in open(os.path.join(ROOT, 'profiles', 'r05_a_box_probes.jsonl'))]
    seen = {'fast': 0, 'slow': 0}
    for r in rows:
        if 'ifetch_64KB_code_256wg' not in (r.get('launch_us') or {}):
            assert box_calib.fetch_state(r.get('launch_us'))['instruction_fetch'] == 'unknown'
            continue
        st = box_calib.fetch_state(r['launch_us'])
        want = 'slow' if r['device_ms'] > 1.03 else 'fast'
        assert st['instruction_fetch'] == want, (r['call'], st)
        lo, hi = st['expect_device_ms_mot17_512']
        assert lo <= r['device_ms'] <= hi, (r['call'], r['device_ms'], st)
        seen[want] += 1
    assert seen['fast'] >= 9 and seen['slow'] >= 1
    assert box_calib.fetch_state({'ifetch_64KB_code_256wg': 47.0})['instruction_fetch'] == 'between'
    assert box_calib.fetch_state({'error': 'x'})['instruction_fetch'] == 'unknown'
