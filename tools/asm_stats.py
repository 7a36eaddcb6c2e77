"""Static per-kernel statistics of a gfx950 assembly listing (hipcc -save-temps): python tools/asm_stats.py file.s [name-substring]..."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
want = sys.argv[2:]
meta = {}
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\d+)', m.group(2)) or [0, '?'])[1]
    meta[m.group(1)] = dict(vgpr=g('next_free_vgpr'), accum=g('accum_offset'), lds=g('group_segment_fixed_size'), scratch=g('private_segment_fixed_size'))
for name, md in meta.items():
    if want and not any(w in name for w in want): continue
    i = s.index('\n' + name + ':'); j = s.index('s_endpgm', i)
    ins = [l.strip() for l in s[i:j].split('\n')]
    ins = [l for l in ins if l and not l.startswith((';', '.')) and not l.endswith(':')]
    c = Counter(l.split()[0] for l in ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    print(f"{name}: vgpr {md['vgpr']} (accum_offset {md['accum']}) lds {md['lds']} scratch {md['scratch']}  static: total {len(ins)} valu {valu} mfma {sum(v for k, v in c.items() if k.startswith('v_mfma'))} "
          f"mad64 {c.get('v_mad_u64_u32', 0) + c.get('v_mad_i64_i32', 0)} ds {sum(v for k, v in c.items() if k.startswith('ds_'))} scratch_ops {sum(v for k, v in c.items() if k.startswith('scratch_'))} s_nop {c.get('s_nop', 0)}")
