"""Instruction mix of one kernel in a hipcc -S listing: python tools/diag/isa_mix.py file.s <mangled-name-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(('EOF', ':')) or (l.startswith('_Z') and key in l and ': ' in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
c = collections.Counter()
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l[0] in '.;' or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
valu = sum(v for k, v in c.items() if k.startswith('v_'))
trans = sum(v for k, v in c.items() if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)', k))
pk = sum(v for k, v in c.items() if k.startswith('v_pk_'))
print(f'lines {start}-{end}  VALU {valu}  transcendental {trans}  packed {pk}  total {sum(c.values())}')
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f'  {k:28s}{v}')
