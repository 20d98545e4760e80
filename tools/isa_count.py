#!/usr/bin/env python3
"""Static instruction counts of one kernel in a hipcc -S listing (development aid): isa_count.py file.s mangled-name-substring"""
import sys
from collections import Counter
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(':')[0] for l in s.split('\n') if l.startswith('_Z') and ':' in l and key in l.split(':')[0]]
for name in names:
    a = s.index('\n' + name + ':')
    b = s.index('.Lfunc_end', a)
    lines = [l.strip() for l in s[a:b].split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.strip().endswith(':')]
    c = Counter(l.split()[0] for l in lines)
    print(name[:70], 'total', len(lines), 'valu', sum(v for k, v in c.items() if k.startswith('v_')), 'salu', sum(v for k, v in c.items() if k.startswith('s_')),
          'writelane', c['v_writelane_b32'], 'readlane', c['v_readlane_b32'], 'ds', sum(v for k, v in c.items() if k.startswith('ds_')))
