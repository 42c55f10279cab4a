#!/usr/bin/env python
"""Where a kernel touches scratch memory: counts of scratch loads / stores between the phase marks (s_setprio), calls and the loop header of one function of a
hipcc -S listing.  usage: isa_scratch_map.py <mangled-name-substring>  (reads /tmp/w1k.s; design aid, round 5)"""
import re,sys
key=sys.argv[1]
lines=open('/tmp/w1k.s').read().split('\n')
start=next(i for i,l in enumerate(lines) if l.startswith('_ZN') and key in l.split(':')[0])
end=next(i for i in range(start+1,len(lines)) if lines[i].startswith('.Lfunc_end'))
body=lines[start:end]
out=[];n=0
for i,l in enumerate(body):
    t=l.strip()
    if t.startswith('s_setprio') or 's_swappc' in t or 'Loop Header' in t:
        if n: out.append(f"   [{n} scratch: {kind}]"); n=0
        out.append(f"{i} {t[:60]}")
    if t.startswith('scratch_'):
        n+=1; kind='store' if 'store' in t else 'load'
if n: out.append(f"   [{n} scratch]")
print(len(body)); print('\n'.join(out))
