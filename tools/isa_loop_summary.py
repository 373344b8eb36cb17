#!/usr/bin/env python3
"""Loops of a kernel in a hipcc --save-temps .s file, one line each: L = vector-memory loads, m = MFMAs, wN = s_waitcnt vmcnt(N), |B| = s_barrier.
   python tools/isa_loop_summary.py FILE.s 'kernel name regex (demangled)'
Back edges: conditional AND unconditional branches to an earlier label (hipcc ends a rotated loop with `s_cbranch exit; s_branch header`)."""
import re,subprocess,sys
txt=open(sys.argv[1]).read(); pat=sys.argv[2]
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    name=subprocess.run(["c++filt", m.group(1)],capture_output=True,text=True).stdout.strip().split("(")[0].replace("void ","")
    if not re.search(pat,name): continue
    L=[l.strip() for l in m.group(2).split("\n")]
    labels={t.split(":")[0]:i for i,t in enumerate(L) if re.match(r"\.LBB\d+_\d+:",t)}
    seen=set()
    for i,t in enumerate(L):
        b=re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)",t)
        if b and b.group(1) in labels and labels[b.group(1)]<i:
            h=labels[b.group(1)]; body=L[h:i]
            mf=sum(1 for q in body if q.startswith('v_mfma'))
            if mf<20 or h in seen: continue
            seen.add(h)
            seq=[]
            for q in body:
                w=re.search(r"vmcnt\((\d+)\)",q)
                if w: seq.append('w%s'%w.group(1))
                elif re.match(r"(buffer|global)_load",q): seq.append('L')
                elif q.startswith('s_barrier'): seq.append('|B|')
                elif q.startswith('v_mfma'): seq.append('m')
            # compress
            out=[];
            for x in seq:
                if out and out[-1][0]==x and x in ('L','m'): out[-1][1]+=1
                else: out.append([x,1])
            print(name[:46],'loop@%d'%h, ' '.join((x if n==1 else '%s%d'%(x,n)) for x,n in out)[:600])
