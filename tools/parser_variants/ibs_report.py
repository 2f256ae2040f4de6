#!/usr/bin/env python3
"""ibs_report.py <harness binary> <PV_SAMPLE=ibs-op raw file> [N]: per instruction of the binary, from AMD IBS op samples:
summed dispatch -> completion latency, data-cache misses and their latency, mispredictions (top N of each)."""
import re, collections, bisect, subprocess, sys
BIN=sys.argv[1]; RAW=sys.argv[2]; N=int(sys.argv[3]) if len(sys.argv)>3 else 25
asm=subprocess.run(['objdump','-d','--no-show-raw-insn',BIN],capture_output=True,text=True).stdout
ins=[]; func=None
for l in asm.splitlines():
    m=re.match(r'^([0-9a-f]+) <(.*)>:',l)
    if m: func=m.group(2); continue
    m=re.match(r'^\s+([0-9a-f]+):\s+(.*)',l)
    if m: ins.append((int(m.group(1),16), m.group(2).strip(), func))
addrs=[a for a,_,_ in ins]
rows={}
for l in open(RAW):
    f=l.split()
    if len(f)<7: continue
    rows[int(f[0],16)]=[int(x) for x in f[1:7]]
tot=[sum(r[i] for r in rows.values()) for i in range(6)]
n,tag,comp,dcm,dcl,brm=tot
print("samples %d  avg tag->ret %.1f  avg comp->ret %.1f  avg exec (tag->comp) %.1f   dc-miss ops %.2f%% avg miss lat %.0f  mispredicted ops %.3f%%"%(n,tag/n,comp/n,(tag-comp)/n,100*dcm/n,dcl/max(dcm,1),100*brm/n))
# exec latency = tag_ret - comp_ret: per instruction total
ex={a:r[1]-r[2] for a,r in rows.items()}
tex=sum(ex.values())
print("--- top instructions by summed execution latency (dispatch->completion), share of all")
for a,v in sorted(ex.items(),key=lambda kv:-kv[1])[:N]:
    i=bisect.bisect_left(addrs,a); r=rows[a]
    print("%5.2f%%  n=%5d avgexec=%6.1f dcmiss=%4d avglat=%5.0f misp=%4d  %x: %-60s"%(100*v/tex,r[0],v/r[0],r[3],r[4]/max(r[3],1),r[5],a,ins[i][1][:60]))
print("--- top by dc miss latency total")
for a,r in sorted(rows.items(),key=lambda kv:-kv[1][4])[:12]:
    i=bisect.bisect_left(addrs,a)
    print("n=%5d dcmiss=%5d avglat=%5.0f  %x: %s"%(r[0],r[3],r[4]/max(r[3],1),a,ins[i][1][:70]))
print("--- top by mispredicts")
for a,r in sorted(rows.items(),key=lambda kv:-kv[1][5])[:12]:
    i=bisect.bisect_left(addrs,a)
    print("n=%5d misp=%5d (%.0f%% of all)  %x: %s"%(r[0],r[5],100*r[5]/max(brm,1),a,ins[i][1][:70]))
