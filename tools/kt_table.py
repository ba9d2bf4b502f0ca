import sys,re,collections
cur=None; d=collections.defaultdict(list)
for l in sys.stdin:
    l=l.rstrip()
    m=re.match(r'== (\S+) (\S+)',l)
    if m: cur=(m.group(1),m.group(2)); continue
    m=re.match(r'\s+(\S.*?)\s+([0-9.]+) us',l)
    if m and cur: d[(m.group(1).replace('mccnn::',''),cur[1],cur[0])].append(float(m.group(2)))
for k in sorted(d):
    if (k[0].startswith('f1') and k[1]=='1to64') or (k[0].startswith('conv_') and k[1]=='3to8') or (k[0].startswith('dw_') and k[1]=='dw256'): print('%-34s %-6s %-18s %s'%(k[0][:34],k[1],k[2],d[k]))
