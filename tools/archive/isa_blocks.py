import re,collections,sys
t=open(sys.argv[1]).read()
m=re.search(r'^%s:.*?s_endpgm'%sys.argv[2], t, re.S|re.M)
body=m.group(0)
blocks=[]; cur=('entry',[])
for l in body.split('\n'):
    if re.match(r'^\.LBB\d+_\d+:',l):
        blocks.append(cur); cur=(l.split(':')[0],[])
    elif l.startswith('\t') and l.strip() and l.strip()[0] not in ';.':
        cur[1].append(l.strip().split()[0])
blocks.append(cur)
for n,ops in blocks:
    c=collections.Counter(ops)
    print(n, len(ops), 'valu', sum(v for k,v in c.items() if k.startswith('v_')), 'nop', c['s_nop'], c.most_common(8))
