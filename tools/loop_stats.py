import re,collections,sys
txt=open('/tmp/attention.dis').read()
want=sys.argv[1]
for f in re.split(r'\n(?=[0-9a-f]+ <)',txt):
    m=re.match(r'[0-9a-f]+ <(.*)>:',f)
    if not m or want not in m.group(1): continue
    ops=[]
    for l in f.split('\n')[1:]:
        t=l.split('//')[0].strip()
        if not t: continue
        mm=re.search(r'//\s*([0-9A-Fa-f]+):',l)
        if mm: ops.append((int(mm.group(1),16),t))
    for i,(a,t) in enumerate(ops):
        m2=re.match(r'(s_cbranch_\w+|s_branch)\s+(\d+)',t)
        if not m2: continue
        off=int(m2.group(2)); off=off-65536 if off>32767 else off
        tgt=a+4+off*4
        if tgt<=a:
            j=[k for k,(aa,_) in enumerate(ops) if aa==tgt][0]
            body=[x for _,x in ops[j:i+1]]
            c=collections.Counter(x.split()[0] for x in body)
            mf=sum(v for k,v in c.items() if 'mfma' in k)
            if mf<10: continue
            cls=collections.Counter()
            for k,v in c.items():
                g='mfma' if 'mfma' in k else 'branch' if k.startswith(('s_cbranch','s_branch')) else 's_' if k.startswith('s_') else 'ds' if k.startswith('ds_') else 'vmem' if k.startswith(('buffer','global')) else 'valu'
                cls[g]+=v
            print(m.group(1)[-40:], f'loop {j}..{i} ({i-j+1} instr)', dict(cls))
