import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print(sys.argv[1], d['value'], d['ms_per_step'], ' | '.join('%s x%g %.1fus %.0fTF'%(n.replace('pconv_','').replace('128x64x64_','').replace('128x64_',''), v['launches_per_step'], 1e3*v['ms_per_step']/v['launches_per_step'], v.get('tflops') or 0) for n,v in k.items() if n.startswith('pconv') and 't4' in n))
