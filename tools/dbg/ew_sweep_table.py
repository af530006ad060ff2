import json,sys
for l in open(sys.argv[1]):
    tag, js = l.split(' ',1)
    d=json.loads(js); k=d['kernels']
    print(tag, d['ms_per_step'], {n: k[n]['ms'] for n in k if 'pool_bwd' in n or 'upadd_bwd' in n})
