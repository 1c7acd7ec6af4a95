import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']
        want = sys.argv[2:] or ['dot_bwd_weight', 'recon_grad_mix', 'dot_bwd_data', 'dot_fwd']
        print(sys.argv[1], d['value'], d['ms_per_step'], ' | '.join('%s x%g %.1fus' % (n, k[n]['launches_per_step'], 1e3 * k[n]['ms_per_step'] / k[n]['launches_per_step']) for n in want if n in k))
