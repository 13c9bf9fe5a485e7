import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/ab.txt'):
    if l.startswith('case'): print("%-60s" % l.strip()[5:], end=" ")
    elif l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%.3f ms %7.1f GB/s | %s %.3f ms x%d (parse %.3f) frac %.3f' % (
            d['ms_per_step'], d['value'], r['kernel'], r.get('%s_ms_per_step' % r['kernel'], 0.0),
            r.get('launches_per_step', 1), r.get('parse_kernel_ms_per_step', 0.0), r['frac']))
    else: print(l.strip())
