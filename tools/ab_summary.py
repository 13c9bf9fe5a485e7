import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/ab.txt'):
    if l.startswith('case'): print("%-60s" % l.strip()[5:], end=" ")
    elif l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%.3f ms %7.1f GB/s | parse %.3f expand %.3f x%d frac %.3f' % (d['ms_per_step'], d['value'], r['parse_kernel_ms_per_step'], r.get('expand_kernel_ms_per_step', r.get('decode_kernel_ms_per_step')), r['launches_per_step'], r['frac']))
    else: print(l.strip())
