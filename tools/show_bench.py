"""Pretty-print a bench.py JSON line: python tools/show_bench.py gpurun_out/x.json"""
import json
import sys
r = json.load(open(sys.argv[1]))
print({k: r[k] for k in ('value', 'ms_per_step')}, 'seq', r['config']['sequential_ms_per_step'], 'strong', r['strong'] and (r['strong']['value'], r['strong']['ms_per_step']))
for k, v in (r.get('layers') or {}).items():
    print(' layer', k, v['ms_per_step'], 'ms', round(v['value'] / 1e6, 1), 'Mpts/s', v.get('conv_ms'), v.get('roofline', {}).get('bound'), v.get('roofline', {}).get('frac'))
for k, v in (r.get('configs') or {}).items():
    if 'error' in v:
        print(k, v['error'][:200])
        continue
    print(k, v['points'], v['level_sizes'], 'ms', v['ms_per_step'], 'Mpts/s', round(v['value'] / 1e6, 2), 'launches', v['library_launches_per_step'], 'hier', v['hierarchy_ms'], 'conv cached', v['conv_fwd_bwd_ms_cached_geometry'])
    c = v.get('cpu_baseline') or {}
    print('   cpu', c.get('value'), c.get('sample', c.get('error')), c.get('gpu_vs_oracle_max_rel_err_f32_layers'))
    if len(sys.argv) > 2:
        for l in v['layers']:
            print('     %-9s F%-4d %s n %6d m %6d E %8d fwd %.4f bwd %.4f  frac %.4f %.4f' % (l['name'], l['fin'], 'c' if l['combin'] else 'd', l['points_in'], l['centres'], l['edges'], l['fwd_ms'], l['bwd_ms'], l['roofline_fwd']['frac'], l['roofline_bwd']['frac']))
