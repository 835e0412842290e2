#!/usr/bin/env python3
"""Turn the raw output of tools/collect_profiles.sh (gpurun_out/profiles/) into the committed evidence under profiles/.
   python tools/make_profiles.py r01"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src, dst = os.path.join(ROOT, 'gpurun_out', 'profiles'), os.path.join(ROOT, 'profiles')
B, A, RES = 1024, 64, 256
ALGO = B * A * 3 * RES * RES * 4

rows = list(csv.reader(open(os.path.join(src, 'bench_kernel_stats.csv'))))
with open(os.path.join(dst, f'{tag}_bench_kernel_stats.csv'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline   (MI355X; top 15 kernels by total time)\n')
    w = csv.writer(f)
    for r in rows[:16]:
        r[0] = r[0][:110]
        w.writerow(r)
log = open(os.path.join(src, 'bench_under_rocprof.log')).read().splitlines()
open(os.path.join(dst, f'{tag}_bench_under_rocprof.log'), 'w').write('\n'.join(l for l in log if l.startswith('{')) + '\n')

pmc = json.load(open(os.path.join(src, 'raster_pmc.json')))
key = max(pmc, key=lambda k: pmc[k].get('WRITE_SIZE', {}).get('mean', 0))
c = {n: v['mean'] for n, v in pmc[key].items()}
wb, fb = c['WRITE_SIZE'] * 1024, c['FETCH_SIZE'] * 1024
out = dict(kernel='raster_scene_bits_kernel<4, 3, float, SceneArgs>', batch=B, agents=A, res=RES,
           command='tools/collect_profiles.sh: rocprofv3 --pmc <set> --kernel-include-regex raster -- python3 tools/profile_raster.py --batch 1024 '
                   '--iters 2 (one pass per counter set, counters only)',
           counters=c, write_bytes_per_launch=wb, fetch_bytes_per_launch_raw=fb, fetch_bytes_per_launch_corrected=2 * fb,
           hbm_bytes_per_launch=wb + 2 * fb, algorithmic_bytes_per_launch=ALGO,
           notes='WRITE_SIZE / FETCH_SIZE are reported in KiB. FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (an upper '
                 'bound here: part of the reads are 32-byte grid-entry gathers). traffic = WRITE_SIZE + 2 FETCH_SIZE = '
                 f'{(wb + 2 * fb) / ALGO:.3f} x the algorithmic bytes (WRITE_SIZE alone: {wb / ALGO:.3f} x).')
json.dump(out, open(os.path.join(dst, f'{tag}_raster_pmc.json'), 'w'), indent=1)
json.dump(dict(batch=B, agents=A, res=RES, hbm_bytes_per_launch=wb + 2 * fb, source=f'profiles/{tag}_raster_pmc.json'),
          open(os.path.join(dst, 'raster_traffic.json'), 'w'), indent=1)
print(open(os.path.join(dst, f'{tag}_bench_kernel_stats.csv')).read()[:900])
print(json.dumps({k: out[k] for k in ('write_bytes_per_launch', 'fetch_bytes_per_launch_raw', 'hbm_bytes_per_launch')}))
