#!/usr/bin/env python3
"""Turn the raw output of tools/collect_profiles.sh (gpurun_out/profiles/) into the committed evidence under profiles/.
   python tools/make_profiles.py r02
Writes  profiles/<tag>_bench_kernel_stats.csv, <tag>_config{2,3,5}_kernel_stats.csv (rocprofv3 --kernel-trace --stats, top kernels),
        profiles/<tag>_raster_pmc.json (counters of the raster kernel, float32 and uint8 output, HBM traffic with the gfx950 corrections)
        profiles/raster_traffic.json   (what bench.py reports as roofline.traffic -- stamped with the hash of the kernel's sources and
                                        refused by bench.py for any other build)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
src, dst = os.path.join(ROOT, 'gpurun_out', 'profiles'), os.path.join(ROOT, 'profiles')
B, A, RES = 1024, 64, 256
stamp = bench.kernel_source_stamp()

commands = dict(bench='python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs',
                config2='python3 tools/bench_configs.py --only config2 --steps 20 --warmup 3',
                config3='python3 tools/bench_configs.py --only config3 --steps 20 --warmup 3',
                config5='python3 tools/bench_configs.py --only config5 --steps 20 --warmup 3')
for name, cmd in commands.items():
    path = os.path.join(src, f'{name}_kernel_stats.csv')
    if not os.path.exists(path):
        print('missing', path)
        continue
    rows = list(csv.reader(open(path)))
    with open(os.path.join(dst, f'{tag}_{name}_kernel_stats.csv'), 'w') as f:
        f.write(f'# rocprofv3 --kernel-trace --stats -- {cmd}   (MI355X; top 15 kernels by total time; kernel source {stamp})\n')
        if name == 'bench':
            # what the raster launches of this trace are: the average in the table is over ALL of them; the bench line's figure is over the
            # launches of its timed region, which the per-dispatch rows of the same trace give too (timed_only_avg_ms)
            try:
                line = [json.loads(l) for l in open(os.path.join(src, 'bench_under_rocprof.log')) if l.startswith('{')][-1]
                roof = line['roofline']
                probe = roof['ring_probe']
                f.write(f"# raster_scene_bits_kernel launches of this run, in order: 3 per candidate of the image ring incl. its first touch (candidates "
                        f"{[round(x, 2) for x in probe['launch_ms']]} ms, first touches {[round(x, 1) for x in probe['first_touch_ms']]} ms, kept {probe['kept']}), "
                        f"3 warm-up, the 20 of the timed region, 6 of the stream-only reference launch ({roof['measured_stream_ms']:.2f} ms), "
                        f"23 of the loop without out= (default_path)\n")
                first, end = roof['timed_raster_calls']
                rows_t = list(csv.DictReader(open(os.path.join(src, 'bench_raster_trace.csv'))))
                rows_t.sort(key=lambda r: int(r['Start_Timestamp']))
                timed = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows_t[first:end]]
                every = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows_t]
                avg = sum(timed) / len(timed)
                f.write(f"# timed_only_avg_ms = {avg:.4f} over launches {first}..{end - 1} of {len(every)} in this trace (min {min(timed):.4f}, max {max(timed):.4f}); "
                        f"the bench line of the same run says avg_launch_ms = {roof['avg_launch_ms']:.4f} (HIP events), i.e. {100 * (roof['avg_launch_ms'] / avg - 1):+.2f} %; "
                        f"all_launches_avg_ms = {sum(every) / len(every):.4f}\n")
            except Exception as exc:          # noqa: BLE001
                f.write(f'# (no bench line / per-dispatch rows beside the trace: {exc})\n')
        w = csv.writer(f)
        for r in rows[:16]:
            r[0] = r[0][:110]
            w.writerow(r)
    log = os.path.join(src, f'{name}_under_rocprof.log')
    if os.path.exists(log):
        lines = [l for l in open(log).read().splitlines() if l.startswith('{')]
        open(os.path.join(dst, f'{tag}_{name}_under_rocprof.log'), 'w').write('\n'.join(lines) + '\n')
    if name == 'bench':
        # the plain runs of the same command on the same box, right before and right after the traced one
        plain = []
        for when in ('before', 'after'):
            q = os.path.join(src, f'bench_plain_{when}.log')
            if os.path.exists(q):
                plain += [f'# plain run {when} the traced one: ' + l for l in open(q).read().splitlines() if l.startswith('{')]
        q = os.path.join(src, 'bench_traced_runs.txt')
        if os.path.exists(q):
            plain = ['# ' + l.strip() for l in open(q)] + plain
        if plain:
            open(os.path.join(dst, f'{tag}_bench_plain_same_box.log'), 'w').write('\n'.join(plain) + '\n')

out, traffic = dict(kernel_source_sha=stamp, batch=B, agents=A, res=RES,
                    command='tools/collect_profiles.sh: rocprofv3 --pmc <set> --kernel-include-regex raster -- python3 tools/profile_raster.py '
                            '--batch 1024 --iters 2 [--u8]  (one pass per counter set, counters only, the PRODUCT library)'), {}
for mode, bpp, mres in (('f32', 4, RES), ('u8', 1, RES), ('f32_128', 4, 128), ('f32_64', 4, 64)):
    path = os.path.join(src, f'raster_pmc_{mode}.json')
    if not os.path.exists(path):
        continue
    pmc = json.load(open(path))
    if mres == RES:
        key = max(pmc, key=lambda k: pmc[k].get('WRITE_SIZE', {}).get('mean', 0))
        c = {n: v['mean'] for n, v in pmc[key].items()}
    else:
        # the split form: two kernels per render call (K3s, K3r) -- the counters of both, summed
        key = ' + '.join(sorted(pmc))
        c = {}
        for k in pmc:
            for n, v in pmc[k].items():
                c[n] = c.get(n, 0.0) + v['mean']
    algo = B * A * 3 * mres * mres * bpp
    wb, fb = c['WRITE_SIZE'] * 1024, c['FETCH_SIZE'] * 1024
    waves = c.get('SQ_WAVES', 0) or 1
    ent = dict(kernel=key, counters=c, write_bytes_per_launch=wb, fetch_bytes_per_launch_raw=fb, fetch_bytes_per_launch_corrected=2 * fb,
               hbm_bytes_per_launch=wb + 2 * fb, algorithmic_bytes_per_launch=algo, traffic_over_algorithmic=(wb + 2 * fb) / algo,
               valu_instructions_per_wave=c.get('SQ_INSTS_VALU', 0) / waves,
               # the launch is persistent since round 3 (a wave renders many images): per image and wave of its workgroup, the figure of rounds 1 - 2
               valu_instructions_per_image_and_wave=c.get('SQ_INSTS_VALU', 0) / (B * A * 4),
               lds_bank_conflict_share=c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0), 1),
               # SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES: the number of SIMDs with a VALU instruction in flight, averaged over the busy cycles
               # of a shader engine -- out of its 32 SIMDs... NOT a percentage (the old name `valu_utilisation` had no unit)
               valu_busy_simds_per_se_of_32=c.get('SQ_ACTIVE_INST_VALU', 0) * 4 / max(c.get('SQ_BUSY_CYCLES', 0), 1) if c.get('SQ_BUSY_CYCLES') else None,
               # calibrated (tools/valu_calibrate.hip, profiles/r04_valu_calibration.log): SQ_ACTIVE_INST_VALU reads 1.000 per v_add_u32 of a saturating
               # dependent chain (3 or 6 waves per SIMD, 64 or 16 active lanes), and the quantity above reads 47.5 there -- so this is the fraction of the
               # VALU issue rate a pure add chain reaches
               valu_issue_fraction_of_add_chain=(c.get('SQ_ACTIVE_INST_VALU', 0) * 4 / max(c.get('SQ_BUSY_CYCLES', 0), 1) / 47.5) if c.get('SQ_BUSY_CYCLES') else None,
               # lanes doing work per VALU instruction: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) (where the pass collected it)
               valu_lane_occupancy=(c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU'])) if c.get('SQ_THREAD_CYCLES_VALU') and c.get('SQ_ACTIVE_INST_VALU') else None)
    out[mode] = ent
    # what bench.py quotes beside a mode that is bound by instruction issue, not by HBM: the share of the SIMDs with a VALU
    # instruction in flight, the average issue cost of one (4 cycles is the rate of a dependent add chain: tools/valu_calibrate.hip), the lanes at
    # work per instruction, and the time the VALU instructions alone take on 1 024 SIMDs at 2.4 GHz
    valu = None
    if ent['valu_busy_simds_per_se_of_32'] is not None:
        # valu_busy_simds_per_se_of_32 is the RAW quantity SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES ("SIMDs of a shader engine with a VALU instruction
        # in flight"); it is not bounded by 32 -- a saturating dependent add chain reads 47.5 (profiles/r04_valu_calibration.log) -- so the calibrated
        # figure beside it, valu_issue_fraction_of_add_chain = raw / 47.5, is the one to read as "how busy"
        valu = dict(valu_busy_simds_per_se_of_32=ent['valu_busy_simds_per_se_of_32'], valu_issue_fraction_of_add_chain=ent['valu_issue_fraction_of_add_chain'],
                    cycles_per_valu_instruction=4.0 / ent['valu_issue_fraction_of_add_chain'] if ent['valu_issue_fraction_of_add_chain'] else None,
                    valu_lane_occupancy=ent['valu_lane_occupancy'],
                    valu_instructions_per_launch=c.get('SQ_INSTS_VALU'),
                    valu_issue_ms_at_4_cycles=c.get('SQ_INSTS_VALU', 0) * 4 / (1024 * 2.4e9) * 1e3)
    traffic[mode] = dict(batch=B, agents=A, res=mres, hbm_bytes_per_launch=wb + 2 * fb, kernel_source_sha=stamp, source=f'profiles/{tag}_raster_pmc.json', valu=valu)
out['notes'] = ('WRITE_SIZE / FETCH_SIZE are reported in KiB. FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (an upper bound here: part '
                'of the reads are 32-byte grid-entry gathers). traffic = WRITE_SIZE + 2 FETCH_SIZE. lds_bank_conflict_share = SQ_LDS_BANK_CONFLICT / '
                'SQ_LDS_IDX_ACTIVE (round 1: 0.49 with the row-major plane layout).')
json.dump(out, open(os.path.join(dst, f'{tag}_raster_pmc.json'), 'w'), indent=1)
if traffic:
    json.dump(traffic, open(os.path.join(dst, 'raster_traffic.json'), 'w'), indent=1)
for mode in ('f32', 'u8', 'f32_128', 'f32_64'):
    if mode in out:
        e = out[mode]
        print(mode, {k: e[k] for k in ('write_bytes_per_launch', 'fetch_bytes_per_launch_raw', 'traffic_over_algorithmic', 'valu_instructions_per_wave',
                                       'lds_bank_conflict_share')})
