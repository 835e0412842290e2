"""bench.py's own multi-GPU launcher (BASELINE.json config 4), end to end on the CPU: `python bench.py --gpus 2 --dry-run`
starts one worker per rank, the workers rendezvous on 127.0.0.1 (gloo), run the barrier / timed loop / max-over-ranks path without
any kernel, and rank 0 prints ONE well-formed JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env=None):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'TDS_BENCH_BACKEND'):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def check_line(line, n, steps, warmup):
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline'):
        assert k in line, k
    assert line['n_gpus'] == n and line['steps'] == steps and line['warmup'] == warmup
    assert line['scaling'] == 'weak' and line['higher_is_better'] is True and line['unit'] == 'agent-steps/s'
    assert line['config']['global_batch'] == n * 1024 and line['config']['agents'] == 64
    assert f'x{n}' in line['config']['parallelism'] and 'no collectives' in line['config']['parallelism']
    assert line['value'] > 0 and line['dry_run'] is True
    # which device every rank ran on is part of the record (none in a dry run: no rank can tell, distinct_devices is null)
    assert 'distinct_devices' in line and line['distinct_devices'] is None
    reports = line['per_rank'] if n > 1 else [dict(device_identity=line['device_identity'])]
    for r in reports:
        assert set(r['device_identity']) >= {'device', 'uuid', 'pci', 'arch', 'compute_units', 'reserved_usable', 'visible'}


def test_self_launch_two_ranks_dry_run():
    line = check = run_bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run')
    check_line(line, 2, 3, 1)
    assert len(line['per_rank_agent_steps_per_s']) == 2 and 'self-launch' in line['launcher']
    assert 'cpu_baseline' not in check                               # rank 0 of N = 1 only


def test_device_list_is_narrowed_per_rank():
    # the launcher hands rank r the r-th entry of an inherited device list
    line = run_bench('--gpus', '2', '--steps', '1', '--warmup', '0', '--dry-run', env=dict(HIP_VISIBLE_DEVICES='5,3,1'))
    check_line(line, 2, 1, 0)
    assert [r['device_identity']['visible'] for r in line['per_rank']] == ['5', '3']


def test_distinct_devices_counts_physical_gpus():
    sys.path.insert(0, ROOT)
    import bench
    rep = lambda uuid, pci=None: dict(device_identity=dict(uuid=uuid, pci=pci))
    assert bench.distinct_devices([rep('a'), rep('b'), rep('a')]) == 2
    assert bench.distinct_devices([rep(None, '0000:05:00'), rep(None, '0000:05:00')]) == 1
    assert bench.distinct_devices([rep(None), rep('None')]) is None


def test_single_process_dry_run_and_torchrun_shape():
    line = run_bench('--steps', '2', '--warmup', '1', '--dry-run')
    check_line(line, 1, 2, 1)
    assert 'per_rank_agent_steps_per_s' not in line
    # the worker also runs under an external launcher that provides RANK / WORLD_SIZE (the driver's torch.distributed.run shape)
    line = run_bench('--gpus', '1', '--steps', '2', '--warmup', '1', '--dry-run', env=dict(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1'))
    check_line(line, 1, 2, 1)


def test_the_barrier_falls_back_to_gloo_when_rccl_cannot_come_up():
    """The process group only carries the barrier and the timing reductions.  Under an external launcher the worker asks for RCCL; where that
    fails (here: no GPU at all) both ranks must fall back to gloo and still deliver the line."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), TDS_BENCH_DRY_RUN_TRY_NCCL='1')
        e.pop('TDS_BENCH_BACKEND', None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--dry-run'],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all('falling back to gloo' in err for _, err in outs), outs
    lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines) == 1 and not outs[1][0].strip()
    line = json.loads(lines[0])
    check_line(line, 2, 2, 1)
    assert 'gloo' in line['launcher']


def test_traffic_figure_is_refused_for_another_build(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    val, note = bench.stamped_traffic(1024, 64)
    tj = json.load(open(os.path.join(ROOT, 'profiles', 'raster_traffic.json')))
    ent = tj.get('f32', tj)
    if ent.get('kernel_source_sha') == bench.kernel_source_stamp():
        assert val == ent['hbm_bytes_per_launch']
    else:
        assert val is None and 'refused' in note
    monkeypatch.setattr(bench, 'kernel_source_stamp', lambda: 'deadbeefdeadbeef')
    val, note = bench.stamped_traffic(1024, 64)
    assert val is None and 'refused' in note


def test_limiter_of_the_instruction_bound_modes_is_stamped_like_the_traffic(tmp_path, monkeypatch):
    """roofline_u8 / _128 / _64 say bound = 'valu' and quote the PMC-derived VALU figures of profiles/raster_traffic.json -- only for the build and the
    workload they were measured on"""
    sys.path.insert(0, ROOT)
    import bench
    fake = dict(u8=dict(batch=1024, agents=64, res=256, hbm_bytes_per_launch=1.3e10, kernel_source_sha='feedfacefeedface', source='profiles/x.json',
                        valu=dict(valu_busy_simds_per_se_of_32=29.3, valu_issue_fraction_of_add_chain=0.617, cycles_per_valu_instruction=6.5, valu_lane_occupancy=0.73, valu_instructions_per_launch=2.75e9, valu_issue_ms_at_4_cycles=4.48)),
                f32_64=dict(batch=1024, agents=64, res=64, hbm_bytes_per_launch=3.3e9, kernel_source_sha='feedfacefeedface', source='profiles/x.json', valu=None))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    os.makedirs(tmp_path / 'profiles')
    json.dump(fake, open(tmp_path / 'profiles' / 'raster_traffic.json', 'w'))
    monkeypatch.setattr(bench, 'kernel_source_stamp', lambda: 'feedfacefeedface')
    lim = bench.stamped_limiter(1024, 64, 'u8')
    assert lim['bound'] == 'valu' and lim['valu_busy_simds_per_se_of_32'] == 29.3 and lim['cycles_per_valu_instruction'] == 6.5 and 'x.json' in lim['source']
    assert bench.stamped_limiter(256, 64, 'u8') == dict(bound='valu', note='PMC figure is for another workload')
    assert 'no PMC figure' in bench.stamped_limiter(1024, 64, 'f32_64', res=64)['note']          # an entry without VALU counters
    assert 'no PMC figure' in bench.stamped_limiter(1024, 64, 'f32_128', res=128)['note']
    monkeypatch.setattr(bench, 'kernel_source_stamp', lambda: 'deadbeefdeadbeef')
    lim = bench.stamped_limiter(1024, 64, 'u8')
    assert lim['bound'] == 'valu' and 'refused' in lim['note'] and 'valu_issue_fraction_of_add_chain' not in lim


def test_a_dying_worker_ends_the_launch_promptly():
    """rank 1 exits at start-up while rank 0 waits for it in the rendezvous: the launcher polls its children, reports the failure and
    ends rank 0 instead of waiting for the timeout"""
    import time
    t0 = time.time()
    e = dict(os.environ, TDS_BENCH_DRY_RUN_FAIL_RANK='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'TDS_BENCH_BACKEND'):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '2', '--warmup', '0'], capture_output=True,
                       text=True, timeout=300, env=e)
    assert r.returncode != 0 and 'worker 1 failed' in (r.stderr + r.stdout) and time.time() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]          # no result line from a failed launch
