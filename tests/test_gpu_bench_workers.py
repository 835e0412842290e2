"""bench.py's multi-process path with REAL workers and REAL kernels on the one GPU a test box has (the launcher had only ever run
--dry-run with more than one worker).  Both ways the driver may start it (BASELINE.json config 4):
  * `python bench.py --gpus 2` by itself with HIP_VISIBLE_DEVICES=0,0: the launcher narrows each child to "its" device (both are GPU 0 here),
    gloo barrier on 127.0.0.1;
  * under `python -m torch.distributed.run --nproc-per-node 2`: the worker asks for RCCL; two ranks on one GPU is something RCCL may refuse
    (duplicate device) -- then every rank must agree on gloo -- or serve; either way ONE well-formed line comes out.
Two ranks share the GPU, so the rates are not a scaling measurement; what is checked is that the sharded path runs kernels and reports."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--gpus', '2', '--batch', '128', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-configs']


def clean_env(**extra):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'TDS_BENCH_BACKEND', 'CUDA_VISIBLE_DEVICES'):
        e.pop(k, None)
    e.update(HSA_ENABLE_IPC_MODE_LEGACY='0', **extra)
    return e


def check(stdout, launcher_words):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1 and line['scaling'] == 'weak'
    assert line['config']['global_batch'] == 256 and 'x2' in line['config']['parallelism'] and 'no collectives' in line['config']['parallelism']
    assert len(line['per_rank_agent_steps_per_s']) == 2 and all(v > 0 for v in line['per_rank_agent_steps_per_s'])
    assert line['value'] > 0 and 'dry_run' not in line and line['data'] == 'synthetic'
    # whole-job rate = all units / slowest rank's time: never above the sum of the per-rank rates
    assert line['value'] <= sum(line['per_rank_agent_steps_per_s']) * 1.0001
    roof = line['roofline']
    assert roof['kernel'] == 'raster_scene_bits_kernel' and roof['avg_launch_ms'] > 0 and 0 < roof['frac'] < 1      # kernels really ran
    assert any(w in line['launcher'] for w in launcher_words), line['launcher']
    assert [p['rank'] for p in line['per_rank']] == [0, 1] and all(p['avg_launch_ms'] > 0 and not p['ring_probe']['aliased'] for p in line['per_rank'])
    return line


def test_self_launch_two_real_workers_on_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *ARGS], capture_output=True, text=True, timeout=900,
                       env=clean_env(HIP_VISIBLE_DEVICES='0,0'))
    assert r.returncode == 0, r.stderr[-3000:]
    check(r.stdout, ['self-launch'])


def test_two_real_workers_under_torch_distributed_run():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), *ARGS], capture_output=True, text=True, timeout=900,
                       env=clean_env(TDS_BENCH_RCCL_TIMEOUT='60'))
    assert r.returncode == 0, r.stderr[-3000:]
    line = check(r.stdout, ['external launcher'])
    assert 'RCCL barrier' in line['launcher'] or ('gloo' in line['launcher'] and 'falling back to gloo' in r.stderr)


def test_eight_real_workers_on_one_gpu():
    """The largest world the driver launches (BASELINE.json config 4: 8 GPUs), with eight real workers sharing the one GPU of a test box:
    rendezvous of eight children, eight image rings probed side by side, eight per-rank reports in the one line."""
    args = ['--gpus', '8', '--batch', '32', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-configs', '--ring-candidates', '2']
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True, timeout=1500,
                       env=clean_env(HIP_VISIBLE_DEVICES='0,0,0,0,0,0,0,0'))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['config']['global_batch'] == 256 and 'x8' in line['config']['parallelism']
    assert len(line['per_rank_agent_steps_per_s']) == 8 and all(v > 0 for v in line['per_rank_agent_steps_per_s'])
    assert line['value'] <= sum(line['per_rank_agent_steps_per_s']) * 1.0001
    per_rank = line['per_rank']
    assert [p['rank'] for p in per_rank] == list(range(8))
    for p in per_rank:
        assert p['avg_launch_ms'] > 0 and len(p['ring_probe']['kept']) == 2 and len(set(p['ring_probe']['kept'])) == 2 and not p['ring_probe']['aliased']
        assert len(p['ring_probe']['launch_ms']) >= 2 and p['ring_probe']['fill_ms'] > 0
    # the record says which physical GPU every rank sat on -- here all eight on the one GPU of the box
    ids = [p['device_identity'] for p in per_rank]
    assert all(i['uuid'] not in (None, 'None') or i['pci'] for i in ids) and all(i['arch'].startswith('gfx950') and i['compute_units'] > 0 for i in ids)
    assert len({(i['uuid'], i['pci']) for i in ids}) == 1 and line['distinct_devices'] == 1
    assert all(i['reserved_usable'] in (True, False) for i in ids)
