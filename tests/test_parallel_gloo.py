"""world_size-2 gloo test of the scene-batch sharding used for multi-GPU runs (no GPU needed: host logic only)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from torchdrivesim_amd import parallel
    from torchdrivesim_amd.kinematic import KinematicBicycle
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.rendering import HipRendererConfig
    from torchdrivesim_amd.simulator import Simulator, TorchDriveConfig
    A = 3
    km = KinematicBicycle()
    km.set_params(lr=torch.ones(B, A))
    km.set_state(torch.arange(B * A * 4, dtype=torch.float32).reshape(B, A, 4))
    from torchdrivesim_amd.lanelet2 import LaneletMap
    lane_maps = [LaneletMap([], np.zeros((0, 3)), []) if b % 2 == 0 else None for b in range(B)]     # per-scene lane maps, some absent
    sim = Simulator(BirdviewMesh.empty(batch_size=B), km, torch.ones(B, A, 2), torch.ones(B, A, dtype=torch.bool),
                    TorchDriveConfig(renderer=HipRendererConfig()), lanelet_map=lane_maps)
    mine = parallel.shard_simulator(sim, rank, world)
    start, stop = parallel.scene_shard(B, rank, world)
    assert mine.batch_size == stop - start
    assert torch.equal(mine.get_state(), sim.get_state()[start:stop])          # shards are plain batch slices
    assert len(mine.lanelet_map) == stop - start and all(a is b for a, b in zip(mine.lanelet_map, lane_maps[start:stop]))
    # no data-path collective: the only exchanges are the timing reductions
    parallel.barrier()
    elapsed = 1.0 + rank                                                        # pretend rank 1 is the slow one
    rate = parallel.aggregate_throughput(mine.batch_size * A * 10, elapsed)
    slow = parallel.max_over_ranks(elapsed)
    np.save(os.path.join(out_dir, f'r{rank}.npy'), np.array([start, stop, rate, slow]))
    dist.destroy_process_group()


def test_scene_shards_cover_the_batch_exactly():
    from torchdrivesim_amd.parallel import scene_shard
    for total in (0, 1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            spans = [scene_shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharding(tmp_path):
    world, B = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / 'r0.npy'), np.load(tmp_path / 'r1.npy')
    assert (r0[0], r0[1], r1[0], r1[1]) == (0, 3, 3, 5)
    assert r0[3] == r1[3] == 2.0                                                # max over ranks
    assert r0[2] == r1[2] == (5 * 3 * 10) / 2.0                                # all units / slowest rank
