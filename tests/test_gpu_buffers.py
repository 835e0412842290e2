"""Where the rendered images live (csrc/alloc.hip, _ops.image_pool / owned_image): buffers whose physical pages are spread
out, handed to torch either through a memory pool (the reference-shaped `render_egocentric()` without `out=`) or as caller-owned tensors
(output rings).  No reference counterpart: rendering/cv2.py:52 allocates a numpy image per call."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_a_buffer_is_built_from_chunks_and_holds_data():
    from torchdrivesim_amd import _native as nat
    L = nat.lib()
    h = ctypes.c_void_p()
    nat.check(L.tds_buffer_create(1 << 30, 0, 0, ctypes.byref(h)), 'tds_buffer_create')
    n, chunks, spread = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
    nat.check(L.tds_buffer_info(h, ctypes.byref(n), ctypes.byref(chunks), ctypes.byref(spread)), 'tds_buffer_info')
    assert (n.value, chunks.value, spread.value) == (1 << 30, 128, 1) and L.tds_buffer_ptr(h) % (2 << 20) == 0
    nat.check(L.tds_buffer_destroy(h), 'tds_buffer_destroy')
    nat.check(L.tds_buffer_create(1 << 20, 0, 0, ctypes.byref(h)), 'tds_buffer_create')            # small: one hipMalloc
    nat.check(L.tds_buffer_info(h, None, ctypes.byref(chunks), ctypes.byref(spread)), 'tds_buffer_info')
    assert (chunks.value, spread.value) == (0, 0)
    nat.check(L.tds_buffer_destroy(h), 'tds_buffer_destroy')
    assert L.tds_buffer_create(0, 0, 0, ctypes.byref(h)) == nat.E_INVAL and L.tds_buffer_create(1 << 20, 0, 64, ctypes.byref(h)) == nat.E_INVAL


def test_owned_image_is_a_tensor_over_library_memory():
    from torchdrivesim_amd import _ops
    free0, _ = torch.cuda.mem_get_info(DEV)
    before = torch.cuda.memory_allocated(DEV)
    t = _ops.owned_image((64, 8, 3, 256, 256), torch.float32, DEV)              # 402 MB
    assert t.shape == (64, 8, 3, 256, 256) and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0
    assert torch.cuda.memory_allocated(DEV) == before                            # not torch's allocator
    t.fill_(3.0)
    v = t[5, 2]                                                                  # a view keeps the buffer alive
    del t
    assert float(v.sum()) == 3.0 * 3 * 256 * 256
    del v
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(DEV)
    assert abs(free1 - free0) < (64 << 20)                                       # back with the driver


def test_images_without_out_come_from_the_pool_and_show_the_same_pixels():
    import bench
    from torchdrivesim_amd import _ops
    from torchdrivesim_amd.utils import Resolution
    sim, actions, _ = bench.build_simulator(32, 16, torch.device(DEV), seed=11)           # 32 x 16 x 3 x 256 x 256 x 4 = 403 MB >= SPREAD_MIN
    sim.step(actions[0])
    res = Resolution(256, 256)
    assert _ops.use_image_pool
    _ops.use_image_pool = False
    try:
        plain = sim.render_egocentric(res=res, fov=35.0)
    finally:
        _ops.use_image_pool = True
    img = sim.render_egocentric(res=res, fov=35.0)
    assert torch.equal(img, plain)
    ptr = img.data_ptr()
    assert ptr % (2 << 20) == 0                                                  # a block of its own, built by tds_torch_alloc
    del img
    again = sim.render_egocentric(res=res, fov=35.0)                              # the block is cached by torch's allocator like any other
    assert again.data_ptr() == ptr and torch.equal(again, plain)
    held = sim.render_egocentric(res=res, fov=35.0)                               # while `again` is alive: another block, never an alias
    assert held.data_ptr() != ptr and torch.equal(held, again)
    small = sim.render_egocentric(res=Resolution(64, 64), fov=35.0)               # below SPREAD_MIN: the default pool
    assert small.shape[-1] == 64
    del again, held
    _ops.release_image_pool()


def test_out_is_validated_with_real_errors():
    import bench
    from torchdrivesim_amd.utils import Resolution
    sim, actions, _ = bench.build_simulator(2, 4, torch.device(DEV), seed=1)
    res = Resolution(64, 64)
    good = torch.empty(2, 4, 3, 64, 64, device=DEV)
    assert sim.render_egocentric(res=res, out=good) is good
    for bad, word in ((torch.empty(2, 4, 3, 64, 32, device=DEV), 'shape'), (torch.empty(2, 4, 3, 64, 64, device=DEV, dtype=torch.uint8), 'float32'),
                      (torch.empty(2, 4, 3, 64, 128, device=DEV)[..., ::2], 'contiguous'),
                      (torch.empty(2 * 4 * 3 * 64 * 64 + 1, device=DEV)[1:].view(2, 4, 3, 64, 64), 'aligned')):
        with pytest.raises(RuntimeError, match=word):
            sim.render_egocentric(res=res, out=bad)


def test_the_workspace_is_sized_for_the_kernels_that_run():
    """tds_raster_scene_workspace_bytes_for: with the number of distinct keys known (tds_map_keys + the actors' keys) and at most
    15 of them the bit-plane kernels serve the launch -- face lists up to 144 x 144 (float32) / 208 x 208 (uint8), above nothing but the 64
    bytes of work queues -- instead of the 32 KB per camera that would serve every path."""
    import bench
    from torchdrivesim_amd import _ops
    from torchdrivesim_amd.utils import Resolution
    sim, actions, _ = bench.build_simulator(4, 16, torch.device(DEV), seed=2)
    keys = sim._scene()['maps'][0][0].face_keys()
    assert keys is not None and 1 <= len(keys) <= 8 and len(set(keys)) == len(keys)
    _ops._workspaces.clear()
    img = sim.render_egocentric(res=Resolution(256, 256), fov=35.0)
    (ws,) = [w for w in _ops._workspaces.values() if w is not False]
    assert ws.numel() <= 4096, ws.numel()                                     # the persistent launch's queues, nothing else
    _ops._workspaces.clear()
    small = sim.render_egocentric(res=Resolution(64, 64), fov=35.0)
    (ws64,) = [w for w in _ops._workspaces.values() if w is not False]
    assert ws64.numel() >= 4 * 16 * 2048 * 16                                   # a list of 2 048 faces per camera for K3s + K3r
    assert bool(img.any()) and bool(small.any())
    # without the key count: the size that serves every path
    import ctypes
    from torchdrivesim_amd import _native as nat
    every, bits, u8hi = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    nat.call('tds_raster_scene_workspace_bytes', torch.device(DEV), 64, 256, ctypes.byref(every))
    nat.call('tds_raster_scene_workspace_bytes_for', torch.device(DEV), 64, 256, nat.OUT_F32, 5, ctypes.byref(bits))
    nat.call('tds_raster_scene_workspace_bytes_for', torch.device(DEV), 64, 256, nat.OUT_F32, 40, ctypes.byref(u8hi))
    assert bits.value <= 256 < every.value == u8hi.value
