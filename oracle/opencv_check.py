"""
Live pin of the oracle's pixel fill against the real OpenCV -- TEST INFRASTRUCTURE (tests/test_oracle_fill.py and
__graft_entry__.smoke() call it wherever `cv2` imports; it is absent from this image and from the GPU box, so the fill rule of
oracle/tds_oracle.c -- orc_fill_convex_poly / orc_line / clip_line -- is PARITY UNPINNED until this has run somewhere).

What is compared: `cv2.fillConvexPoly(img float32 (H,W,3), points int32 (3,2), color, shift=0, lineType=cv2.LINE_AA)`, the exact call of
the reference's CV2 backend (rendering/cv2.py:54-59), against oracle.fill_convex_poly on
  * random triangles whose vertices reach far outside the image (clipLine) and degenerate ones,
  * the reference's own pre-raster call lists for Town01 scenes (tests/golden/g45_mesh_preraster.npz, recorded from the imported reference
    by tools/gen_golden.py): every image replayed call by call, whole images compared.
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def g5_call_lists():
    """[(name, res, tris int32 (n_img, n_calls, 3, 2), colours uint8 (n_img, n_calls, 3))] from the golden file: per image the calls in
    the order the reference made them (every image of a scene has the same number of calls, padding included)"""
    g = np.load(os.path.join(GOLDEN, 'g45_mesh_preraster.npz'), allow_pickle=False)
    return [(m['name'], int(m['res']), g[f"g5_{m['name']}_tris"], g[f"g5_{m['name']}_cols"]) for m in json.loads(str(g['g5_meta']))]


def cross_check(orc, n_random=2000) -> int:
    """Raises AssertionError on the first differing image; returns the number of images compared.  Needs cv2."""
    import cv2
    rng = np.random.default_rng(1)
    n = 0
    for _ in range(n_random):
        pts = rng.integers(-300, 364, size=(3, 2)).astype(np.int32)
        if rng.uniform() < 0.1:
            pts[2] = pts[1]                                       # degenerate: a line or a point
        ref = cv2.fillConvexPoly(img=np.zeros((64, 64, 3), np.float32), points=pts, color=[7, 8, 9], shift=0, lineType=cv2.LINE_AA)
        mine = np.zeros((64, 64, 3), np.float32)
        orc.fill_convex_poly(mine, pts, (7, 8, 9))
        np.testing.assert_array_equal(mine, ref, err_msg=f'triangle {pts.tolist()}')
        n += 1
    for name, res, tris, cols in g5_call_lists():
        for i in range(tris.shape[0]):
            ref = np.zeros((res, res, 3), np.float32)
            mine = np.zeros((res, res, 3), np.float32)
            for t, c in zip(tris[i], cols[i]):
                ref = cv2.fillConvexPoly(img=ref, points=np.ascontiguousarray(t, np.int32), color=[int(v) for v in c], shift=0, lineType=cv2.LINE_AA)
                orc.fill_convex_poly(mine, np.ascontiguousarray(t, np.int32), tuple(int(v) for v in c))
            np.testing.assert_array_equal(mine, ref, err_msg=f'{name}, image {i}')
            n += 1
    return n
