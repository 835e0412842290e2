"""
Known-answer and property tests of the OpenCV fill restatement (oracle/tds_oracle.c: orc_fill_convex_poly).
PARITY UNPINNED against OpenCV itself in this environment (cv2 absent, unpinned in the reference's
pyproject.toml:27); `test_live_opencv_cross_check` runs automatically wherever cv2 is importable.
"""
import numpy as np
import pytest


def fill(oracle, pts, res=16, color=(1, 2, 3)):
    img = np.zeros((res, res, 3), np.float32)
    oracle.fill_convex_poly(img, np.array(pts, np.int32), color)
    return img


def test_degenerate_triangle_is_one_pixel(oracle):
    img = fill(oracle, [[5, 7]] * 3)
    ys, xs = np.nonzero(img[..., 0])
    assert list(zip(xs, ys)) == [(5, 7)]          # OpenCV point (x,y) -> img[y, x]
    assert img[7, 5].tolist() == [1, 2, 3]


def test_offscreen_degenerate_draws_nothing(oracle):
    assert not fill(oracle, [[-3, 4]] * 3).any()
    assert not fill(oracle, [[4, 99]] * 3).any()


def test_axis_aligned_right_triangle(oracle):
    # (2,2),(10,2),(2,10): legs drawn by Line, hypotenuse by Bresenham x+y=12, interior by spans
    img = fill(oracle, [[2, 2], [10, 2], [2, 10]])
    cov = img[..., 0] > 0
    exp = np.zeros((16, 16), bool)
    for y in range(2, 11):
        for x in range(2, 13 - y):
            exp[y, x] = True
    np.testing.assert_array_equal(cov, exp)


def test_rectangle_from_two_triangles_like_an_agent(oracle):
    # faces [0,1,3],[1,3,2] of an axis-aligned 8x4 box (mesh.py:955): union must be the full box, no gaps
    c = [[3, 4], [11, 4], [11, 8], [3, 8]]
    img = np.zeros((16, 16, 3), np.float32)
    oracle.fill_convex_poly(img, np.array([c[0], c[1], c[3]], np.int32), (9, 9, 9))
    oracle.fill_convex_poly(img, np.array([c[1], c[3], c[2]], np.int32), (9, 9, 9))
    exp = np.zeros((16, 16), bool)
    exp[4:9, 3:12] = True
    np.testing.assert_array_equal(img[..., 0] > 0, exp)


def test_vertex_order_and_clipping_properties(oracle):
    rng = np.random.default_rng(0)
    for _ in range(300):
        pts = rng.integers(-20, 52, size=(3, 2))
        a = fill(oracle, pts, res=32)[..., 0] > 0
        # every in-image vertex is drawn (Line end points)
        for x, y in pts:
            if 0 <= x < 32 and 0 <= y < 32:
                assert a[y, x]
        # footprint stays inside the bounding box of the triangle
        ys, xs = np.nonzero(a)
        if len(xs):
            assert xs.min() >= max(0, pts[:, 0].min()) and xs.max() <= min(31, pts[:, 0].max())
            assert ys.min() >= max(0, pts[:, 1].min()) and ys.max() <= min(31, pts[:, 1].max())
        # a fully visible triangle covers at least the pixels strictly inside it (centre sampling)
        if (pts >= 0).all() and (pts < 32).all():
            yy, xx = np.mgrid[0:32, 0:32]
            def e(p, q):
                return (q[0] - p[0]) * (yy - p[1]) - (q[1] - p[1]) * (xx - p[0])
            e0, e1, e2 = e(pts[0], pts[1]), e(pts[1], pts[2]), e(pts[2], pts[0])
            strictly_inside = ((e0 > 0) & (e1 > 0) & (e2 > 0)) | ((e0 < 0) & (e1 < 0) & (e2 < 0))
            assert not (strictly_inside & ~a).any()


def test_g5_call_lists_replay(oracle):
    """the replay half of the live OpenCV check runs everywhere: the reference's recorded call lists painted call by call with the
    oracle's fill give non-trivial images (what cv2 would be compared with, oracle/opencv_check.py)"""
    from oracle import opencv_check
    lists = opencv_check.g5_call_lists()
    assert len(lists) == 4
    name, res, tris, cols = lists[0]
    assert tris.shape[2:] == (3, 2) and tris.dtype == np.int32 and cols.shape[:2] == tris.shape[:2]
    img = np.zeros((res, res, 3), np.float32)
    for t, c in zip(tris[0], cols[0]):
        oracle.fill_convex_poly(img, np.ascontiguousarray(t), tuple(int(v) for v in c))
    assert 0.05 < (img.sum(-1) > 0).mean() < 1.0


@pytest.mark.gpu
def test_live_opencv_cross_check_on_the_gpu_box(oracle):
    """collected by the driver's `-m gpu` run as well: whichever box has OpenCV pins the fill"""
    _live_opencv(oracle)


def test_live_opencv_cross_check(oracle):
    _live_opencv(oracle)


def test_the_live_check_harness_runs(oracle, monkeypatch):
    """oracle/opencv_check.py has never met a real OpenCV (absent here): run its harness against a stand-in `cv2` that paints with the oracle
    itself, so that at least its plumbing (argument forms of the reference's call, the replay of the G5 call lists, shapes) is exercised;
    and against a stand-in that is wrong by one pixel, which it must catch"""
    import sys
    import types
    from oracle import opencv_check

    def make(shift_x):
        fake = types.ModuleType('cv2')
        fake.LINE_AA = 16

        def fillConvexPoly(img, points, color, shift=0, lineType=8):
            assert img.dtype == np.float32 and img.ndim == 3 and points.dtype == np.int32 and points.shape == (3, 2) and shift == 0 and lineType == 16
            pts = np.ascontiguousarray(points + np.array([shift_x, 0], np.int32))
            oracle.fill_convex_poly(img, pts, tuple(color))
            return img
        fake.fillConvexPoly = fillConvexPoly
        return fake
    monkeypatch.setitem(sys.modules, 'cv2', make(0))
    assert opencv_check.cross_check(oracle, n_random=200) > 200
    monkeypatch.setitem(sys.modules, 'cv2', make(1))
    with pytest.raises(AssertionError):
        opencv_check.cross_check(oracle, n_random=200)


def _live_opencv(oracle):
    pytest.importorskip('cv2')
    from oracle import opencv_check
    assert opencv_check.cross_check(oracle) > 2000


def _walk(dx, dy):
    """cv::LineIterator (8-connected, after the left-to-right swap: dx >= 0) from (0, 0): the visited pixels"""
    ady, sy = abs(dy), (1 if dy >= 0 else -1)
    vert = ady > dx
    dmaj, dmin = (ady, dx) if vert else (dx, ady)
    err, plus, minus = dmaj - 2 * dmin, 2 * dmaj, -2 * dmin
    x = y = 0
    out = []
    for _ in range(dmaj + 1):
        out.append((x, y))
        neg = err < 0
        err += minus + (plus if neg else 0)
        if vert:
            y += sy
            x += 1 if neg else 0
        else:
            x += 1
            y += sy if neg else 0
    return out


def test_line_rows_closed_form():
    """The per-row closed form of the Bresenham walk used by the K3 bit-plane kernel (raster.hip: process_batch_bits):
    seen from the top end point, row tau of an edge is one run of pixels given by floor divisions."""
    for dx in range(0, 34):
        for dy in range(-34, 35):
            pix = _walk(dx, dy)
            ady, up = abs(dy), dy < 0
            ytop = -ady if up else 0
            x0 = dx if up else 0
            rows = {}
            for (x, y) in pix:
                rows.setdefault(y - ytop, []).append(x)
            assert sorted(rows) == list(range(ady + 1))
            vert = ady > dx
            if not vert and ady > 0:
                qa, qb = divmod(2 * dx, 2 * ady)
            for tau in range(ady + 1):
                if vert:
                    lo = hi = (2 * dx * tau + ady - (0 if up else 1)) // (2 * ady)
                elif ady == 0:
                    lo, hi = 0, dx
                else:
                    nn = 2 * dx * (tau + 1) + 2 * ady - dx - (1 if up else 0)
                    q2, r2 = divmod(nn, 2 * ady)
                    lo = 0 if tau == 0 else q2 - qa - (1 if r2 < qb else 0)
                    hi = min(q2 - 1, dx)
                xs, xe = (x0 - hi, x0 - lo) if up else (x0 + lo, x0 + hi)
                assert sorted(rows[tau]) == list(range(xs, xe + 1)), (dx, dy, tau)
