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


def test_live_opencv_cross_check(oracle):
    cv2 = pytest.importorskip('cv2')
    rng = np.random.default_rng(1)
    for _ in range(2000):
        pts = rng.integers(-300, 364, size=(3, 2)).astype(np.int32)
        ref = np.zeros((64, 64, 3), np.float32)
        ref = cv2.fillConvexPoly(img=ref, points=pts, color=[7, 8, 9], shift=0, lineType=cv2.LINE_AA)
        np.testing.assert_array_equal(fill(oracle, pts, res=64, color=(7, 8, 9)), ref)
