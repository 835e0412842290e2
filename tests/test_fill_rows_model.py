"""
The row rule of the K3 bit-plane kernel -- outline edges merged into the scan-converted rows (raster.hip: process_batch_bits) -- as
sequential C (tests/fill_rows_model.c), checked triangle by triangle against the oracle's cv::fillConvexPoly restatement
(rendering/cv2.py:59 of the reference is the call being reproduced).  CPU only; the kernel itself is compared with the oracle under -m gpu.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'fill_rows_model.c')
OUT = os.path.join(ROOT, 'tests', '_build', 'fill_rows_model')


@pytest.fixture(scope='module')
def model(oracle):
    libdir = os.path.join(ROOT, 'oracle', '_build')
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(['gcc', '-O2', '-Wall', '-o', OUT, SRC, '-L', libdir, '-ltds_oracle', f'-Wl,-rpath,{libdir}', '-fopenmp'], check=True)
    return OUT


def run(model, *args):
    r = subprocess.run([model, *map(str, args)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_every_triangle_of_a_small_grid(model):
    # all 7^6 vertex triples of a 7 x 7 grid that overhangs a 5 x 5 image by one pixel on every side: every vertex order, every flat,
    # degenerate and clipped configuration at that size
    assert '117649 triangles, 0 differ' in run(model, 'exhaustive', 5, -1, 5)


@pytest.mark.parametrize('res,count,seed', [(64, 150000, 11), (256, 150000, 12), (512, 60000, 13)])
def test_random_triangles(model, res, count, seed):
    # slivers, small faces, flat tops and bottoms, faces that leave the image, edges longer than the merge limits (512: |dy| up to 767)
    assert f'{count} triangles, 0 differ' in run(model, 'random', res, count, seed)


def test_the_check_can_fail(model):
    # without the bias the tie rows of y-major edges and the integer half-row positions of x-major edges go wrong: the harness must say so
    r = subprocess.run([model, 'nobias-random', '256', '100000', '12'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and ' 0 differ' not in r.stdout
