"""
The PAIR rule -- two same-key triangles that share an edge (a triangulated quad of the road / lane-marking mesh) can have their interior rows
painted once, as the hull of the two triangles' row intervals -- as sequential C (tests/fill_quads_model.c), checked pair by pair against the
oracle's cv::fillConvexPoly restatement called once per triangle (rendering/cv2.py:44-59 of the reference: one cv2.fillConvexPoly per face).
Round 5 built the rule into the bit-plane kernel (bit-exact) and took it out again: slower than triangle by triangle (DESIGN_HISTORY.md section 4).
The rendering grid still PAIRS the faces (map.hip) and the scan kernel handles a pair at once; the model documents what a paired
rasteriser has to do.  CPU only.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'fill_quads_model.c')
OUT = os.path.join(ROOT, 'tests', '_build', 'fill_quads_model')


@pytest.fixture(scope='module')
def model(oracle):
    libdir = os.path.join(ROOT, 'oracle', '_build')
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(['gcc', '-O2', '-Wall', '-o', OUT, SRC, '-L', libdir, '-ltds_oracle', f'-Wl,-rpath,{libdir}', '-fopenmp'], check=True)
    return OUT


def run(model, *args):
    r = subprocess.run([model, *map(str, args)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_every_pair_of_a_small_grid(model):
    # the four points anywhere on a 5 x 5 grid that overhangs a 3 x 3 image by one pixel on every side, every choice of the shared edge and every
    # vertex order of the second triangle: 5^8 x 18 pairs -- coinciding points, collinear points, flat tops and bottoms, pairs folded over (both
    # apexes on one side of the shared edge), pairs that leave the image
    out = run(model, 'exhaustive', 3, -1, 3)
    assert '7031250 pairs, 0 differ' in out
    assert '7031250 pairs, 0 differ' in run(model, 'kernel-exhaustive', 3, -1, 3)


@pytest.mark.parametrize('res,count,seed', [(64, 200000, 21), (256, 200000, 22), (512, 80000, 23)])
def test_random_pairs(model, res, count, seed):
    # (also in the two-lane organisation, 'kernel-' modes)
    # lane-marking slivers and road quads at every rotation, with jitter; across the image border; small quads; degenerate ones; edges beyond the
    # merge limits (walked exactly)
    assert f'{count} pairs, 0 differ' in run(model, 'random', res, count, seed)
    assert f'{count // 4} pairs, 0 differ' in run(model, 'kernel-random', res, count // 4, seed + 100)


@pytest.mark.parametrize('mode', ['always', 'noapex'])
def test_the_check_can_fail(model, mode):
    # always-: pairs whose apexes lie on one side of the shared edge painted as ONE hull (the gap between the two triangles is filled);
    # noapex-: a triangle's own interval in the row of the other one's apex left out.  The harness must notice both.
    r = subprocess.run([model, f'{mode}-random', '64', '100000', '21'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and ' 0 differ' not in r.stdout
