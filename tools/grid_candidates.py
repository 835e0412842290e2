#!/usr/bin/env python3
"""How many grid entries the rasteriser's scan offers per view (CPU, numpy): the map's one grid (every face in all the cells its bounding box
touches, per-row cell ranges under the rotated view) against a two-level variant (small faces in ONE cell of a four times finer grid scanned with
the window grown by the cell size).  DESIGN_HISTORY.md section 4, "Two rendering grids": 1 752 against 1 375 candidates for 1 210 accepted faces.
   python tools/grid_candidates.py"""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
v,f,vc,cats = bench.load_town01()
P = v[f]                                  # F,3,2
bmin, bmax = P.min(1), P.max(1)
ox, oy = v[:,0].min(), v[:,1].min()
rng = np.random.default_rng(0)
road = v[vc == cats.index('road')]
def rows_ranges(Q, cell, ext):
    """cells (cx range per cy) under polygon Q (4,2) for min-corner binning with expansion ext"""
    out = []
    cy0 = int(np.floor((Q[:,1].min() - ext - oy)/cell)); cy1 = int(np.floor((Q[:,1].max() - oy)/cell))
    for cy in range(cy0, cy1+1):
        ya, yb = oy + cy*cell, oy + (cy+1)*cell + ext
        xs = []
        for k in range(4):
            x0,y0 = Q[k]; x1,y1 = Q[(k+1)%4]
            if ya <= y0 <= yb: xs.append(x0)
            for yl in (ya, yb):
                if (y0-yl)*(y1-yl) <= 0 and y1 != y0: xs.append(x0 + (yl-y0)/(y1-y0)*(x1-x0))
        if xs:
            out.append((cy, int(np.floor((min(xs) - ext - ox)/cell)), int(np.floor((max(xs) - ox)/cell))))
    return out
def count(cell, single, sel, ext):
    # entries per cell
    if single:
        cx = np.floor((bmin[sel,0]-ox)/cell).astype(int); cy = np.floor((bmin[sel,1]-oy)/cell).astype(int)
        grid = {}
        for a,b in zip(cx,cy): grid[(a,b)] = grid.get((a,b),0)+1
    else:
        grid = {}
        cx0 = np.floor((bmin[sel,0]-ox)/cell).astype(int); cx1 = np.floor((bmax[sel,0]-ox)/cell).astype(int)
        cy0 = np.floor((bmin[sel,1]-oy)/cell).astype(int); cy1 = np.floor((bmax[sel,1]-oy)/cell).astype(int)
        for a0,a1,b0,b1 in zip(cx0,cx1,cy0,cy1):
            for b in range(b0,b1+1):
                for a in range(a0,a1+1): grid[(a,b)] = grid.get((a,b),0)+1
    return grid
fov = 35.0
coarse = 0.65*fov; fine = coarse/4
ext = (bmax-bmin).max(1)
small = ext <= fine*0.999
print('coarse', coarse, 'fine', fine, 'small share', small.mean())
g_all = count(coarse, False, np.ones(len(P),bool), 0)
g_S = count(fine, True, small, fine)
g_L = count(coarse, False, ~small, 0)
tot = dict(all=0, S=0, L=0, rowsS=0, rowsall=0)
N = 300
for _ in range(N):
    c = road[rng.integers(len(road))]; th = rng.uniform(0, 2*np.pi)
    h = fov/2 * (1 + 2/128)
    cor = np.array([[-h,-h],[-h,h],[h,h],[h,-h]])
    R = np.array([[np.cos(th),-np.sin(th)],[np.sin(th),np.cos(th)]])
    Q = cor @ R.T + c
    for name, g, cell, e in (('all', g_all, coarse, 0.0), ('S', g_S, fine, fine), ('L', g_L, coarse, 0.0)):
        rr = rows_ranges(Q, cell, e)
        for cy, a, b in rr:
            for cx in range(a, b+1): tot[name] += g.get((cx,cy),0)
        if name == 'S': tot['rowsS'] += len(rr)
        if name == 'all': tot['rowsall'] += len(rr)
for k in tot: print(k, tot[k]/N)
