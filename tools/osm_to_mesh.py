#!/usr/bin/env python3
"""
Lanelet2 `.osm` -> the BirdviewMesh JSON the simulator loads (the reference's examples/lanelet2_to_birdview_mesh.py without Lanelet2 or
OmegaConf): road surface from the triangulated lanelets + lane markings.

    python tools/osm_to_mesh.py map.osm[.gz] out_mesh.json [--origin LAT LON] [--carla]
`--carla`: the map is in CARLA's left-handed frame: turn the lanelets back around after loading and build the markings left-handed, as the
reference's example does (`revert_map`, `left_handed=True`).  Without it the result is what `MapConfig.road_mesh` builds (map.py:62-72).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchdrivesim_amd import lanelet2  # noqa: E402
from torchdrivesim_amd.mesh import BirdviewMesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('osm')
    ap.add_argument('out')
    ap.add_argument('--origin', type=float, nargs=2, default=(0.0, 0.0), metavar=('LAT', 'LON'))
    ap.add_argument('--carla', action='store_true')
    args = ap.parse_args()
    lanes = lanelet2.load_lanelet_map(args.osm, origin=tuple(args.origin))
    if args.carla:
        lanes = lanelet2.revert_map(lanes)
    road = BirdviewMesh.set_properties(lanelet2.road_mesh_from_lanelet_map(lanes), category='road')
    mesh = lanelet2.lanelet_map_to_lane_mesh(lanes, left_handed=args.carla).merge(road)
    mesh.save(args.out)
    print(f'{len(lanes.laneletLayer)} lanelets -> {mesh.verts_count} vertices, {mesh.faces_count} faces, categories {mesh.categories}: {args.out}')


if __name__ == '__main__':
    main()
