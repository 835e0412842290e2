#!/usr/bin/env python3
"""
Copies the MAP DATA files the lane-map tests use from the reference checkout into tests/golden/ (run in the build container,
where /root/reference exists; the GPU box only sees the copies):
  carla_Town01.osm, carla_Town02.osm -> tests/golden/<name>.osm.gz + town02_mesh.npz (Town02's shipped mesh as arrays)
  carla_Town01.osm                  -> tests/golden/carla_Town01.osm.gz   (the Lanelet2 map whose mesh is tests/golden/town01_mesh.npz;
                                       torchdrivesim/resources/maps/carla_Town01/, origin (0, 0) per its metadata.json)
  tests/resources/testing_lanelet2map.osm -> tests/golden/testing_lanelet2map.osm   (the map of the reference's simulator tests)
These are data (OSM XML: nodes, ways, relations), not code.
"""
import gzip
import os
import shutil

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')

for town in ('carla_Town01', 'carla_Town02'):
    with open(os.path.join(REF, f'torchdrivesim/resources/maps/{town}/{town}.osm'), 'rb') as f:
        data = f.read()
    with open(os.path.join(OUT, f'{town}.osm.gz'), 'wb') as raw:
        with gzip.GzipFile(filename=f'{town}.osm', mode='wb', fileobj=raw, mtime=0) as g:      # mtime=0: reproducible bytes
            g.write(data)

# Town02's shipped mesh (generated upstream from the .osm above with the real Lanelet2) as arrays, like town01_mesh.npz
import json
import numpy as np
m = json.load(open(os.path.join(REF, 'torchdrivesim/resources/maps/carla_Town02/carla_Town02_mesh.json')))
np.savez_compressed(os.path.join(OUT, 'town02_mesh.npz'), verts=np.array(m['verts'][0], np.float32), faces=np.array(m['faces'][0], np.int32),
                    vert_category=np.array(m['vert_category'][0], np.uint8), categories=np.array(m['categories']))
shutil.copyfile(os.path.join(REF, 'tests/resources/testing_lanelet2map.osm'), os.path.join(OUT, 'testing_lanelet2map.osm'))
print('written:', sorted(p for p in os.listdir(OUT) if 'osm' in p))
