"""Host-side map logic on CPU: SDF ingest, the house map, per-env replication, sampler tables."""
import math
import os

import numpy as np
import pytest

from navbot_ppo_amd import maps, sdf_ingest

SDF = """<?xml version='1.0'?><sdf version='1.6'><world name='w'>
<model name='ground_plane'><link name='l'><collision name='c'><geometry><plane/></geometry></collision></link></model>
<model name='m'><pose>1 2 0 0 0 1.5707963267948966</pose>
  <link name='a'><pose>1 0 0 0 0 0</pose>
    <collision name='c'><pose>0 0 0.4 0 0 0</pose><geometry><box><size>2 0.2 0.8</size></box></geometry></collision></link>
  <link name='b'><pose>0 0 0 0 0 0</pose>
    <collision name='c'><pose>0 1 0.5 0 0 0</pose><geometry><cylinder><radius>0.5</radius><length>1</length></cylinder></geometry></collision></link>
  <link name='roof'><collision name='c'><pose>0 0 2.5 0 0 0</pose><geometry><box><size>9 9 0.1</size></box></geometry></collision></link>
  <link name='mesh'><collision name='c'><geometry><mesh><uri>x.dae</uri></mesh></geometry></collision></link>
  <model name='nested'><pose>0 0 0 0 0 -1.5707963267948966</pose>
    <link name='n'><collision name='c'><pose>3 0 0.2 0 0 0</pose><geometry><box><size>1 1 0.4</size></box></geometry></collision></link></model>
</model></world></sdf>"""


def test_sdf_ingest_composes_poses_and_filters_the_scan_plane():
    seg, st = sdf_ingest.sdf_to_segments(SDF, scan_z=0.182, cylinder_sides=8)
    assert st == dict(boxes=2, cylinders=1, meshes_skipped=1, out_of_plane=1, tilted_skipped=0, segments=16)
    # link a: box centre (1,0) in a model at (1,2) rotated +90 deg -> world (1, 3), long axis along y
    a = seg[:4]
    assert np.allclose(a[:, [0, 2]].min(), 0.9, atol=1e-6) and np.allclose(a[:, [0, 2]].max(), 1.1, atol=1e-6)
    assert np.allclose(a[:, [1, 3]].min(), 2.0, atol=1e-6) and np.allclose(a[:, [1, 3]].max(), 4.0, atol=1e-6)
    # cylinder at model-frame (0,1) -> world (0, 2), radius 0.5
    c = seg[4:12]
    assert np.allclose(np.hypot(c[:, 0] - 0.0, c[:, 1] - 2.0), 0.5, atol=1e-6)
    # nested model: origin (1,2), yaw +90 - 90 = 0 -> its box at nested-frame (3,0) sits at world (4, 2), axis-aligned
    n = seg[12:]
    assert np.allclose([n[:, [0, 2]].mean(), n[:, [1, 3]].mean()], [4.0, 2.0], atol=1e-6)
    assert np.allclose([n[:, [0, 2]].min(), n[:, [0, 2]].max()], [3.5, 4.5], atol=1e-6)


def test_house_maps():
    hb = maps.house_base()
    assert hb.shape == (256, 4) and hb.dtype == np.float32
    assert -7.7 < hb[:, [0, 2]].min() and hb[:, [0, 2]].max() < 7.7
    h = maps.house()
    assert h.shape == (2048, 4) and np.array_equal(h[:256], hb) and np.array_equal(h, maps.house())  # seeded
    assert maps.house(n_segments=300, seed=1).shape == (300, 4)
    st, g, lo, hi = maps.spawn_tables("small_house")
    so, go = maps.open_tables(h, st, g)
    assert len(so) >= 8 and len(go) >= 20
    # clutter keeps away from every curated point
    mid = np.stack([(h[256:, 0] + h[256:, 2]) / 2, (h[256:, 1] + h[256:, 3]) / 2], 1)
    pts = np.concatenate([st[:, :2], g])
    assert np.min(np.hypot(mid[:, None, 0] - pts[None, :, 0], mid[:, None, 1] - pts[None, :, 1])) > 0.35
    ref_sdf = "/root/reference/turtlebot3_simulations/turtlebot3_gazebo/models/turtlebot3_house/model.sdf"
    if os.path.exists(ref_sdf):  # the committed asset is what the ingest produces from the reference's SDF
        seg, stt = sdf_ingest.sdf_to_segments(ref_sdf)
        assert np.array_equal(seg, hb) and stt["boxes"] == 52 and stt["cylinders"] == 4
        w, _ = sdf_ingest.sdf_to_segments("/root/reference/turtlebot3_simulations/turtlebot3_gazebo/worlds/train_world_new.world")
        key = lambda a: np.sort(a.view([("", a.dtype)] * 4).ravel())
        assert np.array_equal(key(w), key(maps.stage_1()))  # the hand-typed stage_1 equals the ingested world file


def test_replicate_per_env():
    seg = maps.stage_2()
    assert seg.shape == (128, 4)
    p = maps.replicate_per_env(seg, 6, seed=3)
    assert p.shape == (6, 128, 4) and not np.array_equal(p[0], p[1])
    for i in range(6):  # each copy is a rigid translate (|d| <= 2 cm) of a permutation of the map
        d = np.sort(p[i, :, 0]) - np.sort(seg[:, 0])
        assert np.allclose(d, d[0], atol=1e-6) and abs(d[0]) <= 0.02 + 1e-6
    q = maps.replicate_per_env(seg, 2, shuffle=False, jitter=0.0)
    assert np.array_equal(q[0], seg)


def test_named_maps_and_rects():
    for name, S in (("stage_1", 32), ("stage_2", 128), ("stage_4", 64), ("house", 2048)):
        assert maps.by_name(name).shape == (S, 4)
        rr, rs = maps.goal_rects(name)
        assert rr.shape[1] == 4 and rs.shape[1] == 4
    with pytest.raises(KeyError):
        maps.by_name("stage_9")
