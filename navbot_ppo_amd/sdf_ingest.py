"""SDF world/model -> line-segment map ("map ingest").

The reference's obstacle geometry lives in Gazebo SDF files (``turtlebot3_gazebo/worlds/*.world``,
``models/*/model.sdf``).  The LiDAR of the burger scans a horizontal plane 0.172 m above ``base_link``
(``turtlebot3_burger.urdf.xacro:134-138``), so a static world reduces to the 2-D footprints of the collision
shapes that cross that plane: ``<box>`` -> 4 segments, ``<cylinder>`` -> a regular polygon.  Nested
``<model>``/``<link>``/``<collision>`` poses are composed (planar part: x, y, yaw); ``<mesh>`` collisions have no
analytic footprint and are skipped (reported in the returned stats).

    segs, stats = sdf_to_segments("model.sdf", scan_z=0.182, cylinder_sides=12)
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

from .maps import box_to_segments, cylinder_to_segments


def _pose(el):
    p = el.find("pose") if el is not None else None
    v = [float(t) for t in p.text.split()] if p is not None and p.text else [0.0] * 6
    v += [0.0] * (6 - len(v))
    return v


def _compose(parent, child):
    """Planar composition of SDF poses (x, y, z, roll, pitch, yaw): child expressed in the parent's frame."""
    px, py, pz, _, _, pyaw = parent
    cx, cy, cz, cr, cp, cyaw = child
    c, s = math.cos(pyaw), math.sin(pyaw)
    return [px + c * cx - s * cy, py + s * cx + c * cy, pz + cz, cr, cp, pyaw + cyaw]


def sdf_to_segments(path_or_text, scan_z=0.182, cylinder_sides=12, include_visual_only=False):
    """Returns (segments float32 [S,4], stats dict)."""
    text = open(path_or_text).read() if "<" not in path_or_text else path_or_text
    root = ET.fromstring(text)
    segs = []
    stats = dict(boxes=0, cylinders=0, meshes_skipped=0, out_of_plane=0, tilted_skipped=0)

    def visit_link(link, base):
        lp = _compose(base, _pose(link))
        shapes = link.findall("collision")
        if include_visual_only and not shapes:
            shapes = link.findall("visual")
        for col in shapes:
            cp = _compose(lp, _pose(col))
            geo = col.find("geometry")
            if geo is None:
                continue
            if abs(cp[3]) > 1e-3 or abs(cp[4]) > 1e-3:
                stats["tilted_skipped"] += 1
                continue
            box, cyl = geo.find("box"), geo.find("cylinder")
            if box is not None:
                sx, sy, sz = [float(t) for t in box.find("size").text.split()]
                if not (cp[2] - sz / 2 <= scan_z <= cp[2] + sz / 2):
                    stats["out_of_plane"] += 1
                    continue
                segs.extend(box_to_segments(sx, sy, cp[0], cp[1], cp[5]))
                stats["boxes"] += 1
            elif cyl is not None:
                r = float(cyl.find("radius").text)
                ln = float(cyl.find("length").text)
                if not (cp[2] - ln / 2 <= scan_z <= cp[2] + ln / 2):
                    stats["out_of_plane"] += 1
                    continue
                segs.extend(cylinder_to_segments(r, cp[0], cp[1], sides=cylinder_sides, phase=cp[5]))
                stats["cylinders"] += 1
            elif geo.find("mesh") is not None:
                stats["meshes_skipped"] += 1

    def visit_model(model, base):
        mp = _compose(base, _pose(model))
        for link in model.findall("link"):
            visit_link(link, mp)
        for sub in model.findall("model"):
            visit_model(sub, mp)

    tops = root.findall("model") + [m for w in root.findall("world") for m in w.findall("model")]
    for m in tops:
        if m.get("name") in ("ground_plane",):
            continue
        visit_model(m, [0.0] * 6)
    arr = np.asarray(segs, dtype=np.float64).astype(np.float32).reshape(-1, 4)
    stats["segments"] = int(arr.shape[0])
    return arr, stats
