"""Static line-segment obstacle maps for the batched LiDAR simulator.

A map is a float32 array ``[S, 4]`` of segments ``(ax, ay, bx, by)`` in world metres; a
per-env map buffer is ``[N, S, 4]``.  Geometry is computed in float64 and rounded once.

Only ``stage_1`` has reference geometry
(``turtlebot3_simulations/turtlebot3_gazebo/worlds/train_world_new.world:85-416``: 4 outer
walls + 4 inner boxes, the obstacles the env's goal-rejection rectangles at
``project_ppo/src/environment_new.py:248-251,340-343`` are drawn around).  ``stage_2`` and
``stage_4`` are AUTHORED here (the reference's ``turtlebot3_stage_2.world`` is referenced by
``launch/turtlebot3_stage_2.launch:8`` but missing from the tree; "stage_4" appears nowhere),
so results on them are "parity unpinned (no reference geometry)".
"""
import math

import numpy as np

# Reference goal-rejection rectangles (xmin, xmax, ymin, ymax), inclusive.
RESET_RECTS_STAGE1 = np.array(  # environment_new.py:340-343
    [[1.7, 2.3, -1.2, 1.2], [-2.3, -1.7, -1.2, 1.2], [-1.2, 1.2, 1.7, 2.3], [-1.2, 1.2, -2.3, -1.7]], dtype=np.float64)
RESPAWN_RECTS_STAGE1 = np.array(  # environment_new.py:248-251
    [[1.6, 2.4, -1.4, 1.4], [-2.4, -1.6, -1.4, 1.4], [-1.4, 1.4, 1.6, 2.4], [-1.4, 1.4, -2.4, -1.6]], dtype=np.float64)


def box_to_segments(length, width, cx, cy, yaw):
    """Footprint of an SDF ``<box><size>length width h</size>`` posed at (cx, cy, yaw): 4 segments, CCW."""
    c, s = math.cos(yaw), math.sin(yaw)
    hx, hy = length / 2.0, width / 2.0
    corners = [(-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)]
    pts = [(cx + c * px - s * py, cy + s * px + c * py) for px, py in corners]
    return [[pts[i][0], pts[i][1], pts[(i + 1) % 4][0], pts[(i + 1) % 4][1]] for i in range(4)]


def cylinder_to_segments(radius, cx, cy, sides=24, phase=0.0):
    """Regular ``sides``-gon inscribed in an SDF ``<cylinder>`` footprint, CCW."""
    pts = [(cx + radius * math.cos(phase + 2 * math.pi * k / sides), cy + radius * math.sin(phase + 2 * math.pi * k / sides))
           for k in range(sides)]
    return [[pts[k][0], pts[k][1], pts[(k + 1) % sides][0], pts[(k + 1) % sides][1]] for k in range(sides)]


def _as_map(segs):
    return np.asarray(segs, dtype=np.float64).astype(np.float32).reshape(-1, 4)


# (length, width, cx, cy, yaw) exactly as written in train_world_new.world (link poses at
# :122,:164,:206,:248,:290,:332,:374,:416; collision sizes at :89,:131,:173,:215,:257,:299,:341,:383)
_STAGE1_BOXES = [
    (8.0, 0.2, 4.0, 0.0, -1.5708),
    (8.0002, 0.2, 0.0, -4.0, 3.14159),
    (8.0, 0.2, -4.0, 0.0, 1.5708),
    (8.0, 0.2, 0.0, 4.0, 0.0),
    (2.0, 0.2, 2.0, 0.0, -1.5708),
    (2.0, 0.2, 0.0, -2.0, 3.14159),
    (2.0, 0.2, -2.0, 0.0, 1.5708),
    (2.0, 0.2, 0.0, 2.0, 0.0),
]


def stage_1():
    """8 boxes = 32 segments (reference geometry)."""
    segs = []
    for b in _STAGE1_BOXES:
        segs += box_to_segments(*b)
    return _as_map(segs)


def stage_2(sides=24):
    """AUTHORED: stage_1 shell + four r=0.3 m pillars at (+-2.6, +-2.6); 32 + 4*sides segments (128 by default)."""
    segs = []
    for b in _STAGE1_BOXES:
        segs += box_to_segments(*b)
    for sx in (1, -1):
        for sy in (1, -1):
            segs += cylinder_to_segments(0.3, 2.6 * sx, 2.6 * sy, sides=sides, phase=math.pi / sides)
    return _as_map(segs)


STAGE2_EXTRA_RECTS = np.array([[sx * 2.6 - 0.6, sx * 2.6 + 0.6, sy * 2.6 - 0.6, sy * 2.6 + 0.6]
                               for sx in (1, -1) for sy in (1, -1)], dtype=np.float64)


def stage_4():
    """AUTHORED: stage_1 shell + a pinwheel of four 1.6 m walls and four corner posts; 64 segments."""
    segs = []
    for b in _STAGE1_BOXES:
        segs += box_to_segments(*b)
    for k in range(4):
        a = k * math.pi / 2
        cx, cy = 2.9 * math.cos(a + math.pi / 4), 2.9 * math.sin(a + math.pi / 4)
        segs += box_to_segments(1.6, 0.15, cx, cy, a + 3 * math.pi / 4)
        segs += box_to_segments(0.3, 0.3, 3.3 * math.cos(a + 0.35), 3.3 * math.sin(a + 0.35), a)
    return _as_map(segs)


STAGE4_EXTRA_RECTS = np.array(
    [[2.05 * sx - 1.0, 2.05 * sx + 1.0, 2.05 * sy - 1.0, 2.05 * sy + 1.0] for sx in (1, -1) for sy in (1, -1)],
    dtype=np.float64)


def house_base():
    """The reference's ``models/turtlebot3_house/model.sdf`` cut at the LiDAR plane: 52 boxes + 4 cylinders (12-gons) = 256
    segments, produced by ``tools/gen_house_map.py`` with ``sdf_ingest`` (4 mesh collisions have no footprint and are
    skipped).  15 x 10.7 m."""
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "turtlebot3_house_segments.npy"))


def house(n_segments=2048, seed=0, keep_clear=None, clear_radius=0.6):
    """AUTHORED "small_house ~2k segments" map of BASELINE cfg 5 (``aws_robomaker_small_house_world`` is not vendored in
    the reference, ``navbot_small_house.launch:11``): ``house_base()`` furnished with seeded clutter -- thin table/chair
    legs (5 cm boxes) and round stools/plant pots (16-gons) -- until the map has exactly ``n_segments`` segments.  Clutter
    keeps ``clear_radius`` metres away from every point of ``keep_clear`` (default: the small_house start/goal tables)."""
    base = house_base()
    if keep_clear is None:
        keep_clear = np.concatenate([np.array(_HOUSE_STARTS)[:, :2], np.array(_HOUSE_GOALS)])
    rng = np.random.default_rng(seed)
    segs = [base.astype(np.float64)]
    n = base.shape[0]
    xmin, xmax, ymin, ymax = -7.3, 7.3, -5.1, 5.1
    placed = []
    while n < n_segments:
        left = n_segments - n
        x, y = rng.uniform(xmin, xmax), rng.uniform(ymin, ymax)
        if np.min(np.hypot(keep_clear[:, 0] - x, keep_clear[:, 1] - y)) < clear_radius:
            continue
        if placed and np.min(np.hypot(np.array(placed)[:, 0] - x, np.array(placed)[:, 1] - y)) < 0.25:
            continue
        if left >= 16 and rng.random() < 0.35:
            segs.append(np.array(cylinder_to_segments(rng.uniform(0.1, 0.2), x, y, sides=16, phase=rng.uniform(0, 1))))
            n += 16
        elif left >= 4:
            segs.append(np.array(box_to_segments(0.05, 0.05, x, y, rng.uniform(0, math.pi / 2))))
            n += 4
        else:  # 1..3 segments left: a short free-standing panel per segment
            a = rng.uniform(0, math.pi)
            segs.append(np.array([[x, y, x + 0.3 * math.cos(a), y + 0.3 * math.sin(a)]]))
            n += 1
        placed.append((x, y))
    return _as_map(np.concatenate(segs))


def open_tables(seg, starts, goals, n_beams=10, open_thresh=0.4):
    """Keeps the start poses / goal points whose surroundings are open in map ``seg`` (validate_open_space rule,
    spawn_goal_sampler.py:64-72) using a float64 numpy ray-caster (host-side tooling, not the simulation path)."""
    seg = np.asarray(seg, dtype=np.float64)

    def min_range(x, y):
        ang = np.linspace(-math.pi, math.pi, 72, endpoint=False)
        dx, dy = np.cos(ang)[:, None], np.sin(ang)[:, None]
        ax, ay, bx, by = seg.T
        ex, ey = bx - ax, by - ay
        den = dx * ey - dy * ex
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((ax - x) * ey - (ay - y) * ex) / den
            u = ((ax - x) * dy - (ay - y) * dx) / den
        t = np.where((den != 0) & (t >= 0) & (u >= 0) & (u <= 1), t, np.inf)
        return t.min()

    ks = [k for k, p in enumerate(starts) if min_range(p[0], p[1]) > open_thresh]
    kg = [k for k, p in enumerate(goals) if min_range(p[0], p[1]) > open_thresh]
    return np.asarray(starts)[ks], np.asarray(goals)[kg]


def by_name(name):
    table = {"stage_1": stage_1, "stage_2": stage_2, "stage_4": stage_4, "house": house, "house_base": house_base}
    if name not in table:
        raise KeyError(f"unknown map {name!r}; have {sorted(table)}")
    return table[name]()


def goal_rects(name):
    """(reset_rects, respawn_rects) for a named map."""
    if name == "stage_1":
        return RESET_RECTS_STAGE1, RESPAWN_RECTS_STAGE1
    if name == "stage_2":
        return (np.concatenate([RESET_RECTS_STAGE1, STAGE2_EXTRA_RECTS]),
                np.concatenate([RESPAWN_RECTS_STAGE1, STAGE2_EXTRA_RECTS]))
    if name == "stage_4":
        return (np.concatenate([RESET_RECTS_STAGE1, STAGE4_EXTRA_RECTS]),
                np.concatenate([RESPAWN_RECTS_STAGE1, STAGE4_EXTRA_RECTS]))
    if name in ("house", "house_base"):  # goals come from the curated tables (sampler), not from the uniform box
        return np.zeros((0, 4)), np.zeros((0, 4))
    raise KeyError(name)


def replicate_per_env(seg, n_envs, seed=0, jitter=0.02, shuffle=True):
    """Per-env segment buffers ``[N, S, 4]`` (BASELINE cfg 3: "per-env pose/segment buffers").

    Every env gets its own copy of the map with a small random rigid offset (|d| <= jitter
    metres, so the spawn pose and goal box stay valid) and, optionally, its own segment order,
    so that per-env loads are real and no two envs read identical bytes.
    """
    rng = np.random.default_rng(seed)
    seg = np.asarray(seg, dtype=np.float32).reshape(-1, 4)
    S = seg.shape[0]
    out = np.empty((n_envs, S, 4), dtype=np.float32)
    off = rng.uniform(-jitter, jitter, size=(n_envs, 2))
    for i in range(n_envs):
        order = rng.permutation(S) if shuffle else np.arange(S)
        s64 = seg[order].astype(np.float64)
        s64[:, [0, 2]] += off[i, 0]
        s64[:, [1, 3]] += off[i, 1]
        out[i] = s64.astype(np.float32)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Curated start poses (x, y, yaw) and goal points of the reference's GoalSpawnSampler
# (project_ppo/src/spawn_goal_sampler.py:5-35).  These are scenario DATA of the reference (hand-picked free-space
# coordinates for its two worlds), reproduced so that `--use_external_sampler` runs are drop-in; the sampling rule
# itself is re-implemented in the step/reset kernels (navsim_set_spawn_sampler).
_STAGE1_STARTS = [(0.0, 0.0, 0.0), (0.5, 0.5, 0.785), (-0.5, 0.5, 2.356), (0.5, -0.5, -0.785), (-0.5, -0.5, -2.356),
                  (1.0, 0.0, 0.0), (0.0, 1.0, 1.57), (-1.0, 0.0, 3.14), (0.0, -1.0, -1.57), (1.0, 1.0, 0.785)]
_STAGE1_GOALS = [(3.0, 3.0), (3.5, 2.5), (2.5, 3.5), (4.0, 3.0), (-3.0, 3.0), (-3.5, 2.5), (-2.5, 3.5), (-4.0, 3.0),
                 (3.0, -3.0), (3.5, -2.5), (2.5, -3.5), (4.0, -3.0), (-3.0, -3.0), (-3.5, -2.5), (-2.5, -3.5), (-4.0, -3.0),
                 (4.0, 0.0), (-4.0, 0.0), (0.0, 4.0), (0.0, -4.0), (3.0, 0.0), (-3.0, 0.0), (0.0, 3.0), (0.0, -3.0),
                 (2.0, 2.0), (-2.0, 2.0), (2.0, -2.0), (-2.0, -2.0)]
_HOUSE_STARTS = [(-3.5, 1.0, 0.0), (-3.0, 0.5, 1.57), (-2.5, 1.5, -1.57), (-3.0, 2.0, 0.0), (-1.0, 0.0, 0.0), (-0.5, 0.5, 1.57),
                 (0.0, 0.0, -1.57), (2.0, 1.5, 3.14), (2.5, 0.5, -1.57), (3.0, 1.0, 0.0), (1.0, -2.0, 1.57), (0.5, -2.5, 0.0),
                 (1.5, -2.0, -1.57), (-1.5, 2.5, 0.0), (-2.0, 3.0, 1.57), (0.0, 1.0, 0.0), (-1.0, 1.5, 1.57), (1.0, 1.0, -1.57)]
_HOUSE_GOALS = [(-3.5, 0.5), (-3.0, 1.5), (-2.5, 2.0), (-3.5, 2.5), (-4.0, 1.0), (-2.0, 1.0), (-3.0, 0.0), (-1.0, 0.5),
                (-0.5, 0.0), (0.0, 0.5), (-1.5, 0.0), (0.5, 0.0), (-1.0, -0.5), (2.0, 0.5), (2.5, 1.0), (3.0, 1.5), (2.0, 2.0),
                (3.5, 1.0), (2.5, 0.0), (3.0, 0.5), (1.0, -2.5), (0.5, -2.0), (1.5, -2.5), (1.0, -3.0), (0.0, -2.5), (1.5, -1.5),
                (-1.5, 2.0), (-2.0, 2.5), (-1.0, 3.0), (-2.5, 2.5), (-1.5, 3.5), (0.0, 1.5), (-1.0, 1.0), (1.0, 0.5), (0.5, 1.5),
                (-0.5, 1.0), (0.0, 2.0), (1.0, 1.5), (-4.0, 3.0), (3.5, 2.0), (2.0, -3.0), (-2.0, -1.0)]


def spawn_tables(world_type, min_dist=1.5, max_dist=6.0):
    """(starts [K,3], goals [G,2], min_dist, max_dist) as GoalSpawnSampler(world_type) holds them
    (spawn_goal_sampler.py:38-50; defaults min_dist=1.5, max_dist=6.0 at :38)."""
    if world_type == "stage1":
        st, g = _STAGE1_STARTS, _STAGE1_GOALS
    elif world_type == "small_house":
        st, g = _HOUSE_STARTS, _HOUSE_GOALS
    else:
        raise ValueError(f"Unknown world_type: {world_type}")  # same error as spawn_goal_sampler.py:49
    return np.array(st, dtype=np.float64), np.array(g, dtype=np.float64), float(min_dist), float(max_dist)


def validate_open_space(ranges, range_min=0.12, range_max=3.5, open_thresh=0.4):
    """GoalSpawnSampler.validate_open_space (spawn_goal_sampler.py:64-72) on a raw scan array [..., B]: the smallest
    range strictly inside (range_min, range_max) must exceed open_thresh; no valid range -> False."""
    r = np.asarray(ranges, dtype=np.float64)
    valid = (r > range_min) & (r < range_max)
    mn = np.where(valid, r, np.inf).min(axis=-1)
    return valid.any(axis=-1) & (mn > open_thresh)
