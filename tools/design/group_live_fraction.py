#!/usr/bin/env python3
"""Design aid (CPU): what fraction of a per-env map's segment GROUPS survives a conservative range / behind-the-fan test at the
poses a configs[2] run visits?  Groups = runs of G consecutive segments after a spatial sort.  Runs here (numpy + the oracle)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from navbot_ppo_amd import maps
from oracle import navsim_oracle as O

def poses(N=256, T=300, seed=0):
    sim = O.OracleSim(N, max_episode_steps=500, auto_reset=True, seed=seed)
    seg = maps.stage_2(); sim.set_map(seg)
    rr, rs = maps.goal_rects("stage_2"); sim.set_goal_rects(0, rr); sim.set_goal_rects(1, rs)
    sim.reset()
    rng = np.random.default_rng(seed); out = []
    for t in range(T):
        a = np.stack([rng.uniform(0, 1, N), rng.uniform(-1, 1, N)], 1).astype(np.float32)
        sim.step(a)
        if t % 10 == 9: out.append(sim.get_state()["pose"].copy())
    return np.concatenate(out)

def morton(seg):
    mx, my = 0.5 * (seg[:, 0] + seg[:, 2]), 0.5 * (seg[:, 1] + seg[:, 3])
    def sp(v):
        v = v.astype(np.uint32) & 0xFFFF
        v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    sx = (mx - mx.min()) / max(mx.max() - mx.min(), 1e-9); sy = (my - my.min()) / max(my.max() - my.min(), 1e-9)
    return sp(sx * 65535) | (sp(sy * 65535) << 1)

def order_len_then_morton(seg, long_thr=1.0):
    """long segments first (their boxes are huge: keep them together), the rest in Morton order"""
    ln = np.hypot(seg[:, 2] - seg[:, 0], seg[:, 3] - seg[:, 1])
    key = morton(seg).astype(np.int64)
    key = np.where(ln > long_thr, -1 - np.argsort(np.argsort(-ln)), key)
    return np.argsort(key, kind="stable")

def boxes(seg, G):
    S = seg.shape[0]; nb = (S + G - 1) // G; b = np.empty((nb, 4))
    for k in range(nb):
        g = seg[k * G:(k + 1) * G]
        b[k] = [min(g[:, 0].min(), g[:, 2].min()), min(g[:, 1].min(), g[:, 3].min()), max(g[:, 0].max(), g[:, 2].max()), max(g[:, 1].max(), g[:, 3].max())]
    return b

def live(b, P):
    """[npose, nbox] conservative: not (farther than 3.507 m or all four corners behind the fan)"""
    ox = P[:, 0] - 0.032 * np.cos(P[:, 2]); oy = P[:, 1] - 0.032 * np.sin(P[:, 2]); c, s = np.cos(P[:, 2]), np.sin(P[:, 2])
    dx = np.maximum(np.maximum(b[None, :, 0] - ox[:, None], ox[:, None] - b[None, :, 2]), 0)
    dy = np.maximum(np.maximum(b[None, :, 1] - oy[:, None], oy[:, None] - b[None, :, 3]), 0)
    far = dx * dx + dy * dy > 12.3
    back = np.ones_like(far)
    for xi in (0, 2):
        for yi in (1, 3):
            cx = b[None, :, xi] - ox[:, None]; cy = b[None, :, yi] - oy[:, None]
            X = cx * c[:, None] + cy * s[:, None]; Y = cy * c[:, None] - cx * s[:, None]
            back &= (X + 1e-3 * np.abs(Y) < 0)
    return ~(far | back)

def seg_live(seg, P):
    """per-segment stage A (exact distance + both endpoints behind), [npose, S]"""
    ox = P[:, 0] - 0.032 * np.cos(P[:, 2]); oy = P[:, 1] - 0.032 * np.sin(P[:, 2]); c, s = np.cos(P[:, 2]), np.sin(P[:, 2])
    def fr(px, py):
        cx = px[None] - ox[:, None]; cy = py[None] - oy[:, None]
        return cx * c[:, None] + cy * s[:, None], cy * c[:, None] - cx * s[:, None]
    xa, ya = fr(seg[:, 0], seg[:, 1]); xb, yb = fr(seg[:, 2], seg[:, 3])
    behind = (xa + 1e-3 * np.abs(ya) < 0) & (xb + 1e-3 * np.abs(yb) < 0)
    ex, ey = xb - xa, yb - ya; t = np.clip(-(xa * ex + ya * ey) / (ex * ex + ey * ey), 0, 1)
    d2 = (xa + t * ex) ** 2 + (ya + t * ey) ** 2
    return ~(behind | (d2 > 12.3))

if __name__ == "__main__":
    P = poses()
    print("poses", P.shape, "mean |pos|", np.hypot(P[:, 0], P[:, 1]).mean())
    names = {"stage_2": maps.stage_2(), "stage_2 S=1024": maps.stage_2(sides=248), "stage_4": maps.stage_4(), "house": maps.house(2048)}
    for name, seg in names.items():
        seg = seg.astype(np.float64)
        if name == "house": Pp = np.array(maps._HOUSE_STARTS)[np.random.default_rng(0).integers(0, 18, 2000)] + np.random.default_rng(1).normal(0, 0.3, (2000, 3)) * [1, 1, 3]
        else: Pp = P
        sl = seg_live(seg, Pp).mean()
        print(f"{name}: S={seg.shape[0]} per-segment stage-A survival {sl:.3f}")
        for oname, order in (("morton", np.argsort(morton(seg), kind="stable")), ("long-first+morton", order_len_then_morton(seg)), ("as given", np.arange(seg.shape[0]))):
            for G in (4, 8, 16, 32):
                b = boxes(seg[order], G); f = live(b, Pp).mean()
                print(f"   {oname:18s} G={G:2d}: {b.shape[0]:4d} groups, live fraction {f:.3f}")
