"""GPU tests of the split-bf16 update pass ("bf16x3", csrc/ppo_mlp64.hip: pass_body_x3; PPOConfig.update_arith) -- the gates VERDICT
round 4 set for it to replace the f32-MFMA pass on the timed path:
  * every autograd test of the fused 16-64-64 update passes at UNCHANGED tolerances on it (test_gpu_ppo.py runs them on the
    default arithmetic, which is bf16x3; the parametrised copies here run both arithmetics side by side);
  * against a float64 PyTorch reference on the bench's own batch its error is not above the native-f32 path's error;
  * the native f32 path stays selectable (update_arith="f32") and both give the same statistics.
Reference: project_ppo/src/ppo.py:305-397 (losses, backward, Adam), net_actor.py:147-189."""
import copy
import os

import numpy as np
import pytest
import torch

from navbot_ppo_amd import nets, ppo

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _batch(n, seed, dev, half=False, d=16):
    g = torch.Generator().manual_seed(seed)
    obs = torch.rand((n, d), generator=g)
    acts = torch.stack([torch.rand(n, generator=g), torch.rand(n, generator=g) * 2 - 1], 1)
    acts[torch.rand(n, generator=g) < 0.2, 0] = 0.0
    acts[torch.rand(n, generator=g) < 0.1, 1] = 1.0
    logp = -1.2 - 2.3 * torch.rand(n, generator=g)
    rtg = torch.randn(n, generator=g) * 60 + 20
    adv = torch.randn(n, generator=g)
    adv[torch.rand(n, generator=g) < 0.05] = 0.0
    if half:
        obs = obs.half()
    return [t.to(dev).contiguous() for t in (obs, acts, logp, rtg, adv)]


def _nets(dev, scale=3.0, seed=3, d=16):
    torch.manual_seed(seed)
    a, c = nets.make_policy("mlp64x2", d)
    a.to(dev), c.to(dev)
    with torch.no_grad():
        for p in list(a.parameters()) + list(c.parameters()):
            p.mul_(scale)
    return a, c


def _grad(a, c, arith, batch, dev):
    up = ppo.PPOUpdater(copy.deepcopy(a), copy.deepcopy(c), ppo.PPOConfig(policy="mlp64x2", update_arith=arith), None, dev)
    assert up.fused_mlp64 and up.bf16x3 == (arith == "bf16x3")
    up.fp.grad.fill_(9.0)
    up._fused_loss_grad(*batch, 0.5)
    torch.cuda.synchronize()
    return up, up.fp.grad.clone(), up._fstats.clone()


def _errors_vs_float64(n, half, seed, dev, d=16):
    """Per parameter tensor: rms and max error of both arithmetics against float64 autograd, / the tensor's max |gradient|."""
    a, c = _nets(dev, seed=3 + seed, d=d)
    batch = _batch(n, 100 + n + seed, dev, half, d)
    obs, acts, logp, rtg, adv = batch
    a64, c64 = copy.deepcopy(a).double(), copy.deepcopy(c).double()
    al, cl, _, _, _ = ppo.ppo_losses(a64, c64, obs.double(), acts.double(), logp.double(), rtg.double(), adv.double(),
                                     torch.tensor(0.5, dtype=torch.float64, device=dev), 0.2)
    g64 = torch.cat([t.reshape(-1) for t in torch.autograd.grad(al + cl, list(a64.parameters()) + list(c64.parameters()))])
    out = {}
    for arith in ("f32", "bf16x3"):
        up, g, st = _grad(a, c, arith, batch, dev)
        offs = np.cumsum([0] + [q.numel() for q in up.fp.params])
        mx = [((g64[o:e] - g[o:e].double()).abs().max() / (g64[o:e].abs().max() + 1e-300)).item() for o, e in zip(offs[:-1], offs[1:])]
        rms = [(((g64[o:e] - g[o:e].double()) ** 2).mean().sqrt() / (g64[o:e].abs().max() + 1e-300)).item()
               for o, e in zip(offs[:-1], offs[1:])]
        assert st[0].item() == pytest.approx(al.item(), rel=1e-4, abs=1e-6) and st[4].item() == pytest.approx(cl.item(), rel=1e-4)
        out[arith] = (np.array(mx), np.array(rms), st.cpu().numpy())
    return out


def _kink_free_errors(n, seed, dev, d=16, half=False, with_torch32=False):
    """rms / max error per parameter tensor of both arithmetics (and PyTorch's float32 autograd) against float64 autograd on a batch
    whose samples near a relu / clip kink are replaced (tests/_kinks.py): what is left is arithmetic."""
    from _kinks import replace_kink_samples
    a, c = _nets(dev, seed=3 + seed, d=d)
    batch = _batch(n, 100 + n + seed, dev, half, d)
    obs, acts, logp, rtg, adv = batch
    n_kink = replace_kink_samples(a, c, obs, acts, logp, rtg, adv, 0.5)
    assert n_kink <= 2 + n // 100, n_kink   # the batches are not being edited wholesale (0.4 % at these weights)
    a64, c64 = copy.deepcopy(a).double(), copy.deepcopy(c).double()
    al, cl, _, _, _ = ppo.ppo_losses(a64, c64, obs.double(), acts.double(), logp.double(), rtg.double(), adv.double(),
                                     torch.tensor(0.5, dtype=torch.float64, device=dev), 0.2)
    g64 = torch.cat([t.reshape(-1) for t in torch.autograd.grad(al + cl, list(a64.parameters()) + list(c64.parameters()))])
    rows = {}
    for arith in ("f32", "bf16x3"):
        up, g, st = _grad(a, c, arith, batch, dev)
        assert st[0].item() == pytest.approx(al.item(), rel=1e-4, abs=1e-6) and st[4].item() == pytest.approx(cl.item(), rel=1e-4)
        rows[arith] = g
    if with_torch32:
        a32, c32 = copy.deepcopy(a), copy.deepcopy(c)
        al2, cl2, _, _, _ = ppo.ppo_losses(a32, c32, obs.float(), acts, logp, rtg, adv, torch.tensor(0.5, device=dev), 0.2)
        rows["torch32"] = torch.cat([t.reshape(-1) for t in torch.autograd.grad(al2 + cl2, list(a32.parameters()) + list(c32.parameters()))])
    offs = np.cumsum([0] + [q.numel() for q in up.fp.params])
    out = {}
    for k, g in rows.items():
        rms = [(((g64[o:e] - g[o:e].double()) ** 2).mean().sqrt() / (g64[o:e].abs().max() + 1e-300)).item() for o, e in zip(offs[:-1], offs[1:])]
        mx = [((g64[o:e] - g[o:e].double()).abs().max() / (g64[o:e].abs().max() + 1e-300)).item() for o, e in zip(offs[:-1], offs[1:])]
        out[k] = (np.array(rms), np.array(mx))
    return out


@pytest.mark.parametrize("n,seeds", [(128 * 300 + 7, 6), (512 * 4096, 4)])   # the second one is the bench's own batch
def test_bf16x3_error_against_float64_is_not_above_the_f32_paths_on_kink_free_batches(n, seeds):
    """The gate VERDICT round 5 set for `dtype: "f32"` on the split path: gradients of both arithmetics against float64 autograd of
    the same losses (ppo.py:307-349,386) on KINK-FREE batches -- with the relu / clip events gone the comparison is about arithmetic,
    no slack for events -- several seeds, rms and max error per parameter tensor as a fraction of the tensor's gradient scale.

    What two float32 evaluations of equal quality can and cannot satisfy (tools/bf16x3_error_kinkfree.py prints the table,
    profiles/r06_bf16x3_error_kinkfree.txt): per tensor and seed the ratio of their errors scatters between 0.3 and 3 -- half of the
    14 tensors are vectors or scalars whose error is ONE rounding history -- so "<= 1.0 on every tensor" would fail for the f32 path
    against itself.  What does hold, deterministically for these seeds, and is asserted:
      * pooled over tensors and seeds (root of the mean squared rms error) the split path is NOT ABOVE the f32-MFMA path: ratio <= 1.0
        (measured 0.71 at n = 38,407 and 0.94 / 0.65 at the bench batch with 4 / 8 seeds), and below PyTorch's own float32 autograd
        -- the reference's arithmetic (0.38 / 0.84);
      * the worst element error of any tensor and seed is not above the f32 path's worst (7.4e-7 vs 1.3e-6; 3.43e-7 vs 3.47e-7);
      * no tensor's pooled error is more than 2.5 x the f32 path's (the weight-gradient tensors reach 1.3 - 2 x at the bench batch: a
        wave of the 4-wave pass sums 64 tiles into its accumulators, a wave of the 8-wave f32 pass 32 -- summation order, not product
        arithmetic; at 38,407 samples, one or two tiles per wave in both, they are at 0.9 - 1.25).
    Round 6 found and removed the one systematic term: rounds 4-5 added the layer bias to the ROUNDED matrix product, which leaves a
    mean error of ~2e-8 in the heads' pre-activations that the actor's gradient sums amplify 50-fold (actor tensors were 2 - 3.5 x the
    f32 path's error, kink-free); the accumulators now start as the bias, like the f32 pass's (csrc/ppo_mlp64_x3s.h: relu4)."""
    dev = torch.device("cuda")
    R = {k: [] for k in ("f32", "bf16x3", "torch32")}
    M = {k: [] for k in ("f32", "bf16x3", "torch32")}
    for seed in range(seeds):
        out = _kink_free_errors(n, seed, dev, with_torch32=True)
        for k in R:
            R[k].append(out[k][0])
            M[k].append(out[k][1])
    pooled = {k: np.sqrt(np.mean(np.square(np.stack(R[k])), 0)) for k in R}   # per tensor, over seeds
    tot = {k: float(np.sqrt(np.mean(pooled[k] ** 2))) for k in R}
    worst = {k: float(np.stack(M[k]).max()) for k in M}
    print(f"n = {n}, {seeds} seeds, kink-free: pooled rms error / tensor scale  f32-MFMA {tot['f32']:.2e}  bf16x3 {tot['bf16x3']:.2e}  "
          f"PyTorch float32 {tot['torch32']:.2e};  worst element error {worst['f32']:.2e} / {worst['bf16x3']:.2e} / {worst['torch32']:.2e};  "
          f"per-tensor ratio bf16x3 / f32: " + " ".join(f"{v:.2f}" for v in pooled["bf16x3"] / pooled["f32"]))
    assert tot["bf16x3"] <= 1.0 * tot["f32"], (tot["bf16x3"], tot["f32"])
    assert tot["bf16x3"] <= 1.0 * tot["torch32"], (tot["bf16x3"], tot["torch32"])
    assert worst["bf16x3"] <= 1.0 * worst["f32"], (worst["bf16x3"], worst["f32"])
    assert (pooled["bf16x3"] <= 2.5 * pooled["f32"]).all(), pooled["bf16x3"] / pooled["f32"]
    assert tot["bf16x3"] < 1.5e-7


@pytest.mark.parametrize("n,d", [(128 * 300 + 7, 42), (512 * 4096, 42)])   # 42-column rows: configs[3]'s shard (round 5's compiler-scheduled pass)
def test_bf16x3_error_is_not_above_the_f32_paths_error_against_float64(n, d):
    """The 42-column rows run round 5's split pass (8 / 4 waves, compiler-scheduled; bias added behind the product): its round-5 gate,
    unchanged -- medians over row types and seeds on batches WITH their kink events (critic tensors <= 1.2 x the f32 path, all tensors
    <= 1.75 x and < 1.5e-7), every tensor inside the autograd bound or within 1.5 x the f32 path's worst."""
    dev = torch.device("cuda")
    r32, rx3 = [], []
    for half in (False, True):
        for seed in range(3):
            out = _errors_vs_float64(n, half, seed, dev, d)
            r32.append(out["f32"][1])
            rx3.append(out["bf16x3"][1])
            assert out["bf16x3"][0].max() <= 2e-4 or out["bf16x3"][0].max() <= 1.5 * out["f32"][0].max(), (out["bf16x3"][0], out["f32"][0])
            np.testing.assert_allclose(out["bf16x3"][2], out["f32"][2], rtol=5e-6, atol=1e-7)
    r32, rx3 = np.stack(r32), np.stack(rx3)   # [6 cases, 14 tensors]; tensors 8 .. 13 are the critic's
    m32, mx3 = float(np.median(r32)), float(np.median(rx3))
    c32, cx3 = float(np.median(r32[:, 8:])), float(np.median(rx3[:, 8:]))
    print(f"n = {n}, {d} columns: median rms error / tensor scale, f32-MFMA path vs bf16x3: critic tensors {c32:.2e} vs {cx3:.2e}, all tensors {m32:.2e} vs {mx3:.2e}")
    assert cx3 <= 1.2 * c32 + 1e-9, (cx3, c32)
    assert mx3 <= 1.75 * m32 + 1e-9 and mx3 <= 1.5e-7, (mx3, m32)


@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
@pytest.mark.parametrize("n,d", [(128, 16), (1000, 16), (128 * 300 + 7, 16), (1 << 17, 16), (1, 42), (31, 42), (1000, 42), (128 * 300 + 7, 42)])
def test_gradients_match_autograd_on_both_arithmetics(n, d, arith):
    """test_fused_mlp64_gradients_match_autograd (float32 autograd, 2e-4 of each tensor's scale) with the arithmetic pinned."""
    dev = torch.device("cuda")
    a, c = _nets(dev, d=d)
    batch = _batch(n, n, dev, d=d)
    obs, acts, logp, rtg, adv = batch
    from _kinks import replace_kink_samples
    n_kink = replace_kink_samples(a, c, obs, acts, logp, rtg, adv, 0.5)   # (tests/_kinks.py: the bound below is about arithmetic)
    assert n_kink <= 2 + n // 100, n_kink   # ... on the batch as drawn, not on one edited wholesale
    up, g, st = _grad(a, c, arith, batch, dev)
    a2, c2 = copy.deepcopy(a), copy.deepcopy(c)
    al, cl, ratios, lp, _ = ppo.ppo_losses(a2, c2, obs, acts, logp, rtg, adv, torch.tensor(0.5, device=dev), 0.2)
    g_ref = torch.cat([t.reshape(-1) for t in torch.autograd.grad(al + cl, list(a2.parameters()) + list(c2.parameters()))])
    off = 0
    for prm in up.fp.params:
        k = prm.numel()
        scale = g_ref[off:off + k].abs().max().item() + 1e-12
        assert (g_ref[off:off + k] - g[off:off + k]).abs().max().item() <= 2e-4 * scale + 1e-7, (tuple(prm.shape), arith)
        off += k
    # one-net entry points (the multi-GPU pipeline) write the same slices
    stats = torch.zeros(8, device=dev)
    up.fp.grad.fill_(7.0)
    up._fused_loss_grad_net(0, *batch, 0.5, stats)
    up._fused_loss_grad_net(1, *batch, 0.5, stats)
    torch.cuda.synchronize()
    assert torch.equal(up.fp.grad, g)


@pytest.mark.parametrize("d", [16, 42])
@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
def test_update_tracks_pytorch_on_both_arithmetics(arith, d):
    """test_fused_update_tracks_pytorch_update_over_epochs (10 Adam epochs against PyTorch autograd + torch.optim.Adam) with the
    arithmetic pinned.  (G7, the reference's own learn() golden, is a 512-wide-net fixture: test_gpu_resmlp512.py.)"""
    dev = torch.device("cuda")
    obs, acts, logp, rtg, adv = _batch(1 << 15, 5, dev, d=d)
    res = []
    for fused in (True, False):
        torch.manual_seed(11)
        a, c = nets.make_policy("mlp64x2", d)
        a.to(dev), c.to(dev)
        up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2", n_updates_per_iteration=10, fused_update=fused, update_arith=arith), None, dev)
        st = up.update(obs, acts, logp, rtg, torch.tensor(0.8, device=dev))
        res.append((up.fp.flat.clone(), up.loss_history.clone(), st))
    (w1, h1, s1), (w0, h0, s0) = res
    np.testing.assert_allclose(h1.cpu().numpy(), h0.cpu().numpy(), rtol=2e-4, atol=1e-5)
    assert (w1 - w0).abs().max().item() < 3e-5
    for k in ("actor_loss", "critic_loss", "approx_kl", "clip_frac"):
        assert s1[k] == pytest.approx(s0[k], rel=2e-3, abs=2e-5), k


def test_prepare_is_redone_for_every_update_and_follows_in_place_writes():
    """The split observations are made once per update() -- never reused across updates, because the rollout kernels fill the
    observation buffer behind torch's back -- and the loss_grad entry points re-split when the tensor they were made from changed."""
    dev = torch.device("cuda")
    a, c = _nets(dev, scale=1.0)
    up = ppo.PPOUpdater(a, c, ppo.PPOConfig(policy="mlp64x2", n_updates_per_iteration=2), None, dev)
    assert up.bf16x3
    obs, acts, logp, rtg, adv = _batch(4096, 2, dev)
    up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.5)
    g1 = up.fp.grad.clone()
    # a write that torch does not see (as a HIP kernel through the C ABI would do it): update() must still see the new rows
    import ctypes as C
    new = torch.rand_like(obs)
    torch.cuda.synchronize()
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(obs.data_ptr()), C.c_void_p(new.data_ptr()), C.c_size_t(obs.numel() * 4), 3)
    w0 = up.fp.flat.clone()
    up.update(obs, acts, logp, rtg, torch.tensor(0.5, device=dev))
    a2, c2 = _nets(dev, scale=1.0)
    up2 = ppo.PPOUpdater(a2, c2, ppo.PPOConfig(policy="mlp64x2", n_updates_per_iteration=2), None, dev)
    assert torch.equal(up2.fp.flat, w0)
    up2.update(new.clone(), acts, logp, rtg, torch.tensor(0.5, device=dev))
    assert torch.equal(up.fp.flat, up2.fp.flat)
    obs.mul_(0.5)   # a torch write: the version counter moves, loss_grad re-splits by itself
    up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.5)
    assert not torch.equal(up.fp.grad, g1)


def test_hand_placed_stream_on_ragged_batch_sizes():
    """mlp64_pass_both_x3s (csrc/ppo_mlp64_x3s.h) carries state from tile to tile -- the G1 product of a tile runs behind the NEXT tile's
    MFMAs, rows are requested a tile ahead, a wave's partial sums live in its lanes over all its tiles -- so a batch size decides which wave
    sees how many tiles, whether its last one is ragged and where the tail G1 runs.  The strict gradient test above on sizes around every
    such boundary (one sample, the 32-sample tile, the 4 waves of a workgroup, one tile per wave over all 1024 waves, one more), 16
    columns, against float32 autograd at the suite's bound."""
    dev = torch.device("cuda")
    a, c = _nets(dev)
    from _kinks import replace_kink_samples
    for n in (1, 2, 31, 32, 33, 63, 65, 127, 129, 4095, 4097, 32 * 1024 - 1, 32 * 1024, 32 * 1024 + 1, 32 * 1024 + 33, 3 * 32 * 1024 + 5):
        batch = _batch(n, 7 * n + 1, dev)
        obs, acts, logp, rtg, adv = batch
        n_kink = replace_kink_samples(a, c, obs, acts, logp, rtg, adv, 0.5)
        assert n_kink <= 2 + n // 100, (n, n_kink)
        up, g, st = _grad(a, c, "bf16x3", batch, dev)
        a2, c2 = copy.deepcopy(a), copy.deepcopy(c)
        al, cl, ratios, lp, _ = ppo.ppo_losses(a2, c2, obs, acts, logp, rtg, adv, torch.tensor(0.5, device=dev), 0.2)
        g_ref = torch.cat([t.reshape(-1) for t in torch.autograd.grad(al + cl, list(a2.parameters()) + list(c2.parameters()))])
        off = 0
        for prm in up.fp.params:
            k = prm.numel()
            scale = g_ref[off:off + k].abs().max().item() + 1e-12
            assert (g_ref[off:off + k] - g[off:off + k]).abs().max().item() <= 2e-4 * scale + 1e-7, (n, tuple(prm.shape))
            off += k
