"""PPO-clip rollout / return-scan / update loop over the batched simulator.

Follows ``project_ppo/src/ppo.py`` of the reference function by function, vectorised over N envs:

  rollout()        ppo.py:463-641   T policy steps x N envs; every buffer is a [T,N,...] device tensor
                                    the step kernel writes into directly (no host round trip)
  sample_action    ppo.py:673-706   MVN(mean, var*I) sample, clamp, log-prob of the CLAMPED action
  compute returns  ppo.py:643-671   navsim_rtg_scan (HIP): reward-to-go, no bootstrap
  evaluate         ppo.py:708-737
  update           ppo.py:275-397   A = rtg - V, normalised with the unbiased std (+1e-10); 50 full-batch
                                    epochs of clipped surrogate + MSE critic, Adam(lr 3e-4), no entropy
                                    term, no value clip, no gradient clipping
  learn            ppo.py:218-461   iteration loop, timing, checkpoints (actor_iter%04d_step%08d.pth)

Multi-GPU (absent in the reference): one process per GPU, each owns a contiguous shard
of env ids; after each backward ONE all-reduce (RCCL over xGMI; gloo on CPU for tests) of the single
flat gradient buffer holding actor+critic; the advantage mean/std come from all-reduced
(sum, sum of squares, count) so they equal the single-process values of ppo.py:284.
"""
import dataclasses
import math
import os
import time

import torch
import torch.distributed as dist

from . import nets

LOG_2PI = math.log(2.0 * math.pi)


@dataclasses.dataclass
class PPOConfig:
    rollout_len: int = 512                 # T: policy steps per env per iteration
    max_episode_steps: int = 500           # main.py / arguments.py:30 (timesteps_per_episode)
    gamma: float = 0.99                    # main.py:470
    n_updates_per_iteration: int = 50      # main.py:471
    lr: float = 3e-4                       # main.py:472
    clip: float = 0.2                      # main.py:473
    policy: str = "resmlp512"              # reference nets; "mlp64x2" = BASELINE config 2
    gae_lambda: float = None               # None = the reference's estimator A = rtgs - V (ppo.py:277); a number in [0, 1] turns
                                           # on GAE(lambda) (navsim_gae_scan): advantages and critic targets from the lambda-return
    init_var: float = 0.8                  # ppo.py:123
    var_decay: float = 0.995               # ppo.py:695
    var_floor: float = 0.1                 # ppo.py:694
    var_decay_after: int = 50000           # ppo.py:694
    save_freq: int = 2                     # ppo.py:774
    seed: int = 0
    use_graph: bool = True                 # capture the T-step rollout in one hipGraph
    persistent_rollout: bool = True        # on GPU: all T steps in ONE launch (navsim_rollout_mlp64 / navsim_rollout_resmlp512)
    fused_update: bool = True              # on GPU: fused HIP loss+gradient kernels (csrc/ppo_mlp64.hip, csrc/ppo_resmlp512.hip)
    # arithmetic of the fused update's matrix products: "bf16x3" = float32 products out of operands split into three bf16
    # pieces on the bf16 MFMA (six piece products, float32 accumulate: float32-equivalent by measurement), "f32" =
    # the f32-input MFMA (native float32 fma chains).  Inputs, outputs and everything around the products are float32 either way.
    # mlp64x2: both builds exist.  resmlp512: the fused kernels are "bf16x3" (split products where a k-step of the bf16 MFMA is
    # filled, f32-input MFMA elsewhere; V0 from navppo_resmlp512_value included); "f32" selects the PyTorch float32 path with a warning.
    update_arith: str = "bf16x3"
    # multi-GPU, mlp64x2: False = fused passes of both nets -> ONE all-reduce of the flat gradient -> Adam (the default: at one
    # RCCL rank this path costs 9-24 us per epoch over the single-GPU epoch, the per-net pipeline below 59-74 us, because two
    # pass launches pay the ramp / staging / reduction of the fused one twice -- more than a 43 KB all-reduce costs on the wire);
    # True = two-stage pipeline, each net's all-reduce under the other net's pass (_pipelined_epochs): hides the wire entirely
    overlap_allreduce: bool = False
    output_dir: str = ""                   # "" = no checkpoints / logs
    episode_csv_rows: int = 2000           # per-iteration cap on rows appended to <method>_train_episodes.csv (0 = off)
    tb_episode_rows: int = 256             # per-iteration cap on Episode_Rewards/train points in the TensorBoard file (0 = off)
    method_name: str = "baseline"


# --------------------------------------------------------------------------- distributed context
class DistCtx:
    """torch.distributed wrapper that degrades to a no-op for a single process."""

    def __init__(self, device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # NAVBOT_DIST_FORCE=1: a single rank still creates its process group and sends every collective through the backend --
        # on a one-GPU box this is how the RCCL branch (init with device_id, all-reduces on RCCL's stream, the overlap of the
        # actor's all-reduce with the critic pass, navppo_adam_step's 1/world scale) is executed at all
        self.enabled = self.world > 1 or os.environ.get("NAVBOT_DIST_FORCE") == "1"
        if device is None:
            if torch.cuda.is_available():
                device = torch.device(f"cuda:{self.local_rank % torch.cuda.device_count()}")
            else:
                device = torch.device("cpu")
        self.device = torch.device(device)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        self.backend, self.rccl_version = None, None
        if self.enabled and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            # "nccl" is RCCL on ROCm.  NAVBOT_DIST_BACKEND=gloo lets several ranks share one GPU (tests only:
            # RCCL refuses two ranks on the same device).
            backend = os.environ.get("NAVBOT_DIST_BACKEND") or ("nccl" if self.device.type == "cuda" else "gloo")
            kw = {"device_id": self.device} if backend == "nccl" else {}
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
        if self.enabled:
            # a launcher may have created the group itself (backend=None or "cpu:gloo,cuda:nccl"): ask for the CUDA backend
            try:
                self.backend = dist.get_backend_config().get_device_backend_map().get("cuda", dist.get_backend())
            except Exception:
                self.backend = dist.get_backend()
            if self.device.type == "cuda" and not os.environ.get("NAVBOT_DIST_BACKEND") and "nccl" not in str(self.backend):
                import warnings
                warnings.warn(f"GPU ranks should talk RCCL (torch backend 'nccl'); the process group uses {self.backend!r}")
            if "nccl" in str(self.backend):
                self.backend = "nccl"
                try:
                    self.rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception:
                    self.rccl_version = "unknown"
                if self.rank == 0:   # one line per job: a scaling run shows how many ranks RCCL really connected
                    print(f"[navbot_ppo_amd] RCCL {self.rccl_version}: {dist.get_world_size()} ranks, one per GPU, "
                          f"flat-gradient all-reduce per epoch", flush=True)

    def all_reduce_sum(self, t):
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def all_reduce_max(self, t):
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t

    def broadcast(self, t, src=0):
        if self.enabled:
            dist.broadcast(t, src=src)
        return t

    def barrier(self):
        if self.enabled:
            dist.barrier()

    def shard(self, n_total):
        """Contiguous env-id shard [lo, hi) of this rank."""
        per = n_total // self.world
        if per * self.world != n_total:
            raise ValueError(f"n_envs {n_total} not divisible by world size {self.world}")
        return self.rank * per, (self.rank + 1) * per


# --------------------------------------------------------------------------- flat parameters
class FlatParams:
    """All trainable tensors of actor + critic as views into ONE flat buffer, and their gradients as
    views into ONE flat gradient buffer: a single all-reduce and a single Adam update per epoch.
    (The reference's unused BatchNorm parameters never receive gradients -- net_actor.py:44,48 -- and
    stay ordinary tensors so state_dict keys are unchanged.)"""

    def __init__(self, modules, device):
        per_module = [[p for n, p in m.named_parameters() if ".bn" not in "." + n and not n.startswith("bn")] for m in modules]
        self.params = [p for ps in per_module for p in ps]
        self.module_numel = [sum(p.numel() for p in ps) for ps in per_module]   # [actor, critic] slices of the flat buffers
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            p.grad = self.grad[off:off + n].view_as(p)
            off += n
        self.numel = total
        self.proxy = torch.nn.Parameter(self.flat, requires_grad=True)
        self.proxy.data = self.flat
        self.proxy.grad = self.grad


def gaussian_log_prob(mean, action, var):
    """log N(action; mean, var*I) for a 2-D isotropic Gaussian = MultivariateNormal(mean, diag(var)).log_prob
    (ppo.py:696,704,734-735)."""
    k = mean.shape[-1]
    return -0.5 * (((action - mean) ** 2).sum(-1) / var) - 0.5 * k * LOG_2PI - 0.5 * k * torch.log(var)


def ppo_losses(actor, critic, obs, acts, logp_old, rtg, adv, var, clip):
    """One evaluation of ppo.py:307-343.  Returns (actor_loss, critic_loss, ratios, logp)."""
    V = critic(obs).squeeze(-1)
    mean = actor(obs)
    lo = mean.new_tensor([0.0, -1.0])
    hi = mean.new_tensor([1.0, 1.0])
    mean = torch.max(torch.min(mean, hi), lo)              # ppo.py:730-733 (no-op after sigmoid/tanh)
    logp = gaussian_log_prob(mean, acts, var)
    ratios = torch.exp(logp - logp_old)                    # ppo.py:316
    surr1 = ratios * adv                                   # ppo.py:319
    surr2 = torch.clamp(ratios, 1 - clip, 1 + clip) * adv  # ppo.py:320
    actor_loss = (-torch.min(surr1, surr2)).mean()         # ppo.py:342
    critic_loss = torch.nn.functional.mse_loss(V, rtg)     # ppo.py:343
    return actor_loss, critic_loss, ratios, logp, V


def normalise_advantages(adv, ctx=None):
    """(A - mean) / (std + 1e-10) with torch.std's unbiased estimator (ppo.py:284); with several ranks the
    moments are all-reduced so every rank uses the global mean/std."""
    a64 = adv.double()
    m = torch.stack([a64.sum(), (a64 * a64).sum(), torch.tensor(float(adv.numel()), dtype=torch.float64, device=adv.device)])
    if ctx is not None:
        ctx.all_reduce_sum(m)
    n = m[2]
    mean = m[0] / n
    var = (m[1] - n * mean * mean) / (n - 1)
    std = torch.sqrt(torch.clamp(var, min=0.0))
    return ((adv - mean.float()) / (std.float() + 1e-10))


class PPOUpdater:
    """The update half of PPO.learn (ppo.py:275-397) on a fixed batch; device-agnostic PyTorch."""

    def __init__(self, actor, critic, cfg, ctx=None, device=None):
        self.actor, self.critic, self.cfg, self.ctx = actor, critic, cfg, ctx
        self.device = device or next(actor.parameters()).device
        self.fp = FlatParams([actor, critic], self.device)
        if ctx is not None:
            ctx.broadcast(self.fp.flat, 0)  # identical replicas; identical Adam steps keep them in sync
        fused = self.device.type == "cuda"
        self.opt = torch.optim.Adam([self.fp.proxy], lr=cfg.lr, fused=fused)  # == the two Adam(lr) of ppo.py:116-117
        self.stats = {}
        # HIP paths: fused f32-MFMA kernels instead of ~40 (mlp64x2) / ~120 (resmlp512) PyTorch kernels per epoch
        on_gpu = self.device.type == "cuda" and cfg.fused_update
        # the D-64-64 heads: D = 16 (10 beams) or 42 (36 beams), rows float32 or float16 (obs_f16 envs) -- include/navppo.h
        self.fused_mlp64 = (on_gpu and cfg.policy == "mlp64x2" and isinstance(actor, nets.MLP64Actor)
                            and actor.layer1.in_features in (16, 42))
        self.obs_dim = actor.layer1.in_features if isinstance(actor, nets.MLP64Actor) else actor.rb1.f_in
        if cfg.update_arith not in ("f32", "bf16x3"):
            raise ValueError(f"update_arith {cfg.update_arith!r}: 'f32' or 'bf16x3'")
        self.fused_resmlp512 = (on_gpu and cfg.policy == "resmlp512" and isinstance(actor, nets.ResMLPActor)
                                and actor.rb1.f_in == 16 and actor.rb1.fc1.out_features == 512)
        if self.fused_resmlp512 and cfg.update_arith == "f32":
            # the fused 512-wide kernels exist in ONE arithmetic (csrc/ppo_resmlp512.hip: the products that fill a k-step of the
            # bf16 MFMA are split-bf16 float32 products, the rest f32-input MFMA).  A caller who asks for native float32
            # products gets them -- from PyTorch's float32 GEMMs, ~4 x slower per epoch -- instead of being ignored.
            import warnings
            warnings.warn("update_arith='f32' with policy='resmlp512': the fused 512-wide kernels have no all-f32-MFMA build; "
                          "this updater runs the PyTorch float32 path (slower).  Use update_arith='bf16x3' (default) for the fused kernels.")
            self.fused_resmlp512 = False
        self.fused = "navppo_mlp64" if self.fused_mlp64 else "navppo_resmlp512" if self.fused_resmlp512 else None
        self.bf16x3 = self.fused_mlp64 and cfg.update_arith == "bf16x3"   # (16- and 42-column rows, float32 or float16)
        self._prep = self._prep_key = None
        if self.fused:
            from ._native import lib
            self._n_actor = self.fp.module_numel[0]
            d = self.obs_dim
            assert tuple(self.fp.module_numel) == ((64 * d + 4354, 64 * d + 4289) if self.fused_mlp64 else (50290, 50257))
            self._ws = None
            if self.fused_mlp64:
                self._ws = torch.empty(lib().navppo_mlp64_workspace_bytes(d) // 4, dtype=torch.float32, device=self.device)
            self._fstats = torch.zeros(8, dtype=torch.float32, device=self.device)
            self._fhist = torch.zeros((max(cfg.n_updates_per_iteration, 1), 8), dtype=torch.float32, device=self.device)
            # single GPU: Adam runs inside the kernel that sums the partial gradients (navppo_*_update_epoch)
            self._adam_m = torch.zeros_like(self.fp.flat)
            self._adam_v = torch.zeros_like(self.fp.flat)
            self._adam_t = 0

    def _workspace(self, n):
        """Scratch of the fused kernels: fixed for the 2x64 heads, per-sample partial block outputs for the 512-wide nets."""
        if self.fused_resmlp512:
            from ._native import lib
            need = lib().navppo_resmlp512_workspace_bytes(int(n)) // 4 + 4
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.float32, device=self.device)
        return self._ws

    def _obs_args(self, obs):
        """The observation arguments of the fused entry points -- (pointer, obs_dim, obs_f16) for the D-64-64 heads, (pointer, obs_f16)
        for the 512-wide nets -- after checking what is behind the pointer: rows of self.obs_dim columns, float32 or float16 (the
        kernels widen half rows as they load)."""
        import ctypes as C
        ok = (torch.float32, torch.float16)
        if obs.dim() != 2 or obs.shape[1] != self.obs_dim or obs.dtype not in ok or not obs.is_contiguous():
            raise ValueError(f"{self.fused}: observations must be contiguous [n, {self.obs_dim}] rows of {ok}, got "
                             f"{tuple(obs.shape)} {obs.dtype}")
        p = C.c_void_p(obs.data_ptr())
        f16 = int(obs.dtype == torch.float16)
        return (p, self.obs_dim, f16) if self.fused_mlp64 else (p, f16)

    def prepare(self, obs):
        """bf16x3: split the batch's observations into bf16 pieces (navppo_mlp64_bf16x3_prepare) -- once per update, the rows do not
        change over the epochs.  The epoch entry points use the prepared buffer while `obs` is the tensor it was made from."""
        import ctypes as C
        from ._native import lib
        L = lib()
        p, d, f16 = self._obs_args(obs)
        need = L.navppo_mlp64_bf16x3_prep_bytes(int(obs.shape[0]), self.obs_dim)
        if self._prep is None or self._prep.numel() < need:
            self._prep = None
            self._prep = torch.empty(need, dtype=torch.uint8, device=self.device)
        rc = L.navppo_mlp64_bf16x3_prepare(p, d, f16, int(obs.shape[0]), C.c_void_p(self._prep.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"navppo_mlp64_bf16x3_prepare failed: {L.navppo_last_error().decode()}")
        self._prep_key = (obs.data_ptr(), tuple(obs.shape), obs.dtype, obs._version)

    def invalidate_prepared(self):
        """The rows behind a prepared tensor changed without torch noticing (the HIP rollout / step kernels write the trainer's
        buffers through raw pointers: no version bump).  PPOTrainer.rollout() calls this; any other writer of such a buffer must."""
        self._prep_key = None

    def _prepared(self, obs):
        """The pre-split pieces of `obs`.  Reused while `obs` is the very tensor prepare() saw, unchanged as far as torch knows
        (pointer, shape, dtype, version counter) AND nobody called invalidate_prepared() since."""
        import ctypes as C
        if self._prep_key != (obs.data_ptr(), tuple(obs.shape), obs.dtype, obs._version):
            self.prepare(obs)
        return C.c_void_p(self._prep.data_ptr())

    def _fused_loss_grad(self, obs, acts, logp_old, rtg, adv, var, stats=None):
        """evaluate + losses + backward of ppo.py:307-386 in the HIP kernels of csrc/ppo_mlp64.hip; gradients land
        in the flat gradient buffer, (actor_loss, approx_kl, clip_frac, -, critic_loss) in self._fstats."""
        import ctypes as C
        from ._native import lib
        L = lib()
        ptr = lambda t: C.c_void_p(t.data_ptr())
        for t in (acts, logp_old, rtg, adv):
            assert t.is_contiguous() and t.dtype == torch.float32
        name, oargs = (("navppo_mlp64_bf16x3", (self._prepared(obs), self.obs_dim)) if self.bf16x3 else (self.fused, self._obs_args(obs)))
        rc = getattr(L, name + "_loss_grad")(ptr(self.fp.flat), *oargs, ptr(acts), ptr(logp_old), ptr(rtg), ptr(adv),
                                                   int(obs.shape[0]), float(var), float(self.cfg.clip), ptr(self.fp.grad),
                                                   ptr(self._fstats if stats is None else stats), ptr(self._workspace(obs.shape[0])),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"{self.fused}_loss_grad failed: {L.navppo_last_error().decode()}")

    def _fused_loss_grad_net(self, net, obs, acts, logp_old, rtg, adv, var, stats):
        """One net's half of _fused_loss_grad (navppo_mlp64_loss_grad_net): its slice of the flat gradient, its statistics."""
        import ctypes as C
        from ._native import lib
        L = lib()
        ptr = lambda t: C.c_void_p(t.data_ptr())
        fn, oargs = ((L.navppo_mlp64_bf16x3_loss_grad_net, (self._prepared(obs), self.obs_dim)) if self.bf16x3
                     else (L.navppo_mlp64_loss_grad_net, self._obs_args(obs)))
        rc = fn(int(net), ptr(self.fp.flat), *oargs, ptr(acts), ptr(logp_old), ptr(rtg), ptr(adv),
                                          int(obs.shape[0]), float(var), float(self.cfg.clip), ptr(self.fp.grad), ptr(stats),
                                          ptr(self._workspace(obs.shape[0])), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"navppo_mlp64_loss_grad_net failed: {L.navppo_last_error().decode()}")

    def _fused_adam(self, grad_scale, lo=0, n=None, step=None):
        """Scale + Adam on the flat buffer, or on the slice [lo, lo + n) at optimiser step `step` (one net of the pipelined epoch)."""
        import ctypes as C
        from ._native import lib
        L = lib()
        ptr = lambda t: C.c_void_p(t.data_ptr() + 4 * lo)
        if step is None:
            self._adam_t += 1
            step = self._adam_t
        rc = L.navppo_adam_step(ptr(self.fp.flat), ptr(self.fp.grad), ptr(self._adam_m), ptr(self._adam_v),
                                int(self.fp.numel - lo if n is None else n), float(grad_scale), float(self.cfg.lr), 0.9, 0.999, 1e-8,
                                int(step), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"navppo_adam_step failed: {L.navppo_last_error().decode()}")

    def _pipelined_epochs(self, n_ep, world, obs, acts, logp_old, rtg, adv, var_f):
        """The multi-GPU epochs of the 2x64 heads as a two-stage pipeline.  Actor and critic are disjoint nets whose losses share
        nothing inside the epoch loop (the advantages are fixed before it, ppo.py:275-284), so each net's all-reduce (RCCL's own
        stream) runs under the OTHER net's pass -- the actor's under the critic's pass of the same epoch, the critic's under the
        actor's pass of the next one -- and neither the wire time nor the cross-stream hand-over is on the compute stream's
        critical path.  Same kernels on the same data as the unpipelined order: the weights are bit-identical."""
        n_a = self._n_actor
        n_c = self.fp.numel - n_a
        t0 = self._adam_t
        wa = wc = None
        gn_sq = torch.zeros((max(n_ep, 1), 2), device=obs.device)   # every epoch's squared norms of the summed (actor, critic) gradient
        ga, gc = self.fp.grad[:n_a], self.fp.grad[n_a:]
        for ep in range(n_ep):
            if wa is not None:   # epoch ep - 1's actor gradient has arrived (long ago: it had the critic's pass to do so)
                wa.wait()
                self._fused_adam(1.0 / world, 0, n_a, t0 + ep)
                gn_sq[ep - 1, 0] = torch.dot(ga, ga)
            self._fused_loss_grad_net(0, obs, acts, logp_old, rtg, adv, var_f, self._fhist[ep])
            wa = dist.all_reduce(self.fp.grad[:n_a], op=dist.ReduceOp.SUM, async_op=True)
            if wc is not None:
                wc.wait()
                self._fused_adam(1.0 / world, n_a, n_c, t0 + ep)
                gn_sq[ep - 1, 1] = torch.dot(gc, gc)
            self._fused_loss_grad_net(1, obs, acts, logp_old, rtg, adv, var_f, self._fhist[ep])
            wc = dist.all_reduce(self.fp.grad[n_a:], op=dist.ReduceOp.SUM, async_op=True)
        if wa is not None:
            wa.wait()
            self._fused_adam(1.0 / world, 0, n_a, t0 + n_ep)
            gn_sq[n_ep - 1, 0] = torch.dot(ga, ga)
            wc.wait()
            self._fused_adam(1.0 / world, n_a, n_c, t0 + n_ep)
            gn_sq[n_ep - 1, 1] = torch.dot(gc, gc)
        self._adam_t = t0 + n_ep
        return gn_sq / float(world) ** 2

    def _fused_value(self, obs):
        """V = critic(obs).squeeze() (ppo.py:275) by the forward half of the critic's fused pass."""
        import ctypes as C
        from ._native import lib
        L = lib()
        out = torch.empty(obs.shape[0], dtype=torch.float32, device=obs.device)
        critic = C.c_void_p(self.fp.flat.data_ptr() + 4 * self._n_actor)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self.fused_resmlp512:
            rc = L.navppo_resmlp512_value(critic, *self._obs_args(obs), int(obs.shape[0]), C.c_void_p(out.data_ptr()),
                                          C.c_void_p(self._workspace(obs.shape[0]).data_ptr()), st)
        else:
            rc = L.navppo_mlp64_value(critic, *self._obs_args(obs), int(obs.shape[0]), C.c_void_p(out.data_ptr()), st)
        if rc != 0:
            raise RuntimeError(f"{self.fused}_value failed: {L.navppo_last_error().decode()}")
        return out

    def _fused_epoch(self, obs, acts, logp_old, rtg, adv, var, stats):
        """One epoch of ppo.py:305-392 on one GPU: losses, gradients and both Adam steps in four launches."""
        import ctypes as C
        from ._native import lib
        L = lib()
        ptr = lambda t: C.c_void_p(t.data_ptr())
        self._adam_t += 1
        name, oargs = (("navppo_mlp64_bf16x3", (self._prepared(obs), self.obs_dim)) if self.bf16x3 else (self.fused, self._obs_args(obs)))
        rc = getattr(L, name + "_update_epoch")(ptr(self.fp.flat), *oargs, ptr(acts), ptr(logp_old), ptr(rtg), ptr(adv),
                                                      int(obs.shape[0]), float(var), float(self.cfg.clip), float(self.cfg.lr), 0.9,
                                                      0.999, 1e-8, int(self._adam_t), ptr(self._adam_m), ptr(self._adam_v),
                                                      ptr(self.fp.grad), ptr(stats), ptr(self._workspace(obs.shape[0])),
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"{self.fused}_update_epoch failed: {L.navppo_last_error().decode()}")

    def value(self, obs):
        """V = critic(obs).squeeze() (ppo.py:275) for [n, D] rows."""
        with torch.no_grad():
            if self.fused and obs.is_contiguous() and obs.data_ptr() % 16 == 0 and obs.dtype in (torch.float32, torch.float16):
                return self._fused_value(obs)
            return self.critic(obs.float()).squeeze(-1)

    def update(self, obs, acts, logp_old, rtg, var, adv_raw=None, V0=None):
        """adv_raw / V0: advantages (before normalisation) and values computed by the caller (GAE); default = the reference's
        A = rtg - V (ppo.py:275-277)."""
        cfg, ctx = self.cfg, self.ctx
        world = ctx.world if ctx is not None else 1
        multi = ctx is not None and ctx.enabled   # collectives run (world > 1, or one rank forced through the backend)
        with torch.no_grad():
            if V0 is None:
                V0 = self.value(obs)
            adv = normalise_advantages(rtg - V0 if adv_raw is None else adv_raw, ctx)          # ppo.py:275-284
        n_ep = cfg.n_updates_per_iteration
        flat_before = self.fp.flat.clone()                     # for the parameter-delta diagnostics of ppo.py:402-403
        a_loss = c_loss = torch.zeros((), device=obs.device)   # n_updates_per_iteration == 0: nothing to report
        acc = torch.zeros(6, device=obs.device)                # sums over epochs of diagnostics
        fused_gn_sq = None                                     # fused single-GPU path: [n_ep - 1, 2] squared per-net gradient norms
        net_gn = None                                          # PyTorch path: sums over epochs of the per-net gradient norms
        multi_gn = None                                        # fused multi-GPU path: sums over epochs of (actor, critic, total) norms of the mean gradient
        self.loss_history = torch.zeros((n_ep, 2), device=obs.device)  # per-epoch (actor, critic) loss, ppo.py:396-397
        var_f = float(var) if self.fused else None
        if not (self.fused and obs.dtype == torch.float16):
            obs = obs.float()   # half rows (obs_f16 envs) are consumed as they are by the fused kernels only
        if self.fused:
            obs, acts, logp_old, rtg, adv = (t.contiguous() for t in (obs, acts, logp_old, rtg, adv))
            if self.bf16x3 and n_ep > 0:
                self.prepare(obs)   # ALWAYS here: the rollout kernels fill the buffer behind torch's back (no version bump)
            if self._fhist.shape[0] < n_ep:
                self._fhist = torch.zeros((n_ep, 8), dtype=torch.float32, device=self.device)
        pipelined = self.fused and multi and self.fused_mlp64 and cfg.overlap_allreduce
        if pipelined:
            pg = self._pipelined_epochs(n_ep, world, obs, acts, logp_old, rtg, adv, var_f)   # squared norms of the MEAN gradient, per epoch and net
            if n_ep > 0:
                multi_gn = torch.cat([pg.sqrt().sum(0), pg.sum(1).sqrt().sum().reshape(1)])
        for ep in range(n_ep):                                 # ppo.py:305
            if self.fused:
                # per-epoch diagnostics land in row ep of a device buffer: no extra launches inside the epoch loop
                if pipelined:
                    pass   # all epochs ran above; only the diagnostics of the last one are gathered below
                elif multi:   # fused passes -> ONE all-reduce of the flat gradient (RCCL) -> scale + Adam in one launch
                    self._fused_loss_grad(obs, acts, logp_old, rtg, adv, var_f, stats=self._fhist[ep])
                    ctx.all_reduce_sum(self.fp.grad)
                    with torch.no_grad():   # every epoch's norms of the MEAN gradient (two small launches beside an all-reduce)
                        n_a_ = self.fp.module_numel[0]
                        g3 = torch.stack(torch._foreach_norm([self.fp.grad[:n_a_], self.fp.grad[n_a_:], self.fp.grad])) / world
                        multi_gn = g3 if multi_gn is None else multi_gn + g3
                    self._fused_adam(1.0 / world)
                else:
                    self._fused_epoch(obs, acts, logp_old, rtg, adv, var_f, self._fhist[ep])
                if ep == n_ep - 1:
                    h = self._fhist[:n_ep]
                    self.loss_history = h[:, 0:5:4].clone()   # columns 0 (actor loss) and 4 (critic loss)
                    hs = h.sum(0)
                    # multi-GPU: fp.grad holds the all-reduced SUM (the 1 / world scale is inside navppo_adam_step)
                    gn_last = self.fp.grad.norm() / world
                    if not multi and n_ep > 1:
                        # grad norms as the reference logs them -- every epoch's, averaged (ppo.py:351-352, 389-390) -- without a norm
                        # launch per epoch: the fused epoch leaves the squared per-net norms of the epoch BEFORE in columns 3 / 7 of
                        # its statistics row (reduce_adam / resmlp_reduce), the last epoch's come from the gradient buffer
                        fused_gn_sq = h[1:, 3:8:4].clone()                         # [n_ep - 1, (actor, critic)]
                        gn_sum = fused_gn_sq.sum(1).sqrt().sum() + gn_last
                    elif multi_gn is not None:
                        gn_sum = multi_gn[2]
                    else:   # (the pipelined multi-GPU epochs, one epoch: the last epoch's norm stands in)
                        gn_sum = gn_last * n_ep
                    acc = torch.cat([hs[[0, 4, 1, 2]], torch.stack([gn_sum, V0.mean() * n_ep])])
                    a_loss, c_loss = self.loss_history[-1, 0], self.loss_history[-1, 1]
                continue
            a_loss, c_loss, ratios, logp, _ = ppo_losses(self.actor, self.critic, obs, acts, logp_old, rtg, adv, var, cfg.clip)
            self.fp.grad.zero_()
            (a_loss + c_loss).backward()                       # disjoint nets: same grads as the two backward()s of :349,:386
            if multi:
                ctx.all_reduce_sum(self.fp.grad)
                self.fp.grad.div_(world)
            self.opt.step()                                    # ppo.py:381,392
            with torch.no_grad():                              # ppo.py:323-336
                lr_ = logp.detach() - logp_old
                self.loss_history[ep] = torch.stack([a_loss.detach(), c_loss.detach()])
                acc += torch.stack([a_loss.detach(), c_loss.detach(), ((ratios.detach() - 1) - lr_).mean(),
                                    ((ratios.detach() - 1).abs() > cfg.clip).float().mean(),
                                    self.fp.grad.norm(), V0.mean()])
                n_a_ = self.fp.module_numel[0]
                g2 = torch.stack(torch._foreach_norm([self.fp.grad[:n_a_], self.fp.grad[n_a_:]]))
                net_gn = g2 if net_gn is None else net_gn + g2
        acc = acc / max(n_ep, 1)
        if multi:
            ctx.all_reduce_sum(acc)
            acc = acc / world
        n_a = self.fp.module_numel[0]
        d = self.fp.flat - flat_before
        extra = torch.stack(torch._foreach_norm([self.fp.grad[:n_a], self.fp.grad[n_a:], d[:n_a], d[n_a:]]))
        if self.fused and multi:
            extra = extra * extra.new_tensor([1.0 / world, 1.0 / world, 1.0, 1.0])   # norms of the MEAN gradient, as on one GPU
        if fused_gn_sq is not None:   # per-net norms: the mean over the epochs, like grad_norm (the reference's actor_grad_norm / critic_grad_norm)
            extra = torch.cat([(fused_gn_sq.sqrt().sum(0) + extra[:2]) / n_ep, extra[2:]])
        elif multi_gn is not None:
            extra = torch.cat([multi_gn[:2] / n_ep, extra[2:]])
        elif net_gn is not None:
            extra = torch.cat([net_gn / max(n_ep, 1), extra[2:]])
        self.stats = dict(zip(["actor_loss", "critic_loss", "approx_kl", "clip_frac", "grad_norm", "value_mean",
                               "actor_grad_norm", "critic_grad_norm", "actor_param_delta", "critic_param_delta"],
                              [float(v) for v in torch.cat([acc, extra]).tolist()]))   # grad norms: means over the epochs (multi-GPU fused path: the last epoch's)
        self.last_losses = (a_loss.detach(), c_loss.detach())
        return self.stats


# --------------------------------------------------------------------------- trainer
class PPOTrainer:
    """PPO.learn for a ``VecEnv`` shard.  Construct one per process (= per GPU)."""

    def __init__(self, env, cfg=None, ctx=None):
        self.env, self.cfg = env, cfg or PPOConfig()
        self.ctx = ctx
        self.device = env.device
        cfg = self.cfg
        N, T, D = env.N, cfg.rollout_len, env.D
        torch.manual_seed(cfg.seed)  # same initial weights on every rank (then broadcast anyway)
        self.actor, self.critic = nets.make_policy(cfg.policy, D, 2)
        self.actor.to(self.device)
        self.critic.to(self.device)
        self.updater = PPOUpdater(self.actor, self.critic, cfg, ctx, self.device)
        rank = ctx.rank if ctx is not None else 0
        torch.manual_seed(cfg.seed * 1000003 + 17 + rank)  # exploration noise differs per shard
        dev = self.device
        f32, u8 = torch.float32, torch.uint8
        # the observation rows in the dtype the simulator writes them (float16 on an obs_f16 env: BASELINE configs[4]); half rows
        # are read directly by the fused kernels of both policies (round 6: the 512-wide ones too), a PyTorch policy widens them
        self.obs_buf = torch.zeros((T + 1, N, D), dtype=env.sim.obs_dtype, device=dev)
        self._half_obs = env.sim.obs_dtype == torch.float16
        self.act_buf = torch.zeros((T, N, 2), dtype=f32, device=dev)
        self.logp_buf = torch.zeros((T, N), dtype=f32, device=dev)
        self.rew_buf = torch.zeros((T, N), dtype=f32, device=dev)
        self.done_buf = torch.zeros((T, N), dtype=u8, device=dev)
        self.arrive_buf = torch.zeros((T, N), dtype=u8, device=dev)
        self.ended_buf = torch.zeros((T, N), dtype=u8, device=dev)
        self.epret_buf = torch.zeros((T, N), dtype=f32, device=dev)
        self.eplen_buf = torch.zeros((T, N), dtype=torch.int32, device=dev)
        self.eppath_buf = torch.zeros((T, N), dtype=f32, device=dev)
        self.rtg_buf = torch.zeros((T, N), dtype=f32, device=dev)
        self.var = torch.full((), cfg.init_var, dtype=f32, device=dev)  # ppo.py:123-124 (0.8 * I)
        self.var_host = float(cfg.init_var)   # host mirror, refreshed at every rollout start
        self._lo = torch.tensor([0.0, -1.0], device=dev)
        self._hi = torch.tensor([1.0, 1.0], device=dev)
        self._graph = None
        self._step_base = torch.zeros((), dtype=torch.int32, device=dev)  # rollout steps taken so far (noise counter)
        self._act_seed = (cfg.seed * 0x9E3779B97F4A7C15 + 0xAC7) & 0xFFFFFFFFFFFFFFFF
        self._env_id_base = int(env.sim.cfg.env_id_base)
        self.t_so_far = 0     # completed-episode steps, as the reference counts (ppo.py:258)
        self.env_steps = 0    # all simulated steps
        self.i_so_far = 0
        self.episode_starts = 0
        self.logger = {}

    def _fused_act(self, t, noise=None):
        """PPO.get_action for all envs in ONE launch (csrc/ppo_mlp64.hip: mlp64_act, csrc/ppo_resmlp512.hip: resmlp_act)."""
        import ctypes as C
        from ._native import lib
        ptr = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        L = lib()
        rc = getattr(L, self.updater.fused + "_act")(ptr(self.updater.fp.flat), *self.updater._obs_args(self.obs_buf[t]), ptr(noise), self.env.N,
                                                     ptr(self.var), self._act_seed, self._env_id_base, ptr(self._step_base), t,
                                                     ptr(self.act_buf[t]), ptr(self.logp_buf[t]), None,
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"{self.updater.fused}_act failed: {L.navppo_last_error().decode()}")

    # ---- ppo.py:673-706 + env.step, for rollout step t (all envs)
    def _rollout_step(self, t):
        if self.updater.fused:
            self._fused_act(t)
            self.env.sim.step(self.act_buf[t], self.obs_buf[t + 1], self.rew_buf[t], self.done_buf[t], self.arrive_buf[t],
                              self.ended_buf[t], self.epret_buf[t], self.eplen_buf[t], ep_path=self.eppath_buf[t])
            return
        obs = self.obs_buf[t].float()
        mean = self.actor(obs)
        std = torch.sqrt(self.var)
        raw = torch.addcmul(mean, torch.randn_like(mean), std)          # dist.sample(), ppo.py:698
        act = self.act_buf[t]
        torch.clamp(raw, self._lo, self._hi, out=act)                   # ppo.py:700-703
        self.logp_buf[t] = gaussian_log_prob(mean, act, self.var)       # log-prob of the clamped action, :704
        self.env.sim.step(act, self.obs_buf[t + 1], self.rew_buf[t], self.done_buf[t], self.arrive_buf[t],
                          self.ended_buf[t], self.epret_buf[t], self.eplen_buf[t], ep_path=self.eppath_buf[t])

    def _persistent_rollout(self):
        """ppo.py:505-594 in ONE launch (csrc/navsim.hip: rollout_kernel): policy step and env step alternate inside the
        kernel; same device functions and Philox keys as the per-step path, so the buffers come out bit-identical."""
        import ctypes as C
        from ._native import check, lib
        ptr = lambda x: C.c_void_p(x.data_ptr())
        sim = self.env.sim
        entry = lib().navsim_rollout_resmlp512 if self.updater.fused_resmlp512 else lib().navsim_rollout_mlp64
        with torch.cuda.device(self.device):
            check(entry(sim._h, ptr(self.updater.fp.flat), ptr(self.obs_buf), ptr(self.act_buf),
                                             ptr(self.logp_buf), ptr(self.rew_buf), ptr(self.done_buf), ptr(self.arrive_buf),
                                             ptr(self.ended_buf), ptr(self.epret_buf), ptr(self.eplen_buf), ptr(self.eppath_buf),
                                             ptr(self.var), self._act_seed, ptr(self._step_base), self.cfg.rollout_len,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "navsim_rollout_" + ("resmlp512" if self.updater.fused_resmlp512 else "mlp64"))
        self._step_base += self.cfg.rollout_len

    def _rollout_body(self):
        for t in range(self.cfg.rollout_len):
            self._rollout_step(t)
        self._step_base += self.cfg.rollout_len  # inside the captured graph: every replay draws fresh noise

    @property
    def uses_persistent_rollout(self):
        """All T steps of PPO.rollout in ONE launch: navsim_rollout_mlp64 (the (B + 6)-64-64 actor, 10 or 36 beams, float32 or float16
        rows) or navsim_rollout_resmlp512 (the reference's 512-wide actor, 10 beams, float32 or float16 rows).  Anything else runs the
        hipGraph of per-step launches."""
        return bool(self.cfg.persistent_rollout and ((self.updater.fused_mlp64 and self.env.B in (10, 36)) or
                                                     (self.updater.fused_resmlp512 and self.env.B == 10)))

    @torch.no_grad()
    def rollout(self):
        cfg = self.cfg
        self._decay_exploration()
        self.epret_buf.zero_()
        self.eplen_buf.zero_()
        self.updater.invalidate_prepared()   # the kernels below rewrite obs_buf behind torch's version counter
        self.env.sim.reset(self.obs_buf[0])  # ppo.py:486: every batch starts from a reset
        sim = self.env.sim
        # (round 5: both rollout kernels have the tile-box cast of shared 65..4096-segment maps; until then shards up to 4096 envs on
        # such a map took the hipGraph of per-step launches)
        if self.uses_persistent_rollout:
            self._persistent_rollout()
        elif cfg.use_graph and self.device.type == "cuda":
            if self._graph is not None and self._graph_gen != self.env.sim.generation:
                self._graph = None   # set_map / set_spawn_sampler / set_goal_rects re-allocated what the capture froze
            if self._graph is None:
                self._graph_gen = self.env.sim.generation
                s = torch.cuda.Stream(self.device)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):  # warm-up outside capture (allocator, hipBLASLt workspaces)
                    self._rollout_step(0)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                self.env.sim.reset(self.obs_buf[0])
                self._graph = torch.cuda.CUDAGraph()
                # thread_local: other threads (the RCCL watchdog of torch.distributed) may touch the HIP runtime while
                # this thread captures; the default global mode would turn that into a capture error on multi-GPU runs
                with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                    self._rollout_body()
            self._graph.replay()
        else:
            self._rollout_body()
        from .env import gae_scan, rtg_scan
        self._gae = None
        if cfg.gae_lambda is None:
            rtg_scan(self.rew_buf, self.ended_buf, cfg.gamma, out=self.rtg_buf)  # ppo.py:619 -> 643-671
        else:   # extension: the critic's values of the stored observations -> lambda-returns (critic targets) and advantages
            # lambda < 1: the envs still running at the batch end are bootstrapped with V(obs_buf[T]), the observation behind the last
            # row (an env that ended on the last row has ended[T-1] set, which cuts the bootstrap).  lambda = 1 keeps the reference's
            # convention -- the batch end is terminal, ppo.py:601,658-666 -- and stays bit-identical to compute_rtgs.
            T, N, D = cfg.rollout_len, self.env.N, self.env.D
            boot = cfg.gae_lambda < 1.0
            Vall = self.updater.value(self.obs_buf[:T + 1 if boot else T].reshape(-1, D)).reshape(-1, N)
            V = Vall[:T].contiguous()
            adv, ret = gae_scan(self.rew_buf, self.ended_buf, V, cfg.gamma, cfg.gae_lambda, last_value=Vall[T] if boot else None)
            self.rtg_buf.copy_(ret)
            self._gae = (adv.reshape(T * N), V.reshape(T * N))
        self.env_steps += cfg.rollout_len * self.env.N

    def _decay_exploration(self):
        """ppo.py:694-695 multiplies the covariance by 0.995 at every episode start once t_so_far > 50000 while
        it is >= 0.1, with t_so_far frozen during a rollout.  With N envs there are N x more episode starts per
        iteration, so the decay is applied once per N episode starts (per mean episode) -- identical for N=1
        up to being applied at the rollout boundary; documented deviation."""
        cfg = self.cfg
        v = float(self.var)   # the one read-back of the iteration, at its start (the stream is idle here)
        if self.t_so_far > cfg.var_decay_after:
            k = int(round(self.episode_starts / max(self.env.N, 1)))
            for _ in range(k):
                if v >= cfg.var_floor:
                    v *= cfg.var_decay
            self.var.fill_(v)
        self.var_host = v
        self.episode_starts = 0

    def _rollout_metrics_dev(self):
        """The six sums behind the iteration's episode metrics as ONE device tensor (all-reduced over the ranks); nothing
        here waits for the GPU, so the update can be queued behind it."""
        import ctypes as C   # one pass over the five buffers (navppo_episode_sums), deterministic
        from ._native import lib
        ptr = lambda x: C.c_void_p(x.data_ptr())
        if getattr(self, "_sums_ws", None) is None:
            self._sums_ws = torch.empty(256 * 6, dtype=torch.float64, device=self.device)
        m = torch.empty(6, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = lib().navppo_episode_sums(ptr(self.ended_buf), ptr(self.arrive_buf), ptr(self.done_buf), ptr(self.eplen_buf),
                                           ptr(self.epret_buf), self.ended_buf.numel(), ptr(m), ptr(self._sums_ws),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"navppo_episode_sums failed: {lib().navppo_last_error().decode()}")
        if self.ctx is not None:
            self.ctx.all_reduce_sum(m)
        return m

    def _rollout_metrics(self, m=None):
        ep, succ, coll, tmo, steps, ret = [float(x) for x in (self._rollout_metrics_dev() if m is None else m).tolist()]
        self.episode_starts = ep / (self.ctx.world if self.ctx is not None else 1) + self.env.N
        return dict(episodes=int(ep), successes=int(succ), collisions=int(coll), timeouts=int(tmo),
                    completed_steps=int(steps), avg_ep_rews=(ret / ep if ep else 0.0),       # ppo.py:833
                    avg_ep_lens=(steps / ep if ep else 0.0), success_rate=(succ / ep if ep else 0.0))

    def iteration(self):
        """One pass of the loop of PPO.learn (ppo.py:241-459).  Rollout, metric sums and update are queued back to back and
        the host waits once, at the end; rollout / update times come from events on the stream (wall clock on CPU)."""
        cfg = self.cfg
        cuda = self.device.type == "cuda"
        T, N, D = cfg.rollout_len, self.env.N, self.env.D
        t0 = time.time()
        if cuda:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        self.rollout()
        if cuda:
            ev[1].record()
        t1 = time.time()
        m = self._rollout_metrics_dev()
        stats = self.updater.update(self.obs_buf[:T].reshape(T * N, D), self.act_buf.reshape(T * N, 2),
                                    self.logp_buf.reshape(T * N), self.rtg_buf.reshape(T * N),
                                    self.var_host if self.updater.fused else self.var,
                                    **({} if self._gae is None else dict(adv_raw=self._gae[0], V0=self._gae[1])))
        if cuda:
            ev[2].record()
            torch.cuda.synchronize(self.device)
        t2 = time.time()
        if cuda:   # the host ran ahead of the stream: split the wall clock of the iteration by the stream's own stamps
            r_ms, u_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
            t1 = t0 + (t2 - t0) * r_ms / max(r_ms + u_ms, 1e-9)
        metrics = self._rollout_metrics(m)
        self.t_so_far += metrics["completed_steps"]
        self.i_so_far += 1
        world = self.ctx.world if self.ctx is not None else 1
        self.logger = dict(metrics, **stats, iteration=self.i_so_far, t_so_far=self.t_so_far,
                           rollout_time=t1 - t0, update_time=t2 - t1, iter_time=t2 - t0,
                           steps_per_sec=T * N * world / (t2 - t0),                      # ppo.py:855
                           rollout_steps_per_sec=T * N * world / (t1 - t0), var=self.var_host)
        if cfg.output_dir and (self.ctx is None or self.ctx.rank == 0):
            if self.i_so_far % cfg.save_freq == 0:
                self.save_checkpoint()
            import json
            os.makedirs(self.log_dir(), exist_ok=True)
            with open(os.path.join(self.log_dir(), "scalars.jsonl"), "a") as f:
                f.write(json.dumps(dict(self.tb_scalars(), iteration=self.i_so_far, t_so_far=self.t_so_far)) + "\n")
            if cfg.episode_csv_rows:
                self.write_episode_csv(cfg.episode_csv_rows)
            self.write_tensorboard()
        return self.logger

    # ---- logging surface of the reference: per-episode CSV (ppo.py:159-163,739-746) and the TensorBoard scalar names
    #      (ppo.py:892-939) written as one JSON object per iteration (tensorboardX is not a dependency here)
    EPISODE_CSV_HEADER = ["episode", "timestep", "success", "collision", "timeout", "length", "return", "path_length", "time"]

    def log_dir(self):
        return os.path.join(self.cfg.output_dir, self.cfg.method_name, "logs")

    def write_episode_csv(self, max_rows=None):
        """Appends the episodes finished in the last rollout, in (step, env) order (ppo.py:739-746).  path_length is the
        simulator's per-episode accumulator (ppo.py:533-537 semantics: the final step's displacement is not part of it);
        time is the episode's share of the rollout wall clock, length x (rollout_time / T): the envs advance in lock
        step, so an episode of L steps occupied L / T of the rollout."""
        import csv
        os.makedirs(self.log_dir(), exist_ok=True)
        path = os.path.join(self.log_dir(), f"{self.cfg.method_name}_train_episodes.csv")
        new = not os.path.exists(path)
        ended = self.ended_buf.bool()
        t_idx, n_idx = torch.nonzero(ended, as_tuple=True)
        if max_rows is not None:
            t_idx, n_idx = t_idx[:max_rows], n_idx[:max_rows]
        d = self.done_buf[t_idx, n_idx].cpu().numpy()
        a = self.arrive_buf[t_idx, n_idx].cpu().numpy()
        ln = self.eplen_buf[t_idx, n_idx].cpu().numpy()
        rt = self.epret_buf[t_idx, n_idx].cpu().numpy()
        pl = self.eppath_buf[t_idx, n_idx].cpu().numpy()
        tt = t_idx.cpu().numpy()
        sec_per_step = float(self.logger.get("rollout_time", 0.0)) / max(self.cfg.rollout_len, 1)
        base = getattr(self, "_episode_count", 0)
        with open(path, "a", newline="") as f:
            w = csv.writer(f)
            if new:
                w.writerow(self.EPISODE_CSV_HEADER)
            for k in range(len(tt)):
                succ = int(a[k]); coll = int(d[k] and not a[k]); tmo = int(not d[k] and not a[k])  # ppo.py:558-560
                w.writerow([base + k, self.env_steps - (self.cfg.rollout_len - int(tt[k]) - 1) * self.env.N, succ, coll, tmo,
                            int(ln[k]), float(rt[k]), float(pl[k]), float(ln[k]) * sec_per_step])
        self._episode_count = base + len(tt)
        return path

    def tb_scalars(self):
        """The scalars the reference hands to SummaryWriter.add_scalar once per iteration, under its tag names
        (ppo.py:892-918), for the last iteration."""
        lg = self.logger
        ep = max(lg.get("episodes", 0), 1)
        n_ep = self.cfg.n_updates_per_iteration
        var = float(lg.get("var", self.cfg.init_var))
        sec_per_step = float(lg.get("rollout_time", 0.0)) / max(self.cfg.rollout_len, 1)
        return {"train/success_rate": lg.get("success_rate"), "train/collision_rate": lg.get("collisions", 0) / ep,
                "train/timeout_rate": lg.get("timeouts", 0) / ep, "train/mean_return": lg.get("avg_ep_rews"),
                "train/mean_ep_length": lg.get("avg_ep_lens"), "train/mean_ep_time": lg.get("avg_ep_lens", 0.0) * sec_per_step,
                "loss/actor": lg.get("actor_loss"), "loss/critic": lg.get("critic_loss"), "train/timesteps": lg.get("t_so_far"),
                "time/rollout": lg.get("rollout_time"), "time/update": lg.get("update_time"), "time/iteration": lg.get("iter_time"),
                "perf/steps_per_sec": lg.get("steps_per_sec"), "perf/actor_grad_steps": n_ep, "perf/critic_grad_steps": n_ep,
                "ppo/approx_kl": lg.get("approx_kl"),
                "ppo/entropy": 1.0 + LOG_2PI + math.log(max(var, 1e-30)),   # MultivariateNormal(mean, var I).entropy(), 2-D
                "ppo/clip_frac": lg.get("clip_frac"), "ppo/actor_grad_norm": lg.get("actor_grad_norm"),
                "ppo/critic_grad_norm": lg.get("critic_grad_norm"), "ppo/actor_param_delta": lg.get("actor_param_delta"),
                "ppo/critic_param_delta": lg.get("critic_param_delta")}

    def tb_dir(self):
        return os.path.join(self.cfg.output_dir, self.cfg.method_name, "tb")   # ppo.py:66

    def write_tensorboard(self):
        """The reference's TensorBoard stream (ppo.py:892-939): per-iteration scalars at step i_so_far, the per-epoch
        Actor_loss/train and Critic_loss/train series, Episode_Rewards/train (return / length of every finished episode,
        ppo.py:586, capped per iteration like the CSV) and avg_ep_rews/train."""
        from .tb_writer import SummaryWriter
        if getattr(self, "_tb", None) is None:
            self._tb = SummaryWriter(self.tb_dir())
            self._tb_loss_steps = self._tb_ep_steps = 0
        w, it = self._tb, self.i_so_far
        t0 = time.time()
        recs = [(k, float(v), it) for k, v in self.tb_scalars().items() if v is not None]
        hist = self.updater.loss_history.detach().cpu().numpy()
        for k in range(hist.shape[0]):
            recs.append(("Actor_loss/train", float(hist[k, 0]), self._tb_loss_steps + k))
            recs.append(("Critic_loss/train", float(hist[k, 1]), self._tb_loss_steps + k))
        self._tb_loss_steps += hist.shape[0]
        cap = int(self.cfg.tb_episode_rows or 0)
        if cap:   # own cap, independent of the CSV's: with 4096 envs an iteration finishes ~1e5 episodes
            ended = self.ended_buf.bool()
            t_idx, n_idx = torch.nonzero(ended, as_tuple=True)
            t_idx, n_idx = t_idx[:cap], n_idx[:cap]
            per_step = (self.epret_buf[t_idx, n_idx] / self.eplen_buf[t_idx, n_idx].clamp(min=1)).cpu().numpy()
            recs.extend(("Episode_Rewards/train", float(r), self._tb_ep_steps + k) for k, r in enumerate(per_step))
            self._tb_ep_steps += len(per_step)
        recs.append(("avg_ep_rews/train", float(self.logger.get("avg_ep_rews", 0.0)), it))
        recs.append(("time/log", float(getattr(self, "_last_log_time", 0.0)), it))   # the previous iteration's logging cost
        w.add_scalars(recs)   # one TFRecord per point; CRCs vectorised over the records (tb_writer._masked_crc_many)
        w.flush()
        self._last_log_time = time.time() - t0

    def learn(self, total_timesteps, log=print):
        # The reference counts only COMPLETED episodes toward the budget (ppo.py:258); if a configuration never completes
        # one (rollout shorter than the episode cap) its loop would spin forever -- bound the iterations by the budget in
        # simulated steps instead of hanging.
        world = self.ctx.world if self.ctx is not None else 1
        per_iter = self.cfg.rollout_len * self.env.N * world
        max_iters = 4 * (-(-int(total_timesteps) // per_iter)) + 4
        while self.t_so_far < total_timesteps and self.i_so_far < max_iters:  # ppo.py:245
            lg = self.iteration()
            if log and (self.ctx is None or self.ctx.rank == 0):
                log(f"[iter {lg['iteration']:4d}] t={lg['t_so_far']:>10d} mean_ep_rew={lg['avg_ep_rews']:8.2f} "
                    f"succ={lg['success_rate']:.3f} ep_len={lg['avg_ep_lens']:6.1f} a_loss={lg['actor_loss']:.4f} "
                    f"c_loss={lg['critic_loss']:.2f} kl={lg['approx_kl']:.4f} steps/s={lg['steps_per_sec']:.0f} "
                    f"(rollout {lg['rollout_time']:.3f}s update {lg['update_time']:.3f}s)")
        return self.logger

    # ---- ppo.py:452-457 file naming; state_dict keys match the reference's nets for policy resmlp512
    def checkpoint_dir(self):
        return os.path.join(self.cfg.output_dir, self.cfg.method_name, "checkpoints")

    def save_checkpoint(self):
        d = self.checkpoint_dir()
        os.makedirs(d, exist_ok=True)
        tag = f"iter{self.i_so_far:04d}_step{self.t_so_far:08d}.pth"
        pa, pc = os.path.join(d, "actor_" + tag), os.path.join(d, "critic_" + tag)
        torch.save({k: v.detach().cpu().clone() for k, v in self.actor.state_dict().items()}, pa)
        torch.save({k: v.detach().cpu().clone() for k, v in self.critic.state_dict().items()}, pc)
        return pa, pc

    def load_checkpoint(self, actor_path, critic_path):
        """main.py:52-89: state_dicts only (optimiser state and covariance are not part of a checkpoint)."""
        sa = torch.load(actor_path, map_location="cpu")
        sc = torch.load(critic_path, map_location="cpu")
        with torch.no_grad():
            for mod, sd in ((self.actor, sa), (self.critic, sc)):
                own = mod.state_dict()
                missing = set(own) - set(sd)
                if missing:
                    raise KeyError(f"checkpoint lacks keys {sorted(missing)}")
                for k, v in own.items():
                    v.copy_(sd[k])  # in place: parameters stay views of the flat buffer
