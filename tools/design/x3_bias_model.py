"""Design study (CPU, numpy): the split-bf16 forward of the 16-64-64 actor under a model of the bf16 MFMA measured on gfx950
(tools/ubench/bf16_mfma_rounding.hip: every product and the accumulator aligned to 2^-3 ulp of the largest addend, RNE, one final RNE),
against float64 -- which element of the scheme leaves a MEAN error in the heads pre-activations.  Answer: adding the bias to the rounded
product (rounds 4-5); with the accumulator starting as the bias the mean error drops 20-fold at unchanged rms.
usage: python tools/design/x3_bias_model.py"""
import sys, os, copy
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_bf16x3 import _batch, _nets
def bf16_rne(x):
    x = np.asarray(x, np.float32); u = x.view(np.uint32).astype(np.uint64)
    return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
def split3(x):
    p0 = bf16_rne(x); r = (x - p0).astype(np.float32); p1 = bf16_rne(r); s = (r - p1).astype(np.float32); return p0, p1, bf16_rne(s)
def mfma(acc, A, B, guard=3, mode="rne"):
    """acc [M,N] f32; A [M,16], B [16,N]: align every addend to 2^(emax-23-guard), round each, sum, RNE to f32"""
    prods = A.astype(np.float64)[:, :, None] * B.astype(np.float64)[None, :, :]          # [M,16,N]
    allv = np.concatenate([acc.astype(np.float64)[:, None, :], prods], 1)
    emax = np.floor(np.log2(np.maximum(np.abs(allv).max(1), 1e-300)))
    q = np.exp2(emax - 23 - guard)[:, None, :]
    r = np.rint(allv / q) if mode == "rne" else np.trunc(allv / q)
    return (r * q).sum(1).astype(np.float32)
def x3_product(W, Hs, mode):   # W [M,K] f32 weights, Hs [K,N] activations -> [M,N], small terms of all k-steps first, then the big ones
    Wp, Hp = split3(W), split3(Hs)
    acc = np.zeros((W.shape[0], Hs.shape[1]), np.float32)
    K = W.shape[1]
    for ks in range(K // 16):
        sl = slice(16 * ks, 16 * ks + 16)
        for (i, j) in [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1)]:
            acc = mfma(acc, Wp[i][:, sl], Hp[j][sl, :], mode=mode)
    for ks in range(K // 16):
        sl = slice(16 * ks, 16 * ks + 16)
        acc = mfma(acc, Wp[0][:, sl], Hp[0][sl, :], mode=mode)
    return acc
def f32_chain(W, Hs):
    acc = np.zeros((W.shape[0], Hs.shape[1]), np.float32)
    for k in range(W.shape[1]):
        acc = (acc.astype(np.float64) + W[:, k:k+1].astype(np.float64) * Hs[k:k+1, :].astype(np.float64)).astype(np.float32)
    return acc
dev = "cpu"
N = 4096
for seed in range(3):
    a, c = _nets(torch.device("cpu"), seed=3 + seed)
    obs = _batch(38407, 100 + 38407 + seed, torch.device("cpu"))[0][:N]
    ps = [p.detach().numpy() for p in a.parameters()]
    W1, b1, W2, b2, w3, b3, w4, b4 = ps
    X = obs.numpy().T.copy()                       # [16, N]
    def heads(H2):
        outs = []
        for w, b in ((w3, b3), (w4, b4)):
            z = np.zeros(N, np.float32)
            # kernel order: lane half hi sums rows (r&3)+8(r>>2)+4hi over r of tile t = 0,1; then the halves are added, then the bias
            zh = []
            for hi in range(2):
                zz = np.zeros(N, np.float32)
                for t in range(2):
                    for r in range(16):
                        u = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi
                        zz = (zz.astype(np.float64) + H2[u].astype(np.float64) * np.float64(w[0, u])).astype(np.float32)
                zh.append(zz)
            outs.append(((zh[0] + zh[1]).astype(np.float32) + b[0]).astype(np.float32))
        return outs
    # float64 truth
    H1t = np.maximum(W1.astype(np.float64) @ X.astype(np.float64) + b1.astype(np.float64)[:, None], 0)
    H2t = np.maximum(W2.astype(np.float64) @ H1t + b2.astype(np.float64)[:, None], 0)
    zt = [(w.astype(np.float64) @ H2t)[0] + b.astype(np.float64)[0] for w, b in ((w3, b3), (w4, b4))]
    for name, prod in (("f32 chain", lambda W, H: f32_chain(W, H)), ("x3 rne   ", lambda W, H: x3_product(W, H, "rne")), ("x3 trunc ", lambda W, H: x3_product(W, H, "trunc"))):
        H1 = np.maximum((prod(W1, X) + b1[:, None]).astype(np.float32), 0)
        H2 = np.maximum((prod(W2, H1) + b2[:, None]).astype(np.float32), 0)
        z = heads(H2)
        print(f"seed {seed} {name}: z3 err rms {np.sqrt(((z[0]-zt[0])**2).mean()):.2e} mean {np.mean(z[0]-zt[0]):+.2e}   z4 err rms {np.sqrt(((z[1]-zt[1])**2).mean()):.2e} mean {np.mean(z[1]-zt[1]):+.2e}")
print("---- variants of the split scheme (model), mean z error over 3 seeds x 4096 samples")
def x3_var(W, Hs, guard=3, terms=((2,0),(1,1),(0,2),(1,0),(0,1)), big_last=True, bias=None):
    Wp, Hp = split3(W), split3(Hs)
    acc = np.zeros((W.shape[0], Hs.shape[1]), np.float32) if bias is None else np.repeat(bias[:, None], Hs.shape[1], 1).astype(np.float32)
    K = W.shape[1]
    order = list(terms) + ([] if big_last else [(0, 0)])
    for ks in range(K // 16):
        sl = slice(16 * ks, 16 * ks + 16)
        for (i, j) in order:
            acc = mfma(acc, Wp[i][:, sl], Hp[j][sl, :], guard=guard)
    if big_last:
        for ks in range(K // 16):
            sl = slice(16 * ks, 16 * ks + 16)
            acc = mfma(acc, Wp[0][:, sl], Hp[0][sl, :], guard=guard)
    return acc
import itertools
variants = {"x3 as built": dict(), "exact inner sums": dict(guard=30), "nine products": dict(terms=((2,2),(2,1),(1,2),(2,0),(1,1),(0,2),(1,0),(0,1))),
            "big term inside each k-step": dict(big_last=False), "bias as the initial accumulator": "bias"}
for name, kw in variants.items():
    out = []
    for seed in range(3):
        a, c = _nets(torch.device("cpu"), seed=3 + seed)
        obs = _batch(38407, 100 + 38407 + seed, torch.device("cpu"))[0][:N]
        W1, b1, W2, b2, w3, b3, w4, b4 = [p.detach().numpy() for p in a.parameters()]
        X = obs.numpy().T.copy()
        H1t = np.maximum(W1.astype(np.float64) @ X.astype(np.float64) + b1.astype(np.float64)[:, None], 0)
        H2t = np.maximum(W2.astype(np.float64) @ H1t + b2.astype(np.float64)[:, None], 0)
        if kw == "bias":
            H1 = np.maximum(x3_var(W1, X, bias=b1), 0); H2 = np.maximum(x3_var(W2, H1, bias=b2), 0)
        else:
            H1 = np.maximum((x3_var(W1, X, **kw) + b1[:, None]).astype(np.float32), 0)
            H2 = np.maximum((x3_var(W2, H1, **kw) + b2[:, None]).astype(np.float32), 0)
        # error of H2 itself and of exact heads applied to it (the head's own rounding left out)
        for w, b in ((w3, b3), (w4, b4)):
            zk = (w.astype(np.float64) @ H2.astype(np.float64))[0]; zt = (w.astype(np.float64) @ H2t)[0]
            out.append((np.mean(zk - zt), np.sqrt(np.mean((zk - zt) ** 2))))
    print(f"{name:34s} mean z error (exact head on the kernel's H2): " + " ".join(f"{m:+.1e}" for m, r in out) + "   rms " + " ".join(f"{r:.1e}" for m, r in out))
# the f32 chain the same way
out = []
for seed in range(3):
    a, c = _nets(torch.device("cpu"), seed=3 + seed)
    obs = _batch(38407, 100 + 38407 + seed, torch.device("cpu"))[0][:N]
    W1, b1, W2, b2, w3, b3, w4, b4 = [p.detach().numpy() for p in a.parameters()]
    X = obs.numpy().T.copy()
    H1t = np.maximum(W1.astype(np.float64) @ X.astype(np.float64) + b1.astype(np.float64)[:, None], 0)
    H2t = np.maximum(W2.astype(np.float64) @ H1t + b2.astype(np.float64)[:, None], 0)
    H1 = np.maximum((f32_chain(W1, X) + b1[:, None]).astype(np.float32), 0)
    H2 = np.maximum((f32_chain(W2, H1) + b2[:, None]).astype(np.float32), 0)
    for w, b in ((w3, b3), (w4, b4)):
        zk = (w.astype(np.float64) @ H2.astype(np.float64))[0]; zt = (w.astype(np.float64) @ H2t)[0]
        out.append((np.mean(zk - zt), np.sqrt(np.mean((zk - zt) ** 2))))
print(f"{'f32 fma chain':34s} mean z error (exact head on the kernel's H2): " + " ".join(f"{m:+.1e}" for m, r in out) + "   rms " + " ".join(f"{r:.1e}" for m, r in out))
