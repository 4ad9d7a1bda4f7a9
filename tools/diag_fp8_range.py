"""Diagnostic (GPU): per-row error of the fp8 forward on inputs with a dominant early key, against float64 on the dequantised
inputs, for one or more builds (FA_MI355_LIB-style paths given as arguments).  Prints the error binned by how far the row's first-
tile maximum stands above the row's mean score."""
import math
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def case(kind):
    g = torch.Generator().manual_seed(5)
    Bn, Hh, Sn, Dh = 1, 2, 1024, 128
    if kind == "outlier":
        qf, kf, vf = (torch.randn(Bn, Hh, Sn, Dh, generator=g) for _ in range(3))
        kf[:, :, 5] = qf[:, :, 1000] * 6.0
    else:
        nats = float(kind)
        u = torch.randn(Dh, generator=g)
        u *= math.sqrt(Dh) / u.norm()
        qf = torch.randn(Bn, Hh, Sn, Dh, generator=g) + u
        kf, vf = (torch.randn(Bn, Hh, Sn, Dh, generator=g) for _ in range(2))
        kf[:, :, 0] = u * (nats / math.sqrt(Dh))
    f32 = [qf, kf, vf]
    ds = tuple(float(t.abs().max()) / 448.0 for t in f32)
    return [(t / s).to(torch.float8_e4m3fn) for t, s in zip(f32, ds)], ds


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        import os
        os.environ["FA_MI355_LIB"] = sys.argv[2]
        import flash_attention_impls_amd as fa
        for kind in sys.argv[3:]:
            (q, k, v), ds = case(kind)
            o = fa.flash_attn(q.cuda(), k.cuda(), v.cuda(), False, descale=ds).double().cpu().numpy()
            deq = [t.double().numpy() * s for t, s in zip((q, k, v), ds)]
            s = np.einsum("bhid,bhjd->bhij", deq[0], deq[1]) / math.sqrt(128) * math.log2(math.e)
            w = np.exp2(s - s.max(-1, keepdims=True)); w /= w.sum(-1, keepdims=True)
            ref = np.einsum("bhij,bhjd->bhid", w, deq[2])
            err = np.linalg.norm(o - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-3)
            gap = s[..., :16].max(-1) - s[..., :16].mean(-1)
            over = s.max(-1) - s[..., :16].max(-1)
            print(f"[{os.path.basename(sys.argv[2])}] {kind}: relF {np.linalg.norm(o - ref) / np.linalg.norm(ref):.4f}")
            for lo, hi in [(0, 4), (4, 6), (6, 8), (8, 10), (10, 12), (12, 14), (14, 17), (17, 25), (25, 200)]:
                sel = (gap >= lo) & (gap < hi)
                if sel.any():
                    print(f"   gap [{lo:3d},{hi:3d}) rows {sel.sum():5d}  err median {np.median(err[sel]):.4f} max {err[sel].max():.4f}"
                          f"   max later excess {over[sel].max():.1f}   wg rows hit: {np.unique(np.nonzero(sel)[2] // 256)}")
            wg = err.reshape(1, 2, 4, 256)
            print("   per workgroup (head, 256-row block) median err:", np.round(np.median(wg, -1), 4).tolist())
        return
    kinds = ["outlier", "8", "10"]
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--child", lib] + kinds, check=False)


main()
