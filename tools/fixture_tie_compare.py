"""Compares /tmp/tie_base.npz and /tmp/tie_new.npz (tools/fixture_tie_probe.py): per encoder layer the pooling windows whose
arg-max differs between the two runs, the smallest relative gaps between a window's two largest values, the largest
difference of z between the runs."""
import numpy as np
a = dict(np.load("/tmp/tie_base.npz")); b = dict(np.load("/tmp/tie_new.npz"))
for l in (1, 2, 3):
    za, zb = a[f"z{l}"], b[f"z{l}"]
    N, C, H, W = za.shape
    wa = za.reshape(N, C, H // 2, 2, W // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(N, C, H // 2, W // 2, 4)
    wb = zb.reshape(N, C, H // 2, 2, W // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(N, C, H // 2, W // 2, 4)
    aa, ab = wa.argmax(-1), wb.argmax(-1)
    diff = np.argwhere(aa != ab)
    srt = np.sort(wa, -1); gap = (srt[..., 3] - srt[..., 2]); rel = gap / (np.abs(srt[..., 3]) + 1e-30)
    nz = rel[gap > 0]
    print("layer", l, "windows", aa.size, "argmax differs in", len(diff), "| exact ties", int((gap == 0).sum()), "| smallest nonzero rel gaps", np.sort(nz)[:5], "| max |dz|", np.abs(za - zb).max(), "| lrelu sign flips", int(((wa.max(-1) > 0) != (wb.max(-1) > 0)).sum()))
    for d in diff[:5]:
        print("   ", tuple(d), wa[tuple(d)], wb[tuple(d)])
