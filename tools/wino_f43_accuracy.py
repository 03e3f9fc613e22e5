"""Float32 error of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct form (numpy, CPU; no GPU needed):
`python tools/wino_f43_accuracy.py`.  Planning evidence for DESIGN.md section 8, item 0b: is F(4x4,3x3)'s error small
enough for the 1e-5 loss bar?  One 3x3 "same" convolution, Cin -> Cout channels on a 16x16 map, inputs ~ N(0,1) like
BatchNorm-ed activations, weights ~ U(-1,1)/sqrt(9 Cin); every transform, the position products (accumulated over Cin in
float32, as the MFMA does) and the output transform in float32; reference: the direct form in float64."""
import numpy as np

BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
              np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def direct(x, w):  # x [Cin][H][W], w [Cout][Cin][3][3], float64
    cin, H, W = x.shape
    xp = np.zeros((cin, H + 2, W + 2)); xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W))
    for a in range(3):
        for b in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, a, b], xp[:, a:a + H, b:b + W])
    return out


def wino(x, w, m, BT, G, AT, dt):
    t = m + 2
    cin, H, W = x.shape
    xp = np.zeros((cin, H + 2, W + 2), dt); xp[:, 1:-1, 1:-1] = x.astype(dt)
    BTf, Gf, ATf = BT.astype(dt), G.astype(dt), AT.astype(dt)
    U = np.einsum("ia,ocab,jb->ocij", Gf, w.astype(dt), Gf).astype(dt)       # [Cout][Cin][t][t]
    out = np.zeros((w.shape[0], H, W), dt)
    for i0 in range(0, H, m):
        for j0 in range(0, W, m):
            d = xp[:, i0:i0 + t, j0:j0 + t]
            V = np.einsum("ia,cab,jb->cij", BTf, d, BTf).astype(dt)
            M = np.zeros((w.shape[0], t, t), dt)
            for c in range(cin):  # float32 accumulation over the channels, in order
                M += U[:, c] * V[c][None]
            out[:, i0:i0 + m, j0:j0 + m] = np.einsum("ia,oab,jb->oij", ATf, M, ATf).astype(dt)
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for cin, cout in ((64, 128), (128, 512)):
        x = rng.standard_normal((cin, 16, 16))
        w = rng.uniform(-1, 1, (cout, cin, 3, 3)) / np.sqrt(9 * cin)
        ref = direct(x, w)
        scale = np.abs(ref).max()
        d32 = direct(x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64))
        rows = (("direct, float32 inputs (float64 sums)", d32),
                ("F(2x2,3x3) float32", wino(x, w, 2, BT2, G2, AT2, np.float32).astype(np.float64)),
                ("F(4x4,3x3) float32", wino(x, w, 4, BT4, G4, AT4, np.float32).astype(np.float64)))
        print(f"Cin={cin} Cout={cout}: max|out| = {scale:.3f}")
        for name, y in rows:
            e = y - ref
            print(f"  {name:40s} max |err| / max|out| = {np.abs(e).max() / scale:.2e}   rms err / rms out = "
                  f"{np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean()):.2e}")
