"""CPU study (numpy, float32 arithmetic with float64 references): rounding error of Winograd variants for the U-Net's layer
shapes, to decide which algorithms are worth a kernel (VERDICT r03 item 6 gates: <= 1e-4 of peak per layer, frames within 1e-5
of the direct path).  Weights are transformed in float64 and rounded once (as pack_weight_wino_kernel does); input transform,
channel accumulation and output transform run in float32 (accumulation in chunks of 2 channels like v_mfma_f32_32x32x2_f32:
sequential float32 adds).  Reports max and rms error against a float64 direct convolution, relative to the output rms.
Usage: python scripts/experiments/winograd_accuracy_study.py [C [HW]]"""
import sys
import numpy as np

def cook_toom(points, m, r):
    """Winograd / Cook-Toom matrices AT [m, n], G [n, r], BT [n, n] for F(m, r) with n = m + r - 1 points (last = infinity)."""
    from fractions import Fraction as Fr
    n = m + r - 1
    pts = [Fr(p) for p in points]            # n - 1 finite points
    assert len(pts) == n - 1
    # Lagrange basis denominators
    def poly_mul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    M = [Fr(1)]
    for p in pts:
        M = poly_mul(M, [-p, Fr(1)])          # prod (x - p)
    AT = [[(pts[j] ** i if j < n - 1 else (Fr(1) if i == m - 1 else Fr(0))) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        d = Fr(1)
        for k in range(n - 1):
            if k != j:
                d *= (pts[j] - pts[k])
        G.append([pts[j] ** i / d for i in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    BT = []
    for j in range(n - 1):
        # M(x) / (x - p_j) coefficients
        q = [Fr(1)]
        for k in range(n - 1):
            if k != j:
                q = poly_mul(q, [-pts[k], Fr(1)])
        BT.append(q + [Fr(0)] * (n - len(q)))
    BT.append(list(M) + [Fr(0)] * (n - len(M)))
    f = lambda A: np.array([[float(x) for x in row] for row in A], dtype=np.float64)
    return f(AT), f(G), f(BT)

def check(AT, G, BT, m, r):
    rng = np.random.RandomState(0)
    d, g = rng.randn(m + r - 1), rng.randn(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-9), (y, ref)

def wino_conv(x, w, AT, G, BT, m, r):
    """x [C, H, W] (H, W multiples of m, 'valid' correlation on a pre-padded input of size H + r - 1), w [K, C, r, r]."""
    C, Hp, Wp = x.shape
    K = w.shape[0]
    n = m + r - 1
    H, W = Hp - r + 1, Wp - r + 1
    U = np.einsum('ia,kcab,jb->ijkc', G, w.astype(np.float64), G).astype(np.float32)         # float64, rounded once
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    ty, tx = H // m, W // m
    # input tiles [ty, tx, C, n, n]
    tiles = np.empty((ty, tx, C, n, n), np.float32)
    for i in range(ty):
        for j in range(tx):
            tiles[i, j] = x[:, i * m:i * m + n, j * m:j * m + n]
    # float32 transforms, one matrix product at a time (each product rounds like the kernel's add chains)
    V = np.einsum('ia,yxcab->yxcib', BT32, tiles, dtype=np.float32)
    V = np.einsum('jb,yxcib->yxcij', BT32, V, dtype=np.float32)
    # accumulate over channels in float32, two channels per step (sequential adds)
    Mm = np.zeros((ty, tx, K, n, n), np.float32)
    for c in range(0, C, 2):
        part = np.einsum('ijkc,yxcij->yxkij', U[:, :, :, c:c + 2], V[:, :, c:c + 2], dtype=np.float32)
        Mm = (Mm + part).astype(np.float32)
    Y = np.einsum('ai,yxkij->yxkaj', AT32, Mm, dtype=np.float32)
    Y = np.einsum('bj,yxkaj->yxkab', AT32, Y, dtype=np.float32)
    out = Y.transpose(2, 0, 3, 1, 4).reshape(K, H, W)
    return out

def direct_conv(x, w, dtype):
    C, Hp, Wp = x.shape
    K, _, r, _ = w.shape
    H, W = Hp - r + 1, Wp - r + 1
    out = np.zeros((K, H, W), dtype)
    xx, ww = x.astype(dtype), w.astype(dtype)
    for c in range(0, C, 2):            # two channels per step, taps inside (the direct kernel's order: taps x chunk)
        part = np.zeros((K, H, W), dtype)
        for a in range(r):
            for b in range(r):
                part += np.einsum('kc,chw->khw', ww[:, c:c + 2, a, b], xx[c:c + 2, a:a + H, b:b + W]).astype(dtype)
        out = (out + part).astype(dtype)
    return out

if __name__ == '__main__':
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    HW = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    K = 32
    rng = np.random.RandomState(1)
    variants = [
        ('F(2x2,3x3) pts 0,1,-1          (shipped)', 2, 3, [0, 1, -1]),
        ('F(4x4,3x3) pts 0,1,-1,2,-2     (Lavin)', 4, 3, [0, 1, -1, 2, -2]),
        ('F(4x4,3x3) pts 0,1,-1,1/2,-1/2', 4, 3, [0, 1, -1, '1/2', '-1/2']),
        ('F(4x4,3x3) pts 0,1,-1,1/2,-2', 4, 3, [0, 1, -1, '1/2', -2]),
        ('F(3x3,3x3) pts 0,1,-1,2', 3, 3, [0, 1, -1, 2]),
        ('F(3x3,3x3) pts 0,1,-1,1/2', 3, 3, [0, 1, -1, '1/2']),
        ('F(2x2,2x2) pts 0,-1            (shipped, stride-2 layers)', 2, 2, [0, -1]),
        ('F(3x3,2x2) pts 0,1,-1', 3, 2, [0, 1, -1]),
        ('F(4x4,2x2) pts 0,1,-1,2', 4, 2, [0, 1, -1, 2]),
        ('F(4x4,2x2) pts 0,1,-1,1/2', 4, 2, [0, 1, -1, '1/2']),
    ]
    print('C = %d input channels, %d x %d outputs, %d output channels; activations ~ |N(0,1)| after ReLU-like, weights N(0, 1/(C r^2))' % (C, HW, HW, K))
    for name, m, r, pts in variants:
        AT, G, BT = cook_toom(pts, m, r)
        check(AT, G, BT, m, r)
        hw = (HW // m) * m
        x = np.maximum(rng.randn(C, hw + r - 1, hw + r - 1), 0.2 * rng.randn(C, hw + r - 1, hw + r - 1)).astype(np.float32)
        w = (rng.randn(K, C, r, r) / np.sqrt(C * r * r)).astype(np.float32)
        ref = direct_conv(x, w, np.float64)
        e_w = wino_conv(x, w, AT, G, BT, m, r).astype(np.float64) - ref
        e_d = direct_conv(x, w, np.float32).astype(np.float64) - ref
        rms = np.sqrt((ref ** 2).mean())
        print('%-60s max %.2e  rms %.2e   | direct f32: max %.2e rms %.2e   ratio(max) %.1f  mults/output %.2f' % (
            name, np.abs(e_w).max() / rms, np.sqrt((e_w ** 2).mean()) / rms, np.abs(e_d).max() / rms,
            np.sqrt((e_d ** 2).mean()) / rms, np.abs(e_w).max() / np.abs(e_d).max(), (m + r - 1) ** 2 / m ** 2))
