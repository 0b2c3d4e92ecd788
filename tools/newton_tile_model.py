#!/usr/bin/env python3
"""Lane-level numpy model of the tile algebra in csrc/newton_kernels.hpp (no GPU needed).

The blocked Cholesky of the model solver keeps every 16 x 16 tile in the *operand layout* of
v_mfma_f64_16x16x4_f64 -- element (r, c) at slot (c >> 2) * 64 + (c & 3) * 16 + r -- and computes the
TRANSPOSED products, so that a tile is loaded as an A / B operand and loaded / stored as an accumulator
with the same `slot = step * 64 + lane` rule.  This script emulates one wavefront's MFMA (A[i = l & 15][k =
l >> 4], B[k = l >> 4][j = l & 15], D[i = (l >> 4) + 4 r][j = l & 15] in register r of lane l) and runs the
factorisation and both triangular solves exactly as the kernel indexes them, against numpy.linalg.
"""
import numpy as np

LANES = np.arange(64)


def mfma(a, b, c):
    """a, b: (64,) operand values per lane; c: (64, 4) accumulator; returns D = A B + C in the same layout."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[LANES & 15, LANES >> 4] = a
    B[LANES >> 4, LANES & 15] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[(LANES >> 4) + 4 * r, LANES & 15]
    return out


def to_op(tile):
    """natural [16][16] -> operand layout [256]"""
    out = np.zeros(256)
    for r in range(16):
        for c in range(16):
            out[(c >> 2) * 64 + (c & 3) * 16 + r] = tile[r, c]
    return out


def from_op(op):
    t = np.zeros((16, 16))
    for r in range(16):
        for c in range(16):
            t[r, c] = op[(c >> 2) * 64 + (c & 3) * 16 + r]
    return t


def tile_off(I, J):
    return (I * (I + 1) // 2 + J) * 256


def factor(F, Dinv, T):
    for J in range(T):
        d = from_op(F[tile_off(J, J):tile_off(J, J) + 256])
        L = np.linalg.cholesky(np.tril(d) + np.tril(d, -1).T)  # (the kernel does this with readlanes; reads the lower triangle only)
        Li = np.linalg.inv(L)
        F[tile_off(J, J):tile_off(J, J) + 256] = to_op(L)
        Dinv[J * 256:(J + 1) * 256] = to_op(Li)
        dl = Dinv[J * 256:(J + 1) * 256]
        for I in range(J + 1, T):  # panel: L_IJ = A_IJ Linv^T, computed transposed
            ap = F[tile_off(I, J):tile_off(I, J) + 256]
            acc = np.zeros((64, 4))
            for s in range(4):
                acc = mfma(dl[s * 64 + LANES], ap[s * 64 + LANES], acc)
            for R in range(4):
                ap[R * 64 + LANES] = acc[:, R]
        for I in range(J + 1, T):  # trailing: A_IK -= L_IJ L_KJ^T, computed transposed
            for K in range(J + 1, I + 1):
                li = F[tile_off(I, J):tile_off(I, J) + 256]
                lk = F[tile_off(K, J):tile_off(K, J) + 256]
                cp = F[tile_off(I, K):tile_off(I, K) + 256]
                acc = np.stack([cp[R * 64 + LANES] for R in range(4)], axis=1)
                for s in range(4):
                    acc = mfma(-lk[s * 64 + LANES], li[s * 64 + LANES], acc)
                for R in range(4):
                    cp[R * 64 + LANES] = acc[:, R]


def tile_matvec(op, v, transposed):
    """(tile or tile^T) @ v the way a wavefront does it: lane (r = l & 15, q = l >> 4) sums its four columns,
    the four quarters are added by two xor shuffles"""
    part = np.zeros(64)
    for s in range(4):
        c = (LANES >> 4) + 4 * s
        r = LANES & 15
        if not transposed:
            part += op[s * 64 + LANES] * v[c]
        else:  # element (row c, col r) of the tile
            part += op[(r >> 2) * 64 + (r & 3) * 16 + c] * v[c]
    tot = part.reshape(4, 16).sum(axis=0)
    return tot


def solve(F, Dinv, T, rhs):
    v = rhs.copy()
    for J in range(T):  # forward
        v[16 * J:16 * J + 16] = tile_matvec(Dinv[J * 256:(J + 1) * 256], v[16 * J:16 * J + 16], False)
        for I in range(J + 1, T):
            v[16 * I:16 * I + 16] -= tile_matvec(F[tile_off(I, J):tile_off(I, J) + 256], v[16 * J:16 * J + 16], False)
    for J in range(T - 1, -1, -1):  # backward
        v[16 * J:16 * J + 16] = tile_matvec(Dinv[J * 256:(J + 1) * 256], v[16 * J:16 * J + 16], True)
        for I in range(J):
            v[16 * I:16 * I + 16] -= tile_matvec(F[tile_off(J, I):tile_off(J, I) + 256], v[16 * J:16 * J + 16], True)
    return v


def main():
    rng = np.random.default_rng(0)
    for m in (5, 16, 37, 96):
        T = (m + 15) // 16
        mp = 16 * T
        A = rng.standard_normal((3 * m, m))
        H = np.eye(mp)
        H[:m, :m] = A.T @ A / (3 * m) + 0.1 * np.eye(m)
        F = np.zeros(T * (T + 1) // 2 * 256)
        for I in range(T):
            for J in range(I + 1):
                F[tile_off(I, J):tile_off(I, J) + 256] = to_op(H[16 * I:16 * I + 16, 16 * J:16 * J + 16])
        Dinv = np.zeros(T * 256)
        factor(F, Dinv, T)
        L = np.zeros((mp, mp))
        for I in range(T):
            for J in range(I + 1):
                L[16 * I:16 * I + 16, 16 * J:16 * J + 16] = from_op(F[tile_off(I, J):tile_off(I, J) + 256])
        err_f = np.max(np.abs(L - np.linalg.cholesky(H)))
        rhs = rng.standard_normal(mp)
        x = solve(F, Dinv, T, rhs)
        err_s = np.max(np.abs(x - np.linalg.solve(H, rhs)))
        print(f"m = {m:3d}: factor err {err_f:.1e}, solve err {err_s:.1e}")
        assert err_f < 1e-12 and err_s < 1e-10


if __name__ == "__main__":
    main()
