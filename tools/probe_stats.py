import numpy as np
def hash_slot(key, L):  # common.h: top L bits of key * 0x9E3779B1 (32-bit)
    return ((key.astype(np.uint64) * 0x9E3779B1) & 0xffffffff) >> (32 - L)
def lean_slot(key, L):  # lean.h: bits [24-L, 24) of (key & 0xffffff) * 0x9E3779 (low 32 bits)
    p = ((key.astype(np.uint64) & 0xffffff) * 0x9E3779) & 0xffffffff
    return (p >> (24 - L)) & ((1 << L) - 1)
def probes(keys, L, fn):
    T = 1 << L; tab = np.full(T, -1, dtype=np.int64); tot = 0; mx = 0
    for k, h in zip(keys, fn(keys, L)):
        h = int(h); n = 1
        while tab[h] != -1 and tab[h] != k:
            h = (h + 1) & (T - 1); n += 1
        tab[h] = k; tot += n; mx = max(mx, n)
    return tot / len(keys), mx
rng = np.random.default_rng(0)
def stencil_row(n, c):
    i, j, k = c
    out = []
    for dk in range(-2, 3):
        for dj in range(-2, 3):
            for di in range(-2, 3):
                a, b, d = i + di, j + dj, k + dk
                if 0 <= a < n and 0 <= b < n and 0 <= d < n: out.append(a + n * b + n * n * d)
    return np.array(out)
def pow2_ceil(v): return 1 << int(np.ceil(np.log2(max(v, 1))))
cases = {}
rows = [stencil_row(100, tuple(rng.integers(2, 98, 3))) for _ in range(300)]
cases["stencil 100^3 (125 nnz, T=256)"] = rows
# FEM 3-dof brick rows: node (i,j,k) couples 5x5x5 nodes x 3 dof, 9x9x257
def brick_row(c):
    i, j, k = c; out = []
    for dk in range(-2, 3):
        for dj in range(-2, 3):
            for di in range(-2, 3):
                a, b, d = i + di, j + dj, k + dk
                if 0 <= a < 9 and 0 <= b < 9 and 0 <= d < 257:
                    nd = a + 9 * b + 81 * d
                    out += [3 * nd, 3 * nd + 1, 3 * nd + 2]
    return np.array(out)
cases["brick 3-dof (375 nnz, T=1024)"] = [brick_row((4, 4, int(rng.integers(2, 255)))) for _ in range(100)]
cases["random cols in 1M (n=100, T=256)"] = [rng.choice(1_000_000, 100, replace=False) for _ in range(300)]
cases["multiples of 1024 (R-MAT-like, n=150, T=256)"] = [rng.choice(4096, 150, replace=False) * 1024 for _ in range(300)]
cases["consecutive run (n=170, T=256)"] = [np.arange(170) + int(rng.integers(0, 1 << 20)) for _ in range(200)]
cases["cols >= 2^24 differing above bit 23 (n=100,T=256)"] = [(rng.choice(200, 100, replace=False).astype(np.int64) << 24) + 5 for _ in range(50)]
print("%-52s %22s %22s" % ("rows", "hash_slot avg / max", "lean_slot avg / max"))
for name, rows in cases.items():
    res = []
    for fn in (hash_slot, lean_slot):
        a = [probes(r[rng.permutation(len(r))], int(np.log2(max(64, pow2_ceil(len(r) + len(r) // 2)))), fn) for r in rows]
        res.append((np.mean([x[0] for x in a]), max(x[1] for x in a)))
    print("%-52s %14.3f / %4d %14.3f / %4d" % (name, res[0][0], res[0][1], res[1][0], res[1][1]))
