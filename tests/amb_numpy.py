"""Second, independent restatement of the reference's CSR -> AMB conversion (numpy), written from
cuda-c/src/conversion/convert_amb.cu kernel by kernel, NOT from oracle/nsparse_oracle.c.  Test
infrastructure: tests/test_oracle_golden.py holds the C oracle against it, so that the AMB half of the
oracle no longer rests on one implementation plus format invariants.  Small inputs only (python loops).

Stages and the reference lines they restate:
  segment()   set_segmented_nnz_num / set_segmented_col_val (:138-206): virtual row g*pad_M + i
  sigma_sort  the sort loop of convert_amb_at (:667-696): thrust::stable_sort_by_key, greater<int>,
              per segment and per window [start, min(start + SIGMA, M)) of the REAL rows
  ell()       set_cl / init_cs + scan (:46-102), set_sellcs_col_val (:104-136)
  compress()  get_c_size, set_ushort_col, packed index (:301-386), update / compress / split of the
              write permutation (:253-297)
  block()     set_blocked_cl (:388-429), init_blocked_cs + scan (:431-471), set_blocked_col_val (:473-525)
"""
import numpy as np

USHORT_MAX, SCL_BORDER, SCL_BIT = 65536, 16, 0xFFFF


def csr2amb(A, seg_size=65536, block_size=1, chunk=32, sigma=32768):
    M, N = A["M"], A["N"]
    rpt, col, val = np.asarray(A["rpt"]), np.asarray(A["col"]), np.asarray(A["val"])
    seg_size = min(max(int(seg_size), 1), USHORT_MAX)              # set_plan clamps (nsparse.cu:176-187)
    if not 1 <= block_size <= 20:
        block_size = 1
    pad_M = chunk * -(-M // chunk)
    G = max(1, -(-N // seg_size))
    R = pad_M * G
    # ---- segment(): entries of row i with col // seg_size == g, in storage order
    rows = [[] for _ in range(R)]
    for i in range(M):
        for p in range(rpt[i], rpt[i + 1]):
            rows[(col[p] // seg_size) * pad_M + i].append((int(col[p]), val[p]))
    length = np.array([len(r) for r in rows], dtype=np.int64)
    # ---- sigma_sort(): stable, descending by length, windows over the real rows of every segment
    perm = np.arange(R)
    sg = min(sigma, M)
    if sg > 1:
        for g in range(G):
            for start in range(0, M, sg):
                end = min(start + sg, M)
                lo = g * pad_M + start
                order = np.argsort(-length[lo:lo + end - start], kind="stable")
                perm[lo:lo + end - start] = perm[lo:lo + end - start][order]
                length[lo:lo + end - start] = length[lo:lo + end - start][order]
    # ---- ell(): chunk widths, column-major fill, padding = value 0 and the column the chunk's first
    #      row has at that position
    nchunk = R // chunk
    width = length.reshape(nchunk, chunk).max(axis=1)
    keep = np.flatnonzero(width > 0)                                # compress(): non-empty chunks
    c_size = len(keep)
    out = dict(M=M, N=N, pad_M=pad_M, chunk=chunk, seg_size=seg_size, seg_num=G, c_size=c_size,
               block_size=block_size)
    lanes_col, lanes_val, pcl = [], [], []
    wp = np.zeros(c_size * chunk, dtype=np.int32)
    for k, c in enumerate(keep):
        W = int(width[c])
        first = rows[perm[c * chunk]]
        cc = np.zeros((chunk, W), dtype=np.int64)
        vv = np.zeros((chunk, W), dtype=np.asarray(val).dtype)
        for t in range(chunk):
            own = rows[perm[c * chunk + t]]
            for j in range(W):
                if j < len(own):
                    cc[t, j], vv[t, j] = own[j]
                else:
                    cc[t, j] = first[j][0]
            v = c * chunk + t
            wp[k * chunk + t] = perm[v] - (v // pad_M) * pad_M     # update_write_permutation
        seg = int(cc[0, 0] // seg_size)                            # set_ushort_col: first column of the chunk
        pcl.append((W - 1) | (seg << SCL_BORDER))
        lanes_col.append((cc % seg_size).astype(np.int64))
        lanes_val.append(vv)
    out["write_permutation"] = wp
    out["s_write_permutation"] = (wp % USHORT_MAX).astype(np.uint16)
    out["s_write_permutation_offset"] = (wp.reshape(c_size, chunk)[:, 0] // USHORT_MAX).astype(np.uint16) \
        if c_size else np.zeros(0, np.uint16)
    # ---- block()
    bs = block_size
    cl = np.zeros(c_size, dtype=np.uint32)
    nblks = []
    for k in range(c_size):
        W = (pcl[k] & SCL_BIT) + 1
        mx = 0
        for t in range(chunk):
            base, blocks = lanes_col[k][t, 0], 1
            for j in range(1, W):
                if lanes_col[k][t, j] - base >= bs:
                    base, blocks = lanes_col[k][t, j], blocks + 1
            mx = max(mx, blocks)
        nblks.append(mx)
        cl[k] = (mx - 1) | ((pcl[k] >> SCL_BORDER) << SCL_BORDER)
    cs = np.zeros(c_size, dtype=np.int32)
    if c_size:
        cs[1:] = np.cumsum(np.array(nblks[:-1], dtype=np.int64) * chunk * bs)
    total = int(sum(nblks)) * chunk * bs
    bcol = np.zeros(total // bs, dtype=np.uint16)
    bval = np.zeros(total, dtype=np.asarray(val).dtype)
    for k in range(c_size):
        W = (pcl[k] & SCL_BIT) + 1
        for t in range(chunk):
            s, v = lanes_col[k][t], lanes_val[k][t]
            it = 0
            for b in range(nblks[k]):
                if it < W:
                    base = s[it]
                    bcol[cs[k] // bs + t + b * chunk] = base
                    bval[cs[k] + t + (b * bs) * chunk] = v[it]
                    it += 1
                    for h in range(1, bs):
                        if it < W and s[it] - base == h:
                            bval[cs[k] + t + (b * bs + h) * chunk] = v[it]
                            it += 1
                else:
                    bcol[cs[k] // bs + t + b * chunk] = (s[W - 1] // bs) * bs
    out.update(cs=cs, cl=cl, sellcs_col=bcol, sellcs_val=bval, nnz=total)
    return out


def spmv(amb, x):
    """kernel_spmv_amb_atomic (kernel_spmv_amb.cu:21-79), lane by lane."""
    C, bs = amb["chunk"], amb["block_size"]
    xp = np.zeros(amb["N"] + 20, dtype=amb["sellcs_val"].dtype)
    xp[:amb["N"]] = x
    y = np.zeros(amb["pad_M"], dtype=amb["sellcs_val"].dtype)
    for i in range(amb["c_size"] * C):
        c, lane = divmod(i, C)
        row = int(amb["s_write_permutation"][i]) + int(amb["s_write_permutation_offset"][c]) * USHORT_MAX
        start, colstart = amb["cs"][c] + lane, amb["cs"][c] // bs + lane
        length = int(amb["cl"][c])
        c_off = (length >> SCL_BORDER) * amb["seg_size"]
        acc = 0
        for _ in range((length & SCL_BIT) + 1):
            cc = int(amb["sellcs_col"][colstart]) + c_off
            for b in range(bs):
                acc += amb["sellcs_val"][start] * xp[cc + b]
                start += C
            colstart += C
        y[row] += acc
    return y[:amb["M"]]
