"""Generates tests/golden/rulebook_small.json.

An INDEPENDENT pure-Python transcription of SURVEY.md Appendix A.2/A.3 (dict for the hash,
explicit nested loops) -- not derived from oracle/spconv_ref.c and not from the reference (the
reference holds no fixtures for this path; spconv 1.2.1 is absent offline => parity unpinned).
Run:  python tests/golden/make_rulebook_golden.py
"""
import itertools
import json
import os

import numpy as np


def cdiv(a, b):           # C truncating division
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def valid_out_pos(x, k, s, p, d, out_shape):
    D = len(x)
    lo = [cdiv(x[i] - (k[i] - 1) * d[i] - 1 + s[i] + p[i], s[i]) for i in range(D)]
    hi = [cdiv(x[i] + p[i], s[i]) for i in range(D)]
    cnt = [cdiv(hi[i] - lo[i], d[i]) + 1 for i in range(D)]
    res = []
    for counter in itertools.product(*[range(c) for c in cnt]):      # last dim fastest
        out = [hi[i] - counter[i] * d[i] for i in range(D)]
        m, off = 1, 0
        for j in range(D - 1, -1, -1):
            off += cdiv(m * (x[j] - out[j] * s[j] + p[j]), d[j])
            m *= k[j]
        if all(0 <= out[i] < out_shape[i] for i in range(D)):
            res.append((out, off))
    return res


def rulebook(indices, batch, spatial, k, s, p, d, subm):
    D = len(spatial)
    if subm:
        p = [kk // 2 for kk in k]
        s = [1] * D
        out_shape = list(spatial)
    else:
        out_shape = [(spatial[i] + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 for i in range(D)]
    K = int(np.prod(k))
    N = len(indices)
    pairs = [[[-1] * N for _ in range(K)] for _ in range(2)]
    num = [0] * K
    out_idx = []
    if subm:
        h = {}
        for j, row in enumerate(indices):
            h[(row[0], tuple(row[1:]))] = j
        for j, row in enumerate(indices):
            for out, off in valid_out_pos(list(row[1:]), k, s, p, d, out_shape):
                key = (row[0], tuple(out))
                if key in h:
                    pairs[0][off][num[off]] = j
                    pairs[1][off][num[off]] = h[key]
                    num[off] += 1
        out_idx = [list(map(int, r)) for r in indices]
    else:
        h = {}
        for j, row in enumerate(indices):
            for out, off in valid_out_pos(list(row[1:]), k, s, p, d, out_shape):
                key = (row[0], tuple(out))
                if key not in h:
                    h[key] = len(out_idx)
                    out_idx.append([int(row[0])] + [int(o) for o in out])
                pairs[0][off][num[off]] = j
                pairs[1][off][num[off]] = h[key]
                num[off] += 1
    return out_shape, out_idx, pairs, num


def main():
    rng = np.random.default_rng(20261004)
    cases = []
    specs = [
        (2, [5, 4], [3, 3], [1, 1], [0, 0], [1, 1], True, 12, 2),
        (3, [4, 3, 6], [3, 3, 3], [1, 1, 1], [0, 0, 0], [1, 1, 1], True, 20, 2),
        (2, [7, 6], [3, 3], [2, 2], [1, 1], [1, 1], False, 14, 2),
        (3, [4, 3, 12], [3, 3, 3], [1, 1, 4], [0, 0, 0], [1, 1, 1], False, 24, 2),
        (2, [6, 6], [3, 3], [1, 1], [2, 2], [2, 2], False, 10, 1),
        (2, [5, 5], [2, 2], [2, 2], [0, 0], [1, 1], False, 9, 2),
        (1, [17], [5], [3], [2], [1], False, 8, 2),
        (2, [5, 4], [3, 3], [1, 1], [0, 0], [1, 1], True, 10, 2),      # with duplicate sites below
    ]
    for ci, (D, shape, k, s, p, d, subm, n, B) in enumerate(specs):
        vol = int(np.prod(shape))
        sel = rng.choice(B * vol, size=n, replace=False)
        sel = sel[np.argsort(sel // vol, kind="stable")]
        idx = []
        for v in sel:
            b, pos = int(v // vol), int(v % vol)
            row = [b]
            for dd in range(D):
                st = int(np.prod(shape[dd + 1:]))
                row.append((pos // st) % shape[dd])
            idx.append(row)
        if ci == len(specs) - 1:
            idx[7] = list(idx[2])      # duplicate coordinates: SubM hash keeps the LAST row
        out_shape, out_idx, pairs, num = rulebook(idx, B, shape, k, s, p, d, subm)
        cases.append(dict(ndim=D, batch_size=B, spatial_shape=shape, ksize=k, stride=s, padding=p,
                          dilation=d, subm=subm, indices=idx, out_shape=out_shape,
                          out_indices=out_idx, indice_pairs=pairs, indice_pair_num=num))
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "rulebook_small.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
