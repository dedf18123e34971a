"""Fit the launch planner's duration model (navillm_amd/csrc/gemm_bf16.hip: est_us_256) to tools/gemm_tme_probe.py's measurements.
Usage: python tools/fit_tme_model.py profiles/r03_gemm_tme_probe.txt"""
import re
import sys
import numpy as np
from scipy.optimize import least_squares


def tail_split(rem, KT):
    """gemm_bf16.hip: tail_split()"""
    if rem <= 0 or rem > 128:
        return 1
    max_split, min_slice = (8, 8) if rem <= 16 else (6, 20)
    s = min(256 // rem, max_split, KT // min_slice, 512 // rem)
    return s if s >= 2 else 1


def parse(path):
    pts = []
    for l in open(path):
        m = re.match(r"M=\s*(\d+)\s+(\w+)\s+(NT|NN) N=\s*(\d+) K=\s*(\d+):(.*)best", l)
        if not m:
            continue
        M, lay, N, K = int(m.group(1)), m.group(3), int(m.group(4)), int(m.group(5))
        for t, us in re.findall(r"(\d+):\s*([\d.]+)us", m.group(6)):
            t = int(t)
            if 84 <= t <= 88:
                pts.append((lay, M, N, K, t - 80, float(us)))
    return pts


def model(p, M, N, K, tme):
    tk = p[0:5][tme - 4]
    oh0, c0, fix0, fix1, kfrac = p[5], p[6], p[7], p[8], p[9]
    bme, KT = 32 * tme, (K + 63) // 64
    T = -(-M // bme) * -(-N // 256)
    full, rem = divmod(T, 256)
    oh = oh0 * (0.5 + 0.5 * tme / 8)
    t = full * (KT * tk + oh)
    if rem:
        s = tail_split(rem, KT)
        f = min(1.0, rem * s / 256.0)
        tke = tk * (c0 + (1 - c0) * f)
        if s >= 2:
            t += (KT / s) * tke * kfrac + oh + max(4.0, fix0 + fix1 * s)
        else:
            t += KT * tke + oh
    return t


pts = parse(sys.argv[1])
for lay in ("NT", "NN"):
    d = [q for q in pts if q[0] == lay]
    y = np.array([q[5] for q in d])

    def res(p):
        return np.array([model(p, q[1], q[2], q[3], q[4]) for q in d]) / y - 1.0
    p0 = np.array([0.9, 1.0, 1.1, 1.25, 1.4, 7.0, 0.8, -10, 12, 1.0])
    r = least_squares(res, p0, bounds=([0.3] * 5 + [0, 0.3, -60, 0, 0.8], [3] * 5 + [30, 1.0, 60, 60, 2.0]))
    e = res(r.x)
    print(lay, "tk[4..8] =", np.round(r.x[:5], 3), "oh0 =", round(r.x[5], 2), "c0 =", round(r.x[6], 3), "fix =", np.round(r.x[7:9], 2),
          "kfrac =", round(r.x[9], 3), " rms rel err =", round(float(np.sqrt((e ** 2).mean())), 3), " max =", round(float(np.abs(e).max()), 3))
    # how good are the model's choices?
    tot_best = tot_pick = tot_8 = 0.0
    by = {}
    for q in d:
        by.setdefault(q[1:4], {})[q[4]] = q[5]
    for (M, N, K), row in by.items():
        pick = min(row, key=lambda t: model(r.x, M, N, K, t) * (1.0 if t == 8 else 1.0 / 0.97))
        tot_best += min(row.values()); tot_pick += row[pick]; tot_8 += row[8]
    print(f"   sum over shapes: TME=8 {tot_8:.0f} us, model's picks {tot_pick:.0f} us, oracle picks {tot_best:.0f} us")
