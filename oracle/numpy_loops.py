"""Independent per-node numpy loop restatement of PNA's [mean|min|max|std] x [identity|amplification|attenuation]
aggregation (TEST INFRASTRUCTURE; small cases only).  Written from the published semantics
without sharing code with ``pyg_restatement`` so the two can cross-check each other."""
from __future__ import annotations

import math

import numpy as np


def pna_aggregate_loops(m: np.ndarray, dst: np.ndarray, num_nodes: int, avg_deg_log: float) -> np.ndarray:
    """m float32[E, T, F], dst int[E]  ->  float32[N, T, 12F] in the order
    [mean min max std | amp*(...) | att*(...)]; float32 arithmetic in edge order."""
    E, T, F = m.shape
    out = np.zeros((num_nodes, T, 12 * F), dtype=np.float32)
    f32 = np.float32
    for n in range(num_nodes):
        rows = [e for e in range(E) if dst[e] == n]  # ascending edge id = scatter order
        d = len(rows)
        for t in range(T):
            for f in range(F):
                if d == 0:
                    mean = mn = mx = f32(0.0)
                    mean2 = f32(0.0)
                else:
                    s = f32(0.0)
                    s2 = f32(0.0)
                    mn = f32(np.inf)
                    mx = f32(-np.inf)
                    for e in rows:
                        v = m[e, t, f]
                        s = f32(s + v)
                        s2 = f32(s2 + f32(v * v))
                        mn = min(mn, v)
                        mx = max(mx, v)
                    mean = f32(s / f32(d))
                    mean2 = f32(s2 / f32(d))
                var = f32(mean2 - f32(mean * mean))
                std = f32(np.sqrt(max(var, f32(1e-5))))
                if std <= f32(math.sqrt(1e-5)):
                    std = f32(0.0)
                base = np.array([mean, mn, mx, std], dtype=np.float32)
                amp = f32(f32(np.log(f32(d + 1))) / f32(avg_deg_log))
                att = f32(f32(avg_deg_log) / f32(np.log(f32(max(d, 1) + 1))))
                for a in range(4):
                    out[n, t, a * F + f] = base[a]
                    out[n, t, 4 * F + a * F + f] = f32(base[a] * amp)
                    out[n, t, 8 * F + a * F + f] = f32(base[a] * att)
    return out
