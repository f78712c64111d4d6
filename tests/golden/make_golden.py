#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ (run in the build container).

What each file pins, and where its numbers come from:
  healpix_nside4_ref.npy   192 pixel-centre vectors printed by the REFERENCE's own
                           chealpix.c (pix2vec_nest), compiled as it lies into
                           oracle/_ref/libchealpix_ref.so -- output of the reference itself.
  kat.json                 known-answer values copied from the reference's tests
                           (tests/morton_key/30bit_key.cu:20-26, 63bit_key.cu:20-26) and the
                           three generator outputs recorded in SURVEY.md section 8c.
  pipeline_n4096.npz       oracle outputs (oracle/grace_oracle.c) for 4096 spheres of the
                           reference's test generator: keys30/63, stable sort order, Euclidean
                           deltas, ALBVH leaves/nodes/root for max_per_leaf 1/8/32, 256 rays'
                           brute-force hit counts and fp32/fp64 column densities, and a
                           segmented-scan case.  Data files: inputs + expected outputs only.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402


def main():
    O.build(force=True)
    # --- reference-run outputs: HEALPix directions from the reference's chealpix.c ---------
    ref = O.ref_healpix()
    if ref is None:
        raise SystemExit("oracle/_ref is not built: /root/reference is required")
    import ctypes as C
    v = (C.c_double * 3)()
    dirs = np.empty((192, 3), np.float64)
    for i in range(192):
        ref.pix2vec_nest(4, i, v)
        dirs[i] = list(v)
    np.save(os.path.join(HERE, "healpix_nside4_ref.npy"), dirs)

    # --- KATs held by the reference's own tests ------------------------------------------------
    kat = {
        "morton30": {"x": 309, "y": 942, "z": 619, "spaced_x": 16814145, "spaced_y": 153125448,
                     "spaced_z": 134513161, "key": 861117685,
                     "source": "tests/morton_key/30bit_key.cu:20-26"},
        "morton63": {"x": 1365301, "y": 2014126, "z": 1683051,
                     "spaced_x": 1170975555344961601, "spaced_y": 1317338702596309576,
                     "spaced_z": 1297353911585505801, "key": 8995068606879603957,
                     "source": "tests/morton_key/63bit_key.cu:20-26"},
        "random_real4": {"low": [0, 0, 0, 0], "high": [1, 1, 1, 0.1],
                         "values": [[0.691545367, 0.58562845, 0.869682789, 0.0459066443],
                                    [0.23746188, 0.522250175, 0.539603293, 0.0190835483],
                                    [0.275667191, 0.730803967, 0.63866657, 0.00728325034]],
                         "source": "SURVEY.md section 8c (reference functor, "
                                   "tests/helper/random.cuh:56-111, run by the survey)"},
        "healpix_nside64_pix0": [0.70706841714771695, 0.70706841714771684, 0.010416666666666666],
        "integrate_tolerance": 5e-4,
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

    # --- oracle outputs on the reference's test generator ---------------------------------------
    n = 4096
    s = O.random_real4(n, (0, 0, 0, 0), (1, 1, 1, 0.1))
    bot, top = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    keys30 = O.morton_keys30(s, bot, top)
    keys63 = O.morton_keys63(s, bot, top)
    _, ss, order = O.sort_by_key(keys30, s)
    ss = np.ascontiguousarray(ss)
    deltas = O.deltas_euclid(ss)
    out = dict(spheres=s, keys30=keys30, keys63=keys63, order=order.astype(np.uint32),
               deltas=deltas)
    for mpl in (1, 8, 32):
        nodes, leaves, root, _ = O.albvh(ss, deltas, mpl)
        out["nodes_%d" % mpl] = nodes
        out["leaves_%d" % mpl] = leaves
        out["root_%d" % mpl] = np.int32(root)
    rng = np.random.default_rng(2024)
    d = rng.standard_normal((256, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    rays = np.zeros((256, 7), np.float32)
    rays[:, :3] = d
    rays[:, 3:6] = 0.5
    rays[:, 6] = 2.0
    out["rays"] = rays
    out["hit_counts"] = O.brute_hitcounts(rays, ss)
    c32, c64 = O.brute_cumulative(rays, ss)
    out["cumulative32"] = c32
    out["cumulative64"] = c64
    counts = rng.integers(0, 40, 300)
    counts[::7] = 0
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    data = rng.integers(1, 10, counts.sum()).astype(np.float32)  # 1..9 like the reference test
    out["seg_offsets"] = offs
    out["seg_data"] = data
    out["seg_result"] = O.segscan(offs, data)
    np.savez_compressed(os.path.join(HERE, "pipeline_n4096.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
