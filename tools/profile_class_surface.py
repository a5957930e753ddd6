#!/usr/bin/env python3
"""Where the host time of a scripts/fit.py-shaped loop on the class surface goes (HandModel / ObjectModel /
calculate_energy / MalaStar on the HIP ops, BASELINE configs[1]): cProfile of `iters` iterations, the 45 entries with the
largest own time, and the wall time per iteration with / without the profiler attached.  Development aid (GPU box).

  python tools/profile_class_surface.py [iters] > gpurun_out/class_surface_host_profile.txt
"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch

    import plugin_surface as ps
    from bench import make_initial_state
    from graspqp_amd.hands import get_hand_spec
    from graspqp_amd.utils import meshes

    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    spec = get_hand_spec("allegro")
    fv = meshes.superquadric(0)
    sp = meshes.surface_points(fv, 2500, oversample=4, seed=42)
    hp, idx = make_initial_state(spec, fv, 256, 12, 1000)
    hp, idx = hp.cuda().float(), idx.cuda()
    from graspqp_amd import ops

    for rep in range(2):
        for route in (True, False):
            old = ops.use_dispatcher(route)
            r = ps.class_surface_loop(spec, fv, sp, 256, 12, hp, idx, iters=iters)
            ops.use_dispatcher(old)
            print("registered ops through the dispatcher:" if route else "eager route:", round(r["ms_per_iteration"], 4),
                  "ms / iteration,", round(r["evals_per_s"]), "evals/s")
    pr = cProfile.Profile()
    pr.enable()
    r = ps.class_surface_loop(spec, fv, sp, 256, 12, hp, idx, iters=iters)
    pr.disable()
    print("profiled:", r["ms_per_iteration"], "ms / iteration")
    for key in ("tottime", "cumtime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print(s.getvalue())


if __name__ == "__main__":
    main()
