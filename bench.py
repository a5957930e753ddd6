#!/usr/bin/env python3
"""bench.py -- grasp energy+grad evals/s of the MALA* hot path on MI355X (BASELINE.json metric).

One "step" = one MALA* iteration (propose -> FK/contacts -> object SDF -> friction-cone QP -> E_dis/E_fc/E_joints/
E_pen/E_spen -> backward to hand_pose -> accept) over the whole batch; one eval = one batch row through one step.
Workload at N=1 = BASELINE configs[1]: Allegro, 1 YCB-style mesh, batch_size 256, n_contact 12 (synthetic mesh,
2500 surface points).  N>1: one process per GPU, every rank owns its own object(s) with the same per-rank batch (weak
scaling, no collective in the loop; one gather of the final poses / energies / contact indices at the end).

    python bench.py --gpus N --steps K --warmup W         # N > 1: this process only spawns the N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Timing: W warm-up iterations, then `--windows` (default 3) timed windows of EXACTLY K iterations each, every window
bracketed by barrier + synchronize on both sides, MAX over ranks per window; the line reports the MEDIAN window
(`ms_per_step` = median window / K, `value` = rows of all ranks * K / median window; all windows are listed).
"""
import argparse
import ctypes as _ct
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP32_VECTOR_PEAK_TFLOPS = 157.3
FLOP_PER_TRI_TEST = 70.0      # one point-triangle test ~ 60-80 flop (gq_tri_rank: 45 VALU ops)
PROFILE_TAG = "r03"           # profiles/<tag>_* files written by tools/profile_round.sh from this same command


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="iterations per timed window")
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps iterations; the median is reported")
    ap.add_argument("--event_steps", type=int, default=50, help="extra eager iterations timed with HIP events")
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--n_contact", type=int, default=12)
    ap.add_argument("--n_cone_vecs", type=int, default=4, help="friction-cone edges per contact (BASELINE configs[4] uses 8)")
    ap.add_argument("--n_objects", type=int, default=1, help="objects per rank.  Default 1 = BASELINE configs[1] per GPU at "
                    "every N (weak scaling of the N=1 workload, what the SCALE runs compare); BASELINE configs[3] -- 64 meshes "
                    "sharded over 8 GPUs, 2048 rows per rank -- is `--gpus 8 --n_objects 8`, configs[4] per hand `--gpus 8 "
                    "--n_objects 32 --batch_size 1024 --n_cone_vecs 8 [--hand robotiq3]`")
    ap.add_argument("--hand", default="allegro")
    ap.add_argument("--fork", type=int, default=-1, help="1: every role its own launch, the two branches of the evaluation as "
                    "parallel hipGraph branches; -1 (default): follow the batch size (from 384 rows on)")
    ap.add_argument("--graph_iters", type=int, default=8, help="MALA* iterations captured per hipGraph (reduced to a divisor of --steps)")
    ap.add_argument("--fused", type=int, default=-1, help="1: force-closure and penetration branches share two launches; "
                    "-1 (default): follow the batch size (below 384 rows)")
    ap.add_argument("--graph", type=int, default=1, help="replay the iteration from hipGraphs (1, default) or launch eagerly (0)")
    ap.add_argument("--dist_backend", default="nccl", help="nccl (= RCCL, default) | gloo: rehearsal of the multi-process path "
                    "on fewer GPUs than ranks (ranks share devices round-robin; collectives go through host memory)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--split_self_pen", type=int, default=1, help="0: self penetration stays in the FK forward launch at large "
                    "batches (A/B runs)")
    ap.add_argument("--sdf_topk", type=int, default=0, help="clusters per round of the stand-alone contact-SDF kernel: 0 = "
                    "default (4), 2 / 4 forced (A/B runs)")
    ap.add_argument("--pen_ppt", type=int, default=0, help="surface points per thread of the penetration query: 0 = default, "
                    "1 / 2 forced (A/B runs)")
    ap.add_argument("--pen_caps", type=int, default=0, help="LDS list capacities of the stand-alone penetration query: 0 = by "
                    "launch size, 1 / 2 / 3 = 512 / 256 / 128 entries per block (A/B runs)")
    ap.add_argument("--cpu_rows", type=int, default=8)
    ap.add_argument("--cpu_reps", type=int, default=5)
    ap.add_argument("--point_grid", type=int, default=0, help="cells per axis of the surface-point grid of the link-driven "
                    "penetration query (A/B runs; 0 = the point-driven query, which is faster: one thread per point keeps far\n"
                    "more independent loads in flight than one block per row)")
    ap.add_argument("--sdf_plain_mapping", type=int, default=0, help="1: plain block->query mapping of the object SDF (A/B of the "
                    "XCD-aware placement used with >= 8 meshes)")
    ap.add_argument("--energy_type", default="graspqp", choices=("graspqp", "dexgrasp", "tdg"),
                    help="force-closure energy of scripts/fit.py:343-347 (graspqp = BASELINE's metric; the others: A/B lines)")
    ap.add_argument("--plugin_surface", type=int, default=-1, help="1: after the timed region, time the reference's plugin "
                    "surface on the same workload (compute_sdf at the reference's shapes, QPFunction / SQPLsqSolver.solve, a "
                    "fit.py-shaped loop on the class surface); -1 (default): at N=1 on BASELINE configs[1] only")
    ap.add_argument("--selftest_ranks", action="store_true",
                    help="launcher / rendezvous / collective sequence only (no GPU work): used by the CPU test of --gpus N")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun environment spawns the N ranks itself
# --------------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """Parent of an N-rank run: touches no GPU (no HIP call, not even torch.cuda.is_available()), starts one child per
    rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would set),
    forwards rank 0's JSON line, and exits with the worst child status.  Mirrors the reference's scale-out model: one
    process per batch of objects (scripts/fit_all.sh:181-208)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:  # a dead rank would leave the others waiting in a barrier
                    rc = rc or code
                    for q in procs:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


def make_initial_state(spec, fv, B, n, seed):
    """Hands on a shell of radius extent + U(0.05,0.1) around the object, palm roughly towards it, joints =
    default + 0.1*range jitter, random contact indices (stands in for initialize_convex_hull)."""
    g = torch.Generator().manual_seed(seed)
    ext = float(np.abs(fv).max())
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    t = d * (ext + 0.05 + 0.05 * torch.rand(B, 1, generator=g))
    z = -d  # hand forward axis (z for allegro) looks at the object
    a = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    x = torch.nn.functional.normalize(a - (a * z).sum(-1, keepdim=True) * z, dim=-1)
    y = torch.linalg.cross(z, x)
    six = torch.cat([x, y], dim=1)  # first two columns of R
    lo, hi = torch.tensor(spec.joints_lower), torch.tensor(spec.joints_upper)
    th = torch.tensor(spec.default_state)[None] + 0.1 * (hi - lo) * torch.randn(B, spec.n_dofs, generator=g).clamp(-2, 2)
    th = torch.minimum(torch.maximum(th, lo), hi)
    idx = torch.randint(spec.n_contact_candidates, (B, n), generator=g)
    return torch.cat([t, six, th], 1).float(), idx


def cpu_baseline(spec, fv, sp, n_contact, rows, reps, label):
    """The oracle (CPU torch restatement of the reference algorithm, fp32: qpth-form PDIPM with the 2nz x 2nz Schur
    system, brute-force SDF) on a bounded sample of the workload: `rows` batch rows, all surface points, full meshes;
    energy + backward; 1 warm-up + `reps` timed evaluations, median.  Baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu
    from ref_cpu import models as omodels

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # a GPU box gives one GPU a 16-core share; more threads only oversubscribe
    torch.set_num_threads(cores)
    hp, idx = make_initial_state(spec, fv, rows, n_contact, 123)
    oh = omodels.OracleHand(spec, torch.float32)
    oo = omodels.OracleObject([fv], [sp], rows, torch.float32)

    def one():
        h = hp.clone().requires_grad_()
        oh.set_parameters(h, idx)
        losses = ref_cpu.calculate_energy(oh, oo)
        ref_cpu.total_energy(losses).sum().backward()

    one()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    return {"value": rows / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"{label}: {rows} rows x {reps} timed energy+grad evaluations (median; 1 warm-up), n_contact={n_contact}, "
                      f"{fv.shape[0]}-face mesh, {sp.shape[0]} surface points, oracle/ref_cpu fp32, {dt:.2f} s each"}


def _load_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return None
    return None


def selftest_ranks(args, rank, world):
    """The distributed skeleton of a run without any GPU work: rendezvous, barrier-bracketed window, MAX over ranks,
    gather of a per-rank result.  `python bench.py --gpus 2 --dist_backend gloo --selftest_ranks` runs on a CPU box."""
    import torch.distributed as dist

    from graspqp_amd.parallel import gather_results, shard_objects

    dist.init_process_group(args.dist_backend if args.dist_backend != "nccl" else "gloo")
    objs = shard_objects(args.n_objects * world, world, rank)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0])
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    B = len(objs) * args.batch_size
    pose = torch.full((B, 25), float(rank))
    poses, energies, idxs = gather_results(pose, torch.full((B,), float(rank)), torch.zeros(B, args.n_contact, dtype=torch.long))
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "rows_gathered": int(sum(p.shape[0] for p in poses)),
                          "objects_per_rank": [len(shard_objects(args.n_objects * world, world, r)) for r in range(world)],
                          "window_s_max_over_ranks": float(tt.item())}))
    dist.barrier()
    dist.destroy_process_group()


def rank_main(args):
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.selftest_ranks:
        return selftest_ranks(args, rank, world)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.dist_backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.cuda.set_device(local % torch.cuda.device_count())
            dist.init_process_group(args.dist_backend)
    else:
        torch.cuda.set_device(0)

    from graspqp_amd import _C, ops
    from graspqp_amd.hands import get_hand_spec
    from graspqp_amd.parallel import gather_results, shard_objects
    from graspqp_amd.stepper import GraspStepper
    from graspqp_amd.utils import meshes

    if args.sdf_plain_mapping:
        _C.call("gq_debug_set_sdf_mapping", 1)
    if args.sdf_topk:
        _C.call("gq_debug_set_sdf_topk", int(args.sdf_topk))
    if args.pen_ppt:
        _C.call("gq_debug_set_pen_ppt", int(args.pen_ppt))
    if args.pen_caps:
        _C.call("gq_debug_set_pen_caps", int(args.pen_caps))
    spec = get_hand_spec(args.hand)
    # object ids are global; rank r owns the contiguous block shard_objects gives it (whole objects per rank)
    my_objs = shard_objects(args.n_objects * world, world, rank)
    fvs = [meshes.superquadric(o) for o in my_objs]
    sps = [meshes.surface_points(f, 2500, oversample=4, seed=42) for f in fvs]
    B = len(my_objs) * args.batch_size
    hand = ops.HandHandle(spec)
    st = GraspStepper(hand, ops.MeshSet(fvs), torch.tensor(np.stack(sps)), args.batch_size, args.n_contact,
                      fc_cfg={"n_cone_vecs": args.n_cone_vecs}, seed=1 + rank, point_grid=args.point_grid,
                      split_self_pen=bool(args.split_self_pen), energy_type=args.energy_type)
    hps, idxs = zip(*[make_initial_state(spec, f, args.batch_size, args.n_contact, 1000 + o) for f, o in zip(fvs, my_objs)])
    st.reset(torch.cat(hps).cuda(), torch.cat(idxs).cuda())

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # every timed window is made of whole graph replays: no eager remainder inside the timed region
    g_iters = 0
    if args.graph:
        g_iters = max(d for d in (1, 2, 4, 8, 16, 32, 64) if d <= max(1, args.graph_iters) and args.steps % d == 0)
        st.capture(fork=None if args.fork < 0 else bool(args.fork), fused=None if args.fused < 0 else bool(args.fused),
                   iters=g_iters)
    for _ in range(args.warmup):
        st.step()
    st.realign_draws()  # a warm-up that is not a multiple of the graph length ends with eager iterations (untimed);
    sync()              # the timed windows then start on a fresh draw buffer: whole graph replays only
    st.start_kernel_timing()
    window_s, enq_s = [], []
    cdev = "cuda" if args.dist_backend == "nccl" else "cpu"
    for _ in range(max(1, args.windows)):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.step()
        assert st._graph_pending == 0
        t_enq = time.perf_counter() - t0  # host time to enqueue the window (must stay below dt or the host is the limit)
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        window_s.append(dt)
        enq_s.append(t_enq)
    _, span_ms, n_span = st.kernel_times_ms()
    assert torch.isfinite(st.energy).all(), "non-finite energies"

    # ---- after the timed region --------------------------------------------------------------------------------------
    # (1) the same loop launched eagerly on one stream with a HIP event pair around every hand-penetration query (events
    # cannot bracket a node of a replayed hipGraph) -> the query's isolated duration
    g, st._graph = st._graph, None
    st.start_kernel_timing()
    for _ in range(args.event_steps):
        st.step()
    torch.cuda.synchronize()
    evs, span_iso_ms, _ = st.kernel_times_ms()
    # (2) the fused launch sequence issued eagerly, a torch event pair around every C-ABI call (they run on torch's
    # current stream): live duration of each launch group
    group_ms = {}
    if args.fused != 0 and st.penetration_only == 1:
        names = ["fk_forward(+propose" + ("+object SDF)" if B <= 512 else ")"), "object_sdf", "stage_a+stop+stage_b",
                 "fk_backward(+energies+accept)"]
        acc = {k: [] for k in names}
        for _ in range(max(args.event_steps // 2, 8)):
            st.draw()
            stream = _C.stream_ptr()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            attach = B <= 512
            ev[0].record()
            st._eval_fk(st.pose_new, st.idx_new, stream, True, sdf=attach, spheres=False)
            ev[1].record()
            if not attach:
                _C.call("gq_sdf_forward_meshset", st.objs.handle, _C.f32(st.cpts), B * st.n, st.be * st.n, _C.f32(st.d2),
                        _C.i32(st.sgn), _C.f32(st.onrm), _C.f32(st.closest), stream)
            ev[2].record()
            st._pen_desc.hand_pose = st.pose_new.data_ptr()
            _C.call("gq_fc_pen_step", _ct.byref(st._fc_desc), _ct.byref(st._pen_desc), stream)
            ev[3].record()
            st._eval_tail(st.pose_new, st.idx_new, stream, True)
            ev[4].record()
            torch.cuda.synchronize()
            for i, k in enumerate(names):
                acc[k].append(ev[i].elapsed_time(ev[i + 1]))
        group_ms = {k: float(np.median(v)) for k, v in acc.items() if (k != "object_sdf" or B > 512)}
    # (3) executed work of the two SDF kernels on the final state (debug counters, stand-alone launches)
    executed = {}
    try:
        cnt = torch.zeros(12, dtype=torch.int64, device="cuda")
        _C.call("gq_debug_set_pen_counters", _ct.c_void_p(cnt.data_ptr()))
        # the counting launch of the query (every block adds to the same 12 global counters: ~0.6 ms) runs as the two-points-
        # per-thread instantiation -- the form the query has as the role of stage A -- so that it does not sit in the
        # rocprofv3 statistics of gq_pen_grid_kernel<..., 1>, the kernel the eager pass above times alone
        _C.call("gq_debug_set_pen_ppt", 2)
        stream = _C.stream_ptr()
        st._eval_fk(st.hand_pose, st.contact_idx, stream)
        _C.call("gq_sdf_forward_meshset", st.objs.handle, _C.f32(st.cpts), B * st.n, st.be * st.n, _C.f32(st.d2),
                _C.i32(st.sgn), _C.f32(st.onrm), _C.f32(st.closest), stream)
        if st.grid is not None:
            _C.call("gq_hand_pen_forward_cells", hand.links.handle, st.grid.handle, _C.f32(st.surf), st.n_obj, st.P, st.be,
                    _C.f32(st.hand_pose), st.D, _C.f32(st.Rg), _C.f32(st.link_T), _C.f32(st.pen_dis), _C.i32(st.pen_link),
                    _C.f32(st.pen_gvec), None, None, stream)
        else:
            _C.call("gq_hand_pen_forward", hand.links.handle, _C.f32(st.surf), st.n_obj, st.P, st.be, _C.f32(st.hand_pose),
                    st.D, _C.f32(st.Rg), _C.f32(st.link_T), 1, _C.f32(st.pen_dis), _C.i32(st.pen_link), _C.f32(st.pen_gvec),
                    None, 0, None, None, _C.f32(st.patch), stream)
        torch.cuda.synchronize()
        _C.call("gq_debug_set_pen_counters", None)
        c = [int(v) for v in cnt.tolist()]
        executed = {"object_sdf": {"queries": c[1], "cluster_visits": c[0], "point_triangle_tests": c[0] * 64,
                                   "max_visits_per_query": c[2]},
                    "hand_pen": {"point_link_pairs": B * st.P * hand.L, "pairs_reaching_candidates": c[4],
                                 "point_triangle_tests": c[5], "pairs_ranked_inline": c[6], "blocks": c[7],
                                 "scanning_wavefronts": c[11], "wavefront_link_sphere_tests": c[8],
                                 "wavefront_link_sphere_hits": c[9], "point_link_pairs_in_box": c[10],
                                 "link_cell_pairs_walked": c[0] if st.grid is not None else None,
                                 "query": "link-driven (point grid)" if st.grid is not None else
                                          "point-driven, two surface points per thread (the stage-A role)"}}
    except Exception as e:  # diagnostics must never cost the bench line
        executed = {"error": repr(e)}
    finally:
        _C.call("gq_debug_set_pen_ppt", int(args.pen_ppt or 0))
    st._graph = g

    # ---- end-of-run gather: the only collective of a run (<= 0.5 MB per rank) -----------------------------------------
    per_rank = None
    if dist is not None:
        poses, energies, cidx = gather_results(st.hand_pose.to(cdev), st.energy.to(cdev), st.contact_idx.to(cdev))
        mine = torch.tensor([B * args.steps / float(np.median(window_s))], device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(v.item()) for v in allr]
        if rank == 0:
            assert sum(p.shape[0] for p in poses) == B * world and all(torch.isfinite(e).all() for e in energies)

    if rank == 0:
        dt = float(np.median(window_s))
        total_evals = B * world * args.steps
        nf = hand.links.n_faces
        is_cfg2 = (args.hand, args.n_objects, args.batch_size, args.n_contact, args.n_cone_vecs) == ("allegro", 1, 256, 12, 4)
        is_cfg3 = (args.hand, args.batch_size, args.n_contact, args.n_cone_vecs) == ("allegro", 256, 12, 4) and \
            world * args.n_objects == 64 and world > 1
        roof = None
        if n_span:
            # Dominant SDF kernel = the hand-penetration query (gq_pen_grid_kernel body; in the graph it is the pen role
            # of gq_stage_a_kernel).  `achieved` / `frac` follow the bench contract: ALGORITHMIC bytes of a TorchSDF-shaped
            # dist-only op (16 B per (point, link) query + 36 B per link-mesh face once per launch) over the
            # kernel's mean duration -- an accounting figure, NOT an HBM utilisation: the kernel culls > 99 % of the pairs
            # and moves ~5 % of those bytes (see `utilisation`).  Durations: in-kernel 100 MHz timestamps, first block
            # start -> last block end, averaged over EVERY launch of the timed windows (`kernel_ms`); the same kernel
            # launched alone in the eager pass, by the in-kernel span and by HIP event pairs
            k_ms = span_ms
            k_ms_events = float(np.mean(evs)) if evs else None
            alg = B * st.P * hand.L * 16 + nf * 36
            ach = alg / (k_ms * 1e-3) / 1e9
            # counter bytes are NOT measured by this run (PMC passes need rocprofv3): they are static figures from the newest
            # committed profile of the same command, labelled as such
            tag = next((t for t in (PROFILE_TAG, "r02", "r01") if _load_json(f"{t}_pmc_traffic.json")), None)
            pm = _load_json(f"{tag}_pmc_traffic.json") if tag else None
            src = f"static, from profiles/{tag}_pmc_traffic.json (rocprofv3 --pmc passes of this command at the time of that commit)"
            traffic = None
            pen_keys = [k for k in (pm or {}) if k.startswith("gq_pen_grid_kernel<true")]
            if is_cfg2 and pen_keys:
                traffic = pm[pen_keys[0]]["hbm_bytes_per_launch"]
            t_iso = (k_ms_events or span_iso_ms or k_ms) * 1e-3
            util = {"note": "physical / executed-work utilisation of the same kernel launched alone (what bounds it is L2 "
                            "round-trip latency: ~6 dependent gathers per block, 3-4 blocks per CU)",
                    "kernel_ms_isolated": span_iso_ms, "kernel_ms_isolated_hip_events": k_ms_events}
            if traffic:
                util.update(hbm_bytes_per_launch=traffic, hbm_gbps=traffic / t_iso / 1e9,
                            hbm_frac=traffic / t_iso / 1e9 / HBM_PEAK_GBPS, traffic_source=src)
            hp_ex = executed.get("hand_pen") if isinstance(executed, dict) else None
            if hp_ex:
                fl = hp_ex["point_triangle_tests"] * FLOP_PER_TRI_TEST
                util.update(executed_point_triangle_tests=hp_ex["point_triangle_tests"],
                            executed_tflops=fl / t_iso / 1e12, fp32_alu_frac=fl / t_iso / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                            bruteforce_point_triangle_tests=B * st.P * nf)
            small = getattr(st, "graph_mode", "one grid") == "one grid"
            roof = {"bound": "hbm",
                    "limiter": "latency (L2 round trips); not HBM- or ALU-bound, see utilisation" if small else
                               "VALU issue slots and registers shared with the concurrent force-closure branch (DESIGN.md section 8); "
                               "not HBM-bound",
                    "kernel": "gq_pen_grid_kernel (hand-penetration query; " + ("in the graph it runs as the pen role of gq_stage_a_kernel)"
                                                                               if small else "its own launch on the second graph branch)"),
                    "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": min(ach / HBM_PEAK_GBPS, 1.0),
                    "frac_unclamped": ach / HBM_PEAK_GBPS,
                    "frac_kind": "algorithmic bytes / kernel time / peak (SURVEY 8d accounting of a TorchSDF-shaped op) -- NOT a "
                                 "utilisation: the query culls > 99 % of the (point, link) pairs and never moves those bytes "
                                 "(clamped at 1); the physical figures are `utilisation` (this kernel) and "
                                 "`plugin_surface.compute_sdf` (the TorchSDF-shaped call, where N*44 B really move)",
                    "traffic": traffic, "traffic_source": src if traffic else None, "algorithmic_bytes": alg,
                    "kernel_ms": k_ms, "kernel_launches_timed": n_span, "utilisation": util}
        # per-launch table: durations from the rocprofv3 kernel trace and HBM bytes from the PMC passes of THIS command
        # (tools/profile_round.sh writes profiles/<tag>_launches.json); live numbers: launch_groups_ms above
        ltag = next((t for t in (PROFILE_TAG, "r02") if _load_json(f"{t}_launches.json")), PROFILE_TAG)
        launches = _load_json(f"{ltag}_launches.json") if is_cfg2 and args.energy_type == "graspqp" else None
        if launches:
            launches["source"] = f"static, from profiles/{ltag}_launches.json"
        # executed instructions per wavefront and VALU issue-slot utilisation of every kernel (SQ counters of the same
        # command, tools/pmc_instr.sh): the "how far from the machine's limit" figure of kernels that are neither HBM- nor
        # MFMA-bound
        instr = _load_json(f"{ltag}_cfg2_instr.json") if is_cfg2 else None
        if launches and instr:
            for k in launches.get("kernels", []):
                rec = instr.get("kernels", {}).get(k["name"])
                if rec:
                    k.update(valu_instructions_per_wavefront=rec["valu_per_wave"], wavefronts=rec["wavefronts"],
                             valu_issue_frac=rec["valu_instructions_x4_cycles_over_duration"] if "valu_instructions_x4_cycles_over_duration" in rec
                             else rec["valu_per_wave"] * rec["wavefronts"] * 4.0 / (k["avg_us"] * 1e-6 * instr["clock_hz"] * instr["simds"]))
        hand_label = {"allegro": "Allegro", "shadow_hand": "Shadow Hand", "robotiq3": "Robotiq-3F"}.get(args.hand, args.hand)
        metric_name = f"grasp energy+grad evals/sec ({hand_label}, n_contact={args.n_contact})"
        res = {
            "metric": metric_name, "value": total_evals / dt, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_detail": "kinematics / SDF / energies f32; force-closure QP: PDIPM iterates f32 with the 6x6 Woodbury core "
                            "and the KKT backward in f64 (the reference runs qpth in f32)",
            "data": "synthetic",
            "config": {"workload": f"{args.hand}, {args.n_objects} YCB-style superquadric mesh(es) per GPU "
                                   f"({fvs[0].shape[0]} faces), batch_size={args.batch_size} each, n_contact={args.n_contact}, "
                                   f"2500 surface points, {args.n_cone_vecs}-edge friction cones"
                                   + (f", energy_type={args.energy_type}" if args.energy_type != "graspqp" else "")
                                   + (" (BASELINE configs[1])" if is_cfg2 and world == 1 else "")
                                   + (f" (BASELINE configs[1] on each of {world} GPUs: weak scaling of the N=1 workload)"
                                      if is_cfg2 and world > 1 else "")
                                   + (f" (BASELINE configs[3]: {world * args.n_objects} meshes sharded over {world} GPUs)"
                                      if is_cfg3 else ""),
                       "rows_per_gpu": B, "hip_graph": bool(args.graph), "iterations_per_graph": g_iters,
                       "eager_iterations_in_timed_region": 0,
                       "branches": getattr(st, "graph_mode", "eager") if args.graph else "eager"},
            "timing": {"windows": len(window_s), "window_ms": [w * 1e3 for w in window_s], "statistic": "median",
                       "host_enqueue_ms_per_step": float(np.median(enq_s)) / args.steps * 1e3},
            "mean_energy": float(st.energy.mean()), "accept_rate_last": float(st.accept.float().mean()),
            "roofline": roof, "launch_groups_ms": group_ms, "executed_work": executed, "launches": launches,
        }
        if per_rank is not None:
            res["per_rank_evals_per_s"] = per_rank
        want_plugin = args.plugin_surface == 1 or (args.plugin_surface < 0 and world == 1 and is_cfg2 and args.energy_type == "graspqp")
        if want_plugin:
            # the reference's plugin surface on the same workload (what an unchanged scripts/fit.py calls), after the timed
            # region and BEFORE the CPU baseline (its class-surface loop is bound by host time per call: measured 0.25 M
            # evals/s right after the 16-thread oracle run against 0.40 M before it); tools/plugin_surface.py documents the shapes
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import plugin_surface

                hp0, ix0 = make_initial_state(spec, fvs[0], args.batch_size, args.n_contact, 1000 + my_objs[0])
                res["plugin_surface"] = plugin_surface.measure(spec, fvs[0], sps[0], args.batch_size, args.n_contact, hp0, ix0,
                                                               loop_iters=300, reps=7)
                res["plugin_surface"]["stepper_evals_per_s"] = res["value"]
            except Exception as e:  # diagnostics must never cost the bench line
                res["plugin_surface"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            # the CPU port on configs 1 and 2 of BASELINE.json; `cpu_baseline` = the metric's configuration
            res["cpu_baseline"] = cpu_baseline(spec, fvs[0], sps[0], args.n_contact, args.cpu_rows, args.cpu_reps,
                                               "BASELINE configs[1] sample (Allegro, superquadric mesh)")
            sph = meshes.icosphere(3, 0.05)
            res["cpu_baseline_config0"] = cpu_baseline(get_hand_spec("allegro"), sph,
                                                       meshes.surface_points(sph, 2500, oversample=4, seed=42), 4, 4,
                                                       args.cpu_reps, "BASELINE configs[0] (Allegro, sphere, batch 4, n_contact 4)")
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    if int(env_world or 1) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch N ranks (python bench.py --gpus N, or "
              f"torch.distributed.run --nproc-per-node N)", file=sys.stderr)
        sys.exit(2)
    rank_main(args)


if __name__ == "__main__":
    main()
