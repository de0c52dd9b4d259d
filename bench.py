#!/usr/bin/env python
"""Benchmark of the NEAT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL.  Either launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(RANK / WORLD_SIZE in the environment) or plainly as `python bench.py --gpus N ...`, which re-executes itself under
torch.distributed.run on 127.0.0.1 with a free port.  `--dry-run` (no GPU needed, gloo): launcher + rendezvous + the
flat-bucket gradient all-reduce only, for the CPU test of the multi-rank plumbing.

Workload (BASELINE.json configs[1], "C2"): abc-neat-a networks, 1024 rays x 128 depth samples per rank with the
depth samples GIVEN (sorted stratified U[0,6)), one TRAIN STEP = forward + loss + backward + Adam
(+ gradient all-reduce for N>1).  value = ray-samples/s over all ranks.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_RAYS, S_SAMPLES = int(os.environ.get("NEAT_BENCH_RAYS", "1024")), 128      # the env override is for scaling experiments only
FLOP_PER_RAY_SAMPLE = 9.1254e6        # 4 562 688 MAC per ray-sample per train step
CPU_BASELINE_THREADS = 16
DEFAULT_PRECISION = "bf16"      # BASELINE.json configs[1]: "8x256 SDF MLP, bf16, 1x MI355X"; --precision fp32 = parity build
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0 / 3.0, "fp16": 2500.0, "fp16x3": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peaks (f32-input MFMA / bf16 MFMA)
CPU_BASELINE_RAYS = 256         # bounded sample of the C2 workload for the CPU leg (x 128 samples per ray)


def cpu_baseline(seed):
    """The CPU oracle (oracle/neat_oracle.py = 'port' of the reference's pure-PyTorch path) timed on this box's host
    cores on a bounded sample of the same workload: 256 rays x 128 samples, fwd + loss + bwd + Adam; 3 warm-up + 5 timed
    steps, median, at (i) CPU_BASELINE_THREADS threads and (ii) 1 thread (the reference runner's own setting,
    training/volsdf_train.py:68)."""
    from neat_amd import synth
    from neat_amd.wireframe import WireframeGraph
    from oracle import neat_oracle as O
    R, S = CPU_BASELINE_RAYS, S_SAMPLES
    sd = synth.synth_state_dict(seed, "rough")
    sc = synth.synth_scene(seed=seed, n_rays=R)
    z = torch.tensor(synth.synth_z_vals(seed, R, S))
    wf = WireframeGraph(torch.tensor(sc["wf_vertices"]), torch.tensor(sc["wf_vconf"]), torch.tensor(sc["wf_edges"]),
                        torch.tensor(sc["wf_weights"]), 512, 512)
    inp = {k: torch.tensor(sc[k]) for k in ("intrinsics", "pose", "uv", "uv_proj")}
    lines, verts = wf.line_segments(), wf.vertices
    gt_rgb, gt_l = torch.tensor(sc["gt_rgb"]), torch.tensor(sc["gt_lines2d"])
    WARM, TIMED = 3, 5

    def run(threads):
        torch.set_num_threads(threads)
        p = O.params_from_numpy(sd, requires_grad=True)
        opt = torch.optim.Adam(list(p.values()), lr=5e-4)
        times = []
        for it in range(WARM + TIMED):
            t0 = time.perf_counter()
            rand = {"eik_idx": torch.randint(S, (R,)), "eik_uniform": torch.empty(R, 3).uniform_(-3, 3)}
            out = O.full_forward(p, inp, lines, verts, training=True, rand=rand, z_vals=z)
            lo = O.neat_loss(out, gt_rgb, gt_l)
            opt.zero_grad()
            lo["loss"].backward()
            opt.step()
            if it >= WARM:
                times.append(time.perf_counter() - t0)
        times.sort()
        return R * S / times[len(times) // 2]

    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    import neat_amd
    quota = neat_amd.cpu_quota()                  # the container's cgroup CPU quota (16 on the MI355X boxes of this pool)
    cores = min(quota, CPU_BASELINE_THREADS)
    before = torch.get_num_threads()
    v_all = run(cores)
    v_one = run(1)
    torch.set_num_threads(before)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": v_all, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle (own torch-CPU fp32 restatement of the reference path, pinned to reference goldens) train step on "
                      f"{R} rays x {S} samples of the C2 workload, median of {TIMED} after {WARM} warm-up, {cores} threads",
            "value_1thread": v_one, "cpu_model": model, "host_cpus": avail, "cpu_quota": quota,
            "threads_note": f"{cores} threads = the container's CPU quota ({quota} of {avail} host CPUs): a {R * S}-point batch of 256-wide GEMMs stops scaling in torch-CPU beyond "
                            "a few tens of threads; the 1-thread figure is the reference runner's own configuration"}


def dtu_switches_conf():
    """C3 / C4 name DTU scan24: confs/dtu.conf differs from abc-neat-a in the model switches dbscan_enabled = True, use_median = False and
    1024 global junction latents (the scene itself is not in /root/reference, the rays are synthetic)."""
    import copy
    from neat_amd import synth
    conf = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
    conf.update(dbscan_enabled=True, use_median=False)
    conf["global_junctions"] = dict(conf["global_junctions"], num_junctions=1024)
    return conf


C4_GLOBAL_RAYS = 4096


def make_workload(name, rank, world, dev, precision):
    """The trainer, its batch and the description of a headline workload on this rank.
      c2 (default): BASELINE configs[1] -- 1024 rays x 128 given samples PER GPU (weak scaling);
      c4: BASELINE configs[3] -- 4096 rays per step in total, split evenly over the ranks (512 per rank at 8 GPUs), DTU model
          switches, full losses, RCCL gradient all-reduce (strong scaling)."""
    from neat_amd import dp, synth
    from neat_amd.train import Trainer, synthetic_batch
    seed = dp.rank_seed(42, rank)
    torch.manual_seed(seed)
    if name == "c4":
        rays = dp.shard_rays(C4_GLOBAL_RAYS, world)
        sd = {k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough", num_junctions=1024).items()}
        tr = Trainer(model_conf=dtu_switches_conf(), device=dev, state_dict=sd)
        what = (f"C4: DTU model switches (device DBSCAN, 1024 junction latents), {C4_GLOBAL_RAYS} rays per step in total = {rays} per GPU "
                f"x {S_SAMPLES} given samples, full losses")
        scaling = "strong"
    else:
        rays = R_RAYS
        sd = {k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()}      # same weights on every rank
        tr = Trainer(device=dev, state_dict=sd)
        what = f"C2: abc-neat-a networks, {rays} rays x {S_SAMPLES} samples per GPU, depth samples given"
        scaling = "weak"
    _, inp, gt = synthetic_batch(seed, rays, dev, view=rank)
    tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(seed, rays, S_SAMPLES)).to(dev)
    tr.model.set_precision(precision)
    return tr, inp, gt, rays, what, scaling, sd


def secondary_legs(dev, sd, headline_precision, steps, note):
    """Timed AFTER the headline, on the same GPU, same clock discipline (warm-up, graph capture, `steps` steps between
    synchronisations); rank 0 of a single-GPU run only.  VERDICT r2 #3:
      * c2_parity_grade: the headline workload (C2) at the precision that meets north_star's 1e-4 (fp16x3: 3-product forward, f16
        backward -- the reference goldens pass at the fp32 bars, tests/test_gpu_parity.py fixture `prec`);
      * sampler_step_*: the REAL training step -- conf-default ErrorBoundSampler (N_samples 64 -> 98 samples per ray), rounds decided on
        the device, HIP graph -- at the headline precision and at the parity-grade one."""
    from neat_amd import synth
    from neat_amd.train import Trainer, synthetic_batch
    legs = {}

    def timed(tr, inp, gt, label, rays=R_RAYS):
        for _ in range(3):
            tr.step(inp, gt)
        graphed = tr.capture(inp, gt)
        # untimed: the first ~20 replays of the first sampler graph of a process run 0.6-1.2 ms slow (host-side enqueue of the refilled
        # random draws next to a cold launch path; scripts/probes/sampler_first_steps.py), a one-time 20 ms that is not the steady state
        for _ in range(20):
            tr.step(inp, gt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out, lo = tr.step(inp, gt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        if graphed:
            tr.check_nan()
        pts = out["points"]
        S = pts.shape[0] // rays if pts.dim() == 2 else pts.shape[1]
        note(f"secondary leg {label}: {1e3 * dt:.3f} ms/step")
        return {"ms_per_step": 1e3 * dt, "samples_per_ray": int(S), "value": rays * S / dt, "unit": "ray-samples/s",
                "rays_per_s": rays / dt, "launch": "hip graph replay" if graphed else f"eager ({tr.capture_error!r})",
                "loss": float(lo["loss"].detach()), "steps": steps}

    _, inp, gt = synthetic_batch(42, R_RAYS, dev)
    parity = "fp16x3"
    tr = Trainer(device=dev, state_dict=sd)
    tr.model.set_precision(parity)
    tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, R_RAYS, S_SAMPLES)).to(dev)
    leg = timed(tr, inp, gt, "c2_parity_grade")
    leg.update(precision=parity, workload="C2 (the headline workload) at the 1e-4-grade precision",
               parity="outputs within 1e-4, gradients within 2e-3 of the reference's goldens G2/G3/G7/G8/G11/G12 (tests/test_gpu_parity.py, fixture prec)")
    legs["c2_parity_grade"] = leg
    del tr
    for prec in dict.fromkeys((headline_precision, parity)):
        torch.manual_seed(42)
        tr = Trainer(device=dev, state_dict=sd)
        tr.model.set_precision(prec)
        tr.model.ray_sampler.sync_free = True
        leg = timed(tr, inp, gt, f"sampler_step_{prec}")
        leg.update(precision=prec, rounds=tr.model.ray_sampler.rounds_taken(),
                   workload="train step with the conf-default ErrorBoundSampler (VolSDF Alg. 1, 98 samples per ray), rounds decided on the device")
        legs[f"sampler_step_{prec}"] = leg
        del tr
    # the parity-grade precision with the sampler's SDF queries through the one-product f16 chain (conf key
    # model.hip_sampler_fast_values): main pass and gradients unchanged, sampled depths statistically the reference's
    torch.manual_seed(42)
    tr = Trainer(device=dev, state_dict=sd)
    tr.model.set_precision(parity)
    tr.model.sampler_fast_values = True
    tr.model.ray_sampler.sync_free = True
    leg = timed(tr, inp, gt, f"sampler_step_{parity}_fast_values")
    leg.update(precision=parity, rounds=tr.model.ray_sampler.rounds_taken(),
               workload="train step with the conf-default ErrorBoundSampler, its SDF queries in one-product f16 (hip_sampler_fast_values)")
    legs[f"sampler_step_{parity}_fast_values"] = leg
    del tr
    # ---- north_star's own sub-metric: "the 8-layer x 256-wide SDF MLP at >= 40 % MFMA peak" = the values-mode fused SDF forward (PE ->
    # lin0..lin8 -> sphere clamp, nothing saved), the launch a sampler round makes for its 1024 x 128 new depths -- five per real step.
    # HIP events of the library around THAT kernel (class 2), 20 launches; flops = the library's algorithmic count for the launch: lin0..lin7
    # in full + the ONE row of lin8 a query needs (459 264 MAC = 0.9185 MFLOP per point; SURVEY 8(d)'s 1.0491 MFLOP includes the 256
    # feature rows of lin8, which only the main pass computes).
    from neat_amd import _lib, ops
    lib = _lib.lib()
    for prec in dict.fromkeys((headline_precision, parity)):
        tr = Trainer(device=dev, state_dict=sd)
        tr.model.set_precision(prec).eval()
        net = tr.model.implicit_network
        pts, NL = R_RAYS * S_SAMPLES, 20
        with torch.no_grad():
            dirs, cam = tr.model._rays(inp)
            zq = torch.tensor(synth.synth_z_vals(42, R_RAYS, S_SAMPLES)).to(dev)
            handle = net.handle()
            ws, ldp = ops.sdf_query_workspace(handle, pts, dev)
            # the query points, laid out once by the sampler's prologue launch (as in a real step); then NL launches of the values kernel
            # alone, captured in one HIP graph and replayed between two events: kernel time + the ~1.5 us a graph node costs.  (HIP
            # events around EAGER launches of this kernel read 25-30 us high: 171-179 us against 146 in the graph and in the tracer.)
            ops.sampler_init_rays(zq, tr.model.density.beta, tr.model.density.beta_min, 1.0, 1, cam.contiguous(), dirs.contiguous(), ws, ldp, None, 0, 0, 0)
            run = lambda: ops.sdf_values_laid_out(handle, ws, pts, net.sdf_bounding_sphere, net.sphere_scale)
            for _ in range(3):
                run()
            lib.neat_prof_enable(1)                  # (one eager launch: the library's algorithmic flop count of this launch)
            run()
            torch.cuda.synchronize()
            ms, fl, n, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
            _lib.check(lib.neat_prof_collect(2, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n), ctypes.byref(by)), "neat_prof_collect")
            lib.neat_prof_enable(0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(NL):
                    run()
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (5 * NL)
        flop_pt = fl.value / max(n.value, 1) / pts
        tf = flop_pt * pts / (us * 1e-6) / 1e12
        note(f"secondary leg sdf_mlp_forward_{prec}: {us:.1f} us per {pts} points")
        legs[f"sdf_mlp_forward_{prec}"] = {
            "us_per_launch": us, "points": pts, "launches": 5 * NL, "value": pts / (us * 1e-6), "unit": "SDF-MLP evaluations/s",
            "tflops": tf, "peak_tflops": PEAK_TFLOPS[prec], "frac": tf / PEAK_TFLOPS[prec], "flop_per_point": flop_pt, "precision": prec,
            "frac_at_full_forward_flops": 1.0491e6 * pts / (us * 1e-6) / 1e12 / PEAK_TFLOPS[prec],
            "products_per_mac": 3 if prec == "fp16x3" else 1,
            "workload": "values-mode fused SDF forward (PE-6 -> 8 x 256 softplus MLP -> sdf, sphere clamp) on 1024 rays x 128 depths = one sampler round's query; "
                        f"{NL} launches captured in one HIP graph, replayed 5x between two events; frac = the launch's algorithmic flops (lin0..lin7 + the one row of "
                        "lin8 a query needs: 0.918 MFLOP per point, 1 product per MAC) / dense 16-bit MFMA peak; frac_at_full_forward_flops prices the same "
                        "time with SURVEY 8(d)'s 1.0491 MFLOP (incl. lin8's 256 feature rows, which this launch does not compute)"
                        + (" -- the 3-product f16 build issues 3 MFMAs per algorithmic MAC" if prec == "fp16x3" else "")}
        del tr, g
    # ---- the other BASELINE configs under the same clock; single GPU, HIP-graph replay, synthetic rays
    sd_dtu = {k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough", num_junctions=1024).items()}
    for label, rays, precs, what, parity_note in (
            ("c3_dtu_2048x128", 2048, dict.fromkeys((headline_precision, parity)),
             "C3: DTU model switches (device DBSCAN, use_median off, 1024 junction latents), 2048 rays x 128 given samples, full losses (RGB + eikonal + attraction + junctions)",
             "G11 (the reference's train step with these switches) at the fp32 bars for fp16x3; full-size oracle comparison test_c3_full_size_dtu_step"),
            ("c4_rank_shape_512x128", 512, dict.fromkeys((headline_precision, parity)),
             "C4's per-rank shape: 512 rays x 128 given samples (4096 rays over 8 ranks), DTU switches, world size 1 here (no all-reduce)",
             "test_c4_rank_shape_step_vs_oracle")):
        for prec in precs:
            torch.manual_seed(42)
            tr = Trainer(model_conf=dtu_switches_conf(), device=dev, state_dict=sd_dtu)
            tr.model.set_precision(prec)
            _, inp_w, gt_w = synthetic_batch(42, rays, dev)
            tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, rays, S_SAMPLES)).to(dev)
            leg = timed(tr, inp_w, gt_w, f"{label}_{prec}", rays=rays)
            leg.update(precision=prec, workload=what, parity=parity_note)
            legs[f"{label}_{prec}"] = leg
            del tr
    # C5: hierarchical 64 coarse + 64 fine depths, "fp16 MFMA with fp32 accumulate" (BASELINE configs[4]); per-rank shape 1024 rays
    import copy
    conf5 = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
    conf5.update(hip_sampler="hierarchical", hip_sampler_coarse=64, hip_sampler_fine=64)
    for prec in ("fp16", parity):
        torch.manual_seed(42)
        tr = Trainer(model_conf=conf5, device=dev, state_dict=sd)
        tr.model.set_precision(prec)
        leg = timed(tr, inp, gt, f"c5_hierarchical_64+64_{prec}")
        leg.update(precision=prec, workload="C5: 1024 rays, hierarchical sampler (64 coarse -> SDF values -> weights -> 64 fine = 128 samples per ray) feeding the main pass, abc model",
                   parity="G12 (the reference's hierarchical train step): f16-grade bars for fp16 (test_c5_fp16_train_step_vs_reference_golden), fp32 bars for fp16x3; "
                          "full size: test_c5_full_size_hierarchical_step")
        legs[f"c5_hierarchical_64+64_{prec}"] = leg
        del tr
    # eval chunk as neat-final-parsing.py drives the model (:203-218): forward only, 2048 rays, conf-default sampler in eval mode
    for prec in dict.fromkeys((headline_precision, parity)):
        tr = Trainer(device=dev, state_dict=sd)
        tr.model.set_precision(prec)
        tr.model.eval()
        _, inp_e, _ = synthetic_batch(43, 2048, dev)
        with torch.no_grad():
            for _ in range(3):
                o = tr.model(inp_e)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                o = tr.model(inp_e)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        S_e = o["points"].shape[0] // 2048 if o["points"].dim() == 2 else o["points"].shape[1]
        note(f"secondary leg eval_chunk_2048_{prec}: {1e3 * dt:.3f} ms/chunk")
        legs[f"eval_chunk_2048_{prec}"] = {"ms_per_step": 1e3 * dt, "samples_per_ray": int(S_e), "value": 2048 * S_e / dt, "unit": "ray-samples/s",
                                           "rays_per_s": 2048 / dt, "launch": "eager (host decides the sampler's rounds, as the reference)", "steps": steps,
                                           "precision": prec, "workload": "eval forward of one 2048-ray chunk (sampler + render + junction block), neat-final-parsing.py:203-218",
                                           "parity": "G7 (the reference's eval forward, all keys): test_eval_chunks_like_final_parsing"}
        del tr
    return legs


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """No GPU: rendezvous (gloo) + the data-parallel exchange of the step (ONE flat all-reduce of all 1 219 274 gradients)
    on rank-dependent values, checked against the closed form.  Exercises launcher, env handling and dp.py only."""
    from neat_amd import dp, networks, synth
    import torch.distributed as dist
    rank, world, _ = dp.init_from_env(backend="gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    c4 = args.workload == "c4"
    model = networks.VolSDFNetwork(dtu_switches_conf() if c4 else synth.ABC_NEAT_A_MODEL_CONF)
    rays = dp.shard_rays(C4_GLOBAL_RAYS, world) if c4 else R_RAYS
    params = [p for p in model.parameters() if p.requires_grad]
    n = sum(p.numel() for p in params)
    flat = torch.full((n,), float(rank + 1))
    bucket = dp.FlatGradBucket(params)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        flat.fill_(float(rank + 1))
        bucket.all_reduce_mean(flat)
        # the packed path of a replayed step (Trainer._finish_step): gradients -> flat bucket -> ONE collective -> .grad = views
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (1 + (i % 3)))
        bucket.pack()
        if bucket.active():
            bucket.reduce_packed()
        packed_ok = all(torch.allclose(p.grad, torch.full_like(p, (world + 1) / 2.0 * (1 + (i % 3)))) for i, p in enumerate(params)) \
            if bucket.active() else True
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    expect = (world + 1) / 2.0
    ok = bool(torch.allclose(flat, torch.full_like(flat, expect))) and (packed_ok if args.steps else True)
    if rank == 0:
        print(json.dumps({"metric": "ray-samples/s (train step) on ABC-neat-a", "value": 0.0, "unit": "ray-samples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
                          "higher_is_better": True, "scaling": "strong" if c4 else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "dry_run": True, "backend": "gloo", "world_size": world, "allreduce_elements": n, "allreduce_ok": ok,
                          "config": {"workload": "DRY RUN (no GPU): launcher + gloo rendezvous + flat gradient all-reduce only",
                                     "rays_per_gpu": rays, "global_rays": world * rays, "samples_per_ray": S_SAMPLES, "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("dry run: all-reduce mean is wrong")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (parity-grade precision, sampler step)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-graph", action="store_true",
                    help="run the timed steps eagerly (default: forward+loss+backward of the step replayed from a HIP graph)")
    ap.add_argument("--pt", type=int, default=0, help="(tuning) bf16 layer-kernel point tile: 2 = 64 points, 4 = 128 points")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: launcher + gloo rendezvous + gradient all-reduce only (CPU test)")
    ap.add_argument("--workload", choices=["c2", "c4"], default="c2",
                    help="c2 (default, BASELINE configs[1]): 1024 rays x 128 samples per GPU, weak scaling; c4 (configs[3]): 4096 rays per step "
                         "in total split over the ranks, DTU model switches, strong scaling")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16x3", "fp16", "fp16x3"], default=DEFAULT_PRECISION,
                    help="GEMM build: fp32 = exact-f32 MFMA (parity build); bf16 = bf16 MFMA, fp32 accumulate (BASELINE config 2)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn_under_torchrun(args.gpus))
    if args.dry_run:
        return dry_run(args)

    from neat_amd import _lib, dp, synth
    from neat_amd.train import Trainer, synthetic_batch
    import torch.distributed as dist

    # stdout carries ONE JSON line.  Native libraries write there too (RCCL prints its version banner through C stdio when the first
    # communicator is made, flushed whenever): everything but that line goes to stderr -- fd 1 is pointed at fd 2 for the run, the line is
    # written to a duplicate of the original stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    t_start = time.perf_counter()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # NEAT_DIST_BACKEND=gloo: a functional check of the multi-process path on fewer GPUs than ranks (ranks share devices, the gradient
    # all-reduce goes through the host).  Never a measurement: RCCL needs one GPU per rank and is the default.
    backend = os.environ.get("NEAT_DIST_BACKEND") or None
    ndev = torch.cuda.device_count()
    want_local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend != "gloo" and want_local >= ndev:
        raise SystemExit(f"rank {want_local} of this node has no GPU of its own ({ndev} visible): RCCL needs one GPU per rank")
    torch.cuda.set_device(want_local % ndev)
    rank, world, local = dp.init_from_env(backend=backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _lib.lib()
    if args.pt:
        _lib.check(lib.neat_set_tuning(0, args.pt), "neat_set_tuning")

    tr, inp, gt, rays, workload_text, scaling, sd = make_workload(args.workload, rank, world, dev, args.precision)
    peak = PEAK_TFLOPS[args.precision]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.2f}s] {msg}", file=sys.stderr, flush=True)

    note("model built, starting warm-up")
    for i in range(args.warmup):
        tr.step(inp, gt)
        torch.cuda.synchronize()
        note(f"warm-up step {i} done")
    graphed = False
    if not args.no_graph:
        graphed = tr.capture(inp, gt)          # untimed: 2 more warm-up steps, the capture, one replayed step
        note("HIP graph captured" if graphed else f"graph capture failed, staying eager: {tr.capture_error!r}")
    graph_note = ""
    if world > 1 and not args.no_graph:
        # every rank must take the same path from here on (the eager steps of the roofline pass below issue collectives):
        # a capture that failed on one rank sends all ranks to eager steps.  (Trainer.capture takes warmup + 1 optimizer steps --
        # gradient all-reduces -- whether it succeeds or fails, so the ranks arrive here after the same number of collectives.)
        ok_all = torch.tensor([1.0 if graphed else 0.0], device=dev)
        dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
        if graphed and float(ok_all.item()) == 0.0:
            tr._graphs.clear()
            graphed = False
            graph_note = " (graph capture failed on another rank)"
            note("graph capture failed on another rank: staying eager")
    if graphed and world > 1:
        # insurance for multi-process runs (untimed, 6 steps): if replaying the graph measures slower than stepping eagerly on this
        # node, every rank steps eagerly.  One GPU per rank: the graph wins (3.4 vs 5.5 ms) and stays.  (Not a cure for the
        # functional gloo check with several ranks on ONE device: there, once both processes hold an instantiated graph, replayed AND
        # eager steps take 0.3-1.9 s -- the device's queue scheduler thrashes between the processes; without graphs 9.6 ms.)
        def timed(fn, n=3):
            barrier()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        t_graph = timed(lambda: tr.step(inp, gt))
        t_eager = timed(lambda: tr.step_eager(inp, gt))
        slow = torch.tensor([1.0 if t_graph > 1.5 * t_eager else 0.0], device=dev)
        dist.all_reduce(slow, op=dist.ReduceOp.MAX)
        if float(slow.item()) > 0.0:
            tr._graphs.clear()
            graphed = False
            graph_note = f" (graph replay measured slower than eager here: {1e3 * t_graph:.1f} vs {1e3 * t_eager:.1f} ms per step on rank 0's clock)"
            note("graph replay slower than eager steps on this box: staying eager" + graph_note)
    barrier()
    if not args.no_prof and not graphed:
        lib.neat_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, losses = tr.step(inp, gt)
    barrier()
    elapsed = time.perf_counter() - t0
    note(f"timed region done: {elapsed:.3f}s for {args.steps} steps")
    dist_info = None
    if dist.is_initialized():
        # what a bad scaling curve would be diagnosed from, in the same line: every rank's own time per step (the headline uses the
        # MAX), how long each rank's steps take WITHOUT the exchange being waited on (its GPU work), and the gradient all-reduce alone
        # (the flat 4.9 MB bucket, 20 back-to-back calls between HIP events, after the timed region)
        every_t = torch.zeros(world, device=dev, dtype=torch.float64)      # (as an all-reduce of one-hot rows: gloo has no all_gather for device tensors)
        every_t[rank] = elapsed
        dist.all_reduce(every_t, op=dist.ReduceOp.SUM)
        every = list(every_t)
        tr.bucket._ensure()
        flat = tr.bucket.flat
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        e1.record()
        torch.cuda.synchronize()
        ar = torch.tensor([e0.elapsed_time(e1) / 20.0], device=dev, dtype=torch.float64)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        flat.zero_()
        # host time of a step: what the Python side needs to enqueue one step (batch prefix, graph replay, all-reduce, Adam), with
        # nothing waited on -- as long as it stays below the GPU's time per step the ranks are never host-bound
        # (3 steps at a time from an idle device: the staging ring of the CPU-drawn randoms is 4 deep, a 5th enqueue would wait for the GPU)
        host_s = 0.0
        for _ in range(3):
            barrier()
            th = time.perf_counter()
            for _ in range(3):
                tr.step(inp, gt)
            host_s += time.perf_counter() - th
        host_ms = 1e3 * host_s / 9.0
        barrier()
        dist_info = {"backend": dist.get_backend(), "world": world, "ranks": world, "host_ms_per_step": host_ms,
                     "step_sequence": (("ONE graph launch: forward + loss + backward + gradient pack + all-reduce + Adam on the flat buffer"
                                        if getattr(tr._last, "adam_has", None) is not None else
                                        "graph replay (ends with the gradient pack) -> all-reduce -> Adam on the flat buffer, one stream")
                                       if graphed and getattr(tr._last, "packed", False) else "eager step + pack + all-reduce + Adam"),
                     "ms_per_step_by_rank": [1e3 * float(t.item()) / args.steps for t in every],
                     "allreduce_ms": float(ar.item()), "allreduce_bytes": int(flat.numel() * 4),
                     "allreduce_note": "flat gradient bucket, mean of 20 back-to-back calls after the timed region, max over ranks"}
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if graphed:
        tr.check_nan()
    prof_elapsed, prof_note, n_prof = elapsed, "HIP events over the timed region", args.steps
    if not args.no_prof and graphed:
        # a graph replay does not pass through the library's launch code, so its kernels cannot be bracketed with events:
        # the same steps are run eagerly right after the timed region (same kernels, same arguments) for the roofline
        n_prof = min(args.steps, 10)
        torch.cuda.synchronize()
        lib.neat_prof_enable(1)
        tp = time.perf_counter()
        for _ in range(n_prof):
            tr.step_eager(inp, gt)
        torch.cuda.synchronize()
        prof_elapsed = time.perf_counter() - tp
        prof_note = f"HIP events over {n_prof} eager steps run right after the timed region (the timed steps replay a HIP graph)"

    roofline = None
    kernels = {}
    if not args.no_prof:
        # launch classes of the library's event brackets: 0 per-layer GEMM launches, 1 weight gradients, 2 fused SDF primal chain,
        # 3 fused SDF adjoint chain (the normals), 4 the heads' fused chains (forward and backward)
        for cls, name in ((0, "layer_kernel"), (1, "wgrad_kernel"), (2, "sdf_fused_kernel"), (3, "sdf_adjoint_kernel"), (4, "head_chain_kernel")):
            ms, fl, n, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
            _lib.check(lib.neat_prof_collect(cls, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n), ctypes.byref(by)),
                       "neat_prof_collect")
            if n.value:
                sec = ms.value * 1e-3
                kernels[name] = {"launches": n.value, "total_ms": ms.value, "avg_us": 1e3 * ms.value / n.value,
                                 "tflops": fl.value / sec / 1e12, "flop_per_launch": fl.value / n.value,
                                 "gbytes_per_s": by.value / sec / 1e9, "bytes_per_launch": by.value / n.value,
                                 "flop_per_byte": fl.value / max(by.value, 1.0)}
        lib.neat_prof_enable(0)
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
            k = kernels[dom]
            # HBM bytes per launch of the dominant kernel from the PMC counters: needs rocprofv3 passes around the process, so it
            # cannot be measured in this run; the figure is the committed one of the same command (scripts/refresh_profiles.sh ->
            # scripts/make_traffic.py -> profiles/traffic.json) and is labelled as such
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath)).get(args.precision, {})
                traffic = tj.get(dom, {}).get("hbm_bytes_per_launch")
                if traffic is None and dom == "sdf_fused_kernel":
                    traffic = tj.get("sdf_chain_x3_kernel", {}).get("hbm_bytes_per_launch")
                if traffic is None and dom == "sdf_adjoint_kernel":
                    traffic = tj.get("sdf_adjoint_x3_kernel", {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not measured in this run)"
            # Which roof the dominant class sits on is decided by ITS byte mix, not by SURVEY 8(d)'s hope for a fused path (~5 B per
            # ray-sample): below the ridge (peak flop/s / peak B/s = 312 flop/B for bf16) a launch cannot reach the MFMA roof however
            # well it issues, and `achieved` / `peak` / `frac` are then the HBM figures.  Both fractions are always carried
            # (`frac_mfma`, `frac_hbm`), plus the whole step's fraction of the MFMA peak and the counter-derived MFMA utilisation.
            steps_prof = n_prof if graphed else args.steps
            step_bytes = sum(v["bytes_per_launch"] * v["launches"] for v in kernels.values()) / max(steps_prof, 1)
            ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
            frac_mfma, frac_hbm = k["tflops"] / peak, k["gbytes_per_s"] / PEAK_HBM_GBS
            for v in kernels.values():
                v["bound"] = "hbm" if v["flop_per_byte"] < ridge else "mfma"
                v["frac_mfma"], v["frac_hbm"] = v["tflops"] / peak, v["gbytes_per_s"] / PEAK_HBM_GBS
            # MFMA utilisation from the SQ counters (SQ_VALU_MFMA_BUSY_CYCLES / (launch time x 2.4 GHz x 1024 SIMDs)): like `traffic` a
            # committed figure of the same command (scripts/pmc_sq.sh -> scripts/make_mfma_util.py -> profiles/mfma_util.json)
            mfma_util, upath = None, os.path.join(ROOT, "profiles", "mfma_util.json")
            if os.path.exists(upath):
                uj = json.load(open(upath)).get(args.precision, {}).get("classes", {})
                mfma_util = {c: round(v["mfma_util"], 4) for c, v in uj.items()} or None
            hbm_bound = k["bound"] == "hbm"
            roofline = {"bound": k["bound"], "kernel": dom,
                        "achieved": k["gbytes_per_s"] if hbm_bound else k["tflops"], "peak": PEAK_HBM_GBS if hbm_bound else peak,
                        "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": frac_hbm if hbm_bound else frac_mfma,
                        "frac_mfma": frac_mfma, "frac_hbm": frac_hbm, "achieved_tflops": k["tflops"], "achieved_gbytes_per_s": k["gbytes_per_s"],
                        "ridge_flop_per_byte": ridge,
                        "bound_note": ("the dominant class runs at %.0f flop/B against a ridge of %.0f: HBM-bound by its byte mix (chain variables "
                                       "stream through HBM once per layer); the guide's achievable HBM rate is ~6.3 of the 8 TB/s spec"
                                       % (k["flop_per_byte"], ridge)) if hbm_bound else "above the ridge: priced against the dense MFMA peak",
                        "step_frac_of_mfma_peak": rays * S_SAMPLES * args.steps / elapsed * FLOP_PER_RAY_SAMPLE / 1e12 / peak,
                        "mfma_util_from_counters": mfma_util,
                        "mfma_util_source": "profiles/mfma_util.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass of this command, committed; "
                                            "busy cycles / (launch duration x 2.4 GHz x 1024 SIMDs), not measured in this run)" if mfma_util else None,
                        "traffic": traffic, "traffic_source": traffic_source,
                        # PMC bytes against the library's own algorithmic bytes of the same launches: wasted re-reads show as > 1, a stale
                        # traffic.json as a jump (scripts/check_traffic.py fails the profile refresh beyond +-10 %)
                        "traffic_vs_algorithmic_bytes": (traffic / k["bytes_per_launch"]) if traffic else None,
                        # the committed PMC figure no longer describes the launches of THIS run when it leaves the library's live byte
                        # accounting by more than 15 % (a kernel changed and profiles/traffic.json was not regenerated)
                        "traffic_stale": (abs(traffic / k["bytes_per_launch"] - 1.0) > 0.15) if traffic else None,
                        "avg_launch_us": k["avg_us"], "launches": k["launches"], "flop_per_launch": k["flop_per_launch"],
                        "flop_per_byte": k["flop_per_byte"],
                        "hbm": {"achieved_gbytes_per_s": k["gbytes_per_s"], "peak_gbytes_per_s": PEAK_HBM_GBS,
                                "frac": k["gbytes_per_s"] / PEAK_HBM_GBS, "bytes_per_launch": k["bytes_per_launch"],
                                "gemm_class_bytes_per_step": step_bytes,
                                "note": "bytes the per-layer launches move by construction (operands once, weights once); the fused path's "
                                        "algorithmic minimum is ~5 B per ray-sample (SURVEY 8d), a save-once/read-once backward ~4.3 GB per step"},
                        "kernel_time_share": k["total_ms"] * 1e-3 / prof_elapsed, "measured": prof_note,
                        "all_kernels": kernels}

    if rank == 0:
        samples = world * rays * S_SAMPLES * args.steps
        value = samples / elapsed
        line = {
            "metric": "ray-samples/s (train step) on ABC-neat-a", "value": value, "unit": "ray-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": {"fp32": "f32", "fp16": "f16", "fp16x3": "f16x3 forward / f16 backward"}.get(args.precision, "bf16"), "data": "synthetic",
            "config": {"workload": workload_text + ", train step = forward + loss + backward + Adam" +
                                   ((" + RCCL grad all-reduce" if dist.get_backend() == "nccl" else " + gloo grad all-reduce (functional check)") if dist.is_initialized() else ""),
                       "rays_per_gpu": rays, "samples_per_ray": S_SAMPLES, "global_rays": world * rays,
                       "parallelism": f"dp{world}", "weights": "synthetic 'rough' (seed 42)",
                       "launch": (("hip graph replay, the whole step one launch (forward + loss + backward" + (" + gradient pack + all-reduce" if getattr(tr._last, "packed", False) else "") + " + Adam)")
                                  if getattr(tr._last, "adam_has", None) is not None else "hip graph replay (forward+loss+backward) + eager all-reduce/Adam") if graphed else "eager" + graph_note,
                       "dist_backend": (dist.get_backend() if dist.is_initialized() else None)},
            "rays_per_s": world * rays * args.steps / elapsed,
            "step_tflops": value * FLOP_PER_RAY_SAMPLE / 1e12,
            "step_frac_of_mfma_peak": value * FLOP_PER_RAY_SAMPLE / 1e12 / (peak * world),
            "loss": float(losses["loss"].detach()),
            "roofline": roofline,
            "dist": dist_info,
        }
        if _lib.tuning_overrides:
            line["config"]["tuning_overrides"] = list(_lib.tuning_overrides)      # NEAT_TUNING was set: not the default kernels
        if world == 1 and not args.no_secondary and args.workload == "c2":
            line["secondary"] = secondary_legs(dev, sd, args.precision, args.steps, note)
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU oracle (cpu_baseline)")
            line["cpu_baseline"] = cpu_baseline(42)
            note("cpu_baseline done")
        else:
            line["cpu_baseline"] = None
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if dist.is_initialized():          # (world > 1, or the forced single-rank RCCL group of NEAT_FORCE_DIST=1)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
