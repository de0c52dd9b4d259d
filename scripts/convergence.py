"""Do the fast builds LEARN A REAL SCENE like the fp32 build?  (VERDICT r3 "next" #6)
Eight down-sampled views of ABC 00075213 (tests/golden/scene_abc_00075213_8views.npz, the scene the reference ships), the reference's
abc-neat-a model / loss configuration, `neat_amd.runner` (ErrorBoundSampler on, device-assembled batches, per-view HIP graphs) for N
iterations per precision from the same geometric initialisation and the same random streams.  Prints the mean loss and rgb PSNR over
sixths of the run and the differences of the last sixth against the first precision listed.
    python scripts/convergence.py [iterations] [rays] [precisions, comma separated; "fp32:43" = fp32 with seed 43 instead of 42]"""
import os, sys, tempfile, pathlib, random, json, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neat_amd import synth
from neat_amd.runner import TrainRunner
from neat_amd.synth import hocon_text as _hocon


def run(prec, iters, rays, tmp, res, views):
    seed = 42
    if ":" in prec:                       # "fp32:43" = the same build with other random streams: the spread that is NOT precision
        prec, seed = prec.split(":")[0], int(prec.split(":")[1])
    torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)
    model = dict(synth.ABC_NEAT_A_MODEL_CONF)
    model["hip_precision"] = prec
    conf = {"train": {"expname": "conv", "dataset_class": "datasets.blender_hawp_dataset.BlenderDataset",
                      "model_class": "model.networks.neat_wfr_rend_a.VolSDFNetwork", "loss_class": "model.networks.loss_wfr.VolSDFLoss",
                      "learning_rate": 5.0e-4, "num_pixels": rays, "checkpoint_freq": 10 ** 6},
            "loss": dict(synth.ABC_NEAT_A_LOSS_CONF), "dataset": {"data_dir": "abc/00075213", "img_res": [res, res], "reverse_coordinate": True},
            "model": model}
    path = tmp / f"{prec}_{seed}.conf"
    path.write_text(_hocon(conf))
    r = TrainRunner(str(path), nepochs=iters // views - 1, exps_folder=str(tmp / "exps"), data_root=str(tmp / "data"), log_freq=1)
    r.save_checkpoints = lambda epoch: None
    t0 = time.perf_counter()
    hist = r.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = np.array([h[2] for h in hist]); psnr = np.array([h[3] for h in hist])
    return loss, psnr, dt, r.trainer.replays, r.trainer.eager_steps


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
    rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    precs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fp32", "fp16x3", "bf16"]
    tmp = pathlib.Path(tempfile.mkdtemp())
    res = synth.write_scene_fixture(os.path.join(ROOT, "tests", "golden", "scene_abc_00075213_8views.npz"), str(tmp / "data" / "abc" / "00075213"))
    views = 8
    rows = {}
    for prec in precs:
        loss, psnr, dt, replays, eager = run(prec, iters, rays, tmp, res, views)
        assert np.isfinite(loss).all(), prec
        w = len(loss) // 6
        rows[prec] = (loss, psnr)
        print(f"{prec:7s} loss per sixth: " + " ".join(f"{loss[i * w:(i + 1) * w].mean():.4f}" for i in range(6)) +
              "   psnr: " + " ".join(f"{psnr[i * w:(i + 1) * w].mean():.2f}" for i in range(6)) +
              f"   ({len(loss)} iterations in {dt:.1f} s, replays {replays}, eager {eager})", flush=True)
    ref = rows[precs[0]]
    w = len(ref[0]) // 6
    out = {"iterations": int(len(ref[0])), "rays": rays, "views": views, "resolution": res, "reference_precision": precs[0], "builds": {}}
    for prec in precs:
        l, p = rows[prec]
        out["builds"][prec] = {"loss_last_sixth": float(l[-w:].mean()), "psnr_last_sixth": float(p[-w:].mean()), "psnr_first_sixth": float(p[:w].mean()),
                               "loss_rel_diff": float(abs(l[-w:].mean() - ref[0][-w:].mean()) / ref[0][-w:].mean()),
                               "psnr_diff_db": float(p[-w:].mean() - ref[1][-w:].mean())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
